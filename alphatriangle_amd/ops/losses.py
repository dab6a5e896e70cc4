"""AlphaZero training loss with a C51 distributional value head.

``alphazero_loss`` is the single entry point used by the trainer.  On a GPU with
the kernel library present it runs the fused HIP forward+backward kernel
(csrc/hip/learner_kernels.hip: two-hot projection, both cross-entropies, entropy,
importance weighting and the logits gradients in one pass); everywhere else it
runs the PyTorch composition below, which is also the numerical reference of the
kernel test.  Loss definition: alphatriangle/rl/core/trainer.py:159-271.
"""

from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class LossOutputs:
    total: torch.Tensor        # scalar, differentiable
    policy_loss: torch.Tensor  # scalar (detached)
    value_loss: torch.Tensor   # scalar (detached)
    entropy: torch.Tensor      # scalar mean policy entropy (detached)
    td_errors: torch.Tensor    # [B] per-sample value cross-entropy (detached)


def c51_two_hot(returns: torch.Tensor, v_min: float, v_max: float, num_atoms: int) -> torch.Tensor:
    """Project scalar returns onto the fixed support: mass ``u-b`` on floor, ``b-l`` on ceil."""
    delta_z = (v_max - v_min) / (num_atoms - 1)
    b = (returns.clamp(v_min, v_max) - v_min) / delta_z
    lo = b.floor().long().clamp(0, num_atoms - 1)
    hi = b.ceil().long().clamp(0, num_atoms - 1)
    m = torch.zeros(returns.shape[0], num_atoms, dtype=torch.float32, device=returns.device)
    same = lo == hi
    w_lo = torch.where(same, torch.ones_like(b), hi.float() - b)
    w_hi = torch.where(same, torch.zeros_like(b), b - lo.float())
    m.scatter_add_(1, lo.unsqueeze(1), w_lo.unsqueeze(1))
    m.scatter_add_(1, hi.unsqueeze(1), w_hi.unsqueeze(1))
    return m


def alphazero_loss_reference(
    policy_logits: torch.Tensor,
    value_logits: torch.Tensor,
    policy_target: torch.Tensor,
    returns: torch.Tensor,
    is_weights: torch.Tensor,
    *,
    v_min: float,
    v_max: float,
    policy_weight: float,
    value_weight: float,
    entropy_weight: float,
) -> LossOutputs:
    num_atoms = value_logits.shape[1]
    policy_logits = policy_logits.float()
    value_logits = value_logits.float()
    with torch.no_grad():
        target_dist = c51_two_hot(returns.float(), v_min, v_max, num_atoms)
    value_ce = -(target_dist * F.log_softmax(value_logits, dim=1)).sum(dim=1)
    value_loss = (value_ce * is_weights).mean()
    log_probs = F.log_softmax(policy_logits, dim=1)
    policy_ce = -(torch.nan_to_num(policy_target.float(), nan=0.0) * log_probs).sum(dim=1)
    policy_loss = (policy_ce * is_weights).mean()
    entropy_mean = torch.zeros((), device=policy_logits.device)
    entropy_term = torch.zeros((), device=policy_logits.device)
    if entropy_weight > 0:
        probs = F.softmax(policy_logits, dim=1)
        entropy = -(probs * torch.log(probs + 1e-9)).sum(dim=1)
        entropy_mean = entropy.mean()
        entropy_term = -entropy_weight * entropy_mean
    total = policy_weight * policy_loss + value_weight * value_loss + entropy_term
    return LossOutputs(total, policy_loss.detach(), value_loss.detach(), entropy_mean.detach(), value_ce.detach())


def alphazero_loss(
    policy_logits: torch.Tensor,
    value_logits: torch.Tensor,
    policy_target: torch.Tensor,
    returns: torch.Tensor,
    is_weights: torch.Tensor,
    *,
    v_min: float,
    v_max: float,
    policy_weight: float,
    value_weight: float,
    entropy_weight: float,
    force_reference: bool = False,
) -> LossOutputs:
    if policy_logits.is_cuda and not force_reference:
        from . import hip_learner

        if hip_learner.available():
            return hip_learner.fused_loss(
                policy_logits,
                value_logits,
                policy_target,
                returns,
                is_weights,
                v_min=v_min,
                v_max=v_max,
                policy_weight=policy_weight,
                value_weight=value_weight,
                entropy_weight=entropy_weight,
            )
    return alphazero_loss_reference(
        policy_logits,
        value_logits,
        policy_target,
        returns,
        is_weights,
        v_min=v_min,
        v_max=v_max,
        policy_weight=policy_weight,
        value_weight=value_weight,
        entropy_weight=entropy_weight,
    )
