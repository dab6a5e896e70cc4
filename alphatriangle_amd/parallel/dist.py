"""Process-group runtime: one process per GPU, RCCL over xGMI (``nccl`` backend on ROCm).

The reference has no collective communication at all - its only transport is Ray's
object store (alphatriangle/training/worker_manager.py:44-45,179-194 broadcast
weights by ``ray.put`` + N remote calls; :117-134 gather episode results).  The
equivalents here:

* weights learner -> actors : ONE ``broadcast`` of the flat fp32 parameter buffer
  (12 MB for the default net), optionally on a side stream into a staging buffer so
  self-play never waits for the wire;
* learner gradients (data parallel) : ONE ``all_reduce`` of the flat gradient buffer
  between backward and the fused optimizer step.  xGMI is point-to-point, so a
  message this small is latency-bound; it is deliberately a single un-bucketed call;
* experiences actors -> learner : ``all_gather`` of fixed-stride SoA record blocks
  (parallel/experience.py).

``gloo`` (CPU tests, rehearsals) only ever sees CPU tensors: device tensors are
staged through host memory with synchronous copies.
"""

from __future__ import annotations

import datetime
import logging
import os
from dataclasses import dataclass, field
from typing import Any

import torch
import torch.distributed as dist

logger = logging.getLogger(__name__)


@dataclass
class RoleLayout:
    """Which ranks learn and which ranks self-play."""

    world_size: int
    learner_ranks: list[int]
    actor_ranks: list[int]
    colocated: bool

    @classmethod
    def build(cls, world_size: int, layout: str, num_learners: int) -> "RoleLayout":
        if layout == "colocated" or world_size == 1:
            ranks = list(range(world_size))
            return cls(world_size, ranks, ranks, True)
        n_learn = max(1, min(num_learners, world_size - 1))
        return cls(world_size, list(range(n_learn)), list(range(n_learn, world_size)), False)

    def is_learner(self, rank: int) -> bool:
        return rank in self.learner_ranks

    def is_actor(self, rank: int) -> bool:
        return rank in self.actor_ranks


@dataclass
class DistContext:
    rank: int
    world_size: int
    local_rank: int
    backend: str
    device: torch.device
    layout: RoleLayout
    learner_group: Any = None
    world_group: Any = None
    cohort_group: Any = None          # split layout: one learner + the actors that feed it
    cohort_ranks: list[int] = field(default_factory=list)
    owns_process_group: bool = True
    _side_stream: Any = field(default=None, repr=False)
    _pending_weight_work: Any = field(default=None, repr=False)
    _staging: torch.Tensor | None = field(default=None, repr=False)

    # ------------------------------------------------------------------ helpers
    @property
    def is_learner(self) -> bool:
        return self.layout.is_learner(self.rank)

    @property
    def is_actor(self) -> bool:
        return self.layout.is_actor(self.rank)

    @property
    def num_learners(self) -> int:
        return len(self.layout.learner_ranks)

    def _host_staged(self) -> bool:
        return self.backend == "gloo"

    def _collective(self, fn, tensor: torch.Tensor, **kw) -> None:  # noqa: ANN001
        """Run an in-place collective; with gloo, device tensors round-trip through host memory."""
        if self._host_staged() and tensor.is_cuda:
            host = tensor.detach().to("cpu", non_blocking=False)
            fn(host, **kw)
            tensor.copy_(host, non_blocking=False)
        else:
            fn(tensor, **kw)

    # ------------------------------------------------------------------ collectives
    def grad_sync(self, flat_grad: torch.Tensor) -> None:
        """Sum the flat gradient buffer over the learner ranks (the optimizer divides by the count)."""
        if self.num_learners <= 1 or not self.is_learner:
            return
        self._collective(dist.all_reduce, flat_grad, op=dist.ReduceOp.SUM, group=self.learner_group)

    def broadcast_weights(self, flat_param: torch.Tensor, src: int | None = None) -> None:
        """Blocking broadcast of the flat parameter buffer from the first learner to every rank."""
        src_rank = self.layout.learner_ranks[0] if src is None else src
        self._collective(dist.broadcast, flat_param, src=src_rank, group=self.world_group)

    def broadcast_weights_async(self, flat_param: torch.Tensor) -> None:
        """Start a broadcast into a staging buffer on a side stream; ``finish_weight_broadcast`` swaps it in.

        The side stream uses the default priority.  Senders stage a snapshot so the learner may keep
        updating ``flat_param`` while the wire transfer is in flight.
        """
        src_rank = self.layout.learner_ranks[0]
        if self._host_staged() or not flat_param.is_cuda:
            self.broadcast_weights(flat_param, src_rank)
            return
        if self._staging is None or self._staging.numel() != flat_param.numel():
            self._staging = torch.empty_like(flat_param)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        self._side_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._side_stream):
            if self.rank == src_rank:
                self._staging.copy_(flat_param, non_blocking=True)
            self._pending_weight_work = dist.broadcast(self._staging, src=src_rank, group=self.world_group, async_op=True)

    def finish_weight_broadcast(self, flat_param: torch.Tensor) -> bool:
        """Complete a pending async broadcast and copy the staged weights in (non-source ranks)."""
        if self._pending_weight_work is None:
            return False
        self._pending_weight_work.wait()
        self._pending_weight_work = None
        torch.cuda.current_stream(self.device).wait_stream(self._side_stream)
        if self.rank != self.layout.learner_ranks[0]:
            assert self._staging is not None
            flat_param.copy_(self._staging, non_blocking=True)
        return True

    def all_gather_records(self, block: torch.Tensor) -> torch.Tensor:
        """All-gather equally sized record blocks ``[n, stride]``; returns ``[world, n, stride]``."""
        out_shape = (self.world_size, *block.shape)
        if self._host_staged() and block.is_cuda:
            host = block.detach().to("cpu", non_blocking=False).contiguous()
            out = torch.empty(out_shape, dtype=host.dtype)
            dist.all_gather_into_tensor(out.view(-1), host.view(-1), group=self.world_group)
            return out.to(block.device, non_blocking=False)
        out = torch.empty(out_shape, dtype=block.dtype, device=block.device)
        dist.all_gather_into_tensor(out.view(-1), block.contiguous().view(-1), group=self.world_group)
        return out

    def gather_blocks_to_learner(self, block: torch.Tensor) -> list[torch.Tensor] | None:
        """Split layout: gather equally sized blocks from this rank's cohort onto its learner.

        Returns the per-cohort-member list (cohort order, the learner's own block first) on the
        learner and ``None`` on actors.  Point-to-point under the hood: only the learner receives,
        so actor GPUs spend no xGMI bandwidth on each other's experiences.
        """
        if self.layout.colocated or self.cohort_group is None:
            raise RuntimeError("gather_blocks_to_learner needs the split role layout")
        dst = self.cohort_ranks[0]
        staged = self._host_staged() and block.is_cuda
        send = block.detach().to("cpu", non_blocking=False).contiguous() if staged else block.contiguous()
        if self.rank == dst:
            outs = [torch.empty_like(send) for _ in self.cohort_ranks]
            dist.gather(send, gather_list=outs, dst=dst, group=self.cohort_group)
            return [o.to(block.device, non_blocking=False) for o in outs] if staged else outs
        dist.gather(send, gather_list=None, dst=dst, group=self.cohort_group)
        return None

    def all_reduce_max_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place MAX all-reduce of a small control tensor over every rank."""
        self._collective(dist.all_reduce, t, op=dist.ReduceOp.MAX, group=self.world_group)
        return t

    def all_reduce_scalar(self, value: float, op: str = "sum") -> float:
        t = torch.tensor([float(value)], dtype=torch.float64)
        if not self._host_staged():
            t = t.to(self.device)
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op], group=self.world_group)
        return float(t.item())

    def warm_up(self) -> None:
        """One tiny all-reduce on every group this rank belongs to (world, learners, cohort), in that order.

        RCCL builds a communicator's channels (device allocations, IPC handles) on its first collective; doing that
        here keeps it out of the HIP-graph captures of the first self-play move and the first learner step, during
        which such calls are not permitted.  Every rank has to call this.
        """
        for grp, member in (
            (self.world_group, True),
            (self.learner_group, self.learner_group is not None and self.is_learner),
            (self.cohort_group, self.cohort_group is not None),
        ):
            if not member or (grp is None and grp is not self.world_group):
                continue
            t = torch.zeros(1, dtype=torch.float32, device="cpu" if self._host_staged() else self.device)
            dist.all_reduce(t, group=grp)
        if not self._host_staged():
            torch.cuda.synchronize(self.device)

    def barrier(self) -> None:
        if self._host_staged():
            dist.barrier(group=self.world_group)
        else:
            dist.barrier(group=self.world_group, device_ids=[self.device.index] if self.device.index is not None else None)

    def shutdown(self) -> None:
        if self.owns_process_group and dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:  # pragma: no cover
                logger.debug("destroy_process_group failed", exc_info=True)


def _pick_backend(requested: str) -> str:
    if requested in ("nccl", "gloo"):
        return requested
    return "nccl" if torch.cuda.is_available() else "gloo"


def init_distributed(
    rank: int,
    world_size: int,
    local_rank: int = 0,
    backend: str = "auto",
    layout: str = "colocated",
    num_learners: int = 1,
    init_method: str | None = None,
    timeout_s: float = 600.0,
) -> DistContext:
    """Create the process group (if needed) and the role sub-groups.  Every rank must call this."""
    be = _pick_backend(backend)
    if be == "nccl":
        n_dev = torch.cuda.device_count()
        device = torch.device("cuda", local_rank % max(n_dev, 1))
        torch.cuda.set_device(device)
    else:
        device = torch.device("cuda", local_rank % torch.cuda.device_count()) if torch.cuda.is_available() else torch.device("cpu")
    owns = False
    if not dist.is_initialized():
        kw: dict[str, Any] = {}
        if be == "nccl":
            kw["device_id"] = device
        dist.init_process_group(
            backend=be,
            rank=rank,
            world_size=world_size,
            init_method=init_method,
            timeout=datetime.timedelta(seconds=timeout_s),
            **kw,
        )
        owns = True
    roles = RoleLayout.build(world_size, layout, num_learners)
    # new_group must be called by every rank, in the same order
    learner_group = dist.new_group(ranks=roles.learner_ranks) if len(roles.learner_ranks) < world_size else None
    cohort_group, cohort_ranks = None, []
    if not roles.colocated:
        n_l = len(roles.learner_ranks)
        for li, lr in enumerate(roles.learner_ranks):
            members = [lr] + [a for ai, a in enumerate(roles.actor_ranks) if ai % n_l == li]
            grp = dist.new_group(ranks=members)  # every rank creates every cohort group, in the same order
            if rank in members:
                cohort_group, cohort_ranks = grp, members
    ctx = DistContext(
        rank=rank,
        world_size=world_size,
        local_rank=local_rank,
        backend=be,
        device=device,
        layout=roles,
        learner_group=learner_group,
        world_group=None,
        cohort_group=cohort_group,
        cohort_ranks=cohort_ranks,
        owns_process_group=owns,
    )
    logger.info(
        "rank %d/%d backend=%s device=%s role=%s", rank, world_size, be, device, "learner+actor" if roles.colocated else ("learner" if ctx.is_learner else "actor")
    )
    return ctx


def init_from_env(system_config: Any = None) -> DistContext | None:
    """Initialise from torchrun-style environment variables; ``None`` for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = getattr(system_config, "DIST_BACKEND", "auto") if system_config is not None else "auto"
    layout = getattr(system_config, "ROLE_LAYOUT", "colocated") if system_config is not None else "colocated"
    n_learn = getattr(system_config, "NUM_LEARNER_RANKS", 1) if system_config is not None else 1
    return init_distributed(rank, world, local_rank, backend, layout, n_learn, init_method="env://")
