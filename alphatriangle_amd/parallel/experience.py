"""Experience shipping from actor ranks to learner ranks (split role layout).

The reference returns pickled ``SelfPlayResult`` objects through Ray's object store
(alphatriangle/training/worker_manager.py:117-134); each experience carries a dense
360-entry Python dict.  Here an experience is one fixed-stride fp32 record
``[grid | other | policy | return]`` and a shipment is ONE gather of equally sized record
blocks per round onto each learner from the actors assigned to it (split layout; only the
learner receives, so actor GPUs spend no xGMI bandwidth on each other's data).  Every rank
contributes ``[capacity + 1, stride]`` (row 0 holds the live count).  Fixed sizes keep the
collective free of host-side size negotiation.
"""

from __future__ import annotations

from dataclasses import dataclass

import torch

from .dist import DistContext


@dataclass
class RecordLayout:
    grid: int
    other: int
    policy: int

    @property
    def stride(self) -> int:
        return self.grid + self.other + self.policy + 1

    def pack(self, grid: torch.Tensor, other: torch.Tensor, policy: torch.Tensor, ret: torch.Tensor) -> torch.Tensor:
        n = grid.shape[0]
        return torch.cat([grid.reshape(n, -1).float(), other.float(), policy.float(), ret.float().view(n, 1)], dim=1)

    def unpack(self, rec: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        g, o, p = self.grid, self.other, self.policy
        return rec[:, :g], rec[:, g : g + o], rec[:, g + o : g + o + p], rec[:, g + o + p]


class ExperienceShipper:
    def __init__(self, ctx: DistContext, layout: RecordLayout, capacity: int, device: torch.device):
        self.ctx = ctx
        self.layout = layout
        self.capacity = int(capacity)
        self.device = device
        self.block = torch.zeros(self.capacity + 1, layout.stride, dtype=torch.float32, device=device)
        self.dropped = 0

    def assigned_learner(self, actor_rank: int) -> int:
        learners = self.ctx.layout.learner_ranks
        actors = self.ctx.layout.actor_ranks
        return learners[actors.index(actor_rank) % len(learners)]

    def exchange(self, records: torch.Tensor | None) -> torch.Tensor:
        """Collective: contribute local records (actors) and return the records this rank must ingest (learners).

        Every rank calls this once per round; non-actors pass ``None``.  Records beyond the block
        capacity are dropped and counted.
        """
        n = 0
        if records is not None and records.numel() > 0 and self.ctx.is_actor:
            n = min(int(records.shape[0]), self.capacity)
            self.dropped += int(records.shape[0]) - n
            self.block[1 : 1 + n].copy_(records[:n].to(self.device))
        self.block[0, 0] = float(n)
        if not self.ctx.layout.colocated and self.ctx.cohort_group is not None:
            blocks = self.ctx.gather_blocks_to_learner(self.block)
            if blocks is None:
                return torch.empty(0, self.layout.stride, dtype=torch.float32, device=self.device)
            parts = []
            for blk in blocks[1:]:  # [0] is the learner's own (empty) block
                cnt = int(blk[0, 0].item())
                if cnt > 0:
                    parts.append(blk[1 : 1 + cnt].to(self.device))
            if not parts:
                return torch.empty(0, self.layout.stride, dtype=torch.float32, device=self.device)
            return torch.cat(parts, dim=0)
        gathered = self.ctx.all_gather_records(self.block)  # [world, cap+1, stride]
        if not self.ctx.is_learner:
            return torch.empty(0, self.layout.stride, dtype=torch.float32, device=self.device)
        parts = []
        for r in self.ctx.layout.actor_ranks:
            if self.ctx.layout.colocated and r != self.ctx.rank:
                continue  # colocated ranks keep their own experiences local
            if not self.ctx.layout.colocated and self.assigned_learner(r) != self.ctx.rank:
                continue
            cnt = int(gathered[r, 0, 0].item())
            if cnt > 0:
                parts.append(gathered[r, 1 : 1 + cnt])
        if not parts:
            return torch.empty(0, self.layout.stride, dtype=torch.float32, device=self.device)
        return torch.cat(parts, dim=0)
