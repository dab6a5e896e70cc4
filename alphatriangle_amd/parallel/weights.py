"""Flat weight wire for learner -> actor publication.

The reference ships a pickled CPU ``state_dict`` through Ray's object store on every
update (alphatriangle/training/worker_manager.py:169-209).  Here everything an actor needs
to rebuild its inference engine - every parameter plus the BatchNorm running statistics -
is packed into ONE contiguous fp32 buffer, so a publication is a single RCCL broadcast
(about 12 MB for the default net) followed by one scatter back into the module tensors.
The positional-encoding table is deterministic and ``num_batches_tracked`` counters do not
influence evaluation, so neither travels.
"""

from __future__ import annotations

import torch


class WeightWire:
    def __init__(self, module: torch.nn.Module):
        self.names: list[str] = []
        self.tensors: list[torch.Tensor] = []
        for name, p in module.named_parameters():
            self.names.append(name)
            self.tensors.append(p)
        for name, b in module.named_buffers():
            if not b.dtype.is_floating_point or name.endswith("pos_encoder.pe") or name == "pos_encoder.pe":
                continue
            self.names.append(name)
            self.tensors.append(b)
        if not self.tensors:
            raise ValueError("module has nothing to publish")
        self.sizes = [int(t.numel()) for t in self.tensors]
        self.total = sum(self.sizes)
        self.device = self.tensors[0].device
        self.buffer = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self._views = list(torch.split(self.buffer, self.sizes))

    @torch.no_grad()
    def pack(self) -> torch.Tensor:
        """Module tensors -> wire buffer (one concatenation)."""
        torch.cat([t.detach().reshape(-1).float() for t in self.tensors], out=self.buffer)
        return self.buffer

    @torch.no_grad()
    def unpack(self) -> None:
        """Wire buffer -> module tensors (in place, so flat-parameter views and engine bindings stay valid)."""
        for t, v in zip(self.tensors, self._views, strict=True):
            t.detach().copy_(v.view(t.shape))
