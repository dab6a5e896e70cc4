"""Point-to-point transfers between one learner and one actor whose failure does not take the caller down.

The reference's manager notices a dead worker when its result cannot be fetched, drops it and carries on
(alphatriangle/training/worker_manager.py:153-159).  On RCCL a send or a receive returns as soon as it is enqueued;
a peer that died afterwards never shows up as an exception of that call - the operation just never completes, and the
library's own watchdog would end the whole process after its time-out.  So every transfer here is *posted*
(``isend`` / ``irecv``), its completion is *polled* from the host against a deadline of our own, and a transfer that
misses the deadline tears down the communicator of that one pair (:meth:`PairTransport.abort`) and raises
:class:`TransferTimeout` - the caller retires the peer and keeps every other pair.

``PairTransport`` is the seam the tests use to stand in a transport whose failure surfaces later than the call.
"""

from __future__ import annotations

import datetime
import logging
import time
from typing import Any, Protocol

import torch
import torch.distributed as dist

logger = logging.getLogger(__name__)


class TransferTimeout(RuntimeError):
    """A posted transfer did not complete within its deadline (the peer is gone or stuck)."""


class Transfer(Protocol):
    def done(self) -> bool:
        """Non-blocking.  True: :meth:`finish` will not wait for the peer (a transport that cannot be polled - gloo -
        answers True and does its bounded wait inside :meth:`finish`)."""

    def finish(self, timeout_s: float = 0.0) -> None:
        """Complete the transfer (stream ordering, host-staged copy); raises :class:`TransferTimeout` past ``timeout_s``."""


class PairTransport(Protocol):
    """One learner <-> one actor.  Both sides post their operations in the same order (the order of the pair's
    announcements in the control store), so a single communicator carries both directions without deadlock."""

    def post_send(self, tensor: torch.Tensor, peer: int) -> Transfer: ...
    def post_recv(self, tensor: torch.Tensor, peer: int) -> Transfer: ...
    def abort(self) -> None: ...


def wait_for(transfer: Transfer, timeout_s: float, what: str, transport: PairTransport | None = None) -> None:
    """Poll ``transfer`` until it is done; past the deadline abort the pair's communicator and raise."""
    deadline = time.monotonic() + float(timeout_s)
    pause = 0.0
    try:
        while not transfer.done():
            if time.monotonic() > deadline:
                raise TransferTimeout(f"{what} did not complete within {timeout_s:.1f} s")
            time.sleep(pause)
            pause = min(0.002, pause + 0.0002)  # first polls back to back: a served transfer takes microseconds
        transfer.finish(max(0.05, deadline - time.monotonic()))
    except TransferTimeout:
        if transport is not None:
            try:
                transport.abort()
            except Exception:  # pragma: no cover - the communicator is in an unknown state already
                logger.debug("abort of a timed-out pair failed", exc_info=True)
        raise


class _TorchTransfer:
    def __init__(self, work: Any, pollable: bool, what: str, after: Any = None):
        self.work = work
        self.pollable = pollable  # RCCL: completion is a device event the host can query; gloo: only a (bounded) wait tells
        self.what = what
        self.after = after        # host-staged receive: copy into the device tensor once the data is there
        self._finished = False

    def done(self) -> bool:
        return self._finished or not self.pollable or bool(self.work.is_completed())

    def finish(self, timeout_s: float = 0.0) -> None:
        if self._finished:
            return
        if self.pollable:
            self.work.wait()  # orders the current stream behind the transfer; the host has seen it complete
        else:
            try:
                self.work.wait(datetime.timedelta(seconds=max(float(timeout_s), 0.05)))
            except RuntimeError as exc:  # gloo: time-out (which also closes the pair's connection) or a reset connection
                raise TransferTimeout(f"{self.what}: {exc}") from exc
        self._finished = True
        if self.after is not None:
            self.after()


class TorchPairTransport:
    """``torch.distributed`` 2-rank group (RCCL, or gloo with host staging of device tensors)."""

    def __init__(self, group: Any, host_staged: bool):
        self.group = group
        self.host_staged = host_staged
        self.aborted = False

    def _check(self) -> None:
        if self.aborted:
            raise TransferTimeout("this pair's communicator has been torn down")

    def post_send(self, tensor: torch.Tensor, peer: int) -> Transfer:
        self._check()
        if self.host_staged and tensor.is_cuda:
            tensor = tensor.detach().to("cpu", non_blocking=False)
        return _TorchTransfer(dist.isend(tensor.contiguous(), dst=peer, group=self.group), not self.host_staged, f"send to rank {peer}")

    def post_recv(self, tensor: torch.Tensor, peer: int) -> Transfer:
        self._check()
        if self.host_staged and tensor.is_cuda:
            host = torch.empty(tensor.shape, dtype=tensor.dtype)
            return _TorchTransfer(dist.irecv(host, src=peer, group=self.group), False, f"receive from rank {peer}",
                                  after=lambda: tensor.copy_(host, non_blocking=False))
        return _TorchTransfer(dist.irecv(tensor, src=peer, group=self.group), not self.host_staged, f"receive from rank {peer}")

    def abort(self) -> None:
        """Tear down this pair's communicator only; every other group of the process stays usable."""
        if self.aborted:
            return
        self.aborted = True
        try:
            backend = self.group._get_backend(torch.device("cuda")) if not self.host_staged else None
        except Exception:
            backend = None
        for name in ("abort", "_abort", "_shutdown"):
            fn = getattr(backend, name, None)
            if callable(fn):
                try:
                    fn()
                    logger.error("pair communicator torn down after a transfer missed its deadline.")
                    return
                except Exception:  # pragma: no cover
                    logger.debug("%s() on a pair communicator failed", name, exc_info=True)
