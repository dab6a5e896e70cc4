"""Process-group runtime: one process per GPU, RCCL over xGMI (``nccl`` backend on ROCm).

The reference has no collective communication at all - its only transport is Ray's
object store (alphatriangle/training/worker_manager.py:44-45,179-194 broadcast
weights by ``ray.put`` + N remote calls; :117-134 gather episode results).  The
equivalents here:

* weights learner -> actors : the flat fp32 parameter wire (12 MB for the default net), point to point to
  each actor over its own xGMI link when the actor asks for a newer version at a move boundary
  (training/split_loop.py); ``broadcast_weights`` is the collective form for lock-step callers;
* learner gradients (data parallel) : ONE ``all_reduce`` of the flat gradient buffer
  between backward and the fused optimizer step.  xGMI is point-to-point, so a
  message this small is latency-bound; it is deliberately a single un-bucketed call;
* experiences actors -> learner : point-to-point sends of exactly the rows a move produced, announced in the
  control store first (parallel/experience.py).  Each (learner, actor) pair owns ONE 2-rank communicator; both sides
  post their transfers in the order of the pair's announcements, so the two directions never deadlock on it, and
  every transfer is posted and polled against a deadline (parallel/transport.py): a peer that died costs its pair's
  communicator, not the process.

``gloo`` (CPU tests, rehearsals) only ever sees CPU tensors: device tensors are
staged through host memory with synchronous copies.

Control plane.  Decisions (ready to train? stop? how many steps this move?) never ride
on the RCCL data plane: they go through a host-side channel - a ``gloo`` side group for
the collective "agree" of data-parallel ranks, and the job's rendezvous key/value store
(:class:`ControlPlane`) for the asynchronous split layout, where every transfer is
announced in the store before either side posts it.  A rank that disappears therefore
shows up as :class:`PeerLostError` within ``control_timeout_s`` (connection reset, or the
time-out) instead of stranding the others inside a device collective for minutes - the
reference's manager likewise notices a dead worker and carries on or stops
(alphatriangle/training/worker_manager.py:153-167).
"""

from __future__ import annotations

import datetime
import importlib
import logging
import os
from dataclasses import dataclass, field
from typing import Any

import torch
import torch.distributed as dist

logger = logging.getLogger(__name__)


class PeerLostError(RuntimeError):
    """A rank of the job stopped answering on the control plane (died, hung, or was killed)."""


class ControlPlane:
    """Host-side key/value channel over the job's rendezvous store (TCP store of rank 0, or a file store in tests).

    Only integers travel: counters (``add``), versions and flags (``put`` / ``peek``).  Nothing here touches the GPU
    and nothing blocks: a key that does not exist yet reads as ``default``.
    """

    def __init__(self, store: Any, prefix: str = "atri"):
        self.store = store
        self.prefix = prefix

    def _k(self, key: str) -> str:
        return f"{self.prefix}/{key}"

    def add(self, key: str, amount: int = 1) -> int:
        return int(self.store.add(self._k(key), int(amount)))

    def counter(self, key: str) -> int:
        return int(self.store.add(self._k(key), 0))

    def put(self, key: str, value: int) -> None:
        self.store.set(self._k(key), str(int(value)))

    def peek(self, key: str, default: int | None = None) -> int | None:
        k = self._k(key)
        try:
            if not self.store.check([k]):
                return default
            return int(self.store.get(k))
        except Exception:  # a store whose host went away
            raise PeerLostError(f"control store unreachable while reading '{key}'") from None


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
    owns_process_group: bool = True
    control_group: Any = None         # gloo side group for host-side agreement (None: the world group is gloo already)
    learner_control_group: Any = None  # the same for the learner ranks only (split layout with several learners)
    control: ControlPlane | None = None
    pair_groups: dict[tuple[int, int], Any] = field(default_factory=dict)  # (learner, actor) -> the pair's 2-rank group
    transfer_timeout_s: float = 60.0  # deadline of one posted point-to-point transfer (parallel/transport.py)
    _transports: dict[tuple[int, int], Any] = field(default_factory=dict)

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

    def broadcast_learners(self, tensor: torch.Tensor) -> None:
        """Blocking broadcast from the first learner to the other learner ranks (every learner calls this)."""
        if self.num_learners <= 1 or not self.is_learner:
            return
        self._collective(dist.broadcast, tensor, src=self.layout.learner_ranks[0], group=self.learner_group)

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

    def agree(self, values: list[float], op: str = "min", group: str = "world") -> list[float]:
        """Host-side all-reduce of a few numbers over every rank, or over the learner ranks (control plane, CPU tensors only).

        Raises :class:`PeerLostError` when a rank does not answer within the control time-out.
        """
        members = self.world_size if group == "world" else self.num_learners
        if members <= 1:
            return [float(v) for v in values]
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        rop = {"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op]
        if group == "world":
            grp, side = (self.control_group, True) if self.control_group is not None else (self.world_group, False)
        else:
            grp, side = (self.learner_control_group, True) if self.learner_control_group is not None else (self.learner_group, False)
        if not side and not self._host_staged():
            t = t.to(self.device)  # no side group was created: fall back to the data plane
        try:
            dist.all_reduce(t, op=rop, group=grp)
        except RuntimeError as exc:
            raise PeerLostError(f"rank {self.rank}: control-plane agreement failed ({exc})") from exc
        return [float(x) for x in t.cpu().tolist()]

    # ------------------------------------------------------------------ point to point (split layout)
    def pair_transport(self, learner: int, actor: int) -> Any:
        """The transport of one (learner, actor) pair (created on first use; see parallel/transport.py)."""
        key = (learner, actor)
        tr = self._transports.get(key)
        if tr is None:
            from .transport import TorchPairTransport

            try:
                grp = self.pair_groups[key]
            except KeyError:
                raise RuntimeError(f"no point-to-point group for learner {learner} / actor {actor}") from None
            tr = self._transports[key] = TorchPairTransport(grp, self._host_staged())
        return tr

    def send_to(self, tensor: torch.Tensor, peer: int, *, learner: int, actor: int) -> None:
        """Send ``tensor`` to ``peer`` on the pair's communicator and wait for it (bounded by ``transfer_timeout_s``)."""
        from .transport import TransferTimeout, wait_for

        tr = self.pair_transport(learner, actor)
        try:
            wait_for(tr.post_send(tensor, peer), self.transfer_timeout_s, f"send to rank {peer}", tr)
        except (RuntimeError, TransferTimeout) as exc:
            raise PeerLostError(f"rank {self.rank}: send to rank {peer} failed ({exc})") from exc

    def recv_from(self, tensor: torch.Tensor, peer: int, *, learner: int, actor: int) -> None:
        """Receive into ``tensor`` (contiguous) from ``peer`` on the pair's communicator, bounded like :meth:`send_to`."""
        from .transport import TransferTimeout, wait_for

        tr = self.pair_transport(learner, actor)
        try:
            wait_for(tr.post_recv(tensor, peer), self.transfer_timeout_s, f"receive from rank {peer}", tr)
        except (RuntimeError, TransferTimeout) as exc:
            raise PeerLostError(f"rank {self.rank}: receive from rank {peer} failed ({exc})") from exc

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
        """One tiny transfer on every communicator this rank belongs to (world, learners, its point-to-point pairs).

        RCCL builds a communicator's channels (device allocations, IPC handles) on its first collective; doing that
        here keeps it out of the HIP-graph captures of the first self-play move and the first learner step, during
        which such calls are not permitted.  Every rank has to call this.
        """
        for grp, member in (
            (self.world_group, True),
            (self.learner_group, self.learner_group is not None and self.is_learner),
        ):
            if not member or (grp is None and grp is not self.world_group):
                continue
            t = torch.zeros(1, dtype=torch.float32, device="cpu" if self._host_staged() else self.device)
            dist.all_reduce(t, group=grp)
        # point-to-point communicators of the split layout: one tiny transfer per pair and direction, in a fixed order
        for (learner, actor), _group in sorted(self.pair_groups.items()):
            t = torch.zeros(1, dtype=torch.float32, device="cpu" if self._host_staged() else self.device)
            if self.rank == actor:
                self.send_to(t, learner, learner=learner, actor=actor)
                self.recv_from(t, learner, learner=learner, actor=actor)
            else:
                self.recv_from(t, actor, learner=learner, actor=actor)
                self.send_to(t, actor, learner=learner, actor=actor)
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
    timeout_s: float = 120.0,
    control_timeout_s: float = 60.0,
    transfer_timeout_s: float = 60.0,
) -> DistContext:
    """Create the process group (if needed) and the role sub-groups.  Every rank must call this.

    ``timeout_s`` bounds every data-plane collective (a missing rank ends the others non-zero after it);
    ``control_timeout_s`` bounds the host-side agreements, which is where a lost rank is noticed first.
    """
    be = _pick_backend(backend)
    if be == "nccl":
        n_dev = torch.cuda.device_count()
        device = torch.device("cuda", local_rank % max(n_dev, 1))
        torch.cuda.set_device(device)
    else:
        device = torch.device("cuda", local_rank % torch.cuda.device_count()) if torch.cuda.is_available() else torch.device("cpu")
    owns = False
    if not dist.is_initialized():
        # torch.distributed.nn.functional declares `group=group.WORLD` in ten signatures.  torch imports it lazily (with
        # torch._dynamo, on the first optimizer step); imported while a process group exists, those defaults pin the group
        # for the life of the process, destroy_process_group() no longer ends its worker threads, and a gloo worker that
        # drops the last reference of a tensor while the interpreter finalizes aborts the process ("terminate called
        # without an active exception", exit code 134 after a complete run).  Imported here, the defaults hold None.
        importlib.import_module("torch.distributed.nn.functional")

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
    # control plane: a gloo side group when the data plane is RCCL (host tensors, short time-out)
    control_group = None
    if be == "nccl" and world_size > 1:
        control_group = dist.new_group(ranks=list(range(world_size)), backend="gloo",
                                       timeout=datetime.timedelta(seconds=control_timeout_s))
    learner_control_group = None
    if be == "nccl" and 1 < len(roles.learner_ranks) < world_size:
        learner_control_group = dist.new_group(ranks=roles.learner_ranks, backend="gloo",
                                               timeout=datetime.timedelta(seconds=control_timeout_s))
    # split layout: ONE 2-rank group per (learner, actor) pair.  Both sides post their transfers in the order of the
    # pair's announcements in the control store (the actor initiates every one of them: a shipment, or a request for the
    # published weights), so experience up and weights down share the communicator without deadlock.  A transfer on it is
    # posted, polled and - past transfer_timeout_s - answered by tearing this one communicator down, so the library's
    # own watchdog must neither fire first (long time-out) nor end the process for it (no asynchronous error handling
    # for these groups: the setting is read when a group is constructed).
    pair_groups: dict[tuple[int, int], Any] = {}
    if not roles.colocated:
        n_l = len(roles.learner_ranks)
        saved = os.environ.get("TORCH_NCCL_ASYNC_ERROR_HANDLING")
        if be == "nccl":
            os.environ["TORCH_NCCL_ASYNC_ERROR_HANDLING"] = "0"
        try:
            for ai, a in enumerate(roles.actor_ranks):
                lr = roles.learner_ranks[ai % n_l]
                grp = dist.new_group(ranks=[lr, a], timeout=datetime.timedelta(seconds=max(timeout_s, 10.0 * transfer_timeout_s) + 3600.0))
                if rank in (lr, a):
                    pair_groups[(lr, a)] = grp
        finally:
            if be == "nccl":
                if saved is None:
                    os.environ.pop("TORCH_NCCL_ASYNC_ERROR_HANDLING", None)
                else:
                    os.environ["TORCH_NCCL_ASYNC_ERROR_HANDLING"] = saved
    control = None
    try:
        control = ControlPlane(dist.distributed_c10d._get_default_store())
    except Exception:  # pragma: no cover - a process group created elsewhere without a default store
        logger.debug("no default store available for the control plane", exc_info=True)
    ctx = DistContext(
        rank=rank,
        world_size=world_size,
        local_rank=local_rank,
        backend=be,
        device=device,
        layout=roles,
        learner_group=learner_group,
        world_group=None,
        owns_process_group=owns,
        control_group=control_group,
        learner_control_group=learner_control_group,
        control=control,
        pair_groups=pair_groups,
        transfer_timeout_s=float(transfer_timeout_s),
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
    timeout_s = float(getattr(system_config, "COLLECTIVE_TIMEOUT_S", 120.0)) if system_config is not None else 120.0
    control_s = float(getattr(system_config, "CONTROL_TIMEOUT_S", 60.0)) if system_config is not None else 60.0
    transfer_s = float(getattr(system_config, "TRANSFER_TIMEOUT_S", 60.0)) if system_config is not None else 60.0
    return init_distributed(rank, world, local_rank, backend, layout, n_learn, init_method="env://",
                            timeout_s=timeout_s, control_timeout_s=control_s, transfer_timeout_s=transfer_s)
