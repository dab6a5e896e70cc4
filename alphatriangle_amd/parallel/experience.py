"""The link between an actor rank and the learner rank it feeds (split role layout): experience up, weights down.

The reference returns pickled ``SelfPlayResult`` objects through Ray's object store
(alphatriangle/training/worker_manager.py:117-134; each experience carries a dense 360-entry
Python dict) and pushes weights to every worker with a blocking call (:169-209).  Here an
experience is one fixed-stride fp32 record ``[grid | other | policy | return]``, a shipment is the
exact ``[rows, stride]`` block a move produced, and the weights are one flat buffer
(parallel/weights.py); all of it travels point to point on the pair's ONE communicator.

Operation log.  The actor initiates every transfer of its pair - a shipment, or a request for the
weight version the learner has published - and announces it in the job's control store *before*
posting it: kind, size and counters under ``op/<actor>/<slot>/...``, then ``op/<actor>/n`` += 1.
The learner polls those counters between optimizer steps (host side, no device synchronisation)
and posts the matching receive or send, operation by operation in the announced order.  Both ends
therefore enqueue the same sequence on the pair's communicator, which is what lets the two
directions share it.  No fixed-size padding travels, the learner never waits for an actor that is
still playing, and an actor can ask for weights one move ahead of needing them.

Failure.  Every transfer is posted and polled against ``transfer_timeout_s``
(parallel/transport.py); before a learner posts a receive it checks the actor's heartbeat key,
which a background thread of the actor renews every second.  A stale heartbeat retires the actor
without touching the communicator; a transfer that misses its deadline tears down that pair's
communicator and retires the actor.  Either way the learner trains on with the others, as the
reference's manager does when a worker dies (alphatriangle/training/worker_manager.py:153-159).
"""

from __future__ import annotations

import logging
import threading
import time
from dataclasses import dataclass, field
from typing import Any, Callable

import torch

from .dist import DistContext, PeerLostError
from .transport import PairTransport, Transfer, TransferTimeout, wait_for

logger = logging.getLogger(__name__)


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


COUNTER_FIELDS = ("episodes", "simulations", "moves")
KEY_RING = 64        # announcement keys are reused modulo this many operations
MAX_UNACKED = 32     # an actor pauses when it is this many operations ahead of its learner
OP_EXPERIENCE = 1    # actor -> learner: `rows` records
OP_WEIGHTS = 2       # learner -> actor: the published flat weight buffer (`rows` holds the version asked for)


def assigned_learner(ctx: DistContext, actor_rank: int) -> int:
    learners, actors = ctx.layout.learner_ranks, ctx.layout.actor_ranks
    return learners[actors.index(actor_rank) % len(learners)]


class Heartbeat:
    """Actor side: a daemon thread that bumps ``hb/<rank>`` in the control store every ``period_s``."""

    def __init__(self, control: Any, rank: int, period_s: float = 1.0):
        self.control = control
        self.key = f"hb/{rank}"
        self.period_s = float(period_s)
        self._stop = threading.Event()
        self._thread: threading.Thread | None = None

    def beat(self) -> None:
        try:
            self.control.add(self.key, 1)
        except Exception:  # the store's host is gone: the main loop will find out on its own
            logger.debug("heartbeat could not be written", exc_info=True)

    def start(self) -> "Heartbeat":
        self.beat()
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, name="atri-heartbeat", daemon=True)
            self._thread.start()
        return self

    def _run(self) -> None:
        while not self._stop.wait(self.period_s):
            self.beat()

    def stop(self) -> None:
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0 * self.period_s + 1.0)
            self._thread = None


class ExperienceOutbox:
    """Actor side of the link: ``ship`` one move's records, ``request_weights`` / ``take_weights`` a published version."""

    def __init__(self, ctx: DistContext, layout: RecordLayout, device: torch.device, max_rows: int,
                 transport: PairTransport | None = None, heartbeat_period_s: float = 1.0):
        if ctx.control is None:
            raise RuntimeError("the split layout needs the control store")
        self.ctx = ctx
        self.layout = layout
        self.device = device
        self.max_rows = int(max_rows)
        self.learner = assigned_learner(ctx, ctx.rank)
        self.transport: PairTransport = transport if transport is not None else ctx.pair_transport(self.learner, ctx.rank)
        self.timeout_s = float(getattr(ctx, "transfer_timeout_s", 60.0))
        self.ops = 0            # operations announced so far (shipments and weight requests)
        self.shipments = 0
        self.rows_shipped = 0
        self.dropped = 0
        self.patience_s = 120.0
        self._stream = torch.cuda.Stream(device=device) if (device.type == "cuda" and ctx.backend == "nccl") else None
        self._in_flight: list[tuple[Transfer, torch.Tensor]] = []
        self._weights: tuple[Transfer, int] | None = None
        self.heartbeat = Heartbeat(ctx.control, ctx.rank, heartbeat_period_s).start()

    # -- the operation log
    def _announce(self, kind: int, size: int, counters: dict[str, float] | None) -> int:
        k = self.ops + 1
        a = self.ctx.rank
        cp = self.ctx.control
        self._wait_for_room(k)
        slot = k % KEY_RING
        cp.put(f"op/{a}/{slot}/kind", kind)
        cp.put(f"op/{a}/{slot}/rows", size)
        if counters is not None:
            for f in COUNTER_FIELDS:
                cp.put(f"op/{a}/{slot}/{f}", int(counters.get(f, 0)))
        cp.add(f"op/{a}/n", 1)
        self.ops = k
        return k

    def _wait_for_room(self, k: int) -> None:
        """Back-pressure: never run more than ``MAX_UNACKED`` operations ahead of the learner (the key ring is finite)."""
        cp = self.ctx.control
        deadline = time.monotonic() + self.patience_s
        while k - int(cp.peek(f"ack/{self.ctx.rank}", 0) or 0) > MAX_UNACKED:
            if cp.peek("stop", 0):
                return
            if time.monotonic() > deadline:
                raise PeerLostError(f"rank {self.ctx.rank}: learner {self.learner} stopped taking this actor's operations")
            time.sleep(0.01)

    def _reap(self, block: bool) -> None:
        """Drop finished sends; with ``block`` wait for all of them (each bounded by the transfer deadline)."""
        keep = []
        for tr, payload in self._in_flight:
            if block:
                try:
                    wait_for(tr, self.timeout_s, f"experience send to learner {self.learner}", self.transport)
                except TransferTimeout as exc:
                    raise PeerLostError(f"rank {self.ctx.rank}: {exc}") from exc
            elif not tr.done():
                keep.append((tr, payload))
                continue
            else:
                try:
                    tr.finish(self.timeout_s)
                except TransferTimeout as exc:
                    self.transport.abort()
                    raise PeerLostError(f"rank {self.ctx.rank}: {exc}") from exc
        self._in_flight = keep

    # -- experience
    def ship(self, records: torch.Tensor | None, counters: dict[str, float]) -> int:
        """Announce and send ``records`` (``[n, stride]`` fp32, may be empty).  Returns the rows sent."""
        n = 0 if records is None else int(records.shape[0])
        if n > self.max_rows:
            self.dropped += n - self.max_rows
            records, n = records[: self.max_rows], self.max_rows
        self._reap(block=len(self._in_flight) >= 4)  # a learner this far behind is gone or stuck
        self._announce(OP_EXPERIENCE, n, counters)
        self.shipments += 1
        if n > 0:
            payload = records.contiguous()
            try:
                if self._stream is not None:
                    # the transfer waits for the learner's receive on a side stream; the next move's search is not held up
                    self._stream.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(self._stream):
                        tr = self.transport.post_send(payload, self.learner)
                    payload.record_stream(self._stream)
                else:
                    tr = self.transport.post_send(payload, self.learner)
            except (RuntimeError, TransferTimeout) as exc:
                raise PeerLostError(f"rank {self.ctx.rank}: send to learner {self.learner} failed ({exc})") from exc
            self._in_flight.append((tr, payload))
            self.rows_shipped += n
        return n

    # -- weights
    def request_weights(self, version: int, into: torch.Tensor) -> None:
        """Ask the learner for its published weights (announced like a shipment) and post the receive into ``into``.

        Returns at once; :meth:`take_weights` completes it.  Asked for one move ahead, the transfer is served while this
        actor searches, and taking it costs nothing."""
        if self._weights is not None:
            return
        self._announce(OP_WEIGHTS, int(version), None)
        try:
            tr = self.transport.post_recv(into, self.learner)
        except (RuntimeError, TransferTimeout) as exc:
            raise PeerLostError(f"rank {self.ctx.rank}: weight request to learner {self.learner} failed ({exc})") from exc
        self._weights = (tr, int(version))

    def weights_pending(self) -> bool:
        return self._weights is not None

    def take_weights(self) -> int:
        """Wait (bounded) for the requested weights; returns the version that is now in the buffer."""
        if self._weights is None:
            raise RuntimeError("no weight request is outstanding")
        tr, version = self._weights
        self._weights = None
        try:
            wait_for(tr, self.timeout_s, f"weights from learner {self.learner}", self.transport)
        except TransferTimeout as exc:
            raise PeerLostError(f"rank {self.ctx.rank}: {exc}") from exc
        return version

    def close(self) -> None:
        """Tell the learner how many operations to expect in total; finish what is in flight."""
        self.ctx.control.put(f"done/{self.ctx.rank}", self.ops)
        try:
            if self._weights is not None:
                self.take_weights()
            self._reap(block=True)
        except PeerLostError as exc:
            logger.error("closing the link: %s", exc)
        finally:
            self.heartbeat.stop()
        if self._stream is not None:
            self._stream.synchronize()


@dataclass
class Shipment:
    actor: int
    records: torch.Tensor
    counters: dict[str, int]


@dataclass
class _ActorFeed:
    rank: int
    taken: int = 0                       # operations of this actor served so far
    last_seen: float = field(default_factory=time.monotonic)
    retired: bool = False
    final: int | None = None
    hb_value: int = -1
    hb_at: float = field(default_factory=time.monotonic)
    sends: list[tuple[Transfer, float]] = field(default_factory=list)  # weight deliveries in flight, with their post time
    versions_served: int = 0


class ExperienceInbox:
    """Learner side of the links: ``poll`` serves every operation announced since the last call - shipments are returned,
    weight requests answered from ``weights_provider`` - without ever blocking on an actor that has nothing for it."""

    def __init__(self, ctx: DistContext, layout: RecordLayout, device: torch.device, silence_s: float = 300.0,
                 heartbeat_stale_s: float = 45.0, weights_provider: Callable[[int], torch.Tensor | None] | None = None,
                 transports: dict[int, PairTransport] | None = None):
        if ctx.control is None:
            raise RuntimeError("the split layout needs the control store")
        self.ctx = ctx
        self.layout = layout
        self.device = device
        self.silence_s = float(silence_s)
        self.heartbeat_stale_s = float(heartbeat_stale_s)
        self.timeout_s = float(getattr(ctx, "transfer_timeout_s", 60.0))
        self.weights_provider = weights_provider
        self.feeds = [_ActorFeed(a) for a in ctx.layout.actor_ranks if assigned_learner(ctx, a) == ctx.rank]
        self._transports: dict[int, PairTransport] = dict(transports or {})
        self.rows_received = 0
        self.weights_served = 0

    def _transport(self, actor: int) -> PairTransport:
        tr = self._transports.get(actor)
        if tr is None:
            tr = self._transports[actor] = self.ctx.pair_transport(self.ctx.rank, actor)
        return tr

    def active_actors(self) -> int:
        return sum(1 for f in self.feeds if not f.retired and f.final is None)

    def _retire(self, feed: _ActorFeed, why: str) -> None:
        if not feed.retired:
            feed.retired = True
            logger.error("actor rank %d retired: %s.  Training goes on with %d actor(s).", feed.rank, why, self.active_actors())

    def _heartbeat_fresh(self, feed: _ActorFeed, now: float) -> bool:
        value = self.ctx.control.peek(f"hb/{feed.rank}", None)
        if value is None:
            return True  # not started yet (still building its engine): the silence rule covers an actor that never starts
        if int(value) != feed.hb_value:
            feed.hb_value, feed.hb_at = int(value), now
        return now - feed.hb_at <= self.heartbeat_stale_s

    def _check_sends(self, feed: _ActorFeed, now: float) -> None:
        keep = []
        for tr, posted in feed.sends:
            if tr.done():
                try:
                    tr.finish(max(0.05, self.timeout_s - (now - posted)))
                except TransferTimeout as exc:
                    try:
                        self._transport(feed.rank).abort()
                    finally:
                        self._retire(feed, str(exc))
                    return
            elif now - posted > self.timeout_s:
                try:
                    self._transport(feed.rank).abort()
                finally:
                    self._retire(feed, f"a weight delivery was not taken within {self.timeout_s:.0f} s")
                return
            else:
                keep.append((tr, posted))
        feed.sends = keep

    def poll(self) -> list[Shipment]:
        cp = self.ctx.control
        out: list[Shipment] = []
        now = time.monotonic()
        for feed in self.feeds:
            if feed.retired:
                continue
            self._check_sends(feed, now)
            if feed.retired:
                continue
            announced = cp.counter(f"op/{feed.rank}/n")
            while feed.taken < announced and not feed.retired:
                k = feed.taken + 1
                slot = k % KEY_RING
                kind = int(cp.peek(f"op/{feed.rank}/{slot}/kind", OP_EXPERIENCE) or OP_EXPERIENCE)
                size = int(cp.peek(f"op/{feed.rank}/{slot}/rows", 0) or 0)
                transport = self._transport(feed.rank)
                if kind == OP_WEIGHTS:
                    buf = self.weights_provider(size) if self.weights_provider is not None else None
                    if buf is None:
                        break  # nothing published yet that answers the request: serve it on a later poll
                    if not self._heartbeat_fresh(feed, now):
                        self._retire(feed, "its heartbeat stopped (weight request not served)")
                        break
                    try:
                        feed.sends.append((transport.post_send(buf, feed.rank), now))
                    except (RuntimeError, TransferTimeout) as exc:
                        self._retire(feed, f"weight delivery could not be posted ({exc})")
                        break
                    feed.versions_served += 1
                    self.weights_served += 1
                else:
                    counters = {f: int(cp.peek(f"op/{feed.rank}/{slot}/{f}", 0) or 0) for f in COUNTER_FIELDS}
                    rec = torch.empty(size, self.layout.stride, dtype=torch.float32, device=self.device)
                    if size > 0:
                        # no receive is posted for an actor that may be dead: its send would never come
                        if not self._heartbeat_fresh(feed, now):
                            self._retire(feed, "its heartbeat stopped (announced shipment not received)")
                            break
                        try:
                            wait_for(transport.post_recv(rec, feed.rank), self.timeout_s, f"shipment {k} of actor {feed.rank}", transport)
                        except (RuntimeError, TransferTimeout) as exc:
                            self._retire(feed, str(exc))
                            break
                    self.rows_received += size
                    out.append(Shipment(feed.rank, rec, counters))
                feed.taken = k
                cp.put(f"ack/{feed.rank}", k)
                feed.last_seen = now
            if feed.retired:
                continue
            if feed.final is None:
                feed.final = cp.peek(f"done/{feed.rank}")
            if feed.final is None:
                if now - feed.last_seen > self.silence_s:
                    self._retire(feed, f"no move finished for {self.silence_s:.0f} s")
                elif not self._heartbeat_fresh(feed, now):
                    self._retire(feed, "its heartbeat stopped")
        return out

    def drained(self) -> bool:
        """True once every live actor has said how many operations it made and all of them have been served."""
        return all(f.retired or (f.final is not None and f.taken >= f.final) for f in self.feeds)

    def finish_sends(self) -> None:
        """Wait (bounded) for weight deliveries still in flight; used when the learner leaves."""
        now = time.monotonic()
        for feed in self.feeds:
            for tr, posted in feed.sends:
                try:
                    wait_for(tr, max(0.1, self.timeout_s - (now - posted)), f"weights to actor {feed.rank}", self._transport(feed.rank))
                except TransferTimeout as exc:
                    self._retire(feed, str(exc))
            feed.sends = []
