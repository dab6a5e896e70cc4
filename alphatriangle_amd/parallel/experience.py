"""Experience shipping from actor ranks to learner ranks (split role layout).

The reference returns pickled ``SelfPlayResult`` objects through Ray's object store
(alphatriangle/training/worker_manager.py:117-134); each experience carries a dense
360-entry Python dict.  Here an experience is one fixed-stride fp32 record
``[grid | other | policy | return]`` and a shipment is the exact ``[rows, stride]`` block a
move produced - at most ``games * (n_step + 1)`` rows, usually about ``games`` - sent point
to point from the actor to the one learner it feeds.

Announce, then transfer.  The actor first writes the shipment's row count and its self-play
counters into the job's control store and bumps its shipment counter; only then does it post
the send.  The learner polls the counters between optimizer steps (host side, no device
synchronisation), reads the row count from the store and posts a receive of exactly that
size.  Consequences: no fixed-size padding travels, the learner never waits inside a
collective for an actor that is still playing, and an actor that died simply stops
announcing - the learner carries on with the others.
"""

from __future__ import annotations

import time
from dataclasses import dataclass, field

import torch

from .dist import DistContext, PeerLostError


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
KEY_RING = 64        # announcement keys are reused modulo this many shipments
MAX_UNACKED = 32     # an actor pauses when it is this many shipments ahead of its learner


def assigned_learner(ctx: DistContext, actor_rank: int) -> int:
    learners, actors = ctx.layout.learner_ranks, ctx.layout.actor_ranks
    return learners[actors.index(actor_rank) % len(learners)]


class ExperienceOutbox:
    """Actor side: ``ship`` one move's records to this rank's learner."""

    def __init__(self, ctx: DistContext, layout: RecordLayout, device: torch.device, max_rows: int):
        if ctx.control is None:
            raise RuntimeError("the split layout needs the control store")
        self.ctx = ctx
        self.layout = layout
        self.device = device
        self.max_rows = int(max_rows)
        self.learner = assigned_learner(ctx, ctx.rank)
        self.shipments = 0
        self.rows_shipped = 0
        self.dropped = 0
        self.patience_s = 120.0
        self._stream = torch.cuda.Stream(device=device) if (device.type == "cuda" and ctx.backend == "nccl") else None
        self._in_flight: torch.Tensor | None = None

    def ship(self, records: torch.Tensor | None, counters: dict[str, float]) -> int:
        """Announce and send ``records`` (``[n, stride]`` fp32, may be empty).  Returns the rows sent."""
        n = 0 if records is None else int(records.shape[0])
        if n > self.max_rows:
            self.dropped += n - self.max_rows
            records, n = records[: self.max_rows], self.max_rows
        k = self.shipments + 1
        a = self.ctx.rank
        cp = self.ctx.control
        self._wait_for_room(k)
        slot = k % KEY_RING
        cp.put(f"ship/{a}/{slot}/rows", n)
        for f in COUNTER_FIELDS:
            cp.put(f"ship/{a}/{slot}/{f}", int(counters.get(f, 0)))
        cp.add(f"ship/{a}/n", 1)
        self.shipments = k
        if n > 0:
            payload = records.contiguous()
            if self._stream is not None:
                # the transfer waits for the learner's receive on a side stream; the next move's search is not held up
                self._stream.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self._stream):
                    self.ctx.send_to(payload, self.learner, learner=self.learner, actor=a, weights=False)
                payload.record_stream(self._stream)
                self._in_flight = payload
            else:
                self.ctx.send_to(payload, self.learner, learner=self.learner, actor=a, weights=False)
            self.rows_shipped += n
        return n

    def _wait_for_room(self, k: int) -> None:
        """Back-pressure: never run more than ``MAX_UNACKED`` shipments ahead of the learner (the key ring is finite)."""
        cp = self.ctx.control
        deadline = time.monotonic() + self.patience_s
        while k - int(cp.peek(f"ack/{self.ctx.rank}", 0) or 0) > MAX_UNACKED:
            if cp.peek("stop", 0):
                return
            if time.monotonic() > deadline:
                raise PeerLostError(f"rank {self.ctx.rank}: learner {self.learner} stopped taking experience shipments")
            time.sleep(0.01)

    def close(self) -> None:
        """Tell the learner how many shipments to expect in total."""
        self.ctx.control.put(f"done/{self.ctx.rank}", self.shipments)
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
    received: int = 0
    last_seen: float = field(default_factory=time.monotonic)
    retired: bool = False
    final: int | None = None


class ExperienceInbox:
    """Learner side: ``poll`` returns every shipment announced since the last call, without blocking on absent ones."""

    def __init__(self, ctx: DistContext, layout: RecordLayout, device: torch.device, silence_s: float = 300.0):
        if ctx.control is None:
            raise RuntimeError("the split layout needs the control store")
        self.ctx = ctx
        self.layout = layout
        self.device = device
        self.silence_s = float(silence_s)
        self.feeds = [_ActorFeed(a) for a in ctx.layout.actor_ranks if assigned_learner(ctx, a) == ctx.rank]
        self.rows_received = 0

    def active_actors(self) -> int:
        return sum(1 for f in self.feeds if not f.retired and f.final is None)

    def poll(self) -> list[Shipment]:
        cp = self.ctx.control
        out: list[Shipment] = []
        now = time.monotonic()
        for feed in self.feeds:
            if feed.retired:
                continue
            announced = cp.counter(f"ship/{feed.rank}/n")
            while feed.received < announced:
                k = feed.received + 1
                slot = k % KEY_RING
                rows = int(cp.peek(f"ship/{feed.rank}/{slot}/rows", 0) or 0)
                counters = {f: int(cp.peek(f"ship/{feed.rank}/{slot}/{f}", 0) or 0) for f in COUNTER_FIELDS}
                rec = torch.empty(rows, self.layout.stride, dtype=torch.float32, device=self.device)
                if rows > 0:
                    try:
                        self.ctx.recv_from(rec, feed.rank, learner=self.ctx.rank, actor=feed.rank, weights=False)
                    except PeerLostError:
                        feed.retired = True
                        break
                feed.received = k
                cp.put(f"ack/{feed.rank}", k)
                feed.last_seen = now
                self.rows_received += rows
                out.append(Shipment(feed.rank, rec, counters))
            if feed.retired:
                continue
            if feed.final is None:
                feed.final = cp.peek(f"done/{feed.rank}")
            if feed.final is None and now - feed.last_seen > self.silence_s:
                feed.retired = True  # no move finished for this long: the actor is gone, carry on without it
        return out

    def drained(self) -> bool:
        """True once every live actor has said how many shipments it made and all of them have arrived."""
        return all(f.retired or (f.final is not None and f.received >= f.final) for f in self.feeds)
