"""Aggregates raw metric events into logged metric values (mean / sum / latest / min / max / count / rate)."""

from __future__ import annotations

import time
from collections import defaultdict
from dataclasses import dataclass, field

from .config import MetricConfig, StatsConfig
from .events import RawMetricEvent


@dataclass
class _Pending:
    values: list[float] = field(default_factory=list)
    last_step: int = 0


class StatsAggregator:
    def __init__(self, config: StatsConfig):
        self.config = config
        self._by_event: dict[str, list[MetricConfig]] = defaultdict(list)
        for m in config.metrics:
            self._by_event[m.event_name].append(m)
        self._pending: dict[str, _Pending] = {m.name: _Pending() for m in config.metrics}
        self._last_log_step: dict[str, int] = {m.name: -(10**9) for m in config.metrics}
        self._last_log_time: dict[str, float] = {m.name: time.monotonic() for m in config.metrics}
        self._rate_mark: dict[str, float] = {m.name: time.monotonic() for m in config.metrics}
        self.events_seen = 0

    def add(self, ev: RawMetricEvent) -> None:
        self.events_seen += 1
        for m in self._by_event.get(ev.name, ()):
            if m.context_key is not None:
                if m.context_key not in ev.context:
                    continue
                try:
                    v = float(ev.context[m.context_key])
                except (TypeError, ValueError):
                    continue
            else:
                v = float(ev.value)
            p = self._pending[m.name]
            p.values.append(v)
            p.last_step = max(p.last_step, int(ev.global_step))

    def add_many(self, name: str, values, global_step: int) -> None:  # noqa: ANN001
        """One sample per element of ``values`` for every metric fed by event ``name`` (context-keyed metrics excluded)."""
        vals = [float(v) for v in values]
        if not vals:
            return
        self.events_seen += len(vals)
        for m in self._by_event.get(name, ()):
            if m.context_key is not None:
                continue
            p = self._pending[m.name]
            p.values.extend(vals)
            p.last_step = max(p.last_step, int(global_step))

    def collect(self, global_step: int, force: bool = False) -> list[tuple[MetricConfig, float, int]]:
        """Return ``(metric, value, step)`` for every metric that is due."""
        now = time.monotonic()
        out: list[tuple[MetricConfig, float, int]] = []
        for m in self.config.metrics:
            p = self._pending[m.name]
            if not p.values:
                continue
            due = force
            if not due:
                step_ok = m.log_frequency_steps == 0 or global_step - self._last_log_step[m.name] >= m.log_frequency_steps
                time_ok = m.log_frequency_seconds == 0 or now - self._last_log_time[m.name] >= m.log_frequency_seconds
                due = step_ok and time_ok
            if not due:
                continue
            vals = p.values
            if m.aggregation == "mean":
                v = sum(vals) / len(vals)
            elif m.aggregation == "sum":
                v = sum(vals)
            elif m.aggregation == "latest":
                v = vals[-1]
            elif m.aggregation == "min":
                v = min(vals)
            elif m.aggregation == "max":
                v = max(vals)
            elif m.aggregation == "count":
                v = float(len(vals))
            else:  # rate: summed value per second since the last emission
                dt = max(now - self._rate_mark[m.name], 1e-6)
                weight = sum(vals) if m.event_name == "mcts_step" else float(len(vals))
                v = weight / dt
                self._rate_mark[m.name] = now
            out.append((m, float(v), int(max(global_step, 0))))
            self._pending[m.name] = _Pending()
            self._last_log_step[m.name] = global_step
            self._last_log_time[m.name] = now
        return out
