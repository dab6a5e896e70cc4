"""Dependency-free metric sinks.

* :class:`JsonlSink`        - one JSON object per logged metric value.
* :class:`TensorBoardSink`  - writes genuine ``events.out.tfevents.*`` files (hand-encoded
  protobuf + TFRecord framing) readable by ``tensorboard --logdir``.
* :class:`MlflowFileSink`   - writes the MLflow *file store* layout (``mlruns/<exp>/<run>/
  {metrics,params,tags,meta.yaml}``) so ``mlflow ui --backend-store-uri`` can browse runs
  without the ``mlflow`` package being importable at training time.

The reference logs to MLflow and TensorBoard through the external ``trieye`` actor
(README.md:18-24,50-51); neither package is a hard dependency here.
"""

from __future__ import annotations

import json
import os
import socket
import struct
import threading
import time
import uuid
from pathlib import Path
from typing import Any


class MetricSink:
    name = "sink"

    def log_metric(self, name: str, value: float, step: int, wall_time: float) -> None:  # pragma: no cover
        raise NotImplementedError

    def log_params(self, params: dict[str, Any]) -> None:
        pass

    def flush(self) -> None:
        pass

    def close(self) -> None:
        pass


class JsonlSink(MetricSink):
    name = "jsonl"

    def __init__(self, path: Path):
        path.parent.mkdir(parents=True, exist_ok=True)
        self.path = path
        self._fh = open(path, "a", encoding="utf-8")
        self._lock = threading.Lock()

    def log_metric(self, name: str, value: float, step: int, wall_time: float) -> None:
        with self._lock:
            self._fh.write(json.dumps({"name": name, "value": value, "step": step, "time": wall_time}) + "\n")

    def flush(self) -> None:
        with self._lock:
            self._fh.flush()

    def close(self) -> None:
        with self._lock:
            if not self._fh.closed:
                self._fh.flush()
                self._fh.close()


# ---------------------------------------------------------------- TensorBoard event files
def _crc32c_table() -> list[int]:
    poly = 0x82F63B78
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        table.append(c)
    return table


_CRC_TABLE = _crc32c_table()


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _len_delim(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_scalar_event(tag: str, value: float, step: int, wall_time: float) -> bytes:
    val = _len_delim(1, tag.encode("utf-8")) + _varint((2 << 3) | 5) + struct.pack("<f", value)
    summary = _len_delim(1, val)
    return _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _varint((2 << 3) | 0) + _varint(step) + _len_delim(5, summary)


def encode_version_event(wall_time: float) -> bytes:
    return _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _len_delim(3, b"brain.Event:2")


def frame_record(data: bytes) -> bytes:
    header = struct.pack("<Q", len(data))
    return header + struct.pack("<I", _masked_crc(header)) + data + struct.pack("<I", _masked_crc(data))


class TensorBoardSink(MetricSink):
    name = "tensorboard"

    def __init__(self, log_dir: Path):
        log_dir.mkdir(parents=True, exist_ok=True)
        fname = f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}.atri"
        self.path = log_dir / fname
        self._fh = open(self.path, "ab")
        self._lock = threading.Lock()
        self._fh.write(frame_record(encode_version_event(time.time())))

    def log_metric(self, name: str, value: float, step: int, wall_time: float) -> None:
        with self._lock:
            self._fh.write(frame_record(encode_scalar_event(name, float(value), int(step), wall_time)))

    def flush(self) -> None:
        with self._lock:
            self._fh.flush()

    def close(self) -> None:
        with self._lock:
            if not self._fh.closed:
                self._fh.flush()
                self._fh.close()


# ---------------------------------------------------------------- MLflow file store
def _yaml_scalar(v: Any) -> str:
    if v is None:
        return "null"
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (int, float)):
        return str(v)
    return "'" + str(v).replace("'", "''") + "'"


def _safe(name: str) -> str:
    return name.replace(" ", "_")


class MlflowFileSink(MetricSink):
    name = "mlflow"

    def __init__(self, mlruns_dir: Path, experiment_name: str, run_name: str):
        self.root = mlruns_dir
        self.root.mkdir(parents=True, exist_ok=True)
        self.experiment_id = self._ensure_experiment(experiment_name)
        self.run_id = uuid.uuid4().hex
        self.run_name = run_name
        self.run_dir = self.root / self.experiment_id / self.run_id
        for sub in ("metrics", "params", "tags", "artifacts"):
            (self.run_dir / sub).mkdir(parents=True, exist_ok=True)
        self.start_ms = int(time.time() * 1000)
        self._write_run_meta(status=1, end_time=None)
        (self.run_dir / "tags" / "mlflow.runName").write_text(run_name)
        self._lock = threading.Lock()
        self._buffers: dict[str, list[str]] = {}

    def _ensure_experiment(self, name: str) -> str:
        for child in sorted(self.root.iterdir()):
            meta = child / "meta.yaml"
            if child.is_dir() and meta.exists() and f"name: '{name}'" in meta.read_text():
                return child.name
        ids = [int(c.name) for c in self.root.iterdir() if c.is_dir() and c.name.isdigit()]
        exp_id = str(max(ids) + 1 if ids else 0)
        exp_dir = self.root / exp_id
        exp_dir.mkdir(parents=True, exist_ok=True)
        now = int(time.time() * 1000)
        (exp_dir / "meta.yaml").write_text(
            f"artifact_location: {_yaml_scalar(exp_dir.as_uri())}\ncreation_time: {now}\nexperiment_id: '{exp_id}'\n"
            f"last_update_time: {now}\nlifecycle_stage: active\nname: '{name}'\n"
        )
        return exp_id

    def _write_run_meta(self, status: int, end_time: int | None) -> None:
        art = (self.run_dir / "artifacts").as_uri()
        (self.run_dir / "meta.yaml").write_text(
            f"artifact_uri: {_yaml_scalar(art)}\nend_time: {_yaml_scalar(end_time)}\nentry_point_name: ''\n"
            f"experiment_id: '{self.experiment_id}'\nlifecycle_stage: active\nrun_id: {self.run_id}\n"
            f"run_name: {_yaml_scalar(self.run_name)}\nrun_uuid: {self.run_id}\nsource_name: ''\nsource_type: 4\n"
            f"source_version: ''\nstart_time: {self.start_ms}\nstatus: {status}\ntags: []\nuser_id: atri\n"
        )

    def log_metric(self, name: str, value: float, step: int, wall_time: float) -> None:
        with self._lock:
            self._buffers.setdefault(name, []).append(f"{int(wall_time * 1000)} {value} {step}\n")

    def log_params(self, params: dict[str, Any]) -> None:
        for k, v in params.items():
            p = self.run_dir / "params" / _safe(str(k))
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_text(str(v)[:6000])

    def flush(self) -> None:
        with self._lock:
            bufs, self._buffers = self._buffers, {}
        for name, lines in bufs.items():
            p = self.run_dir / "metrics" / _safe(name)
            p.parent.mkdir(parents=True, exist_ok=True)
            with open(p, "a", encoding="utf-8") as fh:
                fh.writelines(lines)

    def close(self) -> None:
        self.flush()
        self._write_run_meta(status=3, end_time=int(time.time() * 1000))
