import re

from typer.testing import CliRunner

from alphatriangle_amd.cli import app

runner = CliRunner()


def test_help_lists_reference_commands():
    res = runner.invoke(app, ["--help"])
    assert res.exit_code == 0
    for cmd in ("train", "ml", "tb", "ray"):
        assert cmd in res.output
    res = runner.invoke(app, ["train", "--help"])
    out = re.sub(r"\x1b\[[0-9;]*m", "", res.output)
    for opt in ("--log-level", "--seed", "--profile", "--run-name"):
        assert opt in out


def test_ray_command_explains_replacement():
    res = runner.invoke(app, ["ray"])
    assert res.exit_code == 0 and "Ray" in res.output


def test_ml_and_tb_fail_cleanly_without_tools(monkeypatch, tmp_path):
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr("alphatriangle_amd.cli.shutil.which", lambda _name: None)
    for cmd in ("ml", "tb"):
        res = runner.invoke(app, [cmd])
        assert res.exit_code == 1 and "Could not find" in res.output


def test_train_tiny_preset_through_cli(tmp_path):
    res = runner.invoke(app, ["train", "--preset", "tiny", "--max-steps", "12", "--data-root", str(tmp_path), "--run-name", "cli", "--seed", "3", "--log-level", "WARNING"])
    assert res.exit_code == 0, res.output
    assert (tmp_path / "AlphaTriangle" / "runs" / "cli" / "checkpoints" / "latest.pkl").exists()


def test_analyze_profiles(tmp_path):
    import cProfile

    import analyze_profiles

    prof = cProfile.Profile()
    prof.enable()
    sum(range(1000))
    prof.disable()
    path = tmp_path / "w.prof"
    prof.dump_stats(str(path))
    res = CliRunner().invoke(analyze_profiles.app, [str(path), "-n", "5"])
    assert res.exit_code == 0 and "Cumulative Time" in res.output


def test_analyze_profiles_summarises_a_gpu_actor_profile(tmp_path):
    """The --profile artefact of the GPU path: one JSON record per move; the analyzer prints per-phase statistics."""
    import json

    import analyze_profiles
    from alphatriangle_amd.rl.self_play.gpu_profile import summarise

    path = tmp_path / "gpu_actor_rank0.jsonl"
    with open(path, "w") as fh:
        for i in range(20):
            fh.write(json.dumps({"move": i, "rank": 0, "root_eval_ms": 2.0, "waves_ms": 90.0 + i, "commit_ms": 3.0, "learner_ms": 14.0,
                                 "move_ms": 95.0 + i, "games_moved": 4096, "simulations": 4096 * 400, "leaf_share": 0.9}) + "\n")
    s = summarise(path)
    assert s["moves"] == 20 and set(s["phases"]) == {"root_eval", "waves", "commit", "learner", "move"}
    assert abs(s["phases"]["waves"]["mean_ms"] - 99.5) < 1e-9 and s["phases"]["waves"]["p95_ms"] == 109.0
    assert 0.9 < s["phases"]["waves"]["share_of_move"] < 1.0
    assert abs(s["moves_per_second"] - 4096 * 20 / (sum(95.0 + i for i in range(20)) / 1e3)) < 1e-6
    res = CliRunner().invoke(analyze_profiles.app, [str(path)])
    assert res.exit_code == 0, res.output
    assert "GPU actor profile" in res.output and "waves" in res.output and "learner" in res.output
