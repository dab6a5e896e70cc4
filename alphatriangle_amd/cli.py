"""Command line interface (parity: alphatriangle/cli.py:31-326).

Commands ``train``, ``ml``, ``tb`` and ``ray`` keep their names and the four
reference options of ``train`` (``--log-level``, ``--seed``, ``--profile``,
``--run-name``).  ``train`` gains engine / topology options; ``ray`` explains that
no Ray cluster exists here and shows where the equivalent live information is.
"""

from __future__ import annotations

import logging
import shutil
import subprocess
import sys
from typing import Annotated

import typer
from rich.console import Console
from rich.panel import Panel
from rich.text import Text

from .config import APP_NAME, TrainConfig
from .logging_config import setup_logging
from .stats import PersistenceConfig, TrieyeConfig

app = typer.Typer(
    name="alphatriangle-amd",
    help="▲ AlphaTriangle on MI355X: AlphaZero self-play training for the triangle puzzle (headless).",
    add_completion=False,
    rich_markup_mode="markdown",
    pretty_exceptions_show_locals=False,
)
console = Console()

LogLevelOption = Annotated[
    str, typer.Option("--log-level", "-l", help="Logging level (DEBUG, INFO, WARNING, ERROR, CRITICAL).", case_sensitive=False)
]
SeedOption = Annotated[int, typer.Option("--seed", "-s", help="Random seed for reproducibility.")]
ProfileOption = Annotated[bool, typer.Option("--profile", help="cProfile worker 0 (.prof files under the run's profile_data/).", is_flag=True)]
RunNameOption = Annotated[str | None, typer.Option("--run-name", help="Name for the run (default: train_YYYYMMDD_HHMMSS).")]
HostOption = Annotated[str, typer.Option(help="Host IP address to bind the UI server to.")]
PortOption = Annotated[int, typer.Option(help="Port number for the UI server.")]


def _run_external_ui(command_name: str, command_args: list[str], ui_name: str, default_url: str) -> None:
    """Find and run an external UI tool; Ctrl-C exits cleanly."""
    logger = logging.getLogger(__name__)
    executable = shutil.which(command_name)
    if not executable:
        console.print(f"[bold red]Error:[/bold red] Could not find '{command_name}' executable in PATH.")
        console.print(f"Please ensure {ui_name} is installed and accessible.")
        raise typer.Exit(code=1)
    console.print(Panel(f"🚀 Launching [bold cyan]{ui_name}[/]\n   Command: {' '.join([executable, *command_args])}", title="External UI", border_style="blue", expand=False))
    try:
        proc = subprocess.run([executable, *command_args], check=False)
        if proc.returncode != 0:
            console.print(f"[bold red]{ui_name} exited with code {proc.returncode}.[/]")
            console.print(f"Is the port already in use? Default URL: {default_url}")
    except KeyboardInterrupt:
        console.print(f"\n[yellow]🟡 {ui_name} interrupted by user.[/]")
    except Exception as exc:
        logger.error("Error launching %s: %s", ui_name, exc)
        raise typer.Exit(code=1) from exc


@app.command()
def train(
    log_level: LogLevelOption = "INFO",
    seed: SeedOption = 42,
    profile: ProfileOption = False,
    run_name: RunNameOption = None,
    engine: Annotated[str, typer.Option(help="Self-play engine: auto, gpu (HBM-resident batched tree) or cpu.")] = "auto",
    games_per_gpu: Annotated[int | None, typer.Option(help="Concurrent self-play games per actor GPU.")] = None,
    simulations: Annotated[int | None, typer.Option(help="MCTS simulations per move.")] = None,
    max_steps: Annotated[int | None, typer.Option(help="Override MAX_TRAINING_STEPS.")] = None,
    workers: Annotated[int | None, typer.Option(help="CPU self-play workers (cpu engine).")] = None,
    preset: Annotated[str, typer.Option(help="Config preset: default, cnn, tiny, large, huge (20x30 board), reference.")] = "default",
    data_root: Annotated[str | None, typer.Option(help="Root directory for run data (default .trieye_data).")] = None,
    layout: Annotated[str, typer.Option(help="Multi-GPU role layout: colocated (every rank plays and learns) or split (actor ranks + learner ranks).")] = "colocated",
    learners: Annotated[int, typer.Option(help="Learner ranks of the split layout (the rest self-play).")] = 1,
    train_steps_per_round: Annotated[int | None, typer.Option(help="Split layout: learner steps between two polls for shipments and weight requests.")] = None,
    gpus: Annotated[int, typer.Option(help="Number of GPUs of this node to use: one process per GPU, started by this command.")] = 1,
    dtype: Annotated[str | None, typer.Option(help="Leaf-evaluation precision of the GPU engine: fp32 (reference precision) or bf16 (fast).")] = None,
    dist_backend: Annotated[str, typer.Option(help="Process-group backend: auto, nccl (= RCCL) or gloo (CPU rehearsals).")] = "auto",
) -> None:
    """Run the headless training pipeline.  `--gpus N` starts N ranks (one per GPU) from this one command."""
    from .parallel.launch import launch_local_ranks, under_launcher, world_size_from_env

    if gpus < 1:
        raise typer.BadParameter("--gpus must be at least 1")
    if not under_launcher():
        if gpus > 1:
            # no GPU call has been made in this process: start one fresh child per GPU and wait for them
            sys.exit(launch_local_ranks(gpus, [sys.executable, "-m", "alphatriangle_amd", *sys.argv[1:]]))
    elif gpus != world_size_from_env() and gpus != 1:
        raise typer.BadParameter(f"--gpus {gpus} does not match WORLD_SIZE={world_size_from_env()} of the launcher")

    from .parallel.dist import init_from_env
    from .presets import build_preset
    from .training.runner import run_training

    setup_logging(log_level)
    env_cfg, model_cfg, mcts_cfg, train_cfg, sys_cfg = build_preset(preset)
    train_cfg.RANDOM_SEED = seed
    train_cfg.PROFILE_WORKERS = profile
    if max_steps is not None:
        train_cfg = TrainConfig(**{**train_cfg.model_dump(), "MAX_TRAINING_STEPS": max_steps, "LR_SCHEDULER_T_MAX": None, "PER_BETA_ANNEAL_STEPS": None})
    if workers is not None:
        train_cfg.NUM_SELF_PLAY_WORKERS = workers
    if simulations is not None:
        mcts_cfg.max_simulations = simulations
    sys_cfg.ENGINE = engine  # type: ignore[assignment]
    if games_per_gpu is not None:
        sys_cfg.GAMES_PER_GPU = games_per_gpu
    if layout not in ("colocated", "split"):
        raise typer.BadParameter("--layout must be 'colocated' or 'split'")
    sys_cfg.ROLE_LAYOUT = layout  # type: ignore[assignment]
    sys_cfg.NUM_LEARNER_RANKS = learners
    if train_steps_per_round is not None:
        sys_cfg.TRAIN_STEPS_PER_ROUND = train_steps_per_round
    if dtype is not None:
        if dtype not in ("fp32", "bf16"):
            raise typer.BadParameter("--dtype must be 'fp32' or 'bf16'")
        sys_cfg.INFERENCE_DTYPE = dtype  # type: ignore[assignment]
    if dist_backend not in ("auto", "nccl", "gloo"):
        raise typer.BadParameter("--dist-backend must be auto, nccl or gloo")
    sys_cfg.DIST_BACKEND = dist_backend  # type: ignore[assignment]

    trieye_cfg = TrieyeConfig(app_name=APP_NAME, run_name=run_name or train_cfg.RUN_NAME)
    train_cfg.RUN_NAME = trieye_cfg.run_name
    trieye_cfg.persistence.RUN_NAME = trieye_cfg.run_name
    trieye_cfg.persistence.AUTO_RESUME_LATEST = train_cfg.AUTO_RESUME_LATEST
    trieye_cfg.persistence.LOAD_CHECKPOINT_PATH = train_cfg.LOAD_CHECKPOINT_PATH
    trieye_cfg.persistence.LOAD_BUFFER_PATH = train_cfg.LOAD_BUFFER_PATH
    if data_root:
        trieye_cfg.persistence.ROOT_DATA_DIR = data_root

    dist_ctx = init_from_env(sys_cfg)
    if dist_ctx is not None and dist_ctx.rank != 0:
        trieye_cfg.run_name = f"{trieye_cfg.run_name}_rank{dist_ctx.rank}"
        trieye_cfg.persistence.RUN_NAME = trieye_cfg.run_name

    console.print(
        Panel(
            f"Starting Training Run: '[bold cyan]{trieye_cfg.run_name}[/]'\nSeed: {seed}, Log Level: {log_level.upper()}, "
            f"Profiling: {'✅ Enabled' if profile else '❌ Disabled'}, Engine: {engine}, Preset: {preset}",
            title="[bold green]Training Setup[/]",
            border_style="green",
            expand=False,
        )
    )
    exit_code = run_training(
        log_level_str=log_level,
        train_config_override=train_cfg,
        trieye_config_override=trieye_cfg,
        profile=profile,
        model_config_override=model_cfg,
        mcts_config_override=mcts_cfg,
        env_config_override=env_cfg,
        system_config_override=sys_cfg,
        dist_context=dist_ctx,
    )
    if dist_ctx is not None:
        dist_ctx.shutdown()
    if exit_code == 0:
        console.print(Panel(f"✅ Training run '[bold cyan]{trieye_cfg.run_name}[/]' completed successfully.", title="[bold green]Training Finished[/]", border_style="green"))
    else:
        console.print(Panel(f"❌ Training run '[bold cyan]{trieye_cfg.run_name}[/]' failed with exit code {exit_code}.", title="[bold red]Training Failed[/]", border_style="red"))
    sys.exit(exit_code)


@app.command()
def ml(host: HostOption = "127.0.0.1", port: PortOption = 5000) -> None:
    """Launch the MLflow UI on this project's file store (`.trieye_data/<app>/mlruns`)."""
    setup_logging("INFO")
    persist = PersistenceConfig(APP_NAME=APP_NAME)
    mlruns = persist.get_mlflow_abs_path()
    if not mlruns.exists() or not any(mlruns.iterdir()):
        console.print(f"[yellow]Warning:[/yellow] MLflow directory not found or empty at: [dim]{mlruns}[/]")
        console.print("Run `train` first to generate data.")
    _run_external_ui("mlflow", ["ui", "--backend-store-uri", persist.MLFLOW_TRACKING_URI, "--host", host, "--port", str(port)], "MLflow UI", f"http://{host}:{port}")


@app.command()
def tb(host: HostOption = "127.0.0.1", port: PortOption = 6006) -> None:
    """Launch TensorBoard on this project's runs directory (`.trieye_data/<app>/runs`)."""
    setup_logging("INFO")
    persist = PersistenceConfig(APP_NAME=APP_NAME)
    runs = persist.get_runs_root_dir()
    if not runs.exists() or not any(runs.iterdir()):
        console.print(f"[yellow]Warning:[/yellow] runs directory not found or empty at: [dim]{runs}[/]")
        console.print("Run `train` first to generate data.")
    _run_external_ui("tensorboard", ["--logdir", str(runs), "--host", host, "--port", str(port)], "TensorBoard UI", f"http://{host}:{port}")


@app.command()
def ray(host: HostOption = "127.0.0.1", port: PortOption = 8265) -> None:
    """Where the process-level dashboard went: this framework runs without a Ray cluster."""
    setup_logging("INFO")
    persist = PersistenceConfig(APP_NAME=APP_NAME)
    text = Text.from_markup(
        "This framework replaces Ray workers by GPU-resident actors (one process per GPU).\n"
        f"There is no Ray dashboard on http://{host}:{port}. Equivalent live information:\n"
        "  • worker / buffer / throughput gauges: `System/*`, `Buffer/Size`, `Rate/*` metrics "
        "(`alphatriangle-amd tb` or `alphatriangle-amd ml`)\n"
        f"  • per-run metric stream: [dim]{persist.get_runs_root_dir()}/<run>/metrics.jsonl[/]\n"
        "  • GPU utilisation: `rocm-smi`, kernel timelines: `rocprofv3 --kernel-trace --stats -- python bench.py`"
    )
    console.print(Panel(text, title="[bold yellow]Process dashboard[/]", border_style="yellow", expand=False))


@app.command()
def info() -> None:
    """Print the validated default configuration and the native-library status."""
    from . import native
    from .config import print_config_info_and_validate

    print_config_info_and_validate()
    console.print(f"host runtime: {native.host().__file__}")
    console.print(f"HIP kernel library: {native.hip_library_path()} ({'present' if native.hip_library_path().exists() else 'missing'})")


if __name__ == "__main__":
    app()
