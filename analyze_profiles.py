"""Inspect what ``train --profile`` wrote into ``<run>/profile_data/``: a cProfile ``.prof`` file of a CPU worker
(parity: analyze_profiles.py:41-78) or the per-move GPU phase timings of a device-resident actor
(``gpu_actor_rank<r>.jsonl``, written by alphatriangle_amd/rl/self_play/gpu_profile.py)."""

from __future__ import annotations

import pstats
from pathlib import Path
from typing import Annotated

import typer

app = typer.Typer(name="analyze-profiles", help="Analyze profile artefacts: cProfile .prof files and GPU-actor .jsonl files.", add_completion=False)

ProfileArg = Annotated[Path, typer.Argument(help="Path to the .prof file (CPU worker) or gpu_actor_rank<r>.jsonl (GPU actor).", exists=True, file_okay=True, dir_okay=False, readable=True, resolve_path=True)]
LinesOpt = Annotated[int, typer.Option("--lines", "-n", help="Number of lines to show per table.")]


@app.command()
def analyze(profile_file: ProfileArg, num_lines: LinesOpt = 30) -> None:
    """Print the top functions by cumulative time and by own time (.prof), or the per-phase GPU times (.jsonl)."""
    if profile_file.suffix == ".jsonl":
        from alphatriangle_amd.rl.self_play.gpu_profile import format_summary, summarise

        try:
            print(format_summary(summarise(profile_file)))
        except Exception as exc:
            print(f"Error loading profile file: {exc}")
            raise typer.Exit(code=1) from exc
        return
    print(f"\n--- Analyzing Profile: {profile_file.name} ---")
    try:
        stats = pstats.Stats(str(profile_file))
    except Exception as exc:
        print(f"Error loading profile file: {exc}")
        raise typer.Exit(code=1) from exc
    stats.strip_dirs()
    print(f"\n--- Top {num_lines} by Cumulative Time ---")
    stats.sort_stats(pstats.SortKey.CUMULATIVE).print_stats(num_lines)
    print(f"\n--- Top {num_lines} by Total Time (tottime) ---")
    stats.sort_stats(pstats.SortKey.TIME).print_stats(num_lines)
    print("\n--- Analysis Complete ---")


if __name__ == "__main__":
    app()
