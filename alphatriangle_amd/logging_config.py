"""Central console / file logging setup (parity: alphatriangle/logging_config.py:7-104)."""

from __future__ import annotations

import logging
import sys
from pathlib import Path

_RESET = "\x1b[0m"
_COLORS = {
    logging.DEBUG: "\x1b[38;20m",
    logging.INFO: "\x1b[32;20m",
    logging.WARNING: "\x1b[33;20m",
    logging.ERROR: "\x1b[31;20m",
    logging.CRITICAL: "\x1b[31;1m",
}
_BASE = "▲ %(asctime)s [%(levelname)s] %(name)s: %(message)s"
_DEBUG = "▲ %(asctime)s [%(levelname)s] %(name)s (%(filename)s:%(lineno)d): %(message)s"


class CustomFormatter(logging.Formatter):
    """Triangle-prefixed, colour-per-level console formatter."""

    def format(self, record: logging.LogRecord) -> str:
        fmt = _DEBUG if record.levelno == logging.DEBUG else _BASE
        color = _COLORS.get(record.levelno, "")
        return logging.Formatter(color + fmt + _RESET, datefmt="%H:%M:%S").format(record)


def setup_logging(log_level_str: str = "INFO", log_file: str | Path | None = None) -> logging.FileHandler | None:
    """Configure the root logger: colour console handler plus optional plain file handler."""
    level = getattr(logging, str(log_level_str).upper(), logging.INFO)
    root = logging.getLogger()
    root.setLevel(level)
    for handler in list(root.handlers):
        root.removeHandler(handler)
        try:
            handler.close()
        except Exception:  # pragma: no cover
            pass
    console = logging.StreamHandler(sys.stdout)
    console.setLevel(level)
    console.setFormatter(CustomFormatter())
    root.addHandler(console)

    file_handler: logging.FileHandler | None = None
    if log_file is not None:
        try:
            path = Path(log_file)
            path.parent.mkdir(parents=True, exist_ok=True)
            file_handler = logging.FileHandler(path, mode="a", encoding="utf-8")
            file_handler.setLevel(level)
            file_handler.setFormatter(
                logging.Formatter("%(asctime)s [%(levelname)s] %(name)s: %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
            )
            root.addHandler(file_handler)
        except OSError as exc:
            root.error("Could not open log file %s: %s", log_file, exc)
            file_handler = None

    for noisy, lvl in (("matplotlib", logging.WARNING), ("PIL", logging.WARNING), ("torch", logging.WARNING)):
        logging.getLogger(noisy).setLevel(lvl)
    root.info("Logging configured (level %s)%s", logging.getLevelName(level), f", file {log_file}" if file_handler else "")
    return file_handler
