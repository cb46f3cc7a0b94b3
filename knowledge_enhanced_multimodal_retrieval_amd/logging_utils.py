"""Host-side logging and result files.  The three public names and their arguments are the reference's
(src/clip/utils/logging_utils.py:11-54) because its evaluators import them; the result-JSON schema written through
``save_metrics_to_json`` is the evaluators' (evaluator.py:379-387)."""
from __future__ import annotations

import json
import logging
import os
from typing import Any, Dict, Optional

_FORMAT = "%(asctime)s - %(name)s - %(levelname)s - %(message)s"
_DATEFMT = "%Y-%m-%d %H:%M:%S"


def _ensure_parent(path: str) -> str:
    parent = os.path.dirname(os.path.abspath(path))
    os.makedirs(parent, exist_ok=True)
    return path


def _dump(obj: Dict[str, Any], path: str, mode: str, **json_kwargs) -> None:
    with open(_ensure_parent(path), mode, encoding="utf-8") as fh:
        fh.write(json.dumps(obj, **json_kwargs))
        if mode == "a":
            fh.write("\n")


def setup_logger(name: str, log_file: Optional[str] = None, level: int = logging.INFO) -> logging.Logger:
    """A logger that writes to stderr and, when ``log_file`` is given, to that file too; calling it again replaces the
    handlers instead of stacking them."""
    log = logging.getLogger(name)
    for old in list(log.handlers):
        log.removeHandler(old)
    sinks = [logging.StreamHandler()] + ([logging.FileHandler(_ensure_parent(log_file))] if log_file else [])
    formatter = logging.Formatter(_FORMAT, datefmt=_DATEFMT)
    for sink in sinks:
        sink.setFormatter(formatter)
        sink.setLevel(level)
        log.addHandler(sink)
    log.setLevel(level)
    return log


def log_metrics_to_jsonl(metrics: Dict[str, Any], output_file: str) -> None:
    """Append one JSON object per call (a line per evaluation)."""
    _dump(metrics, output_file, "a")


def save_metrics_to_json(metrics: Dict[str, Any], output_file: str) -> None:
    """Write (replace) a pretty-printed result file."""
    _dump(metrics, output_file, "w", indent=2)
