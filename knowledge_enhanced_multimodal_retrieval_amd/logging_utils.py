"""Logger / JSON result helpers with the reference's names (src/clip/utils/logging_utils.py:11-54); the result JSON
schema written by the evaluators is the reference's (evaluator.py:379-387)."""
import json
import logging
from pathlib import Path
from typing import Any, Dict


def setup_logger(name: str, log_file: str = None, level=logging.INFO):
    logger = logging.getLogger(name)
    logger.setLevel(level)
    logger.handlers = []
    fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
    handlers = [logging.StreamHandler()]
    if log_file:
        Path(log_file).parent.mkdir(parents=True, exist_ok=True)
        handlers.append(logging.FileHandler(log_file))
    for h in handlers:
        h.setLevel(level)
        h.setFormatter(fmt)
        logger.addHandler(h)
    return logger


def log_metrics_to_jsonl(metrics: Dict[str, Any], output_file: str):
    Path(output_file).parent.mkdir(parents=True, exist_ok=True)
    with open(output_file, "a", encoding="utf-8") as f:
        f.write(json.dumps(metrics) + "\n")


def save_metrics_to_json(metrics: Dict[str, Any], output_file: str):
    Path(output_file).parent.mkdir(parents=True, exist_ok=True)
    with open(output_file, "w", encoding="utf-8") as f:
        json.dump(metrics, f, indent=2)
