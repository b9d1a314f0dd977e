"""Crash-safe writers for the on-disk formats of the path.

Mirrors ``caliscope/persistence.py:27-52``: data goes to ``<name>.tmp``, is flushed and fsync'ed, then atomically renamed
over the target.  ``rtoml`` is not available here, so the fixed schema of ``camera_array.toml`` is emitted by a small TOML
writer (tables of scalars / nested numeric lists — everything ``CameraArray.from_toml`` and the reference's reader need).
"""

from __future__ import annotations

import os
from pathlib import Path
from typing import Any

import numpy as np
import pandas as pd

CSV_FLOAT_PRECISION = "%.6f"  # micron precision at metre scale (reference persistence.py:27)


class PersistenceError(Exception):
    """Raised when a file of the capture-volume formats cannot be written or read."""


def safe_write_text(text: str, path: Path) -> None:
    path = Path(path)
    tmp = path.with_suffix(path.suffix + ".tmp")
    with open(tmp, "w", encoding="utf-8", newline="") as fh:
        fh.write(text)
        fh.flush()
        os.fsync(fh.fileno())
    os.replace(tmp, path)


def safe_write_csv(df: pd.DataFrame, path: Path, **kwargs: Any) -> None:
    path = Path(path)
    tmp = path.with_suffix(path.suffix + ".tmp")
    with open(tmp, "w", newline="", encoding="utf-8") as fh:
        df.to_csv(fh, **kwargs)
        fh.flush()
        os.fsync(fh.fileno())
    os.replace(tmp, path)


def _value(v: Any) -> str:
    if isinstance(v, (bool, np.bool_)):
        return "true" if v else "false"
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    if isinstance(v, (float, np.floating)):
        f = float(v)
        if np.isnan(f):
            return "nan"
        if np.isinf(f):
            return "inf" if f > 0 else "-inf"
        r = repr(f)
        return r if any(ch in r for ch in ".en") else r + ".0"
    if isinstance(v, str):
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    if isinstance(v, np.ndarray):
        v = v.tolist()
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(_value(x) for x in v) + "]"
    raise TypeError(f"cannot serialise {type(v).__name__} to TOML")


def dumps_toml(data: dict) -> str:
    """Serialise nested dicts of scalars / lists: top-level scalars first, then ``[a.b]`` tables depth-first, then
    ``[[a.b]]`` arrays of tables (non-empty lists of dicts)."""
    lines: list[str] = []

    def is_aot(v: Any) -> bool:
        return isinstance(v, (list, tuple)) and len(v) > 0 and all(isinstance(x, dict) for x in v)

    def emit(table: dict, prefix: str) -> None:
        scalars = {k: v for k, v in table.items() if not isinstance(v, dict) and not is_aot(v)}
        tables = {k: v for k, v in table.items() if isinstance(v, dict)}
        arrays = {k: v for k, v in table.items() if is_aot(v)}
        if prefix and (scalars or not tables):
            lines.append(f"[{prefix}]")
        for k, v in scalars.items():
            lines.append(f"{_key(k)} = {_value(v)}")
        if scalars or (prefix and not tables):
            lines.append("")
        for k, v in tables.items():
            emit(v, f"{prefix}.{_key(k)}" if prefix else _key(k))
        for k, items in arrays.items():
            name = f"{prefix}.{_key(k)}" if prefix else _key(k)
            for item in items:
                if any(isinstance(x, dict) or is_aot(x) for x in item.values()):
                    raise TypeError("nested tables inside an array of tables are not supported")
                lines.append(f"[[{name}]]")
                for ik, iv in item.items():
                    lines.append(f"{_key(ik)} = {_value(iv)}")
                lines.append("")

    def _key(k: Any) -> str:
        k = str(k)
        return k if k.replace("_", "").replace("-", "").isalnum() else _value(k)

    emit(data, "")
    return "\n".join(lines).rstrip("\n") + "\n"


def safe_write_toml(data: dict, path: Path) -> None:
    safe_write_text(dumps_toml(data), Path(path))
