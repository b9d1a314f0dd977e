"""Build libcaliscope_ba.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m caliscope_amd.build [--force]
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "libcaliscope_ba.so"
SOURCES = [CSRC / "cba_lib.hip", CSRC / "cba_solve.cpp"]
DEPENDS = [CSRC / "cba_kernels.h", CSRC / "schur_plan.h", CSRC / "host_plan.h", CSRC / "wg_binding.h", CSRC / "ba_math.h", CSRC / "trf_math.h", HERE.parent / "include" / "caliscope_ba.h"]
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def needs_build() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + DEPENDS)


def source_digest() -> str:
    """sha256 over the sources the library is built from (names and bytes, fixed order): what identifies "the final library" in the committed
    measurements (profiles/parity_r*.json and r*_end_bench.json must carry the same one, tests/test_bench_contract.py)."""
    import hashlib

    h = hashlib.sha256()
    for path in sorted(SOURCES + DEPENDS, key=lambda q: q.name):
        h.update(path.name.encode())
        h.update(path.read_bytes())
    return h.hexdigest()


def built_digest(path: Path = OUT) -> str | None:
    """The source digest the library at `path` was compiled with (the string csrc/cba_lib.hip keeps in the binary), or None."""
    import re

    m = re.search(rb"CBA_SOURCE_DIGEST=([0-9a-f]{64})", path.read_bytes())
    return m.group(1).decode() if m else None


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build() and built_digest() == source_digest():
        return OUT
    cmd = [
        hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC", "-munsafe-fp-atomics",
        "-Wall", "-Wno-unused-function", f'-DCBA_SOURCE_DIGEST="{source_digest()}"', *map(str, SOURCES), "-o", str(OUT), "-pthread", "-lrccl",
        "-lrocprofiler-sdk-roctx",
    ]
    if verbose:
        print("[caliscope_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
