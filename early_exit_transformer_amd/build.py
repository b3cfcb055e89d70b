"""Builds csrc/libeec.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# EEC_LIB_PATH selects an alternative build of the same library (tuning experiments only)
LIB_PATH = os.environ.get("EEC_LIB_PATH") or os.path.join(CSRC, "libeec.so")


def build_library(jobs: int = 6, verbose: bool = False) -> str:
    """make -C csrc; returns the path of the shared library."""
    res = subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building libeec.so failed (see output above)")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(verbose=True))
