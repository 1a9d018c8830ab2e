"""Build libmvdb_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libmvdb_hip.so")
SOURCES = ["engine.hip", "host_graph.cpp"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "mvdb_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, defines=(), out: str | None = None) -> str:
    """Build the library.  `defines`/`out` build a tuning variant (tools/ only) beside it."""
    target = LIB if out is None else os.path.join(LIB_DIR, out)
    if out is None and not force and not _stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
           "-fno-fast-math", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-x", "hip"]
    cmd += ["-D" + d for d in defines]
    cmd += [os.path.join(CSRC, f) for f in SOURCES]
    cmd += ["-o", target]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libmvdb_hip.so")
    if verbose and r.stderr:
        sys.stderr.write(r.stderr)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
