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
    cmd += ["-Rpass-analysis=kernel-resource-usage"]  # -> lib/kernel_resources.txt (see check_resources)
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
        sys.stderr.write("".join(l for l in r.stderr.splitlines(True) if "kernel-resource-usage" not in l))
    if out is None:
        with open(RESOURCES, "w") as f:
            f.write(_resource_table(r.stderr))
        check_resources()
    return target


RESOURCES = os.path.join(LIB_DIR, "kernel_resources.txt")


def _resource_table(stderr: str) -> str:
    """hipcc's per-kernel remarks -> one line per kernel: name VGPRs AGPRs scratch occupancy."""
    import re
    rows, cur = [], None
    for line in stderr.splitlines():
        m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = [m.group(2)]
            rows.append(cur)
        elif cur is not None:
            cur.append(m.group(2))
    return "".join(" ".join(r) + "\n" for r in rows if len(r) == 5)


def check_resources() -> None:
    """The lean traversal is latency-bound: its throughput is waves resident x 1 / chain time.  At 384-512 B rows it
    sits close to 256 VGPRs, where one more live value halves the residency (2 -> 1 waves per SIMD) without any test
    noticing.  Refuse a build in which that happened, or in which the kernel spills."""
    import re
    if not os.path.exists(RESOURCES):
        return
    for line in open(RESOURCES):
        name, vgpr, agpr, scratch, occ = line.split()
        m = re.match(r"_ZN4mvdb20traverse_lean_kernelILi(\d)ELb([01])EEEvNS_12TraverseArgsE", name)
        if not m:
            continue
        j, full = int(m.group(1)), m.group(2) == "1"   # rows of up to j * 128 bytes; full: exactly that many
        # the benchmarked shapes (384- and 512-byte rows) strictly; ragged row sizes may spill a little, and 385..511-byte
        # rows (j = 4, not full) run with one wave per SIMD
        if (j <= 3 and int(occ) < 2) or (full and j <= 4 and int(occ) < 2) or (full and j <= 3 and int(scratch) > 0):
            raise RuntimeError(f"{name}: {vgpr} VGPRs + {agpr} AGPRs, {scratch} B scratch, {occ} wave(s) per SIMD -- "
                               "the lean traversal must keep 2 waves per SIMD and, for full 128/256/384-byte rows, no scratch (see build.py)")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
