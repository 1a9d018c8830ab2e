"""CPU sanitizer builds (the reference has none; GPU sanitizers are not available on this
pool, so host code only).  tests/san/Makefile builds
  * the product's host graph (csrc/host_graph.cpp) + a driver under AddressSanitizer + UBSan: write path, dump round
    trips, ~3000 corrupted / truncated dumps through HostGraph::load,
  * the same driver's multi-threaded builder under ThreadSanitizer (build_parallel: per-node spin locks, the
    entry-point pair read under the global lock),
  * the oracle (oracle/mvdb_oracle.c) + a driver under AddressSanitizer + UBSan.
A sanitizer report makes the binary exit non-zero; none may appear."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "san")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", SAN, "all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(SAN, "_build")


@pytest.mark.parametrize("binary,args,ok", [("host_graph_asan", [], "host_graph driver ok"),
                                            ("oracle_asan", [], "oracle driver ok"),
                                            ("host_graph_tsan", ["threads", "8"], "host_graph threads driver ok")])
def test_driver_runs_clean(built, binary, args, ok):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([os.path.join(built, binary)] + args, capture_output=True, text=True, timeout=600, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and ok in out, out[-3000:]
    assert "Sanitizer" not in out and "runtime error" not in out, out[-3000:]
