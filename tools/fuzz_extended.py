"""Round-end evidence: tests/test_gpu_fuzz.py's random sweep (random graphs x random (k, ef), sorted-list and fused paths,
ids / distances / counters against the oracle, bit for bit) over many more seeds than the test suite runs.
    python tools/fuzz_extended.py [first_seed] [n_seeds] [seconds]   -> one line per 100 seeds, a summary line at the end"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import test_random_graphs_sorted_and_fused as one

if __name__ == "__main__":
    if os.environ.get("MVDB_RELABEL_ALWAYS"):   # every graph of the sweep walked under breadth-first labels (mvdb_set_relabel mode 2)
        from minivectordb_amd import HipBridge
        _init = HipBridge.init
        def init(self, *a, **k):
            r = _init(self, *a, **k)
            self.setRelabel(2)
            return r
        HipBridge.init = init
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
    t0, done, failed = time.time(), 0, []
    for seed in range(first, first + count):
        if time.time() - t0 > budget:
            break
        try:
            one(None, seed)
        except AssertionError as e:
            failed.append((seed, str(e)[:200]))
            print(f"FAILED seed {seed}: {e}", flush=True)
        done += 1
        if done % 100 == 0:
            print(f"{done} seeds ({first}..{seed}) in {time.time() - t0:.0f} s, {len(failed)} failed", flush=True)
    print(f"fuzz_extended: seeds {first}..{first + done - 1}: {done - len(failed)} bit-exact, {len(failed)} failed, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if failed else 0)
