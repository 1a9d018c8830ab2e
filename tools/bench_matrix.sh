#!/bin/bash
# usage: tools/bench_matrix.sh <n> ; runs bench.py variants on one index size (throughput exploration)
n=${1:-200000}
run() { echo "== $*"; timeout -k 10 300 python bench.py --n $n --cpu-sample 0 --no-recall "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" || exit 1; }
run --in-flight 2
run --in-flight 3
run --in-flight 4
run --in-flight 1
run --in-flight 2 --traverse-waves 1
run --in-flight 1 --traverse-waves 1
run --in-flight 2 --nq 2048
run --in-flight 2 --nq 2048 --traverse-waves 1
