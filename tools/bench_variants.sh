#!/bin/bash
# usage: tools/bench_variants.sh <n> <bench args...> -- <variant names>; runs bench.py per variant library
n=$1; shift
mkdir -p gpurun_out  # bench progress goes to a file there: a silent run is taken to be hung
args=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done
shift
for v in default "$@"; do
  if [ "$v" = default ]; then unset MVDB_HIP_LIB; else export MVDB_HIP_LIB=$PWD/minivectordb_amd/lib/libmvdb_hip_$v.so; fi
  for fl in 1 3; do
    echo "== $v in-flight $fl"
    timeout -k 10 300 python bench.py --n $n --cpu-sample 0 --no-recall --no-pcie --sustain-s 0 --second-dataset none --in-flight $fl "${args[@]}" 2>>gpurun_out/bench_variants.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('qps', d['value'], 'ms/step', d['ms_per_step'], 'trav launch ms', d['roofline']['avg_launch_ms'])" || exit 1
  done
done
