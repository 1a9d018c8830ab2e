#!/bin/bash
# usage: tools/ab_both_sets.sh <variant names...>: bench.py (i.i.d. + clustered set) per variant library
mkdir -p gpurun_out
for v in "$@"; do
  MVDB_HIP_LIB=$PWD/minivectordb_amd/lib/libmvdb_hip_$v.so timeout -k 10 300 python bench.py --cpu-sample 0 --no-pcie --sustain-s 0 > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err || exit 1
  python -c "
import json,sys; d=json.loads(open('gpurun_out/bench_$v.json').read()); print('$v', 'iid', round(d['value']), 'solo ms', d['roofline']['solo_avg_launch_ms'], 'clustered', round(d['second_dataset']['value']))"
done
