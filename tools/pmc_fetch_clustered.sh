#!/bin/bash
# FETCH_SIZE pass of the traversal on the clustered set, shipped library and the always-rows-with-visited variant
# (tools/build_variants.py noadapt:MVDB_DUO_SPEC_ADAPT=0).  Usage on the GPU box: bash tools/pmc_fetch_clustered.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
export PMC_CLUSTERED=1 PMC_N=1000000
cd /tmp && export TMPDIR=/tmp
for v in shipped noadapt; do
  if [ $v = shipped ]; then unset MVDB_HIP_LIB; else export MVDB_HIP_LIB=$REPO/minivectordb_amd/lib/libmvdb_hip_$v.so; fi
  rm -rf "$REPO/gpurun_out/pmcc_$v"
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmcc_$v" -- \
      python3 "$REPO/tools/pmc_run.py" > "$REPO/gpurun_out/pmcc_$v.log" 2>&1 || { echo "pass $v failed"; tail -5 "$REPO/gpurun_out/pmcc_$v.log"; exit 1; }
  tail -1 "$REPO/gpurun_out/pmcc_$v.log"
done
