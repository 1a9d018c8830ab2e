#!/bin/bash
# Counter passes of the traversal kernel, one rocprofv3 run per pass (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2;
# SQ has 8).  Usage on the GPU box:
#   bash tools/pmc_passes.sh traffic [N]   HBM traffic  -> python tools/pmc.py traffic gpurun_out profiles/rN <tag>
#   bash tools/pmc_passes.sh sq [N]        instruction mix and wait cycles -> python tools/pmc.py sum gpurun_out/pmc_sq1 ...
# PMC_CLUSTERED=1 selects bench.py's clustered set.  Results under gpurun_out/pmc_*/
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
WHAT=${1:-traffic}
export PMC_N=${2:-1000000}
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  echo "== pass $name: $*"
  rm -rf "$REPO/gpurun_out/pmc_$name"
  timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc_$name" -- \
      python3 "$REPO/tools/pmc_run.py" > "$REPO/gpurun_out/pmc_$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$REPO/gpurun_out/pmc_$name.log"; return 1; }
  tail -1 "$REPO/gpurun_out/pmc_$name.log"
}
if [ "$WHAT" = traffic ]; then
  run fetch FETCH_SIZE &&
  run write WRITE_SIZE TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum &&
  run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum &&
  run hit TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_64B_sum TCC_NORMAL_WRITEBACK_sum
else
  run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES &&
  run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES &&
  python3 "$REPO/tools/pmc.py" sum "$REPO/gpurun_out/pmc_sq1" traverse && python3 "$REPO/tools/pmc.py" sum "$REPO/gpurun_out/pmc_sq2" traverse
fi
