#!/bin/bash
# Round-end evidence run on the GPU box (from the repo root): rocprofv3 kernel stats of the bench legs, the per-launch
# table of the traversal kernel, the stamps and the occupancy probe.  Output under gpurun_out/final/; the summaries
# worth keeping are copied into profiles/rN by hand.   bash tools/final_run.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$name" -o "$name" -- python3 "$R/bench.py" "$@" > "$O/bench_${name}_under_rocprof.json" 2> "$O/bench_$name.err" || { tail "$O/bench_$name.err"; return 1; }
  find "$O/prof_$name" -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats_$name.csv" \;
}
prof 1m_balanced --legs none --cpu-sample 0 --no-pcie &&
python3 "$R/tools/kernel_trace_table.py" "$(find "$O/prof_1m_balanced" -name '*kernel_trace.csv' | head -1)" traverse_lean_kernel reference:8,warmup:5,timed:100,solo:16,resident:100,sustained:* "$O/traverse_launches_1m_balanced.csv" > "$O/traverse_launches_summary.txt" &&
prof scan_1m_512 --workload scan --cpu-sample 0 --second-dataset none &&
prof 10m_balanced --rows 10000000 --legs none --cpu-sample 0 --no-pcie --no-recall &&
find "$O" -name "*kernel_trace.csv" -delete &&
find "$O" -name "*agent_info.csv" -delete &&
timeout -k 10 200 python3 "$R/tools/occupancy_probe.py" > "$O/occupancy_probe_1m_balanced_lean.txt" 2> "$O/occ.err" &&
timeout -k 10 200 python3 "$R/tools/profile_lean.py" 1000000 > "$O/lean_stamps_1m_iid.txt" 2> "$O/stamps.err" &&
timeout -k 10 200 python3 "$R/tools/profile_lean.py" 1000000 2048 > "$O/lean_stamps_1m_iid_2048q.txt" 2>> "$O/stamps.err" &&
timeout -k 10 200 python3 "$R/tools/profile_lean.py" 1000000 1000 --clustered > "$O/lean_stamps_1m_clustered.txt" 2>> "$O/stamps.err" &&
echo final_run ok
