set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu_final.log 2>&1 || { tail -30 gpurun_out/pytest_gpu_final.log; exit 1; }
tail -2 gpurun_out/pytest_gpu_final.log
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/smoke_final.log 2>&1 || { tail gpurun_out/smoke_final.log; exit 1; }
tail -1 gpurun_out/smoke_final.log
timeout -k 10 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || { tail gpurun_out/bench_final.err; exit 1; }
cut -c1-400 gpurun_out/bench_final.json
rm -rf gpurun_out/prof_scan
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_scan -o scan -- python bench.py --workload scan --cpu-sample 0 > gpurun_out/bench_scan_rocprof.json 2> gpurun_out/bench_scan_rocprof.err || { tail gpurun_out/bench_scan_rocprof.err; exit 1; }
find gpurun_out/prof_scan -name "*kernel_stats.csv" -exec cp {} gpurun_out/scan_kernel_stats.csv \;
python tools/trace_launches.py gpurun_out/prof_scan | tail -14 > gpurun_out/scan_launches.txt
find gpurun_out/prof_scan -name "*kernel_trace.csv" -delete
cut -c1-300 gpurun_out/bench_scan_rocprof.json
rm -rf gpurun_out/prof_hnsw
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hnsw -o hnsw -- python bench.py --cpu-sample 0 --second-dataset none --sustain-s 0 --no-pcie > gpurun_out/bench_hnsw_rocprof.json 2> gpurun_out/bench_hnsw_rocprof.err || { tail gpurun_out/bench_hnsw_rocprof.err; exit 1; }
find gpurun_out/prof_hnsw -name "*kernel_stats.csv" -exec cp {} gpurun_out/hnsw_kernel_stats.csv \;
python tools/kernel_trace_table.py $(find gpurun_out/prof_hnsw -name "*kernel_trace.csv" | head -1) traverse_duo 20 gpurun_out/traverse_launches.csv
find gpurun_out/prof_hnsw -name "*kernel_trace.csv" -delete
cut -c1-300 gpurun_out/bench_hnsw_rocprof.json
