#!/bin/bash
# One GPU call that refreshes the round's profile set (run from the repo root on the GPU box; writes under gpurun_out/):
#   bench line, rocprofv3 --kernel-trace --stats of the bench command, one step's kernel timeline, in-kernel stamps, PMC traffic passes.
set -u
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="bench.py --no-cpu-baseline --no-extras --no-configs --steps 100 --warmup 10"
timeout -k 10 480 python bench.py > gpurun_out/${R}_bench_n1.json 2> gpurun_out/${R}_bench_n1.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_prof -- python3 $CMD > gpurun_out/${R}_prof.log 2>&1
s=$(find gpurun_out/${R}_prof -name "*kernel_stats.csv" | head -1)
k=$(find gpurun_out/${R}_prof -name "*kernel_trace.csv" | head -1)
python tools/stats_summary.py "$s" "rocprofv3 --kernel-trace --stats -- python3 $CMD" 30 > gpurun_out/${R}_kernel_stats_step_bf16_graph.md
cp "$s" gpurun_out/${R}_kernel_stats_step_bf16_graph.csv
python tools/step_timeline.py "$k" subsample_fused_kernel > gpurun_out/${R}_step_timeline_bf16.md
rm -rf gpurun_out/${R}_prof
timeout -k 10 200 python tools/persist2_stamps.py 0.2 > gpurun_out/${R}_persist2_stamps.txt 2>&1
timeout -k 10 200 python tools/persist2x_stamps.py 0.2 > gpurun_out/${R}_persist2x_stamps_raw.txt 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_fetch -- python3 tools/run_lstm2_steps.py > gpurun_out/${R}_pmc.log 2>&1 && \
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_write -- python3 tools/run_lstm2_steps.py >> gpurun_out/${R}_pmc.log 2>&1
f=$(find gpurun_out/${R}_pmc_fetch -name "*counter_collection.csv" | head -1); w=$(find gpurun_out/${R}_pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$f" "$w" gpurun_out/${R}_lstm2_chain_traffic.json gpurun_out/${R}_lstm2_chain_traffic.md > /dev/null
rm -rf gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write
echo done; tail -c 300 gpurun_out/${R}_bench_n1.json
