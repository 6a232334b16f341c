cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="${BENCH_CMD:-bench.py --no-cpu-baseline --no-extras --no-configs --steps 100 --warmup 10}"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_prof -- python3 $CMD > gpurun_out/tl_prof.log 2>&1
k=$(find gpurun_out/tl_prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py "$k" subsample_fused_kernel > gpurun_out/tl.md
rm -rf gpurun_out/tl_prof
cat gpurun_out/tl.md
