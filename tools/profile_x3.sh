#!/bin/bash
# One step of the fp32-grade (bf16x3) LSTM-CTC step, launch by launch (run on the GPU box from the repo root).
set -u
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_x3prof -- python3 bench.py --math bf16x3 --no-cpu-baseline --no-extras --no-configs --steps 50 --warmup 10 > gpurun_out/${R}_x3prof.log 2>&1
k=$(find gpurun_out/${R}_x3prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py "$k" subsample_fused_kernel > gpurun_out/${R}_step_timeline_bf16x3.md
rm -rf gpurun_out/${R}_x3prof
