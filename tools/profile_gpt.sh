#!/bin/bash
# GPT-2 small (BASELINE config 3) profile set: bench line, one step's kernel timeline and per-kernel table (run on the GPU box from the repo root).
set -u
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
HALO_MATH=bf16 timeout -k 10 300 python tools/bench_gpt.py --no-cpu-baseline > gpurun_out/${R}_gpt_bench.log 2>&1
HALO_MATH=bf16 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_gptprof -- python3 tools/bench_gpt.py --no-cpu-baseline --no-pmc > gpurun_out/${R}_gptprof.log 2>&1
k=$(find gpurun_out/${R}_gptprof -name "*kernel_trace.csv" | head -1)
python tools/gpt_step_timeline.py "$k" > gpurun_out/${R}_gpt2_small_timeline_bf16.md
python tools/gpt_step_breakdown.py "$k" > gpurun_out/${R}_gpt2_small_step_bf16.md 2>&1
rm -rf gpurun_out/${R}_gptprof

# MFMA utilisation per kernel (its own counter pass)
HALO_MATH=bf16 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${R}_gptmfma -- python3 tools/bench_gpt.py --no-cpu-baseline --no-pmc > gpurun_out/${R}_gptmfma.log 2>&1
c=$(find gpurun_out/${R}_gptmfma -name "*counter_collection.csv" | head -1); k=$(find gpurun_out/${R}_gptmfma -name "*kernel_trace.csv" | head -1)
python3 tools/pmc_mfma_util.py "$c" "$k" > gpurun_out/${R}_gpt2_small_mfma.md
rm -rf gpurun_out/${R}_gptmfma
