cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --no-graph --no-extras --no-configs --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/pmc_mfma.log 2>&1
c=$(find gpurun_out/pmc_mfma -name "*counter_collection.csv" | head -1); k=$(find gpurun_out/pmc_mfma -name "*kernel_trace.csv" | head -1)
python3 tools/pmc_mfma_util.py "$c" "$k" > gpurun_out/r03_mfma_util_lstm.md; cat gpurun_out/r03_mfma_util_lstm.md
HALO_MATH=bf16 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_gpt -- python3 tools/bench_gpt.py --no-cpu-baseline > gpurun_out/pmc_mfma_gpt.log 2>&1
c=$(find gpurun_out/pmc_mfma_gpt -name "*counter_collection.csv" | head -1); k=$(find gpurun_out/pmc_mfma_gpt -name "*kernel_trace.csv" | head -1)
python3 tools/pmc_mfma_util.py "$c" "$k" > gpurun_out/r03_mfma_util_gpt.md; cat gpurun_out/r03_mfma_util_gpt.md
rm -rf gpurun_out/pmc_mfma gpurun_out/pmc_mfma_gpt
