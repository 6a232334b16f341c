# MFMA utilisation of the two-layer launches, one tile per workgroup (B = 64) against two tiles interleaved (B = 128): one PMC pass each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for b in 64 128; do
  export B=$b
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_b$b -- python3 tools/run_lstm2_steps.py > gpurun_out/pmc_mfma_b$b.log 2>&1
  c=$(find gpurun_out/pmc_mfma_b$b -name "*counter_collection.csv" | head -1); k=$(find gpurun_out/pmc_mfma_b$b -name "*kernel_trace.csv" | head -1)
  echo "## B = $b" >> gpurun_out/r04_mfma_util_interleave.md
  python3 tools/pmc_mfma_util.py "$c" "$k" | grep -v "^| \`gemm\|prep_jobs\|splitk\|pack_pair\|Fill" >> gpurun_out/r04_mfma_util_interleave.md
  rm -rf gpurun_out/pmc_mfma_b$b
done
cat gpurun_out/r04_mfma_util_interleave.md
