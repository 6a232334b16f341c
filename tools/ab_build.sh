#!/bin/bash
# Same-box A/B of the working tree against HEAD.  Step 1 (here, no GPU): builds both libraries into haloop_amd/csrc/ab/.
#   bash tools/ab_build.sh
# Step 2 (one gpurun call): bash tools/ab_run.sh [bench.py flags]  -- interleaves old / new twice and prints step, kernel and inference times.
set -e
cd "$(dirname "$0")/.."
mkdir -p haloop_amd/csrc/ab
make -C haloop_amd/csrc 2>&1 | grep -E " error |Error " || true
cp haloop_amd/csrc/libhalo.so haloop_amd/csrc/ab/libhalo_new.so
git stash -q
make -C haloop_amd/csrc 2>&1 | grep -E " error |Error " || true
cp haloop_amd/csrc/libhalo.so haloop_amd/csrc/ab/libhalo_old.so
git stash pop -q
cp haloop_amd/csrc/ab/libhalo_new.so haloop_amd/csrc/libhalo.so
cmp -s haloop_amd/csrc/ab/libhalo_old.so haloop_amd/csrc/ab/libhalo_new.so && echo "WARNING: the two builds are identical" || echo "built old (HEAD) and new (working tree)"
