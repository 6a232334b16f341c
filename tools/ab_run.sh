#!/bin/bash
# Step 2 of tools/ab_build.sh, on the GPU box: old / new / old / new.
cd "$(dirname "$0")/.."
for i in 1 2; do
    for v in old new; do
        cp haloop_amd/csrc/ab/libhalo_$v.so haloop_amd/csrc/libhalo.so
        timeout -k 10 150 python3 bench.py --no-configs --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'step ms', d['ms_per_step'], 'bwd us', d['roofline']['avg_launch_us'], 'fwd us', d['roofline']['forward_twin']['avg_launch_us'], 'loss', d['final_loss'], 'infer ms', d['inference']['ms_per_batch'])"
    done
done
cp haloop_amd/csrc/ab/libhalo_new.so haloop_amd/csrc/libhalo.so
