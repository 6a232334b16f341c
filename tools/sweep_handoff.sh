for nap in 0 1 2 3; do for sh in 3; do
HALO_PERSIST_NAP=$nap HALO_PERSIST_REPLICA_SHIFT=$sh timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --no-configs --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('nap $nap shift $sh', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['forward_twin']['avg_launch_us'])"
done; done
for sh in 0 1 2 4 5 6 7; do
HALO_PERSIST_NAP=2 HALO_PERSIST_REPLICA_SHIFT=$sh timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --no-configs --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('nap 2 shift $sh', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['forward_twin']['avg_launch_us'])"
done
