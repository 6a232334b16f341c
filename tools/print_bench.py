"""Print the numbers of a bench.py line that the round's notes quote."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get('roofline', {})
print('headline', d['value'], d['unit'], d['ms_per_step'], 'ms; roofline frac', r.get('frac'), 'strict', r.get('frac_8d_strict'), '10h', r.get('frac_10h'), 'bwd us', r.get('avg_launch_us'),
      'fwd us', r.get('forward_twin', {}).get('avg_launch_us'))
for k in ('inference', 'inference_bf16x3'):
    if k in d:
        print(k, d[k]['ms_per_batch'], 'ms')
for k in ('stock3', 'H1536', 'bf16x3_mode', 'f32_mode'):
    if k in d:
        print(k, d[k]['ms_per_step'], 'ms', d[k].get('step_frac_of_hbm_peak'))
for k, v in d.get('b_sweep', {}).items():
    print(k, v['ms_per_step'], 'ms', v.get('step_frac_of_hbm_peak'))
if 'gpt2_small' in d:
    print('gpt2_small', d['gpt2_small']['ms_per_step'], 'ms')
if 'asr_transformer32' in d:
    print('asr', d['asr_transformer32'].get('stages'))
if 'cpu_baseline' in d:
    print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('config1_b4', {}).get('value'))
