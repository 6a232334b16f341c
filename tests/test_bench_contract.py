"""The bench.py output contract, checked on the newest committed line (profiles/rNN_bench_n1.json, produced by `python bench.py` on an
MI355X box) and on bench.py's own argument surface -- no GPU needed: the keys the driver reads are there, typed, and consistent with each other."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_bench_n1.json')))[-1]) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_committed_bench_line_meets_the_contract():
    d = _line()
    for k, t in (('metric', str), ('value', float), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int), ('ms_per_step', float),
                 ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str), ('config', dict), ('roofline', dict),
                 ('cpu_baseline', dict)):
        assert isinstance(d[k], t), (k, type(d[k]))
    assert 'vs_baseline' in d and d['vs_baseline'] is None          # BASELINE.md holds no published number for this metric
    assert d['n_gpus'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic'
    assert d['unit'].startswith('utterances') and 'utterances/sec' in d['metric']
    assert 'workload' in d['config'] and 'model' not in d['config']
    B = d['config']['global_batch']
    assert abs(d['value'] - B * 1e3 / d['ms_per_step']) <= 2e-3 * d['value']            # whole-job utterances/s == batch / step time
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) <= 1e-3
    assert r['traffic'] is None or r['traffic'] >= r['algorithmic_bytes_per_launch']         # measured HBM bytes are not below the algorithmic ones
    assert abs(r['achieved'] - r['algorithmic_bytes_per_launch'] / (r['avg_launch_us'] * 1e-6) / 1e9) <= 1e-2 * r['achieved']
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and isinstance(c['sample'], str)
    assert d['value'] >= 10 * c['value']                                                  # the north star's ">= 10x the CPU path"
    assert d['n_ranks_seen'] == 1
    # round 4: the fp32-grade step in the parsed config, the stricter accounting beside the bench's, the reference's own model shapes
    assert d['config']['bf16x3_ms_per_step'] > d['ms_per_step']
    if 'frac_10h' in r:          # round 5 on: `frac` is SURVEY.md 8d's own byte count, the rounds 1-4 yardstick rides beside it
        assert 0 < r['frac'] < r['frac_10h']
    else:
        assert 0 < r['frac_8d_strict'] < r['frac']
    for leg in ('stock3', 'H1536', 'inference', 'inference_bf16x3'):
        assert d[leg]['value'] > 0, leg
    assert d['b_sweep']['B128']['ms_per_step'] < 2 * d['ms_per_step']            # two tiles per workgroup: less than two steps of 64


def test_bench_argument_surface():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ('--gpus', '--steps', '--warmup', '--no-graph', '--graph', '--math', '--dp-algo', '--grad-dtype', 'rs_ag_flat'):
        assert flag in out.stdout, flag
