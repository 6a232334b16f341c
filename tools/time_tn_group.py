"""Times a GPT block's four weight gradients (one grouped launch) and the lm_head's on the 256-row tiles (csrc/gemm_tn_rows.hip) and on the
128 x 128 tiles (HALO_GEMM_TN_ROWS=0): us per launch."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haloop_amd import _lib, ops
_lib.lib(); _lib.lend_scratch(256 << 20); _lib.set_math_mode('bf16')
g = torch.Generator().manual_seed(0)
mk = lambda K, m, n: (torch.randn(K, m, generator=g).cuda().bfloat16(), torch.randn(K, n, generator=g).cuda().bfloat16())
cases = {'block (2304x768, 768x768, 3072x768, 768x3072; K 8192)': [mk(8192, 2304, 768), mk(8192, 768, 768), mk(8192, 3072, 768), mk(8192, 768, 3072)],
         'lm_head (50304x768; K 8192)': [mk(8192, 50304, 768)]}


def t_us(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for name, pairs in cases.items():
    flops = sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for a, b in pairs)
    variants = (('1', ''), ('1', '4'), ('1', '8'), ('0', ''))
    best = {v: 1e30 for v in variants}
    for rep in range(3):                                    # interleaved, the minimum of three rounds per variant
        for env, tn in variants:
            os.environ['HALO_GEMM_TN_ROWS'] = env
            if tn:
                os.environ['HALO_GEMM_TN_ROWS_TN'] = tn
            else:
                os.environ.pop('HALO_GEMM_TN_ROWS_TN', None)
            best[(env, tn)] = min(best[(env, tn)], t_us(lambda: ops.gemm_tn_group(pairs)))
    out = [f"{'256-row tiles' + (' TN ' + tn if tn else ' (chosen width)') if env == '1' else '128 x 128 tiles'}: {us:.1f} us ({flops / us / 1e6:.0f} TF)"
           for (env, tn), us in best.items()]
    print(name + ': ' + ' | '.join(out), flush=True)
