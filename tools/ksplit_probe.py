import os, sys
sys.path.insert(0, '/root/repo')
import torch
from haloop_amd import _lib, ops
_lib.lib(); _lib.lend_scratch(256 << 20); _lib.set_math_mode(os.environ.get('HALO_MATH', 'bf16x3'))
g = torch.Generator().manual_seed(0)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(1280, 1024, 4096), (4096, 128, 1280), (1280, 128, 4096)]
for M, N, K in shapes:
    a = torch.randn(M, K, generator=g).cuda(); b = torch.randn(N, K, generator=g).cuda()
    ai, bi = ops.split_image(a), ops.split_image(b)
    out = torch.empty(M, N, device='cuda')
    print(f'{os.environ.get("HALO_MATH", "bf16x3")} KSPLIT={os.environ.get("HALO_KSPLIT","auto")} M{M} N{N} K{K}: {timeit(lambda: ops.gemm_split(ai, bi, M, N, K, out=out)):.1f} us')
