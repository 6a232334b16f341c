import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from haloop_amd import _lib, ops
_lib.lib(); _lib.lend_scratch()
DEV='cuda'
B,T_in,H,V,S,p = 1,9,128,5,1,0.0
T = (T_in + 6 - 5) // 4 + 1
g = torch.Generator().manual_seed(B * 7 + H)
feats = torch.randn(B, T, H, generator=g).relu().to(DEV)
W = (torch.randn(V, H, generator=g) / H ** 0.5).to(DEV)
b = (torch.randn(V, generator=g) * 0.1).to(DEV)
il = torch.tensor([T_in - 3 * (i % 5) for i in range(B)], dtype=torch.int64).to(DEV)
tg = torch.randint(1, V, (B, S), generator=g).to(DEV)
tl = torch.randint(max(1, S // 2), S + 1, (B,), generator=g).to(DEV)
drop = ops.NO_DROPOUT
sid = _lib.HALO_STREAM_CLASSIFIER
loss0 = torch.zeros((), device=DEV); t0 = torch.zeros(1, device=DEV, dtype=torch.int32)
dW0, db0 = torch.empty_like(W), torch.empty_like(b)
lp0, alpha0, nll0, flen0, go0, (tg64, tl64) = ops.ctc_head_fwd(feats, W, b, drop, sid, il, tg, tl, loss0, t0)
dfeats0 = ops.ctc_head_bwd(feats, W, drop, sid, flen0, tg64, tl64, lp0, alpha0, nll0, go0, dW0, db0)
_lib.set_math_mode('bf16')
loss = torch.full((), -1.0, device=DEV)
ticket = ops.ctc_head_train_ticket(B, H, DEV)
dW, db = torch.empty_like(W), torch.empty_like(b)
for rep in range(1):
    dfeats, nll, flen, lp = ops.ctc_head_train(feats, W, b, drop, sid, il, tg, tl, loss, ticket, dW, db, want_lp=True)
    torch.cuda.synchronize()
    d = (dW - dW0).abs().cpu().numpy()
    print('rep', rep, 'ticket', ticket[:2].tolist(), 'dW bad cols per row', [(np.nonzero(d[v] > 1e-4)[0].tolist()) for v in range(V)][:2])
    A, D = dW.cpu().numpy(), dW0.cpu().numpy()
    for v in range(V):
        print('row', v, 'best match desired row', int(np.argmin([np.abs(A[v] - D[u]).max() for u in range(V)])), 'err', min(np.abs(A[v] - D[u]).max() for u in range(V)), 'ratio', (A[v] / np.where(D[v] == 0, 1, D[v]))[:6])
    print('dfeats err', (dfeats - dfeats0).abs().max().item(), 'db err', (db - db0).abs().max().item(), 'lp err', (lp-lp0).abs().max().item())
