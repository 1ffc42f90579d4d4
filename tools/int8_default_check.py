import sys, numpy as np, torch
sys.path.insert(0, '.')
import bevformer_tensorrt_amd as bev, oracle
g = torch.Generator().manual_seed(0)
bs, levels, nq, P, ppg = 6, [[15, 25]], 2500, 8, 4
L, nk = 1, 375
v = torch.randn(bs, nk, 8, 32, generator=g); r = torch.rand(bs, nq, 1, 2 * ppg, generator=g)
o = torch.randn(bs, nq, 8, L * P * 2, generator=g); w = torch.randn(bs, nq, 8, L * P, generator=g)
q = lambda x: (torch.clamp(torch.round(x / (x.abs().max() / 127)), -127, 127).to(torch.int8), float(x.abs().max() / 127))
(vq, sv), (oq, so), (wq, sw) = q(v), q(o), q(w)
sh = torch.tensor(levels, dtype=torch.int32)
for rdt, u8 in ((torch.float32, False), (torch.float16, True)):
    got = bev.multi_scale_deformable_attn_int8(vq.cuda(), sh.cuda(), r.to(rdt).cuda(), oq.cuda(), wq.cuda(), sv, so, sw, 0.02).cpu().numpy().astype(int)
    want = oracle.msda_s8(vq.numpy(), sv, sh.numpy(), r.to(rdt).float().numpy(), oq.numpy(), so, wq.numpy(), sw, 0.02, u8_weights=u8).astype(int)
    d = np.abs(got - want); print(rdt, "max", d.max(), "frac>0", (d > 0).mean())
