"""Debug aid: where is a forward attention variant wrong?  Per head-dim column, per query position in its 64-block, per key-tile
contribution (V replaced by one-hot tiles).  python tests/tools/attn_cols.py --variant 22"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctrlora_amd import hip

ap = argparse.ArgumentParser(); ap.add_argument("--variant", type=int, default=22); ap.add_argument("--N", type=int, default=1024); ap.add_argument("--B", type=int, default=1); ap.add_argument("--std", type=float, default=1.0)
a = ap.parse_args()
B, H, dh, N = a.B, 8, 40, a.N
inner = H * dh
g = torch.Generator().manual_seed(1)
scale = dh ** -0.5; c = scale * 1.4426950408889634
q32 = (torch.randn(B * N, inner, generator=g) * a.std).cuda()
k = (torch.randn(B * N, inner, generator=g) * a.std).to(torch.bfloat16).cuda()
v = torch.randn(B * N, inner, generator=g).to(torch.bfloat16).cuda()
q = (q32 * c).to(torch.bfloat16)
o = torch.zeros_like(q); lse = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
assert hip.lib().cl_debug_attention_variant(a.variant) == 0
hip.attention_fwd_v2(q, k, v, o, lse, B, H, N, N, dh, scale, q_prescaled=True)
torch.cuda.synchronize()
hip.lib().cl_debug_attention_variant(0)
sp = lambda x: x.double().reshape(B, N, H, dh).permute(0, 2, 1, 3)
s = torch.einsum("bhid,bhjd->bhij", sp(q) / c, sp(k)) * scale
ref = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), sp(v))          # [B,H,N,dh]
got = sp(o)
err = (got - ref)
print("total rel", float(err.norm() / ref.norm()))
print("per column d:", [round(float(err[..., d].norm() / ref[..., d].norm()), 3) for d in range(dh)])
pq = err.reshape(B, H, N // 64, 64, dh)
rq = ref.reshape(B, H, N // 64, 64, dh)
print("per query mod 64:", [round(float(pq[:, :, :, i].norm() / rq[:, :, :, i].norm()), 3) for i in range(64)])
print("per head:", [round(float(err[:, h].norm() / ref[:, h].norm()), 3) for h in range(H)])
pw = err.reshape(B, H, N // 256, 256, dh); rw = ref.reshape(B, H, N // 256, 256, dh)
print("per batch:", [round(float(err[bb].norm() / ref[bb].norm()), 3) for bb in range(B)])
print("per 256-block:", [round(float(pw[:, :, i].norm() / rw[:, :, i].norm()), 3) for i in range(N // 256)])
