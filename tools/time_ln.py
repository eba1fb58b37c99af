"""Micro-timing of cl_layernorm_bwd / cl_colsum variants (GPU box)."""
import torch, sys
sys.path.insert(0, ".")
from ctrlora_amd import hip

dev = "cuda"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for ws in (True, False):
    if ws: hip.ensure_workspace(dev)
    else:
        hip._chk(hip.lib().cl_set_workspace(None, 0), "ws"); hip._workspace = None
    for M, D in [(32768, 320), (8192, 640), (2048, 1280), (512, 1280)]:
        x = torch.randn(M, D, device=dev).bfloat16(); dy = torch.randn(M, D, device=dev).bfloat16()
        acc = torch.randn(M, D, device=dev).bfloat16()
        gamma = torch.ones(D, device=dev); beta = torch.zeros(D, device=dev)
        y = torch.empty_like(x); dx = torch.empty_like(x); stats = torch.empty(M, 2, device=dev)
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        hip.layernorm_fwd(x, y, gamma, beta, 1e-5, stats)
        t_f = timeit(lambda: hip.layernorm_fwd(x, y, gamma, beta, 1e-5, stats))
        t_n = timeit(lambda: hip.layernorm_bwd(x, dy, dx, gamma, stats, accum=acc))
        t_w = timeit(lambda: hip.layernorm_bwd(x, dy, dx, gamma, stats, accum=acc, dgamma=dg, dbeta=db))
        cs = torch.zeros(1, D, device=dev)
        t_c = timeit(lambda: hip.colsum(dy, cs, 1, M, 1.0))
        print(f"ws={ws} M={M} D={D}: ln_fwd {t_f:.1f} us  ln_bwd {t_n:.1f} us  ln_bwd+wg {t_w:.1f} us  colsum {t_c:.1f} us")
