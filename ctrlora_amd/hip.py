"""ctypes binding of libctrlora_hip.so (include/ctrlora_hip.h) for torch tensors.

PyTorch is plumbing here: device memory, the current HIP stream, torch.distributed.
Every function enqueues hand-written gfx950 kernels on `torch.cuda.current_stream()`.
There is NO fallback: if the library is missing or a kernel rejects its arguments the
call raises -- a product path that silently ran ATen / CPU code would void every
parity claim.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTRLORA_LIB") or os.path.join(_HERE, "libctrlora_hip.so")   # (CTRLORA_LIB: A/B builds of the same ABI)

BF16, F32 = 0, 1
LINEAR, CONV_S1, CONV_S2, CONV_UP2, CONV_T2, CONV_S2A, CONV_UP2P, CONV_T2P, CONV_S2K4 = 0, 1, 2, 3, 4, 5, 6, 7, 8
ACT_NONE, ACT_SILU, ACT_GEGLU, ACT_GEGLU_SPLIT = 0, 1, 2, 3


class HipError(RuntimeError):
    pass


class WgradDesc(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("lddy", C.c_long), ("x", C.c_void_p), ("ldx", C.c_long),
                ("dW", C.c_void_p), ("lddw", C.c_long), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("scale", C.c_float), ("tap", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int),
                ("Wout", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("reserved", C.c_int)]


class GemmParams(C.Structure):
    _fields_ = [
        ("A1", C.c_void_p), ("lda1", C.c_long), ("K1", C.c_int),
        ("W1", C.c_void_p), ("ldw1", C.c_long),
        ("A2", C.c_void_p), ("lda2", C.c_long), ("K2", C.c_int),
        ("W2", C.c_void_p), ("ldw2", C.c_long),
        ("M", C.c_int), ("N", C.c_int), ("mode", C.c_int),
        ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
        ("zero_page", C.c_void_p), ("bias", C.c_void_p),
        ("rowbias", C.c_void_p), ("ldrb", C.c_long), ("rows_per_batch", C.c_int),
        ("residual", C.c_void_p), ("ldr", C.c_long),
        ("alpha", C.c_float), ("beta", C.c_float), ("act", C.c_int),
        ("C", C.c_void_p), ("ldc", C.c_long),
        ("out_f32", C.c_int), ("atomic", C.c_int), ("splitk", C.c_int),
        ("a1_group_n", C.c_int), ("a2_group_n", C.c_int), ("alpha_n", C.c_int),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float), ("ln_stats", C.c_void_p),
    ]


_lib = None

# name -> argtypes (restype is int except where noted); mirrors include/ctrlora_hip.h
_P, _L, _I, _F = C.c_void_p, C.c_long, C.c_int, C.c_float
_SIGS = {
    "cl_abi_version": [],
    "cl_last_hip_error": [],
    "cl_set_workspace": [_P, _L],
    "cl_set_stream_workspace": [_P, _P, _L],
    "cl_gemm_force_config": [_I],
    "cl_gemm_force_splitk": [_I],
    "cl_gemm_tune_set": [_I] * 9,
    "cl_gemm_tune_clear": [],
    "cl_gemm_tune_size": [],
    "cl_gemm": [C.POINTER(GemmParams), _I, _P],
    "cl_lora_down": [_I, _P, _L, _P, _I, _P, _L, _I, _I, _P],
    "cl_lora_linear_fwd": [_I, _P, _L, _P, _P, _P, _L, _P, _I, _P, _L, _I, _P, _L, _I, _I, _I, _P],
    "cl_lora_linear_bwd_data": [_I, _P, _L, _P, _P, _P, _I, _P, _L, _P, _L, _P, _L, _I, _I, _I, _P],
    "cl_weight_grad": [_I, _P, _L, _P, _L, _P, _L, _I, _I, _I, _F, _P],
    "cl_weight_grad_tn_group": [_I, _I, _P, _P, _P],
    "cl_weight_grad_tn": [_I, _P, _L, _P, _L, _P, _L, _I, _I, _I, _F, _P, _P],
    "cl_conv3x3_fwd": [_I, _I, _P, _L, _P, _P, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P, _P],
    "cl_conv3x3_bwd_data": [_I, _I, _P, _L, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P, _P],
    "cl_conv1x1_fwd": [_I, _P, _L, _P, _P, _F, _P, _L, _F, _P, _L, _I, _I, _I, _P],
    "cl_groupnorm_ws_floats": [_I, _I, _I],
    "cl_groupnorm_silu_fwd": [_I, _P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P, _P],
    "cl_groupnorm_silu_bwd": [_I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "cl_layernorm_fwd": [_I, _P, _L, _P, _L, _P, _P, _I, _I, _F, _P, _P],
    "cl_layernorm_bwd": [_I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _I, _I, _P, _P, _P],
    "cl_attention_fwd": [_I, _P, _L, _P, _L, _P, _I, _P, _L, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "cl_attention_bwd": [_I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _I, _P, _I, _P, _P, _I, _P, _L, _P, _L,
                         _P, _L, _I, _I, _I, _I, _I, _F, _P],
    "cl_attention_fwd_v2": [_I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "cl_attention_bwd_v2": [_I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _I, _P, _L, _P, _L, _P, _L,
                            _I, _I, _I, _I, _I, _F, _I, _P, _P],
    "cl_geglu_fwd": [_I, _P, _L, _P, _L, _L, _I, _P],
    "cl_geglu_bwd": [_I, _P, _L, _P, _L, _P, _L, _L, _I, _P],
    "cl_silu_fwd": [_I, _P, _P, _L, _P],
    "cl_silu_bwd": [_I, _P, _P, _P, _L, _P],
    "cl_axpby": [_I, _P, _L, _P, _L, _L, _I, _F, _F, _P],
    "cl_transpose": [_I, _I, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _P],
    "cl_nchw_to_tok": [_I, _P, _P, _L, _I, _I, _I, _I, _P],
    "cl_tok_to_nchw": [_I, _P, _L, _P, _I, _I, _I, _F, _F, _P],
    "cl_colsum": [_I, _P, _L, _P, _L, _I, _I, _I, _F, _P],
    "cl_pool2x2": [_I, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P],
    "cl_pack2d": [_I, _P, _L, _P, _L, _L, _I, _I, _P],
    "cl_repack": [_I, _P, _P, _P, _I, _I, _P],
    "cl_timestep_embedding": [_I, _P, _P, _P, _L, _I, _I, _P],
    "cl_qsample": [_P, _P, _P, _P, _P, _P, _I, _L, _P],
    "cl_mse_loss": [_P, _P, _P, _P, _L, _F, _P],
    "cl_p_losses_mse": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _F, _F, _F, _P],
    "cl_zero": [_P, _L, _P],
    "cl_softmax_rows": [_I, _P, _L, _P, _L, _L, _I, _F, _P],
    "cl_conv_tap_gather": [_I, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "cl_ddim_step": [_P, _P, _P, _P, _P, _I, _F, _P, _P, _L, _P],
    "cl_tick": [_P, _P],
    "cl_adamw_dev": [_P, _P, _P, _P, _L, _P, _P, _P],
    "cl_ddim_set_t": [_P, _P, _I, _P, _I, _P],
    "cl_ddim_step_dev": [_P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _L, _P],
    "cl_adamw": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P],
}
EXPORTED = tuple(_SIGS.keys())
# probe hooks (ctrlora_amd/csrc/debug_hooks.h): exported by the library, not part of include/ctrlora_hip.h
_DEBUG_SIGS = {
    "cl_debug_attention_variant": [_I],
    "cl_debug_attention_fuse_delta": [_I],
    "cl_debug_groupnorm_form": [_I, _I],
    "cl_debug_groupnorm_coop": [_I],
    "cl_debug_groupnorm_coop_timeouts": [],
    "cl_debug_gemm_tag": [_I],
    "cl_debug_gemm_xs_rules": [_I],
    "cl_debug_wgrad_ring": [_I],
    "cl_debug_gemm_tag_count": [],
    "cl_debug_gemm_tag_get": [_I, _P],
}


def lib():
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(f"{LIB_PATH} not found -- run `python -m ctrlora_amd.build` "
                           "(the CtrLoRA engine has no non-HIP fallback)")
        L = C.CDLL(LIB_PATH)
        for name, args in {**_SIGS, **_DEBUG_SIGS}.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_long if name == "cl_groupnorm_ws_floats" else C.c_int
        _lib = L
        if os.environ.get("CTRLORA_ATTN_FUSE_DELTA", "1") == "0":      # A/B switch: separate attn_delta launch
            L.cl_debug_attention_fuse_delta(0)
        gn3 = os.environ.get("CTRLORA_GN_THREE_PASS", "0") == "1"      # A/B switch: GroupNorm with the finalize launch
        gn1 = os.environ.get("CTRLORA_GN_ONE_PASS", "1") != "0"        # A/B switch: one-launch (register-resident) GroupNorm
        if gn3 or not gn1:
            L.cl_debug_groupnorm_form(int(gn3), int(gn1))
        if os.environ.get("CTRLORA_GN_COOP", "0") == "1":              # A/B switch: cooperative one-pass GroupNorm (norm_coop.hip; off: no faster)
            L.cl_debug_groupnorm_coop(1)
        L.cl_debug_gemm_xs_rules(int(XS_ENABLED))
        if os.environ.get("CTRLORA_WGRAD_RING"):                     # A/B switch: 4 = the round-1..4 ring depth
            L.cl_debug_wgrad_ring(int(os.environ["CTRLORA_WGRAD_RING"]))
        if os.environ.get("CTRLORA_GEMM_TUNED", "1") != "0":
            load_gemm_table(os.environ.get("CTRLORA_GEMM_TABLE", GEMM_TABLE_PATH))
            # overlays of the built-in table only: a table the user names is taken as it is
            if "CTRLORA_GEMM_TABLE" not in os.environ:
                if XS_ENABLED:  # round 5: signatures the x-stationary kernel (csrc/gemm_xs.hip, configuration 34) won
                    load_gemm_table(GEMM_XS_TABLE_PATH, clear=False)
                if R06_ENABLED and os.path.exists(GEMM_R06_TABLE_PATH):   # round 6: the loader / consumer kernel (csrc/gemm_w4.hip,
                    load_gemm_table(GEMM_R06_TABLE_PATH, clear=False)     # 40 / 41) and the deep-ring small tiles (42 .. 46)
    return _lib


GEMM_TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuned_gfx950.json")
GEMM_XS_TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuned_gfx950_xs.json")
GEMM_R06_TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuned_gfx950_r06.json")
# A/B switch: 0 = the round-6 overlay (configurations 40 .. 46) is not loaded
R06_ENABLED = os.environ.get("CTRLORA_GEMM_R06", "1") != "0"
# A/B switch: 0 = no product goes to the x-stationary streaming kernel (neither through the launch table nor as the fused
# GEGLU projection of the no-grad forwards)
XS_ENABLED = os.environ.get("CTRLORA_GEMM_XS", "1") != "0"


# A/B switch: 0 = every LayerNorm stays its own launch
LN_PROLOGUE = os.environ.get("CTRLORA_LN_PROLOGUE", "1") != "0"


def xs_ln_ok(M: int, N: int, K: int, act: int = 0) -> bool:
    """Can LayerNorm(x) . W^T run as ONE launch (LayerNorm as the x-stationary kernel's prologue, csrc/gemm_xs.hip)?  The
    kernel's own preconditions: bf16 (the caller checks the dtype), K in {320, 640} with no second K segment, whole 32-column
    output blocks, a GEGLU product only where xs_geglu_ok() sends it to this kernel anyway."""
    if not (XS_ENABLED and LN_PROLOGUE) or K not in (320, 640) or M < 128:
        return False
    if act == ACT_GEGLU_SPLIT:
        return N % 64 == 0 and xs_geglu_ok(M, K, 0)
    return act == ACT_NONE and N % 32 == 0


def gemm_tags():
    """Launch-tag table of the contraction kernels (csrc/debug_hooks.h: cl_debug_gemm_tag; tools/prof_shapes.py)."""
    L = lib()
    out = []
    buf = (C.c_long * 12)()
    for i in range(int(L.cl_debug_gemm_tag_count())):
        _chk(L.cl_debug_gemm_tag_get(i, C.cast(buf, C.c_void_p)), "cl_debug_gemm_tag_get")
        v = list(buf)
        out.append(dict(dtype=v[0], mode=v[1], M=v[2], N=v[3], K1=v[4], K2=v[5], act=v[6], residual=v[7], tag=v[8],
                        workgroups=v[9], wg_size=v[10], launches=v[11]))
    return out


def xs_geglu_ok(M: int, K: int, r: int) -> bool:
    """Does the fused GEGLU projection of a no-grad forward go to the x-stationary kernel (act = ACT_GEGLU_SPLIT, natural row
    order)?  Measured against the tile kernels' fused GEGLU (profiles/r05_gemm_xs/probe_xs_fast_gelu.log): K = 320 at every M
    (73 vs 102 us at 32768 rows, 280 vs 450 at 131072), K = 640 from 8192 rows (253 vs 324 us at 32768; 71 vs 78 at 8192)."""
    if not XS_ENABLED or r not in (0, 128):
        return False
    return (K == 320 and M >= 128) or (K == 640 and M >= 8192)


def load_gemm_table(path: str, clear: bool = True) -> int:
    """Register the measured launch table (tools/gemm_autotune.py) with the library: rows
    [dtype, mode, M, N, K1, K2, geglu, cfg, splitk].  A missing file leaves the built-in rules in charge
    (CTRLORA_GEMM_TUNED=0 does the same on purpose, for A/B runs).  Returns the number of entries."""
    import json
    L = lib()
    if clear:
        L.cl_gemm_tune_clear()
    if not path or not os.path.exists(path):
        return 0
    with open(path) as f:
        tab = json.load(f)
    for row in tab.get("entries", []):
        _chk(L.cl_gemm_tune_set(*[int(v) for v in row[:9]]), "cl_gemm_tune_set")
    return int(L.cl_gemm_tune_size())


def _chk(rc: int, what: str):
    if rc != 0:
        detail = "unsupported shape/alignment"
        if rc != 1:
            L = lib()
            L.cl_last_hip_error_string.restype = C.c_char_p
            detail = f"HIP launch error {L.cl_last_hip_error()}: {L.cl_last_hip_error_string().decode()}"
        raise HipError(f"{what} failed with code {rc} ({detail})")


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise HipError(f"unsupported dtype {t.dtype}")


def dt_of(dtype: torch.dtype) -> int:
    return BF16 if dtype == torch.bfloat16 else F32


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def ld(t: Optional[torch.Tensor]) -> int:
    """Row stride (elements) of a 2-D view whose last dim is contiguous."""
    if t is None:
        return 0
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.stride(0)


ZERO_PAGE_BYTES = 16384   # conv halo source: >= 4 * Cin bytes (include/ctrlora_hip.h), 16 KiB covers Cin <= 4096
_zero_pages = {}
_workspace = None
WORKSPACE_BYTES = 64 << 20


_stream_ws = {}


def bind_stream_workspace(stream: "torch.cuda.Stream", nbytes: int = 32 << 20) -> None:
    """Give `stream` its own split-K scratch (needed before running contractions on it concurrently with
    another stream)."""
    key = stream.cuda_stream
    if key not in _stream_ws:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=stream.device)
        _chk(lib().cl_set_stream_workspace(key, buf.data_ptr(), nbytes), "cl_set_stream_workspace")
        _stream_ws[key] = buf


_side_streams = {}


def side_stream(device) -> "torch.cuda.Stream":
    """The process-wide second compute stream of a device (with its own split-K scratch)."""
    key = str(device)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        bind_stream_workspace(st)
        _side_streams[key] = st
    return st


def ensure_workspace(device) -> None:
    """Register the split-K scratch (fp32 partial slabs) with the library once per process."""
    global _workspace
    if _workspace is None:
        _workspace = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
        _chk(lib().cl_set_workspace(_workspace.data_ptr(), WORKSPACE_BYTES), "cl_set_workspace")


def zero_page(device) -> torch.Tensor:
    k = str(device)
    if k not in _zero_pages:
        _zero_pages[k] = torch.zeros(ZERO_PAGE_BYTES, dtype=torch.uint8, device=device)
    return _zero_pages[k]


# ------------------------------------------------------------------ dense contractions

def gemm(a1, w1, out, *, a2=None, w2=None, bias=None, rowbias=None, rows_per_batch=0, residual=None,
         alpha=1.0, beta=0.0, act=ACT_NONE, mode=LINEAR, conv=None, k1=None, out_f32=False, atomic=False,
         splitk=1, M=None, N=None, dtype=None, a1_group_n=0, a2_group_n=0, alpha_n=0, ln=None):
    """out[M,N] = act(a1.w1^T + a2.w2^T + bias + rowbias[m // rows_per_batch]) * alpha + beta * residual.

    a1: [M,K1] (LINEAR) or the NHWC activation [B*Hin*Win, C] (conv modes, conv=(B,Hin,Win,Hout,Wout)).
    ln = (gamma, beta, eps, stats or None): LayerNorm of a1's rows as a prologue of the product (xs_ln_ok() says when).
    """
    p = GemmParams()
    if _workspace is None:
        ensure_workspace(a1.device)
    dty = dt(a1) if dtype is None else dtype
    p.A1 = a1.data_ptr(); p.lda1 = ld(a1); p.K1 = a1.shape[1] if k1 is None else k1
    p.W1 = w1.data_ptr(); p.ldw1 = ld(w1)
    if a2 is not None:
        p.A2 = a2.data_ptr(); p.lda2 = ld(a2); p.W2 = w2.data_ptr(); p.ldw2 = ld(w2)
        p.K2 = w2.shape[1] if a2_group_n else a2.shape[1]        # grouped: a2 holds one K2-wide block per output group
    p.M = out.shape[0] if M is None else M
    p.N = out.shape[1] if N is None else N
    p.mode = mode
    if conv is not None:
        p.B, p.Hin, p.Win, p.Hout, p.Wout = conv
        p.zero_page = zero_page(a1.device).data_ptr()
    p.bias = ptr(bias)
    if rowbias is not None:
        p.rowbias = rowbias.data_ptr(); p.ldrb = ld(rowbias); p.rows_per_batch = rows_per_batch
    if residual is not None:
        p.residual = residual.data_ptr(); p.ldr = ld(residual)
    p.alpha = alpha; p.beta = beta; p.act = act
    p.C = out.data_ptr(); p.ldc = ld(out)
    p.out_f32 = int(out_f32); p.atomic = int(atomic); p.splitk = splitk
    p.a1_group_n = a1_group_n; p.a2_group_n = a2_group_n; p.alpha_n = alpha_n
    if ln is not None:
        p.ln_gamma = ln[0].data_ptr(); p.ln_beta = ln[1].data_ptr(); p.ln_eps = ln[2]; p.ln_stats = ptr(ln[3])
    _chk(lib().cl_gemm(C.byref(p), dty, stream()), "cl_gemm")
    return out


def weight_grad(dyT, xT, dW, scale=1.0):
    """dW[N,K] (fp32) += scale * dyT[N,Mp] . xT[K,Mp]^T  (split-K, fp32 atomics)."""
    _chk(lib().cl_weight_grad(dt(dyT), dyT.data_ptr(), ld(dyT), xT.data_ptr(), ld(xT), dW.data_ptr(), ld(dW),
                              dyT.shape[0], xT.shape[0], dyT.shape[1], scale, stream()), "cl_weight_grad")


def weight_grad_tn_group(problems):
    """problems: list of (dy [M,N] bf16, x [M,K] bf16, dW [N,K] fp32, scale[, conv]): ONE launch for all of them.
    conv = (tap, Hin, Win, Hout, Wout, stride, pad): one tap of a 3x3 conv's weight gradient (x = NHWC input)."""
    n = len(problems)
    if n == 0:
        return
    if _workspace is None:
        ensure_workspace(problems[0][0].device)
    arr = (WgradDesc * n)()
    for d, prob in zip(arr, problems):
        dy, x, dW, scale = prob[:4]
        d.dy = dy.data_ptr(); d.lddy = ld(dy); d.x = x.data_ptr(); d.ldx = ld(x)
        d.dW = dW.data_ptr(); d.lddw = ld(dW); d.M = dy.shape[0]; d.N = dy.shape[1]; d.K = x.shape[1]
        d.scale = scale
        d.tap = -1
        if len(prob) > 4 and prob[4] is not None:
            d.tap, d.Hin, d.Win, d.Hout, d.Wout, d.stride, d.pad = prob[4]
    _chk(lib().cl_weight_grad_tn_group(BF16, n, C.cast(arr, C.c_void_p), zero_page(problems[0][0].device).data_ptr(),
                                       stream()), "cl_weight_grad_tn_group")


def repack(dtype, flat, desc, tile_prefix, ndesc, total_tiles):
    _chk(lib().cl_repack(dt_of(dtype), flat.data_ptr(), desc.data_ptr(), tile_prefix.data_ptr(), ndesc, total_tiles,
                         stream()), "cl_repack")


def weight_grad_tn(dy, x, dW, scale=1.0):
    """dW[N,K] (fp32) += scale * dy[M,N]^T . x[M,K], bf16 operands as they are (no transposes)."""
    if _workspace is None:
        ensure_workspace(dy.device)
    _chk(lib().cl_weight_grad_tn(dt(dy), dy.data_ptr(), ld(dy), x.data_ptr(), ld(x), dW.data_ptr(), ld(dW),
                                 dy.shape[0], dy.shape[1], x.shape[1], scale, zero_page(dy.device).data_ptr(),
                                 stream()), "cl_weight_grad_tn")


# ------------------------------------------------------------------ normalisation

def groupnorm_ws(B, HW, C_) -> int:
    return int(lib().cl_groupnorm_ws_floats(B, HW, C_))


def groupnorm_fwd(x, y, gamma, beta, B, HW, eps, silu, stats, ws, groups=32):
    _chk(lib().cl_groupnorm_silu_fwd(dt(x), x.data_ptr(), ld(x), y.data_ptr(), ld(y), gamma.data_ptr(),
                                     beta.data_ptr(), B, HW, x.shape[1], groups, eps, int(silu), stats.data_ptr(),
                                     ws.data_ptr(), stream()), "cl_groupnorm_silu_fwd")
    return y


def groupnorm_bwd(x, dy, dx, gamma, beta, stats, B, HW, silu, ws, accum=None, dgamma=None, dbeta=None, groups=32):
    _chk(lib().cl_groupnorm_silu_bwd(dt(x), x.data_ptr(), ld(x), dy.data_ptr(), ld(dy), ptr(accum), ld(accum),
                                     dx.data_ptr(), ld(dx), gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(), B, HW,
                                     x.shape[1], groups, int(silu), ptr(dgamma), ptr(dbeta), ws.data_ptr(), stream()),
         "cl_groupnorm_silu_bwd")
    return dx


def layernorm_fwd(x, y, gamma, beta, eps=1e-5, stats=None):
    _chk(lib().cl_layernorm_fwd(dt(x), x.data_ptr(), ld(x), y.data_ptr(), ld(y), gamma.data_ptr(), beta.data_ptr(),
                                x.shape[0], x.shape[1], eps, ptr(stats), stream()), "cl_layernorm_fwd")
    return y


def layernorm_bwd(x, dy, dx, gamma, stats, accum=None, dgamma=None, dbeta=None):
    _chk(lib().cl_layernorm_bwd(dt(x), x.data_ptr(), ld(x), dy.data_ptr(), ld(dy), ptr(accum), ld(accum),
                                dx.data_ptr(), ld(dx), gamma.data_ptr(), stats.data_ptr(), x.shape[0], x.shape[1],
                                ptr(dgamma), ptr(dbeta), stream()), "cl_layernorm_bwd")
    return dx


# ------------------------------------------------------------------ attention

def attention_fwd(q, k, vt, o, lse, B, H, N, Nkv, dh, scale):
    """q [B*N, >=H*dh], k [B*Nkv, ..], vt [B, H*dh, nkv_pad], o [B*N, H*dh], lse [B,H,lse_stride] or None."""
    _chk(lib().cl_attention_fwd(dt(q), q.data_ptr(), ld(q), k.data_ptr(), ld(k), vt.data_ptr(), vt.shape[-1],
                                o.data_ptr(), ld(o), ptr(lse), 0 if lse is None else lse.shape[-1], B, H, N, Nkv, dh,
                                scale, stream()), "cl_attention_fwd")
    return o


def attention_bwd(q, k, v, o, do, qt, dot, kt, lse, delta, dq, dk, dv, B, H, N, Nkv, dh, scale):
    _chk(lib().cl_attention_bwd(dt(q), q.data_ptr(), ld(q), k.data_ptr(), ld(k), v.data_ptr(), ld(v), o.data_ptr(),
                                ld(o), do.data_ptr(), ld(do), qt.data_ptr(), dot.data_ptr(), qt.shape[-1],
                                kt.data_ptr(), kt.shape[-1], lse.data_ptr(), delta.data_ptr(), lse.shape[-1],
                                dq.data_ptr(), ld(dq), ptr(dk), ld(dk), ptr(dv), ld(dv), B, H, N, Nkv, dh, scale,
                                stream()), "cl_attention_bwd")


ATTN_Q_PRESCALED = 1


def attention_fwd_v2(q, k, v, o, lse, B, H, N, Nkv, dh, scale, q_prescaled=False):
    """bf16, transpose-free: v is [B*Nkv, >=H*dh] like k.  q_prescaled: q holds q * scale * log2(e)
    (CL_ATTN_Q_PRESCALED: the to_q projection applied the factor in its epilogue)."""
    _chk(lib().cl_attention_fwd_v2(dt(q), q.data_ptr(), ld(q), k.data_ptr(), ld(k), v.data_ptr(), ld(v), o.data_ptr(),
                                   ld(o), ptr(lse), 0 if lse is None else lse.shape[-1], B, H, N, Nkv, dh, scale,
                                   ATTN_Q_PRESCALED if q_prescaled else 0, stream()), "cl_attention_fwd_v2")
    return o


def attention_bwd_v2(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, N, Nkv, dh, scale, q_prescaled=False, row_ws=None):
    """row_ws: optional uint8 / any tensor of >= B * H * lse.shape[-1] * 32 bytes (see cl_attention_bwd_v2)."""
    assert row_ws is None or row_ws.numel() * row_ws.element_size() >= B * H * lse.shape[-1] * 32
    _chk(lib().cl_attention_bwd_v2(dt(q), q.data_ptr(), ld(q), k.data_ptr(), ld(k), v.data_ptr(), ld(v), o.data_ptr(),
                                   ld(o), do.data_ptr(), ld(do), lse.data_ptr(), delta.data_ptr(), lse.shape[-1],
                                   dq.data_ptr(), ld(dq), ptr(dk), ld(dk), ptr(dv), ld(dv), B, H, N, Nkv, dh, scale,
                                   ATTN_Q_PRESCALED if q_prescaled else 0, ptr(row_ws), stream()), "cl_attention_bwd_v2")


# ------------------------------------------------------------------ elementwise / layout

def geglu_fwd(h, out):
    _chk(lib().cl_geglu_fwd(dt(h), h.data_ptr(), ld(h), out.data_ptr(), ld(out), h.shape[0], out.shape[1], stream()),
         "cl_geglu_fwd")
    return out


def geglu_bwd(h, dout, dh):
    _chk(lib().cl_geglu_bwd(dt(h), h.data_ptr(), ld(h), dout.data_ptr(), ld(dout), dh.data_ptr(), ld(dh), h.shape[0],
                            dout.shape[1], stream()), "cl_geglu_bwd")
    return dh


def silu_fwd(x, y):
    _chk(lib().cl_silu_fwd(dt(x), x.data_ptr(), y.data_ptr(), x.numel(), stream()), "cl_silu_fwd")
    return y


def silu_bwd(x, dy, dx):
    _chk(lib().cl_silu_bwd(dt(x), x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), stream()), "cl_silu_bwd")
    return dx


def axpby(x, y, a=1.0, b=1.0):
    """y = a*x + b*y over 2-D strided views."""
    _chk(lib().cl_axpby(dt(x), x.data_ptr(), ld(x), y.data_ptr(), ld(y), x.shape[0], x.shape[1], a, b, stream()),
         "cl_axpby")
    return y


def transpose(src, dst, Bt, R, C_, Rpad, ldi=None, bsi=None):
    """src [Bt][R][C] (row stride ldi, batch stride bsi) -> dst [Bt][C][Rpad] (zero padded)."""
    ldi = src.stride(-2) if ldi is None else ldi
    bsi = R * ldi if bsi is None else bsi
    _chk(lib().cl_transpose(dt(src), dt(dst), src.data_ptr(), ldi, bsi, dst.data_ptr(), Rpad, C_ * Rpad, Bt, R, C_,
                            Rpad, stream()), "cl_transpose")
    return dst


def nchw_to_tok(x_nchw, out, Cpad=None):
    B, Cin, H, W = x_nchw.shape
    x_nchw = x_nchw.contiguous()
    _chk(lib().cl_nchw_to_tok(dt(out), x_nchw.data_ptr(), out.data_ptr(), ld(out), B, Cin,
                              out.shape[1] if Cpad is None else Cpad, H * W, stream()), "cl_nchw_to_tok")
    return out


def tok_to_nchw(tok, out_nchw, alpha=1.0, beta=0.0):
    B, C_, H, W = out_nchw.shape
    _chk(lib().cl_tok_to_nchw(dt(tok), tok.data_ptr(), ld(tok), out_nchw.data_ptr(), B, C_, H * W, alpha, beta,
                              stream()), "cl_tok_to_nchw")
    return out_nchw


def colsum(x, out_f32, B, HW, scale=1.0):
    _chk(lib().cl_colsum(dt(x), x.data_ptr(), ld(x), out_f32.data_ptr(), ld(out_f32), B, HW, x.shape[1], scale,
                         stream()), "cl_colsum")
    return out_f32


def pool2x2(src, dst, B, H, W, accumulate=False):
    _chk(lib().cl_pool2x2(dt(src), src.data_ptr(), ld(src), dst.data_ptr(), ld(dst), B, H, W, dst.shape[1],
                          int(accumulate), stream()), "cl_pool2x2")
    return dst


def pack2d(src_f32, dst, Cpad=None):
    """fp32 [R, C] (strided) -> dst dtype [R, Cpad] with zero-filled pad columns."""
    R, C_ = src_f32.shape
    _chk(lib().cl_pack2d(dt(dst), src_f32.data_ptr(), ld(src_f32), dst.data_ptr(), ld(dst), R, C_,
                         dst.shape[1] if Cpad is None else Cpad, stream()), "cl_pack2d")
    return dst


def timestep_embedding(t_long, freqs, out):
    _chk(lib().cl_timestep_embedding(dt(out), t_long.data_ptr(), freqs.data_ptr(), out.data_ptr(), ld(out),
                                     t_long.shape[0], freqs.shape[0], stream()), "cl_timestep_embedding")
    return out


def qsample(z, noise, t, sqrt_ac, sqrt_1mac, out):
    B = z.shape[0]
    _chk(lib().cl_qsample(z.data_ptr(), noise.data_ptr(), t.data_ptr(), sqrt_ac.data_ptr(), sqrt_1mac.data_ptr(),
                          out.data_ptr(), B, z.numel() // B, stream()), "cl_qsample")
    return out


def mse_loss(eps, target, d_eps, loss, gscale=1.0):
    _chk(lib().cl_mse_loss(eps.data_ptr(), target.data_ptr(), ptr(d_eps), loss.data_ptr(), eps.numel(), gscale,
                           stream()), "cl_mse_loss")
    return loss


def p_losses_mse(eps, target, d_eps, t, lvlb, out3, scratch, gscale=1.0, w_simple=1.0, w_elbo=0.0, per_sample=None):
    """Deterministic p_losses reduction: out3 = {loss_simple, loss_vlb, loss}; d_eps = d(w_simple*loss_simple)/d eps."""
    B = eps.shape[0]
    assert scratch.numel() >= 16 * B and eps.is_contiguous() and target.is_contiguous()
    _chk(lib().cl_p_losses_mse(eps.data_ptr(), target.data_ptr(), ptr(d_eps), ptr(t), ptr(lvlb), out3.data_ptr(),
                               ptr(per_sample), scratch.data_ptr(), B, eps.numel() // B, gscale, w_simple, w_elbo,
                               stream()), "cl_p_losses_mse")
    return out3


def conv_tap_gather(x, out, B, Hin, Win, Hout, Wout, tap, stride, pad):
    _chk(lib().cl_conv_tap_gather(dt(x), x.data_ptr(), ld(x), out.data_ptr(), ld(out), B, Hin, Win, Hout, Wout, x.shape[1], tap,
                                  stride, pad, stream()), "cl_conv_tap_gather")
    return out


def softmax_rows(scores_f32, probs, scale=1.0):
    """probs[M,N] (engine dtype) = softmax(scores_f32[M,N] * scale) per row."""
    M, N = scores_f32.shape
    _chk(lib().cl_softmax_rows(dt(probs), scores_f32.data_ptr(), ld(scores_f32), probs.data_ptr(), ld(probs), M, N, scale,
                               stream()), "cl_softmax_rows")
    return probs


def zero_(t):
    """In-place clear of a contiguous tensor: a fill kernel on the current stream (no ATen launch, no memset node: DESIGN.md 5)."""
    assert t.is_contiguous()
    _chk(lib().cl_zero(t.data_ptr(), t.numel() * t.element_size(), stream()), "cl_zero")
    return t


def ddim_step(x, e_c, e_u, noise, coef, index, scale, x_prev, pred_x0=None):
    _chk(lib().cl_ddim_step(x.data_ptr(), e_c.data_ptr(), ptr(e_u), ptr(noise), coef.data_ptr(), index, scale,
                            x_prev.data_ptr(), ptr(pred_x0), x.numel(), stream()), "cl_ddim_step")
    return x_prev


def adamw(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
    _chk(lib().cl_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                        weight_decay, step, grad_scale, stream()), "cl_adamw")


def tick(counter):
    _chk(lib().cl_tick(counter.data_ptr(), stream()), "cl_tick")


def adamw_dev(p, g, m, v, hyper, step):
    """AdamW with device-resident hyper-parameters {lr, b1, b2, eps, wd, grad_scale} and step counter
    (hipGraph-replayable: nothing step-dependent is a kernel argument).  The caller advances `step` with
    tick() ONCE per optimizer step, before the first bank."""
    _chk(lib().cl_adamw_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), hyper.data_ptr(),
                            step.data_ptr(), stream()), "cl_adamw_dev")


def ddim_set_t(table, cursor, S, ts):
    _chk(lib().cl_ddim_set_t(table.data_ptr(), cursor.data_ptr(), S, ts.data_ptr(), ts.numel(), stream()), "cl_ddim_set_t")
    return ts


def ddim_step_dev(x, e_c, e_u, noise, coef, cursor, S, scale, x_prev, pred_x0=None):
    _chk(lib().cl_ddim_step_dev(x.data_ptr(), e_c.data_ptr(), ptr(e_u), ptr(noise), coef.data_ptr(), cursor.data_ptr(),
                                S, scale, x_prev.data_ptr(), ptr(pred_x0), x.numel(), stream()), "cl_ddim_step_dev")
    return x_prev
