"""Weight packing for the HIP engine.

The nn.Module mirror (cldm/, ldm/) owns fp32 parameters under the reference's state-dict
names.  The engine consumes *packed* copies in its storage dtype, laid out for the kernels:

  linear  W [N,K]            -> W (NT operand), Wt = W^T [K,N] (data-gradient operand)
  LoRA    down A [r,K], up B [N,r]  -> A, At = A^T [K,r], B, Bt = B^T [r,N]
  conv3x3 W [O,I,3,3]        -> Wp [O][ky][kx][I]  and  Wd [I][2-ky][2-kx][O] (data gradient)
  conv1x1 W [O,I,1,1]        -> as linear

Frozen weights are packed once.  Trainables (LoRA A/B, zero convs, the `norm` layers --
cldm/cldm_ctrlora_finetune.py:84-108) live in ONE flat fp32 master buffer with ONE flat fp32
gradient buffer (ordered by backward completion so that the DP all-reduce can be bucketed and
overlapped); their packed copies are refreshed after every optimizer step by the pack /
transpose kernels.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from .. import hip


def rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class Trainable:
    name: str
    shape: Tuple[int, ...]                  # shape of the MASTER storage (conv3 layout: [O, 9 * I_pad])
    offset: int = 0
    master: Optional[torch.Tensor] = None   # fp32 view into the flat master buffer
    grad: Optional[torch.Tensor] = None     # fp32 view into the flat grad buffer
    conv: Optional[Tuple[int, int, int]] = None   # (O, I, I_pad): 3x3 conv weight kept as [O][ky][kx][I_pad]

    def to_master(self, w: torch.Tensor) -> torch.Tensor:
        """State-dict tensor -> master layout."""
        if self.conv is None:
            return w.reshape(self.shape)
        O, I, Ip = self.conv
        m = torch.zeros(O, 3, 3, Ip, dtype=torch.float32, device=w.device)
        m[..., :I] = w.permute(0, 2, 3, 1)
        return m.reshape(O, 9 * Ip)

    def param_view(self, flat_view: torch.Tensor) -> torch.Tensor:
        """View of master / grad storage in the nn.Parameter's (state-dict) shape."""
        if self.conv is None:
            return flat_view
        O, I, Ip = self.conv
        return flat_view.view(O, 3, 3, Ip)[..., :I].permute(0, 3, 1, 2)


class TrainableSet:
    """Flat fp32 master + gradient storage for the optimizer's parameter subset."""

    def __init__(self):
        self.items: List[Trainable] = []
        self.by_name: Dict[str, Trainable] = {}
        self.flat: Optional[torch.Tensor] = None
        self.flat_grad: Optional[torch.Tensor] = None
        self.numel = 0

    def declare(self, name: str, shape, conv=None) -> Trainable:
        t = Trainable(name, tuple(shape), conv=conv)
        self.items.append(t)
        self.by_name[name] = t
        return t

    def materialize(self, sd: Dict[str, torch.Tensor], device):
        off = 0
        for t in self.items:
            t.offset = off
            n = 1
            for s in t.shape:
                n *= s
            off += rup(n, 64)            # keep every tensor 256-byte aligned
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=device)
        for t in self.items:
            n = 1
            for s in t.shape:
                n *= s
            t.master = self.flat[t.offset:t.offset + n].view(t.shape)
            t.grad = self.flat_grad[t.offset:t.offset + n].view(t.shape)
            t.master.copy_(t.to_master(sd[t.name].to(device=device, dtype=torch.float32)))


class LinearW:
    """nn.Linear / 1x1 conv / LoRACompatibleLinear in packed form."""

    def __init__(self, W: torch.Tensor, bias: Optional[torch.Tensor], dtype, device, need_bwd: bool, keep_f32: bool = False):
        W = W.reshape(W.shape[0], -1).to(device=device, dtype=torch.float32)
        self.N, self.K = W.shape
        self.W = W.to(dtype).contiguous()
        # inference executors that fold the LoRA keep the fp32 base weight: Wm = storage(W_fp32 + B A) is then ONE rounding of
        # the merged weight -- exactly the error any stored weight has -- instead of round(round(W) + B A)
        self.W32 = W.contiguous() if keep_f32 else None
        self.Wt = W.t().to(dtype).contiguous() if need_bwd else None
        self.bias = None if bias is None else bias.to(device=device, dtype=torch.float32).contiguous()
        self.r = 0
        # LoRA (trainable) pieces
        self.tA: Optional[Trainable] = None
        self.tB: Optional[Trainable] = None
        self.A = self.At = self.B = self.Bt = None
        # inference executors: W + B A folded into one packed weight (cldm/lora.py:_fuse_lora does the same on the
        # module; the reference sanctions it for inference) -- no down-projection launch, no second K segment
        self.Wm: Optional[torch.Tensor] = None
        # trainable dense weight (zero convs)
        self.tW: Optional[Trainable] = None
        self.tb: Optional[Trainable] = None
        self.dtype = dtype

    def load(self, W: torch.Tensor, bias: Optional[torch.Tensor]):
        """Refresh the packed copies of a FROZEN weight in place (module.load_state_dict after the executor was
        built: the optimizer keeps pointing at the same executor / flat buffers)."""
        W = W.reshape(W.shape[0], -1).to(device=self.W.device, dtype=torch.float32)
        assert tuple(W.shape) == (self.N, self.K)
        self.W.copy_(W)
        if self.W32 is not None:
            self.W32.copy_(W)
        if self.Wt is not None:
            self.Wt.copy_(W.t())
        if bias is not None and self.bias is not None and self.tb is None:
            self.bias.copy_(bias.to(device=self.W.device, dtype=torch.float32))
        self.invalidate_geglu()

    # ---- GEGLU-fused projection (no-grad forwards): rows permuted so that every 160-column tile of the
    # product holds 80 value columns followed by their 80 gate columns (csrc/gemm.h: ACT_GEGLU)
    def geglu_ok(self) -> bool:
        return self.N % 320 == 0 and (self.N // 2) % 80 == 0

    def geglu_pack(self):
        g = self.__dict__.get("_geglu")
        if g is None:
            half = self.N // 2
            j = torch.arange(half // 80, device=self.W.device).repeat_interleave(160)
            c = torch.arange(160, device=self.W.device).repeat(half // 80)
            perm = torch.where(c < 80, j * 80 + c, half + j * 80 + (c - 80))
            Wg = (self.W if self.Wm is None else self.Wm).index_select(0, perm).contiguous()
            bg = None if self.bias is None else self.bias.index_select(0, perm).contiguous()
            Bg = self.B.index_select(0, perm).contiguous() if (self.r and self.Wm is None) else None
            g = (Wg, bg, Bg)
            self.__dict__["_geglu"] = g
        return g

    def invalidate_geglu(self):
        self.__dict__.pop("_geglu", None)

    def attach_lora(self, tA: Trainable, tB: Trainable, device):
        self.tA, self.tB = tA, tB
        self.r = tA.shape[0]
        self.A = torch.empty(self.r, self.K, dtype=self.dtype, device=device)
        self.At = torch.empty(self.K, self.r, dtype=self.dtype, device=device)
        self.B = torch.empty(self.N, self.r, dtype=self.dtype, device=device)
        self.Bt = torch.empty(self.r, self.N, dtype=self.dtype, device=device)

    def merge_lora(self):
        """Wm = storage-dtype(W + B A) from the packed base weight and the fp32 LoRA masters (weight-load time only)."""
        if self.tA is None:
            return
        base = self.W32 if self.W32 is not None else self.W.float()
        m = torch.addmm(base, self.tB.master.view(self.N, self.r), self.tA.master.view(self.r, self.K))
        if self.Wm is None:
            self.Wm = m.to(self.dtype).contiguous()
        else:
            self.Wm.copy_(m)
        self.invalidate_geglu()

    def attach_trainable_weight(self, tW: Trainable, tb: Optional[Trainable]):
        self.tW, self.tb = tW, tb

    def repack(self):
        """Refresh packed copies from the fp32 masters (pack / transpose kernels)."""
        if self.tA is not None:
            hip.pack2d(self.tA.master, self.A)
            hip.transpose(self.tA.master, self.At, 1, self.r, self.K, self.r)
            hip.pack2d(self.tB.master, self.B)
            hip.transpose(self.tB.master, self.Bt, 1, self.N, self.r, self.N)
        if self.tW is not None:
            w2 = self.tW.master.view(self.N, self.K)
            hip.pack2d(w2, self.W)
            if self.Wt is not None:
                hip.transpose(w2, self.Wt, 1, self.N, self.K, self.N)
            if self.tb is not None:
                self.bias = self.tb.master


class LoraGroup:
    """G LoRACompatibleLinears with the SAME input and shape (to_q | to_k | to_v of a self-attention, to_k | to_v of a
    cross-attention: cldm/lora.py:285-291, attention.py:163-170) packed side by side, so that the G products run as grouped
    launches (csrc/gemm.h: a2_group_n / a1_group_n) instead of G launches each:

        forward    t   = x [A_1; ..; A_G]^T                        one down-projection launch, x read once
                   y   = [x | t_g] . [W_g | B_g]^T  for all g      one launch, output [M, G N]
        backward   u   = [dy_g B_g]_g                               one launch (first segment grouped)
                   dx  = [dy | u] . [W_1^T .. W_G^T | A_1^T .. A_G^T]   one launch, K = G N + G r

    The members' packed tensors become VIEWS of the group's buffers: re-packing (cl_repack writes through the members'
    pointers with explicit row strides), reload_frozen and merge_lora keep working on the members."""

    def __init__(self, members: "List[LinearW]"):
        L0 = members[0]
        assert all(L.N == L0.N and L.K == L0.K and L.r == L0.r and L.r > 0 for L in members)
        assert all((L.bias is None) == (L0.bias is None) for L in members)
        self.members = list(members)
        # biases side by side (frozen fp32 vectors; members keep views so reload_frozen writes through)
        self.bias = None
        if L0.bias is not None:
            self.bias = torch.cat([L.bias for L in members], 0).contiguous()
            for g, L in enumerate(members):
                L.bias = self.bias[g * L0.N:(g + 1) * L0.N]
        G, N, K, r = len(members), L0.N, L0.K, L0.r
        self.G, self.N, self.K, self.r = G, N, K, r
        dev, dt_ = L0.W.device, L0.dtype
        need_bwd = L0.Wt is not None
        self.W = torch.cat([L.W for L in members], 0).contiguous()                       # [G N, K]
        self.Wt = torch.cat([L.Wt for L in members], 1).contiguous() if need_bwd else None   # [K, G N]
        self.A = torch.empty(G * r, K, dtype=dt_, device=dev)
        self.At = torch.empty(K, G * r, dtype=dt_, device=dev)
        self.B = torch.empty(G * N, r, dtype=dt_, device=dev)
        self.Bt = torch.empty(G * r, N, dtype=dt_, device=dev)
        self.Wm = None
        for g, L in enumerate(members):
            L.W = self.W[g * N:(g + 1) * N]
            if need_bwd:
                L.Wt = self.Wt[:, g * N:(g + 1) * N]
            L.A, L.At = self.A[g * r:(g + 1) * r], self.At[:, g * r:(g + 1) * r]
            L.B, L.Bt = self.B[g * N:(g + 1) * N], self.Bt[g * r:(g + 1) * r]
            L.group = self

    def enable_merge(self):
        """Inference executors (W + B A folded): the members' merged weights live side by side -> ONE product for the group."""
        if self.Wm is None:
            self.Wm = torch.empty(self.G * self.N, self.K, dtype=self.members[0].dtype, device=self.W.device)
            for g, L in enumerate(self.members):
                L.Wm = self.Wm[g * self.N:(g + 1) * self.N]


class Conv3W:
    """3x3 conv in implicit-GEMM form; channels padded to multiples of 32 where needed."""

    def __init__(self, W: torch.Tensor, bias: torch.Tensor, dtype, device, need_bwd: bool):
        O, I = W.shape[0], W.shape[1]
        self.O, self.I = O, I
        self.Op, self.Ip = rup(O, 32), rup(I, 32)
        self.Wp = torch.empty(self.Op, 9 * self.Ip, dtype=dtype, device=device)
        self.Wd = torch.empty(self.Ip, 9 * self.Op, dtype=dtype, device=device) if need_bwd else None
        self.bias = torch.zeros(self.Op, dtype=torch.float32, device=device)
        self.tW: Optional[Trainable] = None     # trainable weight (Base-ControlNet pre-training), master [O][9][Ip]
        self.tb: Optional[Trainable] = None
        self._phase = {}                        # phase-packed forms (phase_weights), built on first use, dropped by load()
        self.load(W, bias)

    def phase_weights(self, kind: str) -> torch.Tensor:
        """Weights of the phase-decomposed products (include/ctrlora_hip.h: CL_GEMM_CONV_UP2P / T2P), built once from the packed
        forward weights of a FROZEN conv.

        'up2' (Upsample.conv over the nearest-x2 input, openaimodel.py:108-118): output pixel (2y + a, 2x + b) reads source
        rows {y - 1, y} (a = 0: tap ky = 0 | taps 1 + 2) or {y, y + 1} (a = 1: taps 0 + 1 | tap 2), columns likewise: the 3x3
        taps that land on one source pixel are SUMMED (fp32, one rounding to the compute dtype).  [4 phases][Op][2x2][Ip].
        't2' (data gradient of Downsample's stride-2 conv, :150; in = 2 out + k - 1): an even input row gets tap ky = 1 of
        output row y, an odd one tap 2 of row y and tap 0 of row y + 1.  Phase (a, b): [Ip][(1 + a)(1 + b)][Op], concatenated."""
        if kind in self._phase:
            return self._phase[kind]
        assert self.tW is None, "phase-packed weights are built once: frozen convs only"
        W = self.Wp.float().view(self.Op, 3, 3, self.Ip)                     # [o][ky][kx][i]
        if kind == "up2":
            grp = {(0, 0): [0], (0, 1): [1, 2], (1, 0): [0, 1], (1, 1): [2]}
            out = torch.empty(4, self.Op, 4, self.Ip, dtype=torch.float32, device=W.device)
            for a in range(2):
                for b in range(2):
                    for ty in range(2):
                        for tx in range(2):
                            out[2 * a + b, :, 2 * ty + tx] = W[:, grp[(a, ty)]][:, :, grp[(b, tx)]].sum(dim=(1, 2))
            packed = out.reshape(-1)
        elif kind == "t2":
            taps = {0: [1], 1: [2, 0]}                                        # window offset 0, +1 -> forward tap index
            blocks = []
            for a in range(2):
                for b in range(2):
                    blk = torch.stack([W[:, ky, kx].t() for ky in taps[a] for kx in taps[b]], dim=1)   # [Ip][nt][Op]
                    blocks.append(blk.reshape(-1))
            packed = torch.cat(blocks)
        elif kind == "up2d":
            # data gradient of 'up2' on the source grid (CL_GEMM_CONV_S2K4): window row ky4 = 0..3 <-> upsampled row 2y - 1 + ky4,
            # which the forward reached from (phase a, window tap ty) = (1, 1), (0, 1), (1, 0), (0, 0): taps {2}, {1, 2}, {0, 1}, {0}
            grp4 = [[2], [1, 2], [0, 1], [0]]
            out = torch.empty(self.Ip, 4, 4, self.Op, dtype=torch.float32, device=W.device)
            for ky4 in range(4):
                for kx4 in range(4):
                    out[:, ky4, kx4] = W[:, grp4[ky4]][:, :, grp4[kx4]].sum(dim=(1, 2)).t()
            self._phase[kind] = out.reshape(self.Ip, 16 * self.Op).to(self.Wp.dtype).contiguous()
            return self._phase[kind]
        else:
            raise ValueError(kind)
        self._phase[kind] = packed.to(self.Wp.dtype).contiguous().view(1, -1)     # (ldw1 is not used by these modes)
        return self._phase[kind]

    def attach_trainable(self, tW: Trainable, tb: Trainable):
        assert self.O == self.Op, "trainable 3x3 convs have O % 32 == 0"
        self.tW, self.tb = tW, tb

    def repack_rows(self):
        """cl_repack descriptor rows (8 longs each) refreshing Wp / Wd from the master [O][9][Ip]."""
        es = self.Wp.element_size()
        rows = [[self.tW.offset, (self.O << 32) | (9 * self.Ip), self.Wp.data_ptr(), 0, 0, 0, 0, 0]]
        if self.Wd is not None:
            for t in range(9):     # Wd[i][8 - t][o] = W[o][t][i]
                rows.append([self.tW.offset + t * self.Ip, (self.O << 32) | self.Ip, 0,
                             self.Wd.data_ptr() + (8 - t) * self.Op * es, 9 * self.Ip, 0, 9 * self.Op, 0])
        return rows

    def load(self, W: torch.Tensor, bias: torch.Tensor):
        """(Re)pack in place: [O][ky][kx][I] and the tap-flipped data-gradient form [I][2-ky][2-kx][O]."""
        device = self.Wp.device
        self._phase = {}
        W = W.to(device=device, dtype=torch.float32)
        assert W.shape[0] == self.O and W.shape[1] == self.I
        Wpad = torch.zeros(self.Op, self.Ip, 3, 3, dtype=torch.float32, device=device)
        Wpad[:self.O, :self.I] = W
        self.Wp.copy_(Wpad.permute(0, 2, 3, 1).reshape(self.Op, 9 * self.Ip))
        if self.Wd is not None:
            self.Wd.copy_(Wpad.flip(2, 3).permute(1, 2, 3, 0).reshape(self.Ip, 9 * self.Op))
        self.bias[:self.O] = bias.to(device=device, dtype=torch.float32)


class NormW:
    def __init__(self, gamma, beta, device):
        self.gamma = gamma.to(device=device, dtype=torch.float32).contiguous()
        self.beta = beta.to(device=device, dtype=torch.float32).contiguous()
        self.tg: Optional[Trainable] = None
        self.tb: Optional[Trainable] = None

    def attach(self, tg: Trainable, tb: Trainable):
        self.tg, self.tb = tg, tb

    def load(self, gamma, beta):
        if self.tg is None:      # trainable norms alias the flat masters (updated through the bound Parameters)
            self.gamma.copy_(gamma.to(device=self.gamma.device, dtype=torch.float32))
            self.beta.copy_(beta.to(device=self.beta.device, dtype=torch.float32))

    def repack(self):
        if self.tg is not None:
            self.gamma, self.beta = self.tg.master, self.tb.master

    @property
    def ggamma(self):
        return None if self.tg is None else self.tg.grad

    @property
    def gbeta(self):
        return None if self.tb is None else self.tb.grad
