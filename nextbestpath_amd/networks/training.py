"""Training-mode forward / backward of NBP on the HIP kernels (csrc/nbp_train.hip + the implicit-GEMM
convolution of csrc/nbp_conv.hip).

Reference: next_best_path/utility/nbp_utils.py:340-395 (train_experience_data) drives
``nbp.train(); out1, out2 = nbp(x); loss = nbp.loss(...); loss.backward()``; the layers are
next_best_path/networks/nbp_model.py:8-62.  Here every layer is a ``torch.autograd.Function`` whose
forward and backward are C-ABI kernel launches; torch's autograd engine only threads them together
(plumbing) and torch.optim.AdamW applies the update, as the survey's build plan allows.

Activations are NHWC ``[B,H,W,C]`` fp32; channel counts are padded to multiples of 64 where the
matrix-core kernels need it (network input 5->64, F_int 32->64, final 8/1->64) and sliced back, so
the parameter gradients have the reference's shapes.
"""
from __future__ import annotations

import os

import ctypes

import torch

from .. import _lib


def _st():
    return _lib.current_stream()


def _up(v, m=64):
    return (v + m - 1) // m * m


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _chk(rc, what):
    _lib.check(rc, what)


def _colsum(x2d, rows=None):
    M, C = x2d.shape
    L = _lib.lib()
    out = torch.empty(C, dtype=torch.float32, device=x2d.device)
    ws = _ws(L.nbp_colreduce_workspace_bytes(M, C), x2d.device)
    _chk(L.nbp_colsum_f32(_lib.ptr(x2d), _lib.ptr(rows), M, C, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _st()), "colsum")
    return out


def _pad_channels(x, cout):
    B, H, W, C = x.shape
    if C == cout:
        return x
    out = torch.empty(B, H, W, cout, dtype=torch.float32, device=x.device)
    _chk(_lib.lib().nbp_pad_channels_f32(_lib.ptr(x), B * H * W, C, cout, _lib.ptr(out), _st()), "pad_channels")
    return out


def _slice_channels(x, c0, cs):
    B, H, W, C = x.shape
    if c0 == 0 and cs == C:
        return x
    out = torch.empty(B, H, W, cs, dtype=torch.float32, device=x.device)
    _chk(_lib.lib().nbp_slice_channels_f32(_lib.ptr(x), B * H * W, C, c0, cs, _lib.ptr(out), _st()), "slice_channels")
    return out


def _igemm(src0, src1, ups, wpk, N, ksize, scale, shift, relu):
    L = _lib.lib()
    B, Hs, Ws, C0 = src0.shape
    H, W = (2 * Hs, 2 * Ws) if ups else (Hs, Ws)
    C1 = 0 if src1 is None else src1.shape[3]
    out = torch.empty(B, H, W, N, dtype=torch.float32, device=src0.device)
    ws = _ws(L.nbp_conv_igemm_workspace_bytes(B, H, W, N, 0), src0.device)
    _chk(L.nbp_conv_igemm_f32(_lib.ptr(src0), C0, _lib.ptr(src1), C1, int(ups), B, H, W, ksize, _lib.ptr(wpk), N,
                              _lib.ptr(scale), _lib.ptr(shift), int(relu), _lib.ptr(out), 0, 0, _lib.ptr(ws), ws.numel(),
                              _st()), "conv_igemm")
    return out


# 3x3 convolutions of the forward and of the data gradient on the fp16 matrix pipe through two-piece operand splitting
# (csrc/nbp_split.hip: the fp32 pipe's accuracy at 5.3x its matrix rate); NBP_TRAIN_SPLIT=0 keeps the fp32 MFMA pipe.
_SPLIT = _lib.tune("NBP_TRAIN_SPLIT", "1") != "0"
_WGRAD_SPLIT = _lib.tune("NBP_TRAIN_WGRAD_SPLIT", "1") == "1"      # A/B: weight gradients on the fp32 pipe


def _split_ok(H, W, N, ksize):
    return _SPLIT and ksize == 3 and H % 16 == 0 and ((W % 32 == 0 and N % 64 == 0) or (W % 16 == 0 and N % 128 == 0))


def _pack_split(w_oihw, n_pad, c_total):
    """OIHW fp32 [N, C, 3, 3] -> (hi/lo fp16 planes for N padded to n_pad rows and C to c_total channels, max |w| word)."""
    N, C, k, _ = w_oihw.shape
    # the pack kernel writes every plane entry of the channels it is given: only padded channels need the zero fill
    alloc = torch.zeros if C != c_total else torch.empty
    planes = alloc(c_total // 16 * 9 * 4 * n_pad * 8, dtype=torch.int16, device=w_oihw.device)
    wamax = _fresh_slots(w_oihw.device)[:1]          # a zeroed word of the forward's arena (no memset launch per layer)
    # the pack kernel indexes rows by the padded count: give it a zero-padded weight when N < n_pad
    if N != n_pad:
        wp = torch.zeros(n_pad, C, k, k, dtype=torch.float32, device=w_oihw.device)
        wp[:N] = w_oihw
        w_oihw = wp
    _chk(_lib.lib().nbp_pack_conv_weight_split_prezeroed(_lib.ptr(w_oihw), n_pad, C, 3, c_total, _lib.ptr(planes), _lib.ptr(wamax),
                                                         _st()), "pack_split")
    return planes, wamax


# split-K request of the training convolutions: -1 = slices by occupancy only (csrc/nbp_split.hip: nbp_plan_conv_split; the
# accuracy-driven chain bound of the eval forward costs the step partial sums, ~25 reduce launches and the BatchNorm statistics of
# every layer it splits); NBP_TRAIN_CHAIN_BOUND=1: 0 = the eval forward's plan
_TRAIN_SK = 0 if _lib.tune("NBP_TRAIN_CHAIN_BOUND", "0") == "1" else -1
_CONST = {}


def _const(value, n, device):
    """Read-only vector of n copies of `value` (epilogue scales / shifts that are all ones or zeros): one fill per size, not per call."""
    key = (float(value), int(n), str(device))
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full((n,), float(value), dtype=torch.float32, device=device)
    return t


# ---- hand-off of what a producer already knows about its output to the convolution that consumes it.  The note rides ON the
# tensor object (a Python attribute: it lives and dies with the tensor, and a different tensor that happens to reuse the address
# -- a forward under no_grad, recompute, two forwards before one backward -- can never pick it up; round 3 keyed process-global
# dicts by data_ptr()).  A consumer that gets a tensor without a note (autograd summed two gradients, a slice, a copy) takes its
# own pass.  NBP_TRAIN_FUSE=0 switches the hand-off off altogether.
_FUSE = _lib.tune("NBP_TRAIN_FUSE", "1") == "1"
_BN_EPILOGUE = _lib.tune("NBP_TRAIN_BN_EPILOGUE", "1") == "1"      # BatchNorm statistics from the producing convolution's epilogue (0 = its own pass)
_SLICE_VIEWS = _lib.tune("NBP_TRAIN_SLICE_VIEWS", "1") == "1"      # two-source convolutions return channel-slice VIEWS of dx (0 = copies)
_MASK_FROM_X = _lib.tune("NBP_TRAIN_MASK_FROM_X", "1") == "1"      # BatchNorm backward: ReLU mask rebuilt from x (0 = read y, as round 3)
_ARENA = {}


def _note(t, **kw):
    """Attach producer knowledge to tensor t: amax = 64-word max-|t| slot; colsum = column sums of t (with its max slot)."""
    if _FUSE:
        t._nbp_note = kw
    return t


HANDOFF_STATS = {"hit": 0, "miss": 0}      # notes found / not found by consumers (tools/bench_train.py prints them)


def _noted(t, key):
    n = getattr(t, "_nbp_note", None) if _FUSE and t is not None else None
    v = None if n is None else n.get(key)
    HANDOFF_STATS["hit" if v is not None else "miss"] += 1
    return v



def _fresh_slots(device, n=1):
    """n zeroed 64-word slots out of a per-device arena that is cleared ONCE per training forward (forward_train) instead of a
    torch.zeros per slot; falls back to an allocation when the arena is used up."""
    a = _ARENA.get(device)
    if a is None or a["next"] + n > a["cap"]:
        return torch.zeros(n * 64, dtype=torch.int32, device=device)
    k = a["next"]
    a["next"] += n
    return a["buf"][k * 64:(k + n) * 64]


def _reset_arena(device, cap=1024):
    """A FRESH zeroed arena per training forward (one fill): the slots an earlier forward handed to its autograd contexts are
    views into that forward's own arena, which stays alive (and untouched) for as long as they reference it."""
    _ARENA[device] = {"buf": torch.zeros(cap * 64, dtype=torch.int32, device=device), "cap": cap, "next": 0}


def _amax_slot(*tensors):
    """64-word max-|.| slot (csrc/nbp_split.hip) over the given tensors: one streaming pass each, shared by every kernel that
    scales them (forward + weight gradient for a layer's inputs; data + weight gradient for its output gradient)."""
    live = [t for t in tensors if t is not None and t.numel()]
    if _FUSE and len(live) == 1:
        known = _noted(live[0], "amax")
        if known is not None:
            return known                       # the producer (BatchNorm apply) measured it while writing the tensor
    slot = _fresh_slots(tensors[0].device)
    for t in live:
        known = _noted(t, "amax")
        if known is not None and len(live) > 1:
            # two sources share one slot: fold the known maximum in (element-wise max of the 64 words, non-negative float bits)
            torch.maximum(slot, known, out=slot)
            continue
        _chk(_lib.lib().nbp_amax_f32(_lib.ptr(t), t.numel(), _lib.ptr(slot), _st()), "amax")
    return slot


def _bn_part(out):
    """Buffer for the BatchNorm partial sums a convolution's epilogue may leave beside its output [B, H, W, N] (rows x [2][N] doubles)."""
    B, H, W, N = out.shape
    return torch.empty(_lib.lib().nbp_conv_bn_part_rows(B, H, W) * 2 * N, dtype=torch.float64, device=out.device), ctypes.c_int(0)


def _conv_split(src0, src1, ups, packed, N, scale, shift, relu, amax=None, bn=False):
    L = _lib.lib()
    planes, wamax = packed
    B, Hs, Ws, C0 = src0.shape
    H, W = (2 * Hs, 2 * Ws) if ups else (Hs, Ws)
    C1 = 0 if src1 is None else src1.shape[3]
    out = torch.empty(B, H, W, N, dtype=torch.float32, device=src0.device)
    ws = _ws(L.nbp_conv_split_planned_workspace_bytes_k(B, H, W, C0 + C1, N, int(ups), _TRAIN_SK, None), src0.device)     # the slices the planner will use
    # max |x| of the inputs: the caller's slot, else taken inside the call (autograd hands tensors over without their history)
    if bn:      # the BatchNorm behind this layer gets the column sums of the output from the epilogue (when the launch has them)
        part, rows = _bn_part(out)
        _chk(L.nbp_conv3x3_split_bn_f32(_lib.ptr(src0), C0, _lib.ptr(src1), C1, int(ups), B, H, W, _lib.ptr(planes), _lib.ptr(wamax), N,
                                        _lib.ptr(scale), _lib.ptr(shift), int(relu), _lib.ptr(out), _lib.ptr(amax), None, _TRAIN_SK, _lib.ptr(ws),
                                        ws.numel(), _lib.ptr(part), ctypes.byref(rows), _st()), "conv3x3_split_bn")
        if rows.value > 0:
            _note(out, bnpart=(part, rows.value))
        return out
    _chk(L.nbp_conv3x3_split_f32(_lib.ptr(src0), C0, _lib.ptr(src1), C1, int(ups), B, H, W, _lib.ptr(planes), _lib.ptr(wamax), N,
                                 _lib.ptr(scale), _lib.ptr(shift), int(relu), _lib.ptr(out), _lib.ptr(amax), None, _TRAIN_SK, _lib.ptr(ws),
                                 ws.numel(), _st()), "conv3x3_split")
    return out


# 1x1 layers (the attention gates' W_g / W_x) and their data gradients on the split scheme through the gates' kernel with one source
# (csrc/nbp_split.hip: nbp_conv1x1_split_f32) instead of the fp32 pipe's implicit GEMM: no padding of 32 output channels to 64,
# memory-bound on every level.  NBP_TRAIN_SPLIT_1X1=0: the implicit GEMM.
_SPLIT_1X1 = _lib.tune("NBP_TRAIN_SPLIT_1X1", "1") == "1"


def _conv1x1_ok(M, C, N, c_real):
    return _SPLIT and _SPLIT_1X1 and C % 32 == 0 and N % 32 == 0 and c_real == C and M * C * 4 < 2 ** 31 and N * C * 4 < 2 ** 31


def _conv1x1_split(x, planes, wamax, N, scale, shift, amax):
    """x [B,H,W,C] -> [B,H,W,N] through the 1x1 split kernel; amax = the 64-word max-|x| slot."""
    B, H, W, C = x.shape
    out = torch.empty(B, H, W, N, dtype=torch.float32, device=x.device)
    _chk(_lib.lib().nbp_conv1x1_split_f32(_lib.ptr(x), C, B * H * W, _lib.ptr(planes), _lib.ptr(wamax), N, _lib.ptr(scale),
                                          _lib.ptr(shift), 0, _lib.ptr(out), _lib.ptr(amax), _st()), "conv1x1_split")
    return out


# data gradient of the up_conv layers in parity form (csrc/nbp_split.hip: conv3x3_halo_h2_kernel<..., DG>): dx at the low resolution
# straight from dy, 16 tap-products per low-resolution pixel instead of 36 + a 2x2 sum pass.  NBP_TRAIN_UP_DGRAD=0: round 4's form.
_UP_DGRAD = _lib.tune("NBP_TRAIN_UP_DGRAD", "1") == "1"
# ... and their weight gradient (wgrad_up_split_kernel: 16 tap-GEMMs over M / 4 pixels instead of 9 over M).  NBP_TRAIN_UP_WGRAD=0: the 3x3 form.
_UP_WGRAD = _lib.tune("NBP_TRAIN_UP_WGRAD", "1") == "1"


def _upconv_ok(Hs, Ws, N):
    """up_conv layers (x2 nearest upsample + 3x3): four 2x2 parity convolutions of the low-resolution input when it tiles."""
    return _SPLIT and Hs % 16 == 0 and ((Ws % 32 == 0 and N % 64 == 0) or (Ws % 16 == 0 and N % 128 == 0))


def _upconv_split(src, w_oihw, n_pad, scale, shift, amax=None, bn=False, pre=None):
    L = _lib.lib()
    N, C, _, _ = w_oihw.shape
    B, Hs, Ws, C0 = src.shape
    if pre is not None and N == n_pad and C == C0:
        planes, wamax = pre[3], pre[5]              # packed at the start of the forward (_prepack_run)
    else:
        if N != n_pad or C != C0:       # zero rows / channels up to the padded counts
            wp = torch.zeros(n_pad, C0, 3, 3, dtype=torch.float32, device=w_oihw.device)
            wp[:N, :C] = w_oihw
            w_oihw = wp
        planes = torch.empty(4 * (C0 // 16) * 4 * 4 * n_pad * 8, dtype=torch.int16, device=src.device)
        wamax = torch.empty(1, dtype=torch.int32, device=src.device)             # zeroed by the pack launch itself
        _chk(L.nbp_pack_upconv_weight_split(_lib.ptr(w_oihw), n_pad, C0, _lib.ptr(planes), _lib.ptr(wamax), _st()), "pack_upconv")
    H, W = 2 * Hs, 2 * Ws
    out = torch.empty(B, H, W, n_pad, dtype=torch.float32, device=src.device)
    ws = _ws(L.nbp_conv_split_planned_workspace_bytes_k(B, H, W, C0, n_pad, 1, _TRAIN_SK, None), src.device)
    if bn:
        part, rows = _bn_part(out)
        _chk(L.nbp_upconv3x3_split_bn_f32(_lib.ptr(src), C0, B, H, W, _lib.ptr(planes), _lib.ptr(wamax), n_pad, _lib.ptr(scale),
                                          _lib.ptr(shift), 0, _lib.ptr(out), _lib.ptr(amax), None, _TRAIN_SK, _lib.ptr(ws), ws.numel(),
                                          _lib.ptr(part), ctypes.byref(rows), _st()), "upconv3x3_split_bn")
        if rows.value > 0:
            _note(out, bnpart=(part, rows.value))
        return out
    _chk(L.nbp_upconv3x3_split_f32(_lib.ptr(src), C0, B, H, W, _lib.ptr(planes), _lib.ptr(wamax), n_pad, _lib.ptr(scale),
                                   _lib.ptr(shift), 0, _lib.ptr(out), _lib.ptr(amax), None, _TRAIN_SK, _lib.ptr(ws), ws.numel(), _st()), "upconv3x3_split")
    return out


# ---- all weight packs of a step in two launches (csrc/nbp_split.hip: nbp_prepack_weights_split).  forward_train builds (once per set of
# parameter storages) a device table of the layers the split kernels take -- 3x3, up_conv and the gates' 1x1 layers -- with persistent
# plane buffers, runs the two launches at the start of every forward, and the layers look their planes up by the weight's address;
# a layer called outside forward_train (tests) packs for itself as before.  NBP_TRAIN_PREPACK=0: every layer packs for itself.
_PREPACK = _lib.tune("NBP_TRAIN_PREPACK", "1") == "1"
_PRE_CACHE = {}          # key (device, parameter addresses) -> table
_PRE = None              # the current step's {weight address: (kind, N, C, planes, planes_t, wamax)}


def _prepack_layers(net):
    out = []
    for name, mod in net.named_modules():
        w = getattr(mod, "weight", None)
        if not isinstance(mod, torch.nn.Conv2d) or w is None:
            continue
        N, C, k, _ = w.shape
        if N % 16 or C % 16:
            continue                                   # Conv1.conv.0 (5 channels), psi, the heads
        if k == 3:
            out.append((w, 2 if ".up." in name + "." else 0, N, C))
        elif k == 1 and N % 32 == 0 and C % 32 == 0:
            out.append((w, 1, N, C))
    return out


def _prepack_run(net, dev):
    """Packs every layer's weights for this step; returns the lookup table (None: switched off / nothing to pack)."""
    import numpy as np
    if not (_PREPACK and _SPLIT):
        return None
    L = _lib.lib()
    layers = _prepack_layers(net)
    if not layers:
        return None
    # the key holds what the device table froze at build time: every weight's address AND its shape / dtype / element count (a
    # parameter whose storage was replaced and later landed on the same address with another shape must not reuse stale N / C;
    # ADVICE r05).  Assumptions documented beside ctx.pre: one stream, and weights unchanged between a forward and ITS backward
    # (an optimizer step comes after the backward; the next forward re-packs into the same persistent planes).
    key = (str(dev),) + tuple((w.data_ptr(), tuple(w.shape), str(w.dtype), w.untyped_storage().data_ptr(), w.numel()) for w, *_ in layers)
    tab = _PRE_CACHE.get(key)
    if tab is None:
        while len(_PRE_CACHE) >= 2:                    # (two networks at most -- e.g. a trained and a frozen copy: the buffers are
            _PRE_CACHE.pop(next(iter(_PRE_CACHE)))     #  2 x the weights each; the least recently built one goes)
        assert L.nbp_prepack_desc_bytes() == 48
        n = len(layers)
        wamax = torch.zeros(n, dtype=torch.int32, device=dev)
        rec = np.zeros(n, dtype=np.dtype([("w", "<u8"), ("planes", "<u8"), ("planes_t", "<u8"), ("wamax", "<u8"), ("N", "<i4"), ("C", "<i4"),
                                            ("kind", "<i4"), ("pad", "<i4")]))
        look, keep = {}, []
        for i, (w, kind, N, C) in enumerate(layers):
            taps = {0: 9, 1: 1, 2: 16}[kind]
            planes = torch.empty(taps * 4 * N * C // 2, dtype=torch.int16, device=dev)      # 4 B per (tap, weight): hi + lo fp16
            planes_t = torch.empty(taps * 4 * N * C // 2, dtype=torch.int16, device=dev)
            rec[i] = (w.data_ptr(), planes.data_ptr(), planes_t.data_ptr(), wamax.data_ptr() + 4 * i, N, C, kind, 0)
            look[w.data_ptr()] = (kind, N, C, planes, planes_t, wamax[i:i + 1])
            keep.append((w, w.untyped_storage()))      # the storages themselves, not only the Parameter objects
        descs = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        tab = _PRE_CACHE[key] = {"descs": descs, "wamax": wamax, "look": look, "n": n, "keep": keep}
    _chk(L.nbp_prepack_weights_split(_lib.ptr(tab["descs"]), tab["n"], _lib.ptr(tab["wamax"]), _st()), "prepack")
    return tab["look"]


def _prepacked(w, kind, N, C):
    """This step's planes of weight tensor w, or None."""
    if _PRE is None:
        return None
    e = _PRE.get(w.data_ptr())
    return e if (e is not None and e[0] == kind and e[1] == N and e[2] == C) else None


class ConvFn(torch.autograd.Function):
    """y = conv_k(cat(x0, x1) [x2 nearest-upsampled]) + bias; weight OIHW [N, c_real, k, k].
    x0 / x1 channel counts are multiples of 64 (c_real < C0 only for the zero-padded network input)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, ups, bn_next=False):
        L = _lib.lib()
        N, c_real, k, _ = weight.shape
        C0 = x0.shape[3]
        C1 = 0 if x1 is None else x1.shape[3]
        Ctot, Np = C0 + C1, _up(N)
        dev = x0.device
        M = x0.shape[0] * x0.shape[1] * x0.shape[2]
        one_by_one = k == 1 and not ups and x1 is None and _conv1x1_ok(M, C0, N, c_real)
        if one_by_one:
            Np = N                         # (no padding to 64 columns: the kernel takes 32-column blocks)
        w = weight.detach().contiguous()
        scale = _const(1.0, Np, dev)
        if N == Np:                        # (every 3x3 layer: the bias is the epilogue's shift as it stands)
            shift = bias.detach().contiguous()
        else:
            shift = torch.zeros(Np, dtype=torch.float32, device=dev)
            shift[:N] = bias.detach()
        H, W = (x0.shape[1] * 2, x0.shape[2] * 2) if ups else (x0.shape[1], x0.shape[2])
        xmax = None                      # joint max-|.| slot of the inputs: taken once, reused by the weight gradient
        wmax_fwd = None
        # bn_next: a BatchNorm consumes this output -- its statistics' partial sums come out of the epilogue (not for padded
        # channel counts, whose output is sliced; not under an observer, which may rewrite the output)
        bn = bool(bn_next) and _BN_EPILOGUE and N == Np and _observer is None
        # this step's prepacked planes of the layer (kind, N, C, planes, planes_t, wamax).  They live in ONE persistent buffer per
        # layer that every forward_train() of this network overwrites: backward(t) after a LATER forward_train() is only right while
        # the weights are unchanged in between and everything runs on one stream (true of the trainer: forward, backward, step)
        pre = None
        if ups and k == 3 and x1 is None and _upconv_ok(x0.shape[1], x0.shape[2], Np):
            xmax = _amax_slot(x0)
            pre = _prepacked(w, 2, N, c_real) if (N == Np and c_real == C0) else None
            y = _upconv_split(x0, w, Np, scale, shift, xmax, bn, pre)
        elif _split_ok(H, W, Np, k):
            xmax = _amax_slot(x0, x1)
            pre = _prepacked(w, 0, N, c_real) if (N == Np and c_real == Ctot) else None
            packed = (pre[3], pre[5]) if pre is not None else _pack_split(w, Np, Ctot)
            wmax_fwd = packed[1]                       # max |w|: the data gradient's planes hold the same values
            y = _conv_split(x0, x1, ups, packed, Np, scale, shift, False, xmax, bn)
        elif one_by_one:
            xmax = _amax_slot(x0)
            pre = _prepacked(w, 1, N, C0)
            if pre is not None:
                planes, wamax = pre[3], pre[5]
            else:
                planes = torch.empty(C0 // 16 * 4 * N * 8, dtype=torch.int16, device=dev)
                wamax = torch.empty(1, dtype=torch.int32, device=dev)
                _chk(L.nbp_pack_conv_weight_split(_lib.ptr(w), N, C0, 1, None, 0, C0, _lib.ptr(planes), _lib.ptr(wamax), _st()), "pack_split_1x1")
            y = _conv1x1_split(x0, planes, wamax, N, scale, shift, xmax)
        else:
            wpk = torch.empty(Ctot // 32 * k * k * Np * 32, dtype=torch.float32, device=dev)
            _chk(L.nbp_pack_conv_weight_padded(_lib.ptr(w), N, c_real, k, Ctot, Np, _lib.ptr(wpk), _st()), "pack_fwd")
            y = _igemm(x0, x1, ups, wpk, Np, k, scale, shift, False)
        ctx.save_for_backward(x0, x1 if x1 is not None else torch.empty(0, device=dev), w)
        ctx.xmax = xmax
        ctx.wmax_fwd = wmax_fwd
        ctx.pre = pre
        ctx.meta = (N, c_real, k, C0, C1, Np, bool(ups), x1 is not None, one_by_one)
        return _slice_channels(y, 0, N)

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x0, x1, w = ctx.saved_tensors
        N, c_real, k, C0, C1, Np, ups, has1, one_by_one = ctx.meta
        x1 = x1 if has1 else None
        dev = dy.device
        info = _noted(dy, "colsum") if dy.is_contiguous() else None       # (max-|dy| slot, column sums): this very tensor's
        dy_note = getattr(dy, "_nbp_note", None)
        dy = dy.contiguous()
        dy = _pad_channels(dy, Np)
        B, H, W, _ = dy.shape
        M = B * H * W
        # the bias gradient: the BatchNorm behind this convolution summed dx's columns while writing it
        if info is not None and info[1].numel() == N:
            db = info[1]                               # handed over as it is: nothing else keeps it (the note lets go of it here),
            dy_note.pop("colsum", None)                # so autograd takes the tensor instead of cloning it
        else:
            db = info[1][:N].clone() if info is not None else _colsum(dy.view(M, Np))[:N].clone()
        dw = torch.empty(N, c_real, k, k, dtype=torch.float32, device=dev)
        # (the fp32-pipe weight-gradient kernels of the 1x1 layers take 64-column blocks of dy: a 32-channel gate pads here only)
        # ... unless the 1x1 split form takes the layer (few input channels, many pixels: it masks the missing columns itself)
        own_1x1 = one_by_one and _SPLIT and _WGRAD_SPLIT and C0 % 64 == 0 and C0 <= 128 and (B * H * W) % 64 == 0
        dyw, Nw = (dy, Np) if (Np % 64 == 0 or own_1x1) else (_pad_channels(dy, _up(Np)), _up(Np))
        ws = _ws(L.nbp_conv_wgrad_workspace_bytes(B, H, W, C0, C1, Nw, k), dev)
        # the 3x3 weight gradients take the split scheme too (the entry point falls through to the fp32 pipe for the rest)
        dymax = None
        up_parity = (ups and k == 3 and not has1 and N == Np and c_real == C0 and _SPLIT and _WGRAD_SPLIT and ctx.xmax is not None
                     and B * H * W * Np * 4 < 2 ** 31)
        wsu = L.nbp_upconv_wgrad_split_workspace_bytes(B, H // 2, W // 2, C0, Np) if (up_parity and _UP_WGRAD) else 0
        if wsu:
            dymax = info[0] if info is not None else _amax_slot(dy)
            wsb = _ws(wsu, dev)
            _chk(L.nbp_upconv_wgrad_split_f32(_lib.ptr(x0), C0, B, H // 2, W // 2, _lib.ptr(dy), Np, _lib.ptr(dw), _lib.ptr(ctx.xmax),
                                              _lib.ptr(dymax), _lib.ptr(wsb), wsb.numel(), _st()), "upconv_wgrad_split")
        elif _SPLIT and _WGRAD_SPLIT:
            # max |dy|: shared with the data gradient below; measured by the BatchNorm backward when dy came from one (padding
            # channels are zeros: same maximum)
            dymax = (info[0] if info is not None else _amax_slot(dy)) if (k == 3 or one_by_one) else None
            _chk(L.nbp_conv_wgrad_split_f32(_lib.ptr(x0), C0, _lib.ptr(x1), C1, int(ups), B, H, W, k, _lib.ptr(dyw), Nw, c_real, N,
                                            _lib.ptr(dw), _lib.ptr(ctx.xmax), _lib.ptr(ctx.xmax), _lib.ptr(dymax), _lib.ptr(ws),
                                            ws.numel(), _st()), "conv_wgrad_split")
        else:
            _chk(L.nbp_conv_wgrad_f32(_lib.ptr(x0), C0, _lib.ptr(x1), C1, int(ups), B, H, W, k, _lib.ptr(dyw), Nw, c_real, N,
                                      _lib.ptr(dw), _lib.ptr(ws), ws.numel(), _st()), "conv_wgrad")
        dx0 = dx1 = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            Ctot = C0 + C1
            one, zero = _const(1.0, Ctot, dev), _const(0.0, Ctot, dev)
            if (ups and k == 3 and not has1 and _UP_DGRAD and N == Np and c_real == C0 and _upconv_ok(H // 2, W // 2, C0)
                    and B * H * W * Np * 4 < 2 ** 31):
                pre = getattr(ctx, "pre", None)
                if pre is not None and pre[0] == 2:
                    planes, wamax = pre[4], pre[5]
                else:
                    planes = torch.empty(32 * Np * C0, dtype=torch.int16, device=dev)
                    wamax = torch.empty(1, dtype=torch.int32, device=dev)
                    _chk(L.nbp_pack_upconv_weight_split_dgrad(_lib.ptr(w), Np, C0, _lib.ptr(planes), _lib.ptr(wamax), _st()), "pack_upconv_dgrad")
                if dymax is None:
                    dymax = info[0] if info is not None else _amax_slot(dy)
                dxl = torch.empty(B, H // 2, W // 2, C0, dtype=torch.float32, device=dev)
                wsd = _ws(L.nbp_upconv_split_dgrad_workspace_bytes(B, H // 2, W // 2, Np, C0), dev)
                _chk(L.nbp_upconv3x3_split_dgrad_f32(_lib.ptr(dy), Np, B, H // 2, W // 2, _lib.ptr(planes), _lib.ptr(wamax), C0,
                                                     _lib.ptr(one), _lib.ptr(zero), _lib.ptr(dxl), _lib.ptr(dymax), None, _lib.ptr(wsd),
                                                     wsd.numel(), _st()), "upconv_dgrad_split")
                return dxl, None, dw, db, None, None
            if one_by_one:
                # dx = dy W^T: a 1x1 convolution from N to C0 channels on the same kernel
                pre = getattr(ctx, "pre", None)
                if pre is not None and pre[0] == 1:
                    planes, wamax = pre[4], pre[5]
                else:
                    planes = torch.empty(N // 16 * 4 * C0 * 8, dtype=torch.int16, device=dev)
                    wamax = torch.empty(1, dtype=torch.int32, device=dev)
                    _chk(L.nbp_pack_conv1x1_weight_split_dgrad(_lib.ptr(w), N, C0, _lib.ptr(planes), _lib.ptr(wamax), _st()), "pack_dgrad_1x1")
                if dymax is None:
                    dymax = info[0] if info is not None else _amax_slot(dy)
                dx = _conv1x1_split(dy, planes, wamax, C0, one, zero, dymax)
            elif _split_ok(H, W, Ctot, k):
                # dx = conv3x3(dy, w^T with the taps reversed): output channels = the (padded) input channels
                if c_real == Ctot and N == Np:
                    # flip + permute + pack in one launch (they were an ATen flip, a strided copy and the pack)
                    pre = getattr(ctx, "pre", None)
                    planes = pre[4] if (pre is not None and pre[0] == 0) else torch.empty(Np // 16 * 9 * 4 * Ctot * 8, dtype=torch.int16, device=dev)
                    if pre is not None and pre[0] == 0:
                        wamax = pre[5]
                    elif getattr(ctx, "wmax_fwd", None) is not None:      # the forward's pack of the same weights measured max |w|
                        wamax = ctx.wmax_fwd
                        _chk(L.nbp_pack_conv_weight_split_dgrad_known(_lib.ptr(w), N, c_real, Np, _lib.ptr(planes), _lib.ptr(wamax), _st()),
                             "pack_dgrad_split")
                    else:
                        wamax = torch.empty(1, dtype=torch.int32, device=dev)
                        _chk(L.nbp_pack_conv_weight_split_dgrad(_lib.ptr(w), N, c_real, Np, _lib.ptr(planes), _lib.ptr(wamax), _st()), "pack_dgrad_split")
                    packed = (planes, wamax)
                else:
                    wt = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()            # [c_real, N, 3, 3]
                    packed = _pack_split(wt, Ctot, Np)
                dx = _conv_split(dy, None, False, packed, Ctot, one, zero, False, dymax)
            else:
                wt = torch.empty(Np // 32 * k * k * Ctot * 32, dtype=torch.float32, device=dev)
                _chk(L.nbp_pack_conv_weight_dgrad(_lib.ptr(w), N, c_real, k, Ctot, Np, _lib.ptr(wt), _st()), "pack_dgrad")
                dx = _igemm(dy, None, False, wt, Ctot, k, one, zero, False)        # [B,H,W,Ctot] at output resolution
            if ups:
                low = torch.empty(B, H // 2, W // 2, Ctot, dtype=torch.float32, device=dev)
                _chk(L.nbp_sum2x2_f32(_lib.ptr(dx), B, H // 2, W // 2, Ctot, _lib.ptr(low), _st()), "sum2x2")
                dx = low
            if has1 and _SLICE_VIEWS:
                # views of the joint gradient: the attention gate's RowScaleFn reads its slice in place, autograd's sum of dd's two
                # gradients reads the other (the two slice copies were 2 x (C0 + C1) x M x 4 bytes per decoder level)
                dx0, dx1 = dx[..., :C0], dx[..., C0:]
            elif has1:
                dx0, dx1 = _slice_channels(dx, 0, C0), _slice_channels(dx, C0, C1)
            else:
                dx0 = dx
        return dx0, dx1, dw, db, None, None


# Conv1.conv.0 straight from the NCHW network input (csrc/nbp_first_conv.h forward, wgrad_first_kernel backward) instead of a
# 64-channel padded copy through the 64 -> 64 kernels.  NBP_TRAIN_FIRST_CONV=0: round 4's form.
_FIRST_CONV = _lib.tune("NBP_TRAIN_FIRST_CONV", "1") == "1"


class FirstConvFn(torch.autograd.Function):
    """y [B,S,S,64] (NHWC) = conv3x3(x [B,5,S,S] NCHW) + bias (nbp_model.py:66); the network input takes no gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        L = _lib.lib()
        B, _, H, W = x.shape
        dev = x.device
        w = weight.detach().contiguous()
        y = torch.empty(B, H, W, 64, dtype=torch.float32, device=dev)
        _chk(L.nbp_conv_first_linear_f32(_lib.ptr(x), B, H, W, _lib.ptr(w), _lib.ptr(_const(1.0, 64, dev)), _lib.ptr(bias.detach().contiguous()),
                                  _lib.ptr(y), _st()), "conv_first")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        (x,) = ctx.saved_tensors
        B, _, H, W = x.shape
        dev = dy.device
        info = _noted(dy, "colsum") if dy.is_contiguous() else None
        dy = dy.contiguous()
        db = info[1] if (info is not None and info[1].numel() == 64) else _colsum(dy.view(B * H * W, 64))
        dw = torch.empty(64, 5, 3, 3, dtype=torch.float32, device=dev)
        ws = _ws(L.nbp_conv_first_wgrad_workspace_bytes(), dev)
        _chk(L.nbp_conv_first_wgrad_f32(_lib.ptr(x), B, H, W, _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(ws), ws.numel(), _st()), "conv_first_wgrad")
        return None, dw, db


class BNFn(torch.autograd.Function):
    """nn.BatchNorm2d in training mode (+ optional fused ReLU); updates the running statistics in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu):
        L = _lib.lib()
        shp = x.shape
        C = shp[-1]
        M = x.numel() // C
        dev = x.device
        x = x.contiguous()
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        ws = _ws(L.nbp_colreduce_workspace_bytes(M, C), dev)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        slot = _fresh_slots(dev) if _FUSE and C % 4 == 0 else None
        # the backward rebuilds the ReLU mask from x through the unrounded statistics (two tensor reads less per BatchNorm): not
        # when an observer may rewrite y after the fact (its mask is then y's, and y is what gets saved)
        stat = torch.empty(4 * C, dtype=torch.float64, device=dev) if (_MASK_FROM_X and relu and C % 4 == 0 and _observer is None) else None       # mean | invstd | mask bounds lo | hi
        part = _noted(x, "bnpart") if stat is not None else None
        if part is not None:
            _chk(L.nbp_bn_train_forward_part4_f32(_lib.ptr(x), M, C, _lib.ptr(g), _lib.ptr(b), float(eps), float(momentum),
                                                  _lib.ptr(running_mean), _lib.ptr(running_var), int(relu), _lib.ptr(mean),
                                                  _lib.ptr(invstd), _lib.ptr(y), _lib.ptr(slot), _lib.ptr(stat), stat.numel(),
                                                  _lib.ptr(part[0]), part[1], _lib.ptr(_const(0.0, C, dev)), _st()), "bn_fwd_part")
        elif stat is not None:
            _chk(L.nbp_bn_train_forward_stat4_f32(_lib.ptr(x), M, C, _lib.ptr(g), _lib.ptr(b), float(eps), float(momentum),
                                                  _lib.ptr(running_mean), _lib.ptr(running_var), int(relu), _lib.ptr(mean),
                                                  _lib.ptr(invstd), _lib.ptr(y), _lib.ptr(slot), _lib.ptr(stat), stat.numel(),
                                                  _lib.ptr(ws), ws.numel(), _st()), "bn_fwd")
        else:
            _chk(L.nbp_bn_train_forward_amax_f32(_lib.ptr(x), M, C, _lib.ptr(g), _lib.ptr(b), float(eps), float(momentum),
                                                 _lib.ptr(running_mean), _lib.ptr(running_var), int(relu), _lib.ptr(mean),
                                                 _lib.ptr(invstd), _lib.ptr(y), _lib.ptr(slot), _lib.ptr(ws), ws.numel(), _st()), "bn_fwd")
        if slot is not None:
            _note(y, amax=slot)
        if stat is not None:
            ctx.save_for_backward(x, stat, mean, invstd, g, b)
        else:
            ctx.save_for_backward(x, y, mean, invstd, g)
        ctx.mask_from_x = stat is not None
        ctx.relu = bool(relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        if ctx.mask_from_x:
            x, stat, mean, invstd, g, beta = ctx.saved_tensors
            y = None
        else:
            x, y, mean, invstd, g = ctx.saved_tensors
        C = x.shape[-1]
        M = x.numel() // C
        dev = x.device
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        ws = _ws(L.nbp_colreduce_workspace_bytes(M, C), dev)
        fuse = _FUSE and C % 4 == 0
        slot = _fresh_slots(dev) if fuse else None
        csum = torch.empty(C, dtype=torch.float32, device=dev) if fuse else None
        if ctx.mask_from_x:
            _chk(L.nbp_bn_train_backward_stat_f32(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(stat), _lib.ptr(beta), M, C, _lib.ptr(mean),
                                                  _lib.ptr(invstd), _lib.ptr(g), int(ctx.relu), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db),
                                                  _lib.ptr(csum), _lib.ptr(slot), _lib.ptr(ws), ws.numel(), _st()), "bn_bwd")
        else:
            _chk(L.nbp_bn_train_backward_fused_f32(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(y), M, C, _lib.ptr(mean), _lib.ptr(invstd),
                                                   _lib.ptr(g), int(ctx.relu), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(csum),
                                                   _lib.ptr(slot), _lib.ptr(ws), ws.numel(), _st()), "bn_bwd")
        if fuse:
            _note(dx, colsum=(slot, csum))
        return dx, dg, db, None, None, None, None, None


# An activation with several consumers (the encoder skips: max-pool + two uses per attention gate; x5 and every up_conv output: two)
# goes through FanOutFn: n aliases forward, and the consumers' gradients meet in ONE n-ary sum launch (csrc/nbp_train.hip:
# sum_n4_kernel, n reads + 1 write) instead of autograd's n - 1 binary adds (3 (n - 1) tensor passes; 19 ATen launches per step).
# NBP_TRAIN_FANOUT=0: autograd sums (round 5).
_FANOUT = _lib.tune("NBP_TRAIN_FANOUT", "1") == "1"


class FanOutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        live = [g for g in grads if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        shp = live[0].shape
        C = shp[-1]
        M = live[0].numel() // C
        ok = C % 4 == 0 and len(live) <= 8
        srcs, lds = [], []
        for g in live:
            # a contiguous tensor, or a channel slice of a wider NHWC tensor (rows ld floats apart): read in place
            ld = g.stride(-2) if g.dim() >= 2 else C
            dense = g.dim() == 4 and g.stride(3) == 1 and ld >= C and ld % 4 == 0 and g.stride(1) == shp[2] * ld and \
                g.stride(0) == shp[1] * shp[2] * ld and g.data_ptr() % 16 == 0
            if not dense:
                g, ld = g.contiguous(), C
                ok = ok and g.data_ptr() % 16 == 0
            srcs.append(g)
            lds.append(ld)
        if not ok:
            out = srcs[0] + srcs[1]
            for g in srcs[2:]:
                out = out + g
            return out, None
        out = torch.empty(shp, dtype=torch.float32, device=live[0].device)
        n = len(srcs)
        ptrs = (ctypes.c_void_p * n)(*[g.data_ptr() for g in srcs])
        ldv = (ctypes.c_longlong * n)(*lds)
        _chk(_lib.lib().nbp_sum_n_f32(n, ptrs, ldv, M, C, _lib.ptr(out), _st()), "sum_n")
        return out, None


def _fan(x, n):
    """n aliases of x for its n consumers (each keeps what the producer noted about x)."""
    if not (_FANOUT and n > 1 and torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    outs = FanOutFn.apply(x, n)
    note = getattr(x, "_nbp_note", None)
    if note is not None:
        for o in outs:
            o._nbp_note = dict(note)
    return outs


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, H, W, C = x.shape
        x = x.contiguous()
        y = torch.empty(B, H // 2, W // 2, C, dtype=torch.float32, device=x.device)
        _chk(_lib.lib().nbp_maxpool2_nhwc_f32(_lib.ptr(x), B, H, W, C, _lib.ptr(y), _st()), "maxpool")
        ctx.save_for_backward(x)
        known = _noted(x, "amax")
        if known is not None:
            _note(y, amax=known)                   # a max-pool keeps the maximum
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        _chk(_lib.lib().nbp_maxpool2_backward_f32(_lib.ptr(x), _lib.ptr(dy.contiguous()), B, H, W, C, _lib.ptr(dx), _st()),
             "maxpool_bwd")
        return dx


def _ew(op, a, b=None):
    out = torch.empty_like(a)
    _chk(_lib.lib().nbp_elementwise_f32(op, _lib.ptr(a), _lib.ptr(b), a.numel(), _lib.ptr(out), _st()), "elementwise")
    return out


class AddReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        y = _ew(0, a.contiguous(), b.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        g = _ew(1, dy.contiguous(), y)
        return g, g


class SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        y = _ew(2, a.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return _ew(3, dy.contiguous(), y)


class PsiConvFn(torch.autograd.Function):
    """1x1 convolution F -> 1 (Attention_block.psi.0): p[m] = q[m,:] . w + b."""

    @staticmethod
    def forward(ctx, q, weight, bias):
        L = _lib.lib()
        B, H, W, Fc = q.shape
        M = B * H * W
        q = q.contiguous()
        w = weight.detach().reshape(-1).contiguous()
        p = torch.empty(M, dtype=torch.float32, device=q.device)
        _chk(L.nbp_rowdot_f32(_lib.ptr(q), _lib.ptr(w), 1, M, Fc, _lib.ptr(p), _st()), "rowdot")
        out = _ew(5, p, bias.detach().contiguous())
        ctx.save_for_backward(q, w)
        ctx.wshape = weight.shape
        return out.view(B, H, W, 1)

    @staticmethod
    def backward(ctx, dp):
        L = _lib.lib()
        q, w = ctx.saved_tensors
        B, H, W, Fc = q.shape
        M = B * H * W
        dp = dp.contiguous().view(M)
        dq = torch.empty_like(q)
        _chk(L.nbp_outer_f32(_lib.ptr(dp), _lib.ptr(w), M, Fc, _lib.ptr(dq), _st()), "outer")
        dw = _colsum(q.view(M, Fc), dp).view(ctx.wshape)
        db = _colsum(dp.view(M, 1))
        return dq, dw, db


class RowScaleFn(torch.autograd.Function):
    """out[m, c] = x[m, c] * s[m]  (the attention gate's  x * psi)."""

    @staticmethod
    def forward(ctx, x, s):
        B, H, W, C = x.shape
        x, s = x.contiguous(), s.contiguous()
        out = torch.empty_like(x)
        if _FUSE and C % 4 == 0 and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0:
            slot = _fresh_slots(x.device)              # max |x psi| for the convolution that consumes the gated tensor
            _chk(_lib.lib().nbp_rowscale_amax_f32(_lib.ptr(x), _lib.ptr(s), B * H * W, C, _lib.ptr(out), _lib.ptr(slot), _st()), "rowscale_amax")
            _note(out, amax=slot)
        else:
            _chk(_lib.lib().nbp_rowscale_f32(_lib.ptr(x), _lib.ptr(s), B * H * W, C, _lib.ptr(out), _st()), "rowscale")
        ctx.save_for_backward(x, s)
        return out

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, s = ctx.saved_tensors
        B, H, W, C = x.shape
        M = B * H * W
        dx = torch.empty_like(x)
        ds = torch.empty(M, dtype=torch.float32, device=x.device)
        # one pass over dy for both gradients; a channel slice of a two-source convolution's joint gradient (ConvFn.backward
        # returns views) is read in place through its row stride
        ld = dy.stride(2) if dy.dim() == 4 else 0
        if not (dy.dim() == 4 and dy.stride(3) == 1 and ld >= C and ld % 4 == 0 and dy.stride(1) == W * ld
                and dy.stride(0) == H * W * ld and dy.data_ptr() % 16 == 0):
            dy, ld = dy.contiguous(), C
        if C % 4 == 0:
            _chk(L.nbp_rowscale_backward_f32(dy.data_ptr(), ld, _lib.ptr(x), _lib.ptr(s), M, C, _lib.ptr(dx), _lib.ptr(ds), _st()), "rowscale_bwd")
        else:
            dy = dy.contiguous()
            _chk(L.nbp_rowscale_f32(_lib.ptr(dy), _lib.ptr(s), M, C, _lib.ptr(dx), _st()), "rowscale")
            _chk(L.nbp_rowdot_f32(_lib.ptr(dy), _lib.ptr(x), 0, M, C, _lib.ptr(ds), _st()), "rowdot")
        return dx, ds.view(s.shape)


# The attention gate's element-wise middle as ONE Function (csrc/nbp_train.hip: gate_mid_*): BN_g, BN_x, add-relu and the psi row-dot in one
# pass over the two 1x1 convolutions' outputs, and the whole of its backward in a reduce + an apply pass -- every output bit-identical to
# the separate Functions' (BNFn x 2, AddReluFn, PsiConvFn: rounds 3-5), which NBP_TRAIN_GATE_FUSE=0 brings back.
_GATE_FUSE = _lib.tune("NBP_TRAIN_GATE_FUSE", "1") == "1"


def _gate_mid_ok(gp, xp):
    F = gp.shape[-1]
    F4 = F // 4
    return (_GATE_FUSE and _observer is None and gp.shape == xp.shape and F % 4 == 0 and 4 <= F4 <= 64 and F4 & (F4 - 1) == 0
            and gp.is_contiguous() and xp.is_contiguous() and gp.data_ptr() % 16 == 0 and xp.data_ptr() % 16 == 0)


class GateMidFn(torch.autograd.Function):
    """p = psi_conv(relu(BN_g(g_pre) + BN_x(x_pre))) (nbp_model.py:52-58, train mode); running statistics updated in place."""

    @staticmethod
    def forward(ctx, gp, xp, gam_g, bet_g, rm_g, rv_g, eps_g, mom_g, gam_x, bet_x, rm_x, rv_x, eps_x, mom_x, w_psi, b_psi):
        L = _lib.lib()
        B, H, W, F = gp.shape
        M = B * H * W
        dev = gp.device
        f32 = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
        mean_g, inv_g, mean_x, inv_x = f32(F), f32(F), f32(F), f32(F)
        stat_g = torch.empty(4 * F, dtype=torch.float64, device=dev)
        stat_x = torch.empty(4 * F, dtype=torch.float64, device=dev)
        q = torch.empty_like(gp)
        p = f32(M)
        gg, bg = gam_g.detach().contiguous(), bet_g.detach().contiguous()
        gx, bx = gam_x.detach().contiguous(), bet_x.detach().contiguous()
        w = w_psi.detach().reshape(-1).contiguous()
        ws = _ws(L.nbp_gate_mid_workspace_bytes(M, F), dev)
        _chk(L.nbp_gate_mid_forward_f32(_lib.ptr(gp), _lib.ptr(xp), M, F, _lib.ptr(gg), _lib.ptr(bg), _lib.ptr(rm_g), _lib.ptr(rv_g),
                                        float(eps_g), float(mom_g), _lib.ptr(gx), _lib.ptr(bx), _lib.ptr(rm_x), _lib.ptr(rv_x), float(eps_x),
                                        float(mom_x), _lib.ptr(mean_g), _lib.ptr(inv_g), _lib.ptr(mean_x), _lib.ptr(inv_x),
                                        _lib.ptr(stat_g), _lib.ptr(stat_x), _lib.ptr(w), _lib.ptr(b_psi.detach().contiguous()), _lib.ptr(q),
                                        _lib.ptr(p), _lib.ptr(ws), ws.numel(), _st()), "gate_mid_forward")
        ctx.save_for_backward(gp, xp, q, mean_g, inv_g, gg, mean_x, inv_x, gx, w)
        ctx.wshape = w_psi.shape
        return p.view(B, H, W, 1)

    @staticmethod
    def backward(ctx, dp):
        L = _lib.lib()
        gp, xp, q, mean_g, inv_g, gg, mean_x, inv_x, gx, w = ctx.saved_tensors
        B, H, W, F = gp.shape
        M = B * H * W
        dev = gp.device
        dp = dp.contiguous().view(M)
        f32 = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
        dgp, dxp = torch.empty_like(gp), torch.empty_like(xp)
        dgam_g, dbet_g, dgam_x, dbet_x, dw, cs_g, cs_x = f32(F), f32(F), f32(F), f32(F), f32(F), f32(F), f32(F)
        slots = _fresh_slots(dev, 2) if _FUSE else None
        sl_g, sl_x = (slots[:64], slots[64:]) if slots is not None else (None, None)
        ws = _ws(L.nbp_gate_mid_workspace_bytes(M, F), dev)
        _chk(L.nbp_gate_mid_backward_f32(_lib.ptr(dp), _lib.ptr(w), _lib.ptr(q), _lib.ptr(gp), _lib.ptr(xp), M, F, _lib.ptr(mean_g),
                                         _lib.ptr(inv_g), _lib.ptr(gg), _lib.ptr(mean_x), _lib.ptr(inv_x), _lib.ptr(gx), _lib.ptr(dgp),
                                         _lib.ptr(dxp), _lib.ptr(dgam_g), _lib.ptr(dbet_g), _lib.ptr(dgam_x), _lib.ptr(dbet_x), _lib.ptr(dw),
                                         _lib.ptr(cs_g), _lib.ptr(cs_x), _lib.ptr(sl_g), _lib.ptr(sl_x), _lib.ptr(ws), ws.numel(), _st()),
             "gate_mid_backward")
        if _FUSE:        # what BNFn.backward hands the 1x1 convolutions in front: max |.| slot and column sums of their dy
            _note(dgp, colsum=(sl_g, cs_g))
            _note(dxp, colsum=(sl_x, cs_x))
        db = _colsum(dp.view(M, 1))
        return dgp, dxp, dgam_g, dbet_g, None, None, None, None, dgam_x, dbet_x, None, None, None, None, dw.view(ctx.wshape), db


class ToNCHWFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, H, W, C = x.shape
        out = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
        _chk(_lib.lib().nbp_nhwc_to_nchw_f32(_lib.ptr(x.contiguous()), B, C, H, W, _lib.ptr(out), _st()), "to_nchw")
        return out

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = dy.shape
        out = torch.empty(B, H, W, C, dtype=torch.float32, device=dy.device)
        _chk(_lib.lib().nbp_nchw_to_nhwc_f32(_lib.ptr(dy.contiguous()), B, C, H, W, _lib.ptr(out), _st()), "to_nhwc")
        return out


class GatherValuesFn(torch.autograd.Function):
    """pred[k] = out1[b, c, x, y] (nbp_utils.py:379); coords int64 [K,4] = (b, c, x, y)."""

    @staticmethod
    def forward(ctx, out1, coords):
        B, C, H, W = out1.shape
        K = coords.shape[0]
        coords = coords.contiguous()
        pred = torch.empty(K, dtype=torch.float32, device=out1.device)
        _chk(_lib.lib().nbp_gather_values_f32(_lib.ptr(out1.contiguous()), _lib.ptr(coords), K, C, H, W, _lib.ptr(pred), _st()),
             "gather_values")
        ctx.save_for_backward(coords)
        ctx.shape = (B, C, H, W)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        (coords,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        d = torch.zeros(B, C, H, W, dtype=torch.float32, device=dpred.device)
        _chk(_lib.lib().nbp_scatter_values_f32(_lib.ptr(dpred.contiguous()), _lib.ptr(coords), coords.shape[0], C, H, W,
                                               _lib.ptr(d), _st()), "scatter_values")
        return d, None


class MeanLossFn(torch.autograd.Function):
    """mode 0: F.mse_loss(p, t); mode 1: F.binary_cross_entropy(p, t) (mean reduction)."""

    @staticmethod
    def forward(ctx, p, t, mode):
        p, t = p.contiguous(), t.contiguous()
        acc = torch.empty(1, dtype=torch.float64, device=p.device)
        ws = _ws(512 * 8 + 256, p.device)
        _chk(_lib.lib().nbp_loss_f32(mode, _lib.ptr(p), _lib.ptr(t), p.numel(), 1.0, _lib.ptr(acc), None, _lib.ptr(ws),
                                     ws.numel(), _st()), "loss")
        ctx.save_for_backward(p, t)
        ctx.mode = mode
        return (acc / p.numel()).to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        acc = torch.empty(1, dtype=torch.float64, device=p.device)
        dp = torch.empty_like(p)
        ws = _ws(512 * 8 + 256, p.device)
        _chk(_lib.lib().nbp_loss_f32(ctx.mode, _lib.ptr(p), _lib.ptr(t), p.numel(), float(g.item()), _lib.ptr(acc),
                                     _lib.ptr(dp), _lib.ptr(ws), ws.numel(), _st()), "loss_grad")
        return dp, None, None


# ---------------------------------------------------------------------------------------------- network
# Forward observer: a callable (name, tensor) invoked on every named intermediate (NHWC fp32) right after it is computed -- the
# train-mode counterpart of a module forward hook (the layers here are autograd Functions, not modules).  None by default; what
# an observer does with the tensor is its own business (tests/hip_helpers.py::teacher_forcing overwrites it with an exact
# evaluation's value so that a backward can be checked at that linearisation point).
_observer = None


def set_forward_observer(fn):
    """Installs (or, with None, removes) the observer; returns the previous one."""
    global _observer
    prev, _observer = _observer, fn
    return prev


def _t(name, y):
    if _observer is not None and name is not None:
        _observer(name, y)
        if hasattr(y, "_nbp_note"):
            del y._nbp_note             # an observer may have rewritten the values: what the producer measured no longer holds
    return y


_NBT = None          # inside forward_train: the BatchNorm layers' num_batches_tracked tensors seen so far


def _bn(mod, x, relu, name=None):
    y = BNFn.apply(x, mod.weight, mod.bias, mod.running_mean, mod.running_var, mod.eps, mod.momentum, relu)
    if _NBT is not None:
        _NBT.append(mod.num_batches_tracked)       # forward_train bumps all of a forward's counters in one launch
    else:
        with torch.no_grad():
            mod.num_batches_tracked += 1
    return _t(name, y)


def _block(seq, x0, x1=None, name=""):
    """conv_block (ref :8-21): (conv3x3 -> BN -> ReLU) x 2 on cat(x0, x1)."""
    y = _t(name + ".conv.0", ConvFn.apply(x0, x1, seq[0].weight, seq[0].bias, False, True))
    y = _bn(seq[1], y, True, name + ".conv.1")
    y = _t(name + ".conv.3", ConvFn.apply(y, None, seq[3].weight, seq[3].bias, False, True))
    return _bn(seq[4], y, True, name + ".conv.4")


def _up_conv(seq, x, name=""):
    """up_conv (ref :23-34): nearest x2 -> conv3x3 -> BN -> ReLU (the upsample is fused in the conv gather)."""
    y = _t(name + ".up.1", ConvFn.apply(x, None, seq[1].weight, seq[1].bias, True, True))
    return _bn(seq[2], y, True, name + ".up.2")


def _gate(att, g, x, name="", x_scale=None):
    """Attention_block (ref :36-62).  x_scale: the alias of x the final x * psi reads (FanOutFn), x itself when None."""
    gp = _t(name + ".W_g.0", ConvFn.apply(g, None, att.W_g[0].weight, att.W_g[0].bias, False))
    xp = _t(name + ".W_x.0", ConvFn.apply(x, None, att.W_x[0].weight, att.W_x[0].bias, False))
    if _gate_mid_ok(gp, xp):
        bg, bx = att.W_g[1], att.W_x[1]
        p = GateMidFn.apply(gp, xp, bg.weight, bg.bias, bg.running_mean, bg.running_var, bg.eps, bg.momentum, bx.weight, bx.bias,
                            bx.running_mean, bx.running_var, bx.eps, bx.momentum, att.psi[0].weight, att.psi[0].bias)
        for mod in (bg, bx):
            if _NBT is not None:
                _NBT.append(mod.num_batches_tracked)
            else:
                with torch.no_grad():
                    mod.num_batches_tracked += 1
    else:
        g1 = _bn(att.W_g[1], gp, False, name + ".W_g.1")
        x1 = _bn(att.W_x[1], xp, False, name + ".W_x.1")
        q = _t(name + ".q", AddReluFn.apply(g1, x1))
        p = _t(name + ".psi.0", PsiConvFn.apply(q, att.psi[0].weight, att.psi[0].bias))
    psi = _t(name + ".psi", SigmoidFn.apply(_bn(att.psi[1], p, False, name + ".psi.1")))
    return _t(name + ".out", RowScaleFn.apply(x if x_scale is None else x_scale, psi))


def forward_train(net, x):
    """NBP.forward in train mode (ref :110-160) -> (out1 [B,8,S/4,S/4], out2 [B,1,S,S]) with autograd."""
    L = _lib.lib()
    B, _, S, _ = x.shape
    dev = x.device
    _reset_arena(dev)
    global _NBT, _PRE
    _NBT = []
    _PRE = _prepack_run(net, dev) if _observer is None else None
    try:
        return _forward_train(net, x, L, B, S, dev)
    finally:
        if _NBT:
            with torch.no_grad():
                torch._foreach_add_(_NBT, 1)       # nn.BatchNorm2d's num_batches_tracked += 1 (46 one-element kernels otherwise)
        _NBT = None
        _PRE = None                                # (each layer's context holds its planes for the backward)


def _forward_train(net, x, L, B, S, dev):
    seq = net.Conv1.conv
    if (_FIRST_CONV and S % 32 == 0 and tuple(seq[0].weight.shape) == (64, 5, 3, 3) and B * S * S * 64 * 4 < 2 ** 31):
        # the first layer reads the NCHW input itself; the rest of the block as _block
        y = _t("Conv1.conv.0", FirstConvFn.apply(x.contiguous().float(), seq[0].weight, seq[0].bias))
        y = _bn(seq[1], y, True, "Conv1.conv.1")
        y = _t("Conv1.conv.3", ConvFn.apply(y, None, seq[3].weight, seq[3].bias, False, True))
        x1 = _bn(seq[4], y, True, "Conv1.conv.4")
    else:
        xh = torch.empty(B, S, S, 5, dtype=torch.float32, device=dev)
        _chk(L.nbp_nchw_to_nhwc_f32(_lib.ptr(x.contiguous().float()), B, 5, S, S, _lib.ptr(xh), _st()), "to_nhwc")
        x0 = _pad_channels(xh, 64)
        x1 = _block(net.Conv1.conv, x0, None, "Conv1")
    # every tensor with several consumers goes through _fan: its gradient is one n-ary sum (the skips x1 .. x4 feed the max-pool and,
    # per decoder that reaches their level, a gate's W_x convolution and its x * psi; x5 both decoders; an up_conv output the gate's
    # W_g convolution and the block behind the gate)
    f1 = _fan(x1, 3)
    x2 = _block(net.Conv2.conv, _t("pool1", MaxPoolFn.apply(f1[0])), None, "Conv2")
    f2 = _fan(x2, 3)
    x3 = _block(net.Conv3.conv, _t("pool2", MaxPoolFn.apply(f2[0])), None, "Conv3")
    f3 = _fan(x3, 5)
    x4 = _block(net.Conv4.conv, _t("pool3", MaxPoolFn.apply(f3[0])), None, "Conv4")
    f4 = _fan(x4, 5)
    x5 = _block(net.Conv5.conv, _t("pool4", MaxPoolFn.apply(f4[0])), None, "Conv5")
    f5 = _fan(x5, 2)
    # (level, decoder) -> the two aliases of the skip that decoder's gate reads
    skips = {(5, 1): f4[1:3], (5, 2): f4[3:5], (4, 1): f3[1:3], (4, 2): f3[3:5], (3, 2): f2[1:3], (2, 2): f1[1:3]}
    outs = {}
    for d, levels in ((1, (5, 4)), (2, (5, 4, 3, 2))):
        cur = f5[d - 1]
        for Lv in levels:
            dd = _fan(_up_conv(getattr(net, f"Up{Lv}_{d}").up, cur, f"Up{Lv}_{d}"), 2)
            sk = skips[(Lv, d)]
            a = _gate(getattr(net, f"Att{Lv}_{d}"), dd[0], sk[0], f"Att{Lv}_{d}", x_scale=sk[1])
            cur = _block(getattr(net, f"Up_conv{Lv}_{d}").conv, a, dd[1], f"Up_conv{Lv}_{d}")
        outs[d] = cur
    o1 = ConvFn.apply(outs[1], None, net.Final1.weight, net.Final1.bias, False)          # [B,S/4,S/4,8]
    out1 = ToNCHWFn.apply(o1)
    # Final2 is a 64 -> 1 convolution at full resolution: a row dot product (as the gates' psi layer), not a GEMM padded to 64
    # output channels whose 63 zero columns are written, sliced away and padded back in for the backward
    o2 = PsiConvFn.apply(outs[2], net.Final2[0].weight, net.Final2[0].bias)              # [B,S,S,1]
    out2 = SigmoidFn.apply(o2).reshape(B, 1, S, S)                                       # C == 1: NHWC == NCHW
    return out1, out2


def gather_values(out1, batch_indices, coords):
    """pred_values = predicted_value_map[batch_indices, c, x, y] (nbp_utils.py:379)."""
    full = torch.cat([batch_indices.view(-1, 1).long(), coords.long()], 1)
    return GatherValuesFn.apply(out1, full)


def loss(net, pred1, target1, pred2, target2):
    """NBP.loss (ref :162-173) with the MSE / BCE reductions and their gradients on the device kernels."""
    s = net.log_vars
    mse = MeanLossFn.apply(pred1, target1, 0)
    bce = MeanLossFn.apply(pred2, target2, 1)
    return mse / (2.0 * torch.exp(2 * s[0])) + s[0] + bce / torch.exp(2 * s[1]) + s[1]
