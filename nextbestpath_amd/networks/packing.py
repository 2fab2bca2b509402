"""Eval-mode weight packing and the forward call into libnbp_hip.so.

BatchNorm (eval, running statistics) and the conv bias collapse to one per-output-channel
affine applied in the conv epilogue:  out = act(acc * scale + shift) with
    scale = gamma / sqrt(var + eps),   shift = (bias - mean) * scale + beta
computed here in float64 and rounded once to fp32 (reference semantics:
next_best_path/networks/nbp_model.py:8-62 with nn.BatchNorm2d defaults eps=1e-5).
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib

# canonical conv order of include/nbp_hip.h -> (conv prefix, bn prefix or None)
def canonical_layers():
    out = []
    for e in range(1, 6):
        out.append((f"Conv{e}.conv.0", f"Conv{e}.conv.1"))
        out.append((f"Conv{e}.conv.3", f"Conv{e}.conv.4"))
    for d, levels in ((1, (5, 4)), (2, (5, 4, 3, 2))):
        for L in levels:
            out.append((f"Up{L}_{d}.up.1", f"Up{L}_{d}.up.2"))
            out.append((f"Att{L}_{d}.W_g.0", f"Att{L}_{d}.W_g.1"))
            out.append((f"Att{L}_{d}.W_x.0", f"Att{L}_{d}.W_x.1"))
            out.append((f"Att{L}_{d}.psi.0", f"Att{L}_{d}.psi.1"))
            out.append((f"Up_conv{L}_{d}.conv.0", f"Up_conv{L}_{d}.conv.1"))
            out.append((f"Up_conv{L}_{d}.conv.3", f"Up_conv{L}_{d}.conv.4"))
    out.append(("Final1", None))
    out.append(("Final2.0", None))
    assert len(out) == 48
    return out


def fold_affine(sd, conv, bn, eps=1e-5):
    """(scale, shift) float64 tensors for conv `conv` followed by BatchNorm `bn` (or none)."""
    bias = sd[conv + ".bias"].double()
    if bn is None:
        return torch.ones_like(bias), bias
    g, b = sd[bn + ".weight"].double(), sd[bn + ".bias"].double()
    mu, var = sd[bn + ".running_mean"].double(), sd[bn + ".running_var"].double()
    scale = g / torch.sqrt(var + eps)
    return scale, (bias - mu) * scale + b


class PackedWeights:
    """Owns the device buffer and the C handle; freed explicitly or on GC."""

    def __init__(self, handle, buf, keep, bf16=False):
        self.handle, self.buf, self.keep, self.bf16 = handle, buf, keep, bf16

    def free(self):
        if self.handle:
            _lib.lib().nbp_free_weights(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def pack_state_dict(sd, device, bf16: bool = False) -> PackedWeights:
    """bf16=True packs for nbp_forward_bf16 (bf16 conv weights / activations, fp32 accumulate and epilogues)."""
    L = _lib.lib()
    layers = canonical_layers()
    ws, ss, ts, keep = [], [], [], []
    folded = [fold_affine(sd, c, b) for c, b in layers]
    for i, (conv, bn) in enumerate(layers):
        w = sd[conv + ".weight"].detach().to(device=device, dtype=torch.float32).contiguous()
        scale, shift = folded[i]
        if ".W_g." in conv:       # fused attention GEMM: shift_g + shift_x rides on W_g
            shift = shift + folded[i + 1][1]
        s = scale.to(torch.float32).to(device).contiguous()
        t = shift.to(torch.float32).to(device).contiguous()
        keep += [w, s, t]
        ws.append(w.data_ptr()); ss.append(s.data_ptr()); ts.append(t.data_ptr())
    nbytes = L.nbp_packed_weights_bytes()
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    arr = lambda v: (C.c_void_p * 48)(*v)
    handle = C.c_void_p()
    with torch.cuda.device(device):
        fn = L.nbp_pack_weights_bf16 if bf16 else L.nbp_pack_weights
        rc = fn(arr(ws), arr(ss), arr(ts), buf.data_ptr(), nbytes, _lib.current_stream(), C.byref(handle))
        _lib.check(rc, "nbp_pack_weights")
        torch.cuda.current_stream().synchronize()   # sources in `keep` may now be released
    return PackedWeights(handle, buf, None, bf16)


def pack_eval_weights(module, device) -> PackedWeights:
    return pack_state_dict(module.state_dict(), device, bf16=getattr(module, "conv_precision", "fp32") == "bf16")


_ws_cache = {}


def _workspace(B, S, device, bf16=False):
    # one workspace per (shape, stream): forwards enqueued on different streams may run concurrently
    key = (B, S, str(device), bf16, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None:
        L = _lib.lib()
        n = L.nbp_forward_workspace_bytes_bf16(B, S) if bf16 else L.nbp_forward_workspace_bytes(B, S)
        if n == 0:
            raise _lib.NbpHipError(f"unsupported NBP input size B={B} S={S}")
        ws = torch.empty(n, dtype=torch.uint8, device=device)
        if len(_ws_cache) >= 8:    # keep a few workspaces alive (sizes rarely change); an evicted one may still be
            torch.cuda.synchronize(device)      # in use by a forward in flight on another stream
            _ws_cache.pop(next(iter(_ws_cache)))
        _ws_cache[key] = ws
    return ws


def forward_packed(packed: PackedWeights, x: torch.Tensor):
    B, _, S, _ = x.shape
    x = x.contiguous().float()
    out1 = torch.empty(B, 8, S // 4, S // 4, dtype=torch.float32, device=x.device)
    out2 = torch.empty(B, 1, S, S, dtype=torch.float32, device=x.device)
    ws = _workspace(B, S, x.device, packed.bf16)
    fn = _lib.lib().nbp_forward_bf16 if packed.bf16 else _lib.lib().nbp_forward_f32
    with torch.cuda.device(x.device):
        rc = fn(packed.handle, x.data_ptr(), B, S, out1.data_ptr(), out2.data_ptr(), ws.data_ptr(), ws.numel(),
                _lib.current_stream())
    _lib.check(rc, "nbp_forward_bf16" if packed.bf16 else "nbp_forward_f32")
    return out1, out2


def forward_eval(module, x: torch.Tensor):
    packed = module._ensure_packed(x.device)
    return forward_packed(packed, x)
