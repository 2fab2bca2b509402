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

    def __init__(self, handle, buf, keep, bf16=False, precision=None):
        self.handle, self.buf, self.keep = handle, buf, keep
        self.precision = precision or ("bf16" if bf16 else "fp32")

    @property
    def bf16(self):
        return self.precision == "bf16"

    def free(self):
        if self.handle:
            _lib.lib().nbp_free_weights(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


PRECISIONS = ("fp32", "fp32_split", "bf16")
_PACK = {"fp32": ("nbp_pack_weights", "nbp_packed_weights_bytes"), "bf16": ("nbp_pack_weights_bf16", "nbp_packed_weights_bytes_bf16"),
         "fp32_split": ("nbp_pack_weights_split", "nbp_packed_weights_bytes_split")}
_FWD = {"fp32": ("nbp_forward_f32", "nbp_forward_workspace_bytes"), "bf16": ("nbp_forward_bf16", "nbp_forward_workspace_bytes_bf16"),
        "fp32_split": ("nbp_forward_split_f32", "nbp_forward_workspace_bytes_split")}


def pack_state_dict(sd, device, bf16: bool = False, precision: str = None) -> PackedWeights:
    """precision: "fp32" (fp32 MFMA), "fp32_split" (fp32 tensors, 3x3 layers as three exact fp16 MFMAs per product on
    two-piece operands -- the same accuracy at 5.3x the matrix rate; its handle also serves "fp32"), "bf16" (bf16 conv weights / activations, fp32
    accumulate and epilogues; nbp_forward_bf16).  bf16=True is the older spelling of precision="bf16"."""
    precision = precision or ("bf16" if bf16 else "fp32")
    assert precision in PRECISIONS, precision
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
    nbytes = getattr(L, _PACK[precision][1])()
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    arr = lambda v: (C.c_void_p * 48)(*v)
    handle = C.c_void_p()
    with torch.cuda.device(device):
        fn = getattr(L, _PACK[precision][0])
        rc = fn(arr(ws), arr(ss), arr(ts), buf.data_ptr(), nbytes, _lib.current_stream(), C.byref(handle))
        _lib.check(rc, "nbp_pack_weights")
        torch.cuda.current_stream().synchronize()   # sources in `keep` may now be released
    return PackedWeights(handle, buf, None, precision=precision)


def pack_eval_weights(module, device) -> PackedWeights:
    return pack_state_dict(module.state_dict(), device, precision=getattr(module, "conv_precision", "fp32"))


_ws_cache = {}


def _workspace(B, S, device, precision="fp32"):
    # one workspace per (shape, stream): forwards enqueued on different streams may run concurrently
    key = (B, S, device.index if isinstance(device, torch.device) else str(device), precision, _lib.current_stream())
    ws = _ws_cache.get(key)
    if ws is None:
        L = _lib.lib()
        n = getattr(L, _FWD[precision][1])(B, S)
        if n == 0:
            raise _lib.NbpHipError(f"unsupported NBP input size B={B} S={S}")
        ws = torch.empty(n, dtype=torch.uint8, device=device)
        if len(_ws_cache) >= 8:    # keep a few workspaces alive (sizes rarely change); an evicted one may still be
            torch.cuda.synchronize(device)      # in use by a forward in flight on another stream
            _ws_cache.pop(next(iter(_ws_cache)))
        _ws_cache[key] = ws
    return ws


def forward_packed(packed: PackedWeights, x: torch.Tensor, precision: str = None, out=None, ws=None):
    """precision overrides the handle's own only where the handle allows it ("fp32" on a "fp32_split" handle).
    out = (out1, out2) / ws: caller-owned result tensors and workspace (ForwardGraph: everything a captured launch touches
    must outlive the graph)."""
    precision = precision or packed.precision
    if precision != packed.precision and not (precision == "fp32" and packed.precision == "fp32_split"):
        raise ValueError(f"weights packed for {packed.precision!r} cannot run the {precision!r} forward")
    B, _, S, _ = x.shape
    x = x.contiguous().float()
    if out is None:
        out1 = torch.empty(B, 8, S // 4, S // 4, dtype=torch.float32, device=x.device)
        out2 = torch.empty(B, 1, S, S, dtype=torch.float32, device=x.device)
    else:
        out1, out2 = out
    fn = getattr(_lib.lib(), _FWD[precision][0])
    if torch.cuda.current_device() == x.device.index:       # the common case: no device-guard objects in the step loop
        if ws is None:
            ws = _workspace(B, S, x.device, precision)      # (keyed by the current stream: looked up under the right device)
        rc = fn(packed.handle, x.data_ptr(), B, S, out1.data_ptr(), out2.data_ptr(), ws.data_ptr(), ws.numel(),
                _lib.current_stream())
    else:
        with torch.cuda.device(x.device):
            if ws is None:
                ws = _workspace(B, S, x.device, precision)
            rc = fn(packed.handle, x.data_ptr(), B, S, out1.data_ptr(), out2.data_ptr(), ws.data_ptr(), ws.numel(),
                    _lib.current_stream())
    _lib.check(rc, _FWD[precision][0])
    return out1, out2


class ForwardGraph:
    """The eval forward on FIXED buffers, captured once into a hipGraph and replayed.

    The reference's own usage is one forward per exploration step on one map (nbp_planning.py:166): ~100 kernel launches of a
    few microseconds each, 0.8 ms of host enqueue time per forward against 0.03 ms for a replay (the GPU time is the same:
    profiles/r04/fwd_graph_ab.txt) -- the single-rollout loop is host-bound without it.  A rollout's network input is a persistent
    tensor (RolloutState.net_in), so the whole launch sequence is replayable: same kernels, same arguments, same order -- the
    outputs are bit-identical to the eager call's (tests/test_gpu_network.py::test_forward_graph_is_bit_identical).
    `x` must stay alive and keep its address; `out1` / `out2` are overwritten by every replay (consume them on the replaying
    stream before the next one)."""

    def __init__(self, packed: PackedWeights, x: torch.Tensor, precision: str = None):
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
        precision = precision or packed.precision
        B, _, S, _ = x.shape
        L = _lib.lib()
        self.packed, self.x, self.precision = packed, x, precision
        with torch.cuda.device(x.device):
            n = getattr(L, _FWD[precision][1])(B, S)
            if n == 0:
                raise _lib.NbpHipError(f"unsupported NBP input size B={B} S={S}")
            self.ws = torch.empty(n, dtype=torch.uint8, device=x.device)
            self.out1 = torch.empty(B, 8, S // 4, S // 4, dtype=torch.float32, device=x.device)
            self.out2 = torch.empty(B, 1, S, S, dtype=torch.float32, device=x.device)
            cur = torch.cuda.current_stream(x.device)
            side = torch.cuda.Stream(x.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):           # warm-up outside the capture: one-time function attributes, lazy module loads
                forward_packed(packed, x, precision, (self.out1, self.out2), self.ws)
            cur.wait_stream(side)
            torch.cuda.synchronize(x.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                forward_packed(packed, x, precision, (self.out1, self.out2), self.ws)

    def __call__(self):
        self.graph.replay()
        return self.out1, self.out2


def forward_eval(module, x: torch.Tensor):
    packed = module._ensure_packed(x.device)
    return forward_packed(packed, x)
