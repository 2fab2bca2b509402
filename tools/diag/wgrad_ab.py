"""Diagnostic (GPU box): nbp_conv_wgrad_split_f32 against nbp_conv_wgrad_f32 at the training step's real layer shapes."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nextbestpath_amd import _lib
L = _lib.lib()
shapes = [(32, 256, 64, 0, 64, 0, 64, 64), (32, 256, 64, 0, 64, 0, 5, 64), (32, 128, 64, 0, 128, 0, 64, 128), (32, 64, 128, 0, 256, 0, 128, 256),
          (32, 32, 256, 0, 512, 0, 256, 512), (32, 32, 512, 512, 512, 0, 1024, 512), (32, 64, 512, 0, 256, 1, 512, 256),
          (32, 256, 128, 0, 64, 1, 128, 64), (32, 256, 64, 64, 64, 0, 128, 64), (32, 256, 64, 0, 64, 0, 64, 1)]
for B, H, C0, C1, N, ups, c_real, n_real in shapes:
    torch.manual_seed(1)
    Hs = H // 2 if ups else H
    x0 = torch.randn(B, Hs, Hs, C0, device="cuda").relu_()
    x1 = torch.randn(B, Hs, Hs, C1, device="cuda") if C1 else None
    dy = torch.randn(B, H, H, N, device="cuda") * 1e-4
    out = []
    for entry in ("nbp_conv_wgrad_f32", "nbp_conv_wgrad_split_f32"):
        dw = torch.zeros(n_real, c_real, 3, 3, device="cuda")
        ws = torch.empty(L.nbp_conv_wgrad_workspace_bytes(B, H, H, C0, C1, N, 3), dtype=torch.uint8, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k in range(2):
            e0.record()
            extra = (None, None, None) if "split" in entry else ()
            rc = getattr(L, entry)(_lib.ptr(x0), C0, _lib.ptr(x1), C1, ups, B, H, H, 3, _lib.ptr(dy), N, c_real, n_real,
                                   _lib.ptr(dw), *extra, _lib.ptr(ws), ws.numel(), _lib.current_stream())
            e1.record()
        torch.cuda.synchronize()
        assert rc == 0, (entry, rc)
        out.append((dw.clone(), e0.elapsed_time(e1)))
    a, b = out[0][0].double(), out[1][0].double()
    print(f"B={B} H={H} C0={C0} C1={C1} N={N} ups={ups} c_real={c_real} n_real={n_real}: fp32 {out[0][1]:.3f} ms, split {out[1][1]:.3f} ms; "
          f"max |diff| / max |ref| = {(a - b).abs().max().item() / a.abs().max().item():.2e}; finite {bool(torch.isfinite(b).all())}")
