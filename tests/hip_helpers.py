"""ctypes call helpers for the GPU parity tests (thin: pointers + sizes only)."""
import torch

from nextbestpath_amd import _lib


def stream():
    return _lib.current_stream()


def pack_conv(w_oihw, scale=None, c_off=0, c_total=None, dst=None):
    L = _lib.lib()
    N, C, k, _ = w_oihw.shape
    c_total = c_total or C
    if dst is None:
        dst = torch.zeros(c_total // 32 * k * k * N * 32, dtype=torch.float32, device=w_oihw.device)
    rc = L.nbp_pack_conv_weight(_lib.ptr(w_oihw), N, C, k, _lib.ptr(scale), c_off, c_total, _lib.ptr(dst), stream())
    _lib.check(rc, "pack")
    return dst


def conv_igemm(src0, src1, ups, wpk, N, ksize, scale, shift, relu, split_k=0, tile=0):
    """src*: NHWC cuda tensors [B,Hs,Ws,C]; returns NHWC [B,H,W,N]."""
    L = _lib.lib()
    B, Hs, Ws, C0 = src0.shape
    H, W = (Hs * 2, Ws * 2) if ups else (Hs, Ws)
    C1 = 0 if src1 is None else src1.shape[3]
    out = torch.empty(B, H, W, N, dtype=torch.float32, device=src0.device)
    nws = L.nbp_conv_igemm_workspace_bytes(B, H, W, N, split_k)
    ws = torch.empty(max(nws, 256), dtype=torch.uint8, device=src0.device)
    rc = L.nbp_conv_igemm_f32(_lib.ptr(src0), C0, _lib.ptr(src1), C1, int(ups), B, H, W, ksize, _lib.ptr(wpk), N,
                              _lib.ptr(scale), _lib.ptr(shift), int(relu), _lib.ptr(out), split_k, tile,
                              _lib.ptr(ws), ws.numel(), stream())
    _lib.check(rc, "conv_igemm")
    return out


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


# ---- bf16 path (uint16 storage on the device; torch.bfloat16 views share the bit pattern)
def to_bf16_dev(x_f32_cuda):
    return x_f32_cuda.to(torch.bfloat16).contiguous()


def pack_conv_bf16(w_oihw, scale=None, c_off=0, c_total=None, dst=None):
    L = _lib.lib()
    N, C, k, _ = w_oihw.shape
    c_total = c_total or C
    if dst is None:
        dst = torch.zeros(c_total // 64 * k * k * N * 64, dtype=torch.bfloat16, device=w_oihw.device)
    rc = L.nbp_pack_conv_weight_bf16(_lib.ptr(w_oihw), N, C, k, _lib.ptr(scale), c_off, c_total, _lib.ptr(dst), stream())
    _lib.check(rc, "pack_bf16")
    return dst


def conv_igemm_bf16(src0, src1, ups, wpk, N, ksize, scale, shift, relu, split_k=0, tile=0):
    """src*: NHWC cuda bfloat16 tensors [B,Hs,Ws,C]; returns NHWC bfloat16 [B,H,W,N]."""
    L = _lib.lib()
    B, Hs, Ws, C0 = src0.shape
    H, W = (Hs * 2, Ws * 2) if ups else (Hs, Ws)
    C1 = 0 if src1 is None else src1.shape[3]
    out = torch.empty(B, H, W, N, dtype=torch.bfloat16, device=src0.device)
    nws = L.nbp_conv_igemm_bf16_workspace_bytes(B, H, W, N, 64)
    ws = torch.empty(max(nws, 256), dtype=torch.uint8, device=src0.device)
    rc = L.nbp_conv_igemm_bf16(_lib.ptr(src0), C0, _lib.ptr(src1), C1, int(ups), B, H, W, ksize, _lib.ptr(wpk), N,
                               _lib.ptr(scale), _lib.ptr(shift), int(relu), _lib.ptr(out), split_k, tile,
                               _lib.ptr(ws), ws.numel(), stream())
    _lib.check(rc, "conv_igemm_bf16")
    return out


# ---- split path (fp32 tensors; 3x3 layers on the fp16 matrix pipe through two-piece operand splitting)
def pack_conv_split(w_oihw, scale=None, c_total=None):
    """-> (planes int16 [c_total/16 * 9 * 4 * N * 8], wamax int32[1] = float bits of max |w|)."""
    L = _lib.lib()
    N, C, k, _ = w_oihw.shape
    c_total = c_total or C
    dst = torch.zeros(c_total // 16 * k * k * 4 * N * 8, dtype=torch.int16, device=w_oihw.device)
    wamax = torch.zeros(1, dtype=torch.int32, device=w_oihw.device)
    rc = L.nbp_pack_conv_weight_split(_lib.ptr(w_oihw), N, C, k, _lib.ptr(scale), 0, c_total, _lib.ptr(dst), _lib.ptr(wamax), stream())
    _lib.check(rc, "pack_split")
    return dst, wamax


def conv3x3_split(src0, src1, ups, packed, N, scale, shift, relu, split_k=0, amax_in=None, amax_out=None):
    """src*: NHWC cuda fp32 tensors [B,Hs,Ws,C]; packed = pack_conv_split(...); returns NHWC fp32 [B,H,W,N]."""
    L = _lib.lib()
    planes, wamax = packed
    B, Hs, Ws, C0 = src0.shape
    H, W = (Hs * 2, Ws * 2) if ups else (Hs, Ws)
    C1 = 0 if src1 is None else src1.shape[3]
    out = torch.empty(B, H, W, N, dtype=torch.float32, device=src0.device)
    nws = L.nbp_conv_split_workspace_bytes(B, H, W, N, split_k)
    ws = torch.empty(max(nws, 256), dtype=torch.uint8, device=src0.device)
    rc = L.nbp_conv3x3_split_f32(_lib.ptr(src0), C0, _lib.ptr(src1), C1, int(ups), B, H, W, _lib.ptr(planes), _lib.ptr(wamax), N,
                                 _lib.ptr(scale), _lib.ptr(shift), int(relu), _lib.ptr(out), _lib.ptr(amax_in), _lib.ptr(amax_out),
                                 split_k, _lib.ptr(ws), ws.numel(), stream())
    _lib.check(rc, "conv3x3_split")
    return out


def pack_upconv_split(w_oihw):
    """-> (parity planes int16 [4 * C/16 * 4 * 4 * N * 8], wamax int32[1])."""
    L = _lib.lib()
    N, C, _, _ = w_oihw.shape
    dst = torch.zeros(4 * (C // 16) * 4 * 4 * N * 8, dtype=torch.int16, device=w_oihw.device)
    wamax = torch.zeros(1, dtype=torch.int32, device=w_oihw.device)
    _lib.check(L.nbp_pack_upconv_weight_split(_lib.ptr(w_oihw), N, C, _lib.ptr(dst), _lib.ptr(wamax), stream()), "pack_upconv")
    return dst, wamax


def upconv3x3_split(src, packed, N, scale, shift, relu, split_k=0):
    """src NHWC [B,H/2,W/2,C] -> NHWC [B,H,W,N] (x2 nearest upsample + 3x3 convolution through the parity kernels)."""
    L = _lib.lib()
    planes, wamax = packed
    B, Hs, Ws, C = src.shape
    H, W = 2 * Hs, 2 * Ws
    out = torch.empty(B, H, W, N, dtype=torch.float32, device=src.device)
    ws = torch.empty(max(L.nbp_conv_split_workspace_bytes(B, H, W, N, split_k), 256), dtype=torch.uint8, device=src.device)
    rc = L.nbp_upconv3x3_split_f32(_lib.ptr(src), C, B, H, W, _lib.ptr(planes), _lib.ptr(wamax), N, _lib.ptr(scale), _lib.ptr(shift),
                                   int(relu), _lib.ptr(out), None, None, split_k, _lib.ptr(ws), ws.numel(), stream())
    _lib.check(rc, "upconv3x3_split")
    return out
