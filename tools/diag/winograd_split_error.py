"""Prototype (CPU, torch): would Winograd F(2x2, 3x3) on the split scheme meet the accuracy bar?  (kill if the
error against fp64 exceeds 2x the fp32 pipe's, or if the layer is not >= 1.3x faster.)

The numerics of the candidate kernel are emulated step by step on the host:
  input transform  V = B^T d B   in fp32 (additions only), per 4x4 patch (stride 2)
  filter transform U = G g G^T   in float64, rounded once
  both cut into two fp16 pieces under a per-tensor power-of-two scale (max -> [2^14, 2^15)), products hi hi + hi lo + lo hi
  summed over the channels per transformed position (exactly here: the accumulation rounding is the same K-long fp32 chain as the
  direct kernel's and is left out on both sides), output transform Y = A^T M A in fp32.
Compared with the direct split scheme (same representation, no transforms) and with plain fp32 operands, all against float64, on
(a) uniform activations, (b) rollout-like activations: non-negative, 1e4 on 'wall' pixels beside O(1) elsewhere."""
import sys
import torch

torch.manual_seed(0)
EXP = 14


def split(x):
    """fp32 tensor -> (hi + lo) as float64 under the per-tensor scale (what the MFMAs multiply), lo.lo dropped later"""
    m = x.abs().max().item()
    e = torch.floor(torch.log2(torch.tensor(m))).item() if m > 0 else EXP
    s = 2.0 ** (EXP - e)
    xs = x.float() * s
    hi = xs.half()
    lo = (xs - hi.float()).half()
    return hi.double(), lo.double(), s


def split_matmul(a, b):
    """sum_k a[..., k] b[k, ...] with both operands split: hi hi + hi lo + lo hi, exact accumulation, one fp32 rounding"""
    ah, al, sa = split(a)
    bh, bl, sb = split(b)
    return (((ah @ bh) + (ah @ bl) + (al @ bh)) / (sa * sb)).float()


def run(name, x, w):
    B, C, H, W = x.shape
    N = w.shape[0]
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    # ---- direct: im2col rows x [9 C, N]
    cols = torch.nn.functional.unfold(x, 3, padding=1).transpose(1, 2).reshape(-1, 9 * C)          # fp32
    wm = w.reshape(N, 9 * C).t().contiguous()
    d_f32 = (cols.double() @ wm.double()).float()          # fp32 operands, exact accumulation
    d_split = split_matmul(cols, wm)
    shape = lambda t: t.reshape(B, H * W, N).transpose(1, 2).reshape(B, N, H, W).double()
    # ---- Winograd F(2x2, 3x3)
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    patches = xp.unfold(2, 4, 2).unfold(3, 4, 2)                         # [B, C, H/2, W/2, 4, 4]
    V = torch.einsum("ij,bchwjk->bchwik", Bt, patches)                   # fp32 additions, rows then columns
    V = torch.einsum("bchwik,lk->bchwil", V, Bt)
    U = torch.einsum("ij,ncjk,lk->ncil", G, w.double(), G).float()       # [N, C, 4, 4], rounded once
    T = (H // 2) * (W // 2) * B
    M = torch.empty(4, 4, T, N)
    M32 = torch.empty(4, 4, T, N)
    for i in range(4):
        for j in range(4):
            a = V[..., i, j].permute(0, 2, 3, 1).reshape(T, C)
            b = U[..., i, j].t().contiguous()
            M[i, j] = split_matmul(a, b)
            M32[i, j] = (a.double() @ b.double()).float()
    def out_tf(Mx):
        Y = torch.einsum("ij,jktn->iktn", At, Mx)
        Y = torch.einsum("iktn,lk->iltn", Y, At)                          # [2, 2, T, N]
        return Y.reshape(2, 2, B, H // 2, W // 2, N).permute(2, 5, 3, 0, 4, 1).reshape(B, N, H, W).double()
    w_split, w_f32 = out_tf(M), out_tf(M32)
    rng = ref.abs().max().item()
    e = lambda t: ((t - ref).abs().mean().item() / rng, (t - ref).abs().max().item() / rng)
    r = {"direct fp32 operands": e(shape(d_f32)), "direct split": e(shape(d_split)), "winograd fp32 operands": e(w_f32), "winograd split": e(w_split)}
    print(f"{name}: C={C} N={N} {H}x{W}, range {rng:.3g}")
    for k, (m, mx) in r.items():
        print(f"   {k:24s} mean {m:.2e}  max {mx:.2e}   ({m / r['direct fp32 operands'][0]:.1f}x the direct fp32-operand mean)")
    return r


C, N, S = 64, 64, 64
w = (torch.rand(N, C, 3, 3) * 2 - 1) * (6.0 / (C * 9)) ** 0.5
run("uniform activations", torch.rand(1, C, S, S) * 2 - 1, w)
x = torch.rand(1, C, S, S)
walls = torch.rand(1, 1, S, S) < 0.05
x = torch.where(walls, x * 1e4, x)                      # rollout-like: wall pixels 1e4 beside O(1)
run("rollout-like activations", x, w)
