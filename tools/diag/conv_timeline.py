"""Diagnostic (GPU box, library built with NBP_EXTRA_FLAGS=-DNBP_DBG_TS): phase durations of one wave of the split conv kernel
from in-kernel cycle-counter stamps (1 stage start, 2 before the DMA wait, 3 after it, 4 after the stage barrier, 5 after the
halo staging, 6 after its barrier, 7 kernel end).   python tools/diag/conv_timeline.py [C] [N] [H] [B]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hip_helpers import conv3x3_split, pack_conv_split
from nextbestpath_amd import _lib
Cc, N, H, B = (int(v) for v in (sys.argv[1:5] + ["512", "256", "64", "8"][len(sys.argv) - 1:]))
x = torch.randn(B, H, H, Cc, device="cuda")
w = torch.randn(N, Cc, 3, 3, device="cuda") * 0.05
packed = pack_conv_split(w)
sc, sh = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
for _ in range(3):
    y = conv3x3_split(x, None, 0, packed, N, sc, sh, 1, split_k=1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y = conv3x3_split(x, None, 0, packed, N, sc, sh, 1, split_k=1); e1.record(); torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * 8192)(); n = C.c_uint(0)
L.nbp_dbg_read(buf, C.byref(n))
a = np.array(buf[:2 * min(n.value, 4000)], dtype=np.uint64).reshape(-1, 2)
tags, ts = a[:, 0].astype(int), a[:, 1].astype(np.int64)
w0, w1 = ts[tags == 100][0], ts[tags == 103][0]
c0, c1 = ts[tags == 101][0], ts[tags == 102][0]
print(f"calibration: wave alive {(w1 - w0) / 100.0:.1f} us (100 MHz wall clock), {c1 - c0} cycle-counter ticks -> {(c1 - c0) / ((w1 - w0) / 100.0) / 1e3:.3f} ticks per ns")
keep = tags < 100
tags, ts = tags[keep], ts[keep]
print(f"C={Cc} N={N} H={H} B={B}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {len(tags)} stamps, span {(ts[-1]-ts[0])} ticks")
d = np.diff(ts)
names = {(1, 2): "reads + MFMA issue", (2, 3): "wait weight DMA", (3, 4): "stage barrier", (4, 1): "loop back", (4, 5): "halo split + ds_write",
         (5, 6): "staging barrier", (6, 1): "loop back (chunk)", (4, 7): "epilogue"}
acc = {}
for k in range(len(d)):
    key = (tags[k], tags[k + 1])
    acc.setdefault(key, []).append(d[k])
tot = float(ts[-1] - ts[0])
for key, v in sorted(acc.items()):
    v = np.array(v, float)
    print(f"  {names.get(key, str(key)):26s} n={len(v):4d} mean {v.mean():9.1f} median {np.median(v):9.1f} max {v.max():9.1f} ticks  = {v.sum()/tot*100:5.1f} % of the wave's time")
