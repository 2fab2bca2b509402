"""TEST INFRASTRUCTURE ONLY.  numpy restatement (float32, same operation order) of nbp_slice_obstacle_f32, the GT
obstacle label that replaces get_binary_obstacle_array (next_best_path/utility/utils.py:226-262).

Parity unpinned: the reference renders the mesh / plane intersection through trimesh 4.1.2 (absent),
matplotlib and PIL (anti-aliased 1.5 pt lines, PNG round trip, LANCZOS resize, threshold 128); this restatement
fixes the definition "pixel centre within 1.04 px of the intersection segment, window [lo,hi] around the camera,
column ~ -(x - cx), row ~ -(z - cz)" and is checked by analytic cases in tests/test_oracle_planner_sim.py."""
from __future__ import annotations

import numpy as np

f32 = np.float32


def slice_obstacle(verts, faces, y0, cx, cz, S=256, lo=-40.0, hi=40.0, half_width=1.04):
    verts = np.asarray(verts, f32)
    faces = np.asarray(faces, np.int64)
    out = np.zeros((S, S), f32)
    y0, cx, cz, hi_f, hw = f32(y0), f32(cx), f32(cz), f32(hi), f32(half_width)
    scale = f32(float(S) / (float(hi) - float(lo)))
    pad = hw + f32(1.0)
    h2 = hw * hw
    for f in range(len(faces)):
        p = verts[faces[f]]
        d = (p[:, 1] - y0).astype(f32)
        seg = []
        for k in range(3):
            q = (k + 1) % 3
            if (d[k] < 0) != (d[q] < 0):
                t = f32(d[k] / f32(d[k] - d[q]))
                x = f32(p[k, 0] + f32(t * f32(p[q, 0] - p[k, 0])))
                z = f32(p[k, 2] + f32(t * f32(p[q, 2] - p[k, 2])))
                seg.append((f32(f32(f32(cx - x) + hi_f) * scale), f32(f32(f32(cz - z) + hi_f) * scale)))
        if len(seg) != 2:
            continue
        (u0, v0), (u1, v1) = seg
        wu, wv = f32(u1 - u0), f32(v1 - v0)
        L2 = f32(f32(wu * wu) + f32(wv * wv))
        c0 = max(0, int(np.floor(f32(min(u0, u1) - pad)))); c1 = min(S - 1, int(np.ceil(f32(max(u0, u1) + pad))))
        r0 = max(0, int(np.floor(f32(min(v0, v1) - pad)))); r1 = min(S - 1, int(np.ceil(f32(max(v0, v1) + pad))))
        if c1 < c0 or r1 < r0:
            continue
        cc, rr = np.meshgrid(np.arange(c0, c1 + 1), np.arange(r0, r1 + 1))
        qu = ((cc.astype(f32) + f32(0.5)) - u0).astype(f32)
        qv = ((rr.astype(f32) + f32(0.5)) - v0).astype(f32)
        if L2 > 0:
            t = ((qu * wu).astype(f32) + (qv * wv).astype(f32)).astype(f32) / L2
            t = np.minimum(np.maximum(t.astype(f32), f32(0)), f32(1)).astype(f32)
        else:
            t = np.zeros_like(qu)
        du = (qu - (t * wu).astype(f32)).astype(f32)
        dv = (qv - (t * wv).astype(f32)).astype(f32)
        hit = ((du * du).astype(f32) + (dv * dv).astype(f32)).astype(f32) <= h2
        out[rr[hit], cc[hit]] = 1.0
    return out
