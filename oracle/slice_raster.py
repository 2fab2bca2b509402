"""TEST INFRASTRUCTURE ONLY.  numpy restatement (float32, same operation order) of nbp_slice_obstacle_f32, the GT
obstacle label that replaces get_binary_obstacle_array (next_best_path/utility/utils.py:226-262).

Two forms.  slice_obstacle: "pixel centre within 1.04 px of the intersection segment, window [lo,hi] around the camera,
column ~ -(x - cx), row ~ -(z - cz)" (rounds 1-5; analytic cases in tests/test_oracle_planner_sim.py).  slice_obstacle_fig
(round 6): the same strokes on the REFERENCE's pixel grid -- its matplotlib figure's axes box saved, resized and flipped:
geometry() below -- as rectangles with projecting caps.  Pinned by tests/golden/obstacle_label.npz, labels produced by the
reference's own draw -> PNG -> resize -> threshold code (utils.py:232-258; matplotlib and PIL are installed) from THIS
restatement's mesh / plane segments -- trimesh 4.1.2 (absent) is the one stub.  What remains unpinned is the anti-aliasing of
Agg and PIL's Lanczos filter: agreement is "within one pixel of line position", not pixel for pixel."""
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


def geometry(S=256, view_size=80.0):
    """nextbestpath_amd/utility/hipops.py::reference_figure_geometry, restated: (half_u, scale_u, half_v, scale_v, half_width, cap)."""
    fig = 2.56 * 100.0
    ax_w, ax_h = 0.775 * fig, 0.77 * fig
    png_w, png_h = int(ax_w), int(ax_h)
    ppu = ax_w / view_size
    su, sv = ppu * S / png_w, ppu * S / png_h
    cu = S - (ax_w / 2.0) * S / png_w
    cv = (png_h - ax_h / 2.0) * S / png_h
    hw = 0.5 * (1.5 * 100.0 / 72.0) * 0.5 * (S / png_w + S / png_h)
    return cu / su, su, cv / sv, sv, hw, hw


def plane_segments(verts, faces, y0):
    """The mesh / plane y = y0 intersection as [n,2,3] float32 segments, with the kernel's arithmetic (what trimesh.intersections.
    mesh_plane returns in the reference, utils.py:230; used by tests/golden/make_golden.py to feed the reference's drawing stage)."""
    verts = np.asarray(verts, f32)
    faces = np.asarray(faces, np.int64)
    y0 = f32(y0)
    segs = []
    for f in range(len(faces)):
        p = verts[faces[f]]
        d = (p[:, 1] - y0).astype(f32)
        s = []
        for k in range(3):
            q = (k + 1) % 3
            if (d[k] < 0) != (d[q] < 0):
                t = f32(d[k] / f32(d[k] - d[q]))
                s.append([f32(p[k, 0] + f32(t * f32(p[q, 0] - p[k, 0]))), y0, f32(p[k, 2] + f32(t * f32(p[q, 2] - p[k, 2])))])
        if len(s) == 2:
            segs.append(s)
    return np.asarray(segs, f32).reshape(-1, 2, 3)


def slice_obstacle_fig(verts, faces, y0, cx, cz, S=256, view_size=80.0):
    """float32 restatement of nbp_slice_obstacle_fig_f32 (slice_obstacle_kernel with cap > 0), same operation order."""
    hu, su, hv, sv, hw, cap = (f32(v) for v in geometry(S, view_size))
    cx, cz = f32(cx), f32(cz)
    out = np.zeros((S, S), f32)
    pad = f32(max(hw, cap) + f32(1.0))
    for a, b in plane_segments(verts, faces, y0):
        u0, v0 = f32(f32(f32(cx - a[0]) + hu) * su), f32(f32(f32(cz - a[2]) + hv) * sv)
        u1, v1 = f32(f32(f32(cx - b[0]) + hu) * su), f32(f32(f32(cz - b[2]) + hv) * sv)
        wu, wv = f32(u1 - u0), f32(v1 - v0)
        L2 = f32(f32(wu * wu) + f32(wv * wv))
        c0 = max(0, int(np.floor(f32(min(u0, u1) - pad)))); c1 = min(S - 1, int(np.ceil(f32(max(u0, u1) + pad))))
        r0 = max(0, int(np.floor(f32(min(v0, v1) - pad)))); r1 = min(S - 1, int(np.ceil(f32(max(v0, v1) + pad))))
        if c1 < c0 or r1 < r0 or not L2 > 0:
            continue
        L = np.sqrt(L2).astype(f32)
        cc, rr = np.meshgrid(np.arange(c0, c1 + 1), np.arange(r0, r1 + 1))
        qu = ((cc.astype(f32) + f32(0.5)) - u0).astype(f32)
        qv = ((rr.astype(f32) + f32(0.5)) - v0).astype(f32)
        ta = (((qu * wu).astype(f32) + (qv * wv).astype(f32)).astype(f32) / L).astype(f32)
        tn = (np.abs(((qv * wu).astype(f32) - (qu * wv).astype(f32)).astype(f32)) / L).astype(f32)
        hit = (ta >= -cap) & (ta <= f32(L + cap)) & (tn <= hw)
        out[rr[hit], cc[hit]] = 1.0
    return out
