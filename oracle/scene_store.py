"""numpy restatement of the GT-surface sampler and the Scene / Cell point store (TEST INFRASTRUCTURE).

PINNED by tests/golden/scene.npz (outputs of the reference's own functions run on CPU tensors):
  * compute_mesh_face_area, sample_mesh_triangle, sample_points_on_mesh_faces
    (macarons/utility/utils.py:1301-1455), get_scene_gt_surface (macarons/utility/macarons_utils.py:612-637)
  * Scene.get_pts_in_bounding_box / get_cells_for_each_pt / fill_cells / return_entire_pt_cloud and Cell.__init__ /
    Cell.fill (macarons_utils.py:2952-3234); floor_divide (macarons/utility/utils.py:113-117)
The reference's random draws (torch.rand for the sampler, torch.randperm for the capacity cap) come from the global
generator; here the uniforms are an INPUT (so the fixture's recorded stream can be replayed) and the cap uses the
seeded index bijection of sampling.py (an exact-size subset, like randperm[:capacity])."""
import numpy as np

from . import sampling

f32 = np.float32


def face_areas(verts, faces):
    """compute_mesh_face_area: Heron's formula in the reference's factored form (fp32)."""
    fc = np.asarray(verts, f32)[np.asarray(faces)]
    def norm(d):
        return np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=f32)
    a, b, c = norm(fc[:, 0] - fc[:, 1]), norm(fc[:, 1] - fc[:, 2]), norm(fc[:, 2] - fc[:, 0])
    p = (a + b + c) / f32(2)
    if not np.any(p <= 0):
        res = ((p - a) / p) * ((p - b) / p) * ((p - c) / p)        # f1 = 1.
        res = np.sqrt(np.maximum(res, f32(0)), dtype=f32)
        res = res * (p * p)
    else:
        res = np.sqrt(np.maximum(p * (p - a) * (p - b) * (p - c), f32(0)), dtype=f32)
    return res.astype(f32)


def faces_inside(verts, faces, x_min, x_max):
    v = np.asarray(verts, f32)
    inside = np.all((v >= np.asarray(x_min, f32)) & (v <= np.asarray(x_max, f32)), axis=1)
    return inside, np.asarray(faces)[inside[np.asarray(faces)].all(1)]


def sample_surface(verts, faces, u_face, u_alpha, u_beta):
    """sample_mesh_triangle + sample_points_on_mesh_faces with the uniforms given: face = first index whose cumulative
    area probability is >= u; point = o + alpha a + beta b with (alpha, beta) reflected into the triangle."""
    v = np.asarray(verts, f32)
    fa = np.asarray(faces)
    area = face_areas(v, fa)
    prob = area / area.sum(dtype=f32)
    cum = np.cumsum(prob, dtype=f32)
    u = np.asarray(u_face, f32)
    pick = np.searchsorted(cum, u, side="left")          # cum - u >= 0  <=>  cum >= u; argmin of the non-negative gaps
    diff_all_neg = pick >= len(cum)                      # every gap negative -> all set to 2 -> argmin = 0
    pick = np.where(diff_all_neg, 0, pick)
    tri = v[fa[pick]]
    o, a, b = tri[:, 2], tri[:, 0] - tri[:, 2], tri[:, 1] - tri[:, 2]
    al, be = np.asarray(u_alpha, f32).copy(), np.asarray(u_beta, f32).copy()
    flip = al + be > f32(1)
    al[flip] = f32(1) - al[flip]
    be[flip] = f32(1) - be[flip]
    return (o + al[:, None] * a + be[:, None] * b).astype(f32), pick


class Cell:
    def __init__(self, center, l, w, h, capacity, resolution):
        center = np.asarray(center, f32)
        half = np.array([l / f32(2), w / f32(2), h / f32(2)], f32)
        self.x_min, self.x_max = center - half, center + half
        l, w, h = f32(l), f32(w), f32(h)
        area = max(float(l * np.sqrt(w * w + h * h, dtype=f32)), float(w * np.sqrt(h * h + l * l, dtype=f32)),
                   float(h * np.sqrt(l * l + w * w, dtype=f32)))
        if resolution is None:
            self.capacity = capacity
            self.resolution = 2 * np.sqrt(area / capacity / np.pi)
        elif capacity is None:
            self.resolution = resolution
            self.capacity = int(area // (np.pi * (resolution / 2.0) ** 2))
        else:
            self.resolution, self.capacity = resolution, capacity
        self.pts = np.zeros((0, 3), f32)

    def fill(self, pts, n_point_min=0, seed=0):
        add = pts[(pts - self.x_max).max(-1) < 0]
        if len(add) == 0:
            return
        add = add[(add - self.x_min).min(-1) > 0]
        if len(add) <= n_point_min:
            return
        if len(self.pts) > 0:
            stored = self.pts.astype(np.float64)
            dist = np.empty(len(add))
            for i in range(0, len(add), 256):                       # blocks: the full matrix would not fit for 30 k x 20 k
                d = add[i:i + 256].astype(np.float64)[:, None, :] - stored[None, :, :]
                dist[i:i + 256] = np.sqrt((d * d).sum(-1)).min(-1)
            add = add[dist > self.resolution]
        allp = np.concatenate([self.pts, add], 0)
        if len(allp) > self.capacity:
            allp = allp[sampling.perm_index(np.arange(self.capacity), len(allp), seed & sampling.M32)]
        self.pts = allp


class Scene:
    def __init__(self, x_min, x_max, grid_l, grid_w, grid_h, cell_capacity, cell_resolution):
        self.x_min, self.x_max = np.asarray(x_min, f32), np.asarray(x_max, f32)
        self.grid = (grid_l, grid_w, grid_h)
        d = self.x_max - self.x_min
        self.l, self.w, self.h = d[0] / f32(grid_l), d[1] / f32(grid_w), d[2] / f32(grid_h)
        self.cells = {}
        for i in range(grid_l):
            for j in range(grid_w):
                for k in range(grid_h):
                    c = np.array([self.x_min[0] + f32(0.5 + i) * self.l, self.x_min[1] + f32(0.5 + j) * self.w,
                                  self.x_min[2] + f32(0.5 + k) * self.h], f32)
                    cell = Cell(c, self.l, self.w, self.h, cell_capacity, cell_resolution)
                    self.cells[(i, j, k)] = cell
                    cell_resolution = cell.resolution if cell_resolution is None else cell_resolution
                    cell_capacity = cell.capacity if cell_capacity is None else cell_capacity

    def cells_for_each_pt(self, pts):
        p = np.asarray(pts, f32) - self.x_min
        out = np.empty(p.shape, np.int64)
        for ax, (step, n) in enumerate(zip((self.l, self.w, self.h), self.grid)):
            x = p[:, ax]
            q = (x - np.mod(x, step)) / step                   # floor_divide(x, d) = (x - x % d) / d, torch % = python mod
            q = np.where(q >= n, f32(n - 1), q)
            out[:, ax] = q.astype(np.int64)                    # .long() truncates toward zero
        return np.maximum(out, 0)

    def fill_cells(self, pts, n_point_min=0, seed=0):
        p = np.asarray(pts, f32)
        inside = p[np.all((p >= self.x_min) & (p <= self.x_max), axis=1)]
        for key in sorted({tuple(r) for r in self.cells_for_each_pt(inside).tolist()}):
            i, j, k = key
            lin = (i * self.grid[1] + j) * self.grid[2] + k
            self.cells[key].fill(inside, n_point_min, seed + 0x9E3779B1 * (lin + 1))

    def return_entire_pt_cloud(self):
        return np.concatenate([c.pts for c in self.cells.values()], 0)


def scene_coverage(gt_scene, rec_scene, epsilon):
    """Scene.scene_coverage (mu:3512-3539): per cell, GT points whose nearest recovered point of the same cell is
    closer than epsilon (fp64, heaviside(eps - d, 0) => strict) -> (covered, n_gt)."""
    covered = n_gt = 0
    for key, gc in gt_scene.cells.items():
        if len(gc.pts) == 0:
            continue
        n_gt += len(gc.pts)
        rp = rec_scene.cells[key].pts
        if len(rp) == 0:
            continue
        for i in range(0, len(gc.pts), 256):
            d = gc.pts[i:i + 256].astype(np.float64)[:, None, :] - rp.astype(np.float64)[None, :, :]
            covered += int((np.sqrt((d * d).sum(-1)).min(-1) < epsilon).sum())
    return covered, n_gt
