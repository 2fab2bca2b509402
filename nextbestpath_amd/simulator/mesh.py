"""Triangle meshes for the simulator: OBJ loading (replaces pytorch3d.io.load_objs_as_meshes /
trimesh.load at macarons/utility/macarons_utils.py:554-572 and
next_best_path/testers/nbp_planning.py:454-455) and a seeded procedural maze generator that
stands in for the AiMDoom scenes, which are not available offline."""
from __future__ import annotations

import json
import os

import numpy as np


def load_obj(path, with_colors=False):
    """Vertices [V,3] fp32 and triangle faces [F,3] int32 (polygons are fan-triangulated;
    negative / slash-separated indices handled).  Texture and normal records are ignored.  with_colors: also the
    per-vertex colours [V,3] of "v x y z r g b" records -- 0.5 gray where a record has none, which is what the reference
    gives an untextured mesh (macarons_utils.py:596-606, TexturesVertex of 0.5)."""
    verts, faces, cols = [], [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
                cols.append((float(p[4]), float(p[5]), float(p[6])) if len(p) >= 7 else (0.5, 0.5, 0.5))
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    v, f = np.asarray(verts, np.float32).reshape(-1, 3), np.asarray(faces, np.int32).reshape(-1, 3)
    if with_colors:
        return v, f, np.asarray(cols, np.float32).reshape(-1, 3)
    return v, f


def save_obj(path, verts, faces, colors=None):
    with open(path, "w") as fh:
        for i, v in enumerate(verts):
            if colors is not None:
                c = colors[i]
                fh.write(f"v {v[0]:.6f} {v[1]:.6f} {v[2]:.6f} {c[0]:.4f} {c[1]:.4f} {c[2]:.4f}\n")
                continue
            fh.write(f"v {v[0]:.6f} {v[1]:.6f} {v[2]:.6f}\n")
        for f in faces:
            fh.write(f"f {f[0] + 1} {f[1] + 1} {f[2] + 1}\n")


class _Builder:
    def __init__(self):
        self.v, self.f = [], []

    def quad(self, p0, e1, e2, n1, n2):
        """Rectangle p0 + s e1 + t e2 tessellated into n1 x n2 cells (2 triangles each)."""
        base = len(self.v)
        p0, e1, e2 = map(lambda a: np.asarray(a, np.float64), (p0, e1, e2))
        for i in range(n1 + 1):
            for j in range(n2 + 1):
                self.v.append(p0 + e1 * (i / n1) + e2 * (j / n2))
        for i in range(n1):
            for j in range(n2):
                a = base + i * (n2 + 1) + j
                b, c, d = a + 1, a + (n2 + 1), a + (n2 + 1) + 1
                self.f += [(a, c, d), (a, d, b)]

    def box(self, lo, hi, pitch):
        lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
        d = hi - lo
        n = np.maximum(1, np.round(d / pitch).astype(int))
        ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
        self.quad(lo, ey, ez, n[1], n[2]); self.quad(lo + ex, ez, ey, n[2], n[1])       # x faces
        self.quad(lo, ez, ex, n[2], n[0]); self.quad(lo + ey, ex, ez, n[0], n[2])       # y faces
        self.quad(lo, ex, ey, n[0], n[1]); self.quad(lo + ez, ey, ex, n[1], n[0])       # z faces

    def arrays(self):
        return np.asarray(self.v, np.float32), np.asarray(self.f, np.int32)


def make_maze_mesh(seed=0, cells=10, size=60.0, height=12.0, wall=0.6, tess=2.5, extra_openings=0.25, hull="slab"):
    """Single-floor maze: floor, ceiling, outer walls and interior walls as closed boxes of
    thickness `wall`, footprint [-size/2, size/2]^2, y in [0, height]; every surface is
    tessellated at pitch `tess` (so the face count scales like AiMDoom's 5-50 k faces).
    Connectivity: random spanning tree over the cell grid + `extra_openings` of the remaining
    walls removed.  hull="shell" makes floor, ceiling and outer walls ONE closed surface around the free
    space (as in a game level), so that check_camera_in_mesh's "odd hit count along +X, +Y, +Z" means
    "inside the level" (training scenes); hull="slab" gives them thickness (the evaluation scenes)."""
    rng = np.random.default_rng(seed)
    half, pitch = size / 2.0, size / cells
    b = _Builder()
    if hull == "shell":
        b.box((-half, 0.0, -half), (half, height, half), tess)
    else:
        b.box((-half - wall, -wall, -half - wall), (half + wall, 0.0, half + wall), tess * 2)           # floor slab
        b.box((-half - wall, height, -half - wall), (half + wall, height + wall, half + wall), tess * 2)   # ceiling slab
    for s in (-1, 1) if hull != "shell" else ():                                                  # outer walls
        x0 = s * half if s > 0 else -half - wall
        b.box((x0, 0.0, -half - wall), (x0 + wall, height, half + wall), tess)
        b.box((-half, 0.0, x0), (half, height, x0 + wall), tess)
    # spanning tree (iterative DFS) over cells; walls[(i,j,axis)] between cell (i,j) and its +axis neighbour
    walls = {(i, j, a) for i in range(cells) for j in range(cells) for a in (0, 1)
             if (i + 1 < cells if a == 0 else j + 1 < cells)}
    seen = {(0, 0)}
    stack = [(0, 0)]
    while stack:
        i, j = stack[-1]
        nbrs = [(i + di, j + dj) for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1))
                if 0 <= i + di < cells and 0 <= j + dj < cells and (i + di, j + dj) not in seen]
        if not nbrs:
            stack.pop()
            continue
        ni, nj = nbrs[rng.integers(len(nbrs))]
        walls.discard((min(i, ni), min(j, nj), 0 if ni != i else 1))
        seen.add((ni, nj))
        stack.append((ni, nj))
    for w in sorted(walls):
        if rng.random() < extra_openings:
            continue
        i, j, a = w
        if a == 0:      # wall on the x = const plane between (i,j) and (i+1,j)
            x = -half + (i + 1) * pitch
            b.box((x - wall / 2, 0.0, -half + j * pitch), (x + wall / 2, height, -half + (j + 1) * pitch), tess)
        else:
            z = -half + (j + 1) * pitch
            b.box((-half + i * pitch, 0.0, z - wall / 2), (-half + (i + 1) * pitch, height, z + wall / 2), tess)
    return b.arrays()


def vertex_colors_for(verts, seed=0):
    """Deterministic per-vertex colours for the procedural mazes (smooth in space, so interpolation is exercised)."""
    v = np.asarray(verts, np.float64)
    ph = 0.37 * seed
    c = np.stack([0.5 + 0.4 * np.sin(1.3 * v[:, 0] + ph), 0.5 + 0.4 * np.sin(2.1 * v[:, 1] + 1.0 + ph),
                  0.5 + 0.4 * np.sin(0.9 * v[:, 2] + 2.0 + ph)], 1)
    return np.round(c, 4).astype(np.float32)


def make_maze_scene(scene_dir, seed=0, cells=10, size=6.0, height=1.2, tess=0.25, n_starts=1, scale=10.0, hull="slab",
                    colors=True):
    """Writes <scene_dir>/<name>.obj + settings.json in UNSCALED units (the drivers multiply by
    scene_scale_factor = 10 on load, like the reference: nbp_planning.py:442,455) with the
    reference's settings schema (macarons/utility/macarons_utils.py:2152-2190)."""
    os.makedirs(scene_dir, exist_ok=True)
    v, f = make_maze_mesh(seed, cells, size, height, wall=0.06, tess=tess, hull=hull)
    name = os.path.basename(os.path.normpath(scene_dir))
    save_obj(os.path.join(scene_dir, name + ".obj"), v, f, vertex_colors_for(v, seed) if colors else None)
    half = size / 2.0
    lattice = int((size * scale - 6.0) // 3.0) + 1          # 3-unit lattice inside [x_min+3, x_max-3] (scaled)
    rng = np.random.default_rng(seed + 1000)
    starts = [[int(rng.integers(1, lattice - 1)), 0, int(rng.integers(1, lattice - 1)), 2, int(rng.integers(0, 8))]
              for _ in range(n_starts)]
    settings = {
        "scene": {"grid_l": 3, "grid_w": 1, "grid_h": 3, "cell_capacity": 20000, "cell_resolution": 0.05,
                  "x_min": [-half, 0.0, -half], "x_max": [half, height, half]},
        "camera": {"x_min": [-half + 0.3, 0.0, -half + 0.3], "x_max": [half - 0.3, height, half - 0.3],
                   "pose_l": lattice, "pose_w": 1, "pose_h": lattice, "pose_n_theta": 5, "pose_n_azim": 8,
                   "start_positions": starts, "contrast_factor": 1.0},
    }
    with open(os.path.join(scene_dir, "settings.json"), "w") as fh:
        json.dump(settings, fh, indent=1)
    return name
