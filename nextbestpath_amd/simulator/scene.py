"""Scene-side setup for the rollout: settings.json schema, scene listing, device mesh, GT surface
cloud.  Mirrors (for the NBP path only) macarons/utility/CustomDataset.py:313-361 (SceneDataset),
macarons/utility/macarons_utils.py:2152-2190 (Settings), :554-572 (load_scene), :612-637 +
macarons/utility/utils.py:1301-1455 (area-weighted GT surface sampling) and the resolution
thinning of Scene.fill_cells (:2952-3036).  Setup-time code: runs once per start pose on the
host (numpy); the per-step work is all in libnbp_hip.so."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass

import numpy as np
import torch

from .mesh import load_obj

f32 = np.float32


class _NS:
    pass


class Settings:
    """settings.json -> .camera / .scene with the scale factor applied to the bounds (mu:2152-2190)."""

    def __init__(self, d, scene_scale_factor=1.0):
        s, c = d["scene"], d["camera"]
        self.scene, self.camera = _NS(), _NS()
        for k in ("grid_l", "grid_w", "grid_h", "cell_capacity", "cell_resolution"):
            setattr(self.scene, k, s[k])
        self.scene.x_min = f32(scene_scale_factor) * np.asarray(s["x_min"], f32)
        self.scene.x_max = f32(scene_scale_factor) * np.asarray(s["x_max"], f32)
        self.camera.x_min = f32(scene_scale_factor) * np.asarray(c["x_min"], f32)
        self.camera.x_max = f32(scene_scale_factor) * np.asarray(c["x_max"], f32)
        for k in ("pose_l", "pose_w", "pose_h"):
            setattr(self.camera, k, c[k])
        self.camera.pose_n_elev, self.camera.pose_n_azim = c["pose_n_theta"], c["pose_n_azim"]
        self.camera.start_positions = [tuple(int(v) for v in p) for p in c["start_positions"]]
        self.camera.contrast_factor = c.get("contrast_factor", 1.0)


class SceneDataset:
    """Lists <data_path>/<scene>/ directories holding one .obj and a settings.json."""

    def __init__(self, data_path, scene_names=None):
        self.data_path = data_path
        names = scene_names if scene_names else sorted(
            d for d in os.listdir(data_path) if os.path.isdir(os.path.join(data_path, d)))
        self.scenes = []
        for name in names:
            p = os.path.join(data_path, name)
            objs = sorted(f for f in os.listdir(p) if f.endswith(".obj"))
            if not objs:
                raise FileNotFoundError(f"no .obj in {p}")
            with open(os.path.join(p, "settings.json")) as fh:
                settings = json.load(fh)
            self.scenes.append({"scene_name": name, "obj_name": objs[0], "settings": settings})

    def __len__(self):
        return len(self.scenes)

    def __getitem__(self, i):
        return self.scenes[i]


@dataclass
class DeviceMesh:
    verts: torch.Tensor      # [V,3] fp32 device (scaled)
    faces: torch.Tensor      # [F,3] int32 device
    verts_host: np.ndarray
    faces_host: np.ndarray
    bin_cap: int = 4096


def load_scene(mesh_path, scene_scale_factor, device):
    """load_scene (mu:554-572) + the trimesh twin (nbp_planning.py:454-455): one scaled mesh for
    rendering AND for the collision tests."""
    v, f = load_obj(mesh_path)
    v = (v * f32(scene_scale_factor)).astype(f32)
    return DeviceMesh(torch.from_numpy(v).to(device), torch.from_numpy(f).to(device), v, f)


def face_areas(verts, faces):
    a, b, c = verts[faces[:, 0]].astype(np.float64), verts[faces[:, 1]].astype(np.float64), verts[faces[:, 2]].astype(
        np.float64)
    return 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)


def sample_gt_surface(verts, faces, n_points, x_min, x_max, resolution, seed=0):
    """GT surface cloud: area-weighted triangle choice + uniform barycentric point (utils.py:1332-1439),
    restricted to faces whose vertices lie inside the scene box (mu:625-626), then thinned so that no
    two kept points share a voxel of size `resolution` (stands in for Cell.fill's cdist thinning,
    mu:3000-3028; same target density, deterministic)."""
    rng = np.random.default_rng(seed)
    inside = np.all((verts >= x_min) & (verts <= x_max), axis=1)
    fsel = faces[inside[faces].all(1)]
    if len(fsel) == 0:
        return np.zeros((0, 3), f32)
    area = face_areas(verts, fsel)
    pick = rng.choice(len(fsel), size=n_points, p=area / area.sum())
    tri = verts[fsel[pick]].astype(np.float64)
    o, a, b = tri[:, 2], tri[:, 0] - tri[:, 2], tri[:, 1] - tri[:, 2]
    al, be = rng.random(n_points), rng.random(n_points)
    flip = al + be > 1.0
    al[flip], be[flip] = 1.0 - al[flip], 1.0 - be[flip]
    pts = (o + al[:, None] * a + be[:, None] * b).astype(f32)
    if resolution and resolution > 0:
        vox = np.floor((pts - x_min) / f32(resolution)).astype(np.int64)
        _, first = np.unique(vox, axis=0, return_index=True)
        pts = pts[np.sort(first)]
    return pts


def y_bins_for(verts_host, n_pieces=4):
    """nbp_planning.py:446-451: torch.arange on python floats (the length rule is torch's)."""
    min_y = float(verts_host[:, 1].min()) + 0.5
    max_y = float(verts_host[:, 1].max()) - 0.5
    w = (max_y - min_y) / n_pieces
    return torch.arange(min_y, max_y + w, w)
