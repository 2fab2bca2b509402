"""Scene-side setup for the rollout: settings.json schema, scene listing, device mesh, GT surface
cloud.  Mirrors (for the NBP path only) macarons/utility/CustomDataset.py:313-361 (SceneDataset),
macarons/utility/macarons_utils.py:2152-2190 (Settings), :554-572 (load_scene), :612-637 +
macarons/utility/utils.py:1301-1455 (area-weighted GT surface sampling) and the resolution
Scene / Cell point store (:2952-3234, :3512-3539).  The surface sampler is setup-time host numpy (once per
start pose, as in the reference); the store and everything per step is in libnbp_hip.so."""
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
    colors: torch.Tensor = None          # [V,3] fp32 device vertex colours (0.5 gray for untextured meshes)
    colors_host: np.ndarray = None


def load_scene(mesh_path, scene_scale_factor, device):
    """load_scene (mu:554-572) + the trimesh twin (nbp_planning.py:454-455): one scaled mesh for
    rendering AND for the collision tests."""
    v, f, c = load_obj(mesh_path, with_colors=True)
    v = (v * f32(scene_scale_factor)).astype(f32)
    return DeviceMesh(torch.from_numpy(v).to(device), torch.from_numpy(f).to(device), v, f,
                      colors=torch.from_numpy(c).to(device), colors_host=c)


def face_areas(verts, faces):
    """compute_mesh_face_area (macarons/utility/utils.py:1301-1330): Heron's formula in the reference's factored
    form, fp32."""
    fc = np.asarray(verts, f32)[np.asarray(faces)]

    def norm(d):
        return np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=f32)

    a, b, c = norm(fc[:, 0] - fc[:, 1]), norm(fc[:, 1] - fc[:, 2]), norm(fc[:, 2] - fc[:, 0])
    p = (a + b + c) / f32(2)
    if np.any(p <= 0):
        return np.sqrt(np.maximum(p * (p - a) * (p - b) * (p - c), f32(0)), dtype=f32)
    res = ((p - a) / p) * ((p - b) / p) * ((p - c) / p)
    return (np.sqrt(np.maximum(res, f32(0)), dtype=f32) * (p * p)).astype(f32)


def sample_gt_surface(verts, faces, n_points, x_min, x_max, seed=0, uniforms=None):
    """get_scene_gt_surface (mu:612-637) + sample_points_on_mesh_surface (macarons/utility/utils.py:1332-1455):
    faces with all three vertices inside the scene box (inclusive), a face per draw with probability ~ its area
    (first index whose cumulative probability reaches the uniform), a uniform point on it
    (o + alpha a + beta b, (alpha, beta) reflected into the triangle).  No thinning: that is the scene store's
    business (Cell.fill does none on an empty cell, mu:3016).  The reference draws from torch's global generator;
    here the three uniform vectors come from `seed` (or are passed in: tests replay the reference's stream)."""
    verts = np.asarray(verts, f32)
    inside = np.all((verts >= np.asarray(x_min, f32)) & (verts <= np.asarray(x_max, f32)), axis=1)
    fsel = np.asarray(faces)[inside[np.asarray(faces)].all(1)]
    if len(fsel) == 0:
        return np.zeros((0, 3), f32)
    if uniforms is None:
        rng = np.random.default_rng(seed)
        uniforms = tuple(rng.random(n_points, dtype=f32) for _ in range(3))
    u_face, al, be = (np.asarray(u, f32).copy() for u in uniforms)
    area = face_areas(verts, fsel)
    cum = np.cumsum(area / area.sum(dtype=f32), dtype=f32)
    pick = np.searchsorted(cum, u_face, side="left")
    pick = np.where(pick >= len(cum), 0, pick)               # every gap negative -> the reference's argmin returns 0
    tri = verts[fsel[pick]]
    o, a, b = tri[:, 2], tri[:, 0] - tri[:, 2], tri[:, 1] - tri[:, 2]
    flip = al + be > f32(1)
    al[flip], be[flip] = f32(1) - al[flip], f32(1) - be[flip]
    return (o + al[:, None] * a + be[:, None] * b).astype(f32)


class Scene:
    """Device-resident point store with the semantics of the reference's Scene / Cell (mu:2952-3234, 3512-3539):
    a grid of cells over [x_min, x_max], each holding up to `capacity` points; fill_cells thins incoming points at the
    cell resolution against what the cell already holds (never on its first fill) and caps by a random subset;
    scene_coverage compares two stores cell by cell.  The cells are ONE [n_cells, capacity, 3] tensor + counts; every
    operation is a kernel sequence of libnbp_hip.so (csrc/nbp_scene.hip) without a host sync.  Also carries the proxy
    point state of the depth-map carving (mu:3239-3249, 3329-3363).  Point features (colours) are not stored."""

    def __init__(self, x_min, x_max, grid_l, grid_w, grid_h, cell_capacity, cell_resolution, n_proxy_points, device,
                 score_threshold=1.0, seed=0, view_state_n_elev=7, view_state_n_azim=2 * 7):
        self.x_min, self.x_max = np.asarray(x_min, f32).copy(), np.asarray(x_max, f32).copy()
        self.grid_l, self.grid_w, self.grid_h = int(grid_l), int(grid_w), int(grid_h)
        self.device, self.seed, self.n_fills = device, int(seed), 0
        d = self.x_max - self.x_min
        self.l, self.w, self.h = d[0] / f32(self.grid_l), d[1] / f32(self.grid_w), d[2] / f32(self.grid_h)
        # Cell.__init__ (mu:2962-2990): the missing one of (capacity, resolution) follows from the cell's largest face
        l, w, h = self.l, self.w, self.h
        area = max(float(l * np.sqrt(w * w + h * h, dtype=f32)), float(w * np.sqrt(h * h + l * l, dtype=f32)),
                   float(h * np.sqrt(l * l + w * w, dtype=f32)))
        if cell_resolution is None:
            if cell_capacity is None:
                raise NameError("Please choose a capacity or a resolution.")
            cell_resolution = 2 * np.sqrt(area / cell_capacity / np.pi)
        elif cell_capacity is None:
            cell_capacity = int(area // (np.pi * (cell_resolution / 2.0) ** 2))
        self.cell_capacity, self.cell_resolution = int(cell_capacity), float(cell_resolution)
        self.n_cells = self.grid_l * self.grid_w * self.grid_h
        self.cell_pts = torch.zeros(self.n_cells, self.cell_capacity, 3, dtype=torch.float32, device=device)
        self.cell_count = torch.zeros(self.n_cells, dtype=torch.int32, device=device)
        self.n_proxy_points, self.score_threshold = int(n_proxy_points), float(score_threshold)
        self.view_state_n_elev, self.view_state_n_azim = int(view_state_n_elev), int(view_state_n_azim)      # mu:3042, 3106-3109
        self.n_view_state_cameras = self.view_state_n_elev * self.view_state_n_azim
        self.proxy_points = self.view_states = None
        n_per_cell = self.n_proxy_points / self.n_cells
        vol = float(self.l * self.w * self.h) / max(n_per_cell, 1e-30)
        self.distance_between_proxy_points = 2 * np.power(3 * vol / (4 * np.pi), 1.0 / 3.0)

    # ---- geometry handed to the kernels
    def box6(self):
        return np.concatenate([self.x_min, self.x_max]).astype(f32)

    def grid3(self):
        return np.array([self.grid_l, self.grid_w, self.grid_h], np.int32)

    def cell_keys(self):
        return [(i, j, k) for i in range(self.grid_l) for j in range(self.grid_w) for k in range(self.grid_h)]

    # ---- Scene.fill_cells / empty_cells / return_entire_pt_cloud
    def fill_cells(self, pts, features=None, n_point_min=0, n_dev=None):
        from ..utility import hipops
        if features is not None:
            raise NotImplementedError("point features (colours) are not carried by the device store")
        if pts.shape[0] == 0:
            return
        self.n_fills += 1
        hipops.scene_fill_cells(self, pts.contiguous(), n_point_min, seed=self.seed + 7919 * self.n_fills, n_dev=n_dev)

    def empty_cells(self):
        self.cell_count.zero_()

    def return_entire_pt_cloud(self, return_features=False):
        from ..utility import hipops
        return hipops.scene_gather(self)

    def cell_points(self, key):
        """Host-side view of one cell (tests / inspection; one device sync)."""
        i, j, k = key
        c = (i * self.grid_w + j) * self.grid_h + k
        return self.cell_pts[c, :int(self.cell_count[c].item())]

    def scene_coverage(self, recovered_scene, surface_epsilon=None):
        """mu:3512-3539 -> (coverage, n_gt_pts); coverage is 0.0 when nothing matches (one device sync, like
        the reference's .item())."""
        from ..utility import hipops
        eps = 2.0 * self.cell_resolution if surface_epsilon is None else float(surface_epsilon)
        covered, n_gt = hipops.scene_coverage(self, recovered_scene, eps).tolist()
        return (covered / n_gt if n_gt else 0.0), n_gt

    # ---- proxy points of the depth-map carving
    def sample_in_box(self, n_sample, generator=None):
        u = torch.rand(n_sample, 3, generator=generator).to(self.device)
        return torch.from_numpy(self.x_min).to(self.device) + torch.from_numpy(self.x_max - self.x_min).to(self.device) * u

    def initialize_proxy_points(self, n_proxy_points=None, default_proba_value=0.5):
        n = self.n_proxy_points if n_proxy_points is None else int(n_proxy_points)
        dev = self.device
        self.proxy_points = self.sample_in_box(n, torch.Generator().manual_seed(self.seed)).contiguous()
        self.proxy_proba = torch.full((n, 1), default_proba_value, device=dev)
        self.proxy_supervision_occ = torch.ones(n, 1, device=dev)
        self.out_of_field = torch.ones(n, 1, device=dev)
        self.proxy_n_inside_fov = torch.zeros(n, 1, device=dev)
        self.proxy_n_behind_depth = torch.zeros(n, 1, device=dev)
        self.view_states = torch.zeros(n, self.n_view_state_cameras, device=dev)          # mu:3245

    def carve(self, depth, cam12_host, zfar, fov_range, tol, X_cam=None):
        """One depth frame: Camera.get_points_in_fov + get_signed_distance_to_depth_maps (mu:2849-2949) +
        update_proxy_supervision_occ + update_proxy_out_of_field (mu:3329-3363), one fused launch.  With the camera position
        X_cam [3] the same launch also updates the view-state vectors (update_proxy_view_states, mu:3268-3327, as the drivers
        call it: points in the field of view whose signed distance is below 3 x distance_between_proxy_points)."""
        from ..utility import hipops
        if X_cam is None:
            hipops.carve_update(self.proxy_points, depth, None, cam12_host, zfar, fov_range, tol, self.score_threshold,
                                self.proxy_n_inside_fov, self.proxy_n_behind_depth, self.proxy_supervision_occ,
                                self.out_of_field)
            return
        hipops.carve_view_update(self.proxy_points, depth, None, cam12_host, zfar, fov_range, tol, self.score_threshold,
                                 self.proxy_n_inside_fov, self.proxy_n_behind_depth, self.proxy_supervision_occ,
                                 self.out_of_field, X_cam, self.view_state_n_elev, self.view_state_n_azim,
                                 3.0 * self.distance_between_proxy_points, self.view_states)

    def update_proxy_view_states(self, camera, proxy_mask, signed_distances=None, distance_to_surface=None, X_cam=None):
        """mu:3268-3327 with the reference's arguments; proxy_mask [P] bool / uint8; signed_distances either as the reference
        passes them -- one value per point INSIDE the mask (mu:3299-3302: `update_mask[proxy_mask] = sd < dist`), scattered here
        into a [P] buffer -- or already [P] fp32 over all proxy points (entries outside the mask are ignored); any other size
        raises.  X_cam [3] or [1, 3] (default: the camera's position)."""
        from ..utility import hipops
        if X_cam is None:
            X_cam = camera.X_cam
        xc = X_cam.detach().cpu().numpy() if torch.is_tensor(X_cam) else np.asarray(X_cam)
        P = self.proxy_points.shape[0]
        if proxy_mask.numel() != P:
            raise ValueError(f"update_proxy_view_states: proxy_mask has {proxy_mask.numel()} entries, the scene {P} proxy points")
        if signed_distances is not None:
            sd = signed_distances.reshape(-1).to(torch.float32)
            if sd.numel() != P:
                mask_b = proxy_mask.reshape(-1).to(torch.bool)
                n_in = int(mask_b.sum().item())
                if sd.numel() != n_in:
                    raise ValueError(f"update_proxy_view_states: signed_distances has {sd.numel()} values; expected one per masked "
                                     f"point ({n_in}) or one per proxy point ({P})")
                full = torch.full((P,), float("inf"), dtype=torch.float32, device=self.proxy_points.device)
                full[mask_b] = sd.to(full.device)
                sd = full
            signed_distances = sd
        if signed_distances is not None and distance_to_surface is None:
            distance_to_surface = 3 * self.distance_between_proxy_points
        hipops.view_state_update(self.proxy_points, xc.reshape(-1, 3)[:1], self.view_state_n_elev, self.view_state_n_azim,
                                 self.view_states, mask=proxy_mask.to(torch.uint8).contiguous(),
                                 sd=None if signed_distances is None else signed_distances.reshape(-1).contiguous(),
                                 distance_to_surface=distance_to_surface or 0.0)


def fill_surface_scene(surface_scene, full_pc, n_dev=None, random_sampling_max_size=200000, min_n_points_per_cell_fill=3,
                       progressive_fill=True, max_n_points_per_fill=1000, seed=0):
    """mu:691-753: empty the scene and refill it from a shuffled subset of the cloud, progressively (chunks thinned
    against what the earlier chunks stored).  The reference's torch.randperm becomes the seeded index bijection; one
    device sync (the subset size decides the number of chunks, as len(full_pc) does in the reference)."""
    from ..utility import hipops
    sample, m = hipops.sample_points(full_pc, random_sampling_max_size, seed=seed, n_dev=n_dev)
    sample = sample[:int(m.item())]
    surface_scene.empty_cells()
    if not progressive_fill:
        surface_scene.fill_cells(sample, n_point_min=min_n_points_per_cell_fill)
        return
    n_fill = random_sampling_max_size // max_n_points_per_fill + (1 if random_sampling_max_size % max_n_points_per_fill else 0)
    for q in range(n_fill):
        lo = q * max_n_points_per_fill
        hi = lo + max_n_points_per_fill
        chunk = sample[lo:-1] if q == random_sampling_max_size // max_n_points_per_fill else sample[lo:hi]   # ref :741-742
        if len(chunk):
            surface_scene.fill_cells(chunk, n_point_min=min_n_points_per_cell_fill)


def setup_test_scenes(params, settings, mesh, device, test_resolution=0.05, seed=0, n_gt_points=None):
    """setup_test_scene (macarons/testers/scene.py:119-217): gt_scene (filled with the GT surface), covered_scene,
    surface_scene (resolution derived from the capacity) and proxy_scene (proxy points initialised)."""
    x_min, x_max = settings.scene.x_min - f32(0.2), settings.scene.x_max + f32(0.2)
    g = (settings.scene.grid_l, settings.scene.grid_w, settings.scene.grid_h)
    gt_scene, _ = setup_gt_scene(params, settings, mesh, device, test_resolution, seed, n_gt_points)
    covered = Scene(x_min, x_max, *g, params.surface_cell_capacity, test_resolution * params.scene_scale_factor,
                    params.n_proxy_points, device, seed=seed + 1)
    surface = Scene(x_min, x_max, *g, params.surface_cell_capacity, None, params.n_proxy_points, device, seed=seed + 2)
    proxy = Scene(x_min, x_max, *g, params.proxy_cell_capacity, params.proxy_cell_resolution, params.n_proxy_points, device,
                  score_threshold=params.score_threshold, seed=seed + 3)
    proxy.initialize_proxy_points()
    return gt_scene, covered, surface, proxy


def setup_gt_scene(params, settings, mesh, device, test_resolution=0.05, seed=0, n_points=None):
    """setup_test_scene's gt_scene (macarons/testers/scene.py:139-177): a Scene over the scene box grown by 0.2 with
    the surface-cell capacity and resolution test_resolution * scale, filled ONCE with n_gt_surface_points samples of
    the mesh surface -> (gt_scene, gt_scene_pc [G,3] device)."""
    gt_scene = Scene(settings.scene.x_min - f32(0.2), settings.scene.x_max + f32(0.2), settings.scene.grid_l,
                     settings.scene.grid_w, settings.scene.grid_h, params.surface_cell_capacity,
                     test_resolution * params.scene_scale_factor, params.n_proxy_points, device, seed=seed)
    pts = sample_gt_surface(mesh.verts_host, mesh.faces_host, n_points or params.n_gt_surface_points, gt_scene.x_min,
                            gt_scene.x_max, seed=seed)
    gt_scene.fill_cells(torch.from_numpy(pts).to(device))
    return gt_scene, gt_scene.return_entire_pt_cloud()


def y_bins_for(verts_host, n_pieces=4):
    """nbp_planning.py:446-451: torch.arange on python floats (the length rule is torch's)."""
    min_y = float(verts_host[:, 1].min()) + 0.5
    max_y = float(verts_host[:, 1].max()) - 0.5
    w = (max_y - min_y) / n_pieces
    return torch.arange(min_y, max_y + w, w)
