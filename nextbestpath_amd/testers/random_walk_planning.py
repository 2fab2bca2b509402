"""Random-walk / next-best-view baseline rollout on the accelerated simulator -- host-side mirror of
compute_random_walk_trajectory (macarons/testers/random_walk_planning.py:25-400), SURVEY.md section 8(f) rank 3.

What the reference's loop does per pose, and what runs here:
  * covered_scene.fill_cells(current frame) + gt_scene.scene_coverage(covered_scene)  (:64-92)   -> nbp_scene.hip
  * surface_scene.fill_cells(current frame), full_pc append                            (:118-136) -> nbp_scene.hip
  * proxy points in the field of view, signed distance to the depth map, view-state vectors (compute_view_state,
    macarons/utility/scone_utils.py:799-862), supervision occupancy and out-of-field flags
    (:140-166, 305-385; Camera.get_points_in_fov / get_signed_distance_to_depth_maps, Scene.update_proxy_*)
                                                                                                   -> nbp_carve_view_update_f32
  * valid neighbours of the pose lattice (non-empty field of view, :181-182)                      -> nbp_points_in_fov_u8
  * the move (4 interpolated poses, one raster launch) and the 4 supervision frames (:258-360)     -> nbp_sim.hip
  * every recompute_surface_every_n_loop poses the surface scene is rebuilt progressively (:52-59)  -> fill_surface_scene
The occupancy field and the coverage-gain prediction of the reference need the MACARONS / SCONE networks
(compute_scene_occupancy_probability_field, predict_coverage_gain_for_single_camera, :169-215), which are not
released (out of scope): `coverage_gain_fn(camera, neighbour_idx) -> float` may supply a gain
model; without one every step takes a uniformly random valid neighbour (the reference's own 20 % exploration branch,
:250-251); testers/scene.py holds the next-best-view loop that consumes such a model.  The proxy cells (which only index
the proxy points for those networks) are not maintained; the view-state vectors are.  Random draws are
seeded (sub-sampling bijection, random.Random) like the NBP driver."""
from __future__ import annotations

import random
import time

import numpy as np
import torch

from ..simulator import scene as sim_scene
from ..utility import hipops


class RandomWalkRollout:
    def __init__(self, params, camera, gt_scene, surface_scene, proxy_scene, covered_scene, mesh, device,
                 test_resolution=0.05, coverage_gain_fn=None, seed=0, cloud_capacity=6_000_000):
        self.params, self.camera, self.mesh, self.device = params, camera, mesh, device
        self.gt_scene, self.surface_scene, self.proxy_scene, self.covered_scene = gt_scene, surface_scene, proxy_scene, covered_scene
        self.eps = 2 * test_resolution * params.scene_scale_factor
        self.gain_fn = coverage_gain_fn
        self.rng = random.Random(seed)
        self.seed = seed * 1_000_003
        self.full_pc = torch.zeros(cloud_capacity, 3, dtype=torch.float32, device=device)
        self.full_count = torch.zeros(1, dtype=torch.int64, device=device)
        self.part = torch.zeros(8 * 6000 * 4, 3, dtype=torch.float32, device=device)        # one step's partial clouds
        self.part_count = torch.zeros(1, dtype=torch.int64, device=device)
        self.coverage_evolution = []
        self.pose_i = 0

    def _partial(self, which, seed):
        """compute_partial_point_cloud of frames `which` into the scratch cloud -> (points view, device count)."""
        depth, cams = self.camera.frames_batch(which)
        self.part_count.zero_()
        hipops.unproject_append(depth, None, cams, self.part, self.part_count, self.params.gathering_factor,
                                self.params.sensor_range, seed=seed)
        return depth, cams

    def _append_full(self):
        n = int(self.part_count.item())
        c = int(self.full_count.item())
        self.full_pc[c:c + n] = self.part[:n]
        self.full_count += n

    def choose(self, valid):
        """the random walk's rule (:250-251 plus the optional gain model); testers/scene.py::NBVRollout overrides it"""
        if self.gain_fn is not None and self.rng.random() >= 0.2:
            gains = [self.gain_fn(self.camera, n) for n in valid]
            return valid[int(np.argmax(gains))]
        return self.rng.choice(valid)

    def step(self):
        p, cam, pose_i = self.params, self.camera, self.pose_i
        if pose_i > 0 and pose_i % p.recompute_surface_every_n_loop == 0:
            sim_scene.fill_surface_scene(self.surface_scene, self.full_pc, n_dev=self.full_count,
                                         random_sampling_max_size=p.n_gt_surface_points, min_n_points_per_cell_fill=3,
                                         progressive_fill=p.progressive_fill,
                                         max_n_points_per_fill=p.max_points_per_progressive_fill, seed=self.seed + 13 * pose_i)
        # GT surface points of the current frame -> covered scene -> true coverage (:64-92)
        self._partial([-1], self.seed + 11 * pose_i)
        self.covered_scene.fill_cells(self.part, n_dev=self.part_count)
        cov, _ = self.gt_scene.scene_coverage(self.covered_scene, surface_epsilon=self.eps)
        self.coverage_evolution.append(cov)
        # the same frame through the (perfect) depth path -> surface scene + full cloud (:118-136)
        depth, cams = self._partial([-1], self.seed + 11 * pose_i + 3)
        self.surface_scene.fill_cells(self.part, n_dev=self.part_count)
        self._append_full()
        # proxy points against the current depth map (:140-166)
        self.proxy_scene.carve(depth[0], cams[0], p.zfar, p.sensor_range, p.carving_tolerance, X_cam=cam.X_cam)     # + view states
        # next pose among the valid neighbours (:181-251)
        valid = cam.get_valid_neighbors(cam.get_neighboring_poses_2d(), self.mesh)
        next_idx = self.choose(valid)
        cam.move_and_capture(self.mesh, next_idx)
        # the 4 supervision frames: surface points, carving per frame (:305-385)
        depth, cams = self._partial([-5, -4, -3, -2], self.seed + 11 * pose_i + 5)
        self.surface_scene.fill_cells(self.part, n_dev=self.part_count)
        self._append_full()
        for i in range(depth.shape[0]):
            self.proxy_scene.carve(depth[i], cams[i], p.zfar, p.sensor_range, p.carving_tolerance, X_cam=hipops.camera_center(cams[i]))
        self.pose_i += 1


def compute_random_walk_trajectory(params, macarons, camera, gt_scene, surface_scene, proxy_scene, covered_scene, mesh, device,
                                   test_resolution=0.05, use_perfect_depth_map=True, compute_collision=False, n_poses=200,
                                   coverage_gain_fn=None, seed=0):
    """Same leading arguments and return tuple as the reference (random_walk_planning.py:25-400); `macarons` (the depth /
    occupancy networks) is unused: only the perfect-depth path is available."""
    if not use_perfect_depth_map:
        raise NotImplementedError("the MACARONS depth network is not part of this build (SURVEY.md section 2)")
    t1 = time.time()
    ro = RandomWalkRollout(params, camera, gt_scene, surface_scene, proxy_scene, covered_scene, mesh, device, test_resolution,
                           coverage_gain_fn, seed)
    for _ in range(n_poses):
        ro.step()
    n = int(ro.full_count.item())
    print("Time: ", time.time() - t1)
    print("Coverage Evolution:", ro.coverage_evolution)
    return ro.coverage_evolution, camera.X_cam_history, camera.V_cam_history, gt_scene, surface_scene, ro.full_pc[:n], None


def test_random_walk_planning(params_file, model_file, results_json_file, numGPU, test_scenes, test_resolution=0.05,
                              use_perfect_depth_map=True, compute_collision=False, load_json=False, dataset_path=None,
                              configs_dir=None, results_dir=None, n_poses=200, seed=8):
    """Same arguments as the reference (random_walk_planning.py:402-411): every (scene, start pose) of the dataset is
    rolled out and {scene: {start: {coverage, X_cam_history, V_cam_history}}} is written as JSON.  `model_file` (the
    MACARONS weights) is not loaded: perfect depth only."""
    import json
    import os
    from .nbp_planning import load_params, setup_test_camera
    here = os.path.dirname(os.path.abspath(__file__))
    configs_dir = configs_dir or os.path.join(here, "../../configs/macarons")
    results_dir = results_dir or os.path.join(here, "../../data")
    params = load_params(os.path.join(configs_dir, params_file))
    device = torch.device("cuda", numGPU)
    torch.cuda.set_device(device)
    dataset = sim_scene.SceneDataset(dataset_path, test_scenes)
    path = os.path.join(results_dir, results_json_file)
    results = json.load(open(path)) if load_json and os.path.exists(path) else {}
    for si in range(len(dataset)):
        sd = dataset[si]
        settings = sim_scene.Settings(sd["settings"], params.scene_scale_factor)
        mesh = sim_scene.load_scene(os.path.join(dataset.data_path, sd["scene_name"], sd["obj_name"]),
                                    params.scene_scale_factor, device)
        results[sd["scene_name"]] = {}
        for k, start in enumerate(settings.camera.start_positions):
            s = seed + 1000 * si + k
            gt_scene, covered, surface, proxy = sim_scene.setup_test_scenes(params, settings, mesh, device, test_resolution, seed=s)
            camera = setup_test_camera(params, mesh, start, settings, device, seed=s)
            cov, X, V, *_ = compute_random_walk_trajectory(params, None, camera, gt_scene, surface, proxy, covered, mesh, device,
                                                           test_resolution, use_perfect_depth_map, compute_collision, n_poses,
                                                           seed=s)
            results[sd["scene_name"]][str(k)] = {"coverage": cov, "X_cam_history": X.tolist(), "V_cam_history": V.tolist()}
    os.makedirs(results_dir, exist_ok=True)
    with open(path, "w") as fh:
        json.dump(results, fh)
    print("Saved data about test losses in", results_json_file)
    return results
