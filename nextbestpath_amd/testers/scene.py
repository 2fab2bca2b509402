"""Next-best-view rollout on the accelerated simulator -- host-side mirror of compute_trajectory
(macarons/testers/scene.py:491-826), SURVEY.md section 8(f) rank 3.

The reference's loop per pose: true coverage of the current frame (:521-547), surface points + full cloud (:565-583), proxy
points against the depth map with view-state / occupancy / out-of-field updates (:587-610), the occupancy field and, for
every valid neighbour, the coverage gain predicted by the SCONE network (:614-672), the move to the best neighbour (:678-690)
and the four supervision frames with the same proxy updates (:693-815).  Everything but the two network calls is the
random-walk driver's step (testers/random_walk_planning.py); this driver adds the reference's selection rule -- the first
neighbour with the strictly largest gain (:666-668) -- over a gain model:

  * `coverage_gain_fn(rollout, neighbour_idx) -> float` if given (a SCONE replacement plugs in here; it sees the proxy scene
    with its view-state vectors, occupancy and carving counters, the surface scene and the camera);
  * otherwise the geometric model of nbp_view_gain_i32: occupied proxy points inside the neighbour's field of view that have
    not been observed from its direction yet -- one launch for all neighbours.
The MACARONS depth / occupancy networks are not released: perfect depth only."""
from __future__ import annotations

import time

import numpy as np

from ..utility import hipops
from .random_walk_planning import RandomWalkRollout


class NBVRollout(RandomWalkRollout):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.last_gains = None

    def geometric_gains(self, neighbours):
        """[n] int gains of the neighbours under the view-state model (device -> host: one sync, like the reference's
        `coverage_gain > max_coverage_gain` per neighbour)."""
        cam, ps, p = self.camera, self.proxy_scene, self.params
        poses = [cam.pose_from_idx(n) for n in neighbours]
        cams = np.stack([cam.cam12_of_pose(q) for q in poses])
        xs = np.stack([np.asarray(q[:3], np.float32) for q in poses])
        g = hipops.view_gain(ps.proxy_points, ps.proxy_supervision_occ, ps.view_states, cams, xs, ps.view_state_n_elev,
                             ps.view_state_n_azim, cam.image_height, cam.image_width, p.sensor_range)
        return g.cpu().tolist()

    def choose(self, valid):
        """macarons/testers/scene.py:640-672: max_coverage_gain = -1, next_idx = valid[0]; a neighbour replaces it only with a
        strictly larger gain."""
        if self.gain_fn is not None:
            gains = [float(self.gain_fn(self, n)) for n in valid]
        else:
            gains = self.geometric_gains(valid)
        self.last_gains = gains
        best, next_idx = -1.0, valid[0]
        for n, gval in zip(valid, gains):
            if gval > best:
                best, next_idx = gval, n
        return next_idx


def compute_trajectory(params, macarons, camera, gt_scene, surface_scene, proxy_scene, covered_scene, mesh, device,
                       test_resolution=0.05, use_perfect_depth_map=True, compute_collision=False, coverage_gain_fn=None, seed=0,
                       n_poses=None):
    """Same leading arguments and return tuple as the reference (macarons/testers/scene.py:491-826); `macarons` is unused."""
    if not use_perfect_depth_map:
        raise NotImplementedError("the MACARONS depth network is not part of this build (SURVEY.md section 2)")
    t0 = time.time()
    ro = NBVRollout(params, camera, gt_scene, surface_scene, proxy_scene, covered_scene, mesh, device, test_resolution,
                    coverage_gain_fn, seed)
    n = params.n_poses_in_trajectory if n_poses is None else n_poses
    for _ in range(n):
        ro.step()
    print("Trajectory computed in", time.time() - t0, "seconds.")
    print("Coverage Evolution:", ro.coverage_evolution)
    return ro.coverage_evolution, camera.X_cam_history, camera.V_cam_history
