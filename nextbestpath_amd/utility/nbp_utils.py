"""Training data path -- host-side mirror of next_best_path/utility/nbp_utils.py: the replay store
(store_experience :32-44, store_validation_data :78-100, read_random_data_readonly :63-76, read_combined_data
:103-141) and the exploration-with-hindsight trajectory collection that fills it (trajectory_collection :470-852).

Record schema and value encoding are the reference's: a msgpack map with 'current_model_input' [1,5,S,S] f32,
'current_gt_2d_layout' [1,1,S,S] f32, 'target_value_map_pixel' [K,3] i64 (heading, row, col),
'actual_coverage_gain' [K] f32, 'pose_i'; arrays in msgpack-numpy's {nd, type, kind, shape, data} form; keys are
zero-padded millisecond timestamps (`%012d`, 13 digits today).  The container is LMDB: through the `lmdb` module when it is importable,
otherwise through this package's own reader / writer of LMDB's file format (MdbEnv -> csrc/nbp_mdb.cpp; lmdb is not installable in this
image; msgpack-numpy is absent too, hence the explicit encoder below).  LogEnv (an append-only log, rounds 2-5) still opens its own files.

All map work of the collection (cloud accumulation, slab maps, trajectory image, GT obstacle label, coverage,
rendering, un-projection, NBP forward) runs on the HIP kernels; the host keeps the reference's control flow:
Boltzmann goal sampling, uniform-cost search on the mesh-free lattice edges, 60 % random headings, hindsight
relabelling of every (earlier, later) pair of a finished path.
"""
from __future__ import annotations

import math
import os
import random
import struct
import time

import msgpack
import numpy as np
import torch

from . import hipops
from . import planner_host
from . import utils as hu


# ------------------------------------------------------------------ record encoding (msgpack-numpy layout)
def _encode(obj):
    if isinstance(obj, np.ndarray):
        return {b"nd": True, b"type": obj.dtype.str, b"kind": b"", b"shape": list(obj.shape),
                b"data": np.ascontiguousarray(obj).tobytes()}
    if isinstance(obj, (np.bool_, np.number)):
        return {b"nd": False, b"type": obj.dtype.str, b"data": obj.tobytes()}
    raise TypeError(f"cannot pack {type(obj)}")


def _decode(obj):
    nd = obj.get(b"nd", obj.get("nd"))
    if nd is None:
        return obj
    g = lambda k: obj.get(k.encode(), obj.get(k))
    dt = g("type")
    dt = np.dtype(dt.decode() if isinstance(dt, bytes) else dt)
    if nd:
        return np.frombuffer(g("data"), dtype=dt).reshape(g("shape")).copy()
    return np.frombuffer(g("data"), dtype=dt)[0]


def pack_record(data) -> bytes:
    """ref :35-41 (tensors -> numpy -> msgpack with use_bin_type)."""
    np_ = lambda v: v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    return msgpack.packb({
        "current_model_input": np_(data["current_model_input"]),
        "current_gt_2d_layout": np_(data["current_gt_2d_layout"]),
        "target_value_map_pixel": np_(data["target_value_map_pixel"]),
        "actual_coverage_gain": np_(data["actual_coverage_gain"]),
        "pose_i": np.array(data["pose_i"]),
    }, use_bin_type=True, default=_encode)


def unpack_record(value: bytes):
    rec = msgpack.unpackb(value, object_hook=_decode, raw=False, strict_map_key=False)
    rec["pose_i"] = int(np.asarray(rec["pose_i"]))
    return rec


# ------------------------------------------------------------------ containers
class LogEnv:
    """Append-only key/value log with LMDB's ordered-cursor semantics for the few calls the trainer makes.
    File layout: repeated [u16 key length][key][u32 length][payload]; length 0xFFFFFFFF marks a deleted key."""

    def __init__(self, path):
        os.makedirs(path, exist_ok=True)
        self.file = os.path.join(path, "data.log")
        self.index = {}          # key -> (offset, length)
        if os.path.exists(self.file):
            with open(self.file, "rb") as fh:
                while True:
                    kl = fh.read(2)
                    if len(kl) < 2:
                        break
                    klen = struct.unpack("<H", kl)[0]
                    head = fh.read(klen + 4)
                    if len(head) < klen + 4:
                        break
                    key, n = head[:klen], struct.unpack("<I", head[klen:])[0]
                    if n == 0xFFFFFFFF:
                        self.index.pop(key, None)
                        continue
                    self.index[key] = (fh.tell(), n)
                    fh.seek(n, 1)

    def put(self, key: bytes, value: bytes):
        with open(self.file, "ab") as fh:
            fh.write(struct.pack("<H", len(key)) + key + struct.pack("<I", len(value)))
            off = fh.tell()
            fh.write(value)
        self.index[key] = (off, len(value))

    def delete(self, key: bytes):
        if key in self.index:
            with open(self.file, "ab") as fh:
                fh.write(struct.pack("<H", len(key)) + key + struct.pack("<I", 0xFFFFFFFF))
            del self.index[key]

    def entries(self):
        return len(self.index)

    def keys(self):
        return sorted(self.index)

    def items(self):
        with open(self.file, "rb") as fh:
            for k in self.keys():
                off, n = self.index[k]
                fh.seek(off)
                yield k, fh.read(n)

    def close(self):
        pass


class LmdbEnv:
    """The reference's container (train_nbp_model.py:61-63); used when `lmdb` is importable."""

    def __init__(self, path, map_size):
        import lmdb
        os.makedirs(path, exist_ok=True)
        self.env = lmdb.open(path, map_size=map_size)

    def put(self, key, value):
        with self.env.begin(write=True) as txn:
            txn.put(key, value)

    def delete(self, key):
        with self.env.begin(write=True) as txn:
            txn.delete(key)

    def entries(self):
        return self.env.stat()["entries"]

    def keys(self):
        with self.env.begin(write=False) as txn:
            return [bytes(k) for k, _ in txn.cursor()]

    def items(self):
        with self.env.begin(write=False) as txn:
            for k, v in txn.cursor():
                yield bytes(k), bytes(v)

    def close(self):
        self.env.close()


class MdbEnv:
    """<path>/data.mdb in LMDB's on-disk format through the native container of this package (csrc/nbp_mdb.cpp: nbp_mdb_*), for
    hosts without the `lmdb` module: a store written here opens with lmdb.open(path) and one the reference wrote opens here.  Same
    interface as LmdbEnv / LogEnv; every put / delete is its own committed transaction, as in the reference."""

    def __init__(self, path, map_size=200 * 1024 ** 3, sync=False):
        import ctypes as C
        from .. import _lib
        self._L, self._C = _lib.lib(), C
        os.makedirs(path, exist_ok=True)          # (lmdb.open creates the last component only; the trainers hand over nested paths)
        h = C.c_void_p()
        _lib.check(self._L.nbp_mdb_open(os.fsencode(path), int(map_size), int(bool(sync)), C.byref(h)), "nbp_mdb_open")
        self._h = h

    def put(self, key: bytes, value: bytes):
        rc = self._L.nbp_mdb_put(self._h, key, len(key), value, len(value))
        if rc:
            raise RuntimeError(f"nbp_mdb_put failed ({rc})")

    def delete(self, key: bytes):
        rc = self._L.nbp_mdb_del(self._h, key, len(key))
        if rc not in (0, 1):
            raise RuntimeError(f"nbp_mdb_del failed ({rc})")
        return rc == 0

    def get(self, key: bytes):
        C = self._C
        n = C.c_size_t()
        rc = self._L.nbp_mdb_get(self._h, key, len(key), None, 0, C.byref(n))
        if rc == 1:
            return None
        buf = C.create_string_buffer(max(n.value, 1))
        rc = self._L.nbp_mdb_get(self._h, key, len(key), buf, n.value, C.byref(n))
        if rc:
            raise RuntimeError(f"nbp_mdb_get failed ({rc})")
        return buf.raw[:n.value]

    def entries(self):
        return int(self._L.nbp_mdb_entries(self._h))

    def stat(self):
        out = (self._C.c_ulonglong * 8)()
        self._L.nbp_mdb_stat(self._h, out)
        names = ("depth", "branch_pages", "leaf_pages", "overflow_pages", "entries", "last_pgno", "last_txnid", "psize")
        return dict(zip(names, (int(v) for v in out)))

    def keys(self):
        C = self._C
        n = C.c_size_t()
        self._L.nbp_mdb_keys(self._h, None, 0, C.byref(n))
        buf = C.create_string_buffer(max(n.value, 1))
        self._L.nbp_mdb_keys(self._h, buf, n.value, C.byref(n))
        raw, out, i = buf.raw[:n.value], [], 0
        while i < len(raw):
            k = struct.unpack_from("<H", raw, i)[0]
            out.append(raw[i + 2:i + 2 + k])
            i += 2 + k
        return out

    def items(self):
        for k in self.keys():              # (a snapshot of the keys: deleting while iterating is safe, as with a write cursor)
            v = self.get(k)
            if v is not None:
                yield k, v

    def close(self):
        if self._h:
            self._L.nbp_mdb_close(self._h)
            self._h = None

    def __del__(self):                     # (an environment dropped without close(): the file descriptor goes with it)
        try:
            self.close()
        except Exception:
            pass


def open_experience_db(path, map_size=200 * 1024 ** 3):
    """train_nbp_model.py:61-63.  The `lmdb` module when it is importable; otherwise LMDB's file format through the native container
    (MdbEnv) -- unless `path` already holds the append-only log of rounds 2-5 (data.log), which keeps opening as it was written."""
    try:
        import lmdb  # noqa: F401
        return LmdbEnv(path, map_size)
    except ImportError:
        pass
    if os.path.exists(os.path.join(path, "data.log")) and not os.path.exists(os.path.join(path, "data.mdb")):
        return LogEnv(path)
    return MdbEnv(path, map_size)


_last_key = [0]


def store_experience(env, data):
    """ref :32-44.  The key is the millisecond clock; two records in the same millisecond would overwrite each
    other in the reference -- here the key is bumped so that none is lost."""
    ms = max(int(time.time() * 1000), _last_key[0] + 1)
    _last_key[0] = ms
    env.put(f"{ms:012d}".encode(), pack_record(data))


def store_validation_data(env, num=600 * 2):
    """ref :78-100: every ceil(total/num)-th record, up to `num`, is MOVED out of the store."""
    total = env.entries()
    print("Number of total data in the database:", total)
    n = max(math.ceil(total / num), 1)
    selected, delete_keys = [], []
    for count, (key, value) in enumerate(env.items()):
        if count % n == 0 and len(selected) < num:
            selected.append(unpack_record(value))
            delete_keys.append(key)
            if len(selected) == num:
                break
    for key in delete_keys:
        env.delete(key)
    return selected


def store_validation_data_readonly(env, num=600 * 2):
    """ref :46-61."""
    total = env.entries()
    n = max(math.ceil(total / num), 1)
    selected = []
    for count, (key, value) in enumerate(env.items()):
        if count % n == 0 and len(selected) < num:
            selected.append(unpack_record(value))
    return selected


def read_random_data_readonly(env, num_samples=64):
    """ref :63-76."""
    indices = set(random.sample(range(env.entries()), num_samples))
    return [unpack_record(v) for i, (k, v) in enumerate(env.items()) if i in indices]


def read_combined_data(env, sample_m=2304 * 2, sample_size=2176 * 2):
    """ref :103-141: a random sample of the older records + the newest `sample_m` in order."""
    total = env.entries()
    print("number of total data in the database:", total)
    if sample_m is None:
        return [unpack_record(v) for _, v in env.items()]
    n = total - sample_m
    if n < 0:
        n = 1
    sample_indices = set(random.sample(range(n), min(sample_size, n)))
    selected, tail = [], []
    first_tail = max(total - sample_m, 0)
    for i, (key, value) in enumerate(env.items()):
        if i < n and i in sample_indices:
            selected.append(unpack_record(value))
        if i >= first_tail:
            tail.append(unpack_record(value))
    return selected + tail


# ------------------------------------------------------------------ GT obstacle label
def get_binary_obstacle_array(mesh, camera_pose, view_size=80, grid_size=256, reference_label_semantics=True):
    """ref utils.py:226-262 -> [S,S] fp32 {0,1} on the device.  reference_label_semantics (default): the label sits on the pixel
    grid of the reference's matplotlib figure -- 80 units across the columns, 79.48 across the rows, 2.7-px strokes with projecting
    caps (hipops.reference_figure_geometry; pinned by tests/golden/obstacle_label.npz to within one pixel of line position).  False:
    the isotropic +-view_size/2 window at 1.04-px round strokes of rounds 1-5 (nbp_slice_obstacle_f32)."""
    x, y, z = (float(v) for v in list(camera_pose)[:3])
    if reference_label_semantics:
        return hipops.slice_obstacle_fig(mesh.verts, mesh.faces, y, x, z, grid_size, float(view_size))
    return hipops.slice_obstacle(mesh.verts, mesh.faces, y, x, z, grid_size, (-view_size / 2, view_size / 2))


# ------------------------------------------------------------------ trajectory collection
class CollectionRollout:
    """One training rollout of trajectory_collection (ref :470-852) on one scene."""

    BETA = 0.5                     # Boltzmann temperature, ref :719
    P_RANDOM_HEADING = 0.6         # ref :768

    def __init__(self, params, nbp, camera, gt_scene_pc, mesh, y_bins, device, db_env, seed=0, grid=256,
                 value_size=64, grid_range=(-40, 40)):
        from ..testers.nbp_planning import RolloutState
        from .long_term_utils import LatticePlanner
        self.params, self.nbp, self.camera, self.mesh, self.device, self.db = params, nbp, camera, mesh, device, db_env
        self.y_bins, self.S, self.V, self.grid_range = y_bins, grid, value_size, grid_range
        self.st = RolloutState(device, grid=grid)
        self.st.cloud_count.zero_(); self.st.coverage_counts.zero_()
        self.rng = random.Random(seed)
        self.planner = LatticePlanner(camera, mesh, device, value_size, grid, grid_range, rng=self.rng)
        self.gt = gt_scene_pc.contiguous()
        self.bbox = (self.gt.min(0).values.tolist(), self.gt.max(0).values.tolist())
        self.cov_plan = hipops.CoveragePlan(self.gt, 1.0, 2, self.bbox)
        self.gen = torch.Generator().manual_seed(seed)
        self.seed = seed * 1_000_003
        # check_camera_in_mesh for every lattice position, once per scene (static mesh)
        cnt = hipops.axis_ray_counts(mesh.verts, mesh.faces, self.planner.pos_dev).cpu().numpy()
        self.inside = np.all(cnt % 2 == 1, axis=1)
        self.edge_ok = ~self.planner.mesh_hit.astype(bool)             # training_flag branch of get_neighbors
        self.path, self.path_record = [], 0
        self.unreachable = set()
        self.experiences = []
        self.coverage_evolution = []
        self.n_stored = 0
        from ..testers.nbp_planning import _settle_gc
        _settle_gc()                   # the planner's long-lived tables leave the cyclic collector's walks (see there)

    # -- S1-S8 of a step: coverage, current frame, maps, GT label, trajectory image
    def _observe(self, pose_i):
        st, cam, params = self.st, self.camera, self.params
        out = st.coverage_counts[pose_i % st.coverage_counts.shape[0]]
        self.cov_plan.count(st.cloud, out, n_dev=st.cloud_count, n=st.cloud.shape[0], seed=self.seed + 7 * pose_i)
        cov = float(np.float32(out[0].item()) / np.float32(len(self.gt)))
        return cov

    def _inputs(self, pose_i):
        st, cam, params = self.st, self.camera, self.params
        depth, cams = cam.frames_batch([-1])
        hipops.unproject_append(depth, None, cams, st.cloud, st.cloud_count, params.gathering_factor,
                                params.sensor_range, seed=self.seed + 11 * pose_i)
        pose, _ = cam.get_pose_from_idx(cam.cam_idx)
        hu.accumulate_step_maps(st.cloud, pose, self.y_bins, self.S, self.grid_range, n_dev=st.cloud_count, out=st.maps6)
        traj2d = hu.transform_points_to_n_pieces(cam.trajectory_points(), pose)
        traj_img = hu.map_points_to_n_imgs(traj2d, (self.S, self.S), self.grid_range)
        model_input = torch.cat((st.maps6[:4], traj_img), 0).unsqueeze(0).clone()
        gt_obs = get_binary_obstacle_array(self.mesh, pose, self.grid_range[1] * 2, self.S).reshape(1, 1, self.S, self.S)
        return pose, model_input, gt_obs

    def _flush_experiences(self, pose_i):
        """Hindsight relabelling (ref :655-693): every later pose of the finished path that falls inside the
        value map of an earlier one becomes a target pixel (its heading, its cell) with the coverage gained."""
        ex_list = self.experiences
        for a in range(len(ex_list)):
            later = ex_list[a + 1:]
            if not later:
                continue
            pts = torch.tensor([list(e[3][:3]) for e in later], dtype=torch.float32, device=self.device)
            p2d = hu.transform_points_to_n_pieces(pts, ex_list[a][3])
            cells = hu.get_point_position_in_the_img(p2d.squeeze(0), (self.V, self.V), self.grid_range)
            cells = cells.reshape(2, -1).cpu().numpy()
            pixels, gains = [], []
            for j, e in enumerate(later):
                r, c = int(cells[0, j]), int(cells[1, j])
                if 0 <= r < self.V and 0 <= c < self.V:
                    d = e[0] - ex_list[a][0]
                    pixels.append([int(e[4]), r, c])
                    gains.append(d * 100 if d > 0 else 0)
            if pixels:
                store_experience(self.db, {
                    "current_model_input": ex_list[a][1], "current_gt_2d_layout": ex_list[a][2],
                    "target_value_map_pixel": np.asarray(pixels, np.int64),
                    "actual_coverage_gain": np.asarray(gains, np.float32), "pose_i": pose_i})
                self.n_stored += 1
        self.experiences = []

    def _replan(self, pose, model_input):
        """Boltzmann goal sampling + search (ref :695-745).  Returns the path or None."""
        pl, cam = self.planner, self.camera
        with torch.no_grad():
            out1, _ = self.nbp(model_input)
        o1 = out1[0]
        p2d = hu.transform_points_to_n_pieces(pl.pos_dev, pose)
        cells = hu.get_point_position_in_the_img(p2d.squeeze(0), (self.V, self.V), self.grid_range).reshape(2, -1)
        max_gain = o1.amax(0)
        ok = (cells[0] >= 0) & (cells[0] < self.V) & (cells[1] >= 0) & (cells[1] < self.V)
        vals = max_gain[cells[0].clamp(0, self.V - 1), cells[1].clamp(0, self.V - 1)]
        ok_h, vals_h, out1_h = ok.cpu().numpy(), vals.cpu().numpy(), o1.cpu().numpy()
        start_id = pl.node_index[tuple(cam.cam_idx[:3])]
        cand = [n for n in range(len(pl.idx3)) if ok_h[n] and n != start_id]
        if not cand:
            return None
        probs = torch.softmax(torch.from_numpy(vals_h[cand]).double() / self.BETA, 0)
        first = int(torch.multinomial(probs, 1, generator=self.gen).item())
        cand.insert(0, cand.pop(first))
        tree = planner_host.level_order_tree(pl.nbrs, self.edge_ok, start_id)
        hist = np.asarray(cam.cam_idx_history, np.int64).reshape(-1, 5)
        for n in cand:
            if not self.inside[n] or n in self.unreachable:
                continue
            if n not in tree:
                self.unreachable.add(n)
                continue
            ids, cur = [], n
            while cur >= 0:
                ids.append(cur)
                cur = tree[cur]
            nodes = [tuple(pl.idx3[m].tolist()) for m in ids[::-1]]
            full = planner_host.choose_headings(nodes, pl.xyz, pl.node_index, pose, out1_h, hist, self.V, self.grid_range,
                                                rng=self.rng)
            return full[1:]
        return None

    def run(self, n_poses=100, coverage_after_trajectory=None):
        cam, params, st = self.camera, self.params, self.st
        for pose_i in range(n_poses):
            cov = self._observe(pose_i)
            self.coverage_evolution.append(cov)
            if coverage_after_trajectory is not None and pose_i == getattr(params, "n_poses_in_trajectory", -1):
                coverage_after_trajectory.append(cov)
            if cov > 0.95:
                break
            pose, model_input, gt_obs = self._inputs(pose_i)
            if self.path is not None and self.path_record + 1 > len(self.path):
                if self.experiences:
                    self._flush_experiences(pose_i)
                self.path_record = 0
                self.path = self._replan(pose, model_input)
            if self.path is None or len(self.path) == 0:
                break
            self.experiences.append([cov, model_input, gt_obs, list(pose), int(cam.cam_idx[4])])
            if self.path_record >= len(self.path):
                break
            next_idx = list(self.path[self.path_record])
            if self.rng.random() <= self.P_RANDOM_HEADING:
                next_idx[4] = self.rng.randrange(8)
            cam.move_and_capture(self.mesh, next_idx)
            depth, cams = cam.frames_batch([-5, -4, -3, -2])
            hipops.unproject_append(depth, None, cams, st.cloud, st.cloud_count, params.gathering_factor,
                                    params.sensor_range, seed=self.seed + 11 * pose_i + 5)
            self.path_record += 1
        return self.coverage_evolution


def trajectory_collection(params, current_epoch, dataset, db_env, pc2img_size, value_map_size, prediction_range, nbp,
                          coverage_after_trajectory, memory, device, folder_img_path=None, rank=0, world=1, n_poses=100,
                          n_gt_points=None):
    """ref :470-852.  `dataset` is a simulator.scene.SceneDataset; with world > 1 each rank collects the scenes
    rank, rank + world, ... into its own store (collection is embarrassingly parallel)."""
    from ..simulator import scene as sim_scene
    from ..testers.nbp_planning import setup_test_camera
    nbp.eval()
    stored = 0
    for si in range(rank, len(dataset), world):
        sd = dataset[si]
        settings = sim_scene.Settings(sd["settings"], params.scene_scale_factor)
        mesh = sim_scene.load_scene(os.path.join(dataset.data_path, sd["scene_name"], sd["obj_name"]),
                                    params.scene_scale_factor, device)
        y_bins = sim_scene.y_bins_for(mesh.verts_host, 4)
        seed = 7919 * current_epoch + si
        _, gt_dev = sim_scene.setup_gt_scene(params, settings, mesh, device, 0.05, seed=seed, n_points=n_gt_points)
        camera = setup_test_camera(params, mesh, settings.camera.start_positions[0], settings, device, seed=seed)
        ro = CollectionRollout(params, nbp, camera, gt_dev, mesh, y_bins, device, db_env, seed,
                               pc2img_size[0], value_map_size[0], prediction_range)
        ro.run(n_poses, coverage_after_trajectory)
        stored += ro.n_stored
        del ro
        torch.cuda.empty_cache()
    return stored
