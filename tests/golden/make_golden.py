"""Generates the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference; the GPU box has neither the
reference nor this need -- it consumes the committed .npz files).  The reference is pure
Python with heavy third-party imports that are absent here (pytorch3d, trimesh, torchvision,
lmdb, ...): those are stubbed with MagicMock modules, which is enough because every function
captured below is pure torch / pure Python arithmetic.  Nothing from the reference (source,
bytecode) is written anywhere: the outputs are plain arrays.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
from __future__ import annotations

import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


ABSENT = ("torchsummary", "trimesh", "lmdb", "msgpack_numpy", "torchvision", "pytorch3d", "plotly", "rtree")


class _StubFinder:
    """Serves a MagicMock-backed module for every import under an absent top-level package."""

    @staticmethod
    def find_spec(name, path=None, target=None):
        import importlib.machinery
        if name.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(name, _StubFinder, is_package=True)
        return None

    @staticmethod
    def create_module(spec):
        m = types.ModuleType(spec.name)
        m.__getattr__ = lambda attr, _n=spec.name: MagicMock(name=f"{_n}.{attr}")   # type: ignore
        m.__path__ = []
        return m

    @staticmethod
    def exec_module(module):
        pass


def import_reference():
    sys.meta_path.insert(0, _StubFinder)
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    # entry-point import order (the reverse order hits the reference's import cycle)
    import next_best_path.testers.nbp_planning as planning   # noqa: F401
    import next_best_path.networks.nbp_model as model
    import next_best_path.utility.utils as utils
    import next_best_path.utility.long_term_utils as ltu
    import macarons.utility.macarons_utils as mu
    return model, utils, ltu, mu


def gen_network(model):
    from nextbestpath_amd.utility.synthetic import make_nbp_state_dict, make_count_maps
    sd = make_nbp_state_dict(9)
    net = model.NBP()
    net.load_state_dict(sd, strict=True)          # also proves the 327 keys/shapes match
    net.eval()
    torch.set_num_threads(8)
    # pin the weight generator itself
    probe = {k: float(sd[k].double().sum()) for k in ["Conv1.conv.0.weight", "Conv5.conv.3.weight",
                                                      "Att4_2.W_x.1.running_var", "Final1.weight"]}
    for tag, B, S, seed in [("S32", 1, 32, 11), ("S64B2", 2, 64, 12), ("S128", 1, 128, 13)]:
        x = make_count_maps(B, S, seed=seed)
        with torch.no_grad():
            o1, o2 = net(x)
        np.savez_compressed(os.path.join(HERE, f"nbp_fwd_{tag}.npz"), x=x.numpy(), out1=o1.numpy(), out2=o2.numpy(),
                            weight_seed=9, input_seed=seed,
                            probe_keys=np.array(list(probe)), probe_sums=np.array(list(probe.values())))
        print(tag, "out1", float(o1.abs().max()), "out2", float(o2.min()), float(o2.max()))


def gen_training(model, B=2, S=32, K=7, tag="S32B2"):
    """Train-mode forward (batch-statistics BatchNorm), NBP.loss, parameter gradients and the running statistics
    after one forward of the REFERENCE module: pins oracle/nbp_net.py's train path and nbp_loss.  S32B2 is the small
    (ill-conditioned: the bottleneck BatchNorm sees 8 samples) case, S128B4 the well-conditioned one (256 samples)."""
    from nextbestpath_amd.utility.synthetic import make_nbp_state_dict, make_count_maps
    torch.set_num_threads(8)
    sd = make_nbp_state_dict(9)
    net = model.NBP()
    net.load_state_dict(sd, strict=True)
    net.train()
    x = make_count_maps(B, S, seed=31)
    g = torch.Generator().manual_seed(32)
    coords = torch.stack([torch.randint(0, B, (K,), generator=g), torch.randint(0, 8, (K,), generator=g),
                          torch.randint(0, S // 4, (K,), generator=g), torch.randint(0, S // 4, (K,), generator=g)], 1)
    gains = torch.rand(K, generator=g) * 5
    gt = (torch.rand(B, 1, S, S, generator=g) < 0.1).float()
    o1, o2 = net(x)
    pred = o1[coords[:, 0], coords[:, 1], coords[:, 2], coords[:, 3]]          # nbp_utils.py:373-381
    loss = net.loss(pred, gains, o2, gt)
    loss.backward()
    keys = ["Conv1.conv.0.weight", "Conv1.conv.1.weight", "Conv3.conv.3.weight", "Conv5.conv.4.bias", "Up5_1.up.1.weight",
            "Att4_2.W_g.0.weight", "Att3_2.psi.0.weight", "Att2_2.psi.1.bias", "Up_conv2_2.conv.3.bias", "Final1.weight",
            "Final2.0.bias", "log_vars"]
    named = dict(net.named_parameters())
    grads = {}
    for k in keys:       # small fixtures: at most 4096 strided entries per gradient + its sum and |sum|
        gflat = named[k].grad.detach().double().flatten()
        stride = max(1, (gflat.numel() + 4095) // 4096)
        kk = k.replace(".", "__")
        grads[kk] = gflat[::stride].float().numpy()
        grads[kk + "__stats"] = np.array([stride, float(gflat.sum()), float(gflat.abs().sum())])
    bufs = dict(net.named_buffers())
    np.savez_compressed(os.path.join(HERE, f"nbp_train_{tag}.npz"), x=x.numpy(), coords=coords.numpy(), gains=gains.numpy(),
                        gt=gt.numpy(), out1=o1.detach().numpy(), out2=o2.detach().numpy(), loss=float(loss),
                        grad_keys=np.array(keys), run_mean_Conv1=bufs["Conv1.conv.1.running_mean"].numpy(),
                        run_var_Up5_2=bufs["Up5_2.up.2.running_var"].numpy(), **grads)
    print(f"train {tag}: loss", float(loss), "|grad Conv1|", float(named[keys[0]].grad.abs().max()))


def _sample(t, cap=4096):
    """At most `cap` strided entries of a tensor + (stride, sum, sum |.|, max |.|) of ALL its entries, in float64."""
    f = t.detach().double().flatten()
    stride = max(1, (f.numel() + cap - 1) // cap)
    return f[::stride].numpy(), np.array([stride, float(f.sum()), float(f.abs().sum()), float(f.abs().max())])


def gen_blocks(model):
    """Forward + backward of the reference's block classes (conv_block :8-21, up_conv :23-34, Attention_block :36-62) in train mode on
    the seeded cases of nextbestpath_amd/utility/synthetic.py::BLOCK_CASES -- evaluated in float64 (the module cast with .double():
    the reference's own code path, rounding removed) and in float32 (its distance to the float64 run is recorded per tensor, for
    context).  Stored per case and tensor: <= 4096 strided samples + sums; inputs and parameters are regenerated from the seed by the
    consumer and pinned here by their sums."""
    from nextbestpath_amd.utility.synthetic import BLOCK_CASES, make_block_case
    torch.set_num_threads(8)
    out = {"tags": np.array([r[0] for r in BLOCK_CASES])}
    for row in BLOCK_CASES:
        tag = row[0]
        kind, cin, cout, sd, inputs, dy = make_block_case(tag)
        res = {}
        for dt in (torch.float64, torch.float32):
            blk = {"conv_block": lambda: model.conv_block(sum(cin), cout), "up_conv": lambda: model.up_conv(cin[0], cout),
                   "attention": lambda: model.Attention_block(cin[0], cin[1], cout)}[kind]()
            blk.load_state_dict(sd, strict=True)
            blk = blk.to(dt).train()
            xs = [x.to(dt).clone().requires_grad_(True) for x in inputs]
            if kind == "conv_block":
                xin = torch.cat(xs, 1) if len(xs) > 1 else xs[0]
                y = blk(xin)
            elif kind == "up_conv":
                y = blk(xs[0])
            else:
                y = blk(xs[0], xs[1])          # (g, x)
            y.backward(dy.to(dt))
            r = {"y": y.detach()}
            for i, x in enumerate(xs):
                r[f"dx{i}"] = x.grad
            for k, p_ in blk.named_parameters():
                r["d__" + k.replace(".", "__")] = p_.grad
            for k, b_ in blk.named_buffers():
                if k.endswith("running_mean") or k.endswith("running_var"):
                    r["buf__" + k.replace(".", "__")] = b_.detach()
            res[dt] = r
        r64, r32 = res[torch.float64], res[torch.float32]
        worst = 0.0
        for k, v in r64.items():
            smp, st = _sample(v)
            out[f"{tag}__{k}"] = smp
            out[f"{tag}__{k}__stats"] = st
            e32 = float((r32[k].double() - v).abs().max()) / max(float(v.abs().max()), 1e-30)
            out[f"{tag}__{k}__fp32_err"] = np.array(e32)
            worst = max(worst, e32)
        # the ReLUs really are open (the premise of the 1e-5 comparison): smallest pre-activation margin is recorded by the consumer's
        # own check; here the inputs / parameters are pinned
        out[f"{tag}__pin"] = np.array([float(sum(x.double().sum() for x in inputs)), float(dy.double().sum()),
                                       float(sum(v.double().sum() for v in sd.values() if v.is_floating_point()))])
        print(f"block {tag}: {kind} {cin}->{cout}: {len(r64)} tensors, worst torch-fp32 distance to fp64 {worst:.2e} of the tensor's max")
    np.savez_compressed(os.path.join(HERE, "nbp_blocks_bwd.npz"), **out)


def label_scene():
    """A small walled room with an inner wall, a box and a rotated box (oblique segments), y up, 3 units high: [V,3] f32, [F,3] i64."""
    v, f = [], []

    def wall(a, b, h=3.0):
        i = len(v)
        v.extend([[a[0], 0, a[1]], [b[0], 0, b[1]], [b[0], h, b[1]], [a[0], h, a[1]]])
        f.extend([[i, i + 1, i + 2], [i, i + 2, i + 3]])

    def loop(pts):
        for k in range(len(pts)):
            wall(pts[k], pts[(k + 1) % len(pts)])
    loop([(-20, -10), (25, -10), (25, 30), (-20, 30)])
    loop([(0, 0), (5, 0), (5, 12), (0, 12)])
    wall((-20, 8), (-6, 8))
    c, s_ = np.cos(0.5), np.sin(0.5)
    loop([(12 + c * a - s_ * b, 18 + s_ * a + c * b) for a, b in [(-6, -3), (6, -3), (6, 3), (-6, 3)]])
    loop([(-12 + 4 * np.cos(t), 20 + 4 * np.sin(t)) for t in np.linspace(0, 2 * np.pi, 12, endpoint=False)])
    return np.asarray(v, np.float32), np.asarray(f, np.int64)


def gen_obstacle_label(utils):
    """get_binary_obstacle_array (next_best_path/utility/utils.py:226-262) RUN AS IT IS -- matplotlib figure, PNG, PIL resize,
    flip, threshold -- with trimesh (absent) as the one stub: mesh_plane returns the segments of oracle/slice_raster.py::
    plane_segments, load_path a path of one two-point entity per segment.  Pillow >= 10 dropped Image.ANTIALIAS (an alias of
    LANCZOS): restored for the call.  Stored: the scene, the poses, the labels (bit-packed)."""
    from PIL import Image
    from oracle.slice_raster import plane_segments
    if not hasattr(Image, "ANTIALIAS"):
        Image.ANTIALIAS = Image.LANCZOS
    verts, faces = label_scene()
    poses = np.array([[3.0, 1.5, 4.0, 0, 0], [-30.0, 1.5, 10.0, 0, 0], [10.0, 0.7, -20.0, 0, 0], [40.0, 2.2, 45.0, 0, 0],
                      [-9.3, 1.1, 21.7, 0, 0]], np.float32)
    labels = []
    for pose in poses:
        segs = plane_segments(verts, faces, pose[1]).astype(np.float64)
        path = types.SimpleNamespace(vertices=segs.reshape(-1, 3),
                                     entities=[types.SimpleNamespace(points=np.array([2 * i, 2 * i + 1])) for i in range(len(segs))])
        utils.trimesh = types.SimpleNamespace(intersections=types.SimpleNamespace(mesh_plane=lambda mesh, n, o, _s=segs: _s),
                                              load_path=lambda inter, _p=path: _p)
        lab = utils.get_binary_obstacle_array(None, torch.from_numpy(pose))
        assert lab.shape == (256, 256) and set(np.unique(lab)) <= {0, 1}
        labels.append(np.packbits(lab.astype(np.uint8), axis=None))
        print("obstacle label", pose[:3], "segments", len(segs), "pixels", int(lab.sum()))
    import matplotlib, PIL
    np.savez_compressed(os.path.join(HERE, "obstacle_label.npz"), verts=verts, faces=faces, poses=poses, labels=np.stack(labels),
                        versions=np.array([matplotlib.__version__, PIL.__version__]))


def gen_carve(mu):
    """Depth-map space carving through the REFERENCE's own code: Camera.get_points_in_fov (macarons_utils.py:
    2849-2884), Camera.get_signed_distance_to_depth_maps (:2900-2949: mask fill, NDC -> grid scaling, F.grid_sample bilinear /
    border / align_corners=False, the subtraction) and Scene.update_proxy_out_of_field / update_proxy_supervision_occ (:3329-3363).
    The one stub is the PyTorch3D camera object: its two transforms return the view-space points and NDC projections that
    oracle/camera.py's restatement of the library's conventions computes (fp32, R / T from its look_at) -- so what this fixture pins
    is everything DOWNSTREAM of the projection; the projection itself stays parity-unpinned (package absent)."""
    from oracle import camera as ocam
    H, W = 64, 114
    renderer = types.SimpleNamespace(rasterizer=types.SimpleNamespace(raster_settings=types.SimpleNamespace(image_size=(H, W))))
    x_min, x_max = torch.tensor([-24.0, 0.0, -21.0]), torch.tensor([24.0, 12.0, 21.0])
    cam = mu.Camera(x_min, x_max, 15, 1, 13, 5, 8, 4, 1000.0, renderer, "cpu", contrast_factor=1.0, gathering_factor=0.05)
    rng = np.random.default_rng(20)
    out = {"HW": np.array([H, W]), "zfar": np.array(1000.0, np.float32), "tol": np.array(10.0, np.float32),
           "score_threshold": np.array(0.95, np.float32)}
    P = 6000
    pts = rng.uniform([-60, -5, -60], [60, 25, 60], (P, 3)).astype(np.float32)
    n_inside = rng.integers(0, 4, (P, 1)).astype(np.float32)
    n_behind = np.minimum(rng.integers(0, 4, (P, 1)), n_inside).astype(np.float32)
    occ = np.ones((P, 1), np.float32)
    oof = np.ones((P, 1), np.float32)
    out.update(pts=pts, n_inside_init=n_inside.copy(), n_behind_init=n_behind.copy())
    scene = types.SimpleNamespace(proxy_n_inside_fov=torch.from_numpy(n_inside), proxy_n_behind_depth=torch.from_numpy(n_behind),
                                  proxy_supervision_occ=torch.from_numpy(occ), out_of_field=torch.from_numpy(oof), score_threshold=0.95)
    for k, (X, V, fov_range) in enumerate([((3.0, 3.3, -6.0), (0.0, 45.0), 70.0), ((-15.0, 3.3, 12.0), (-30.0, 200.0), 70.0),
                                           ((20.0, 3.3, 18.0), (30.0, 315.0), 40.0)]):
        R, T = ocam.camera_RT(np.array(X), np.array(V))
        # depth map of this view: a smooth surface 5 .. 60 units away with "no hit" holes (zbuf -1, mask False)
        yy, xx = np.mgrid[0:H, 0:W]
        depth = (20.0 + 15.0 * np.sin(xx / 9.0 + k) + 10.0 * np.cos(yy / 7.0) + rng.uniform(0, 3, (H, W))).astype(np.float32)
        hole = rng.random((H, W)) < 0.15
        depth[hole] = -1.0
        mask = ~hole
        # the stub camera: view-space points and NDC projections by oracle/camera.py's formulae (fp32)
        p = pts
        v = np.empty_like(p)
        for j in range(3):
            v[:, j] = ((p[:, 0] * R[0, j] + p[:, 1] * R[1, j]) + p[:, 2] * R[2, j]) + T[j]
        with np.errstate(divide="ignore", invalid="ignore"):
            nx = v[:, 0] / (v[:, 2] * ocam.TAN_HALF_FOV)
            ny = v[:, 1] / (v[:, 2] * ocam.TAN_HALF_FOV)
        proj = np.stack([nx, ny, np.ones_like(nx)], 1).astype(np.float32)
        C = ocam.camera_center(R, T)
        index = {pp.tobytes(): i for i, pp in enumerate(p)}

        def rows(q):
            return np.array([index[r.tobytes()] for r in q.numpy()], np.int64)

        class _Tf:
            def __init__(self, table):
                self.table = table

            def transform_points(self, q):
                return torch.from_numpy(self.table[rows(q)].copy())
        fov_camera = types.SimpleNamespace(get_full_projection_transform=lambda: _Tf(proj), get_world_to_view_transform=lambda: _Tf(v),
                                           get_camera_center=lambda: torch.from_numpy(C).view(1, 3), R=torch.from_numpy(R).view(1, 3, 3))
        tp = torch.from_numpy(pts)
        fov_pts, fov_mask = cam.get_points_in_fov(tp, return_mask=True, fov_camera=fov_camera, fov_range=fov_range)
        sd = cam.get_signed_distance_to_depth_maps(pts=fov_pts, depth_maps=torch.from_numpy(depth).view(1, H, W, 1),
                                                   mask=torch.from_numpy(mask).view(1, H, W, 1), fov_camera=fov_camera)
        mu.Scene.update_proxy_out_of_field(scene, fov_mask)
        mu.Scene.update_proxy_supervision_occ(scene, fov_mask, sd, tol=10.0)
        out.update({f"R{k}": R, f"T{k}": T, f"depth{k}": depth, f"mask{k}": mask, f"fov_range{k}": np.array(fov_range, np.float32),
                    f"fov_mask{k}": fov_mask.numpy(), f"sd{k}": sd.view(-1).numpy(),
                    f"n_inside{k}": scene.proxy_n_inside_fov.numpy().copy(), f"n_behind{k}": scene.proxy_n_behind_depth.numpy().copy(),
                    f"occ{k}": scene.proxy_supervision_occ.numpy().copy(), f"oof{k}": scene.out_of_field.numpy().copy()})
        print(f"carve view {k}: {int(fov_mask.sum())} of {P} points in the field of view, sd range {float(sd.min()):.1f} .. {float(sd.max()):.1f}, "
              f"occ = 0 on {int((scene.proxy_supervision_occ == 0).sum())}")
    out["ndc_minmax"] = np.array([float(cam.min_ndc_x), float(cam.max_ndc_x), float(cam.min_ndc_y), float(cam.max_ndc_y)], np.float32)
    np.savez_compressed(os.path.join(HERE, "carve.npz"), **out)


def gen_maps(utils):
    rng = np.random.default_rng(21)
    dev = torch.device("cpu")
    S = 256
    pose = torch.tensor([3.25, 13.3, -7.5, 0.0, 90.0])
    # random cloud + adversarial points: exact half-integer cells, the +-40 window edge
    n = 20000
    pts = rng.uniform(-60, 60, (n, 3)).astype(np.float32)
    pts[:, 1] = rng.uniform(5, 35, n)
    special = []
    scale = np.float32(256 / 80)
    for cell in [0.5, 1.5, 2.5, 127.5, 128.5, 254.5, 255.5, -0.5, 255.4999, 256.0, 255.0, 0.0]:
        v = np.float32(cell) / scale - np.float32(40)        # v such that (v+40)*scale ~ cell
        for dv in (0.0, 1e-6, -1e-6):
            # v0 = -(z - cz)  =>  z = cz - v0
            special.append([pose[0].item() - (v + dv), 13.3, pose[2].item() - (v + dv)])
    pts = np.concatenate([pts, np.array(special, np.float32)], 0)
    p = torch.from_numpy(pts)
    t2d = utils.transform_points_to_n_pieces(p, pose, dev)
    img = utils.map_points_to_n_imgs(t2d, (S, S), (-40, 40), dev)
    pos256 = utils.get_point_position_in_the_img(t2d.squeeze(0)[:64], (S, S), (-40, 40))
    pos64 = utils.get_point_position_in_the_img(t2d.squeeze(0)[:64], (64, 64), (-40, 40))
    pos1 = utils.get_point_position_in_the_img(t2d.squeeze(0)[5:6].squeeze(0), (S, S), (-40, 40))
    # slab split: the inline code of next_best_path/testers/nbp_planning.py:114-127,446-451
    out = {}
    for tag, (min_v, max_v) in {"nominal": (4.2, 36.1), "six_bins": (0.0, 30.0)}.items():
        n_pieces = 4
        min_y, max_y = min_v + 0.5, max_v - 0.5
        bin_width = (max_y - min_y) / n_pieces
        y_bins = torch.arange(min_y, max_y + bin_width, bin_width)
        if tag == "six_bins":       # a 6-entry y_bins (float arange may overshoot): slab 4 must be dropped
            y_bins = torch.tensor([0.5, 6.0, 11.5, 17.0, 22.5, 28.0])
        bins = torch.bucketize(p[:, 1], y_bins[:-1]) - 1
        imgs = []
        for i in range(n_pieces):
            g = p[bins == i]
            if len(g) > 0:
                imgs.append(utils.map_points_to_n_imgs(utils.transform_points_to_n_pieces(g, pose, dev), (S, S),
                                                       (-40, 40), dev))
            else:
                imgs.append(torch.zeros(1, S, S))
        out[f"ybins_{tag}"] = y_bins.numpy()
        out[f"slabs_{tag}"] = torch.cat(imgs, 0).numpy().astype(np.uint16)
    # height band (nbp_planning.py:178-183)
    cy = pose[1].item()
    m = (p[:, 1] < cy + 0.1) & (p[:, 1] > cy - 0.1)
    band = utils.map_points_to_n_imgs(utils.transform_points_to_n_pieces(p[m], pose, dev), (S, S), (-40, 40), dev)
    # batched form n=2
    t2 = torch.stack([t2d[0, :5000], t2d[0, 5000:10000]])
    img2 = utils.map_points_to_n_imgs(t2, (128, 128), (-40, 40), dev)
    np.savez_compressed(os.path.join(HERE, "maps.npz"), points=pts, pose=pose.numpy(), t2d=t2d.numpy(),
                        img=img.numpy().astype(np.uint16), pos256=pos256.numpy(), pos64=pos64.numpy(),
                        pos1=pos1.numpy(), band=band.numpy().astype(np.uint16),
                        img2=img2.numpy().astype(np.uint16), **out)
    print("maps: total count", float(img.sum()), "six-bin len", len(out["ybins_six_bins"]))


def gen_planner(ltu, mu):
    rng = np.random.default_rng(31)
    # Bresenham truth table (long_term_utils.py:277-298)
    ends = rng.integers(0, 64, (200, 4))
    ends[:8] = [[0, 0, 0, 0], [0, 0, 5, 0], [0, 0, 0, 5], [5, 5, 0, 0], [3, 7, 9, 2], [63, 0, 0, 63], [10, 10, 11, 40],
                [40, 11, 10, 10]]
    lines, lens = [], []
    for x0, y0, x1, y1 in ends.tolist():
        pts = ltu.bresenham_line(x0, y0, x1, y1)
        lens.append(len(pts))
        lines.extend(pts)
    # edge test (long_term_utils.py:300-331) on a random obstacle layout
    S = 256
    layout = (torch.from_numpy(rng.random((1, 1, S, S))) < 0.08).float()
    pose = torch.tensor([1.0, 13.3, -2.0, 0.0, 0.0])
    p1 = torch.from_numpy(rng.uniform(-45, 45, (300, 3)).astype(np.float32))
    step = torch.from_numpy(rng.choice([-3.0, 0.0, 3.0], (300, 3)).astype(np.float32))
    step[:, 1] = 0
    p2 = p1 + step
    blocked = [bool(ltu.line_across_image_pixel(p1[i], p2[i], pose, (S, S), (-40, 40), layout, torch.device("cpu")))
               for i in range(300)]
    # check_pixel_values (macarons_utils.py:86-100)
    proj = torch.from_numpy(rng.poisson(0.002, (1, 1, S, S)).astype(np.float32))
    proj[proj > 1] = 1
    cells = rng.integers(0, S, (200, 2))
    cpv = [bool(mu.check_pixel_values(proj, torch.tensor(c))) for c in cells.tolist()]
    # coverage (long_term_utils.py:437-468)
    gt = torch.from_numpy(rng.uniform(-20, 20, (1500, 3)).astype(np.float32))
    pc = torch.from_numpy(rng.uniform(-20, 20, (9000, 3)).astype(np.float32))
    pc[:, 1] *= 0.3
    torch.manual_seed(5)
    cov = ltu.calculate_coverage_percentage(gt, pc)
    torch.manual_seed(5)
    perm = torch.randperm(pc.shape[0])[:int(len(gt) * 2)]
    cov_small = ltu.calculate_coverage_percentage(gt, pc[:1000])      # no subsampling branch
    cov_empty = ltu.calculate_coverage_percentage(gt, pc[:0])
    auc = ltu.compute_auc(np.linspace(0, 0.8, 101))
    np.savez_compressed(os.path.join(HERE, "planner.npz"), ends=ends, line_pts=np.array(lines), line_lens=np.array(lens),
                        layout=layout.numpy().astype(np.uint8), edge_pose=pose.numpy(), edge_p1=p1.numpy(),
                        edge_p2=p2.numpy(), edge_blocked=np.array(blocked), proj=proj.numpy().astype(np.uint8),
                        cpv_cells=cells, cpv=np.array(cpv), cov_gt=gt.numpy(), cov_pc=pc.numpy(),
                        cov_perm=perm.numpy(), cov=cov, cov_small=cov_small, cov_empty=cov_empty, auc=auc)
    print("planner: blocked", sum(blocked), "/300  cpv", sum(cpv), "/200  cov", cov, cov_small, cov_empty)


def gen_replan(utils, ltu, mu):
    """Obstacle fusion + candidate scoring (nbp_planning.py:166-233) and generate_Dijkstra_path
    (long_term_utils.py:334-418) on a synthetic lattice; the inline tester code is restated with
    the reference's own functions doing the arithmetic."""
    import types as _t
    rng = np.random.default_rng(41)
    dev = torch.device("cpu")
    S, V = 256, 64
    pose = torch.tensor([4.0, 13.3, -5.0, 0.0, 0.0])
    # lattice 12 x 12 positions, 3-unit pitch, partly outside the +-40 window
    ii, kk = np.meshgrid(np.arange(12), np.arange(12), indexing="ij")
    idx = np.stack([ii.ravel(), np.zeros(144, int), kk.ravel()], 1)
    pos = np.stack([-20.0 + 3.0 * idx[:, 0] + 4.0, np.full(144, 13.3), -50.0 + 3.0 * idx[:, 2] - 5.0], 1).astype(np.float32)
    out1 = torch.from_numpy(rng.normal(0, 1, (1, 8, V, V)).astype(np.float32))
    out2 = torch.from_numpy(np.where(rng.random((1, 1, S, S)) < 0.06, 0.13 + 0.8 * rng.random((1, 1, S, S)),
                                     0.13 * rng.random((1, 1, S, S))).astype(np.float32))
    out2[0, 0, 100, 100] = 0.13        # threshold is >= (nbp_planning.py:168)
    full = torch.from_numpy(rng.poisson(0.03, (1, 1, S, S)).astype(np.float32))
    band = torch.from_numpy((rng.random((1, 1, S, S)) < 0.3).astype(np.float32)) * (full > 0)
    traj = torch.from_numpy((rng.random((1, 1, S, S)) < 0.01).astype(np.float32))
    # --- fusion (nbp_planning.py:166-191)
    obst = (out2 >= 0.13).float()
    fullproj = full.clone()
    fullproj[fullproj > 1] = 1
    filt = band.clone()
    filt[filt > 0] = 1
    mask_layout = fullproj > 0
    obst[mask_layout] = filt[mask_layout]
    obst[traj > 0] = 0
    max_gain, _ = torch.max(out1, dim=1, keepdim=True)
    # --- scoring (nbp_planning.py:203-233)
    skip = rng.random(144) < 0.05
    rows = []
    for i in range(144):
        if skip[i]:
            continue
        p3 = torch.from_numpy(pos[i])
        p2 = utils.transform_points_to_n_pieces(p3.unsqueeze(0), pose, dev)
        g = utils.get_point_position_in_the_img(p2.squeeze(0), (V, V), (-40, 40))
        if 0 <= g[0] < V and 0 <= g[1] < V:
            val = max_gain[0, 0, g[0], g[1]]
            sel = utils.get_point_position_in_the_img(p2.squeeze(0), (S, S), (-40, 40))
            selp = torch.tensor([sel[0], sel[1]])
            dens = fullproj[0, 0, selp[0], selp[1]]
            if mu.check_pixel_values(fullproj, selp):
                rows.append([i, int(g[0]), int(g[1]), val.item() - 10 * dens.item()])
    order = sorted(range(len(rows)), key=lambda r: rows[r][-1], reverse=True)     # list.sort(reverse=True) is stable
    # --- Dijkstra (long_term_utils.py:334-418)
    pose_space = {str([int(a), int(b), int(c)]).replace(", ", ",  "): torch.from_numpy(pos[n])
                  for n, (a, b, c) in enumerate(idx.tolist())}
    cam = _t.SimpleNamespace(cam_idx_history=torch.tensor([[5., 0., 6., 2., 3.], [5., 0., 7., 2., 1.]]))
    start = [5, 0, 7]
    goals, paths, plens = [], [], []
    collision = [[[5, 0, 7], [5, 0, 8]], [[5, 0, 8], [5, 0, 7]]]
    passable = [[[5, 0, 7], [5, 0, 6]], [[5, 0, 6], [5, 0, 7]]]
    import random as _r
    for gi in [rows[r][0] for r in order[:12]] + [0, 143]:
        _r.seed(3)
        goal = idx[gi].tolist()
        pth = ltu.generate_Dijkstra_path(pose_space, start, goal, None, pose, cam, (V, V), (-40, 40), out1, dev,
                                         layout_image=obst, layout_size=(S, S), collision_list=collision,
                                         training_flag=False, passable_list=passable)
        goals.append(gi)
        if pth is None:
            plens.append(-1)
        else:
            plens.append(len(pth))
            paths.extend(pth.long().tolist())
    np.savez_compressed(os.path.join(HERE, "replan.npz"), pose=pose.numpy(), idx=idx, pos=pos, out1=out1.numpy(),
                        out2=out2.numpy()[0, 0], full=full.numpy()[0, 0].astype(np.uint8), band=band.numpy()[0, 0].astype(np.uint8),
                        traj=traj.numpy()[0, 0].astype(np.uint8), obst=obst.numpy()[0, 0].astype(np.uint8),
                        fullproj=fullproj.numpy()[0, 0].astype(np.uint8), skip=skip,
                        cand=np.array([r[:3] for r in rows]), cand_score=np.array([r[3] for r in rows]),
                        cand_order=np.array(order), start=np.array(start), goals=np.array(goals),
                        path_lens=np.array(plens), paths=np.array(paths), cam_hist=cam.cam_idx_history.numpy(),
                        collision=np.array(collision), passable=np.array(passable))
    print("replan: candidates", len(rows), "paths", plens)


def gen_scene(mu):
    """GT surface store: compute_mesh_face_area / sample_mesh_triangle /
    sample_points_on_mesh_faces (macarons/utility/utils.py:1301-1455), get_scene_gt_surface (macarons_utils.py:612-637),
    Scene.get_cells_for_each_pt / fill_cells / return_entire_pt_cloud and Cell.fill (macarons_utils.py:2952-3234) run
    UNMODIFIED on CPU tensors; the only shim is Tensor.get_device() -> 'cpu' (the reference passes its result to
    .to(), which rejects the -1 a CPU tensor reports).  The uniform draws the reference consumed are recorded (same
    seed, same shapes, same order) so that a restatement can be fed the identical stream."""
    import macarons.utility.utils as mutils
    from macarons.utility.CustomGeometry import get_cartesian_coords
    from nextbestpath_amd.simulator.mesh import make_maze_mesh
    orig_get_device = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: "cpu"
    try:
        dev = "cpu"
        v, f = make_maze_mesh(seed=2, cells=3, size=18.0, height=6.0, wall=0.6, tess=3.0)
        verts, faces = torch.from_numpy(v), torch.from_numpy(f.astype(np.int64))
        x_min, x_max = torch.tensor([-9.0, 0.0, -9.0]) - 0.2, torch.tensor([9.0, 6.0, 9.0]) + 0.2
        n_pts = 3000

        def scene(capacity, resolution):
            return mu.Scene(x_min=x_min, x_max=x_max, grid_l=3, grid_w=1, grid_h=3, cell_capacity=capacity,
                            cell_resolution=resolution, n_proxy_points=900, device=dev, feature_dim=0)

        sc = scene(2000, 0.5)
        areas = mutils.compute_mesh_face_area(verts, faces)
        torch.manual_seed(21)
        gt = mu.get_scene_gt_surface(sc, verts, faces, n_pts)
        # the draws get_scene_gt_surface consumed: sample_mesh_triangle (one batch: 1e7 / n_faces >= n_pts), alpha, beta
        torch.manual_seed(21)
        u_face = torch.rand(n_pts, 1)
        u_alpha = torch.rand(n_pts, 1)
        u_beta = torch.rand(n_pts, 1)
        _, inside = sc.get_pts_in_bounding_box(verts, return_mask=True)
        cells_of = sc.get_cells_for_each_pt(gt)
        # --- first fill (empty cells: no thinning), second fill (fp64 cdist thinning at the resolution), capacity cap
        torch.manual_seed(22)
        sc.fill_cells(gt)
        keys = list(sc.cells.keys())
        first = [sc.cells[k].cell_pts.clone() for k in keys]
        entire_first = sc.return_entire_pt_cloud(return_features=False)
        extra = gt[:1200] + 0.37 * torch.randn(1200, 3, generator=torch.Generator().manual_seed(23))
        sc.fill_cells(extra)
        second = [sc.cells[k].cell_pts.clone() for k in keys]
        capped = scene(150, 0.5)
        torch.manual_seed(24)
        capped.fill_cells(gt)
        cap_pts = [capped.cells[k].cell_pts.clone() for k in keys]
        auto = mu.Cell(center=torch.tensor([[0.0, 3.0, 0.0]]), l=torch.tensor(6.0), w=torch.tensor(6.4), h=torch.tensor(6.0),
                       capacity=None, resolution=0.5, device=dev)          # capacity derived from the resolution
        auto2 = mu.Cell(center=torch.tensor([[0.0, 3.0, 0.0]]), l=torch.tensor(6.0), w=torch.tensor(6.4), h=torch.tensor(6.0),
                        capacity=500, resolution=None, device=dev)         # resolution derived from the capacity
        # --- get_cartesian_coords (CustomGeometry.py:5-24) as get_camera_RT calls it (macarons_utils.py:949-952)
        g = torch.Generator().manual_seed(25)
        elev = (torch.rand(64, 1, generator=g) * 180 - 90)
        azim = (torch.rand(64, 1, generator=g) * 720 - 360)
        elev[:5, 0] = torch.tensor([-90.0, 90.0, 0.0, -60.0, 30.0])
        azim[:5, 0] = torch.tensor([0.0, 45.0, 180.0, 315.0, 360.0])
        rays = -get_cartesian_coords(r=torch.ones(64, 1), elev=-1 * elev, azim=180.0 + azim, in_degrees=True)

        def pack(lst):
            return np.concatenate([t.numpy() for t in lst], 0), np.array([len(t) for t in lst])

        p1, n1 = pack(first); p2, n2 = pack(second); p3, n3 = pack(cap_pts)
        np.savez_compressed(os.path.join(HERE, "scene.npz"), verts=v, faces=f, x_min=x_min.numpy(), x_max=x_max.numpy(),
                            areas=areas.numpy(), inside=inside.numpy(), u_face=u_face.numpy()[:, 0],
                            u_alpha=u_alpha.numpy()[:, 0], u_beta=u_beta.numpy()[:, 0], gt=gt.numpy(),
                            cells_of=cells_of.numpy(), cell_keys=np.array([eval(k) for k in keys]),
                            first_pts=p1, first_n=n1, entire_first=entire_first.numpy(), extra=extra.numpy(),
                            second_pts=p2, second_n=n2, cap_pts=p3, cap_n=n3,
                            auto_capacity=auto.capacity, auto_resolution=auto2.resolution,
                            cart_elev=elev.numpy()[:, 0], cart_azim=azim.numpy()[:, 0], cart_rays=rays.numpy())
        print("scene: gt", tuple(gt.shape), "first", n1.tolist(), "second", n2.tolist(), "capped", n3.tolist(),
              "auto", auto.capacity, auto2.resolution)
    finally:
        torch.Tensor.get_device = orig_get_device


def gen_viewstate(mu):
    """View-state vectors of the proxy points: compute_view_state (macarons/utility/scone_utils.py:799-862)
    on random points / camera positions incl. the degenerate directions (straight up / down, x = 0, azimuth +-pi), and
    Scene.update_proxy_view_states (macarons_utils.py:3268-3327) called as the NBV driver calls it
    (macarons/testers/scene.py:598-601: signed distances given, distance_to_surface=None, X_cam=None) on a reference Scene.
    Same shim as gen_scene: Tensor.get_device() -> 'cpu'."""
    import macarons.utility.scone_utils as su
    orig_get_device = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: "cpu"
    try:
        g = torch.Generator().manual_seed(31)
        pts = (torch.rand(4000, 3, generator=g) * 2 - 1) * torch.tensor([40.0, 6.0, 40.0])
        views = (torch.rand(6, 3, generator=g) * 2 - 1) * torch.tensor([30.0, 10.0, 30.0])
        # degenerate rays from the first camera: straight up / down, in the x = 0 plane on both sides, along +-x, equal points
        pts[0] = views[0] + torch.tensor([0.0, -5.0, 0.0]); pts[1] = views[0] + torch.tensor([0.0, 5.0, 0.0])
        pts[2] = views[0] + torch.tensor([0.0, 0.0, -7.0]); pts[3] = views[0] + torch.tensor([0.0, 0.0, 7.0])
        pts[4] = views[0] + torch.tensor([3.0, 0.0, 0.0]); pts[5] = views[0] + torch.tensor([-3.0, 0.0, 0.0])
        pts[6] = views[0] + torch.tensor([0.0, 1.0, 1.0]); pts[7] = views[0] + torch.tensor([-1e-3, 0.0, -9.0])
        vs = su.compute_view_state(pts.view(1, -1, 3), views, 7, 14).view(-1, 98)
        vs1 = su.compute_view_state(pts.view(1, -1, 3), views[:1], 7, 14).view(-1, 98)
        vs_small = su.compute_view_state(pts[:500].view(1, -1, 3), views[:2], 4, 6).view(-1, 24)
        # --- Scene.update_proxy_view_states, twice (the vectors accumulate: heaviside of the sum)
        sc = mu.Scene(x_min=torch.tensor([-20.0, 0.0, -20.0]), x_max=torch.tensor([20.0, 8.0, 20.0]), grid_l=2, grid_w=1, grid_h=2,
                      cell_capacity=100, cell_resolution=0.5, n_proxy_points=3000, device="cpu", feature_dim=0)
        torch.manual_seed(32)
        sc.initialize_proxy_points()
        proxy = sc.proxy_points.clone()
        states, masks, sds, cams = [], [], [], []
        for k in range(2):
            cam = types.SimpleNamespace(X_cam=views[k:k + 1].clone())
            mask = torch.rand(3000, generator=g) < 0.6
            sd = (torch.rand(int(mask.sum()), 1, generator=g) * 2 - 1) * 4 * sc.distance_between_proxy_points
            sc.update_proxy_view_states(cam, mask, signed_distances=sd, distance_to_surface=None, X_cam=None)
            states.append(sc.view_states.clone()); masks.append(mask.clone()); sds.append(sd.view(-1).clone()); cams.append(cam.X_cam[0].clone())
        sd_full = [torch.zeros(3000).masked_scatter(m, s) for m, s in zip(masks, sds)]
        np.savez_compressed(os.path.join(HERE, "viewstate.npz"), pts=pts.numpy(), views=views.numpy(), vs=vs.numpy().astype(np.uint8),
                            vs1=vs1.numpy().astype(np.uint8), vs_small=vs_small.numpy().astype(np.uint8), proxy=proxy.numpy(),
                            dist_between=np.float64(sc.distance_between_proxy_points),
                            upd_mask=np.stack([m.numpy() for m in masks]), upd_sd=np.stack([s.numpy() for s in sd_full]),
                            upd_cam=np.stack([c.numpy() for c in cams]),
                            upd_state=np.stack([s.numpy().astype(np.uint8) for s in states]))
        print("viewstate: set bits", int(vs.sum()), int(vs1.sum()), int(vs_small.sum()), "after updates", [int(s.sum()) for s in states],
              "dist", float(sc.distance_between_proxy_points))
    finally:
        torch.Tensor.get_device = orig_get_device


def gen_camera(ltu, mu):
    """The pure-torch pieces of the simulator rows A13 / A14 that the reference holds itself (no PyTorch3D arithmetic involved):
    the NDC tables and pose lattice of Camera.__init__ (macarons_utils.py:2270-2279, 2283-2327), the mask / keep-count logic of
    compute_partial_point_cloud (:2811-2838; the un-projection itself is PyTorch3D's and is replaced by the identity here, so the
    returned rows are (ndc_x, ndc_y, depth) of the kept pixels), and obtain_depth's outputs and random draws
    (long_term_utils.py:50-155) with use_perfect_depth."""
    H, W = 256, 456
    renderer = types.SimpleNamespace(rasterizer=types.SimpleNamespace(raster_settings=types.SimpleNamespace(image_size=(H, W))))
    x_min, x_max = torch.tensor([-24.0, 0.0, -21.0]), torch.tensor([24.0, 12.0, 21.0])
    cam = mu.Camera(x_min, x_max, 15, 1, 13, 5, 8, 4, 1000.0, renderer, "cpu", contrast_factor=1.0, gathering_factor=0.05)
    keys = list(cam.pose_space.keys())
    poses = torch.stack([cam.pose_space[k] for k in keys]).numpy()
    # ---- compute_partial_point_cloud: identity un-projection, recorded randperm
    g = torch.Generator().manual_seed(77)
    depth = torch.rand(1, H, W, 1, generator=g) * 100.0
    mask = torch.rand(1, H, W, 1, generator=g) < 0.7
    depth[0, :40] = -1.0                                          # "no hit" rows as zbuf has them
    mask = mask & (depth > -1)
    ident = types.SimpleNamespace(unproject_points=lambda pts, scaled_depth_input=False: pts)
    out = {}
    for tag, gf, fov_range in [("a", 0.05, 70.0), ("b", 0.3, None), ("c", 0.05, 5.0)]:
        torch.manual_seed(5)
        pts = cam.compute_partial_point_cloud(depth, mask, fov_cameras=ident, gathering_factor=gf, fov_range=fov_range)
        out[f"ppc_{tag}"] = pts.numpy()
        out[f"ppc_{tag}_args"] = np.array([gf, -1.0 if fov_range is None else fov_range])
    # ---- obtain_depth with perfect depth: what the rollout consumes is (depth, mask); count the numpy draws it makes
    params = types.SimpleNamespace(data_augmentation=False, jitter_probability=0.5, symmetry_probability=0.5, znear=0.5, zfar=1000.0,
                                   n_alpha=2, use_depth_mask=True, pose_factor=1.0, image_height=32, image_width=57, min_depth=0.5, max_depth=1000.0,
                                   height=32, width=57)
    B, h, w = 2, 32, 57
    zb = torch.rand(B, h, w, 1, generator=g) * 60.0
    zb[:, :5] = -1.0
    bmask = zb > -1
    img = torch.rand(B, h, w, 3, generator=g)
    Rm = torch.eye(3).view(1, 3, 3).repeat(B, 1, 1)
    Tm = torch.rand(B, 3, generator=g)
    bd = {"images": img, "mask": bmask, "R": Rm, "T": Tm, "zfar": 1000.0, "zbuf": zb}
    ad = {"images": img.view(B, 1, h, w, 3).repeat(1, 2, 1, 1, 1), "mask": bmask.view(B, 1, h, w, 1).repeat(1, 2, 1, 1, 1),
          "R": Rm.view(B, 1, 3, 3).repeat(1, 2, 1, 1), "T": Tm.view(B, 1, 3).repeat(1, 2, 1), "zfar": 1000.0, "zbuf": zb}
    draws = {"n": 0}
    orig_rand = np.random.rand

    def counting_rand(*a):
        draws["n"] += 1
        return orig_rand(*a)
    # third-party pieces on obtain_depth's way (stubbed modules here): the relative pose goes through PyTorch3D's quaternion
    # helpers -- replaced by zeros, the rollout never reads it (nbp_planning.py:90-92) --, and the error mask pads with
    # torchvision's `pad`, whose reflect mode is torch.nn.functional.pad's
    ltu.convert_matrix_to_pose = lambda params, R, T, aR, aT: torch.zeros(T.shape[0], aT.shape[1], 6)
    ltu.pad = lambda img, padding, padding_mode: torch.nn.functional.pad(img, (padding,) * 4, mode=padding_mode)
    od = {}
    np.random.rand = counting_rand
    try:
        for tag, aug in (("", False), ("_aug", True)):
            params.data_augmentation = aug
            params.jitter_probability = params.symmetry_probability = 0.0        # the draws are made, neither branch is taken
            draws["n"] = 0
            d, m, em, pose, gtp = ltu.obtain_depth(params, bd, ad, "cpu", use_perfect_depth=True)
            od.update({f"od_depth{tag}": d.numpy(), f"od_mask{tag}": m.numpy(), f"od_draws{tag}": np.array(draws["n"])})
        od["od_zbuf"] = zb.numpy()
        od["od_znear_zfar"] = np.array([params.znear, params.zfar], np.float32)
    finally:
        np.random.rand = orig_rand
    np.savez_compressed(os.path.join(HERE, "camera.npz"), ndc_x=cam.ndc_x_tab.numpy(), ndc_y=cam.ndc_y_tab.numpy(),
                        ndc_minmax=np.array([float(cam.min_ndc_x), float(cam.max_ndc_x), float(cam.min_ndc_y), float(cam.max_ndc_y)], np.float32),
                        cam_x_min=cam.x_min.numpy(), pose_keys=np.array(keys), poses=poses, pose_shift=cam.pose_shift.numpy(),
                        dims=np.array([15, 1, 13, 5, 8]), x_min_in=x_min.numpy(),
                        depth=depth.numpy()[0, :, :, 0], mask=mask.numpy()[0, :, :, 0], **out, **od)
    print("camera: ndc_x", float(cam.ndc_x_tab.min()), float(cam.ndc_x_tab.max()), "poses", poses.shape,
          "ppc", {k: v.shape for k, v in out.items() if not k.endswith("args")}, "obtain_depth", {k: getattr(v, "shape", v) for k, v in od.items()})


if __name__ == "__main__":
    model, utils, ltu, mu = import_reference()
    if "--only-camera" in sys.argv:
        gen_camera(ltu, mu)
        sys.exit(0)
    if "--only-viewstate" in sys.argv:
        gen_viewstate(mu)
        sys.exit(0)
    if "--only-scene" in sys.argv:
        gen_scene(mu)
        sys.exit(0)
    if "--only-carve" in sys.argv:
        gen_carve(mu)
        sys.exit(0)
    if "--only-label" in sys.argv:
        gen_obstacle_label(utils)
        sys.exit(0)
    if "--only-blocks" in sys.argv:
        gen_blocks(model)
        sys.exit(0)
    if "--only-train-large" in sys.argv:
        gen_training(model, B=4, S=128, K=40, tag="S128B4")
        sys.exit(0)
    gen_maps(utils)
    gen_planner(ltu, mu)
    gen_replan(utils, ltu, mu)
    gen_scene(mu)
    gen_viewstate(mu)
    gen_camera(ltu, mu)
    gen_network(model)
    gen_training(model)
    gen_training(model, B=4, S=128, K=40, tag="S128B4")
    gen_blocks(model)
    gen_obstacle_label(utils)
    gen_carve(mu)
