"""numpy restatement of the mesh ray queries.  PARITY UNPINNED against trimesh 4.1.2
(third-party, absent): restates ray.intersects_location semantics at the reference's call
sites -- line_segment_mesh_intersection (macarons/utility/macarons_utils.py:120-151: hit iff
some hit distance < segment length) and check_camera_in_mesh
(next_best_path/utility/long_term_utils.py:158-170: hit counts along +Y, +X, +Z all odd).
Same Moller-Trumbore algebra as nextbestpath_amd/csrc/nbp_sim.hip (fp32)."""
import numpy as np

f32 = np.float32


def ray_tri_all(o, d, verts, faces):
    """Distances t (or -1) of ray o + t d with every face; fp32, kernel op order."""
    v = np.asarray(verts, f32)
    a, b, c = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    o, d = np.asarray(o, f32), np.asarray(d, f32)
    e1, e2 = b - a, c - a
    p = np.stack([d[1] * e2[:, 2] - d[2] * e2[:, 1], d[2] * e2[:, 0] - d[0] * e2[:, 2],
                  d[0] * e2[:, 1] - d[1] * e2[:, 0]], 1)
    det = (e1[:, 0] * p[:, 0] + e1[:, 1] * p[:, 1]) + e1[:, 2] * p[:, 2]
    ok = np.abs(det) >= f32(1e-12)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = f32(1) / det
        tv = o[None, :] - a
        u = ((tv[:, 0] * p[:, 0] + tv[:, 1] * p[:, 1]) + tv[:, 2] * p[:, 2]) * inv
        ok &= (u >= 0) & (u <= 1)
        q = np.stack([tv[:, 1] * e1[:, 2] - tv[:, 2] * e1[:, 1], tv[:, 2] * e1[:, 0] - tv[:, 0] * e1[:, 2],
                      tv[:, 0] * e1[:, 1] - tv[:, 1] * e1[:, 0]], 1)
        vv = ((d[0] * q[:, 0] + d[1] * q[:, 1]) + d[2] * q[:, 2]) * inv
        ok &= (vv >= 0) & (u + vv <= 1)
        t = ((e2[:, 0] * q[:, 0] + e2[:, 1] * q[:, 1]) + e2[:, 2] * q[:, 2]) * inv
        ok &= t > 0
    return np.where(ok, t, f32(-1))


def segment_hits_mesh(p0, p1, verts, faces):
    p0, p1 = np.asarray(p0, f32), np.asarray(p1, f32)
    d = p1 - p0
    ln = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2], dtype=f32)
    if not ln > 0:
        return False
    t = ray_tri_all(p0, d / ln, verts, faces)
    return bool(np.any((t >= 0) & (t < ln)))


def axis_ray_counts(p, verts, faces):
    return [int(np.sum(ray_tri_all(p, d, verts, faces) >= 0)) for d in ((0, 1, 0), (1, 0, 0), (0, 0, 1))]


def point_in_mesh(p, verts, faces):
    return all(c % 2 == 1 for c in axis_ray_counts(p, verts, faces))
