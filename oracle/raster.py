"""numpy restatement of the z-buffer rasteriser (brute force: every pixel x every face).
PARITY UNPINNED against pytorch3d 0.7.4 MeshRasterizer (third-party, absent): restates its
documented semantics at the reference's call site (macarons/utility/macarons_utils.py:905-937,
2743-2786: image_size=(256,456), blur_radius=0, faces_per_pixel=1, perspective-correct z,
z_clip = znear/2 = 0.5): zbuf = view-space z of the nearest face through the pixel centre,
-1 for background; pixel centres at ndc_x = (W-(2c+1))/s, ndc_y = (H-(2r+1))/s, s = min(H,W).
Same ray/triangle algebra as nextbestpath_amd/csrc/nbp_sim.hip (fp32): per-face plane forms (plane_forms), bit for bit."""
import numpy as np

f32 = np.float32


def to_view(verts, R, T):
    v = np.asarray(verts, f32)
    R = np.asarray(R, f32)
    out = np.empty_like(v)
    for j in range(3):
        out[:, j] = ((v[:, 0] * R[0, j] + v[:, 1] * R[1, j]) + v[:, 2] * R[2, j]) + f32(T[j])
    return out


def plane_forms(v0, e1, e2):
    """The ray (dx, dy, 1) through a pixel centre hits the triangle (v0, v0 + e1, v0 + e2) where det = e1 . (d x e2), u = -(v0 . (d x e2)) /
    det, v = (d . (e1 x v0)) / det: all three numerators are LINEAR in (dx, dy), so a face is three coefficient triples (fp32 cross
    products, rounded once per face) and a pixel costs three linear forms and one reciprocal.  -> (det coefficients a, u-numerator
    coefficients un); the v numerator's are q = e1 x v0 as before."""
    a = np.array([e1[2] * e2[1] - e1[1] * e2[2], e1[0] * e2[2] - e1[2] * e2[0], e1[1] * e2[0] - e1[0] * e2[1]], f32)
    un = np.array([v0[1] * e2[2] - v0[2] * e2[1], v0[2] * e2[0] - v0[0] * e2[2], v0[0] * e2[1] - v0[1] * e2[0]], f32)
    return a, un


def raster_zbuf(verts, faces, R, T, H, W, tan_half_fov, z_clip=0.5, eps=1e-6):
    vv = to_view(verts, R, T)
    s = min(H, W)
    col = np.arange(W, dtype=f32)[None, :]
    row = np.arange(H, dtype=f32)[:, None]
    dx = ((f32(W) - (f32(2) * col + f32(1))) / f32(s) * f32(tan_half_fov)) + np.zeros((H, 1), f32)
    dy = ((f32(H) - (f32(2) * row + f32(1))) / f32(s) * f32(tan_half_fov)) + np.zeros((1, W), f32)
    zb = np.full((H, W), 3.0e38, f32)
    zc = f32(z_clip)
    for f in np.asarray(faces):
        v0, v1, v2 = vv[f[0]], vv[f[1]], vv[f[2]]
        if v0[2] <= zc and v1[2] <= zc and v2[2] <= zc:
            continue
        e1, e2 = v1 - v0, v2 - v0
        q = np.array([e1[1] * v0[2] - e1[2] * v0[1], e1[2] * v0[0] - e1[0] * v0[2], e1[0] * v0[1] - e1[1] * v0[0]], f32)
        tnum = (e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]
        a, un = plane_forms(v0, e1, e2)
        det = (a[0] * dx + a[1] * dy) + a[2]
        ok = np.abs(det) >= f32(1e-12)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            inv = f32(1) / det
            u = ((un[0] * dx + un[1] * dy) + un[2]) * inv
            v = ((dx * q[0] + dy * q[1]) + q[2]) * inv
            z = tnum * inv
            hit = ok & (u >= -f32(eps)) & (v >= -f32(eps)) & (u + v <= f32(1) + f32(eps)) & (z > zc) & (z < zb)
        zb = np.where(hit, z, zb)
    return np.where(zb < 1.0e38, zb, f32(-1)).astype(f32)


def raster_rgbz(verts, faces, colors, R, T, H, W, tan_half_fov, ambient=0.85, contrast=1.0, z_clip=0.5, eps=1e-6):
    """Depth and colours (Camera.capture_image, mu:2743-2763; PARITY UNPINNED like the depth): SoftPhongShader under
    AmbientLights on a TexturesVertex mesh = ambient x barycentric interpolation of the winning face's vertex colours
    (perspective-correct barycentrics = those of the view-space ray cast), white background; the soft-blend terms of
    softmax_rgb_blend vanish for faces_per_pixel=1, sigma=gamma=1e-4 and z <= sensor range; torchvision adjust_contrast.
    Equal depths: the lowest face index wins (first strict improvement in index order).  -> (zbuf, rgb [H,W,3])."""
    vv = to_view(verts, R, T)
    s = min(H, W)
    col = np.arange(W, dtype=f32)[None, :]
    row = np.arange(H, dtype=f32)[:, None]
    dx = ((f32(W) - (f32(2) * col + f32(1))) / f32(s) * f32(tan_half_fov)) + np.zeros((H, 1), f32)
    dy = ((f32(H) - (f32(2) * row + f32(1))) / f32(s) * f32(tan_half_fov)) + np.zeros((1, W), f32)
    zb = np.full((H, W), 3.0e38, f32)
    ub, vb = np.zeros((H, W), f32), np.zeros((H, W), f32)
    fb = np.full((H, W), -1, np.int64)
    zc = f32(z_clip)
    for fi, f in enumerate(np.asarray(faces)):
        v0, v1, v2 = vv[f[0]], vv[f[1]], vv[f[2]]
        if v0[2] <= zc and v1[2] <= zc and v2[2] <= zc:
            continue
        e1, e2 = v1 - v0, v2 - v0
        q = np.array([e1[1] * v0[2] - e1[2] * v0[1], e1[2] * v0[0] - e1[0] * v0[2], e1[0] * v0[1] - e1[1] * v0[0]], f32)
        tnum = (e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]
        a, un = plane_forms(v0, e1, e2)
        det = (a[0] * dx + a[1] * dy) + a[2]
        ok = np.abs(det) >= f32(1e-12)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            inv = f32(1) / det
            u = ((un[0] * dx + un[1] * dy) + un[2]) * inv
            v = ((dx * q[0] + dy * q[1]) + q[2]) * inv
            z = tnum * inv
            hit = ok & (u >= -f32(eps)) & (v >= -f32(eps)) & (u + v <= f32(1) + f32(eps)) & (z > zc) & (z < zb)
        zb = np.where(hit, z, zb)
        ub, vb = np.where(hit, u, ub), np.where(hit, v, vb)
        fb = np.where(hit, fi, fb)
    have = fb >= 0
    fa = np.asarray(faces)[np.maximum(fb, 0)]
    c = np.asarray(colors, f32)
    w0 = (f32(1) - ub) - vb
    rgb = np.ones((H, W, 3), f32)
    for k in range(3):
        tex = (w0 * c[fa[..., 0], k] + ub * c[fa[..., 1], k]) + vb * c[fa[..., 2], k]
        rgb[..., k] = np.where(have, f32(ambient) * tex, f32(1))
    if contrast != 1.0:
        gray = (f32(0.299) * rgb[..., 0] + f32(0.587) * rgb[..., 1]) + f32(0.114) * rgb[..., 2]
        mean = f32(gray.astype(np.float64).sum() / (H * W))
        rgb = np.clip(f32(contrast) * rgb + (f32(1) - f32(contrast)) * mean, 0, 1).astype(f32)
    return np.where(zb < 1.0e38, zb, f32(-1)).astype(f32), rgb
