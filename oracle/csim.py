"""ctypes front of oracle/csrc/oracle_sim.c (TEST INFRASTRUCTURE, see oracle/__init__.py): the C twins of
raster.py::raster_zbuf, camera.py::unproject and the inner loop of planner.py::coverage, for rollout-length
checks and bench.py's cpu_baseline leg.  `make -C oracle` builds the library (gcc only; __graft_entry__.build()
runs it); when the .so is missing it is built on first use."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32
_libs = {}


def lib(omp=False):
    name = "liboracle_sim_omp.so" if omp else "liboracle_sim.so"
    if name not in _libs:
        path = os.path.join(HERE, "_build", name)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", HERE])
        L = C.CDLL(path)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.oracle_raster_zbuf.argtypes = [fp, C.c_int, ip, C.c_int, fp, fp, C.c_int, C.c_int, C.c_float, C.c_float,
                                         C.c_float, fp]
        L.oracle_raster_zbuf.restype = None
        L.oracle_raster_zbuf_frames.argtypes = [fp, C.c_int, ip, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, C.c_float, C.c_float,
                                                C.c_float, C.c_int, fp]
        L.oracle_raster_zbuf_frames.restype = None
        L.oracle_accumulate_step_maps.argtypes = [fp, C.c_longlong, C.c_float, C.c_float, fp, C.c_int, C.c_int, C.c_float,
                                                  C.c_float, C.c_int, C.c_float, C.c_float, C.c_int, fp]
        L.oracle_accumulate_step_maps.restype = None
        L.oracle_raster_rgbz.argtypes = [fp, C.c_int, ip, C.c_int, fp, fp, fp, C.c_int, C.c_int, C.c_float, C.c_float,
                                         C.c_float, C.c_float, C.c_float, fp, fp]
        L.oracle_raster_rgbz.restype = None
        L.oracle_unproject.argtypes = [fp, C.c_int, C.c_int, C.c_float, fp, fp, fp]
        L.oracle_unproject.restype = None
        L.oracle_coverage_count.argtypes = [fp, C.c_longlong, fp, C.c_longlong, C.c_float]
        L.oracle_coverage_count.restype = C.c_longlong
        _libs[name] = L
    return _libs[name]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def raster_zbuf(verts, faces, R, T, H, W, tan_half_fov, z_clip=0.5, eps=1e-6):
    v = np.ascontiguousarray(verts, f32)
    f = np.ascontiguousarray(faces, np.int32)
    R = np.ascontiguousarray(R, f32).reshape(9)
    T = np.ascontiguousarray(T, f32).reshape(3)
    out = np.empty((H, W), f32)
    lib().oracle_raster_zbuf(_fp(v), len(v), f.ctypes.data_as(C.POINTER(C.c_int)), len(f), _fp(R), _fp(T), H, W,
                             float(tan_half_fov), float(z_clip), float(eps), _fp(out))
    return out


def raster_zbuf_frames(verts, faces, Rs, Ts, H, W, tan_half_fov, z_clip=0.5, eps=1e-6, band_rows=16, omp=True):
    """[n_frames,H,W] z-buffers; (frame, row band) tasks over the host's threads with omp=True.  == raster_zbuf per frame."""
    v = np.ascontiguousarray(verts, f32)
    f = np.ascontiguousarray(faces, np.int32)
    Rs = np.ascontiguousarray(Rs, f32).reshape(-1, 9)
    Ts = np.ascontiguousarray(Ts, f32).reshape(-1, 3)
    out = np.empty((len(Rs), H, W), f32)
    lib(omp).oracle_raster_zbuf_frames(_fp(v), len(v), f.ctypes.data_as(C.POINTER(C.c_int)), len(f), len(Rs), _fp(Rs), _fp(Ts), H, W,
                                       float(tan_half_fov), float(z_clip), float(eps), int(band_rows), _fp(out))
    return out


def accumulate_step_maps(full_pc, camera_pose, y_bins, S=256, grid_range=(-40, 40), band=0.1, n_pieces=4, omp=True, max_threads=0):
    """== oracle/maps.py::accumulate_step_maps ([6,S,S] fp32 counts), threads over the cloud with omp=True."""
    p = np.ascontiguousarray(full_pc, f32).reshape(-1, 3)
    bounds = np.ascontiguousarray(np.asarray(y_bins, f32)[:-1])
    lo, hi = grid_range
    cy = float(f32(camera_pose[1]))
    out = np.empty((6, S, S), f32)
    lib(omp).oracle_accumulate_step_maps(_fp(p), len(p), float(f32(camera_pose[0])), float(f32(camera_pose[2])), _fp(bounds),
                                         len(bounds), n_pieces, float(f32(cy - band)), float(f32(cy + band)), S, float(f32(lo)),
                                         float(f32(S / (hi - lo))), int(max_threads), _fp(out))
    return out


def raster_rgbz(verts, faces, colors, R, T, H, W, tan_half_fov, ambient=0.85, contrast=1.0, z_clip=0.5, eps=1e-6):
    v = np.ascontiguousarray(verts, f32)
    f = np.ascontiguousarray(faces, np.int32)
    c = np.ascontiguousarray(colors, f32)
    R = np.ascontiguousarray(R, f32).reshape(9)
    T = np.ascontiguousarray(T, f32).reshape(3)
    z, rgb = np.empty((H, W), f32), np.empty((H, W, 3), f32)
    lib().oracle_raster_rgbz(_fp(v), len(v), f.ctypes.data_as(C.POINTER(C.c_int)), len(f), _fp(c), _fp(R), _fp(T), H, W,
                             float(tan_half_fov), float(z_clip), float(eps), float(ambient), float(contrast), _fp(z), _fp(rgb))
    return z, rgb


def unproject(depth, R, T, tan_half_fov):
    d = np.ascontiguousarray(depth, f32)
    H, W = d.shape
    R = np.ascontiguousarray(R, f32).reshape(9)
    T = np.ascontiguousarray(T, f32).reshape(3)
    out = np.empty((H * W, 3), f32)
    lib().oracle_unproject(_fp(d), H, W, float(tan_half_fov), _fp(R), _fp(T), _fp(out))
    return out


def coverage_count(gt, pc, threshold=1.0, omp=False):
    g = np.ascontiguousarray(gt, f32)
    p = np.ascontiguousarray(pc, f32)
    return int(lib(omp).oracle_coverage_count(_fp(g), len(g), _fp(p), len(p), float(threshold)))
