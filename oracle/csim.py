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
