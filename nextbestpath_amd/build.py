"""Builds libnbp_hip.so (gfx950 only) in-tree with hipcc.  No torch extension machinery:
the library is a plain C-ABI shared object (include/nbp_hip.h) loaded through ctypes."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libnbp_hip.so")
STAMP = os.path.join(HERE, ".libnbp_hip.stamp")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libnbp_hip.so)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest() -> str:
    h = hashlib.sha256()
    files = sources() + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    files += [os.path.join(INCLUDE, f) for f in sorted(os.listdir(INCLUDE))]
    for f in files:
        h.update(f.encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-I", CSRC]


def _up_to_date(dig: str) -> bool:
    if os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            return fh.read().strip() == dig
    return False


def build(force: bool = False, verbose: bool = True) -> str:
    extra = os.environ.get("NBP_EXTRA_FLAGS", "").split()
    dig = _digest() + " ".join(extra)
    if not force and _up_to_date(dig):
        return LIB
    # one builder at a time: the ranks of a torchrun job import the package concurrently
    import fcntl
    lock = open(os.path.join(HERE, ".build.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and _up_to_date(dig):
            return LIB
        return _build_locked(dig, extra, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(dig, extra, verbose) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc, *FLAGS, *extra, "-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode:
            failed = True
            print(f"[build] FAILED: {src}\n{out}", file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
