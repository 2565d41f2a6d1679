"""Builds roitr_amd/lib/libroitr_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

One object per source under csrc/, compiled in parallel, linked into a single C-ABI shared
library with no torch dependency.  Rebuilds only what changed (mtime of source and headers).
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libroitr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize (round 6): hipcc's SLP vectoriser packs neighbouring scalar fp32 operations into v_pk_{mul,add,fma}_f32.  The
# three angle polynomials of the point-pair feature (common.h roitr_ppf4), packed two at a time with their constants in SGPR pairs,
# returned WRONG values for the low half in whole 16-lane groups, a few hundred entries in 16 million, differently from run to run
# (found by the bitwise batch-vs-single tests of tests/test_timed_shape_gpu.py; reproduced on the round-5 build; gone with scalar
# code: scripts history of round 6).  Packed fp32 buys nothing next to MFMAs anyway (MI355X_MICROARCH.md), and
# the kernels that want it (pointops_fps.hip) write it by hand.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         "-I", CSRC, "-I", os.path.join(os.path.dirname(HERE), "include")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            lang = ["-x", "hip"] if s.endswith(".hip") else []
            jobs.append([HIPCC] + FLAGS + lang + ["-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for cmd, r in ex.map(run, jobs):
            if verbose or r.returncode != 0:
                sys.stderr.write(" ".join(cmd[-3:]) + "\n" + r.stdout + r.stderr)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed for " + cmd[-3])
    if jobs or force or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        # -shared links happily with undefined symbols (a kernel whose host stub was not emitted): load it now, in a child
        chk = subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({LIB!r})"], capture_output=True, text=True)
        if chk.returncode != 0:
            os.remove(LIB)
            raise RuntimeError("libroitr_hip.so does not load: " + chk.stderr.strip().splitlines()[-1])
    return LIB


def source_hash():
    """sha256 (first 16 hex digits) over the kernel and ABI sources (csrc/*, include/*.h, in name order): what profiles/*.json are
    stamped with when they are collected, and what bench.py checks before it attaches counters from them to a fresh timing."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h")))
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    if "--source-hash" in sys.argv:
        print(source_hash())
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
