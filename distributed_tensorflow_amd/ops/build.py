"""Build the gfx950 HIP kernel library in-tree.

    python -m distributed_tensorflow_amd.ops.build          # all objects + link
    python -m distributed_tensorflow_amd.ops.build --force

Every ``csrc/*.hip`` translation unit is compiled with
``hipcc --offload-arch=gfx950`` into ``ops/_build/*.o`` and linked into
``ops/libdtf_hip.so`` (a plain C-ABI shared library loaded through ctypes: no
torch headers, so a unit compiles in seconds and the library does not depend on
the torch ABI).  The ``.so`` is git-ignored but travels with ``gpurun``
snapshots; nothing is JIT-compiled at job submit time.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libdtf_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=fast",
         "-Wall", "-Wno-unused-function"]


# per-file additions: the attention kernels take row maxima of finite scores; without nnan every fmaxf
# input is first canonicalised with its own v_max_f32 (196 extra VALU instructions per 128 scores)
EXTRA_FLAGS = {"attn_kernels.hip": ["-fno-honor-nans"]}


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src,) + tuple(extra))


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    return tuple(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))


def build(force=False, verbose=True, save_temps=False, debug_bounds=False):
    """``debug_bounds``: compile with -DDTF_DEBUG_BOUNDS into ``libdtf_hip_debug.so`` (own object directory); run a
    process against it with ``DTF_HIP_LIB=<path>`` -- see ``scripts/gpu_debug_bounds.sh``."""
    global OBJ, LIB
    if debug_bounds:
        OBJ, LIB = os.path.join(HERE, "_build_debug"), os.path.join(HERE, "libdtf_hip_debug.so")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = headers()
    jobs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        if force or _newer(src, obj, hdrs):
            cmd = [HIPCC] + FLAGS + (["-DDTF_DEBUG_BOUNDS"] if debug_bounds else []) + \
                EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
            if save_temps:
                cmd += ["-save-temps=obj"]
            jobs.append((src, cmd))

    def run(job):
        src, cmd = job
        if verbose:
            print("[dtf build] hipcc", os.path.basename(src), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in sources()]
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[dtf build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, debug_bounds="--debug-bounds" in sys.argv)
