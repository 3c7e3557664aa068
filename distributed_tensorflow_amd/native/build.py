"""Build the native (C++) runtime pieces in-tree: ``dtf-stored``, the RESP2 job-store server, and
``libdtf_rccl.so``, the native RCCL communicator (``rccl_comm.cpp``, used through ``parallel/rccl_native.py``).

    python -m distributed_tensorflow_amd.native.build
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
STORED = os.path.join(HERE, "dtf-stored")


RCCL_LIB = os.path.join(HERE, "libdtf_rccl.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build_rccl(force=False, verbose=True):
    """``libdtf_rccl.so``: host-only C++ against the HIP runtime and RCCL (no device code of its own).  At run time
    it binds to the ``librccl.so.1`` / ``libamdhip64`` the process has already loaded -- torch's bundled copies when
    torch is imported first, which is what ``parallel/rccl_native.py`` does."""
    src = os.path.join(HERE, "rccl_comm.cpp")
    if force or not os.path.exists(RCCL_LIB) or os.path.getmtime(src) > os.path.getmtime(RCCL_LIB):
        cmd = [os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc")), "-O2", "-std=c++17", "-fPIC", "-shared",
               "-Wall", src, "-o", RCCL_LIB, "-L" + os.path.join(ROCM, "lib"), "-lrccl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for rccl_comm.cpp:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[dtf build] built", RCCL_LIB, flush=True)
    return RCCL_LIB


def build(force=False, verbose=True):
    build_rccl(force, verbose)
    src = os.path.join(HERE, "store_server.cpp")
    if force or not os.path.exists(STORED) or os.path.getmtime(src) > os.path.getmtime(STORED):
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-o", STORED, src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[dtf build] built", STORED, flush=True)
    return STORED


def spawn_store(port=0, host="127.0.0.1"):
    """Start ``dtf-stored``; returns ``(Popen, port)`` once it is listening."""
    exe = os.environ.get("DTF_STORED_EXE") or build(verbose=False)     # e.g. the ASan/UBSan variant
    p = subprocess.Popen([exe, "--host", host, "--port", str(port)], stdout=subprocess.PIPE, text=True)
    line = p.stdout.readline()
    if "listening" not in line:
        p.kill()
        raise RuntimeError("dtf-stored did not start: %r" % line)
    return p, int(line.rsplit(":", 1)[1])


if __name__ == "__main__":
    build(force="--force" in sys.argv)
