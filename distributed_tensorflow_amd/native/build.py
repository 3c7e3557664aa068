"""Build the native (C++) runtime pieces in-tree: ``dtf-stored``, the RESP2 job-store server.

    python -m distributed_tensorflow_amd.native.build
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
STORED = os.path.join(HERE, "dtf-stored")


def build(force=False, verbose=True):
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
