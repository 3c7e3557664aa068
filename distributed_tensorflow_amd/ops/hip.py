"""ctypes loader for ``libdtf_hip.so`` (the in-tree gfx950 kernel library).

Policy: on a CUDA/HIP tensor the HIP kernel is *the* path -- if the library is
missing the op raises (never a silent PyTorch fallback on a GPU box).  CPU
tensors take the plain PyTorch reference implementation in each op module,
which is also what the numerics tests compare the kernels against.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DTF_HIP_LIB: another build of the kernel library, e.g. the -DDTF_DEBUG_BOUNDS one (libdtf_hip_debug.so)
LIB_PATH = os.environ.get("DTF_HIP_LIB") or os.path.join(_HERE, "libdtf_hip.so")

DTF_F32, DTF_BF16 = 0, 1
_DT = {torch.float32: DTF_F32, torch.bfloat16: DTF_BF16}

_lib = None
_lock = threading.Lock()

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class KernelLibraryMissing(RuntimeError):
    pass


def _declare(lib):
    P, I, L, F = c_void_p, c_int, c_int64, c_float
    sigs = {
        "dtf_sgd_momentum_apply": [P, I, I, L, P, P, P, I, L, P, P, F, F, F, F, F, I, P],
        "dtf_adam_apply": [P, I, I, L, P, P, P, P, I, L, P, P, F, F, F, F, F, F, F, F, I, I, P],
        "dtf_bucket_pack": [I, P, P, P, P, P, I, F, I, P],
        "dtf_sumsq": [P, I, L, I, L, P, P, I, P],
        "dtf_cast_scale": [P, I, P, I, L, F, P],
        "dtf_abi_version": [],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int
    # optional symbols (other translation units); declared by their op modules
    return lib


def lib():
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise KernelLibraryMissing(
                        "%s not found: run `python -m distributed_tensorflow_amd.ops.build`" % LIB_PATH)
                _lib = _declare(ctypes.CDLL(LIB_PATH))
    return _lib


def available():
    return os.path.exists(LIB_PATH)


def declare(name, argtypes, restype=c_int):
    fn = getattr(lib(), name)
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


def dt(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise TypeError("dtype %s not supported by the HIP kernels (float32 / bfloat16 only)" % (t,))


def ptr(t):
    return 0 if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        if rc == -16:
            raise ValueError("%s: operand not 16-byte aligned" % what)
        if rc == -17:
            raise ValueError("%s: source stride must be a multiple of 8 elements" % what)
        raise RuntimeError("%s failed with code %d" % (what, rc))


def on_gpu(*tensors):
    return any(t is not None and t.is_cuda for t in tensors)
