"""Native RCCL communicator for the PS data path (new layer N1 in C++: ``native/rccl_comm.cpp``; replaces
external X2, the TF1 gRPC mesh on port 2222, ``/root/reference/handlers/train_handler.py:31-35,84,99-100``).

:class:`NativeComm` has the surface of :class:`~.comm.Comm` (``reduce_scatter`` / ``all_gather`` / ``all_to_all`` /
``all_reduce`` / ``broadcast`` / ``reduce`` / ``barrier`` / ``close``), so ``ShardedPS`` and ``ShardedEmbedding`` run on
it unchanged -- but a collective is ONE ctypes call that enqueues RCCL's kernel on the *current HIP stream*: no c10d
``Work`` object, no watchdog thread, no event polling.  Consequences:

* host cost per collective 4.6-11 us instead of 11-18 us (``scripts/comm_host_cost.py``,
  ``profiles/r3_comm_host_cost.txt``: 1.4x-3.9x less);
* a step with its collectives is hipGraph-capturable as is (nothing queries events behind the capture's back);
* teardown is explicit: ``close()`` = ``ncclCommDestroy`` after a barrier, ``abort()`` = ``ncclCommAbort`` for ranks
  whose peers are gone (the launcher's SIGTERM -> SIGKILL escalation stays as the last resort).

Rendezvous: rank 0 creates the 128-byte ``ncclUniqueId`` and publishes it in a ``TCPStore`` on
``MASTER_ADDR:MASTER_PORT`` -- the address of rank 0's headless Service, exactly where the reference's workers find
their PS -- under ``dtf/rccl_id/<generation>``; the other ranks read it and everybody calls ``ncclCommInitRank``.

Select with ``DTF_COMM_BACKEND=rccl-native`` (``bench.py``, ``service/entry.py``) or ``NativeComm.from_env()``.
torch.distributed stays the backend for CPU / gloo jobs, for object exchange and for the dedicated placement's
point-to-point control traffic.
"""
from __future__ import annotations

import ctypes
import datetime
import os

import torch

from .comm import Comm

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DTF_RCCL_LIB") or os.path.normpath(os.path.join(_HERE, "..", "native", "libdtf_rccl.so"))

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.int64: 2, torch.int32: 3, torch.float64: 4, torch.uint8: 5}
_OPS = {"sum": 0, "max": 1, "min": 2, "prod": 3}
_lib = None


class NativeCommError(RuntimeError):
    pass


def lib():
    """``libdtf_rccl.so`` (torch is imported first on purpose: the library then binds to the RCCL / HIP runtime copies
    torch has already loaded instead of pulling second ones into the process)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeCommError("%s not found: run `python -m distributed_tensorflow_amd.native.build`" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        P, I, L64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        PL = ctypes.POINTER(ctypes.c_int64)
        sig = {
            "dtf_rccl_abi_version": [], "dtf_rccl_version": [ctypes.POINTER(I)], "dtf_rccl_unique_id_bytes": [],
            "dtf_rccl_unique_id": [P], "dtf_rccl_init": [ctypes.POINTER(P), P, I, I, I],
            "dtf_rccl_info": [P, ctypes.POINTER(I), ctypes.POINTER(I), ctypes.POINTER(I),
                              ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)],
            "dtf_rccl_destroy": [P], "dtf_rccl_abort": [P], "dtf_rccl_async_error": [P],
            "dtf_rccl_reduce_scatter": [P, P, P, L64, I, P], "dtf_rccl_all_gather": [P, P, P, L64, I, P],
            "dtf_rccl_all_to_all": [P, P, P, L64, I, P],
            "dtf_rccl_all_to_all_v": [P, P, PL, PL, P, PL, PL, I, P],
            "dtf_rccl_all_reduce": [P, P, P, L64, I, I, P], "dtf_rccl_broadcast": [P, P, P, L64, I, I, P],
            "dtf_rccl_reduce": [P, P, P, L64, I, I, P], "dtf_rccl_send": [P, P, L64, I, I, P],
            "dtf_rccl_recv": [P, P, L64, I, I, P],
            "dtf_rccl_pull_and_push": [P, P, P, L64, P, P, L64, I, P],
        }
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.argtypes, fn.restype = args, I
        L.dtf_rccl_last_error.argtypes, L.dtf_rccl_last_error.restype = [], ctypes.c_char_p
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise NativeCommError("%s failed (%d): %s" % (what, rc, lib().dtf_rccl_last_error().decode(errors="replace")))


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError("dtype %s not supported by the native communicator" % t.dtype)


def _op_code(op):
    if op is None:
        return 0
    name = str(op).lower()
    for k, v in _OPS.items():
        if k in name:
            return v
    raise ValueError("reduce op %r not supported" % (op,))


_generation = 0


class NativeComm(Comm):
    backend_name = "rccl-native"

    def __init__(self, rank, world_size, device, handle, store=None):
        super().__init__(rank, world_size, device, None, self.backend_name, live=True)
        self._h = ctypes.c_void_p(handle)
        self._store = store
        self._flag = torch.zeros(1, dtype=torch.float32, device=device)

    # -- construction ---------------------------------------------------------
    @classmethod
    def from_env(cls, device=None, timeout_s=600, store=None):
        """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  A 1-rank communicator is a real
        RCCL communicator too (the whole call path runs on one GPU)."""
        global _generation
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        if device is None:
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        assert device.type == "cuda", "the native communicator is for GPU ranks"
        torch.cuda.set_device(device)
        L = lib()
        n = int(L.dtf_rccl_unique_id_bytes())
        buf = ctypes.create_string_buffer(n)
        gen = _generation
        _generation += 1
        if world > 1:
            if store is None:
                import torch.distributed as dist

                # under torchrun the launcher agent already serves a TCPStore on MASTER_PORT
                # (TORCHELASTIC_USE_AGENT_STORE): every rank is a client then; otherwise rank 0 hosts it
                agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
                store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29533")),
                                      world, (rank == 0) and not agent, timeout=datetime.timedelta(seconds=timeout_s),
                                      wait_for_workers=False)
                if agent:
                    key_prefix = "dtf/%s/" % os.environ.get("TORCHELASTIC_RUN_ID", "run")
                    store = dist.PrefixStore(key_prefix, store)
            key = "dtf/rccl_id/%d" % gen
            if rank == 0:
                _check(L.dtf_rccl_unique_id(buf), "ncclGetUniqueId")
                store.set(key, bytes(buf.raw))
            else:
                raw = store.get(key)
                assert len(raw) == n, "unique id of %d bytes, expected %d" % (len(raw), n)
                buf = ctypes.create_string_buffer(bytes(raw), n)
        else:
            _check(L.dtf_rccl_unique_id(buf), "ncclGetUniqueId")
        h = ctypes.c_void_p()
        _check(L.dtf_rccl_init(ctypes.byref(h), buf, world, rank, device.index or 0), "ncclCommInitRank")
        return cls(rank, world, device, h.value, store)

    def subgroup(self, ranks):
        raise NotImplementedError("sub-communicators are not needed by the colocated PS data path")

    # -- helpers ---------------------------------------------------------------
    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _p(t):
        assert t.is_cuda and t.is_contiguous(), "native collectives take contiguous device tensors"
        return ctypes.c_void_p(t.data_ptr())

    def _done(self, async_op):
        # stream-ordered: there is nothing to wait for on the host.  Callers that asked for a handle get one whose
        # wait() is a no-op (the next kernel on the stream is already ordered behind the collective).
        return _Done() if async_op else None

    # -- collectives ------------------------------------------------------------
    def reduce_scatter(self, out_shard, in_full, async_op=False):
        n = out_shard.numel()
        assert in_full.numel() == n * self.world_size and in_full.dtype == out_shard.dtype
        _check(lib().dtf_rccl_reduce_scatter(self._h, self._p(in_full), self._p(out_shard), n, _dt(in_full),
                                             self._stream()), "reduce_scatter")
        return self._done(async_op)

    def all_gather(self, out_full, in_shard, async_op=False):
        n = in_shard.numel()
        assert out_full.numel() == n * self.world_size and out_full.dtype == in_shard.dtype
        _check(lib().dtf_rccl_all_gather(self._h, self._p(in_shard), self._p(out_full), n, _dt(in_shard),
                                         self._stream()), "all_gather")
        return self._done(async_op)

    def all_to_all(self, out, inp, out_splits=None, in_splits=None, async_op=False):
        W = self.world_size
        if out_splits is None and in_splits is None:
            assert inp.numel() % W == 0 and out.numel() == inp.numel() and out.dtype == inp.dtype
            _check(lib().dtf_rccl_all_to_all(self._h, self._p(inp), self._p(out), inp.numel() // W, _dt(inp),
                                             self._stream()), "all_to_all")
            return self._done(async_op)
        # variable splits (rows of ``width`` elements each): counts / displacements in elements, host arrays
        width = 1
        for d in inp.shape[1:]:
            width *= d
        sc = [int(x) * width for x in (in_splits if in_splits is not None else [inp.shape[0] // W] * W)]
        rc = [int(x) * width for x in (out_splits if out_splits is not None else [out.shape[0] // W] * W)]
        assert sum(sc) == inp.numel() and sum(rc) == out.numel(), "split sizes do not cover the tensors"
        arr = ctypes.c_int64 * W
        sd, rd, a, b = [], [], 0, 0
        for p in range(W):
            sd.append(a)
            rd.append(b)
            a += sc[p]
            b += rc[p]
        # empty tensors have a null data pointer: give RCCL a valid one (nothing is read or written through it)
        src = inp if inp.numel() else self._flag
        dst = out if out.numel() else self._flag
        _check(lib().dtf_rccl_all_to_all_v(self._h, self._p(src), arr(*sc), arr(*sd), self._p(dst), arr(*rc), arr(*rd),
                                           _dt(inp), self._stream()), "all_to_all_v")
        return self._done(async_op)

    def all_reduce(self, t, op=None, async_op=False):
        _check(lib().dtf_rccl_all_reduce(self._h, self._p(t), self._p(t), t.numel(), _dt(t), _op_code(op),
                                         self._stream()), "all_reduce")
        return self._done(async_op)

    def broadcast(self, t, src=0, async_op=False):
        _check(lib().dtf_rccl_broadcast(self._h, self._p(t), self._p(t), t.numel(), _dt(t), int(src), self._stream()),
               "broadcast")
        return self._done(async_op)

    def reduce(self, t, dst=0, async_op=False):
        _check(lib().dtf_rccl_reduce(self._h, self._p(t), self._p(t), t.numel(), _dt(t), int(dst), self._stream()),
               "reduce")
        return self._done(async_op)

    def send(self, t, dst, tag=0):
        _check(lib().dtf_rccl_send(self._h, self._p(t), t.numel(), _dt(t), int(dst), self._stream()), "send")

    def recv(self, t, src=None, tag=0):
        assert src is not None, "the native communicator needs the source rank"
        _check(lib().dtf_rccl_recv(self._h, self._p(t), t.numel(), _dt(t), int(src), self._stream()), "recv")
        return src

    def isend(self, t, dst, tag=0):
        self.send(t, dst, tag)
        return _Done()

    def irecv(self, t, src, tag=0):
        self.recv(t, src, tag)
        return _Done()

    def pull_and_push(self, pull_out_full, pull_in_shard, push_out_shard, push_in_full):
        """The pull of one bucket and the push of the next as ONE grouped RCCL launch (both directions of every xGMI
        link busy at once)."""
        assert pull_out_full.dtype == push_in_full.dtype
        _check(lib().dtf_rccl_pull_and_push(self._h, self._p(pull_in_shard), self._p(pull_out_full), pull_in_shard.numel(),
                                            self._p(push_in_full), self._p(push_out_shard), push_out_shard.numel(),
                                            _dt(push_in_full), self._stream()), "pull_and_push")

    def barrier(self):
        """A 1-element all-reduce + a wait for the stream it ran on: every rank has reached this point."""
        self._flag.zero_()
        self.all_reduce(self._flag)
        torch.cuda.current_stream(self.device).synchronize()

    # -- diagnostics / teardown -----------------------------------------------------
    def stats(self):
        r, n, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        calls, nbytes = ctypes.c_uint64(), ctypes.c_uint64()
        _check(lib().dtf_rccl_info(self._h, ctypes.byref(r), ctypes.byref(n), ctypes.byref(d), ctypes.byref(calls),
                                   ctypes.byref(nbytes)), "info")
        return {"rank": r.value, "nranks": n.value, "device": d.value, "collectives": calls.value, "bytes": nbytes.value}

    def healthy(self):
        """False once RCCL has recorded an asynchronous error (a peer died): the caller should ``abort()``."""
        return lib().dtf_rccl_async_error(self._h) == 0

    def close(self, timeout_s=None, on_timeout_exit=None):
        if not self.live or not self._h:
            return
        timer = None
        if timeout_s and on_timeout_exit is not None:
            import sys
            import threading

            def _bail():
                try:
                    sys.stdout.flush()
                    sys.stderr.flush()
                finally:
                    os._exit(int(on_timeout_exit))

            timer = threading.Timer(float(timeout_s), _bail)
            timer.daemon = True
            timer.start()
        try:
            torch.cuda.synchronize(self.device)
            h, self._h = self._h, None
            _check(lib().dtf_rccl_destroy(h), "ncclCommDestroy")
        finally:
            if timer is not None:
                timer.cancel()
            self.live = False

    def abort(self):
        if self._h:
            h, self._h = self._h, None
            self.live = False
            _check(lib().dtf_rccl_abort(h), "ncclCommAbort")


class _Done:
    """Handle of a stream-ordered collective: already 'complete' from the host's point of view."""

    def wait(self, timeout=None):
        return True

    def is_completed(self):
        return True
