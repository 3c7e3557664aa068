"""Multi-tensor bucket pack / unpack (kernel K5).

The TF1 runtime behind the reference moves every variable as its own
``RecvTensor`` message (external X2).  Here gradients are flattened into a few
large wire-dtype buckets so that one RCCL reduce-scatter moves MBs per xGMI
link; the pack is one launch per <=96 tensors with the descriptor table in the
kernel arguments.  ``PackPlan`` holds the static part (offsets, sizes); only
the tensors' current addresses are collected per call.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import hip


class PackPlan:
    """Static layout of one bucket: tensor ``i`` occupies
    ``flat[flat_off[i] : flat_off[i] + numel[i]]`` in raw storage order."""

    def __init__(self, numels, flat_offs):
        self.n = len(numels)
        self.numel = np.ascontiguousarray(numels, dtype=np.int64)
        self.flat_off = np.ascontiguousarray(flat_offs, dtype=np.int64)
        self._ptrs = np.zeros(self.n, dtype=np.uint64)
        self._dts = np.zeros(self.n, dtype=np.int32)

    def _fill(self, tensors):
        assert len(tensors) == self.n
        for i, t in enumerate(tensors):
            self._ptrs[i] = t.data_ptr()
            self._dts[i] = hip.dt(t)

    def _launch(self, tensors, flat, scale, unpack):
        self._fill(tensors)
        rc = hip.lib().dtf_bucket_pack(
            self.n, self._ptrs.ctypes.data_as(ctypes.c_void_p), self.numel.ctypes.data_as(ctypes.c_void_p),
            self.flat_off.ctypes.data_as(ctypes.c_void_p), self._dts.ctypes.data_as(ctypes.c_void_p),
            hip.ptr(flat), hip.dt(flat), float(scale), int(unpack), hip.stream())
        hip.check(rc, "dtf_bucket_pack")

    def pack(self, tensors, flat, scale=1.0):
        """``flat[off_i : off_i+n_i] = storage(tensors[i]) * scale`` (cast to flat's dtype).
        Every tensor must be dense (its storage order is what gets copied)."""
        if not flat.is_cuda:
            for t, n, off in zip(tensors, self.numel, self.flat_off):
                flat[off:off + n].copy_(dense_storage_view(t).float() * scale)
            return
        self._launch(tensors, flat, scale, 0)

    def unpack(self, flat, tensors):
        if not flat.is_cuda:
            for t, n, off in zip(tensors, self.numel, self.flat_off):
                dense_storage_view(t).copy_(flat[off:off + n])
            return
        self._launch(tensors, flat, 1.0, 1)


def dense_storage_view(t):
    """1-D view of a dense (possibly permuted, e.g. channels_last) tensor in
    storage order."""
    if t.is_contiguous():
        return t.view(-1)
    assert is_dense(t), "tensor is not non-overlapping and dense"
    return torch.as_strided(t, (t.numel(),), (1,), t.storage_offset())


def is_dense(t):
    if t.numel() == 0 or t.is_contiguous():
        return True
    sizes_strides = sorted(((st, sz) for sz, st in zip(t.shape, t.stride()) if sz != 1))
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            return False
        expect *= sz
    return True
