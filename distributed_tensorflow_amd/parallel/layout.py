"""Flat variable placement (new layer N2; replaces external X3, TF1's
``replica_device_setter`` that decides which PS task owns which variable).

TF1 places whole variables round-robin on the ``ps`` tasks.  Here the dense
variables of a model are laid out in one flat address space cut into
*buckets* (the unit of one push/pull collective), and every bucket is split
into ``world`` equal, 16-byte-aligned slices: shard ``r`` (the PS state hosted
on rank ``r``) is the concatenation of slice ``r`` of every bucket.  A variable
may straddle two shards; the checkpoint index records that.

Bucket order is reverse registration order, so the bucket holding the last
layers -- whose gradients backward produces first -- is pushed first, and the
bucket holding the first layers is pulled last... i.e. lands right before the
next forward needs it (SURVEY 5.8 "reverse-layer for push").

Sizing for 7 point-to-point xGMI links: a bucket of B bytes puts B/8 on every
link, and per-link messages should be >= 1-4 MB, hence the 32 MB default.

The LAST bucket is the exception (``tail_elems``): it holds the first layers, whose
gradients appear at the very end of backward, so its push -> apply -> pull cannot hide
behind any compute.  With 32 MB buckets ResNet-50's 51 MB of gradients left ~19 MB
exposed every step; the first layers are therefore cut off into their own small bucket
(a few MB: stem + the first stages) and only that is on the critical path.
"""
from __future__ import annotations

from dataclasses import dataclass, field

ALIGN = 8  # elements: 16 B in bf16, 32 B in fp32 -> every slice start is 16-B aligned
# Every VARIABLE starts on a 128-byte line of the flat buffer (64 bf16 elements).  With 16-B alignment one odd-sized
# variable (ResNet's 1000-element classifier bias, first in pack order) shifted every following weight matrix to
# 80 B past a line: the 64-B row segments that the GEMM / convolution kernels stage then straddle two lines, and
# K13's forward calls (weights read in place from this buffer) ran 1.25-1.5x slower than its data-gradient calls
# (flipped filter in a fresh, aligned allocation) on the same problem.
PARAM_ALIGN = 64


def _round_up(x, m):
    return (x + m - 1) // m * m


@dataclass
class ParamSlot:
    name: str
    numel: int
    index: int          # position in the registration-order list
    bucket: int = -1
    flat_off: int = -1  # element offset in the flat space


@dataclass
class Bucket:
    index: int
    start: int = 0      # element offset of the bucket in the flat space
    size: int = 0       # elements, multiple of world * ALIGN
    shard: int = 0      # size // world
    shard_off: int = 0  # offset of this bucket's slice inside a rank's shard
    params: list = field(default_factory=list)  # ParamSlot indices, in pack order
    group: int = 0      # hyper-parameter group (0: decayed, 1: not decayed)


class FlatLayout:
    def __init__(self, named_numels, world, bucket_elems, tail_elems=0):
        """``named_numels``: ``(name, numel)`` or ``(name, numel, group)`` in registration order.  Variables of
        different ``group`` (e.g. weight-decayed / not decayed) never share a bucket, so one shard-apply launch
        has one set of hyper-parameters."""
        self.world = int(world)
        self.slots = []
        groups = []
        for i, item in enumerate(named_numels):
            self.slots.append(ParamSlot(item[0], int(item[1]), i))
            groups.append(int(item[2]) if len(item) > 2 else 0)
        self.buckets = []
        # leading (first-layer) variables that go into the small tail bucket
        n_tail, acc = 0, 0
        if tail_elems and tail_elems < bucket_elems:
            for slot in self.slots:
                acc += _round_up(slot.numel, PARAM_ALIGN)
                if acc > tail_elems:
                    break
                n_tail += 1
            if n_tail == len(self.slots):
                n_tail = 0              # the whole model fits: one ordinary bucket
        open_b = {}                     # group -> [bucket, fill]
        for slot in reversed(self.slots):
            need = _round_up(slot.numel, PARAM_ALIGN)
            g = groups[slot.index]
            if slot.index == n_tail - 1:
                # tail boundary: everything from here on is the exposed tail
                for ob in open_b.values():
                    self._close(*ob)
                open_b = {}
            cur = open_b.get(g)
            if cur is not None and cur[1] and cur[1] + need > bucket_elems:
                self._close(*cur)
                cur = None
            if cur is None:
                cur = open_b[g] = [Bucket(len(self.buckets), group=g), 0]
                self.buckets.append(cur[0])
            slot.bucket = cur[0].index
            slot.flat_off = cur[1]          # relative for now
            cur[0].params.append(slot.index)
            cur[1] += need
        for ob in open_b.values():
            self._close(*ob)
        start = shard_off = 0
        for b in self.buckets:
            b.start, b.shard_off = start, shard_off
            for i in b.params:
                self.slots[i].flat_off += start
            start += b.size
            shard_off += b.shard
        self.total = start
        self.shard_total = shard_off

    def _close(self, b, fill):
        b.size = _round_up(max(fill, 1), self.world * PARAM_ALIGN)      # the next bucket starts on a line too
        b.shard = b.size // self.world

    # -- address translation ------------------------------------------------
    def shard_range(self, bucket, rank):
        """Flat-space element range owned by ``rank`` inside ``bucket``."""
        b = self.buckets[bucket]
        return b.start + rank * b.shard, b.start + (rank + 1) * b.shard

    def owner_pieces(self, slot_index):
        """``[(rank, shard_elem_off, param_elem_off, length), ...]`` describing
        where the elements of a variable live in the ranks' shards."""
        s = self.slots[slot_index]
        b = self.buckets[s.bucket]
        out = []
        pos, end = s.flat_off, s.flat_off + s.numel
        while pos < end:
            rel = pos - b.start
            rank = rel // b.shard
            in_slice = rel - rank * b.shard
            length = min(end - pos, b.shard - in_slice)
            out.append((rank, b.shard_off + in_slice, pos - s.flat_off, length))
            pos += length
        return out

    def describe(self):
        return {
            "world": self.world, "total": self.total, "shard_total": self.shard_total,
            "buckets": [{"start": b.start, "size": b.size, "shard": b.shard, "shard_off": b.shard_off,
                         "params": [self.slots[i].name for i in b.params]} for b in self.buckets],
        }
