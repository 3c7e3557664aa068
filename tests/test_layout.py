"""Flat variable placement (parallel/layout.py): buckets, shards, the small tail bucket."""
import random

from distributed_tensorflow_amd.parallel.layout import ALIGN, PARAM_ALIGN, FlatLayout


def _check_invariants(L, named, world):
    # every variable lives in exactly one bucket, offsets do not overlap, shards tile the flat space
    spans = sorted((s.flat_off, s.flat_off + s.numel) for s in L.slots)
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0
    assert spans[-1][1] <= L.total
    assert sum(b.size for b in L.buckets) == L.total and sum(b.shard for b in L.buckets) == L.shard_total
    for b in L.buckets:
        assert b.size % (world * ALIGN) == 0 and b.shard * world == b.size
        assert b.start % PARAM_ALIGN == 0
        for i in b.params:
            assert L.slots[i].flat_off % PARAM_ALIGN == 0          # every variable starts on a 128-byte line (bf16)
            assert b.start <= L.slots[i].flat_off and L.slots[i].flat_off + L.slots[i].numel <= b.start + b.size
    for s in L.slots:
        pieces = L.owner_pieces(s.index)
        assert sum(p[3] for p in pieces) == s.numel
        assert [p[2] for p in pieces] == sorted(p[2] for p in pieces)


def test_buckets_are_in_reverse_registration_order():
    named = [("v%d" % i, 100 + i) for i in range(20)]
    L = FlatLayout(named, 4, 600)
    _check_invariants(L, named, 4)
    assert L.slots[L.buckets[0].params[0]].name == "v19"
    assert L.slots[L.buckets[-1].params[-1]].name == "v0"
    assert len(L.buckets) > 2


def test_tail_bucket_holds_only_the_first_layers():
    named = [("stem", 500), ("l1", 2000), ("l2", 6000), ("l3", 40000), ("l4", 90000), ("fc", 20000)]
    plain = FlatLayout(named, 8, 1 << 20)
    assert len(plain.buckets) == 1
    L = FlatLayout(named, 8, 1 << 20, tail_elems=10000)
    _check_invariants(L, named, 8)
    assert len(L.buckets) == 2
    assert [L.slots[i].name for i in L.buckets[-1].params] == ["l2", "l1", "stem"]
    assert [L.slots[i].name for i in L.buckets[0].params] == ["fc", "l4", "l3"]
    # the whole model below the cap, or a first variable above it: no special bucket
    assert len(FlatLayout(named, 8, 1 << 20, tail_elems=1 << 19).buckets) == 1
    assert len(FlatLayout(named, 8, 1 << 20, tail_elems=100).buckets) == 1


def test_random_layouts_keep_invariants():
    rng = random.Random(0)
    for _ in range(50):
        named = [("p%d" % i, rng.randint(1, 5000)) for i in range(rng.randint(1, 40))]
        world = rng.choice([1, 2, 3, 8])
        L = FlatLayout(named, world, rng.randint(100, 20000), tail_elems=rng.choice([0, 50, 3000, 100000]))
        _check_invariants(L, named, world)
