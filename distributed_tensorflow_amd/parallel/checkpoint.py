"""Per-shard checkpoints with TF1 variable naming (replaces
external X6, ``tf.train.Saver`` driven by the job's ``export`` field --
argv[7] of the reference's entry command, ``handlers/train_handler.py:83``).

Files under ``<export>/`` (names follow ``tf.train.Saver``):

``model.ckpt-<step>.data-<rank:05d>-of-<W:05d>``
    what rank ``rank`` hosts: a self-describing container
    ``[u64 header_len][JSON header][raw sections]``; the header maps section
    name -> (byte offset, byte length, dtype).  Sections of a dense PS shard are
    ``""`` (fp32 master weights) and one per optimizer slot (``Momentum`` /
    ``Adam`` / ``Adam_1``); sharded embedding tables add ``<table>/part_<k>``.
    Every rank writes its own file in parallel -- nothing is gathered.
``model.ckpt-<step>.index``
    JSON written by the chief: variable name -> dtype, shape and the list of
    pieces ``(rank, section, element offset, length, offset inside the
    variable)`` -- a variable may straddle shards.  Optimizer slots appear as
    ``<var>/Momentum``, ``<var>/Adam``, ``<var>/Adam_1`` like TF slot variables.
    The shard bytes are the parameter's STORAGE order; ``shape`` is the logical
    shape and ``storage_order`` (present when it is not the identity) lists the
    logical dimensions from slowest to fastest varying -- a channels_last conv
    filter ``[O, I, H, W]`` is stored ``O, H, W, I`` = ``[0, 2, 3, 1]``.
    :meth:`CheckpointReader.read` returns the logical tensor; ``restore_ps``
    converts between layouts, so a checkpoint written by an NHWC GPU job
    restores into a contiguous CPU model and vice versa.  One torch variable
    may appear as several TF variables (BERT's packed ``qkv`` ->
    ``query`` / ``key`` / ``value``): row ranges of the same bytes.
``model.ckpt-<step>.replicated``
    non-PS state of the chief (``global_step``, BatchNorm moving statistics).
``checkpoint``
    ``model_checkpoint_path: "model.ckpt-<step>"`` (TF's CheckpointState text).

Restore re-shards: pieces are read by variable and scattered into whatever
layout (world size, bucket size) the restoring job uses.
"""
from __future__ import annotations

import json
import os
import struct

import numpy as np
import torch

_NP = {"float32": np.float32, "bfloat16": np.uint16, "int64": np.int64, "float16": np.float16}
PREFIX = "model.ckpt"


def data_path(export_dir, step, rank, world):
    return os.path.join(export_dir, "%s-%d.data-%05d-of-%05d" % (PREFIX, step, rank, world))


def index_path(export_dir, step):
    return os.path.join(export_dir, "%s-%d.index" % (PREFIX, step))


def _dtype_name(t):
    return str(t.dtype).replace("torch.", "")


def _to_numpy(t):
    t = t.detach().contiguous().cpu()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def write_shard(export_dir, step, rank, world, sections):
    """``sections``: ordered ``{name: tensor}``; atomic (tmp + rename)."""
    os.makedirs(export_dir, exist_ok=True)
    header, off = {}, 0
    arrays = []
    for name, t in sections.items():
        a = _to_numpy(t).reshape(-1)
        header[name] = {"offset": off, "nbytes": int(a.nbytes), "dtype": _dtype_name(t)}
        arrays.append(a)
        off += (a.nbytes + 63) // 64 * 64
    hj = json.dumps({"sections": header, "rank": rank, "world": world, "step": step}).encode()
    path = data_path(export_dir, step, rank, world)
    tmp = path + ".tmp.%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        base = f.tell()
        for name, a in zip(header, arrays):
            f.seek(base + header[name]["offset"])
            a.tofile(f)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)
    return path


class ShardReader:
    def __init__(self, path):
        self.path = path
        with open(path, "rb") as f:
            (n,) = struct.unpack("<Q", f.read(8))
            self.header = json.loads(f.read(n).decode())
            self.base = 8 + n
        self._mm = np.memmap(path, mode="r")

    def read(self, section, elem_off, length):
        s = self.header["sections"][section]
        dt = np.dtype(_NP[s["dtype"]])
        start = self.base + s["offset"] + elem_off * dt.itemsize
        raw = self._mm[start:start + length * dt.itemsize]
        a = np.frombuffer(raw, dtype=dt)
        t = torch.from_numpy(a.copy())
        if s["dtype"] == "bfloat16":
            t = t.view(torch.int16).view(torch.bfloat16)
        return t


def storage_order(t):
    """Logical dimensions of a dense tensor from slowest to fastest varying in its storage
    (identity for a contiguous tensor; ``[0, 2, 3, 1]`` for a channels_last 4-D tensor)."""
    if t.dim() <= 1 or t.is_contiguous():
        return list(range(t.dim()))
    # stable sort by decreasing stride; size-1 dimensions keep their logical position among equals
    return sorted(range(t.dim()), key=lambda d: (-t.stride(d), d))


def _segments(tf_names, slot_name, shape, numel):
    """``[(variable name, first element, end element, shape)]`` of one torch variable: itself, or the row ranges
    ``tf_names`` splits it into (``[(tf_name, row0, row1), ...]``)."""
    spec = tf_names.get(slot_name, slot_name)
    if isinstance(spec, str):
        return [(spec, 0, numel, list(shape))]
    inner = numel // shape[0] if shape and shape[0] else 1
    return [(n, r0 * inner, r1 * inner, [r1 - r0] + list(shape[1:])) for n, r0, r1 in spec]


def dense_variable_index(layout, tf_names, shapes, slot_names, orders=None):
    """Index entries of the dense PS variables of a :class:`FlatLayout`."""
    out = {}
    orders = orders or {}
    for slot in layout.slots:
        shape = list(shapes[slot.name])
        pieces = layout.owner_pieces(slot.index)
        order = orders.get(slot.name)
        identity = order is None or list(order) == list(range(len(shape)))
        segs = _segments(tf_names, slot.name, shape, slot.numel)
        assert identity or len(segs) == 1, "a variable split into several names must be stored contiguously"
        for name, lo, hi, sshape in segs:
            for sec in ("",) + tuple(slot_names):
                ent = {"dtype": "float32", "shape": sshape, "torch_name": slot.name, "pieces": []}
                if not identity:
                    ent["storage_order"] = list(order)
                for r, so, vo, ln in pieces:
                    a, b = max(vo, lo), min(vo + ln, hi)
                    if a < b:
                        ent["pieces"].append({"rank": r, "section": sec, "offset": so + (a - vo), "length": b - a,
                                              "var_offset": a - lo})
                out[name if sec == "" else "%s/%s" % (name, sec)] = ent
    return out


def write_index(export_dir, step, world, variables, extra=None):
    os.makedirs(export_dir, exist_ok=True)
    doc = {"format": "dtf-ckpt-1", "step": step, "world": world, "variables": variables}
    if extra:
        doc.update(extra)
    p = index_path(export_dir, step)
    tmp = p + ".tmp"
    with open(tmp, "w") as f:
        json.dump(doc, f)
    os.replace(tmp, p)
    kept = prune(export_dir, KEEP, newest=step)
    with open(os.path.join(export_dir, "checkpoint.tmp"), "w") as f:
        f.write('model_checkpoint_path: "%s-%d"\n' % (PREFIX, step))
        for s_ in kept:
            f.write('all_model_checkpoint_paths: "%s-%d"\n' % (PREFIX, s_))
    os.replace(os.path.join(export_dir, "checkpoint.tmp"), os.path.join(export_dir, "checkpoint"))
    return p


# How many complete checkpoints stay in an export directory (TF1 Saver's max_to_keep, default 5; 0 = all of them): a job
# that checkpoints every N steps would otherwise fill the disk (12 checkpoints of ResNet-50 = 2.3 GB).  Set from the
# job option `keep_checkpoints` by the trainer.
KEEP = int(os.environ.get("DTF_KEEP_CHECKPOINTS", "5"))


def checkpoint_steps(export_dir):
    """Steps that have an index file (= complete checkpoints), ascending."""
    out = []
    pre, suf = PREFIX + "-", ".index"
    for f in os.listdir(export_dir) if os.path.isdir(export_dir) else []:
        if f.startswith(pre) and f.endswith(suf):
            try:
                out.append(int(f[len(pre):-len(suf)]))
            except ValueError:
                pass
    return sorted(out)


def prune(export_dir, keep, newest=None):
    """Delete all files of the checkpoints older than the ``keep`` newest complete ones (never ``newest``, never files of
    a step that has no index yet: its shards are still being written).  Called by whoever publishes the index (the
    chief), after the new index is in place.  -> the steps that remain, ascending."""
    steps = checkpoint_steps(export_dir)
    if keep <= 0 or len(steps) <= keep:
        return steps
    drop = [s_ for s_ in steps[:-keep] if s_ != newest]
    for s_ in drop:
        pre = "%s-%d." % (PREFIX, s_)
        for f in os.listdir(export_dir):
            if f.startswith(pre):
                try:
                    os.remove(os.path.join(export_dir, f))
                except OSError:
                    pass
    return [s_ for s_ in steps if s_ not in drop]


def write_replicated(export_dir, step, replicated):
    """Non-PS state of the chief (``global_step``, BatchNorm moving statistics): atomic write."""
    os.makedirs(export_dir, exist_ok=True)
    rp = os.path.join(export_dir, "%s-%d.replicated" % (PREFIX, step))
    torch.save({k: (v.detach().cpu() if isinstance(v, torch.Tensor) else torch.tensor(v))
                for k, v in replicated.items()}, rp + ".tmp")
    os.replace(rp + ".tmp", rp)
    return rp


def latest_step(export_dir):
    p = os.path.join(export_dir, "checkpoint")
    if not os.path.isfile(p):
        return None
    with open(p) as f:
        for line in f:
            if line.startswith("model_checkpoint_path"):
                name = line.split('"')[1]
                return int(name.rsplit("-", 1)[1])
    return None


def read_index(export_dir, step=None):
    step = latest_step(export_dir) if step is None else step
    if step is None:
        return None
    with open(index_path(export_dir, step)) as f:
        return json.load(f)


class CheckpointReader:
    """Random access to variables of a checkpoint, whatever its sharding."""

    def __init__(self, export_dir, step=None):
        self.dir = export_dir
        self.index = read_index(export_dir, step)
        if self.index is None:
            raise FileNotFoundError("no checkpoint under %s" % export_dir)
        self.step, self.world = self.index["step"], self.index["world"]
        self._readers = {}

    def _reader(self, rank):
        r = self._readers.get(rank)
        if r is None:
            r = ShardReader(data_path(self.dir, self.step, rank, self.world))
            self._readers[rank] = r
        return r

    def names(self):
        return sorted(self.index["variables"])

    def has(self, name):
        return name in self.index["variables"]

    def read_flat(self, name, start=0, length=None):
        """Elements ``[start, start+length)`` of a variable (flattened in storage order)."""
        v = self.index["variables"][name]
        total = int(np.prod(v["shape"])) if v["shape"] else 1
        length = total - start if length is None else length
        out = None
        for pc in v["pieces"]:
            lo = max(start, pc["var_offset"])
            hi = min(start + length, pc["var_offset"] + pc["length"])
            if lo >= hi:
                continue
            t = self._reader(pc["rank"]).read(pc["section"], pc["offset"] + (lo - pc["var_offset"]), hi - lo)
            if out is None:
                out = torch.empty(length, dtype=t.dtype)
            out[lo - start:hi - start] = t
        return out

    def read(self, name):
        """The variable as a LOGICAL tensor of its recorded shape (whatever order its bytes are stored in)."""
        v = self.index["variables"][name]
        if not v["shape"]:
            return self.read_flat(name)[0]
        order = v.get("storage_order")
        if not order or order == list(range(len(v["shape"]))):
            return self.read_flat(name).reshape(v["shape"])
        stored = self.read_flat(name).reshape([v["shape"][d] for d in order])
        inverse = [order.index(d) for d in range(len(order))]
        return stored.permute(inverse)

    def read_storage(self, name, order):
        """The variable flattened in the storage order ``order`` (converted when the file's differs)."""
        v = self.index["variables"][name]
        have = v.get("storage_order") or list(range(len(v["shape"])))
        if list(order) == list(have):
            return self.read_flat(name)
        return self.read(name).permute(list(order)).contiguous().reshape(-1)

    def replicated(self):
        p = os.path.join(self.dir, "%s-%d.replicated" % (PREFIX, self.step))
        if not os.path.isfile(p):
            return {}
        return torch.load(p, map_location="cpu", weights_only=True)


# ---------------------------------------------------------------------------
# ShardedPS <-> checkpoint
# ---------------------------------------------------------------------------

def _ps_sections(ps):
    sec = {"": ps.master}
    slots = ps.opt.slots
    if len(slots) >= 1 and ps.slot_m is not None:
        sec[slots[0]] = ps.slot_m
    if len(slots) >= 2 and ps.slot_v is not None:
        sec[slots[1]] = ps.slot_v
    return sec


def save_ps(ps, export_dir, step=None, tf_names=None, replicated=None, rank=None, world=None, comm=None,
            extra_sections=None, extra_variables=None):
    """Every rank calls this; the chief (rank 0) also writes index + replicated."""
    comm = comm or ps.comm
    rank = comm.rank if rank is None else rank
    world = comm.world_size if world is None else world
    step = ps.global_step if step is None else step
    if hasattr(ps, "flush"):
        ps.flush()
    if ps.master.is_cuda:
        torch.cuda.synchronize(ps.master.device)
    sections = _ps_sections(ps)
    if extra_sections:
        sections.update(extra_sections)
    write_shard(export_dir, step, rank, world, sections)
    comm.barrier()
    if rank == 0:
        shapes = {n: tuple(p.shape) for n, p in ps.named_params}
        orders = {n: storage_order(p) for n, p in ps.named_params}
        variables = dense_variable_index(ps.layout, tf_names or {}, shapes, ps.opt.slots, orders)
        if extra_variables:
            variables.update(extra_variables)
        if replicated is not None:
            write_replicated(export_dir, step, replicated)
        write_index(export_dir, step, world, variables,
                    extra={"optimizer": ps.opt.kind, "layout": {"total": ps.layout.total,
                                                                "buckets": len(ps.layout.buckets)}})
    comm.barrier()
    return step


def restore_ps(ps, export_dir, step=None, tf_names=None):
    """Load master weights + optimizer slots into ``ps`` (any world size, bucket size or memory format), and
    refresh the wire-dtype parameters the model reads.  Returns the reader."""
    rd = CheckpointReader(export_dir, step)
    tf_names = tf_names or {}
    L, r = ps.layout, ps.comm.rank
    hosted = getattr(ps, "hosted", True)
    targets = [("", ps.master)]
    slots = ps.opt.slots
    if len(slots) >= 1 and ps.slot_m is not None:
        targets.append((slots[0], ps.slot_m))
    if len(slots) >= 2 and ps.slot_v is not None:
        targets.append((slots[1], ps.slot_v))
    for slot, p in zip(L.slots, ps.params):
        order = storage_order(p)
        segs = _segments(tf_names, slot.name, list(p.shape), slot.numel)
        missing = [n for n, _, _, _ in segs if not rd.has(n)]
        if missing:
            raise KeyError("checkpoint %s has no variable %s (for %s)" % (export_dir, missing[0], slot.name))
        cache = {}

        def flat_of(vname, order=order, cache=cache):
            # the variable in THIS model's storage order; read once, sliced per piece
            if vname not in cache:
                cache[vname] = rd.read_storage(vname, order)
            return cache[vname]

        for sec, dst in targets:
            for name, lo, hi, _ in segs:
                vname = name if sec == "" else "%s/%s" % (name, sec)
                if not rd.has(vname):
                    continue
                for rank, shard_off, var_off, length in L.owner_pieces(slot.index):
                    if rank != r or not hosted:
                        continue
                    a, b = max(var_off, lo), min(var_off + length, hi)
                    if a < b:
                        dst[shard_off + (a - var_off):shard_off + (b - var_off)].copy_(flat_of(vname)[a - lo:b - lo])
        # the full variable, for the local wire-dtype copy the model computes with (a dedicated PS task has none)
        if getattr(ps, "param_flat", None) is not None:
            for name, lo, hi, _ in segs:
                ps.param_flat[slot.flat_off + lo:slot.flat_off + hi].copy_(flat_of(name))
    ps.global_step = rd.step
    if getattr(ps, "hyper", None) is not None:
        ps.hyper[2:3].fill_(float(rd.step + 1))
    if getattr(ps, "_unpack_needed", False):
        ps._unpack_params()
    return rd
