"""Input pipeline of the built-in workloads: the job's ``data`` field.

The reference forwards ``data`` verbatim to the training script as argv[6] (``/root/reference/handlers/train_handler.py:82-83``;
its sample job trains the TF1 MNIST replica example, which reads the four ``*-idx?-ubyte`` files from that directory).  The
built-in workloads here read the same directory:

* **MNIST IDX files** (``train-images-idx3-ubyte[.gz]`` + ``train-labels-idx1-ubyte[.gz]``: what
  ``input_data.read_data_sets`` reads) -- ``mnist_mlp``;
* **array files**: ``images.npy`` + ``labels.npy`` (or ``x.npy`` + ``y.npy``), or any tuple of arrays listed by a
  ``manifest.json`` (``{"fields": ["a.npy", "b.npy", ...]}``) in the order of the model's batch tuple; ``.npy`` files are
  memory-mapped, so a dataset larger than the host's memory streams from the page cache.

A directory that holds none of these (the shipped ``post-data`` names ``/data/imagenet-synthetic``) means synthetic data.

Sharding follows between-graph replication: every worker draws the SAME per-epoch permutation (seeded), global batch ``g`` is
``perm[g * W * B : (g + 1) * W * B]`` and worker ``w`` takes rows ``[w * B, (w + 1) * B)`` of it -- disjoint across workers,
every sample once per epoch, the trailing partial global batch dropped.

``Prefetcher`` keeps the device fed: a host thread gathers + pins batch ``k + 2`` while batch ``k + 1`` is being copied and
transformed (uint8 -> normalised bf16 NHWC, ...) on a copy stream and batch ``k`` trains; the training stream only ever waits
for an event.
"""
from __future__ import annotations

import gzip
import json
import os
import queue
import struct
import threading

import numpy as np
import torch

_IDX_DTYPES = {0x08: np.uint8, 0x09: np.int8, 0x0B: ">i2", 0x0C: ">i4", 0x0D: ">f4", 0x0E: ">f8"}


def read_idx(path):
    """An IDX file (the MNIST container: 2 zero bytes, dtype code, rank, big-endian dims, data), plain or gzipped."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        head = f.read(4)
        if len(head) != 4 or head[0] != 0 or head[1] != 0 or head[2] not in _IDX_DTYPES:
            raise ValueError("%s is not an IDX file" % path)
        dims = struct.unpack(">" + "I" * head[3], f.read(4 * head[3]))
        data = np.frombuffer(f.read(), dtype=_IDX_DTYPES[head[2]])
    n = int(np.prod(dims)) if dims else 1
    if data.size < n:
        raise ValueError("%s is truncated: %d of %d elements" % (path, data.size, n))
    return data[:n].reshape(dims)


def write_idx(path, array):
    """Inverse of :func:`read_idx` for uint8 arrays (tests, and a way to hand a dataset to the MNIST workload)."""
    a = np.ascontiguousarray(array, dtype=np.uint8)
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(bytes([0, 0, 0x08, a.ndim]) + struct.pack(">" + "I" * a.ndim, *a.shape) + a.tobytes())


class ArrayDataset:
    """A tuple of arrays with a common leading dimension; ``kind`` picks the default transform."""

    def __init__(self, fields, kind="generic", source=""):
        n = {int(a.shape[0]) for a in fields}
        if len(n) != 1:
            raise ValueError("dataset fields disagree on the number of samples: %s" % sorted(n))
        self.fields, self.kind, self.source = list(fields), kind, source

    def __len__(self):
        return int(self.fields[0].shape[0])


def _first(data_dir, names):
    for n in names:
        p = os.path.join(data_dir, n)
        if os.path.isfile(p):
            return p
    return None


def open_dataset(data_dir, split="train"):
    """-> :class:`ArrayDataset` for a directory in one of the layouts of the module docstring, else ``None``.
    ``split``: which MNIST files to take when the directory holds both -- ``train`` (``train-*``) or ``test``
    (``t10k-*``); a directory that holds only one of the two is read whatever ``split`` says."""
    if not data_dir or not os.path.isdir(data_dir):
        return None
    man = os.path.join(data_dir, "manifest.json")
    if os.path.isfile(man):
        with open(man) as f:
            m = json.load(f)
        names = [os.path.basename(str(n)) for n in m["fields"]]         # files of THIS directory only
        return ArrayDataset([np.load(os.path.join(data_dir, n), mmap_mode="r", allow_pickle=False) for n in names],
                            kind=str(m.get("kind", "generic")), source=man)
    for pre in (("train", "t10k") if split != "test" else ("t10k", "train")):
        img = _first(data_dir, tuple(pre + e for e in ("-images-idx3-ubyte", "-images-idx3-ubyte.gz", "-images.idx3-ubyte")))
        lab = _first(data_dir, tuple(pre + e for e in ("-labels-idx1-ubyte", "-labels-idx1-ubyte.gz", "-labels.idx1-ubyte")))
        if img and lab:
            return ArrayDataset([read_idx(img), read_idx(lab)], kind="mnist", source=img)
    for xs, ys in (("images.npy", "labels.npy"), ("x.npy", "y.npy")):
        x, y = _first(data_dir, (xs,)), _first(data_dir, (ys,))
        if x and y:
            return ArrayDataset([np.load(x, mmap_mode="r", allow_pickle=False), np.load(y, mmap_mode="r", allow_pickle=False)],
                                kind="images", source=x)
    return None


class BatchStream:
    """Endless iterator of this worker's batches (tuples of numpy arrays), see the module docstring for the sharding."""

    def __init__(self, dataset, batch, worker=0, num_workers=1, seed=0, shuffle=True, start_step=0):
        self.ds, self.batch, self.worker, self.num_workers = dataset, int(batch), int(worker), int(num_workers)
        self.seed, self.shuffle = int(seed), bool(shuffle)
        self.global_batch = self.batch * self.num_workers
        self.per_epoch = len(dataset) // self.global_batch
        if self.per_epoch < 1:
            raise ValueError("dataset of %d samples is smaller than one global batch (%d workers x %d)" % (
                len(dataset), self.num_workers, self.batch))
        self.step = int(start_step)          # a resumed job continues where the data order left off
        self._epoch, self._perm = -1, None

    def _permutation(self, epoch):
        if epoch != self._epoch:
            n = len(self.ds)
            self._perm = np.random.default_rng(self.seed + epoch).permutation(n) if self.shuffle else np.arange(n)
            self._epoch = epoch
        return self._perm

    def indices(self, step):
        epoch, g = divmod(step, self.per_epoch)
        lo = g * self.global_batch + self.worker * self.batch
        return self._permutation(epoch)[lo:lo + self.batch]

    def __iter__(self):
        return self

    def __next__(self):
        idx = np.sort(self.indices(self.step))          # ascending rows: kinder to a memory-mapped file
        self.step += 1
        return tuple(np.ascontiguousarray(a[idx]) for a in self.ds.fields)


_IMAGENET_MEAN = (0.485, 0.456, 0.406)
_IMAGENET_STD = (0.229, 0.224, 0.225)


def transform_for(mod, dataset, device):
    """``fn(tensors already on `device`) -> the model's batch tuple``.  A model module may bring its own
    ``batch_from_arrays(tensors, device)``; otherwise by dataset kind:

    * ``mnist``: uint8 ``[B, 28, 28]`` -> float ``[B, 784]`` in [0, 1]; labels int64;
    * ``images``: uint8 ``[B, H, W, C]`` -> (x / 255 - mean) / std, or float ``[B, H, W, C]`` / ``[B, C, H, W]`` as it is,
      -> ``[B, C, H, W]`` channels-last, bf16 on the GPU; labels int64;
    * ``generic``: floating arrays in the training dtype, integer arrays as int64."""
    device = torch.device(device)
    if hasattr(mod, "batch_from_arrays"):
        return lambda ts: mod.batch_from_arrays(ts, device)
    fdt = torch.bfloat16 if device.type == "cuda" else torch.float32
    kind = dataset.kind

    def mnist(ts):
        x, y = ts
        return x.reshape(x.shape[0], -1).to(torch.float32).div_(255.0), y.to(torch.int64)

    def images(ts):
        x, y = ts[0], ts[1]
        if x.dim() != 4:
            raise ValueError("images dataset: expected [B, H, W, C] or [B, C, H, W], got %s" % (tuple(x.shape),))
        nhwc = x.shape[-1] in (1, 3, 4) and x.shape[1] not in (1, 3, 4)
        if x.dtype == torch.uint8:
            c = x.shape[-1] if nhwc else x.shape[1]
            mean = torch.tensor((_IMAGENET_MEAN if c == 3 else (0.5,) * c), device=x.device, dtype=torch.float32)
            std = torch.tensor((_IMAGENET_STD if c == 3 else (0.5,) * c), device=x.device, dtype=torch.float32)
            shape = (1, 1, 1, c) if nhwc else (1, c, 1, 1)
            x = x.to(torch.float32).div_(255.0).sub_(mean.view(shape)).div_(std.view(shape))
        if nhwc:
            x = x.permute(0, 3, 1, 2)               # NHWC memory seen as NCHW = channels-last: no data movement
        x = x.to(fdt).contiguous(memory_format=torch.channels_last)
        return (x, y.to(torch.int64)) + tuple(ts[2:])

    def generic(ts):
        return tuple(t.to(fdt) if t.is_floating_point() else t.to(torch.int64) for t in ts)

    return {"mnist": mnist, "images": images}.get(kind, generic)


class Prefetcher:
    """Iterator of device batches over a :class:`BatchStream` (or any iterator of numpy tuples).

    Host side: a daemon thread gathers the next batches and pins them (``depth`` deep).  Device side: batch ``k + 1`` is
    copied and transformed on a copy stream while batch ``k`` trains; ``next()`` makes the caller's stream wait for the
    copy's event and hands over the tensors (their memory is marked as used by the caller's stream, so the allocator
    does not recycle it under the training kernels)."""

    _END = object()

    def __init__(self, stream, transform, device, depth=2):
        self.device = torch.device(device)
        self.transform = transform
        self._it = iter(stream)
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._cuda = self.device.type == "cuda"
        self._copy_stream = torch.cuda.Stream(self.device) if self._cuda else None
        self._staged, self._error, self._done = None, None, False
        self._thread = threading.Thread(target=self._produce, name="dtf-data", daemon=True)
        self._thread.start()

    def _produce(self):
        try:
            while not self._stop.is_set():
                arrays = next(self._it)
                ts = [torch.from_numpy(a) for a in arrays]
                if self._cuda:
                    ts = [t.pin_memory() for t in ts]
                while not self._stop.is_set():
                    try:
                        self._q.put(ts, timeout=0.2)
                        break
                    except queue.Full:
                        continue
        except StopIteration:
            self._q.put(self._END)
        except BaseException as e:          # noqa: BLE001 -- handed to the consumer, which re-raises it
            self._q.put(e)

    def _stage(self):
        host = self._q.get()
        if host is self._END:
            return None
        if isinstance(host, BaseException):
            raise host
        if not self._cuda:
            return (self.transform(host), None, host)
        with torch.cuda.stream(self._copy_stream):
            dev = [t.to(self.device, non_blocking=True) for t in host]
            batch = self.transform(dev)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        return (batch, ev, host)          # `host`: the pinned source stays alive until the copy has been waited for

    def __iter__(self):
        return self

    def __next__(self):
        if self._error is not None:          # the producer failed while THIS batch was being prepared
            err, self._error = self._error, None
            raise err
        if self._staged is None and not self._done:
            self._staged = self._stage()
        if self._staged is None:
            self._done = True
            raise StopIteration
        batch, ev, _host = self._staged
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in batch:
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)
        try:
            self._staged = self._stage()     # start the next copy before the caller launches this batch's step
            self._done = self._staged is None
        except BaseException as e:           # noqa: BLE001 -- belongs to the NEXT batch: raised when that one is asked for
            self._staged, self._error = None, e
        return batch

    def close(self):
        self._stop.set()
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=2.0)


class Cycle:
    """``K`` device-resident batches served round-robin (``synthetic_batches=K``: synthetic data that is not one fixed batch)."""

    def __init__(self, batches):
        self.batches, self.i = list(batches), 0

    def __iter__(self):
        return self

    def __next__(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return b

    def close(self):
        pass
