"""The training curve of a job: what a TF1 script leaves for TensorBoard with ``tf.summary.scalar`` (an events file in
the job's ``export`` directory), as plain text.

* ``<export>/scalars.jsonl`` -- one JSON object per logged step (``step``, ``wall``, ``loss``, ``items_per_sec``, anything
  else passed), appended by the chief; a resumed job appends to the same file;
* the job's ``<uid>:metrics`` hash gets a thinned copy (``curve``: at most ~400 points, older ones at half the density each
  time the limit is reached), so ``GET /v1/train?uid=`` can draw the curve without access to the export directory.

Values are read where the training loop synchronises anyway (its ``log_every`` line): no extra device sync."""
from __future__ import annotations

import json
import os
import time


class ScalarLog:
    def __init__(self, export_dir=None, sink=None, limit=400):
        self.path = None
        if export_dir and "://" not in str(export_dir):
            try:
                os.makedirs(export_dir, exist_ok=True)
                self.path = os.path.join(export_dir, "scalars.jsonl")
            except OSError:
                self.path = None
        self.sink, self.limit, self.points = sink, max(8, int(limit)), []

    def add(self, step, **scalars):
        row = {"step": int(step), "wall": round(time.time(), 3)}
        row.update({k: (float(v) if v is not None else None) for k, v in scalars.items()})
        if self.path:
            try:
                with open(self.path, "a") as f:
                    f.write(json.dumps(row) + "\n")
            except OSError:
                self.path = None        # a full or vanished disk must not stop the training
        self.points.append([row["step"]] + [row.get(k) for k in ("loss", "items_per_sec")])
        if len(self.points) > self.limit:
            half = len(self.points) // 2
            self.points = self.points[:half][::2] + self.points[half:]
        if self.sink is not None:
            self.sink(curve=json.dumps(self.points))
        return row


def read_scalars(export_dir):
    """-> list of the rows of ``<export>/scalars.jsonl`` (empty when there is none)."""
    p = os.path.join(export_dir, "scalars.jsonl")
    out = []
    if os.path.isfile(p):
        with open(p) as f:
            for line in f:
                line = line.strip()
                if line:
                    try:
                        out.append(json.loads(line))
                    except ValueError:
                        pass            # a torn last line of a killed job
    return out
