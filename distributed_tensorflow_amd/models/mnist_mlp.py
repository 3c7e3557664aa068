"""MNIST MLP 784-100-10 (config #1): the model of TensorFlow's classic
``mnist_replica.py`` between-graph-replication example, which is the kind of
script the reference's jobs run through ``/notebooks/entry.sh``
(``handlers/train_handler.py:82-83``).  Variables carry that script's names:
``hid_w``, ``hid_b``, ``sm_w``, ``sm_b`` (+ ``global_step``), stored
``[in, out]`` like TF so a checkpoint shard is layout-compatible.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

IMAGE_PIXELS = 28


class MnistMLP(nn.Module):
    def __init__(self, hidden_units=100, num_classes=10):
        super().__init__()
        d = IMAGE_PIXELS * IMAGE_PIXELS
        # TF: truncated_normal(stddev = 1/IMAGE_PIXELS) and 1/sqrt(hidden)
        self.hid_w = nn.Parameter(torch.empty(d, hidden_units))
        self.hid_b = nn.Parameter(torch.zeros(hidden_units))
        self.sm_w = nn.Parameter(torch.empty(hidden_units, num_classes))
        self.sm_b = nn.Parameter(torch.zeros(num_classes))
        nn.init.trunc_normal_(self.hid_w, std=1.0 / IMAGE_PIXELS, a=-2.0 / IMAGE_PIXELS, b=2.0 / IMAGE_PIXELS)
        s = 1.0 / math.sqrt(hidden_units)
        nn.init.trunc_normal_(self.sm_w, std=s, a=-2 * s, b=2 * s)

    def forward(self, x):
        x = x.reshape(x.shape[0], -1)
        hid = F.relu(torch.addmm(self.hid_b, x, self.hid_w))
        return torch.addmm(self.sm_b, hid, self.sm_w)


def build(hidden_units=100, **_):
    return MnistMLP(hidden_units)


def tf_variable_names(model):
    return {n: n for n, _ in model.named_parameters()}


def synthetic_batch(batch, device="cpu", dtype=torch.float32, seed=None, **_):
    """Synthetic digits with learnable structure: class k lights up a band of
    pixels, plus noise, so the loss must fall when training works."""
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    y = torch.randint(0, 10, (batch,), generator=g)
    x = torch.rand(batch, IMAGE_PIXELS * IMAGE_PIXELS, generator=g) * 0.5
    band = IMAGE_PIXELS * IMAGE_PIXELS // 10
    idx = torch.arange(IMAGE_PIXELS * IMAGE_PIXELS).unsqueeze(0)
    mask = (idx >= (y * band).unsqueeze(1)) & (idx < ((y + 1) * band).unsqueeze(1))
    x = x + mask.float()
    return x.to(device=device, dtype=dtype), y.to(device)


def loss_fn(model, batch):
    x, y = batch
    return F.cross_entropy(model(x).float(), y)


def default_optimizer():
    from ..parallel.ps_runtime import OptimizerSpec

    # mnist_replica: AdamOptimizer(learning_rate=0.01); plain SGD also fine
    return OptimizerSpec(kind="sgd", lr=0.1)
