"""Model zoo (new layer N3): the five workloads of BASELINE.json ``configs``.

The reference ships no model code at all -- the user's TF1 script lives in the
opaque ``tf-dis:4.0`` image (external X1/E0) -- so these are the
in-repo stand-ins for that script, one per north-star config:

  mnist_mlp     config #1  MNIST MLP 784-100-10 (TF ``mnist_replica`` style)
  resnet50      config #2  ResNet-50 v1, synthetic ImageNet, bf16 channels_last
  wide_deep     config #3  Wide&Deep CTR with row-sharded embedding tables
  bert_base     config #4  BERT-base pretraining (MLM + NSP), seq 512
  inception_v3  config #5  Inception-v3, sync-vs-async PS update

Each module exposes ``build(**kw)``, ``tf_variable_names(model)`` (checkpoint
naming), ``synthetic_batch(batch, device, ...)`` and
``loss_fn(model, batch)``.
"""
import importlib

_REGISTRY = {
    "mnist_mlp": "mnist_mlp",
    "mnist": "mnist_mlp",
    "resnet50": "resnet",
    "resnet101": "resnet",
    "resnet152": "resnet",
    "inception_v3": "inception_v3",
    "inception3": "inception_v3",
    "wide_deep": "wide_deep",
    "bert_base": "bert",
    "bert": "bert",
}


def get_model_module(name):
    key = name.lower().replace("-", "_")
    if key not in _REGISTRY:
        raise KeyError("unknown model %r (known: %s)" % (name, ", ".join(sorted(_REGISTRY))))
    return importlib.import_module(__name__ + "." + _REGISTRY[key])


def model_names():
    return sorted(_REGISTRY)
