import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-process / end-to-end tests")


def free_port():
    """A port for a rendezvous that starts a moment later.  NOT from the kernel's ephemeral range (``bind(0)``): any
    outgoing connection of any process may be given that very port as its source port between this call and the
    listener's bind -- seen once as EADDRINUSE in a GPU suite run.  Ports 10000-19999 are outside that range and below
    the launcher's own (23000+)."""
    import random

    for _ in range(200):
        port = random.randint(10000, 19999)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture
def port():
    return free_port()


@pytest.fixture
def store():
    from distributed_tensorflow_amd.utils.store import MemoryStore, RedisHelper

    st = MemoryStore()
    RedisHelper.install(st)
    yield st
    RedisHelper.reset()


@pytest.fixture(autouse=True)
def _clean_config():
    from distributed_tensorflow_amd.utils.config import ApiConfig

    yield
    ApiConfig.clear_overrides()


@pytest.fixture(scope="session", autouse=True)
def _close_process_group():
    """GPU tests keep one live 1-rank RCCL group for the session; close it explicitly (an open group at exit is a
    warning today and a hang source with more ranks)."""
    yield
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    except Exception:
        pass
