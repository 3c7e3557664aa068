"""distributed_tensorflow_amd: parameter-server training service and data plane for MI355X (gfx950)."""
import os as _os


def _ensure_hw_queues(minimum=8):
    """HIP maps the streams of a process onto ``GPU_MAX_HW_QUEUES`` hardware queues (4 unless set); streams that land on
    the same queue run in series.  A training rank with a live communicator uses the compute stream, the weight-gradient
    stream (``ops/streams.py``), the parameter-server comm stream, the process group's collective stream and RCCL's
    internal ones -- with four queues the weight-gradient stream shared a queue with the compute stream and the overlap it
    exists for was gone: ResNet-50 b256, one MI355X, live 1-rank RCCL group, 21.01 ms/step with 4 queues, 19.96 with 6 or
    8, 19.93 without a communicator (``scripts/gpu_r3_p.sh``).  The variable is read when the HIP runtime initialises,
    so importing this package before the first GPU call is early enough.  A larger value that is already set is kept;
    ``DTF_KEEP_HW_QUEUES=1`` leaves the variable alone."""
    global _hw_queues_before
    if _os.environ.get("DTF_KEEP_HW_QUEUES") == "1":
        return
    try:
        cur = int(_os.environ.get("GPU_MAX_HW_QUEUES", "0"))
    except ValueError:
        cur = 0
    if cur < minimum:
        if _hw_queues_before is _UNSET:
            _hw_queues_before = _os.environ.get("GPU_MAX_HW_QUEUES")
        _os.environ["GPU_MAX_HW_QUEUES"] = str(minimum)


def restore_hw_queues():
    """Undo :func:`_ensure_hw_queues` -- for a process that will replay the step as ONE hipGraph.  ROCm 7.2's graph
    executor got slower with more hardware queues: a captured ResNet-50 step replays in 21.0 ms with 4 queues and in
    29.1 ms with 8 (single-stream capture; 28.8 vs 21.0 ms with a live group; ``scripts/gpu_r3_t.sh``), while the eager
    step needs the queues (see above).  Only effective before the HIP runtime initialises; returns False when the
    runtime is already up (``torch.cuda.is_initialized()``) and the setting can no longer change."""
    global _hw_queues_before
    try:
        import sys

        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_initialized():
            return False
    except Exception:
        pass
    if _hw_queues_before is not _UNSET:
        if _hw_queues_before is None:
            _os.environ.pop("GPU_MAX_HW_QUEUES", None)
        else:
            _os.environ["GPU_MAX_HW_QUEUES"] = _hw_queues_before
        _hw_queues_before = _UNSET
    return True


_UNSET = object()
_hw_queues_before = _UNSET
_ensure_hw_queues()
