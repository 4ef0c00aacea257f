"""MI355X-native distributed fine-tuning framework (BERT-class sequence classification).

Capabilities of the two training scripts of qqaatw/pytorch-distributed-training
(/root/reference/test_data_parallelism.py, /root/reference/test_model_parallelism.py), rebuilt for
gfx950: hand-written HIP kernels for the encoder hot path, a C++ bucket reducer over RCCL, a 2-stage
micro-batched pipeline composed with data parallelism, HF-compatible checkpoints.
"""
import os as _os

# dmabuf IPC handles are the only kind the host drivers of MI355X nodes export; the HIP runtime reads this when it
# initialises, so it is set here, at package import, before any HIP call (see launch.export_gpu_ipc_env)
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _hardware_queues() -> None:
    """HIP gives every stream one of GPU_MAX_HW_QUEUES hardware queues (4 by default) and hands out the least-used one once
    all exist; streams that share a queue run strictly one after the other.  A training rank here has the compute stream,
    the weight-gradient lane, the communication stream, RCCL's own streams and the micro-batch lanes: with 4 queues which
    two of them end up sharing depends on creation order - measured on MI355X (BERT-large, 96 samples per step, 1-rank RCCL
    rehearsal): 36.0 ms per step with 4 queues (weight-gradient lane serialised behind the compute stream), 40.0 ms with 8,
    34.4 ms with 16, where every stream has a queue of its own; no eager configuration got slower with 16.  hipGraph
    replays do (96 -> 235 ms for the captured 12 x 8 step: utils/graph.py warns; ``bench.py --graph`` keeps the default).
    ``PDT_HW_QUEUES`` sets another count (0: leave the environment alone).  Read by the HIP runtime when it initialises."""
    want = int(_os.environ.get("PDT_HW_QUEUES", "16"))
    if want > 0 and int(_os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < want:
        _os.environ["GPU_MAX_HW_QUEUES"] = str(min(want, 32))


_hardware_queues()

__version__ = "0.1.0"
