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

__version__ = "0.1.0"
