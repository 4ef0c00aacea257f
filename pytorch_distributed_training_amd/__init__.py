"""MI355X-native distributed fine-tuning framework (BERT-class sequence classification).

Capabilities of the two training scripts of qqaatw/pytorch-distributed-training
(/root/reference/test_data_parallelism.py, /root/reference/test_model_parallelism.py), rebuilt for
gfx950: hand-written HIP kernels for the encoder hot path, a C++ bucket reducer over RCCL, a 2-stage
micro-batched pipeline composed with data parallelism, HF-compatible checkpoints.
"""
__version__ = "0.1.0"
