"""DeviceLoader on a GPU: prefetched batches are copied on a side stream and consumed on the compute stream."""
import pytest
import torch
from torch.utils.data import DataLoader

from pytorch_distributed_training_amd.data import PadCollator, SyntheticPairDataset
from pytorch_distributed_training_amd.data.loader import DeviceLoader, shard_dataloader

pytestmark = pytest.mark.gpu


def test_prefetched_batches_survive_a_lagging_compute_stream():
    """The host runs ahead of the GPU: every batch is dropped by Python while the kernel that reads it is still
    queued behind earlier work.  The memory of a dropped batch must not be handed to a later prefetch copy before
    that kernel has run (the loader records the consumer stream on the batch tensors) - otherwise a consumer reads
    another batch's token ids, or token ids where it expects token types."""
    ds = SyntheticPairDataset(96 * 12, 28996, max_len=64, seed=3)
    collate = PadCollator(0, "max_length", 128)  # equal shapes: a freed block fits the next copy exactly
    cpu_loader = shard_dataloader(DataLoader(ds, shuffle=False, collate_fn=collate, batch_size=96), 1, 0, seed=0)
    expected = [{k: v.clone() for k, v in b.items()} for b in cpu_loader]
    loader = DeviceLoader(cpu_loader, torch.device("cuda"), prefetch=2)
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2e9))  # ~1 s of work queued on the compute stream: nothing below runs before the loop ends
    seen = []
    for batch in loader:
        seen.append({k: v.clone() for k, v in batch.items()})  # the consumer kernel, queued behind the sleep
        del batch
    torch.cuda.synchronize()
    assert len(seen) == len(expected) == 12
    for got, ref in zip(seen, expected):
        for k in ref:
            assert torch.equal(got[k].cpu(), ref[k]), k
