"""Shows what the stream recording in DeviceLoader prevents: with it patched out, batches read behind a lagging
compute stream come back as other batches' data."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from torch.utils.data import DataLoader
from pytorch_distributed_training_amd.data import PadCollator, SyntheticPairDataset
from pytorch_distributed_training_amd.data.loader import DeviceLoader, shard_dataloader

def old_hand_over(self, moved, ev):
    torch.cuda.current_stream(self.device).wait_event(ev)
    return moved

def run(patched):
    if patched:
        DeviceLoader._hand_over = old_hand_over
    ds = SyntheticPairDataset(96 * 12, 28996, max_len=64, seed=3)
    cpu_loader = shard_dataloader(DataLoader(ds, shuffle=False, collate_fn=PadCollator(0, "max_length", 128), batch_size=96), 1, 0, seed=0)
    expected = [{k: v.clone() for k, v in b.items()} for b in cpu_loader]
    loader = DeviceLoader(cpu_loader, torch.device("cuda"), prefetch=2)
    torch.cuda.synchronize(); torch.cuda._sleep(int(2e9))
    seen = []
    for batch in loader:
        seen.append({k: v.clone() for k, v in batch.items()}); del batch
    torch.cuda.synchronize()
    bad = sum(int(not torch.equal(g[k].cpu(), r[k])) for g, r in zip(seen, expected) for k in r)
    worst = max(int(g["token_type_ids"].max()) for g in seen)
    print(f"patched_out={patched}: {bad} of {4 * len(seen)} tensors differ; largest token type id seen {worst}")
run(False); run(True)
