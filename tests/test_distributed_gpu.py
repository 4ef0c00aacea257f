"""Multi-process rehearsals of the distributed native path on the one GPU of the test box (gloo transport;
see dist_workers.py): DP gradient equality, DP x 2-stage pipeline and tower-parallel loss curves vs one process."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dist_workers as W  # noqa: E402
from pytorch_distributed_training_amd.testing import run_distributed  # noqa: E402

pytestmark = pytest.mark.gpu


def test_dp_native_gradients_match_big_batch_two_ranks_one_gpu():
    for r in run_distributed(W.gpu_dp_native, 2):
        assert r["err"] < 2e-2 and r["same"]
        assert r["order"] == list(range(r["nb"]))


@pytest.mark.parametrize("world", [2, 4])
def test_pipeline_native_loss_curve(world):
    ref = W.gpu_single_reference("pipeline")
    res = run_distributed(W.gpu_pipeline_recipe, world, args=("pipeline",))
    got = [r for r in res if r["stage"] == 1 and r["replica"] == 0][0]["losses"]
    if world == 4:  # 2 replicas x per-replica batch 16 = global batch 32 => 2 steps per epoch, other batches
        assert len(got) == 4 and all(x == x and x < 1.5 for x in got)
        return
    assert len(got) == len(ref) == 6
    for a, b in zip(got, ref):
        assert abs(a - b) < 2e-2, (got, ref)


def test_tower_parallel_native_loss_curve():
    ref = W.gpu_single_reference("towers", steps=4)
    res = run_distributed(W.gpu_pipeline_recipe, 2, args=("towers",))
    got = [r for r in res if r["stage"] == 0][0]["losses"][:4]
    for a, b in zip(got, ref):
        assert abs(a - b) < 2e-2, (got, ref)


def test_rccl_call_path_single_rank():
    """A 1-rank "nccl" (= RCCL) group with forced collectives: broadcast, bucketed all-reduce with the native
    AVG reduction on the communicator's stream, barrier - the API path the multi-GPU runs take."""
    import subprocess

    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
                  PDT_FORCE_COLLECTIVES="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from pytorch_distributed_training_amd.models.bert import build_model
from pytorch_distributed_training_amd.ops.arena import ParamArena
from pytorch_distributed_training_amd.parallel.dp import DataParallel
from pytorch_distributed_training_amd.data.synthetic import fixed_batch
torch.manual_seed(0)
m = build_model("bert-tiny").cuda()
arena = ParamArena(m, torch.device("cuda"), torch.float16, min_bucket_elems=1 << 14)
m.enable_native(arena, seed=1)
dp = DataParallel(m, arena)
assert dp.reducer.world_size() == 1
b = fixed_batch(8, 32, 512, "cuda")
dp.train()
dp(**b).loss.backward()
torch.cuda.synchronize()
g1 = arena.grad.clone()
st = dp.reducer.stats()
assert st["collectives"] == dp.reducer.num_buckets() >= 3, st
assert dp.reducer.launch_order() == list(range(dp.reducer.num_buckets()))
arena.zero_grad()
m(**b).loss.backward()          # same micro-batch index => same dropout masks?  no: compare norms only
torch.cuda.synchronize()
assert torch.isfinite(g1).all() and g1.abs().sum() > 0
t = torch.ones(4, device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
dist.destroy_process_group()
print("RCCL_PATH_OK")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "RCCL_PATH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
