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
