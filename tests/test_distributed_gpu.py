"""Multi-process rehearsals of the distributed native path on the one GPU of the test box (gloo transport;
see dist_workers.py): DP gradient equality, DP x 2-stage pipeline and tower-parallel loss curves vs one process."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dist_workers as W  # noqa: E402
from pytorch_distributed_training_amd.testing import run_distributed  # noqa: E402

pytestmark = pytest.mark.gpu


# bert-tiny curves (losses ~0.7, lr 2e-3 - 100x the reference's).  A run with ``-s`` prints what the layouts achieve:
# 8e-7 ... 1.2e-5 on MI355X for every layout below; the tolerance leaves a factor of 15 over the worst of them
TOL_TINY = 2e-4


def _maxdiff(name, got, ref):
    """Largest |loss difference| of two curves; printed so that a run with ``-s`` shows what the tolerances leave."""
    assert len(got) == len(ref), (name, got, ref)
    d = max(abs(a - b) for a, b in zip(got, ref))
    print(f"PARITY {name}: max |dloss| {d:.2e} over {len(ref)} steps")
    return d


def test_dp_native_gradients_match_big_batch_two_ranks_one_gpu():
    for r in run_distributed(W.gpu_dp_native, 2):
        assert r["err"] < 2e-2 and r["same"]
        assert r["order"] == list(range(r["nb"]))


def test_dp_accumulation_window_with_two_gradient_lanes_two_ranks_one_gpu():
    for r in run_distributed(W.gpu_dp_accumulate_two_lanes, 2):
        assert r["lanes"] and r["err"] < 2e-2, r
        assert r["order"] == list(range(r["nb"]))


def test_unusable_rccl_communicator_falls_back_to_the_process_group():
    """Two ranks on ONE GPU try to open the framework's RCCL communicator (forced; RCCL refuses the duplicate device,
    and rank 1 is told to fail its self test anyway): the ranks agree over the process group, abort, and train
    through the process group with identical replicas."""
    for r in run_distributed(W.gpu_dp_comm_fallback, 2):
        assert r["label"] == "c10d:gloo" and r["comm_is_none"] and r["same"] and r["err"] < 2e-2


@pytest.mark.parametrize("world", [2, 4])
def test_pipeline_native_loss_curve(world):
    """1 or 2 replicas of the 2-stage pipeline against ONE process on the same global batch (16 resp. 32 samples)."""
    replicas = world // 2
    steps = 6 if replicas == 1 else 4  # 64 examples, 2 epochs: 4 resp. 2 optimizer steps per epoch
    ref = W.gpu_single_reference("pipeline", steps=steps, replicas=replicas)
    res = run_distributed(W.gpu_pipeline_recipe, world, args=("pipeline",))
    got = [r for r in res if r["stage"] == 1 and r["replica"] == 0][0]["losses"]
    assert len(got) == len(ref) == steps, (got, ref)
    assert _maxdiff(f"pipeline nodrop world {world}", got, ref) < TOL_TINY, (got, ref)
    assert all(r["replicas_bit_equal"] in (True, None) for r in res)


def test_pipeline_hands_packed_rows_across_the_link():
    """Dynamically padded batches: both stages run on the real tokens only and only those rows (and their gradients)
    cross the link; the loss curve is the one of the padded run of the same recipe (dropout on: the masks are keyed
    by sample and position, not by packed row)."""
    runs = {}
    for varlen in (True, False):
        res = run_distributed(W.gpu_pipeline_recipe, 2, args=("pipeline", False, "bert-tiny", varlen))
        runs[varlen] = [r for r in res if r["stage"] == 1][0]
    packed, padded = runs[True], runs[False]
    assert packed["packed"] is True and padded["packed"] is False
    assert len(packed["losses"]) == len(padded["losses"]) == 6
    assert _maxdiff("packed vs padded", packed["losses"], padded["losses"]) < TOL_TINY, (packed["losses"], padded["losses"])
    assert packed["link_mb"] < 0.9 * padded["link_mb"], (packed["link_mb"], padded["link_mb"])


@pytest.mark.parametrize("world", [2, 4])
def test_tower_parallel_native_loss_curve(world):
    """TriBert towers over rank pairs; with 4 ranks two such pairs are data parallel (fp16, debug guards on): the
    logged loss is the one of a single-process TriBert on the global batch and the replicas stay bit-identical -
    the embedding bucket of the first tower rank completes in a second backward call and must still be reduced in
    the step it belongs to (/root/reference/test_model_parallelism.py:248-253,292-299)."""
    replicas = world // 2
    ref = W.gpu_single_reference("towers", steps=4, replicas=replicas)
    res = run_distributed(W.gpu_pipeline_recipe, world, args=("towers", world == 4))
    got = [r for r in res if r["stage"] == 0 and r["replica"] == 0][0]["losses"][:4]
    assert len(got) == 4
    assert _maxdiff(f"towers nodrop world {world}", got, ref) < TOL_TINY, (got, ref)
    if world == 4:
        assert all(r["replicas_bit_equal"] is True for r in res)


def test_dropout_masks_follow_the_sample_across_parallel_layouts():
    """Dropout 0.1 ON (the reference trains with it in both scripts, /root/reference/test_model_parallelism.py:230-239):
    masks are keyed by (seed, step, global sample id, position), so 2 data-parallel ranks, a 2-stage pipeline cut into
    4 micro-batches, 2 replicas x 2 stages and the tower layout all reproduce the loss curve of ONE process on the same
    global batch to fp16 tolerance - rank- or micro-batch-keyed masks differ by ~0.05 in the loss from step one."""
    ref = W.gpu_single_reference("dp", steps=6, replicas=2, model="bert-tiny")
    for got in run_distributed(W.gpu_dp_dropout, 2):
        assert _maxdiff("dp2 dropout", got, ref) < TOL_TINY, (got, ref)
    for world, steps in ((2, 6), (4, 4)):
        ref = W.gpu_single_reference("pipeline", steps=steps, replicas=world // 2, model="bert-tiny")
        res = run_distributed(W.gpu_pipeline_recipe, world, args=("pipeline", False, "bert-tiny"))
        got = [r for r in res if r["stage"] == 1 and r["replica"] == 0][0]["losses"]
        assert _maxdiff(f"pipeline dropout world {world}", got, ref) < TOL_TINY, (world, got, ref)
    ref = W.gpu_single_reference("towers", steps=4, replicas=2, model="bert-tiny")
    res = run_distributed(W.gpu_pipeline_recipe, 4, args=("towers", False, "bert-tiny"))
    got = [r for r in res if r["stage"] == 0 and r["replica"] == 0][0]["losses"][:4]
    assert _maxdiff("towers dropout world 4", got, ref) < TOL_TINY, (got, ref)


def test_pipeline_stages_skip_overflowed_steps_together():
    """fp16 with a loss scale that overflows on the first steps (2^24 is halved several times before a gradient is finite):
    the stages of a replica agree on every skip - the pipeline's loss curve is the one of a single process started with
    the same scale, and differs from the curve of a run whose scale never overflows."""
    ref = W.gpu_single_reference("pipeline", steps=6, replicas=1, init_scale=2.0 ** 24)
    calm = W.gpu_single_reference("pipeline", steps=6, replicas=1)
    assert max(abs(a - b) for a, b in zip(ref, calm)) > 1e-3, (ref, calm)  # updates were skipped
    res = run_distributed(W.gpu_pipeline_recipe, 2, args=("pipeline", False, "bert-tiny-nodrop", True, 64, 2e-3, 2.0 ** 24))
    got = [r for r in res if r["stage"] == 1][0]["losses"]
    assert _maxdiff("pipeline overflowing scale", got, ref) < TOL_TINY, (got, ref)


def test_eight_ranks_as_four_pipeline_pairs_reproduce_the_single_process_loss_curve():
    """The layout of a whole node - 4 data-parallel replicas x 2 pipeline stages, rank = replica * 2 + stage
    (/root/reference/test_model_parallelism.py:244-253 on 8 devices) - rehearsed with 8 processes on the one GPU, fp16,
    dropout 0.1 on: six optimizer steps on a global batch of 64 give the loss curve of ONE process on that batch, and
    the four replicas of each stage end bit-identical."""
    ref = W.gpu_single_reference("pipeline", steps=6, replicas=4, model="bert-tiny", examples=256)
    res = run_distributed(W.gpu_pipeline_recipe, 8, args=("pipeline", False, "bert-tiny", True, 256))
    got = [r for r in res if r["stage"] == 1 and r["replica"] == 0][0]["losses"]
    assert _maxdiff("8 ranks = 4 x pp2", got, ref) < TOL_TINY, (got, ref)
    assert all(r["replicas_bit_equal"] is True for r in res)


def test_bert_base_two_replicas_of_two_stages_match_one_process():
    """Real model size in the driver-run suite: bert-base-cased (110 M parameters), 4 ranks = 2 data-parallel replicas x 2
    pipeline stages on the one GPU, fp16, dropout 0.1 on, the reference's learning-rate scale: four optimizer steps on a
    global batch of 32 reproduce the loss curve of ONE process on that batch to 5e-4 (the losses are ~0.7; fp16
    activations crossing the link and a different summation order of the gradient are all that differs), and the
    replicas of each stage stay bit-identical."""
    ref = W.gpu_single_reference("pipeline", steps=4, replicas=2, model="bert-base-cased", examples=128, lr=5e-5)
    res = run_distributed(W.gpu_pipeline_recipe, 4, args=("pipeline", False, "bert-base-cased", True, 128, 5e-5))
    got = [r for r in res if r["stage"] == 1 and r["replica"] == 0][0]["losses"][:4]
    assert _maxdiff("bert-base 2 x pp2", got, ref) <= 5e-4, (got, ref)
    assert all(r["replicas_bit_equal"] is True for r in res)


def test_rccl_call_path_single_rank():
    """A 1-rank "nccl" (= RCCL) group with forced collectives: broadcast, bucketed all-reduce with the native
    AVG reduction on the communicator's stream, barrier - the API path the multi-GPU runs take."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["PDT_TEST_PORT"],
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
    _run_snippet(code, "RCCL_PATH_OK")


def _run_snippet(code: str, marker: str, timeout: int = 180):
    import subprocess

    from pytorch_distributed_training_amd.launch import find_free_port

    # (PDT_TEST_PORT: a free rendezvous port for snippets that open a process group of their own)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       env=dict(os.environ, PDT_TEST_PORT=str(find_free_port())))
    assert marker in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_native_rccl_communicator_single_rank():
    """The framework's own communicator (RCCL C API + its own HIP stream) in a 1-rank group: every collective
    and a self send/recv pair really run through RCCL; results are checked against their 1-rank definitions,
    and the stream ordering (fork / join) against a kernel enqueued right before the collective."""
    code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from pytorch_distributed_training_amd.parallel.comm import create_rccl_comm, close_all
torch.cuda.set_device(0)
comm = create_rccl_comm()
assert comm.world_size == 1 and comm.rank == 0 and comm.rccl_version() > 20000
x = torch.randn(1 << 22, device="cuda")
y = x * 3.0                      # producer kernel on the compute stream; the all-reduce must see its result
comm.all_reduce(y, average=True)
z = y + 1.0                      # consumer kernel on the compute stream; must see the all-reduce's result
torch.cuda.synchronize()
assert torch.equal(z, x * 3.0 + 1.0)
for dt in (torch.float16, torch.bfloat16, torch.float32, torch.int64, torch.int32):
    t = torch.arange(1000).to(dt).cuda(); ref = t.clone()
    comm.all_reduce(t); comm.broadcast(t, 0)           # (no host synchronisation between the calls)
    out = torch.empty_like(t); comm.all_gather(out, t)
    rs = torch.empty_like(t); comm.reduce_scatter(rs, out)
    assert torch.equal(out, ref) and torch.equal(rs, ref), dt
# 40 fork -> collective -> join rounds back to back on the null stream, nothing else in the queues: with a
# communication stream of elevated priority this chain never drained (bench/dev/rccl_stall.cpp)
import time
t = torch.arange(1000, dtype=torch.float32).cuda(); ref = t.clone(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(40):
    comm.all_reduce(t)
torch.cuda.synchronize()
assert time.time() - t0 < 5.0, f"chain of 40 collectives took {time.time() - t0:.1f} s"
assert torch.equal(t, ref) and comm.comm_count() == 1
src = torch.randn(4096, 64, device="cuda", dtype=torch.float16); dst = torch.zeros_like(src)
comm.p2p([0, 1], [src, dst], [0, 0])          # batched send + recv (to itself: one group call)
torch.cuda.synchronize()
assert torch.equal(src, dst)
comm.barrier(); comm.check_async_error()
# the start-up self test of multi-rank communicators (bounded wait on the communication stream + value check)
from pytorch_distributed_training_amd.parallel.comm import self_test
assert self_test(comm) == "" and comm.wait_idle(1.0)
assert comm.collectives() >= 63 and comm.bytes() > 0
close_all()
print("NATIVE_RCCL_OK")
'''
    _run_snippet(code, "NATIVE_RCCL_OK")


def test_bucket_reducer_over_native_rccl_communicator():
    """Data-parallel backward with the bucket reducer on the native communicator (1-rank group, collectives
    forced): one all-reduce per bucket in bucket order on the communication stream, gradients equal to the
    run without data parallelism (same seeds => same dropout masks), also with a 16-bit wire dtype."""
    code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["PDT_FORCE_COLLECTIVES"] = "1"
from pytorch_distributed_training_amd.models.bert import build_model
from pytorch_distributed_training_amd.ops.arena import ParamArena
from pytorch_distributed_training_amd.parallel.dp import make_reducer
from pytorch_distributed_training_amd.parallel.comm import create_rccl_comm
from pytorch_distributed_training_amd.data.synthetic import fixed_batch
torch.cuda.set_device(0)
b = fixed_batch(8, 32, 512, "cuda")
def grads(mode):
    torch.manual_seed(0)
    m = build_model("bert-tiny").cuda()
    arena = ParamArena(m, torch.device("cuda"), torch.float16, min_bucket_elems=1 << 14)
    m.enable_native(arena, seed=1)
    m.train()
    red = None
    if mode is not None:
        red, label, _ = make_reducer(None, arena, list(arena.bucket_segments), True, mode[1], comm_backend="rccl")
        assert label == "rccl"
        seg_buckets = [s.buckets for s in arena.segments]
        m.runtime.segment_done = lambda seg: [red.mark_ready(b, 1) for b in seg_buckets[seg]]
        red.begin(True)
    m(**b).loss.backward()
    if red is not None:
        red.finish()
        assert red.launch_order() == list(range(red.num_buckets())), red.launch_order()
        assert red.stats()["collectives"] == red.num_buckets() >= 3
    torch.cuda.synchronize()
    return arena.grad.clone()
g0 = grads(None)
g1 = grads(("rccl", None))
g2 = grads(("rccl", torch.bfloat16))
os.environ["PDT_COMM_ALGO"] = "rs_ag"      # buckets as reduce-scatter + all-gather (in place) instead of one all-reduce
g3 = grads(("rccl", None))
assert torch.equal(g3, g1)
assert g0.abs().sum() > 0 and torch.equal(g1, g0)   # (every gradient reduction is deterministic)
assert (g2 - g0).abs().max() <= 1e-2 * g0.abs().max() + 1e-6
assert create_rccl_comm().collectives() >= 6
# the loss scaler's overflow flag set bucket by bucket on the communication stream, behind each all-reduce
def overflow_flag(loss_factor):
    torch.manual_seed(0)
    m = build_model("bert-tiny").cuda()
    arena = ParamArena(m, torch.device("cuda"), torch.float16, min_bucket_elems=1 << 14)
    m.enable_native(arena, seed=1)
    m.train()
    red, _, _ = make_reducer(None, arena, list(arena.bucket_segments), True, None, comm_backend="rccl")
    state = torch.tensor([1.0, 0.0, 0.0, 0.0], device="cuda")
    red.set_overflow_state(state)
    assert red.checks_overflow()
    seg_buckets = [s.buckets for s in arena.segments]
    m.runtime.segment_done = lambda seg: [red.mark_ready(b, 1) for b in seg_buckets[seg]]
    red.begin(True)
    (m(**b).loss * loss_factor).backward()
    red.finish()
    torch.cuda.synchronize()
    return float(state[2])
assert overflow_flag(1.0) == 0.0 and overflow_flag(1e30) == 1.0
print("REDUCER_NATIVE_RCCL_OK")
'''
    _run_snippet(code, "REDUCER_NATIVE_RCCL_OK")


def test_adamw_runs_bucket_by_bucket_behind_the_all_reduces_in_bf16():
    """bf16 run on the native communicator (1-rank group, collectives forced): the reducer applies AdamW to every bucket
    on the communication stream right behind its all-reduce, the optimizer step only handles what is left (left-over
    buckets) and the scaler bookkeeping.  Three steps end in bit-identical parameters, 16-bit copies and moments to
    the run that updates the whole arena in optimizer.step(); large buckets are cut into several that all wait for
    the same segment.  fp16 runs (dynamic loss scale: one skip decision for all buckets) refuse the hook."""
    code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["PDT_FORCE_COLLECTIVES"] = "1"
from pytorch_distributed_training_amd.models.bert import build_model
from pytorch_distributed_training_amd.ops.arena import ParamArena
from pytorch_distributed_training_amd.optim import FusedAdamW, LossScaler
from pytorch_distributed_training_amd.parallel.dp import make_reducer
from pytorch_distributed_training_amd.data.synthetic import fixed_batch
torch.cuda.set_device(0)
b = fixed_batch(8, 32, 512, "cuda")
def run(per_bucket, dtype=torch.bfloat16, steps=3):
    torch.manual_seed(0)
    m = build_model("bert-tiny").cuda()
    arena = ParamArena(m, torch.device("cuda"), dtype, min_bucket_elems=1 << 14, max_bucket_elems=1 << 17,
                       tail_bucket_elems=1 << 16)
    m.enable_native(arena, seed=1)
    m.train()
    red, label, _ = make_reducer(None, arena, list(arena.bucket_segments), True, None, comm_backend="rccl")
    seg_buckets = [s.buckets for s in arena.segments]
    assert max(len(x) for x in seg_buckets) > 1          # the embedding table is cut into several buckets
    m.runtime.segment_done = lambda seg: [red.mark_ready(k, 1) for k in seg_buckets[seg]]
    opt = FusedAdamW(arena, lr=1e-3, weight_decay=0.01, scaler=LossScaler(torch.device("cuda"), enabled=dtype == torch.float16))
    took = opt.attach_reducer(red) if per_bucket else False
    for _ in range(steps):
        red.begin(True)
        m(**b).loss.backward()
        red.finish()
        applied = list(red.optimizer_applied())
        opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()
    return took, applied, [t.clone() for t in (arena.master, arena.p16, opt.m, opt.v, opt.step_count)]
took, applied, a = run(True)
assert took and sum(applied) >= len(applied) - 1 and sum(applied) >= 3, applied
took0, applied0, ref = run(False)
assert not took0 and not any(applied0)
for x, y in zip(a, ref):
    assert torch.equal(x, y)
assert int(a[4]) == 3
os.environ["PDT_ADAMW_PER_BUCKET"] = "0"
assert run(True, steps=1)[0] is False
del os.environ["PDT_ADAMW_PER_BUCKET"]
assert run(True, torch.float16, steps=1)[0] is False
print("ADAMW_PER_BUCKET_OK")
'''
    _run_snippet(code, "ADAMW_PER_BUCKET_OK")


def _torchrun_bench(extra, port=None, timeout=240, bare=False, world=2):
    """bench.py with two ranks: under the launcher (the driver's command line), or - `bare` - as plain
    ``python bench.py --gpus 2`` with no launcher environment, where the script starts its own rank processes."""
    import json
    import subprocess

    from pytorch_distributed_training_amd.launch import find_free_port

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tail = ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1"] + extra
    if bare:
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(port or find_free_port())] + tail
    env = dict(os.environ, PDT_HANG_DUMP="150")
    if bare:  # no launcher environment at all: that is the point
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                  "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_USE_AGENT_STORE"):
            env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads(lines[0])


def test_bench_two_ranks_share_the_gpu_data_parallel_and_pipeline():
    """bench.py under the launcher with two ranks on the one GPU (gloo rehearsal; multi-rank runs keep the persistent GEMM launches of
    the single-GPU run): data parallel (bucket reducer staged
    through the host) and DP x 2-stage pipeline in fp16 (overflow flag agreed across the stages)."""
    assert_mode = 1  # what bench.py must report as gemm_persistent_mode for a multi-rank run (the single-GPU default)
    dp = _torchrun_bench(["--model", "bert-base-cased", "--batch", "128"])
    assert dp["n_gpus"] == 2 and dp["config"]["parallelism"] == "dp2" and dp["value"] > 0
    assert dp["config"]["gemm_persistent_mode"] == assert_mode and dp["replicas_bit_equal"] is True
    assert dp["last_step_loss"] == dp["last_step_loss"] and dp["comm_exposed_ms_per_step"] is not None
    pp = _torchrun_bench(["--model", "bert-base-cased", "--batch", "256", "--pp", "2", "--micro-batches", "4"])
    assert pp["config"]["parallelism"].startswith("dp1 x pp2") and pp["value"] > 0
    assert pp["last_step_loss"] == pp["last_step_loss"] and pp["config"]["global_batch"] == 512


def test_bench_starts_its_own_ranks_without_a_launcher():
    """``python bench.py --gpus 2`` with no launcher environment (the reference's model-parallel script starts its own
    ranks the same way, /root/reference/test_model_parallelism.py:331-335): the parent never touches the GPU, runs
    torch.distributed.run with two fresh rank processes as a child and returns its status; one JSON line comes back."""
    out = _torchrun_bench(["--model", "bert-base-cased", "--batch", "128"], bare=True)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    assert out["config"]["parallelism"] == "dp2" and out["replicas_bit_equal"] is True


def test_bench_eight_ranks_share_the_gpu():
    """The driver's 8-rank command line (torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8) rehearsed with
    eight processes on the one GPU over gloo at BERT-base size: rank arithmetic, 8-way replica verification, the bucket
    list of eight ranks, the MAX over ranks of the timed region and the bit-equality check of all eight replicas."""
    out = _torchrun_bench(["--model", "bert-base-cased", "--batch", "32"], world=8, timeout=400)
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "dp8" and out["config"]["global_batch"] == 256
    assert out["value"] > 0 and out["replicas_bit_equal"] is True and out["last_step_loss"] == out["last_step_loss"]
