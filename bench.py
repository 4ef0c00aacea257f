#!/usr/bin/env python3
"""Headline benchmark: BERT-large sequence-classification fine-tuning throughput (samples/s, whole job).

The configuration is the reference's data-parallel launch (bert-large-cased, MRPC-shaped sentence pairs,
16-bit mixed precision, AdamW lr 2e-5 with linear warm-up, dropout 0.1;
/root/reference/test_data_parallelism.py:69-150,174) on synthetic token batches of seq 128 with random-init
weights.  Every timed step is a full optimizer step: forward, backward (with the bucketed gradient
all-reduce when N > 1), fused AdamW with dynamic loss scaling, LR schedule, zero_grad.  The per-GPU batch
defaults to 131072 tokens (1024 samples of 128 tokens) in one micro-batch (weak scaling: it stays fixed as N
grows); the reference's literal "96 = 12 x 8" accumulation layout is `--batch 96 --accum 12`.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W        # no launcher environment: starts N rank processes itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Started without a launcher environment and with ``--gpus N`` > 1, the script launches ``torch.distributed.run``
with N fresh rank processes as a child (the way the reference's model-parallel script spawns its own ranks,
/root/reference/test_model_parallelism.py:331-335) before anything has touched a GPU, and exits with the child's
status; rank 0 prints the one JSON line either way.  Multi-rank runs verify after the warm-up steps and again
after the timed steps that all data-parallel replicas hold bit-identical parameters and fail loudly otherwise.
"""
from __future__ import annotations

import argparse
import json
import os

# dmabuf IPC (the only mode the host driver of these nodes supports) for RCCL / cross-process GPU memory:
# must be in the environment before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if "--graph" in __import__("sys").argv:
    # graph replays want HIP's default of 4 hardware queues (16, the package default, makes them 2.4x slower: see
    # utils/graph.py); the count is fixed when the HIP runtime starts, i.e. before the package is imported
    os.environ.setdefault("PDT_HW_QUEUES", "0")
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from pytorch_distributed_training_amd.data.synthetic import fixed_batch  # noqa: E402
from pytorch_distributed_training_amd.engine import Engine  # noqa: E402
from pytorch_distributed_training_amd.models.bert import build_model  # noqa: E402
from pytorch_distributed_training_amd.optim import AdamW, get_linear_schedule_with_warmup  # noqa: E402


# samples/s of the reference's software stack (transformers + torch AMP, SDPA attention) on one MI355X: bert-large-cased,
# seq 128, batch 1024, fp16 - bench/eager_baseline.py; 2014 / 2018 / 1993 in three table runs (profiles/perf_table_*.jsonl)
REFERENCE_STACK_1GPU = 2014.0
POOL_BATCHES = 16  # distinct synthetic batches a run cycles through


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="bert-large-cased")
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--batch", type=int, default=None,
                    help="per-GPU samples per optimizer step; default 131072 tokens / seq_len (1024 at seq 128: ~105 GB of "
                         "activations, sized for 288 GB HBM, not for the reference's 8-sample micro-batches)")
    ap.add_argument("--accum", type=int, default=1, help="micro-batches per optimizer step")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--comm-dtype", default=None, choices=[None, "fp16", "bf16"])
    ap.add_argument("--comm", default=None, choices=[None, "c10d", "rccl"],
                    help="gradient all-reduce and pipeline send/recv through the framework's own RCCL communicator and "
                         "communication stream (rccl, default) or through torch's RCCL process group (c10d)")
    ap.add_argument("--comm-algo", default=None, choices=[None, "allreduce", "rs_ag"],
                    help="gradient buckets as one all-reduce (default) or as reduce-scatter + all-gather (own RCCL communicator)")
    ap.add_argument("--pp", type=int, default=1, choices=[1, 2],
                    help="2 = every data-parallel replica is a 2-stage layer-split pipeline over a pair of GPUs (1F1B over "
                         "send/recv); N GPUs = N/2 replicas.  The per-replica batch is 2 x --batch, so the work per GPU is "
                         "the same as in the pure data-parallel run")
    ap.add_argument("--micro-batches", type=int, default=8, help="pipeline micro-batches per optimizer step (--pp 2)")
    ap.add_argument("--rehearse-dp", action="store_true",
                    help="single GPU only: run the data-parallel machinery anyway (1-rank RCCL group, every bucket "
                         "all-reduce really issued on RCCL's stream) to see what the gradient path costs besides the wire")
    ap.add_argument("--gemm-persist", type=int, default=None, choices=[0, 1, 2],
                    help="persistent launches of the large GEMMs (256 workgroups draw the output tiles from ticket counters): "
                         "0 never (one workgroup per tile), 1 from 4 tiles per workgroup (default, single- and multi-rank), "
                         "2 from 2")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (all micro-batches + optimizer) in one hipGraph (single GPU only)")
    args = ap.parse_args()
    if os.environ.get("PDT_HANG_DUMP"):  # debugging aid: Python stacks of every thread after N seconds, then exit
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["PDT_HANG_DUMP"]), exit=True)
    if args.batch is None:
        args.batch = max(1, 131072 // args.seq_len)
    if args.comm_algo:
        os.environ["PDT_COMM_ALGO"] = args.comm_algo
    if args.gemm_persist is not None:
        os.environ["PDT_GEMM_PERSIST"] = str(args.gemm_persist)  # (read by the extension and honoured by Engine)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(_self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    comm = {None: None, "fp16": torch.float16, "bf16": torch.bfloat16}[args.comm_dtype]
    if args.pp == 2:
        return bench_pipeline(args, world, comm)
    if args.rehearse_dp:
        if world != 1:
            raise SystemExit("--rehearse-dp is a single-GPU rehearsal")
        os.environ["PDT_FORCE_COLLECTIVES"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from pytorch_distributed_training_amd.launch import find_free_port

        os.environ.setdefault("MASTER_PORT", str(find_free_port()))
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    eng = Engine(fp16=args.dtype == "fp16", bf16=args.dtype == "bf16", seed=42, comm_dtype=comm, comm_backend=args.comm)
    if eng.device.type != "cuda":
        raise SystemExit("bench.py needs a GPU")
    torch.manual_seed(42)
    model = build_model(args.model)
    optimizer = AdamW(params=model.parameters(), lr=2e-5, correct_bias=True)
    model, optimizer = eng.prepare(model, optimizer)
    if args.rehearse_dp:
        from pytorch_distributed_training_amd.parallel.dp import DataParallel

        model = DataParallel(model, model.arena, None, comm, comm_backend=args.comm)
        eng._models[-1] = model
        optimizer.optimizer.grads_checked_by_reducer = model.attach_scaler(eng.scaler)
        optimizer.optimizer.attach_reducer(model.reducer)  # (bf16: AdamW per bucket behind the all-reduces)
    sched = get_linear_schedule_with_warmup(optimizer, 100, 10000)
    model.train()
    micro = args.batch // args.accum
    vocab = eng.unwrap_model(model).config.vocab_size
    # a pool of fixed synthetic batches, cycled: the timed steps read no new data, so the loss reported at the end is
    # the loss of a batch the model has seen before (a long run memorises its pool) - a liveness signal, not convergence
    pool = [fixed_batch(micro, args.seq_len, vocab, eng.device, seed=1000 * eng.process_index + i,
                        first_id=eng.process_index * args.batch + (i % args.accum) * micro) for i in range(POOL_BATCHES)]

    if os.environ.get("PDT_BENCH_NO_SAMPLE_IDS"):  # A/B aid: rank-keyed dropout streams instead of sample-keyed ones
        for b in pool:
            b.pop("sample_ids", None)

    def step(i: int) -> torch.Tensor:
        if args.accum > 1:
            # one accumulation window: micro-batch i+1's forward is enqueued next to micro-batch i's backward
            # (Engine.accumulate; PDT_OVERLAP_MICRO=0 gives the plain loop), one gradient reduction at its end
            losses = eng.accumulate(model, [pool[(i * args.accum + a) % len(pool)] for a in range(args.accum)])
            loss = losses[-1] / args.accum
        else:
            out = model(**pool[i % len(pool)])
            loss = out.loss
            eng.backward(loss)
        optimizer.step()
        sched.step()
        optimizer.zero_grad()
        return loss

    if args.graph:
        from pytorch_distributed_training_amd.utils.graph import GraphedStep

        graphed = GraphedStep(eng, model, optimizer, warmup=2)

        def step(i: int) -> torch.Tensor:  # noqa: F811 - the same step, replayed from the captured graph
            loss = graphed([pool[(i * args.accum + a) % len(pool)] for a in range(args.accum)])
            sched.step()
            return loss / args.accum

    for i in range(max(args.warmup, 3 if args.graph else 0)):
        step(i)
    _fence(eng)
    _check_replicas(model)  # (outside the timed region)
    # two events per backward around the waits for the bucket collectives: reports how much of the gradient
    # communication backward did not hide (diagnostic only, nothing is synchronised inside the timed region)
    timed_dp = hasattr(model, "comm_exposed_ms")
    if timed_dp:
        model.measure_exposed = True
        model.comm_exposed_ms()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    _fence(eng)
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=eng.device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    exposed = model.comm_exposed_ms() if timed_dp else None
    exposed = round(exposed, 3) if exposed is not None else None
    replicas_equal = _check_replicas(model)
    global_batch = args.batch * world
    value = global_batch * args.steps / dt
    if eng.is_main_process:
        print(json.dumps({
            "metric": f"training samples/sec (node) {_pretty(args.model)} seq{args.seq_len}",
            "value": round(value, 2),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1000 * dt / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic MRPC-shaped token batches (random-init weights)",
            "last_step_loss": float(loss.item()) * args.accum,
            "loss_note": f"batches cycle through a pool of {POOL_BATCHES} fixed synthetic batches: not a convergence signal",
            "tokens_per_s": round(value * args.seq_len, 1),
            # model FLOPs (6 x linear-layer parameters + attention products, forward + backward) actually delivered per GPU
            "model_tflops_per_gpu": round(value / world * _flops_per_sample(eng.unwrap_model(model).config, args.seq_len) / 1e12, 1),
            "comm_exposed_ms_per_step": exposed,
            "rccl_ranks": model.rccl_ranks() if hasattr(model, "rccl_ranks") else 0,
            "replicas_bit_equal": replicas_equal,
            # not a published number (the reference has none, hence vs_baseline = null): the reference's software
            # stack measured on the same kind of GPU at this configuration (REFERENCE_STACK_1GPU above)
            "vs_reference_stack_1gpu": (round(value / REFERENCE_STACK_1GPU, 3)
                                        if (world == 1 and args.model == "bert-large-cased" and args.seq_len == 128
                                            and args.batch == 1024) else None),
            "config": {"model": args.model, "global_batch": global_batch, "per_gpu_batch": args.batch,
                       "micro_batches": args.accum, "seq_len": args.seq_len,
                       "parallelism": f"dp{world}" + (" (1-rank RCCL rehearsal)" if args.rehearse_dp else ""),
                       "comm_backend": getattr(model, "backend_label", "none"),
                       "comm_algo": os.environ.get("PDT_COMM_ALGO", "allreduce"),
                       "gemm_persistent_mode": _persist_mode(),
                       "hip_graph": bool(args.graph),
                       "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1) if eng.device.type == "cuda" else None,
                       "optimizer": "fused AdamW + dynamic loss scale" if args.dtype == "fp16" else "fused AdamW",
                       "adamw_per_bucket": getattr(optimizer.optimizer, "_bucket_reducer", None) is not None,
                       "dropout": 0.1},
        }))
    if dist.is_initialized():
        dist.barrier()
        _teardown()


def _self_launch(gpus: int) -> int:
    """`python bench.py --gpus N` outside a launcher: run N fresh rank processes under torch.distributed.run as a
    child of this process (which has not touched a GPU and never will) and hand its exit status back.  With fewer
    GPUs than ranks (a rehearsal on a one-GPU box) the ranks share the card over gloo, see launch.default_backend."""
    import subprocess

    from pytorch_distributed_training_amd.launch import find_free_port

    # Under a profiler that preloads itself into this process (rocprofv3 does, and with counter collection it starts the
    # GPU runtime before main() runs) this process HAS touched the GPU, and the launcher it would start re-executes
    # Python: profile multi-rank runs by putting the profiler in front of the rank processes instead.
    preload = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "") + os.environ.get("ROCPROFILER_LIBRARY", "")
    if "rocprof" in preload.lower() or "ROCPROFILER_SDK_TOOL" in "".join(os.environ.keys()):
        print("bench.py --gpus N > 1 does not start its own ranks under a profiler: run\n  python -m torch.distributed.run "
              "--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...\nand put the profiler "
              "inside (e.g. profile one rank), or profile a single-GPU run.", file=sys.stderr)
        return 2

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(find_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.run(cmd).returncode


def _check_replicas(model):
    """True after every data-parallel replica was verified to hold bit-identical parameters (raises otherwise);
    None when the model is not data parallel."""
    if not hasattr(model, "check_replicas"):
        return None
    model.check_replicas()
    return True


def _teardown() -> None:
    """End of a multi-rank run: the result line is out; leave without running the interpreter's exit-time destructors
    (communicator worker threads torn down in arbitrary order at exit can turn a finished run into a non-zero exit)."""
    sys.stdout.flush()
    sys.stderr.flush()
    torch.cuda.synchronize()
    multi = dist.get_world_size() > 1
    if multi:
        # the measurement is complete and printed: a teardown handshake that does not return (a peer already gone) must
        # not turn the run into a hang - leave after 30 s whatever destroy_process_group() is doing
        import threading

        t = threading.Timer(30.0, lambda: os._exit(0))
        t.daemon = True
        t.start()
    dist.destroy_process_group()
    if multi:  # (a 1-rank rehearsal exits normally: a profiler attached to it writes its output at exit)
        os._exit(0)


def bench_pipeline(args, world: int, comm) -> None:
    """DP x 2-stage pipeline (the layout of /root/reference/test_model_parallelism.py: data parallel around a
    model split over two devices) - rank = replica * 2 + stage, one GPU per rank."""
    from pytorch_distributed_training_amd import launch
    from pytorch_distributed_training_amd.models.config import preset
    from pytorch_distributed_training_amd.parallel.groups import ParallelMesh
    from pytorch_distributed_training_amd.parallel.pp import PipelineStage, build_stage

    if world % 2:
        raise SystemExit("--pp 2 needs an even number of ranks")
    env, device = launch.init_distributed()
    mesh = ParallelMesh(world, env.rank, pp=2).build_groups()
    # (update_in_backward=False, like the pipeline recipe: the optimizer runs after the schedule's last backward)
    eng = Engine(fp16=args.dtype == "fp16", bf16=args.dtype == "bf16", seed=42, comm_dtype=comm, comm_backend=args.comm,
                 process_group=mesh.dp_group, update_in_backward=False)
    if eng.device.type != "cuda":
        raise SystemExit("bench.py needs a GPU")
    model_cfg = preset(args.model)
    stage = build_stage(model_cfg, mesh.stage, 2, seed=42)
    optimizer = AdamW(params=stage.parameters(), lr=2e-5, correct_bias=True)
    stage, optimizer = eng.prepare(stage, optimizer)
    sched = get_linear_schedule_with_warmup(optimizer, 100, 10000)
    pipe = PipelineStage(stage, mesh, eng, micro_batches=args.micro_batches)
    replica_batch = 2 * args.batch
    # both stage ranks of a replica read the same batches (only hidden states cross the link)
    pool = [fixed_batch(replica_batch, args.seq_len, model_cfg.vocab_size, eng.device, seed=1000 * mesh.replica + i,
                        first_id=mesh.replica * replica_batch) for i in range(POOL_BATCHES)]
    loss = None

    def step(i: int):
        out = pipe.train_step(pool[i % len(pool)])
        pipe.optimizer_step(optimizer, sched)
        return out

    for i in range(args.warmup):
        step(i)
    dist.barrier()
    torch.cuda.synchronize()
    _check_replicas(stage)
    timed_dp = hasattr(stage, "comm_exposed_ms")
    if timed_dp:
        stage.measure_exposed = True
        stage.comm_exposed_ms()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    exposed = stage.comm_exposed_ms() if timed_dp else None
    replicas_equal = _check_replicas(stage)
    t = torch.tensor([dt], dtype=torch.float64, device=eng.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    box = [float(loss.item()) if (loss is not None and mesh.replica == 0) else None]
    dist.broadcast_object_list(box, src=mesh.peer(1) if mesh.replica == 0 else 1)
    global_batch = replica_batch * mesh.dp
    value = global_batch * args.steps / dt
    if env.rank == 0:
        print(json.dumps({
            "metric": f"training samples/sec (node) {_pretty(args.model)} seq{args.seq_len}",
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic MRPC-shaped token batches (random-init weights)",
            "last_step_loss": box[0],
            "loss_note": f"batches cycle through a pool of {POOL_BATCHES} fixed synthetic batches: not a convergence signal",
            "tokens_per_s": round(value * args.seq_len, 1),
            "model_tflops_per_gpu": round(value / world * _flops_per_sample(model_cfg, args.seq_len) / 1e12, 1),
            "comm_exposed_ms_per_step": round(exposed, 3) if exposed is not None else None,
            "rccl_ranks": stage.rccl_ranks() if hasattr(stage, "rccl_ranks") else 0,
            "pipeline_p2p": "rccl" if pipe.comm is not None else "c10d:" + dist.get_backend(),
            "replicas_bit_equal": replicas_equal,
            "config": {"model": args.model, "global_batch": global_batch, "per_replica_batch": replica_batch,
                       "micro_batches": args.micro_batches, "seq_len": args.seq_len,
                       "parallelism": f"dp{mesh.dp} x pp2 (1F1B)", "backend": dist.get_backend(),
                       "comm_backend": getattr(stage, "backend_label", "none"),
                       "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                       "optimizer": "fused AdamW + dynamic loss scale", "dropout": 0.1},
        }))
    dist.barrier()
    _teardown()


def _flops_per_sample(cfg, seq: int) -> float:
    """Training FLOPs of one sample of `seq` tokens: 6 per linear-layer parameter and token (forward + both backward
    GEMMs), 12 * seq * hidden per token and layer for the two attention products, the pooler once per sample."""
    L, H, F = cfg.num_hidden_layers, cfg.hidden_size, cfg.intermediate_size
    linear = L * (4 * H * H + 2 * H * F)
    return seq * (6.0 * linear + 12.0 * L * seq * H) + 6.0 * H * H


def _persist_mode() -> int:
    from pytorch_distributed_training_amd.ops import native

    return int(native().gemm_persistent_mode())


def _pretty(model: str) -> str:
    return {"bert-large-cased": "BERT-large", "bert-base-cased": "BERT-base", "roberta-large": "RoBERTa-large",
            "roberta-base": "RoBERTa-base"}.get(model, model)


def _fence(eng: Engine) -> None:
    if eng.num_processes > 1:
        dist.barrier()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
