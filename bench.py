#!/usr/bin/env python
"""bench.py — images/sec of the CenterNet hot path (forward + gather_detection2d [+ RCCL all-gather]) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 either launched under
`python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU, RCCL) or started plainly — without WORLD_SIZE in the environment
the script launches those N ranks itself (`self_launch`) and exits non-zero if any rank fails.  Rank 0 prints ONE JSON line.

  step      = one pass of the hot path over one synthetic batch per GPU:
              CenterNet.forward (ResNet-34 -> neck -> heads, sigmoid) + gather_detection2d (k=100, nms 3)
              [+ the all-gather of the packed detections when N > 1: started on a side stream after the decode and collected one
              step later (collate.Collator), i.e. overlapped with the next batch's forward; the last one is drained inside the
              timed region].  Inputs are resident in HBM.
  workload  = BASELINE.json configs[1] (C1): ResNet34 + simple upsample neck, batch 32 per GPU, 512x512, 80 classes
              (`--config fpn --batch 64` = C2/C3, `--config tracking --batch 32 --height 608 --width 1088` = C4).  The default run
              appends short lines of the other BASELINE configurations under `also` (10 steps after 3 warm-ups each; `also_workloads`):
              at N = 1  C2 (FPN, 64 images) and C4's per-GPU share (tracking, 32 x 608x1088);
              at N > 1  C3 (FPN, 64 images per GPU: N = 8 is BASELINE's 512-image job) and C4 (tracking, 32 per GPU x 608x1088: N = 8 is
              its 256-image job) — every rank runs them, sharded like the main line, each with its RCCL all-gather and `collate_ms`.
  value     = total images / max-over-ranks wall time of the K timed steps (weak scaling: batch per GPU fixed).
  roofline  = the dominant kernel by time, measured with HIP events on the launch stream around every conv launch of one step.
              The long 3x3 layers form every fp32 product on the fp16 matrix cores (scaled two-way fp16 split, three MFMAs per
              product, fp32 accumulation) as row-Winograd F(2,3) (`cnl_wino9` / `cnl_wino10`) or 2-D F(2x2,3x3) (`cnl_wino5/6`); peak =
              2.5 PFLOP/s dense fp16.  `achieved` counts the matrix-core flops
              the kernel EXECUTES, per launch (direct-conv flops x 6/9 [row F(2,3)] or 16/36 [F(2x2)], x 3 for the split; x 2/3 again for a row-Winograd
              launch behind a folded upsample that runs on row-pair weights, cnl_conv_params.w_up: 96 of 144 MFMAs per chunk), so `frac` is an honest
              hardware fraction — Winograd trades executed flops for transform work, which is why `effective_tflops` (the same time
              against the direct-conv, i.e. algorithmic, flops) is reported beside it.  `traffic` is NOT measured in this run: it is
              the rocprofv3 PMC figure of the named profiles/ file (null where no profile of that configuration exists).
  variants  = the same job in the other arithmetic classes of the plan (KernelOptions.algo; in-process, short runs):
              f32 = fp32 matrix cores only; f43 = "auto" with KernelOptions(f43=True): the 256-channel head blocks on the F(4,3) row kernel
              (csrc/winograd13.hip, CNL_ALGO_F43: 2.4-4.6 x the fp32 matrix core's rounding error — an opt-in class, see `accuracy`).
  accuracy  = max |feature - float64 oracle| / max |float64 oracle| at the neck output and at each head's last 256-channel block
              output (what out_conv reads), for auto / f32 / f43 and for the CPU fp32 oracle itself, on 2 images of the bench shape.
              (The post-sigmoid heatmap hides feature error by ~3 orders of magnitude; it is reported too.)  Backbone / ConvBnAct
              parity is "unpinned" by the reference itself (torchvision / vision_toolbox absent): the oracle is this repo's restatement.
  decode    = decode p50 on the forward's own outputs: bytes that must move, GB/s, fraction of 8 TB/s; with a separate sigmoid pass
              in front (what a caller holding logits pays) and without (the path: sigmoid is the heatmap conv's epilogue).
  cpu_baseline = the CPU oracle (oracle/ref_cpu.py + oracle/decode_ref.py: the plain PyTorch restatement of the reference path —
              kind "port") timed on this box's host cores: C0 exactly (1x3x512x512) and the bench shape at N = 32 (the GPU leg's batch),
              in child processes under two OpenMP placements (runtime default / OMP_PROC_BIND=close OMP_PLACES=cores) and 16 / 32 / 64
              threads; 1 warm-up + 3 timed passes of the best, median; with --cpu-multi also SEVERAL oracle processes at once on disjoint
              core sets (one per NUMA node, one per 16 cores), throughput summed — measured once per round, it does not beat one process
              on the GPU box's host (profiles/r04_bench_c1_with_cpu_multi.json) and costs minutes; `value` = the best of the legs that ran,
              `cores` = the cores that leg used; `host` = affinity mask size, cgroup CPU quota, NUMA nodes.  Decode p50 on the CPU beside the GPU's.  Rank 0, N = 1 only, AFTER all GPU
              legs (so that the GPU work of the run is contiguous).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "centernet-lightning_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import centernet_lightning_amd as cl  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: dense fp16 / bf16 (v_mfma_f32_32x32x16_f16)
HBM_PEAK_GBPS = 8000.0            # same guide: HBM3E spec peak (6.29 TB/s measured float4 copy)
# rocprofv3 figures this line quotes but does not measure itself (HBM bytes per launch from the PMC passes: FETCH_SIZE x 2 + WRITE_SIZE, see
# tools/rocpd_summary.py json; kernel-trace averages of the decode kernels): read from the committed profile JSON of the same command, so a
# quoted number is byte-equal to a field of that file.  Written on the GPU box by tools/profile_round.sh.
PROFILE_JSON = {("simple", 32, 512, 512): "profiles/r06_profile_c1.json", ("fpn", 64, 512, 512): "profiles/r06_profile_c2.json",
                ("tracking", 32, 608, 1088): "profiles/r06_profile_c4.json"}
# kernel-name PREFIX in the profile JSON (template variants of one kernel — with / without a residual — are combined, weighted by calls)
KIND_KERNEL = {"winograd_row_f16x2": "cnl_wino9::winograd9_kernel", "winograd_row4_f16x2": "cnl_wino10::winograd10_kernel", "winograd_row_f43_f16x2": "cnl_wino13::winograd13_kernel",
               "winograd_f16x2": "cnl_wino5::winograd5_kernel", "winograd_f32": "cnl_wino2::winograd2_kernel"}


def profiled(config, B, H, W):
    path = PROFILE_JSON.get((config, B, H, W))
    if not path or not os.path.exists(os.path.join(ROOT, path)):
        return None, path
    return json.load(open(os.path.join(ROOT, path))), path


def profile_matches_sources(prof):
    """Was the committed profile taken from the kernel sources of this tree?  (meta.sources_sha16, written by tools/profile_round.sh; profiles of rounds 1-5 carry none: None.)"""
    want = (prof or {}).get("meta", {}).get("sources_sha16")
    return None if want is None else (want == sources_sha16())


CONFIGS = {"simple": "resnet34_simple.yaml", "fpn": "resnet34_fpn.yaml", "tracking": "tracking_resnet34_fpn.yaml"}
KIND_NAMES = {"winograd_row4_f16x2": "cnl_wino10::winograd10_kernel (the row-Winograd arithmetic of cnl_wino9 on 4-row x 64-pixel x 64-cout work items, two "
                                     "workgroups per CU)",
              "winograd_f16x2": "cnl_wino5::winograd5_kernel / cnl_wino6::winograd6_kernel (Winograd F(2x2,3x3); the same split arithmetic)",
              "winograd_row_f16x2": "cnl_wino9::winograd9_kernel (1-D Winograd F(2,3) along x, the three kernel rows in the reduction: 6 of the direct conv's 9 multiplies "
                                    "per output; fp32 operands scaled per image / per output channel by a power of two and split into 2 fp16 pieces, "
                                    "3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulate)",
              "winograd_row_f43_f16x2": "cnl_wino13::winograd13_kernel (1-D Winograd F(4,3) along x, the three kernel rows in the reduction: 4.5 of the direct conv's 9 multiplies "
                                        "per output, the same split arithmetic; the opt-in class KernelOptions(f43=True) / CNL_ALGO_F43)",
              "winograd_f32": "cnl_wino2::winograd2_kernel (Winograd F(2x2,3x3), fp32 v_mfma_f32_32x32x2_f32)",
              "direct_f16x2": "cnl_conv::conv_f16x2_kernel (direct implicit GEMM, fp16 matrix cores, scaled two-way split)",
              "direct": "cnl_conv::conv_mfma_kernel (direct implicit GEMM, fp32 v_mfma_f32_32x32x2_f32)",
              "fused_out_reduce": "cnl_fused::reduce_kernel (fixed-order sum of the per-32-channel partial sums of a 1x1 out_conv folded into the row-Winograd epilogue; HBM-bound)"}
# executed matrix flops / direct-conv flops, and the peak they run against
EXEC = {"winograd_row4_f16x2": (6.0 / 9.0 * 3.0, F16_MFMA_PEAK_TFLOPS), "winograd_f16x2": (16.0 / 36.0 * 3.0, F16_MFMA_PEAK_TFLOPS),
        "winograd_row_f16x2": (6.0 / 9.0 * 3.0, F16_MFMA_PEAK_TFLOPS), "winograd_row_f43_f16x2": (4.5 / 9.0 * 3.0, F16_MFMA_PEAK_TFLOPS),
        "winograd_f32": (16.0 / 36.0, FP32_MFMA_PEAK_TFLOPS), "direct_f16x2": (3.0, F16_MFMA_PEAK_TFLOPS), "direct": (1.0, FP32_MFMA_PEAK_TFLOPS),
        "fused_out_reduce": (0.0, FP32_MFMA_PEAK_TFLOPS)}      # (the reduce only adds partial sums: no matrix flops — reported against the HBM peak below; the folded 1x1 conv's multiplies run in winograd9's epilogue on the vector ALUs)


def synthetic_weights_(model, seed=0):
    """Random-init weights of the named architecture (no checkpoints offline).  Convs: Kaiming fan_out (reference
    layers.py:77).  BN affine/stats randomised so folding is exercised; the residual-branch BN (bn2) gets a small gamma
    so activations stay O(1) through 16 residual blocks without a calibration pass.  out_conv: N(0, 0.01^2), biases
    keep init_bias (-2.19 / 10 / 0)."""
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    for k, v in sd.items():
        if k.endswith("running_var"):
            base = k[: -len("running_var")]
            lo, hi = (0.15, 0.35) if base.endswith("bn2.") else (0.7, 1.3)
            sd[base + "weight"].copy_(torch.rand(v.shape, generator=g) * (hi - lo) + lo)
            sd[base + "bias"].copy_(torch.randn(v.shape, generator=g) * 0.1)
            sd[base + "running_mean"].copy_(torch.randn(v.shape, generator=g) * 0.1)
            v.copy_(torch.rand(v.shape, generator=g) * 0.4 + 0.8)
        elif k.endswith("out_conv.weight"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.01)
        elif v.dim() == 4:
            fan_out = v.shape[0] * v.shape[2] * v.shape[3]
            v.copy_(torch.randn(v.shape, generator=g) * (2.0 / fan_out) ** 0.5)
    model.load_state_dict(sd)
    return model


def build_model(config, **options):
    torch.manual_seed(0)
    model = synthetic_weights_(cl.build_centernet(os.path.join(ROOT, "centernet-lightning_amd", "configs", CONFIGS[config])))
    if options:
        model.set_kernel_options(**options)
    return model.cuda()


def setup_distributed(gpus):
    """(rank, world, local_rank) from the torchrun environment; one rank per GPU over RCCL ("nccl" IS RCCL on ROCm).
    CNL_BENCH_BACKEND=gloo is a functional check of the N>1 flow on a box with fewer GPUs than ranks (ranks then share devices — or have
    none at all in the CPU test of tests/test_host.py — and the collective goes through the host); the measured configuration is always
    nccl = RCCL, one GPU per rank."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != gpus:                    # (a plain `bench.py --gpus N` never gets here: main() starts the ranks itself, `self_launch`)
        raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}")
    backend = os.environ.get("CNL_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > 1 and torch.cuda.is_available() and torch.cuda.device_count() < world:
        # RCCL needs one device per rank; two ranks on one GPU end in an abort deep inside the communicator set-up
        raise SystemExit(f"bench.py --gpus {gpus}: {torch.cuda.device_count()} GPU(s) visible to rank {rank} — the RCCL (nccl) backend needs one device per rank; "
                         f"for a functional run of the N > 1 flow on fewer devices set CNL_BENCH_BACKEND=gloo (its line says so: collective.backend)")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))     # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def self_launch(gpus, argv):
    """`python bench.py --gpus N` with N > 1 and NO launcher environment (WORLD_SIZE unset): start the N ranks ourselves — the same
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <argv>` the driver's
    contract names, on a free port — pass rank 0's one JSON line through on stdout and return the launcher's exit code (non-zero if ANY
    rank failed).  Under torchrun (WORLD_SIZE set) this is never called: that path is unchanged."""
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


class _StubModel:
    """CNL_BENCH_STUB=1 — the launcher self-test of tests/test_host.py on a box WITHOUT GPUs: stands in for the model leg only (a "forward" that
    hands back host records), so that everything around it — self_launch, the rendezvous, barriers, the pipelined Collator over gloo, max over
    ranks, rank 0's single JSON line, the exit code — runs for real.  Its line says `"data": "STUB ..."`: never a measurement."""

    def __init__(self, rank):
        self.rank = rank

    def __call__(self, x):
        return x

    def gather_detection2d(self, o, num_detections=100):
        return torch.full((o.shape[0], num_detections, 6), float(self.rank))

    gather_tracking2d = gather_detection2d


class _StubCollator:
    """records in, records out through the product's Collator (submit_records / result_records: the gloo path of the protocol tests)."""

    def __init__(self):
        self.c = cl.Collator()

    def submit(self, rec):
        return self.c.submit_records(rec)

    def result(self, h):
        return self.c.result_records(h)


def stub_main(args, rank, world):
    if os.environ.get("CNL_BENCH_STUB_FAIL_RANK") == str(rank):      # (the self-test's "one rank dies" case: the launcher must report it)
        raise SystemExit(f"rank {rank}: failing on request (CNL_BENCH_STUB_FAIL_RANK)")
    x = torch.zeros(args.batch, 1)
    barrier = dist.barrier if world > 1 else (lambda: None)
    collator = _StubCollator()
    model = _StubModel(rank)
    elapsed = timed(model, x, False, args.k, args.warmup, args.steps, collator, barrier)
    out = run_steps(model, x, False, args.k, 1, collator)
    ok = tuple(out.shape) == (world * args.batch, args.k, 6) and all(bool((out[q * args.batch:(q + 1) * args.batch] == q).all()) for q in range(world))
    coll = collective_info("cpu")
    par = parallelism_text(world)
    if world > 1:
        elapsed = max_over_ranks(elapsed, "cpu")
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(f"rank {rank}: gathered records are not in rank order")
    if rank == 0:
        print(json.dumps({"metric": "images/sec @512x512 ResNet34 CenterNet forward + gather_detection2d", "value": round(job_throughput(args.batch, world, args.steps, elapsed), 2),
                          "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "STUB (CNL_BENCH_STUB=1: launcher / collate self-test without a GPU — NOT a measurement)",
                          "collective": coll,
                          "config": {"workload": "stub", "global_batch": world * args.batch, "per_gpu_batch": args.batch, "parallelism": par}}), flush=True)


def collective_info(device):
    """What the communicator of THIS run was (the N > 1 line must prove an RCCL run, not assert it): backend, RCCL version, the world size the
    process group reports, and every rank's device — ordinal, name and PCI bus id, all-gathered — so that "8 ranks on 8 distinct devices" is a field of the line.
    Every rank calls this (it is a collective); world size 1: no process group, no collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1, "note": "one rank: no process group, collate is a no-op (reference eval/coco.py:11-13)"}
    backend = dist.get_backend()
    mine = {"rank": dist.get_rank(), "device": None, "pci_bus_id": None, "name": None, "host": os.uname().nodename, "pid": os.getpid()}
    if torch.cuda.is_available():
        d = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(d)
        mine.update(device=d, name=pr.name, pci_bus_id=getattr(pr, "pci_bus_id", None) if not hasattr(pr, "pci_domain_id") else f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}")
        if isinstance(mine["pci_bus_id"], int):
            mine["pci_bus_id"] = f"{mine['pci_bus_id']:02x}"
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:          # noqa: BLE001 — reported, never fatal
            ver = repr(e)
    ids = [(e["host"], e["pci_bus_id"] if e["pci_bus_id"] is not None else e["device"]) for e in everyone]
    return {"backend": backend, "rccl_version": ver, "world_size": dist.get_world_size(), "devices": everyone,
            "distinct_devices": len(set(ids)) if all(i[1] is not None for i in ids) else 0,
            "note": "backend 'nccl' IS RCCL on ROCm (collectives over xGMI between the GPUs of one node); 'gloo' = a functional run through the host, never a measurement of the collective"}


def parallelism_text(world):
    if world <= 1:
        return "batch-shard x1"
    backend = dist.get_backend() if dist.is_initialized() else "none"
    how = "RCCL all-gather of detections over xGMI" if backend == "nccl" else f"{backend} all-gather of detections through the host (FUNCTIONAL run, not RCCL)"
    return f"batch-shard x{world} + {how} (side stream, one step behind)"


def sources_sha16():
    """First 16 hex digits of the SHA-256 over csrc/*.hip, csrc/*.h and include/*.h (sorted by name): what a committed profile was taken from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "centernet-lightning_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "centernet-lightning_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def max_over_ranks(elapsed, device):
    """The slowest rank's wall time of the timed region: what the whole job took."""
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(images_per_rank, world, steps, elapsed_max):
    """Whole-job images/s: every rank processes its own batch per step (weak scaling), the job ends when the slowest rank does."""
    return images_per_rank * world * steps / elapsed_max


def run_steps(model, x, tracking, k, steps, collator):
    """`steps` passes of the hot path; the all-gather of step i is collected during step i+1 and the last one drained."""
    pending, out = None, None
    for _ in range(steps):
        o = model(x)
        dets = model.gather_tracking2d(o, num_detections=k) if tracking else model.gather_detection2d(o, num_detections=k)
        h = collator.submit(dets)
        if pending is not None:
            out = collator.result(pending)
        pending = h
    if pending is not None:
        out = collator.result(pending)
    return out


def timed(model, x, tracking, k, warmup, steps, collator, barrier):
    with torch.no_grad():
        run_steps(model, x, tracking, k, warmup, collator)
        barrier()
        t0 = time.perf_counter()
        run_steps(model, x, tracking, k, steps, collator)
        barrier()
        return time.perf_counter() - t0


def conv_kernel_profile(model, x, reps=3):
    """HIP-event timing of every conv launch of one forward (events on torch's current stream == the launch stream).
    Returns per-launch rows (what, algorithmic flops, ms, kind, algorithmic bytes, EXECUTED matrix flops), scaled to the whole batch."""
    import ctypes
    eng = model._engine
    model(x)                                            # make sure the plan exists
    # batches whose activations would exceed the kernels' 4 GiB addressing run as equal sub-batches (Engine.forward): time one and scale
    n_sub = eng.sub_batch(x.shape[0], x.shape[2], x.shape[3])
    scale = x.shape[0] / n_sub
    x = x[:n_sub]
    plan = eng.plan_for(x, sigmoid=True)
    lib = plan.lib
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    convs = [L for L in plan.launches if L.fn in (lib.cnl_conv2d_nhwc_f32, lib.cnl_conv3x3_winograd_f32, lib.cnl_conv3x3_up2_nhwc_f32, lib.cnl_fused_out_reduce_f32)]
    acc = [0.0] * len(convs)
    for _ in range(reps):
        torch.cuda.synchronize()
        evs = []
        if plan.absmax is not None:
            plan.absmax.zero_()
        for L in plan.launches:                         # replay the WHOLE plan in order (buffers are reused by liveness), timing the convs
            e0 = e1 = None
            if L in convs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rc = plan.launch(L, x, stream)
            assert rc == 0, L.what
            if e0 is not None:
                e1.record()
                evs.append((e0, e1))
        torch.cuda.synchronize()
        for i, (e0, e1) in enumerate(evs):
            acc[i] += e0.elapsed_time(e1)

    def kind(L):
        if L.fn is lib.cnl_conv3x3_up2_nhwc_f32:        # four sub-pixel phase convs on the direct kernels
            return "direct_f16x2" if lib.cnl_conv3x3_up2_kernel(ctypes.byref(L.args)) == 5 else "direct"
        if L.fn is not lib.cnl_conv3x3_winograd_f32:
            return "direct_f16x2" if lib.cnl_conv2d_kernel(ctypes.byref(L.args)) == 5 else "direct"
        return {5: "winograd_f16x2", 6: "winograd_f16x2", 9: "winograd_row_f16x2", 10: "winograd_row4_f16x2", 11: "winograd_row4_f16x2", 13: "winograd_row_f43_f16x2"}.get(lib.cnl_conv3x3_winograd_variant(ctypes.byref(L.args)), "winograd_f32")

    rows = []
    for i, L in enumerate(convs):
        p = L.args
        if L.fn is lib.cnl_fused_out_reduce_f32:        # second half of an out_conv folded into the 3x3 block before it: [part, blocks, M, C2, ...]
            # HBM-bound: it reads the [blocks][M][4] partial sums and writes [M][C2]; its "flops" are the folded 1x1 conv's ALGORITHMIC flops (counted once, here,
            # for the conv stack's total) and it executes no matrix instruction
            fl_r = 2.0 * p[2] * p[1] * 32 * p[3] * scale
            rows.append((L.what, fl_r, acc[i] / reps * scale, "fused_out_reduce", (p[1] * p[2] * 16 + p[2] * p[3] * 4) * scale, 0.0))
            continue
        up_in = 2 if p.flags & 4 else 1
        ho = (p.H_in * up_in + 2 * p.pad - p.KH) // p.stride + 1
        wo = (p.W_in * up_in + 2 * p.pad - p.KW) // p.stride + 1
        out_px = p.N * ho * wo * (4 if p.flags & 8 else 1)
        # algorithmic HBM bytes of a conv launch: input + weights + bias + output (+ residual), each touched once
        nbytes = 4 * (p.N * p.H_in * p.W_in * p.Cin + p.Cout * p.KH * p.KW * p.Cin + p.Cout + out_px * p.Cout + (out_px * p.Cout if p.residual else 0))
        kd = kind(L)
        ratio = EXEC[kd][0]
        if kd == "winograd_row_f16x2" and L.fn is lib.cnl_conv3x3_winograd_f32 and p.w_up and (p.flags & 4) and not p.residual and not p.fuse_w:
            ratio *= 2.0 / 3.0                          # row-pair weights behind a folded upsample (cnl_conv_params.w_up): 96 of the 144 MFMAs per chunk are issued
        rows.append((L.what, L.flops * scale, acc[i] / reps * scale, kd, nbytes * scale, L.flops * scale * ratio))
    return rows, plan


def roofline_block(rows, config, B, H, W):
    agg = {}
    for what, fl, ms, kd, nb, fx in rows:
        a = agg.setdefault(kd, [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += fl; a[3] += nb; a[4] += fx
    dom = max(agg, key=lambda k_: agg[k_][1])
    n, ms, fl, nb, fx = agg[dom]
    ratio, peak = EXEC[dom]
    exec_tf = fx / (ms * 1e-3) / 1e12
    prof, prof_path = profiled(config, B, H, W)
    pks = {k_: v for k_, v in (prof or {}).get("kernels", {}).items() if dom in KIND_KERNEL and k_.startswith(KIND_KERNEL[dom]) and "hbm_MB_per_launch" in v}
    traffic = None
    if pks:      # all template variants of the dominant kernel, weighted by their launch counts: the same population as avg_launch_us
        calls = sum(v["calls"] for v in pks.values())
        mb = sum(v["calls"] * v["hbm_MB_per_launch"] for v in pks.values()) / calls
        us = sum(v["calls"] * v["avg_us"] for v in pks.values()) / calls
        traffic = (round(mb * 1e6), f"{prof_path} [kernels][{' + '.join(sorted(pks))}]: hbm_MB_per_launch and avg_us weighted by calls over {calls} launches "
                                    f"(all launches of the kernel in the profiled steps; rocprofv3 average duration there {us:.2f} us)")
    roof = {"bound": "mfma", "kernel": KIND_NAMES[dom], "achieved": round(exec_tf, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(exec_tf / peak, 4),
            "achieved_counts": f"EXECUTED matrix-core flops = direct-conv flops x {ratio:.4f} (Winograd multiplies x split terms; x {ratio * 2 / 3:.4f} for the launches behind a folded "
                               f"upsample that run on row-pair weights, cnl_conv_params.w_up: 96 of 144 MFMAs per chunk) = {fx / fl:.4f} x over this kernel's launches; the algorithmic rate is effective_tflops",
            "effective_tflops": round(fl / (ms * 1e-3) / 1e12, 2),
            "effective_frac_of_fp32_mfma_peak": round(fl / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 3),
            "launches_per_step": n, "kernel_ms_per_step": round(ms, 3), "algorithmic_gflop_per_step": round(fl / 1e9, 2),
            "avg_launch_us": round(ms * 1e3 / n, 2),
            "algorithmic_bytes_per_launch": round(nb / n),
            "traffic": traffic[0] if traffic else None,
            "traffic_source": (traffic[1] + " — rocprofv3 PMC of an earlier run of this command, not measured here") if traffic else
                              "no PMC profile of this kernel / configuration committed yet (see profiles/)",
            "profile_matches_sources": profile_matches_sources(prof),
            "sustained_clock_note": "power-limited DVFS: on the 256-channel head blocks this kernel holds 1.37-1.48 GHz of 2.4 under real data (GRBM_GUI_ACTIVE / duration, profiles/r04_winograd_variants.txt, r04_winograd9_skip4.txt): a denser schedule lowers the clock, 25 % fewer MFMAs buy 12.6 % (DESIGN.md 3.1); a loop of nothing but this MFMA on random fp16 operands sustains 1.84-1.90 PFLOP/s at 1.84-1.90 GHz (profiles/r04_mfma_order.txt), 0.74-0.76 of `peak`"}
    def other(k_, v):
        if k_ == "fused_out_reduce":      # HBM-bound: bytes / s against the HBM peak, no executed matrix flops
            gbps = v[3] / (v[1] * 1e-3) / 1e9 if v[1] else 0.0
            return {"launches_per_step": v[0], "kernel_ms_per_step": round(v[1], 3), "bound": "hbm", "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / HBM_PEAK_GBPS, 4),
                    "executed_matrix_flops": 0}
        return {"launches_per_step": v[0], "kernel_ms_per_step": round(v[1], 3), "effective_tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2) if v[1] else 0.0,
                "executed_frac_of_its_peak": round(v[4] / (v[1] * 1e-3) / 1e12 / EXEC[k_][1], 4) if v[1] else 0.0}
    roof["other_kernels"] = {KIND_NAMES[k_].split(" (")[0]: other(k_, v) for k_, v in agg.items() if k_ != dom}
    conv_ms, conv_fl = sum(r[2] for r in rows), sum(r[1] for r in rows)
    stack = {"algorithmic_gflop_per_step": round(conv_fl / 1e9, 2), "kernel_ms_per_step": round(conv_ms, 3),
             "effective_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2)}
    return roof, stack


def p50_ms(fn, reps=30, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    lat = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        lat.append(e0.elapsed_time(e1))
    lat.sort()
    return lat[len(lat) // 2]


def gpu_ms_back_to_back(fn, calls=50, rounds=5):
    """GPU time per call: `calls` calls enqueued without a host sync between HIP events (the host runs ahead, as inside a step), best round."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(calls):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / calls)
    return best


def decode_profiled(config, B, H, W):
    prof, path = profiled(config, B, H, W)
    if not prof:
        return None
    out = {"source": f"{path} — rocprofv3 kernel-trace / PMC of an earlier run of this command, not measured by this run", "profile_matches_sources": profile_matches_sources(prof)}
    for name, k in prof["kernels"].items():
        if name.startswith("cnl_decode::"):
            out[name] = {f: k[f] for f in ("avg_us", "hbm_MB_per_launch", "calls") if f in k}
    return out


def decode_block(model, x, tracking, k, config=None):
    with torch.no_grad():
        out = model(x)
        logits = model.get_encoded_outputs(x)
        gather = model.gather_tracking2d if tracking else model.gather_detection2d
        p_wo = p50_ms(lambda: gather(out, num_detections=k))
        g_wo = gpu_ms_back_to_back(lambda: gather(out, num_detections=k))
        rest = tuple(out[1:])
        p_w = p50_ms(lambda: gather((torch.sigmoid(logits["heatmap"]),) + rest, num_detections=k))
    N, C, h, w = out[0].shape
    E = out[2].shape[1] if tracking else 0
    must = N * (4 * C * h * w + k * (16 + 4 * E) + k * (4 + 8 + 8 + 16 + 4 * E))      # heatmap once + k gathers + outputs
    gbps = must / (g_wo * 1e-3) / 1e9
    return {"gpu_ms": round(g_wo, 4), "p50_ms_without_sigmoid": round(p_wo, 4), "p50_ms_with_separate_sigmoid_pass": round(p_w, 4),
            "must_move_bytes": must, "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / HBM_PEAK_GBPS, 4),
            "GBps_single_call": round(must / (p_wo * 1e-3) / 1e9, 1),
            "kernels_profiled": decode_profiled(config, N, x.shape[2], x.shape[3]),
            "note": "gpu_ms = GPU time of one decode (both kernels): 50 calls enqueued back to back between HIP events, as inside a step where the "
                    "host runs ahead — GBps / frac_of_8TBps use it; p50 = one call at a time from Python with the GPU idle before it (HIP events "
                    "around gather_detection2d on the forward's own outputs, median of 30): it adds the host's argument set-up and launch "
                    "latency before the first kernel starts (~8 us).  without = the path (sigmoid is the heatmap out_conv's epilogue); with = torch.sigmoid(logits) + decode, "
                    "what a caller holding logits pays"}


_REF64 = {}


def feature_errors(config, x2, algo):
    """max |feature - float64 oracle| / max |float64 oracle| at the neck output and each head's last-block output + the heatmap."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu
    if algo == "f43":        # the opt-in F(4,3) class (KernelOptions.f43): AUTO's plan with the 256-channel head blocks on csrc/winograd13.hip
        model = build_model(config, algo="auto", f43=True, reuse_buffers=False)
    else:
        model = build_model(config, algo=algo, reuse_buffers=False) if algo != "cpu" else build_model(config)
    sd = {k_: v.detach().cpu() for k_, v in model.state_dict().items()}
    key = (config, tuple(x2.shape), float(x2.double().sum()))          # the float64 oracle once per (weights recipe, input): the same for every class
    if key not in _REF64:
        _REF64.clear()
        _REF64[key] = ref_cpu.forward_float64(sd, x2, sigmoid=True, return_intermediates="heads")
    out64, _, neck64, heads64 = _REF64[key]
    if algo == "cpu":
        out, _, neck, heads = ref_cpu.forward(sd, x2, sigmoid=True, return_intermediates="heads")
        heat = out["heatmap"]
    else:
        with torch.no_grad():
            heat = model(x2.cuda())[0].cpu()
        torch.cuda.synchronize()
        plan = model._engine.plan_for(x2.cuda(), sigmoid=True)
        nb, _, _, nc, nup = plan.neck_out
        neck = plan.tensor(nb)[..., :nc].permute(0, 3, 1, 2).cpu()
        if nup:
            neck = torch.nn.functional.interpolate(neck, scale_factor=2, mode="nearest")
        heads = {name: plan.tensor(buf)[..., off:off + c].permute(0, 3, 1, 2).cpu() for name, (buf, ld, off, c, _, _, _) in plan.head_features.items()}
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    e = {"neck": rel(neck, neck64)}
    for name in heads64:
        e["head." + name] = rel(heads[name], heads64[name])
    e["heatmap_abs"] = float((heat.double() - out64["heatmap"]).abs().max())
    return {k_: float(f"{v:.3e}") for k_, v in e.items()}


def _cpu_leg_child(spec):
    """Runs in a CHILD process (python bench.py --cpu-leg-child '<json>'): OpenMP reads OMP_PROC_BIND / OMP_PLACES once, when torch is
    imported, so each thread placement needs its own process.  Prints one JSON line."""
    if spec.get("affinity"):
        os.sched_setaffinity(0, set(spec["affinity"]))      # one oracle process per NUMA node / core group (cpu_baseline's multi-process legs)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import decode_ref
    import ref_cpu
    torch.manual_seed(0)
    model = synthetic_weights_(cl.build_centernet(os.path.join(ROOT, "centernet-lightning_amd", "configs", CONFIGS[spec["config"]])))
    sd = {k_: v.detach().cpu() for k_, v in model.state_dict().items()}
    tracking, k, n, h, w = spec["config"] == "tracking", spec["k"], spec["n"], spec["h"], spec["w"]

    def one(x):
        o = ref_cpu.forward(sd, x, sigmoid=True)
        t0 = time.perf_counter()
        decode_ref.decode_detections_torch(o["heatmap"], o["box_2d"], k, 3, reid=o["reid"] if tracking else None)
        return time.perf_counter() - t0

    x0 = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(1234))
    torch.set_num_threads(spec["threads"][0])
    one(x0[:1])                                                             # global warm-up (oneDNN primitive caches, allocator)
    probes = []
    for threads in spec["threads"]:
        for cl_ in (True, False) if spec.get("both_layouts") else (True,):
            torch.set_num_threads(threads)
            x = x0.contiguous(memory_format=torch.channels_last) if cl_ else x0
            t0 = time.perf_counter()
            one(x)
            probes.append((time.perf_counter() - t0, threads, cl_))
    _, threads, cl_ = min(probes)
    torch.set_num_threads(threads)
    x = x0.contiguous(memory_format=torch.channels_last) if cl_ else x0
    for _ in range(spec["warmup"]):
        one(x)
    if spec.get("t_start"):                                  # concurrent processes start their timed passes together
        time.sleep(max(0.0, spec["t_start"] - time.time()))
    t_begin = time.time()
    ts, dec = [], []
    for _ in range(spec["passes"]):
        t0 = time.perf_counter()
        dec.append(one(x))
        ts.append(time.perf_counter() - t0)
    ts.sort(); dec.sort()
    print(json.dumps({"threads": threads, "channels_last": cl_, "images_per_s": round(n / ts[len(ts) // 2], 3), "timed_passes": len(ts), "n": n,
                      "t_begin": t_begin, "t_end": time.time(),
                      "decode_p50_ms": round(dec[len(dec) // 2] * 1e3, 3), "omp": {k_: os.environ.get(k_) for k_ in ("OMP_PROC_BIND", "OMP_PLACES")},
                      "probes_images_per_s": {f"{t_}thr{'_cl' if c_ else ''}": round(n / s_, 3) for s_, t_, c_ in probes}}))


MULTI_NOTE = ("not run (--cpu-multi): measured once per round — profiles/r04_bench_c1_with_cpu_multi.json: 16 oracle processes x 16 logical CPUs on disjoint core "
              "sets sum to 9.3 images/s (2.5 by wall clock, stragglers included), 2 x 128: 0.95 — no more than ONE 16-thread process (9.9): this host does not feed "
              "the oracle beyond ~10 images/s however it is placed")


def host_limits():
    """What the container may use of the host: logical CPUs in the affinity mask, the cgroup CPU quota, NUMA nodes."""
    out = {"affinity_cpus": len(os.sched_getaffinity(0))}
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            out["cgroup:" + f.rsplit("/", 1)[1]] = open(f).read().strip()
        except OSError:
            pass
    try:
        out["numa_nodes"] = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        pass
    return out


def cpu_quota():
    """CPUs' worth of time the cgroup grants per period (cpu.max "quota period" / cfs_quota_us), or None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(q) // int(per))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, q // per)
    except (OSError, ValueError):
        return None


def cpu_groups(cpus):
    """Process counts for the multi-process CPU legs: one per NUMA node, and one per 16 cores (a pair of CCDs)."""
    counts = set()
    try:
        nodes = [d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]
        if len(nodes) > 1 and len(cpus) // len(nodes) >= 8:
            counts.add(len(nodes))
    except OSError:
        pass
    if len(cpus) >= 32:
        counts.add(len(cpus) // 16)
    return sorted(c for c in counts if c >= 2)


def cpu_baseline(config, k, H, W, gpu_decode_p50_ms, multi_process=False):
    """Oracle leg: the CPU restatement of forward + decode (kind "port") on the box's host cores — C0 exactly (1 x 3 x 512 x 512) and the bench
    shape at N = 32 (the GPU leg's own batch: ~30 s of CPU work), each under two OpenMP thread placements (the runtime's default, and threads
    pinned to consecutive cores: OMP_PROC_BIND=close OMP_PLACES=cores) and the probed thread counts; the best is `value`, every probe is listed."""
    import subprocess
    cores = os.cpu_count()
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    quota = cpu_quota()                        # more threads than the cgroup's CPU quota only share it (the GPU boxes: 16 of a 256-CPU host)
    cap = min(cores, quota) if quota else cores
    tlist = sorted({max(1, min(t_, cap)) for t_ in (16, 32, 64)})
    quota_note = f", capped by the cgroup quota of {quota} CPUs" if quota else ""

    def child(spec, pinned):
        env = dict(os.environ)
        env.pop("OMP_PROC_BIND", None); env.pop("OMP_PLACES", None)
        if pinned:
            env.update({"OMP_PROC_BIND": "close", "OMP_PLACES": "cores"})
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg-child", json.dumps(spec)], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": r.stderr[-400:]}
        return json.loads(r.stdout.strip().splitlines()[-1])

    legs = {}
    for pinned in (False, True):
        tag = "pinned_close_cores" if pinned else "omp_default"
        legs["N32_" + tag] = child({"config": config, "k": k, "n": 32, "h": H, "w": W, "threads": tlist, "warmup": 1, "passes": 3}, pinned)
    # one oracle process per NUMA node / per group of 16 cores, running concurrently on disjoint cores, throughput summed (a single
    # process does not feed a 2 x 64-core host — 64 threads were slower than 16)
    multi = {}
    try:
        cpus = sorted(os.sched_getaffinity(0))
        for P_ in (cpu_groups(cpus) if multi_process else []):
            groups = [cpus[i * len(cpus) // P_:(i + 1) * len(cpus) // P_] for i in range(P_)]
            n_each = max(1, 32 // P_)
            t_start = time.time() + 30.0                      # children import torch, build the model, warm up; then start together
            procs = []
            for g_ in groups:
                spec = {"config": config, "k": k, "n": n_each, "h": H, "w": W, "threads": [len(g_)], "warmup": 1, "passes": 2, "affinity": g_, "t_start": t_start}
                env = dict(os.environ)
                env.pop("OMP_PROC_BIND", None); env.pop("OMP_PLACES", None)
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-leg-child", json.dumps(spec)], env=env, stdout=subprocess.PIPE,
                                              stderr=subprocess.PIPE, text=True))
            outs = []
            for pr in procs:
                so, se = pr.communicate(timeout=900)
                outs.append(json.loads(so.strip().splitlines()[-1]) if pr.returncode == 0 else {"error": se[-300:]})
            if all("t_end" in o for o in outs):
                wall = max(o["t_end"] for o in outs) - min(o["t_begin"] for o in outs)
                late = max(o["t_begin"] for o in outs) - t_start
                multi[f"{P_}_processes_x_{len(groups[0])}_cores"] = {"images_per_s": round(sum(o["n"] * o["timed_passes"] for o in outs) / wall, 3), "wall_s": round(wall, 2),
                                                                    "images_per_process_per_pass": n_each, "passes": 2, "started_late_s": round(max(0.0, late), 2),
                                                                    "per_process_images_per_s": [o["images_per_s"] for o in outs]}
            else:
                multi[f"{P_}_processes"] = {"error": [o.get("error") for o in outs if "error" in o][:1]}
    except Exception as e:
        multi["error"] = repr(e)
    c0 = child({"config": config, "k": k, "n": 1, "h": 512, "w": 512, "threads": sorted({max(1, min(t_, cap)) for t_ in (8, 16, 32)}), "warmup": 2, "passes": 5,
                "both_layouts": True}, True)
    good = {k_: v for k_, v in legs.items() if "images_per_s" in v}
    if not good:
        return {"error": legs}
    best_tag = max(good, key=lambda k_: good[k_]["images_per_s"])
    cn = good[best_tag]
    mgood = {k_: v for k_, v in multi.items() if isinstance(v, dict) and "images_per_s" in v}
    if mgood and max(v["images_per_s"] for v in mgood.values()) > cn["images_per_s"]:
        mtag = max(mgood, key=lambda k_: mgood[k_]["images_per_s"])
        mv = mgood[mtag]
        return {"value": mv["images_per_s"], "unit": "images/s", "cores": len(os.sched_getaffinity(0)), "kind": "port", "os_cpu_count": cores, "cpu_model": cpu_model,
                "sample": f"oracle/ref_cpu.forward + decode_ref.decode_detections_torch (same weights), {mtag.replace('_', ' ')} run CONCURRENTLY on disjoint core sets (os.sched_setaffinity), "
                          f"each {mv['images_per_process_per_pass']} x 3 x {H} x {W} images per pass, 1 warm-up + 2 timed passes started together; value = all images of the timed passes / "
                          f"(last end - first start); best of {{one process: legs, several: multi_process_legs}}; torch {torch.__version__} CPU fp32",
                "multi_process_legs": multi, "host": host_limits(), "legs": legs, "C0_1x3x512x512": c0, "single_process_best": {"images_per_s": cn["images_per_s"], "threads": cn["threads"], "placement": best_tag},
                "decode_p50_ms": {"cpu_N32": cn["decode_p50_ms"], "cpu_N1_C0": c0.get("decode_p50_ms"), "gpu_full_batch": gpu_decode_p50_ms}}
    return {"value": cn["images_per_s"], "unit": "images/s", "cores": cn["threads"], "kind": "port", "multi_process_legs": multi or MULTI_NOTE,
            "host": host_limits(),
            "os_cpu_count": cores, "cpu_model": cpu_model,
            "sample": f"oracle/ref_cpu.forward + decode_ref.decode_detections_torch (same weights) on 32x3x{H}x{W} (the GPU leg's batch), 1 warm-up + {cn['timed_passes']} timed "
                      f"passes, median; torch {torch.__version__} CPU fp32, {cn['threads']} threads{', channels_last' if cn['channels_last'] else ''}, placement {best_tag} "
                      f"(best of two OpenMP placements x thread counts {tlist}{quota_note}: legs[*].probes_images_per_s); decode = the reference's torch op sequence on CPU",
            "legs": legs, "C0_1x3x512x512": c0,
            "decode_p50_ms": {"cpu_N32": cn["decode_p50_ms"], "cpu_N1_C0": c0.get("decode_p50_ms"), "gpu_full_batch": gpu_decode_p50_ms}}


def also_workloads(world, config, B, H, W):
    """The other BASELINE configurations a default run (the C1 line) appends under `also`: at one GPU C2 and C4's per-GPU share, at N > 1 the
    two configurations BASELINE.json DEFINES on 8 GPUs — C3 (FPN, 512 images = 64 per GPU) and C4 (tracking, 256 images = 32 per GPU at
    608x1088) — sharded over the ranks like the main line, each with its all-gather (reference collective: eval/coco.py:10-18)."""
    if config != "simple" or (B, H, W) != (32, 512, 512):
        return []
    fpn = {"config": "fpn", "batch": 64, "height": 512, "width": 512, "k": 100}
    trk = {"config": "tracking", "batch": 32, "height": 608, "width": 1088, "k": 100}
    if world == 1:
        return [dict(fpn, name="C2"), dict(trk, name="C4 (one GPU's share: 32 of its 256 images)")]
    return [dict(fpn, name=f"C3 ({64 * world} images over {world} GPUs)"), dict(trk, name=f"C4 ({32 * world} images over {world} GPUs)")]


def collate_alone_ms(model, x, tracking, k, collator, barrier):
    """The collate step alone, synchronously (pack + all-gather + unpack): median of 10."""
    with torch.no_grad():
        o = model(x)
        dets = model.gather_tracking2d(o, num_detections=k) if tracking else model.gather_detection2d(o, num_detections=k)
        cts = []
        for _ in range(10):
            barrier()
            t0 = time.perf_counter()
            collator.result(collator.submit(dets))
            torch.cuda.synchronize()
            cts.append(time.perf_counter() - t0)
    cts.sort()
    return cts[len(cts) // 2] * 1e3


def short_line(spec, rank=0, world=1, barrier=None, steps=10, warmup=3):
    """A short run of another BASELINE configuration (`also_workloads`): EVERY rank runs it on its own batch (weak scaling, the all-gather
    of the detections behind each step when world > 1); rank 0 returns the line, the others None."""
    config, B, H, W, k = spec["config"], spec["batch"], spec["height"], spec["width"], spec["k"]
    tracking = config == "tracking"
    barrier = barrier or torch.cuda.synchronize
    model = build_model(config)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1234 + rank)).cuda()
    collator = cl.Collator()
    el = timed(model, x, tracking, k, warmup, steps, collator, barrier)
    collate_ms = None
    if world > 1:
        el = max_over_ranks(el, "cuda")
        collate_ms = collate_alone_ms(model, x, tracking, k, collator, barrier)
    if rank != 0:
        return None
    with torch.no_grad():
        rows, plan = conv_kernel_profile(model, x, reps=2)
    roof, stack = roofline_block(rows, config, B, H, W)
    eng = model._engine
    line = {"config": {"workload": f"BASELINE {spec['name']}: ResNet34 + {config} neck, {B} img/GPU x {H}x{W}, k={k}", "global_batch": world * B, "per_gpu_batch": B,
                       "parallelism": parallelism_text(world)},
            "value": round(job_throughput(B, world, steps, el), 2), "unit": "images/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(el / steps * 1e3, 3),
            "roofline": {kk: roof[kk] for kk in ("kernel", "achieved", "peak", "frac", "effective_tflops", "launches_per_step", "kernel_ms_per_step", "avg_launch_us",
                                                 "algorithmic_bytes_per_launch", "traffic", "traffic_source")},
            "conv_stack": stack,
            "activation_arena_MB": {"with_liveness_reuse": round(plan.arena_bytes / 1e6, 1), "every_buffer_separate": round(plan.bytes_without_reuse / 1e6, 1),
                                    "sub_batch": eng.sub_batch(B, H, W)}}
    if collate_ms is not None:
        line["collate_ms"] = round(collate_ms, 4)
    return line


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="simple")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--algo", choices=["auto", "f32"], default="auto", help="KernelOptions.algo of the measured job")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-multi", action="store_true", help="CPU baseline: also run several oracle processes at once on disjoint core sets (one per NUMA node, one per 16 "
                    "cores), throughput summed — minutes of wall time; measured once per round (profiles/r04_bench_c1_with_cpu_multi.json: no gain on the GPU box's host)")
    ap.add_argument("--no-variants", action="store_true", help="skip the f2 / f32 legs")
    ap.add_argument("--no-also", action="store_true", help="skip the short C2 / C4 lines")
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--collate-probe", action="store_true", help="N=1 only: also time the steps with the RCCL all-gather of the detections forced "
                    "through a world-size-1 process group (side stream, one step behind) — the N>1 data path on the one GPU a box has")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer conv table to stderr")
    return ap.parse_args(argv)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-leg-child":
        _cpu_leg_child(json.loads(sys.argv[2]))
        return
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))          # no launcher around us: become one

    rank, world, local_rank = setup_distributed(args.gpus)
    if os.environ.get("CNL_BENCH_STUB") == "1":
        if torch.cuda.is_available():
            raise SystemExit("CNL_BENCH_STUB=1 is the no-GPU launcher self-test; on a GPU box the bench measures the real model")
        return stub_main(args, rank, world)
    t_wall = {"start": time.perf_counter()}

    def lap(name):                      # wall seconds per section of this run (rank 0's line: `bench_wall_s`)
        now = time.perf_counter()
        t_wall[name] = round(now - t_wall.get("_last", t_wall["start"]), 1)
        t_wall["_last"] = now

    tracking = args.config == "tracking"
    model = build_model(args.config, algo=args.algo)
    B, H, W = args.batch, args.height, args.width
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1234 + rank)).cuda()
    collator = cl.Collator()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    lap("build")
    elapsed = timed(model, x, tracking, args.k, args.warmup, args.steps, collator, barrier)
    lap("timed_steps")
    if world > 1:
        elapsed = max_over_ranks(elapsed, "cuda")
        collate_ms = collate_alone_ms(model, x, tracking, args.k, collator, barrier)
    coll = collective_info("cuda")          # (a collective at world > 1: every rank)

    if rank == 0:
        with torch.no_grad():
            rows, plan = conv_kernel_profile(model, x)
        roof, stack = roofline_block(rows, args.config, B, H, W)
        dec = decode_block(model, x, tracking, args.k, args.config)
        eng = model._engine
        result = {
            "metric": "images/sec @512x512 ResNet34 CenterNet forward + gather_detection2d",
            "value": round(job_throughput(B, world, args.steps, elapsed), 2),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "timed_region_s": round(elapsed, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "collective": coll,
            "dtype": "f32",
            "dtype_note": "fp32 in / fp32 accumulate / fp32 out; where it pays, each fp32 product is formed on the fp16 matrix cores from a two-way fp16 split of "
                          "both (power-of-two scaled) operands (3 cross terms); KernelOptions.algo = " + args.algo + " (auto: every kernel's error at or below the fp32 MFMA's; "
                          "f32: fp32 matrix cores only): see `variants` and `accuracy`",
            "data": "synthetic (seeded rand images; random-init weights of the named architecture)",
            "config": {"workload": f"BASELINE C{'1' if args.config == 'simple' else ('4' if tracking else '2/3')}: ResNet34 + {args.config} neck, "
                                   f"{B} img/GPU x {H}x{W}, heads {'2+4+reid64' if tracking else '80+4'} (w256), k={args.k}, nms 3",
                       "global_batch": world * B, "per_gpu_batch": B, "parallelism": parallelism_text(world)},
            "roofline": roof,
            "conv_stack": stack,
            "decode": dec,
            "decode_p50_ms": dec["p50_ms_without_sigmoid"], "decode_gpu_ms": dec["gpu_ms"],
            "activation_arena_MB": {"with_liveness_reuse": round(plan.arena_bytes / 1e6, 1), "every_buffer_separate": round(plan.bytes_without_reuse / 1e6, 1),
                                    "sub_batch": eng.sub_batch(B, H, W)},
        }
        if world > 1:
            result["collate_ms"] = round(collate_ms, 4)
            result["collate_note"] = "pack + all_gather_into_tensor + unpack run synchronously, median of 10 (inside `value` the gather runs on a side stream behind the next forward)"
        if args.layers:
            for what, fl, ms, kd, _nb, _fx in rows:
                print(f"{what:44s} {kd:16s} {fl / 1e9:10.2f} GFLOP {ms * 1e3:10.1f} us {fl / (ms * 1e-3) / 1e12 if ms else 0:8.1f} TF", file=sys.stderr)
        lap("roofline+decode")
        if world == 1:
            sync = torch.cuda.synchronize
            if not args.no_variants:
                result["variants"] = {}
                for algo in ("f32", "f43"):
                    if algo == args.algo:
                        continue
                    try:
                        m2 = build_model(args.config, algo="auto", f43=True) if algo == "f43" else build_model(args.config, algo=algo)
                        st = min(max(args.steps // 2, 3), 20)
                        el = timed(m2, x, tracking, args.k, min(args.warmup, 3), st, cl.Collator(), sync)
                        with torch.no_grad():
                            r2, _ = conv_kernel_profile(m2, x, reps=2)
                        rf2, _ = roofline_block(r2, args.config, B, H, W)
                        result["variants"][algo] = {"value": round(B * st / el, 2), "unit": "images/s", "ms_per_step": round(el / st * 1e3, 3), "steps": st,
                                                    "dominant_kernel": rf2["kernel"].split(" (")[0], "roofline_frac": rf2["frac"], "effective_tflops": rf2["effective_tflops"]}
                        del m2
                    except Exception as e:      # reported, never fatal: `value` above is the measurement
                        result["variants"][algo] = {"error": repr(e)}
            lap("variants")
            if args.collate_probe:
                try:
                    import socket
                    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
                    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
                    forced = cl.Collator(force=True)
                    el_f = timed(model, x, tracking, args.k, args.warmup, args.steps, forced, sync)
                    el_0 = timed(model, x, tracking, args.k, args.warmup, args.steps, cl.Collator(), sync)
                    with torch.no_grad():
                        o = model(x)
                        dets = model.gather_tracking2d(o, num_detections=args.k) if tracking else model.gather_detection2d(o, num_detections=args.k)
                        cts = []
                        for _ in range(10):
                            sync(); t0 = time.perf_counter()
                            forced.result(forced.submit(dets)); sync()
                            cts.append(time.perf_counter() - t0)
                    cts.sort()
                    result["collate_probe"] = {"ms_per_step_with_forced_rccl_gather": round(el_f / args.steps * 1e3, 3), "ms_per_step_without": round(el_0 / args.steps * 1e3, 3),
                                               "collate_ms_synchronous": round(cts[len(cts) // 2] * 1e3, 4),
                                               "note": "world-size-1 nccl (= RCCL) group, Collator(force=True): pack kernel -> all_gather_into_tensor on the side stream "
                                                       "behind an event -> unpack one step later; the same code path N > 1 ranks take, minus the wire"}
                    dist.destroy_process_group()
                except Exception as e:
                    result["collate_probe"] = {"error": repr(e)}
            if not args.no_also:
                # one image (BASELINE C0 shape when the bench runs C1): forward + decode, back to back, default plan and latency mode
                try:
                    lat = {}
                    for name, opts in (("default", {}), ("split_small", {"split_small": True}), ("latency", {"latency": True}), ("latency+split_small", {"latency": True, "split_small": True})):
                        m3 = build_model(args.config, **opts)
                        x1 = x[:1]
                        gather3 = m3.gather_tracking2d if tracking else m3.gather_detection2d
                        with torch.no_grad():
                            lat[name] = round(gpu_ms_back_to_back(lambda: gather3(m3(x1), num_detections=args.k), calls=30, rounds=3), 4)
                        del m3
                    result["latency_ms_N1"] = dict(lat, note=f"1 x 3 x {H} x {W} (BASELINE C0's shape), forward + decode, 30 calls back to back; default: launches of at most 128 winograd9 work items "
                                                             "take the bit-identical 4-row x 32-cout items (csrc/winograd10.hip); latency = KernelOptions(latency=True): every eligible "
                                                             "3x3 / stride-1 layer does, also those whose default is another kernel; split_small = KernelOptions(split_small=True) "
                                                             "(reduction of the small-grid convs split over workgroups); both options off by default")
                except Exception as e:
                    result["latency_ms_N1"] = {"error": repr(e)}
    lap("latency_N1")
    # the other BASELINE configurations: EVERY rank takes part (at N > 1 these are C3 and C4, the two configurations defined on 8 GPUs)
    if not args.no_also:
        lines = []
        if args.steps < 200:
            # the main line's job once more over 200 steps (the driver's --steps 20 makes `value` a 150-ms sample; this is the same job as a 1.5-s one)
            el200 = timed(model, x, tracking, args.k, 0, 200, collator, barrier)
            if world > 1:
                el200 = max_over_ranks(el200, "cuda")
            if rank == 0:
                lines.append({"config": {"workload": "the main line's job again, 200 steps (no further warm-up)", "global_batch": world * B, "per_gpu_batch": B, "parallelism": parallelism_text(world)},
                              "value": round(job_throughput(B, world, 200, el200), 2), "unit": "images/s", "n_gpus": world, "steps": 200, "warmup": 0,
                              "ms_per_step": round(el200 / 200 * 1e3, 3), "timed_region_s": round(el200, 4)})
        for spec in also_workloads(world, args.config, B, H, W):
            try:
                ln = short_line(spec, rank, world, barrier)
                if ln is not None or world == 1:
                    lines.append(ln)
            except Exception as e:
                if world > 1:
                    raise                                   # a rank that drops out of a collective would hang the others: fail the job loudly
                lines.append({"config": spec, "error": repr(e)})
        if rank == 0 and lines:
            result["also"] = lines
    lap("also")
    if rank == 0:
        if world == 1:
            torch.cuda.empty_cache()
            if not args.no_accuracy:
                x2 = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(4242))
                try:
                    acc = {a: feature_errors(args.config, x2, a) for a in ("auto", "f32", "f43", "cpu")}
                    result["accuracy"] = {"max_err_over_max_ref_vs_float64_oracle": acc, "tolerance": 1e-4,
                                          "sample": f"2 images of the bench shape, same weights; 'cpu' = the CPU fp32 oracle's own distance from float64",
                                          "parity_note": "backbone + ConvBnAct parity is unpinned by the reference (torchvision / vision_toolbox absent): the oracle is this repo's restatement; "
                                                         "decode / head wiring / neck options are pinned by goldens generated from the reference (tests/golden/)"}
                except Exception as e:
                    result["accuracy"] = {"error": repr(e)}
            lap("accuracy")
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(args.config, args.k, H, W, dec["p50_ms_without_sigmoid"], multi_process=args.cpu_multi)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        lap("cpu_baseline")
        result["bench_wall_s"] = {k_: v for k_, v in t_wall.items() if k_ not in ("start", "_last")}
        # the ONE JSON line goes out last: RCCL writes a version banner through C stdio, which a pipe would otherwise deliver after Python's line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
