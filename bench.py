#!/usr/bin/env python
"""bench.py — images/sec of the CenterNet hot path (forward + gather_detection2d [+ RCCL all-gather]) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under
`python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

  step      = one pass of the hot path over one synthetic batch per GPU:
              CenterNet.forward (ResNet-34 -> neck -> heads, sigmoid) + gather_detection2d (k=100, nms 3)
              [+ all-gather of the packed detections when N > 1].  Inputs are resident in HBM.
  workload  = BASELINE.json configs[1] (C1): ResNet34 + simple upsample neck, batch 32 per GPU, 512x512, 80 classes
              (`--config fpn --batch 64` = C2/C3, `--config tracking --batch 32 --height 608 --width 1088` = C4).
  value     = total images / max-over-ranks wall time of the K timed steps (weak scaling: batch per GPU fixed).
  roofline  = the dominant kernel by time, measured with HIP events on the launch stream around every launch of one step.  The
              3x3/s1 layers run Winograd F(2x2,3x3) on one of two multiplier arrays, chosen per layer SHAPE (include/centernet_gfx950.h):
              `cnl_wino5::winograd5_kernel` (Cin >= 128 or Cout >= 512: fp32 operands scaled by a per-tensor power of two and split
              into two fp16 pieces, three fp16 MFMAs per product, fp32 accumulation — fp32-grade accuracy on the 16x faster fp16
              matrix core; peak = 2.5 PFLOP/s dense fp16) or `cnl_wino2::winograd2_kernel` (fp32 MFMA, peak 157.3).  `achieved` counts the matrix-core flops the kernel EXECUTES
              (direct-conv flops x 16/36, x 3 for the split), so `frac` is an honest hardware fraction; `effective_tflops` is the same
              time against the direct-conv (algorithmic) flops.  The other conv kernels are reported under `other_kernels`.
  dtype     = "f32": inputs, weights, accumulation and outputs are fp32 and every product is formed to fp32 accuracy (the dropped
              split terms are <= 2^-24 relative); `cpu_baseline.sample` carries the max |GPU - CPU oracle| of this very run.
  cpu_baseline = the CPU oracle (oracle/ref_cpu.py + oracle/decode_ref.py: the plain PyTorch restatement of the
              reference path — kind "port") timed on this box's host cores on a bounded sample; rank 0, N=1 only.
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
BF16_MFMA_PEAK_TFLOPS = 2500.0    # same guide: dense bf16 (v_mfma_f32_32x32x16_bf16), 16x the fp32 MFMA rate
F16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: dense fp16 = bf16 rate
# HBM bytes per launch from rocprofv3 PMC passes of this same command (profiles/r01_pmc_traffic_final9.txt: 29 winograd5 launches): per kernel, mean
# FETCH_SIZE x 2 (gfx950 reports half of a 16 B/lane stream, MI355X_MICROARCH.md §HBM) + mean WRITE_SIZE over the launches
# of a C1 step (fp16-split Winograd: 186.2 MB x 2 + 146.6 MB = 518.9 MB).  Other configs: not profiled -> null.
MEASURED_TRAFFIC_BYTES_PER_LAUNCH = {("simple", 32, 512, 512, "cnl_conv::conv_mfma_kernel"): 237.0e6,
                                     ("simple", 32, 512, 512, "cnl_wino2::winograd2_kernel"): 328.9e6,
                                     ("simple", 32, 512, 512, "cnl_wino3::winograd3_kernel"): 731.5e6,
                                     ("simple", 32, 512, 512, "cnl_wino5::winograd5_kernel"): 484.3e6}
CONFIGS = {"simple": "resnet34_simple.yaml", "fpn": "resnet34_fpn.yaml", "tracking": "tracking_resnet34_fpn.yaml"}


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


def step(model, x, tracking, k):
    out = model(x)
    dets = model.gather_tracking2d(out, num_detections=k) if tracking else model.gather_detection2d(out, num_detections=k)
    return model.collate(dets)


def conv_kernel_profile(model, x, reps=3):
    """HIP-event timing of every conv launch of one forward (events on torch's current stream == the launch
    stream).  Returns (sum of conv durations per step [ms], conv flops per step, launches per step, per-layer rows)."""
    import ctypes
    eng = model._engine
    model(x)                                            # make sure the plan exists
    # batches whose activations would exceed the kernels' 4 GiB addressing run as equal sub-batches (Engine.forward): time one
    # sub-batch and scale
    n_sub = eng.sub_batch(x.shape[0], x.shape[2], x.shape[3])
    scale = x.shape[0] / n_sub
    x = x[:n_sub]
    plan = eng.plans[(n_sub, x.shape[2], x.shape[3], True)]
    lib = plan.lib
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    convs = [L for L in plan.launches if L.fn in (lib.cnl_conv2d_nhwc_f32, lib.cnl_conv3x3_winograd_f32, lib.cnl_conv3x3_up2_nhwc_f32)]
    acc = [0.0] * len(convs)
    for _ in range(reps):
        model(x)                                        # refresh inputs of every layer
        torch.cuda.synchronize()
        evs = []
        for L in convs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.fn(ctypes.byref(L.args), stream)
            e1.record()
            assert rc == 0, L.what
            evs.append((e0, e1))
        torch.cuda.synchronize()
        for i, (e0, e1) in enumerate(evs):
            acc[i] += e0.elapsed_time(e1)
    def kind(L):
        if L.fn is lib.cnl_conv3x3_up2_nhwc_f32:        # four sub-pixel phase convs on the direct kernels
            return "direct_f16x2" if lib.cnl_conv3x3_up2_kernel(ctypes.byref(L.args)) == 5 else "direct"
        if L.fn is not lib.cnl_conv3x3_winograd_f32:
            return "direct_f16x2" if lib.cnl_conv2d_kernel(ctypes.byref(L.args)) == 5 else "direct"
        return {3: "winograd_bf16x3", 5: "winograd_f16x2"}.get(lib.cnl_conv3x3_winograd_kernel(ctypes.byref(L.args)), "winograd_f32")
    rows = [(L.what, L.flops * scale, acc[i] / reps * scale, kind(L)) for i, L in enumerate(convs)]
    # algorithmic HBM bytes of a conv launch: input + weights + bias + output (+ residual), each touched once
    nbytes = 0
    bytes_by_kind = {}
    for L in convs:
        p = L.args
        up_in = 2 if p.flags & 4 else 1
        ho = (p.H_in * up_in + 2 * p.pad - p.KH) // p.stride + 1
        wo = (p.W_in * up_in + 2 * p.pad - p.KW) // p.stride + 1
        up_out = 4 if p.flags & 8 else 1
        out_px = p.N * ho * wo * up_out
        b_ = 4 * (p.N * p.H_in * p.W_in * p.Cin + p.Cout * p.KH * p.KW * p.Cin + p.Cout + out_px * p.Cout
                  + (out_px * p.Cout if p.residual else 0))
        nbytes += b_
        bytes_by_kind[kind(L)] = bytes_by_kind.get(kind(L), 0) + b_ * scale
    return sum(r[2] for r in rows), sum(r[1] for r in rows), len(rows), rows, bytes_by_kind


def cpu_baseline(model, tracking, k, H, W, budget_s=20.0):
    """Oracle leg: CPU restatement of forward + decode on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import decode_ref
    import ref_cpu
    sd = {k_: v.detach().cpu() for k_, v in model.state_dict().items()}
    n = 2
    x = torch.rand(n, 3, H, W, generator=torch.Generator().manual_seed(1234))
    cores = torch.get_num_threads()

    def one():
        o = ref_cpu.forward(sd, x, sigmoid=True)
        d = decode_ref.decode_detections(o["heatmap"].numpy(), o["box_2d"].numpy(), k, 3,
                                         reid=o["reid"].numpy() if tracking else None)
        return o, d

    t0 = time.perf_counter()
    o, d = one()                                        # warm-up (also used as the checker below)
    warm = time.perf_counter() - t0
    iters, spent = 0, 0.0
    while iters < 1 or (spent + warm < budget_s and iters < 5):
        t0 = time.perf_counter()
        one()
        spent += time.perf_counter() - t0
        iters += 1
    # checker: the HIP path on the same sample
    with torch.no_grad():
        out = model(x.cuda())
    err = float((out[0].cpu() - o["heatmap"]).abs().max())
    return {"value": round(n * iters / spent, 3), "unit": "images/s", "cores": cores, "kind": "port", "max_abs_err_heatmap": err,
            "sample": f"{iters} timed passes of oracle/ref_cpu.forward + decode_ref on {n}x3x{H}x{W} (same weights), "
                      f"torch {torch.__version__} CPU fp32, {cores} threads; max |heatmap_gpu - heatmap_cpu| = {err:.2e}"}


def oracle_check(model, H, W):
    """max |heatmap_gpu - heatmap_cpu| on a 2-image sample: one pass of the CPU oracle, no timing (the checker, not the product)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu
    sd = {k_: v.detach().cpu() for k_, v in model.state_dict().items()}
    x = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(1234))
    o = ref_cpu.forward(sd, x, sigmoid=True)
    with torch.no_grad():
        out = model(x.cuda())
    return float((out[0].cpu() - o["heatmap"]).abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="simple")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer conv table to stderr")
    ap.add_argument("--oracle-check", action="store_true", help="add max |heatmap - CPU oracle| of this configuration (one oracle pass)")
    ap.add_argument("--no-fp32-mfma-leg", action="store_true",
                    help="skip the extra short run with every conv on the fp32 matrix core (CNL_WINO=2 CNL_CONV_F16X2=0 CNL_STEM_F16X2=0), reported beside `value`")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    # CNL_BENCH_BACKEND=gloo is a functional check of the N>1 flow on a box with fewer GPUs than ranks (ranks then share
    # devices and the collective goes through the host); the measured configuration is always nccl = RCCL, one GPU per rank
    backend = os.environ.get("CNL_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    else:
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))     # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    tracking = args.config == "tracking"
    torch.manual_seed(0)
    model = synthetic_weights_(cl.build_centernet(os.path.join(ROOT, "centernet-lightning_amd", "configs", CONFIGS[args.config]))).cuda()
    B, H, W = args.batch, args.height, args.width
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1234 + rank)).cuda()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step(model, x, tracking, args.k)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(model, x, tracking, args.k)
        barrier()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        with torch.no_grad():
            conv_ms, conv_flops, n_launch, rows, conv_bytes = conv_kernel_profile(model, x)
            # decode-only latency (p50) on the forward's own outputs
            out = model(x)
            lat = []
            for _ in range(30):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                (model.gather_tracking2d if tracking else model.gather_detection2d)(out, num_detections=args.k)
                e1.record()
                torch.cuda.synchronize()
                lat.append(e0.elapsed_time(e1))
            lat.sort()
        def agg(kind):
            sel = [r for r in rows if r[3] == kind]
            ms = sum(r[2] for r in sel)
            fl = sum(r[1] for r in sel)
            return len(sel), ms, fl
        n_w, ms_w, fl_w = agg("winograd_f32")
        n_b, ms_b, fl_b = agg("winograd_bf16x3")
        n_h, ms_h, fl_h = agg("winograd_f16x2")
        n_d, ms_d, fl_d = agg("direct")
        n_d5, ms_d5, fl_d5 = agg("direct_f16x2")
        direct_tf = fl_d / (ms_d * 1e-3) / 1e12 if ms_d else 0.0
        f32_exec_tf = fl_w * (16.0 / 36.0) / (ms_w * 1e-3) / 1e12 if ms_w else 0.0
        if ms_h >= ms_b and ms_h >= ms_w and ms_h >= ms_d:
            # dominant: Winograd on the fp16 matrix cores (scaled two-way split, 3 MFMAs per product); HIP events around the entry
            # point include the absmax pass over the input that fixes the scale
            exec_tf = fl_h * (16.0 / 36.0) * 3.0 / (ms_h * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "cnl_wino5::winograd5_kernel (F(2x2,3x3); fp32 operands scaled by a power of two and split into 2 fp16 "
                                               "pieces, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; incl. cnl_wino6::winograd6_kernel and cnl_wino7::winograd7_kernel, its 128-cout and two-waves-per-SIMD forms, and the few absmax_kernel passes)",
                    "achieved": round(exec_tf, 2), "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(exec_tf / F16_MFMA_PEAK_TFLOPS, 4),
                    "achieved_counts": "executed fp16 matrix-core flops = direct-conv flops x 16/36 (Winograd) x 3 (split terms)",
                    "effective_tflops": round(fl_h / (ms_h * 1e-3) / 1e12, 2),
                    "launches_per_step": n_h, "kernel_ms_per_step": round(ms_h, 3),
                    "algorithmic_gflop_per_step": round(fl_h / 1e9, 2), "avg_launch_us": round(ms_h * 1e3 / n_h, 2),
                    "sustained_clock_note": "power-limited DVFS: the split-operand kernels hold 1.8-2.0 GHz of 2.4 (tools/clk_probe.sh)"}
        elif ms_b >= ms_w and ms_b >= ms_d:      # dominant kernel: Winograd on the bf16 matrix cores (exact 3-way split, 6 MFMAs per product)
            exec_tf = fl_b * (16.0 / 36.0) * 6.0 / (ms_b * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "cnl_wino3::winograd3_kernel (F(2x2,3x3); fp32 operands split exactly into 3 bf16 pieces, "
                                               "6 x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate)",
                    "achieved": round(exec_tf, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(exec_tf / BF16_MFMA_PEAK_TFLOPS, 4),
                    "achieved_counts": "executed bf16 matrix-core flops = direct-conv flops x 16/36 (Winograd) x 6 (split terms)",
                    "effective_tflops": round(fl_b / (ms_b * 1e-3) / 1e12, 2),
                    "launches_per_step": n_b, "kernel_ms_per_step": round(ms_b, 3),
                    "algorithmic_gflop_per_step": round(fl_b / 1e9, 2), "avg_launch_us": round(ms_b * 1e3 / n_b, 2),
                    "sustained_clock_note": "the chip sustains ~1.6-1.65 GHz under this kernel (power-limited DVFS; tools/wino3_trace.py): "
                                            "bf16 ceiling at that clock ~1.7 PFLOP/s"}
        elif ms_w >= ms_d:          # dominant kernel: Winograd on the fp32 matrix cores
            roof = {"bound": "mfma", "kernel": "cnl_wino2::winograd2_kernel (F(2x2,3x3), fp32 v_mfma_f32_32x32x2_f32)",
                    "achieved": round(f32_exec_tf, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(f32_exec_tf / FP32_MFMA_PEAK_TFLOPS, 4),
                    "achieved_counts": "executed matrix-core flops = direct-conv flops x 16/36",
                    "effective_tflops": round(fl_w / (ms_w * 1e-3) / 1e12, 2),
                    "launches_per_step": n_w, "kernel_ms_per_step": round(ms_w, 3),
                    "algorithmic_gflop_per_step": round(fl_w / 1e9, 2), "avg_launch_us": round(ms_w * 1e3 / n_w, 2),
                    "sustained_clock_note": "chip sustains ~2.1 GHz under this load (DVFS; profiles/r01_mfma_peak_onbox.txt), i.e. ~140 TFLOP/s ceiling"}
        else:
            roof = {"bound": "mfma", "kernel": "cnl_conv::conv_mfma_kernel (fp32 v_mfma_f32_32x32x2_f32 implicit GEMM)",
                    "achieved": round(direct_tf, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(direct_tf / FP32_MFMA_PEAK_TFLOPS, 4), "launches_per_step": n_d,
                    "kernel_ms_per_step": round(ms_d, 3), "algorithmic_gflop_per_step": round(fl_d / 1e9, 2),
                    "avg_launch_us": round(ms_d * 1e3 / max(n_d, 1), 2)}
        roof["traffic"] = MEASURED_TRAFFIC_BYTES_PER_LAUNCH.get((args.config, B, H, W, roof["kernel"].split(" ")[0]))
        roof["traffic_unit"] = "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_traffic_final9.txt: 29 winograd5 launches)"
        dom = "winograd_f16x2" if "wino5" in roof["kernel"] else "winograd_bf16x3" if "wino3" in roof["kernel"] else ("winograd_f32" if "wino2" in roof["kernel"] else "direct")
        roof["algorithmic_bytes_per_launch"] = round(conv_bytes.get(dom, 0) / max(roof["launches_per_step"], 1))
        roof["other_kernels"] = {"cnl_wino2::winograd2_kernel (fp32 MFMA)": {"launches_per_step": n_w, "kernel_ms_per_step": round(ms_w, 3),
                                                                             "achieved_tflops_executed": round(f32_exec_tf, 2),
                                                                             "frac_of_fp32_mfma_peak": round(f32_exec_tf / FP32_MFMA_PEAK_TFLOPS, 4)},
                                 "cnl_wino3::winograd3_kernel (bf16 MFMA, exact 3-way split)": {"launches_per_step": n_b, "kernel_ms_per_step": round(ms_b, 3)},
                                 "cnl_wino5::winograd5_kernel (fp16 MFMA, scaled 2-way split)": {"launches_per_step": n_h, "kernel_ms_per_step": round(ms_h, 3)},
                                 "cnl_conv::conv_mfma_kernel": {"launches_per_step": n_d, "kernel_ms_per_step": round(ms_d, 3),
                                                                "achieved_tflops": round(direct_tf, 2)},
                                 "cnl_conv::conv_f16x2_kernel (direct conv, fp16 MFMA, scaled 2-way split)": {
                                     "launches_per_step": n_d5, "kernel_ms_per_step": round(ms_d5, 3),
                                     "effective_tflops": round(fl_d5 / (ms_d5 * 1e-3) / 1e12, 2) if ms_d5 else 0.0}}
        ms_per_step = elapsed / args.steps * 1e3
        result = {
            "metric": "images/sec @512x512 ResNet34 CenterNet forward + gather_detection2d",
            "value": round(world * B * args.steps / elapsed, 2),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": "fp32 in / fp32 accumulate / fp32 out; the long-channel 3x3 layers, the stride-2 / heatmap direct convs and the stem form each fp32 product on the fp16 matrix cores from "
                          "a two-way fp16 split of both (power-of-two scaled) operands (3 cross terms, error <= the fp32 MFMA's: tools/bf16x3_probe.hip, "
                          "tests/test_gpu_conv.py::test_winograd_split_kernels_error_not_above_fp32_mfma, test_conv_f16x2_error_not_above_fp32_mfma, test_stem_f16x2_error_not_above_fp32_mfma; fp32_mfma_only = the same job with no split operands anywhere)",
            "data": "synthetic (seeded rand images; random-init weights of the named architecture)",
            "config": {"workload": f"BASELINE C{'1' if args.config == 'simple' else ('4' if tracking else '2/3')}: ResNet34 + {args.config} neck, "
                                   f"{B} img/GPU x {H}x{W}, heads {'2+4+reid64' if tracking else '80+4'} (w256), k={args.k}, nms 3",
                       "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"batch-shard x{world}" + (" + RCCL all-gather of detections" if world > 1 else "")},
            "roofline": roof,
            "conv_stack": {"algorithmic_gflop_per_step": round(conv_flops / 1e9, 2), "kernel_ms_per_step": round(conv_ms, 3),
                           "effective_tflops": round(conv_flops / (conv_ms * 1e-3) / 1e12, 2)},
            "decode_p50_ms": round(lat[len(lat) // 2], 4),
        }
        if args.layers:
            for what, fl, ms, _kind in rows:
                print(f"{what:44s} {fl / 1e9:10.2f} GFLOP {ms * 1e3:10.1f} us {fl / (ms * 1e-3) / 1e12 if ms else 0:8.1f} TF", file=sys.stderr)
        if world == 1 and not args.no_fp32_mfma_leg and not os.environ.get("CNL_WINO"):
            # the same job with every 3x3 layer on the fp32 matrix core (the kernel choice is read once per process: child process)
            import subprocess
            env = dict(os.environ, CNL_WINO="2", CNL_CONV_F16X2="0", CNL_STEM_F16X2="0")
            cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(max(args.steps // 2, 3)), "--warmup", str(min(args.warmup, 3)),
                   "--config", args.config, "--batch", str(B), "--height", str(H), "--width", str(W), "--k", str(args.k),
                   "--no-cpu-baseline", "--no-fp32-mfma-leg", "--oracle-check"]
            try:
                out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1]
                alt = json.loads(out)
                result["fp32_mfma_only"] = {"value": alt["value"], "unit": alt["unit"], "ms_per_step": alt["ms_per_step"], "steps": alt["steps"],
                                            "roofline_frac_of_fp32_mfma_peak": alt["roofline"]["frac"],
                                            "max_abs_err_heatmap_vs_cpu_oracle": alt.get("oracle_check", {}).get("max_abs_err_heatmap"),
                                            "note": "CNL_WINO=2 CNL_CONV_F16X2=0 CNL_STEM_F16X2=0: every conv on v_mfma_f32_32x32x2_f32 (no split operands anywhere)"}
            except Exception as e:      # reported, never fatal: `value` above is the measurement
                result["fp32_mfma_only"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(model, tracking, args.k, H, W)
            result["accuracy"] = {"max_abs_err_heatmap_vs_cpu_oracle": result["cpu_baseline"]["max_abs_err_heatmap"],
                                  "same_with_every_layer_on_the_fp32_mfma": result.get("fp32_mfma_only", {}).get("max_abs_err_heatmap_vs_cpu_oracle"),
                                  "tolerance": 1e-4, "sample": "2 images of the bench shape, same weights"}
        elif world == 1 and args.oracle_check:
            result["oracle_check"] = {"max_abs_err_heatmap": oracle_check(model, H, W)}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
