#!/usr/bin/env python3
"""Benchmark of the SR hot path: training frames/s at BASELINE cfg2 (scale=2, feat=64, blocks=8,
T=3, 540p -> 1080p), one process per GPU, weak scaling.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Run without WORLD_SIZE in the environment, `--gpus N` (N > 1) starts the N ranks itself: this process launches
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a CHILD process before it
touches the GPU (never an exec), relays rank 0's JSON line and exits with the child's return code.  Launched by
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank of that job.

A step = zero_grad -> forward -> MSE -> backward (-> RCCL all-reduce of the flat gradient
bucket) -> AdamW step, on synthetic U[0,1) clips with T distinct frames, resident in HBM before
the timed region.  Rank 0 prints ONE JSON line (contract in the round prompt).  `roofline` is the
dominant kernel's algorithmic FLOP rate measured with HIP events around each of its launches in
the timed region; `cpu_baseline` is the CPU oracle (a PyTorch-CPU restatement of the reference)
timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 (matrix)
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 MFMA peak (same guide); ridge = 2500 / 8 = 312 FLOP/B
# algorithmic bytes of one train step per clip in bf16 storage (3 x the forward's 16 566 planes at cfg2)
ALGO_GB_PER_CLIP_STEP = {"cfg2": 51.53, "cfg4 (SR part)": 16.66}


def sources_sha() -> str:
    """Fingerprint of what determines the kernels' HBM traffic: every HIP source / header and the launch schedule.
    tools/pmc_traffic.py stores it in the PMC summary it writes; a summary taken at other sources is not quoted."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.h")))
    files += [os.path.join(PKG, "nerve_cl", "_engine.py"), os.path.join(PKG, "nerve_cl", "_nvq.py")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel: str, batch: int, workload: str):
    """(GB per launch, file name) of `kernel` from a committed rocprofv3 PMC summary (FETCH_SIZE and WRITE_SIZE collected
    in two separate passes of this script, corrected as MI355X_MICROARCH.md prescribes) - but only from a summary whose
    recorded source fingerprint, workload and batch are THIS run's; (None, None) otherwise.  bench.py cannot collect PMC
    counters itself."""
    import glob
    sha = sources_sha()
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "*_traffic_pmc.json")), reverse=True):
        try:
            blob = json.load(open(path))
        except (OSError, ValueError):
            continue
        if blob.get("sources_sha") != sha or blob.get("workload") != workload or blob.get("batch") != batch:
            continue
        v = blob.get("kernels", {}).get(kernel)
        if v:
            return v["hbm_bytes_per_launch"] / 1e9, os.path.basename(path)
    return None, None


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(args, cfg, parity_probe=None):
    """Time the oracle's training step on the host cores on a BOUNDED sample: one clip at 1/16 of
    the pixels (135x240) and scale the rate by the pixel ratio.  The CPU path is super-linear in
    pixels (1.05 s at 135x240, 16.5 s at 270x480, ~75 GB of autograd state at
    540p), so this extrapolation flatters the CPU; it is a reported baseline, not the target.

    The same leg is the run's parity check ("PSNR vs ref", the second half of BASELINE.json's metric): parity_probe =
    (state_dict, clips, targets, GPU output, GPU loss) of a small crop that went through the benchmarked mode; the oracle
    evaluates it in fp32 and PSNR (formula of experiments/train_baseline.py:27-32) / relative loss are reported."""
    from oracle import sr_oracle
    from nerve_cl.models import SuperResolutionNet
    cores = host_cores()
    torch.set_num_threads(cores)
    parity = None
    if parity_probe is not None:
        sd, xs, ys, out_gpu, loss_gpu = parity_probe
        ora = sr_oracle.OracleSR(3, cfg["scale"], cfg["F"], cfg["blocks"], cfg["window"])
        ora.load_named(sd)
        ora.train()
        with torch.no_grad():
            out_cpu = ora(xs)
            loss_cpu = F.mse_loss(out_cpu, ys).item()
        parity = {"psnr_vs_oracle_db": sr_oracle.compute_psnr(out_gpu, out_cpu),
                  "loss_rel_vs_oracle": abs(loss_gpu - loss_cpu) / loss_cpu,
                  "max_abs_err": (out_gpu - out_cpu).abs().max().item(),
                  "probe": f"{xs.shape[0]} clips of {xs.shape[-2]}x{xs.shape[-1]} cropped from the benchmark batch, the "
                           f"benchmarked weights and mode (train-mode BatchNorm) vs the fp32 CPU oracle"}
    Hc, Wc = cfg["H"] // 4, cfg["W"] // 4
    torch.manual_seed(0)
    ora = sr_oracle.OracleSR(3, cfg["scale"], cfg["F"], cfg["blocks"], cfg["window"])
    ora.load_named(SuperResolutionNet(3, cfg["scale"], cfg["F"], cfg["blocks"], cfg["window"]).state_dict())
    ora.train()
    opt = torch.optim.AdamW(ora.parameters(), lr=1e-4, weight_decay=1e-5)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(1, cfg["T"], 3, Hc, Wc, generator=g)
    y = torch.rand(1, 3, Hc * cfg["scale"], Wc * cfg["scale"], generator=g)
    times = []
    budget_t0 = time.perf_counter()
    for i in range(1 + args.cpu_steps):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = F.mse_loss(ora(x), y)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu baseline step {i}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
        if i >= 1 and time.perf_counter() - budget_t0 > 45.0:
            break
    timed = sorted(times[1:])
    t = timed[len(timed) // 2]
    ratio = (cfg["H"] * cfg["W"]) / (Hc * Wc)
    extra = {}
    if not args.no_cpu_extras:
        # the reference's own CPU-runnable case (BASELINE configs[0]: F=32, 4 blocks, B=16, 64x64) and the
        # measured full-size INFERENCE forward beside the extrapolated training rate
        try:
            torch.manual_seed(0)
            o1 = sr_oracle.OracleSR(3, 2, 32, 4, 1)
            o1.load_named(SuperResolutionNet(3, 2, 32, 4, 1).state_dict())
            o1.train()
            op1 = torch.optim.AdamW(o1.parameters(), lr=1e-3, weight_decay=1e-5)
            x1, y1 = torch.rand(16, 3, 3, 64, 64, generator=g), torch.rand(16, 3, 128, 128, generator=g)
            ts = []
            for i in range(3):
                t0 = time.perf_counter()
                op1.zero_grad()
                F.mse_loss(o1(x1), y1).backward()
                op1.step()
                ts.append(time.perf_counter() - t0)
                print(f"[bench] cpu cfg1 step {i}: {ts[-1]:.2f} s", file=sys.stderr, flush=True)
                if time.perf_counter() - budget_t0 > 90.0:
                    break
            if len(ts) > 1:
                extra["cfg1_train_clips_per_s"] = 16.0 / min(ts[1:])
                extra["cfg1_sample"] = (f"F=32, 4 blocks, T=3, B=16, 64x64 train step, best of {len(ts) - 1} after one "
                                        f"warm-up: {min(ts[1:]):.2f} s")
            if time.perf_counter() - budget_t0 < 100.0:
                ora.eval()
                xf = torch.rand(1, cfg["T"], 3, cfg["H"], cfg["W"], generator=g)
                with torch.no_grad():
                    t0 = time.perf_counter()
                    ora(xf)
                    tf = time.perf_counter() - t0
                extra["full_size_inference_forward_s"] = tf
                extra["full_size_inference_sample"] = (f"one eval-mode no-grad forward of one {cfg['H']}x{cfg['W']} clip "
                                                       f"(first call at this shape, primitive creation included)")
                print(f"[bench] cpu {cfg['H']}x{cfg['W']} inference forward: {tf:.2f} s", file=sys.stderr, flush=True)
        except MemoryError:
            pass
    return {
        "value": (1.0 / t) / ratio,
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "sample": f"median of {len(timed)} timed train step(s) (fwd+MSE+bwd+AdamW, fp32, {cores} threads) of the "
                  f"same net on one {Hc}x{Wc} clip = 1/{ratio:.0f} of the 540p pixels: {t:.2f} s/step; rate divided "
                  f"by {ratio:.0f} (flatters the CPU, whose cost grows faster than the pixel count); 1 warm-up "
                  f"step excluded",
        "parity": parity,
        **extra,
    }


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torch.distributed.run job (one process
    per GPU, rendezvous on 127.0.0.1) and return its exit code.  Called before this process has imported anything that
    touches the GPU; the child's stdout (rank 0's JSON line) and stderr pass straight through."""
    import socket
    import subprocess
    with socket.socket() as s:                  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--features", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--window", type=int, default=1)
    ap.add_argument("--scale", type=int, default=2)
    ap.add_argument("--math", choices=["bf16", "f32"], default="bf16",
                    help="MFMA operand precision of the convolutions (accumulation and storage are fp32)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--fp32-acts", action="store_true",
                    help="with --math bf16: keep the conv-internal tensors in fp32 (default: bf16 storage)")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-extras", action="store_true",
                    help="skip the cfg1 CPU timing and the full-size CPU inference forward of the cpu_baseline leg")
    ap.add_argument("--recovery", action="store_true",
                    help="BASELINE cfg4: EnhancementEngine = this SR net + the FrameRecoveryNet inpainting head (base 64), a "
                         "rectangular mask over 25 %% of the frame, loss = MSE(enhanced, hr) + MSE(recovered, lr_centre) "
                         "(SURVEY.md 8d)")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--detail", action="store_true", help="print a per-shape table of the conv launches to stderr")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))       # nothing above has initialised the GPU

    from nerve_cl import _nvq, ops, parallel
    from nerve_cl.models import SuperResolutionNet

    if args.backend != "nccl":
        local_dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_dev)
    rank, world, local = parallel.init_from_env(args.backend)
    if args.backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)     # rehearsal: several ranks may share one GPU
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is what runs",
                  file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = dict(F=args.features, blocks=args.blocks, window=args.window, T=2 * args.window + 1,
               scale=args.scale, H=args.height, W=args.width)

    is_cfg2 = (cfg["scale"], cfg["F"], cfg["blocks"], cfg["T"], cfg["H"], cfg["W"]) == (2, 64, 8, 3, 540, 960)
    is_cfg4 = (cfg["scale"], cfg["F"], cfg["blocks"], cfg["T"], cfg["H"], cfg["W"]) == (4, 64, 8, 5, 270, 480)
    tag = "cfg2" if is_cfg2 else ("cfg4" if args.recovery else "cfg4 (SR part)") if is_cfg4 else "custom"
    workload = (f"{tag}: SuperResolutionNet(scale={cfg['scale']}, feat={cfg['F']}, blocks={cfg['blocks']}, T={cfg['T']}) "
                + ("+ FrameRecoveryNet(base=64) inpainting head (25 % mask, two-term loss) " if args.recovery else "")
                + f"train step on {cfg['H']}x{cfg['W']} -> {cfg['H'] * cfg['scale']}x{cfg['W'] * cfg['scale']} clips")

    torch.manual_seed(0)
    engine = None
    if args.recovery:
        from nerve_cl.models import EnhancementConfig, EnhancementEngine
        engine = EnhancementEngine(EnhancementConfig(
            scale_factor=cfg["scale"], sr_num_features=cfg["F"], sr_num_residual_blocks=cfg["blocks"],
            sr_temporal_window=cfg["window"], recovery_base_channels=64, recovery_temporal_window=cfg["window"])).to(dev).train()
        net = engine.super_resolution
        engine.frame_recovery.math_mode = _nvq.MATH_BF16 if args.math == "bf16" else _nvq.MATH_F32
    else:
        net = SuperResolutionNet(3, cfg["scale"], cfg["F"], cfg["blocks"], cfg["window"]).to(dev).train()
    model = engine if engine is not None else net
    net.math_mode = _nvq.MATH_BF16 if args.math == "bf16" else _nvq.MATH_F32
    net.bf16_activations = args.math == "bf16" and not args.fp32_acts
    if world > 1:
        parallel.enable_data_parallel(model)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    B = args.batch
    x = torch.rand(B, cfg["T"], 3, cfg["H"], cfg["W"], device=dev, generator=g)
    y = torch.rand(B, 3, cfg["H"] * cfg["scale"], cfg["W"] * cfg["scale"], device=dev, generator=g)
    mask = None
    if args.recovery:                            # ones over the central quarter of the frame
        mask = torch.zeros(B, 1, cfg["H"], cfg["W"], device=dev)
        mask[:, :, cfg["H"] // 4:cfg["H"] // 4 + cfg["H"] // 2, cfg["W"] // 4:cfg["W"] // 4 + cfg["W"] // 2] = 1.0
        x_centre = x[:, cfg["T"] // 2].contiguous()

    def step():
        opt.zero_grad()
        if engine is None:
            loss = ops.mse_loss(net(x), y)      # nn.MSELoss of the reference loop as libnvq kernels
        else:
            # the second term is what gives the recovery head a gradient: its output does not feed the SR result
            # (reference enhancement_engine.py:143-148)
            res = engine(x, corruption_mask=mask)
            loss = ops.mse_loss(res["enhanced"], y) + ops.mse_loss(res["recovered"], x_centre)
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()

    def timed_pass():
        if world > 1:
            parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        if world > 1:
            parallel.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = tt.item()
        return dt, loss

    # pass A: exactly K steps, nothing but the workload on the stream -> `value`
    dt, loss = timed_pass()
    # pass B: the same K steps again with a HIP event pair around every conv launch -> `roofline`
    # (the event records cost ~10 % of a step, so they are kept out of pass A)
    timer = None
    dt_b = None
    if not args.no_kernel_timer:
        timer = _nvq.KernelTimer()
        _nvq.TIMER = timer
        dt_b, loss = timed_pass()
        _nvq.TIMER = None
    final_loss = loss.item()

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    roofline = None
    residual_stack = None
    kernels = {}
    if timer is not None:
        kernels = timer.summary()
        if kernels:
            name, d = max(kernels.items(), key=lambda kv: kv[1]["ms_total"])
            tflops = d["flops"] / (d["ms_total"] * 1e-3) / 1e12
            gbs = d["bytes"] / (d["ms_total"] * 1e-3) / 1e9
            traffic, tfile = pmc_traffic(name, B, workload) if args.math == "bf16" and not args.fp32_acts else (None, None)
            common = {"traffic": traffic, "traffic_unit": f"GB per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC, "
                                                          f"profiles/{tfile}, same sources fingerprint {sources_sha()})"
                      if traffic else None,
                      "algorithmic_gb_per_launch": d["bytes"] / d["launches"] / 1e9,
                      "kernel": name, "launches": d["launches"],
                      "avg_launch_ms": d["ms_total"] / d["launches"],
                      "share_of_step": d["ms_total"] / (dt_b * 1e3),
                      "measured": "second pass of the same K steps with a HIP event pair around each conv launch; "
                                  f"{getattr(timer, 'outliers', 0)} of {len(timer.records)} pairs that also caught a host launch "
                                  "gap (> 2x the median of their shape) counted at that median"}
            if args.math == "f32":      # exact-fp32 MFMA: compute-bound (157 TF vs 8 TB/s => ridge at 20 FLOP/B)
                roofline = {"bound": "mfma", "achieved": tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": tflops / PEAK_F32_MFMA_TFLOPS, "algorithmic_gbs": gbs, **common}
            else:                       # bf16 MFMA on fp32-stored activations: HBM-bound
                roofline = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": gbs / PEAK_HBM_GBS, "algorithmic_tflops": tflops, **common}
                # the whole step against the layer-wise HBM bound of SURVEY.md 8(d): (algorithmic GB per clip and step x
                # clips per second of pass A) / peak - tracks the headline, not one kernel
                if tag in ALGO_GB_PER_CLIP_STEP and net.bf16_activations:
                    roofline["whole_step_frac"] = ALGO_GB_PER_CLIP_STEP[tag] * value / PEAK_HBM_GBS
                    roofline["whole_step_algorithmic_gb_per_clip"] = ALGO_GB_PER_CLIP_STEP[tag]
    if timer is not None and kernels:
        # the residual dense stack as a whole (north-star: >= 40 % of the HBM roofline): every conv / weight-gradient launch
        # of the 8 dense blocks, forward and backward, algorithmic bytes over measured time
        rdb = timer.by_tag().get("rdb")
        if rdb and rdb["ms_total"] > 0:
            gbs = rdb["bytes"] / (rdb["ms_total"] * 1e-3) / 1e9
            residual_stack = {"launches_per_step": rdb["launches"] // args.steps,
                              "ms_per_clip": rdb["ms_total"] / args.steps / B,
                              "algorithmic_gb_per_clip": rdb["bytes"] / args.steps / B / 1e9,
                              "achieved_gbs": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS,
                              "tflops": rdb["flops"] / (rdb["ms_total"] * 1e-3) / 1e12}
    if timer is not None and args.detail:
        rows = sorted(timer.by_shape().items(), key=lambda kv: -kv[1]["ms_total"])
        # `of bound`: the shape's fraction of ITS roofline - HBM (8 TB/s) when its arithmetic intensity is under the bf16
        # ridge (312 FLOP/B; 20 FLOP/B with exact-fp32 MFMA operands), the MFMA peak above it
        peak_tf = PEAK_BF16_MFMA_TFLOPS if args.math == "bf16" else PEAK_F32_MFMA_TFLOPS
        ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
        print("%-24s %-38s %5s %9s %9s %9s %7s %9s" % ("kernel", "shape", "calls", "ms/step", "TFLOP/s", "alg GB/s",
                                                      "FLOP/B", "of bound"), file=sys.stderr)
        for (label, shape), d in rows[:48]:
            sec = d["ms_total"] * 1e-3
            ai = d["flops"] / max(d["bytes"], 1)
            frac = d["flops"] / sec / 1e12 / peak_tf if ai >= ridge else d["bytes"] / sec / 1e9 / PEAK_HBM_GBS
            print("%-24s %-38s %5d %9.3f %9.1f %9.0f %7.0f %6.2f %s" % (
                label, shape, d["launches"] // args.steps, d["ms_total"] / args.steps, d["flops"] / sec / 1e12,
                d["bytes"] / sec / 1e9, ai, frac, "mfma" if ai >= ridge else "hbm"), file=sys.stderr)
    # "PSNR vs ref" (second half of BASELINE.json's metric): a crop of the benchmark batch through the benchmarked mode,
    # checked against the fp32 CPU oracle inside the cpu_baseline leg
    parity_probe = None
    if world == 1 and not args.no_cpu_baseline:
        hs, ws_ = min(cfg["H"], 64), min(cfg["W"], 96)
        xs = x[:2, :, :, :hs, :ws_].contiguous()
        ys = y[:2, :, :hs * cfg["scale"], :ws_ * cfg["scale"]].contiguous()
        with torch.no_grad():
            out_s = net(xs)
            loss_s = ops.mse_loss(out_s, ys).item()
        parity_probe = ({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, xs.cpu(), ys.cpu(),
                        out_s.cpu(), loss_s)
    line = {
        "metric": "train frames/sec (2x SR, T=3, 540p->1080p)" if is_cfg2 else
                  f"train frames/sec ({cfg['scale']}x SR, T={cfg['T']}, {cfg['H']}x{cfg['W']})",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.math == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "lr_input_frames_per_s": value * cfg["T"], "final_loss": final_loss,
                   "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                   "conv_internal_storage": "bf16" if net.bf16_activations else "f32"},
        "roofline": roofline,
        "residual_stack": residual_stack,
    }
    if kernels:
        line["ms_per_step_with_kernel_events"] = dt_b / args.steps * 1e3
        line["kernel_ms_per_step"] = {k: round(v["ms_total"] / args.steps, 3) for k, v in
                                      sorted(kernels.items(), key=lambda kv: -kv[1]["ms_total"])}
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args, cfg, parity_probe)
        par = line["cpu_baseline"].get("parity")
        if par:
            line["psnr_vs_oracle_db"] = par["psnr_vs_oracle_db"]
            line["loss_rel_vs_oracle"] = par["loss_rel_vs_oracle"]
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
