#!/usr/bin/env python3
"""Benchmark of the denoising hot path: SR3 UNet p_sample steps over GAE-latent batches on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one reverse-diffusion step (noise embedding + full 97.8 M-parameter UNet forward + fused
posterior update) over this GPU's batch of latents: `--patches` CAVE patches x 5 spectral groups, each a
(3,128,128) latent conditioned on its low-resolution latent (BASELINE.json configs[1]; cosine T=1000).
Inputs are resident in HBM before the timed region; the step is a captured HIP graph.  Rank 0 prints ONE
JSON line.  value = steps * total batch / seconds  ("UNet denoise-steps/sec x batch").

Extra objects in the line:
  roofline     - the implicit-GEMM conv kernel family (the dominant kernel): algorithmic FLOPs of its launches
                 in one step / their HIP-event durations, against the dense bf16 MFMA peak.
  cpu_baseline - the oracle (CPU restatement of the reference, oracle/) timed on this host, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FULL_CFG = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8],
                attn_res=[16], res_blocks=2, image_size=128)
SCHED = dict(schedule="cosine", n_timestep=1000, linear_start=1e-6, linear_end=1e-2)
GROUPS = 5                 # CAVE: 31 bands, n_subs=8, n_ovls=2 -> 5 spectral groups (AE.py:263)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0           # HBM3E spec (6290 measured by a streaming read), MI355X_MICROARCH.md


def build_model(dev, precision):
    from hsi_dmgasr_amd.init import init_weights_orthogonal
    from hsi_dmgasr_amd.sr3_modules import diffusion, unet
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8],
                  attn_res=[16], res_blocks=2, dropout=0.2, image_size=128, precision=precision)
    init_weights_orthogonal(u, seed=0)                      # model/networks.py:110-112
    gd = diffusion.GaussianDiffusion(u, image_size=128, channels=3, loss_type="l1", conditional=True)
    gd = gd.to(dev).eval()
    gd.set_loss(dev)
    gd.set_new_noise_schedule(SCHED, dev)
    gd.noise, gd.seed = "philox", 2
    return gd


def conv_roofline(run, batch, reps=3):
    """Roofline of the DOMINANT kernel of one step.  Every conv launch of one eager step is bracketed with HIP events
    on the launch stream (best of `reps`); launches are grouped by the kernel the C ABI dispatches to
    (hsidm_conv_kernel_id) and the group with the largest total time is reported:
    achieved = sum of algorithmic FLOPs of its launches / sum of their durations."""
    from hsi_dmgasr_amd import ops
    best = None
    for _ in range(reps):
        recs = []
        ops.set_conv_probe(recs)
        run._enqueue()
        ops.set_conv_probe(None)
        torch.cuda.synchronize()
        for r in recs:
            r["ms"] = r["e0"].elapsed_time(r["e1"])
        if best is None:
            best = recs
        else:
            for a, b in zip(best, recs):
                a["ms"] = min(a["ms"], b["ms"])
    groups = {}
    for r in best:
        g = groups.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
        g["ms"] += r["ms"]; g["flops"] += r["flops"]; g["bytes"] += r["bytes"]; g["n"] += 1
    name, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
    all_ms = sum(g["ms"] for g in groups.values())
    all_fl = sum(g["flops"] for g in groups.values())
    top = max((r for r in best if r["kernel"] == name), key=lambda r: r["flops"] / r["ms"])
    traffic = None
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")     # PMC passes (FETCH_SIZE x2 + WRITE_SIZE), see profiles/README.md
    if os.path.exists(tf):
        traffic = json.load(open(tf)).get(name)
        if traffic is not None and traffic.get("batch_per_gpu") not in (None, batch):
            traffic = None                                      # counters were collected at another batch: not this run's traffic
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    # the launch closest to the HBM roof (north-star: "fraction of the HBM roofline on the fused ResnetBlock kernel"): a 3x3
    # GN+SiLU conv of the 128x128 level; algorithmic bytes = activations in + out + weights, against the 8 TB/s spec
    hb = max((r for r in best if r["ksize"] == 3 and r["stride"] == 1 and "gn+silu" in r["kernel"]), key=lambda r: r["bytes"] / r["ms"])
    hbm_view = dict(kernel=hb["kernel"], cin=hb["cin"], cout=hb["cout"], hw=list(hb["hw"]), us=hb["ms"] * 1e3,
                    algorithmic_bytes=hb["bytes"], achieved_GBps=hb["bytes"] / (hb["ms"] * 1e-3) / 1e9, peak_GBps=HBM_PEAK_GBPS,
                    frac=hb["bytes"] / (hb["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    tflops=hb["flops"] / (hb["ms"] * 1e-3) / 1e12)
    return dict(bound="mfma", hbm_view=hbm_view, achieved=ach, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / MFMA_BF16_PEAK_TFLOPS,
                traffic=traffic, kernel=name, launches=dom["n"], avg_launch_us=dom["ms"] / dom["n"] * 1e3,
                algorithmic_flops_per_launch=dom["flops"] / dom["n"], algorithmic_bytes_per_launch=dom["bytes"] / dom["n"],
                share_of_conv_time=dom["ms"] / all_ms,
                best_launch=dict(cin=top["cin"], cout=top["cout"], hw=list(top["hw"]), us=top["ms"] * 1e3,
                                 tflops=top["flops"] / (top["ms"] * 1e-3) / 1e12),
                all_conv_kernels=dict(launches=len(best), flops_per_step=all_fl, ms_per_step=all_ms,
                                      tflops=all_fl / (all_ms * 1e-3) / 1e12))


def cpu_baseline(batch=1, steps=160, warm=2):
    """The oracle on the host CPU: full-size UNet p_sample steps (fp32), a bounded sample (about 10-25 s) of the same
    workload: 160 of the 1000 reverse steps at batch 1 (every step costs the same)."""
    from oracle import diffusion as odiff, sr3_unet
    from hsi_dmgasr_amd.init import init_weights_orthogonal
    from hsi_dmgasr_amd.sr3_modules import unet
    threads = min(usable_cpus(), 64)
    torch.set_num_threads(threads)
    u = unet.UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8],
                  attn_res=[16], res_blocks=2, dropout=0.2, image_size=128)
    init_weights_orthogonal(u, seed=0)
    sd = {k: v.detach() for k, v in u.state_dict().items()}
    sched = odiff.noise_schedule(SCHED)
    g = torch.Generator().manual_seed(1)
    cond = torch.randn((batch, 3, 128, 128), generator=g).clamp(-2.5, 2.5)
    x = torch.randn((batch, 3, 128, 128), generator=g)
    den = lambda xx, gam: sr3_unet.unet_forward(sd, FULL_CFG, xx, gam)
    with torch.no_grad():
        t0 = None
        for k in range(warm + steps):
            if k == warm:
                t0 = time.perf_counter()
            i = 999 - k
            x = odiff.p_sample_step(den, sched, x, cond, i, torch.randn(x.shape, generator=g))
        dt = time.perf_counter() - t0
    return dict(value=steps * batch / dt, unit="denoise-steps*batch/s", cores=threads, kind="port",
                sample="%d p_sample steps of the full UNet at batch %d (fp32 oracle, %d threads), %.1f s" %
                       (steps, batch, threads, dt))


def log(msg):
    if os.environ.get("HSIDM_BENCH_VERBOSE"):
        print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def usable_cpus():
    """CPUs this process may really use: affinity mask and cgroup quota, not the machine's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--patches", type=int, default=48,
                    help="CAVE patches per GPU (x5 spectral groups = batch of 240 latents); throughput saturates around here: "
                         "120 latents -3 %, 640 latents +2 % (sweep in DESIGN.md)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    torch.set_num_threads(min(usable_cpus(), 64))      # the host may expose 256 CPUs behind a 16-CPU quota
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("HSIDM_FORCE_DIST")        # FORCE: exercise the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm

    from hsi_dmgasr_amd import parallel
    log('building model')
    gd = build_model(dev, args.precision)
    log('model on device')
    parallel.broadcast_module_(gd, src=0)                  # one-time weight broadcast over xGMI
    batch = args.patches * GROUPS
    g = torch.Generator(device="cpu").manual_seed(1 + rank)
    cond = torch.randn((batch, 3, 128, 128), generator=g).clamp(-2.5, 2.5).to(dev)   # GAE latent range
    run = gd.make_run(cond, wrap=True)

    with torch.no_grad():
        for _ in range(max(args.warmup, 2)):               # >= 2: eager step, then graph capture + first replay
            run.step()
        torch.cuda.synchronize()
        log('warmup done')
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run.step()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        el = torch.tensor([dt], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dt = float(el.item())
        assert torch.isfinite(run.x).all(), "sampler state diverged"
        log('timed region done: %.3f s' % dt)

        roof = None
        if rank == 0 and not args.no_roofline:
            roof = conv_roofline(run, batch)
    log('roofline done')
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
        log('cpu baseline done')

    if rank == 0:
        total_batch = batch * world
        line = {
            "metric": "UNet denoise-steps/sec x batch, CAVE 31-band 16->128, 1000-step p_sample_loop",
            "value": args.steps * total_batch / dt,
            "unit": "denoise-steps*batch/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision if args.precision == "bf16" else "fp32 (bf16x3 split)",
            "data": "synthetic (orthogonal-init weights seed 0, N(0,1) latents clipped to +-2.5, Philox noise)",
            "config": {"workload": "SR3 UNet 97.8M (6->3 ch, inner 64, mults 1-2-4-8-8, attn@16) p_sample step on "
                                   "GAE latents 3x128x128, cosine T=1000, BASELINE configs[1]",
                       "patches_per_gpu": args.patches, "groups_per_patch": GROUPS,
                       "batch_per_gpu": batch, "global_batch": total_batch, "parallelism": "dp%d" % world},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
