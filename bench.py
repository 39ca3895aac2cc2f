"""Headline benchmark: volumes/s of a full VNet3d train step (fwd + BinaryDiceLoss + backward +
[RCCL grad all-reduce] + AdamW + weight re-pack) at 4 x 1 x 96^3 fp16 per GPU — BASELINE.json
configs[2], the configuration the metric is quoted on.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; the batch shards on the batch axis (weak scaling: 4 volumes per GPU); the flat
fp32 gradient buffer is all-reduced over RCCL (sum, then 1/N) before the fused optimiser step.  Inputs
are resident in HBM before the timed region.  Prints ONE JSON line on rank 0 with the extra objects
  "roofline"           – the kernel FAMILY with the most GPU time of the step, chosen from the complete table (every launch of both streams bracketed with HIP
                         events in R instrumented steps run AFTER the timed region); "roofline_2" / "roofline_3" are the next two, "kernel_families" the table
  "other_configs"      – ms per train step of the other BASELINE.json configs and of the f32 (parity-exact) run dtype of the headline workload, 10 steps each
  "cpu_baseline"       – the oracle (torch-CPU port of the reference path) on this box's host cores
  "gpu_torch_baseline" – the same oracle functions on this GPU through stock PyTorch-ROCm / MIOpen (fp32 and autocast f16):
                         the number the hand-written engine has to beat.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # see pytorchdeeplearing_amd/__init__.py (set before the HIP runtime starts)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per 96^3 volume and train step
GFLOP_PER_VOLUME_96 = 216.5
GB_PER_VOLUME_96 = 2.85
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (~6.3 TB/s achievable)
PEAK_MFMA_TFLOPS = 2500.0      # dense f16/bf16 MFMA
# Kernel FAMILIES of the train step (every launch of the step belongs to exactly one; the profile classes of include/segengine.h grouped):
# which roofline bounds the family, what it is, and which kernel symbols of the rocprofv3 PMC summary (profiles/summarize_pmc.py) belong to it
# ("count": the symbols whose launches are one bracketed op of the family - a weight gradient is main kernel + partial-tile reduce).
FAMILIES = {
    "halo_conv": dict(classes=["conv3", "conv3_smallbox"], bound="mfma", queue="main",
                      what="3^d halo convolutions, forward + data-gradient (c3x::conv3x_kernel / conv3x16_kernel / conv3x16r_kernel, all tilings)",
                      pmc=["_ZN3seg3c3x13conv3x_kernel", "_ZN3seg3c3x15conv3x16_kernel", "_ZN3seg3c3x16conv3x16r_kernel", "conv3_kernel"], count=None),
    "halo_wgrad": dict(classes=["wgrad3"], bound="mfma", queue="weight-gradient stream",
                       what="weight gradients of the 3^d convolutions (wgrad3_kernel + wgrad3_reduce_kernel)",
                       pmc=["wgrad3_kernel", "wgrad3_reduce_kernel"], count=["wgrad3_kernel"]),
    "generic_wgrad": dict(classes=["wgrad_generic"], bound="hbm", queue="weight-gradient stream",
                          what="weight gradients of the 2^d stride-2 / transposed / 1^d convolutions (wgrad_kernel / wgrad_direct_kernel + wgrad_reduce_kernel)",
                          pmc=["wgrad_kernel", "wgrad_direct_kernel", "wgrad_reduce_kernel"], count=["wgrad_kernel", "wgrad_direct_kernel"]),
    "generic_conv": dict(classes=["conv_generic"], bound="hbm", queue="main",
                         what="2^d stride-2, transposed and 1^d convolutions, forward + data-gradient (conv_stream_kernel / conv_igemm_kernel)",
                         pmc=["conv_stream_kernel", "conv_igemm_kernel"], count=None),
    "input_block": dict(classes=["stem"], bound="hbm", queue="main", what="fused input block, forward and backward (stemx_kernel, 4 passes)",
                        pmc=["stemx_kernel", "stemx_wgrad_reduce_kernel", "gn_finalize_kernel", "gn_bwd_finalize_kernel"], count=["stemx_kernel"]),
    "gn_act": dict(classes=["gn_act"], bound="hbm", queue="main", what="GroupNorm + dropout + ReLU (+ residual) forward (gn_act_kernel)",
                   pmc=["gn_act_kernel"], count=None),
    "gn_bwd_reduce": dict(classes=["gn_bwd_reduce"], bound="hbm", queue="main", what="GroupNorm backward, reduction pass (gn_bwd_reduce_kernel)",
                          pmc=["gn_bwd_reduce_kernel"], count=None),
    "gn_bwd_apply": dict(classes=["gn_bwd_apply"], bound="hbm", queue="main", what="GroupNorm backward, elementwise pass (gn_bwd_apply_kernel)",
                         pmc=["gn_bwd_apply_kernel"], count=None),
    "gn_small": dict(classes=["gn_group"], bound="hbm", queue="main",
                     what="one-launch GroupNorm passes of the >= 64-channel levels (gn_bwd_coop_kernel: 24^3 ... 6^3 backward; gn_fwd_group_kernel: 6^3 forward)",
                     pmc=["gn_fwd_group_kernel", "gn_bwd_group_kernel", "gn_bwd_coop_kernel"], count=None),
    "head": dict(classes=["head"], bound="hbm", queue="main", what="1^d head backward (head_bwd_kernel; the forward head runs inside the last gn_act launch)",
                 pmc=["head_fwd_kernel", "head_bwd_kernel"], count=None),
    "misc": dict(classes=["misc"], bound="hbm", queue="main", what="fill + ingest, loss, fused AdamW, weight re-pack",
                 pmc=["ingest", "loss_", "adam_kernel", "grad_check_kernel", "pack_kernel", "__amd_rocclr_fillBufferAligned", "dropout_mask_kernel"], count=None),
}


def _pmc_file():
    """newest committed PMC summary (profiles/rNN_pmc_fetch_write_per_kernel.json, regenerated from the final binary of a round by
    tools/gpu_final.sh with this file's own command line)"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_fetch_write_per_kernel.json")))
    return c[-1] if c else None


def _library_build():
    try:
        from pytorchdeeplearing_amd import _capi
        return _capi.product_library().build_info()
    except Exception:
        return None


def pmc_traffic(family):
    """HBM bytes per bracketed op of a kernel family from the rocprofv3 PMC passes committed under profiles/ (separate --pmc FETCH_SIZE /
    --pmc WRITE_SIZE runs of this same command; FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes): (bytes per op, file).
    (None, None) when the summary is not available."""
    try:
        f = _pmc_file()
        with open(f) as fh:
            table = json.load(fh)
        # the summary is stamped with seg_build_info() of the binary the PMC passes ran (profiles/summarize_pmc.py): figures of another binary are not reported
        if table.get("_build") != _library_build():
            return None, "%s is from build '%s', not the loaded library: traffic dropped" % (os.path.basename(f), table.get("_build"))
        fam = FAMILIES[family]
        rows = {k: v for k, v in table.items() if isinstance(v, dict) and any(k.startswith(p) or (p in k and not p.startswith("_Z")) for p in fam["pmc"])}
        # a family name that is a prefix of another family's symbol must not swallow it (wgrad_kernel vs wgrad3_kernel)
        if family == "generic_wgrad":
            rows = {k: v for k, v in rows.items() if "wgrad3" not in k}
        if family == "generic_conv":
            rows = rows
        total = sum(r["launches"] * (2.0 * r["fetch_kb_raw_per_launch"] + (r["write_kb_per_launch"] or 0.0)) for r in rows.values()) * 1024
        cnt = fam["count"]
        n = sum(r["launches"] for k, r in rows.items() if cnt is None or any(k.startswith(c) for c in cnt))
        return (int(total / n), os.path.basename(f)) if n else (None, None)
    except Exception:
        return None, None


# the other BASELINE.json configs (and the headline config in its parity-exact f32 run dtype), timed after the headline measurement:
# tag -> (net, ndim, shape, classes, loss, run dtype, soft-clDice weight, fused-bound GB per step from SURVEY.md section 8d)
OTHER_CONFIGS = {
    "C2 VNet2d 16x512^2 f16, 2 classes (BASELINE configs[1])": ("vnet", 2, (16, 1, 512, 512), 2, "MutilDiceLoss", "f16", 0.0, 21.72),
    "C4 UNet3d 2x128^3 f16, 4 classes (BASELINE configs[3], per-GPU shard)": ("unet", 3, (2, 1, 128, 128, 128), 4, "MutilDiceLoss", "f16", 0.0, 11.34),
    "C5 VNet3d 1x160^3 bf16, BCE + Dice (BASELINE configs[4] without the clDice term)": ("vnet", 3, (1, 1, 160, 160, 160), 1, "BinaryCrossEntropyDiceLoss", "bf16", 0.0, 13.14),
    "C5 VNet3d 1x160^3 bf16, Dice + soft-clDice (BASELINE configs[4] as worded)": ("vnet", 3, (1, 1, 160, 160, 160), 1, "BinaryDiceLoss", "bf16", 1.0, 13.14),
    "C3 VNet3d 4x96^3 in the f32 run dtype (the parity-exact mode: masks identical to the fp32 reference)": ("vnet", 3, (4, 1, 96, 96, 96), 1, "BinaryDiceLoss", "f32", 0.0, 22.5),
}


def other_configs(dev, steps=10):
    """ms per train step of the other BASELINE configs on this GPU (10 timed steps after 3 warm-up steps each, inputs resident): driver-visible
    numbers for configs[1], [3], [4] and for the f32 run dtype of the headline workload."""
    from pytorchdeeplearing_amd import SegEngine, synthetic
    out = {}
    for tag, (kind, ndim, shape, ncls, loss, dt, cld, gb) in OTHER_CONFIGS.items():
        try:
            e = SegEngine(kind, ndim, shape[1], ncls, dtype=dt, device=dev)
            synthetic.init_engine(e, seed=0)
            x, y = synthetic.synthetic_batch(shape[0], shape[2:], shape[1], ncls, seed=1)
            x, y = x.to(dev), y.to(dev)
            kw = {"class_alpha": torch.ones(ncls, device=dev)}
            if cld:
                kw["cldice_weight"] = cld
            for _ in range(6):
                e.train_step(x, y, loss, **kw)
                torch.cuda.synchronize()          # (the caching allocator settles step by step on the clDice path, which still allocates per step)
            t0 = time.perf_counter()
            for _ in range(steps):
                out3 = e.train_step(x, y, loss, **kw)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            out[tag] = {"ms_per_step": round(ms, 3), "samples_per_s": round(shape[0] / ms * 1e3, 1), "steps": steps, "final_loss": round(float(out3[0]), 5),
                        "fused_bound_GB_per_step": gb, "hbm_frac_of_fused_bound": round(gb / ms / PEAK_HBM_GBS * 1e3, 4)}
            del e, x, y
            torch.cuda.empty_cache()
        except Exception as ex:          # a side measurement must not take the contract line down
            out[tag] = {"error": str(ex)[:200]}
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--condition-seconds", type=float, default=0.75, help="un-timed conditioning loop BEFORE the warm-up the caller asks for: "
                    "train steps until this much wall time has passed (clocks / power state / allocator / lazy code-object loads settle); "
                    "reported in the JSON line as conditioning_steps")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the cpu_baseline / dice_vs_ref / gpu_torch_baseline legs")
    ap.add_argument("--roofline-steps", type=int, default=5, help="instrumented steps run AFTER the timed region: EVERY launch of the step (all kernel "
                    "families, both streams) is bracketed by HIP events on its launch stream (each bracket idles the stream for ~5 us, so none of "
                    "them is inside the timed region); the family with the most GPU time is \"roofline\", the next two \"roofline_2\" / \"roofline_3\"")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs leg (the other BASELINE configs + the f32 run dtype, 10 steps each)")
    ap.add_argument("--launch", default="auto", choices=["auto", "stream", "graph"], help="how the single-rank step reaches the GPU: stream = ~250 "
                    "launches enqueued per step by one library call; graph = the step captured once as a HIP graph, one hipGraphLaunch per step; "
                    "auto = both are timed for a few steps inside the un-timed conditioning phase and the faster one runs the warm-up and the "
                    "timed steps (a host that cannot keep up with the GPU is the case for the graph: BENCH_r02 lost 17 %% that way)")
    ap.add_argument("--single-allreduce", action="store_true", help="one blocking all-reduce after backward instead of two overlapped buckets")
    ap.add_argument("--global-loss", action="store_true", help="exact global-batch Dice across ranks (parallel.GlobalBatchLoss: 32 fp64 sums "
                    "all-reduced between the loss reduction and its finalize; gradients summed) instead of DDP semantics")
    return ap.parse_args(argv)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown CPU"


def _quiet_stdout(fn, *args, **kw):
    """Run fn with file descriptors 1 AND 2 pointed at a scratch file: MIOpen / composable_kernel print ~1 MB of solver diagnostics
    ("dimension check failure" ...) with C-level printf while stock PyTorch searches its convolution kernels; stdout of this script
    carries exactly ONE JSON line and the driver's stderr tail should show this script's own messages.  The scratch file's tail is
    replayed on stderr only if fn raises."""
    import ctypes
    import tempfile
    libc = ctypes.CDLL(None)
    sys.stdout.flush(); sys.stderr.flush()
    libc.fflush(None)
    saved1, saved2 = os.dup(1), os.dup(2)
    scratch = tempfile.TemporaryFile()
    try:
        os.dup2(scratch.fileno(), 1)
        os.dup2(scratch.fileno(), 2)
        return fn(*args, **kw)
    except BaseException:
        sys.stdout.flush(); sys.stderr.flush(); libc.fflush(None)
        os.dup2(saved2, 2)
        scratch.seek(max(0, scratch.tell() - 4096))
        sys.stderr.write(scratch.read().decode(errors="replace"))
        raise
    finally:
        sys.stdout.flush(); sys.stderr.flush()
        libc.fflush(None)          # the C library's own buffers (printf from MIOpen / CK) must drain while the descriptors still point at the scratch file
        os.dup2(saved1, 1)
        os.dup2(saved2, 2)
        os.close(saved1); os.close(saved2)
        scratch.close()


def _gpu_torch_baseline(seg, batch, size, dev, steps=3):
    """Stock PyTorch-ROCm on the SAME GPU: the oracle's functional restatement of the reference
    path (F.conv3d / group_norm / dropout3d-style masks / AdamW as torch ops, NCDHW) on `dev`, fp32 and autocast-f16, the full
    batch x 1 x size^3 workload, dropout on.  MIOpen picks the convolution kernels; first calls include its search.
    Part of the baseline leg: called from cpu_baseline() with its oracle module."""
    out = {}
    x, y = seg.synthetic_batch(batch, (size,) * 3, 1, 1, seed=1234)
    x, y = x.to(dev), y.to(dev)
    for mode in ("fp32", "autocast_f16"):
        try:
            params = {k: v.to(dev) for k, v in seg.init_params("vnet", 3, 1, 1, seed=0).items()}
            g = torch.Generator().manual_seed(0)
            st = {}
            times = []
            for it in range(steps + 2):
                masks = [m.to(dev) for m in seg.draw_masks("vnet", batch, generator=g)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if mode == "fp32":
                    r = seg.forward_backward("vnet", params, x, y, "BinaryDiceLoss", masks=masks)
                else:
                    with torch.autocast("cuda", dtype=torch.float16):
                        r = seg.forward_backward("vnet", params, x, y, "BinaryDiceLoss", masks=masks)
                params = seg.adamw_step(params, r["grads"], st)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            best = min(times[2:])
            out[mode] = {"value": round(batch / best, 2), "unit": "volumes/s", "ms_per_step": round(best * 1e3, 2),
                         "sample": "%d timed train steps (2 warm-up) of VNet3d %dx1x%d^3, oracle functions on %s, torch %s"
                                   % (steps, batch, size, torch.cuda.get_device_name(dev), torch.__version__)}
            del params, st, r
            torch.cuda.empty_cache()
        except Exception as ex:      # a baseline that cannot run must not take the contract line down
            out[mode] = {"error": str(ex)[:200]}
    return out


def cpu_baseline(size, trained_state, dev, dtype, seconds_budget=25.0, batch=4, gpu_leg=False):
    """The oracle (torch-CPU port of the reference path, oracle/seg_oracle.py) on this box's host cores,
    bounded sample of the same workload: VNet3d `batch` x 1 x size^3 train steps (the benchmark's own 4-volume batch; fp32, dropout on;
    one warm-up + two timed steps, best of the timed ones - about 15 s).
    Returns (cpu_baseline, dice_vs_ref, gpu_torch_baseline or None): the only place of this file that touches oracle/."""
    from oracle import seg_oracle as seg
    # torch-CPU convolutions stop scaling (and then collapse) beyond a few dozen threads on a many-core host:
    # 32 threads is the fastest setting for this workload on the 256-core GPU box (all 256: 134 s/step)
    ncores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(ncores)
    params = seg.init_params("vnet", 3, 1, 1, seed=0)
    x, y = seg.synthetic_batch(batch, (size,) * 3, 1, 1, seed=1234)
    g = torch.Generator().manual_seed(0)
    st = {}
    times = []
    t_start = time.time()
    for it in range(3):
        masks = seg.draw_masks("vnet", batch, generator=g)
        t0 = time.time()
        r = seg.forward_backward("vnet", params, x, y, "BinaryDiceLoss", masks=masks)
        params = seg.adamw_step(params, r["grads"], st)
        times.append(time.time() - t0)
        if time.time() - t_start > seconds_budget:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    base = {"value": round(batch / best, 4), "unit": "volumes/s", "cores": ncores, "kind": "port",
            "sample": "%d train steps (the first is warm-up) of VNet3d %dx1x%d^3 fp32 - the benchmark's own batch -, torch %s CPU, %d of the %d hardware "
                      "threads of %s, best step %.3f s" % (len(times), batch, size, torch.__version__, ncores, os.cpu_count() or 1, cpu_model(), best)}
    dice = _dice_vs_reference(seg, trained_state, dev, dtype, size=size)
    return base, dice, (_quiet_stdout(_gpu_torch_baseline, seg, batch, size, dev) if gpu_leg else None)


def _dice_vs_reference(seg, trained_state, dev, dtype, size=96):
    """BASELINE metric, second half ("Dice vs ref"): one eval forward of a 1 x 1 x size^3 volume (the workload's own 96^3) through the engine (run dtype and
    f32) and through the oracle on the host, same weights; integer-mask Dice against the synthetic label must be identical
    when the masks are (model/metric.py:146-155).  Weights: the benchmark's random init with biases / GroupNorm affine perturbed
    (the weights after the timed steps predict all-foreground on random labels: a trivial mask).  Called from cpu_baseline only."""
    from pytorchdeeplearing_amd import SegEngine, synthetic
    init = synthetic.init_engine(SegEngine("vnet", 3, 1, 1, dtype="f32", device=dev), seed=0).state_dict()
    sd = seg.perturb_params({k: v.detach().float().cpu() for k, v in init.items()}, seed=7)
    x, y = seg.synthetic_batch(1, (size,) * 3, 1, 1, seed=4321)
    logits_ref, probs_ref = seg.net_forward("vnet", sd, x)
    mask_ref = probs_ref > 0.5
    dice_ref = float(seg.dice_coeff(probs_ref, y))
    out = {"case": "VNet3d eval forward, 1x1x%d^3, random-init weights (perturbed affine), vs the torch-CPU oracle (fp32)" % size,
           "dice_oracle": round(dice_ref, 7), "foreground_fraction_oracle": round(float(mask_ref.float().mean()), 4)}
    for dt in dict.fromkeys([dtype, "f32"]):
        e = SegEngine("vnet", 3, 1, 1, dtype=dt, device=dev)
        e.load_state_dict(sd)
        logits, probs = e.forward(x.to(dev))
        out3 = e.loss_forward(logits, y.to(dev), "BinaryDiceLoss").cpu()
        mism = int(((probs.cpu() > 0.5) != mask_ref).sum())
        out[dt] = {"dice_engine": round(float(out3[1]), 7), "mask_mismatch_voxels": mism, "voxels": int(mask_ref.numel()),
                   "logits_max_abs_diff": float((logits.cpu() - logits_ref).abs().max())}
        del e
    return out



def main(argv=None, checker_device=None):
    """checker_device: TEST-ONLY (tests/test_parallel.py): run the control flow of this file on the host-side kernel checker
    with the gloo backend, so the N > 1 path (barriers, bucketed exchange, MAX over ranks, the JSON line) is exercised on
    the GPU-less build box.  None = the real thing: one process per GPU, RCCL."""
    a = parse(argv)
    # stdout carries exactly ONE JSON line: file descriptor 1 points at stderr for the whole run (RCCL prints a version banner, MIOpen / CK
    # print solver diagnostics, all with C-level stdio) and comes back on EVERY rank and exit path (finally) before the line is printed
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)
    line = None
    try:
        line = _run(a, checker_device)
    finally:
        sys.stdout.flush()
        libc.fflush(None)
        os.dup2(saved_stdout_fd, 1)
        os.close(saved_stdout_fd)
    if line is not None:
        print(json.dumps(line))
        sys.stdout.flush()


def _run(a, checker_device):
    """everything between the stdout redirect and the JSON line; returns the line on rank 0, None elsewhere"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if a.gpus != world:
        # the world size comes from the launcher (one process per GPU): a plain `python bench.py --gpus 8` would silently measure ONE GPU
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d - launch the N-GPU run as `python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py --gpus %d ...`" % (a.gpus, world, a.gpus, a.gpus))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    on_gpu = checker_device is None
    if on_gpu:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device(checker_device)
    gpu_sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    if world > 1:
        import torch.distributed as dist
        if on_gpu:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    try:
        return _bench(a, dev, on_gpu, gpu_sync, dist, world, rank)
    finally:
        if dist:
            dist.destroy_process_group()


def _bench(a, dev, on_gpu, gpu_sync, dist, world, rank):
    from pytorchdeeplearing_amd import SegEngine, synthetic   # oracle/ is touched by the cpu_baseline leg only
    from pytorchdeeplearing_amd.parallel import BucketedGradAllReduce, GlobalBatchLoss, GradAllReduce

    S = a.size
    # the other BASELINE configs first, before anything of the headline run exists in the process: measured after the HIP-graph probe of `--launch auto`
    # the clDice config (the one path that uses a second torch stream per step) ran 11 ms instead of 6.4 (profiles/r04_bench_other_configs_order.txt)
    others = other_configs(dev) if (on_gpu and world == 1 and rank == 0 and not a.no_other_configs) else None
    e = SegEngine("vnet", 3, 1, 1, dtype=a.dtype, device=dev)
    synthetic.init_engine(e, seed=0)
    if world > 1:
        from pytorchdeeplearing_amd.parallel import broadcast_parameters
        broadcast_parameters(e, src=0)          # every replica starts from rank 0's weights (the dropout streams differ per rank)
    x, y = synthetic.synthetic_batch(a.batch, (S, S, S), 1, 1, seed=1234 + rank)
    x, y = x.to(dev), y.to(dev)
    logits = torch.empty((a.batch, 1, S, S, S), dtype=torch.float32, device=dev)
    probs = torch.empty_like(logits)
    allreduce = (GradAllReduce(world) if a.single_allreduce else BucketedGradAllReduce(world)) if world > 1 else None

    exchange = GlobalBatchLoss(world, equal_shards=True) if (a.global_loss and world > 1) else None
    kw = {"loss_exchange": exchange} if exchange is not None else {}

    can_graph = on_gpu and world == 1
    launch = {"mode": a.launch if (can_graph and a.launch != "auto") else "stream"}
    kw["launch"] = "stream"

    def step():
        kw["launch"] = launch["mode"]
        return e.train_step(x, y, "BinaryDiceLoss", lr=1e-3, allreduce=allreduce, logits=logits, probs=probs, **kw)

    probe = None
    if can_graph and a.launch == "auto":
        # un-timed: 8 steps to settle, then the same number of steps through each launch path; the faster one is used from here on
        def timed_steps(mode, n):
            launch["mode"] = mode
            for _ in range(3):
                step()
            gpu_sync()
            t = time.perf_counter()
            for _ in range(n):
                step()
            gpu_sync()
            return (time.perf_counter() - t) / n * 1e3
        for _ in range(8):
            step()
        probe = {"stream": round(timed_steps("stream", 24), 3)}
        try:
            probe["graph"] = round(timed_steps("graph", 24), 3)
            if getattr(e, "_graph_key", None) is None:
                probe["graph"] = None                     # the capture was refused: the steps above ran through the stream path
                probe["graph_error"] = getattr(e, "graph_error", None)
        except RuntimeError as ex:
            probe["graph"], probe["graph_error"] = None, str(ex)[:160]
        launch["mode"] = "graph" if (probe["graph"] is not None and probe["graph"] < 0.985 * probe["stream"]) else "stream"

    # ---- conditioning (un-timed, before the warm-up the caller asks for): the first tens of milliseconds after a cold start do not run
    # at steady-state speed (round 2: 727 volumes/s at --steps 20 --warmup 5 on the driver's box against 880 at 50/10 on the builder's;
    # 5 warm-up steps are 25 ms).  Every rank runs the SAME number of steps (rank 0 decides), so collectives stay matched.
    ncond = 0
    if a.condition_seconds > 0:
        gpu_sync()
        tc = time.perf_counter()
        chunk = 8 if on_gpu else 1
        while True:
            for _ in range(chunk):
                out3 = step()
            ncond += chunk
            gpu_sync()
            more = 1 if (time.perf_counter() - tc < a.condition_seconds and ncond < 4096) else 0
            if dist:
                flag = torch.tensor([more], device=dev, dtype=torch.int32)
                dist.broadcast(flag, src=0)
                more = int(flag)
            if not more:
                break
    for _ in range(a.warmup):
        out3 = step()
    gpu_sync()
    if dist:
        dist.barrier()
    gpu_sync()
    # ---- the timed region: exactly K steps, no instrumentation of any kind inside
    t0 = time.perf_counter()
    nhost, t_enqueued = min(8, a.steps), 0.0
    for i in range(a.steps):
        out3 = step()
        if i == nhost - 1:
            t_enqueued = time.perf_counter() - t0      # host time to enqueue the first steps: it runs ahead of the GPU (later the full queue throttles it)
    gpu_sync()
    if dist:
        dist.barrier()
    gpu_sync()
    dt = time.perf_counter() - t0
    loss = float(out3[0])
    # ---- roofline: R instrumented steps AFTER the timed region.  EVERY launch of the step (all profile classes, main stream and weight-gradient
    # stream) is bracketed with hipEventRecord on its launch stream; the classes are grouped into kernel families (FAMILIES) and the family with
    # the most GPU time is "roofline" - the choice is made from the complete table, not from a short list.
    from pytorchdeeplearing_amd import _capi
    nprof = max(0, a.roofline_steps)
    prof = {}
    if nprof:
        e.profile_enable(_capi.KERNEL_CLASSES)
        if on_gpu:
            step()                            # creates the event pool (hipEventCreate) outside the measured brackets
            gpu_sync()
            e.profile_read()
        for _ in range(nprof):
            step()
        gpu_sync()
        prof = e.profile_read()
        e.profile_enable([])
        if dist:
            dist.barrier()
    # an EMPTY bracket (two event records back to back on the launch stream) is not zero: each record is a barrier packet
    # with a timestamp.  Measured live and subtracted per launch below, so that the event-based average can be compared
    # with rocprofv3's kernel durations (which carry no brackets); both the raw and the corrected figures are reported.
    bracket_us = 0.0
    if on_gpu:
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for ea, eb in pairs:
            ea.record(); eb.record()
        gpu_sync()
        bracket_us = sorted(ea.elapsed_time(eb) for ea, eb in pairs)[len(pairs) // 2] * 1e3
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms = dt / a.steps * 1e3
    vols = world * a.batch * a.steps / dt

    if rank == 0:
        scale = (S / 96.0) ** 3
        line = {
            "metric": "volumes/sec VNet3d 96^3 fp16 train step", "value": round(vols, 2), "unit": "volumes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "VNet3d(1,1) binary seg, %dx1x%d^3 per GPU, BinaryDiceLoss + Dice metric, AdamW, dropout p=0.2 on, "
                                   "random-init weights (BASELINE.json configs[2])" % (a.batch, S),
                       "global_batch": a.batch * world, "parallelism": "dp%d" % world,
                       "loss_semantics": "global-batch (sums exchanged)" if exchange is not None else "per-rank (DDP)"},
            "final_loss": round(loss, 5),
            "conditioning_steps": ncond, "conditioning_seconds": a.condition_seconds,
            "launch_mode": launch["mode"], "launch_probe_ms_per_step": probe,
            "host_enqueue_ms_per_step": round(t_enqueued / nhost * 1e3, 3),
            "whole_step": {"hbm_frac_of_fused_bound": round(GB_PER_VOLUME_96 * scale * vols / world / PEAK_HBM_GBS, 4),
                           "mfma_frac": round(GFLOP_PER_VOLUME_96 * scale * vols / world / 1e3 / PEAK_MFMA_TFLOPS, 4)},
        }
        line["timed_region_s"] = round(dt, 4)
        line["build"] = _library_build() if on_gpu else None      # seg_build_info() of the loaded library (a sha of its sources): which binary this line is from
        # ---- kernel families: the profile classes grouped, event time corrected by the empty-bracket time per launch
        fams = {}
        for name, f in FAMILIES.items():
            calls = sum(prof.get(c, {}).get("calls", 0) for c in f["classes"])
            raw_ms = sum(prof.get(c, {}).get("ms", 0.0) for c in f["classes"])
            if not calls or raw_ms <= 0:
                continue
            nbytes = sum(prof[c]["bytes"] for c in f["classes"] if c in prof)
            flops = sum(prof[c]["flops"] for c in f["classes"] if c in prof)
            ms_c = max(raw_ms - calls * bracket_us * 1e-3, 0.25 * raw_ms)
            fams[name] = dict(calls=calls, raw_ms=raw_ms, ms=ms_c, bytes=nbytes, flops=flops)
        nprof = max(nprof, 1)
        order = sorted(fams, key=lambda k: -fams[k]["ms"])
        total_ms = sum(v["ms"] for v in fams.values()) or 1.0

        def roofline_block(name, rank_):
            f, p = FAMILIES[name], fams[name]
            mfma = f["bound"] == "mfma" and p["flops"] > 0
            if mfma:
                ach, ach_raw, peak, unit = p["flops"] / (p["ms"] * 1e-3) / 1e12, p["flops"] / (p["raw_ms"] * 1e-3) / 1e12, PEAK_MFMA_TFLOPS, "TFLOP/s"
            else:
                ach, ach_raw, peak, unit = p["bytes"] / (p["ms"] * 1e-3) / 1e9, p["bytes"] / (p["raw_ms"] * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
            traffic, tfile = pmc_traffic(name) if (a.dtype == "f16" and S == 96 and a.batch == 4) else (None, None)
            blk = {"kernel": "%s: %s" % (name, f["what"]), "queue": f["queue"], "bound": "mfma" if mfma else "hbm",
                   "achieved": round(ach_raw, 1), "peak": peak, "unit": unit, "frac": round(ach_raw / peak, 4),
                   "traffic": traffic, "traffic_source": tfile,
                   "launches_per_step": p["calls"] // nprof, "avg_launch_us": round(p["raw_ms"] / p["calls"] * 1e3, 2),
                   "ms_per_step": round(p["raw_ms"] / nprof, 3), "share_of_gpu_time": round(p["ms"] / total_ms, 4),
                   "avg_launch_us_minus_bracket": round(p["ms"] / p["calls"] * 1e3, 2), "bracket_overhead_us": round(bracket_us, 2),
                   "achieved_minus_bracket": round(ach, 1), "frac_minus_bracket": round(ach / peak, 4), "instrumented_steps": nprof,
                   "algorithmic_bytes_per_launch": int(p["bytes"] / p["calls"])}
            if p["flops"] > 0:
                blk["algorithmic_flops_per_launch"] = int(p["flops"] / p["calls"])
                if mfma:      # the same family against the other roofline (the 16-channel finest-level launches of an MFMA family are HBM-bound)
                    blk["hbm_GBs_minus_bracket"] = round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1)
            blk["note"] = ("rank %d of %d kernel families by GPU time (all launches of the step, both streams, bracketed with hipEventRecord on their launch stream in %d "
                           "instrumented steps AFTER the timed region; the brackets serialise the two streams less than the profiler-free step, so per-launch "
                           "times are those of the instrumented steps); achieved = sum of algorithmic %s over the family's launches / sum of event time; "
                           "*_minus_bracket subtracts the empty-bracket time measured live per launch; algorithmic work per launch: DESIGN.md section 5"
                           % (rank_, len(fams), nprof, "flops (2*voxels*taps*Cin*Cout)" if mfma else "bytes (operands read once + results written once)"))
            return blk

        for i, name in enumerate(order[:3]):
            line["roofline" if i == 0 else "roofline_%d" % (i + 1)] = roofline_block(name, i + 1)
        line["kernel_families"] = {k: {"queue": FAMILIES[k]["queue"], "launches_per_step": fams[k]["calls"] // nprof, "ms_per_step": round(fams[k]["ms"] / nprof, 3),
                                       "frac": round((fams[k]["flops"] / (fams[k]["ms"] * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS) if (FAMILIES[k]["bound"] == "mfma" and fams[k]["flops"] > 0)
                                                     else (fams[k]["bytes"] / (fams[k]["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS), 4),
                                       "bound": FAMILIES[k]["bound"]} for k in order}
        if others is not None:
            line["other_configs"] = others
        if not a.no_cpu_baseline and world == 1:
            sd = None                      # (dice_vs_ref builds its own perturbed weights: the trained ones predict a trivial mask)
            del e
            if on_gpu:
                torch.cuda.empty_cache()
            line["cpu_baseline"], line["dice_vs_ref"], gt = cpu_baseline(S, sd, dev, a.dtype, batch=a.batch, gpu_leg=on_gpu)
            if gt is not None:
                line["gpu_torch_baseline"] = gt
        return line
    return None


if __name__ == "__main__":
    main()
