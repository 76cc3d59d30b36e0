#!/usr/bin/env python3
"""Headline benchmark: images/sec of the VQ-W-Net first training step on synthetic 256x256 slices.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N>1 with torch.distributed.run); per-GPU batch is fixed
(32 source images = 64 views per step) so scaling is weak.  A step = 2 views x (encoder + VQ + decoder)
forward, embedding / reconstruction losses, backward and two Adam updates (the reference's
trainers/single_window_trainer.py:68-147), all in the HIP kernels behind libvqwnet_hip.so.
Rank 0 prints ONE JSON line.  Extra fields: `roofline` for the dominant kernel family (timed with HIP
events on the launch stream inside the timed region) and `cpu_baseline` (the CPU oracle timed on the
host cores, rank 0, N=1 only, bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "medical-image-editing_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic work per source image at 256x256 with the R-cfg model
FLOP_PER_IMAGE = 490.3e9
BYTES_PER_IMAGE = 3.28e9
PEAK_FP32_MFMA = 157.3e12        # MI355X_MICROARCH.md: fp32-input MFMA = vector fp32 peak
PEAK_HBM = 8.0e12


def synthetic_batch(batch, size, seed, device):
    """smooth random field + noise in [-1, 1]; noise for the second view's noised copy."""
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(batch, 1, size // 8, size // 8, generator=g)
    field = torch.nn.functional.interpolate(low, size=(size, size), mode="bilinear", align_corners=False)
    img = (field * 0.6 + 0.05 * torch.randn(batch, 1, size, size, generator=g)).clamp_(-1, 1)
    noise = 0.02 * torch.randn(batch, 1, size, size, generator=g)
    return img.to(device), noise.to(device)


def csrc_digest():
    """sha1 over the convolution kernel sources and the host code that picks a kernel form per layer: ties a committed PMC
    traffic file to the kernels and the dispatch it was measured on."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "medical-image-editing_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith("conv") and f.endswith((".hip", ".h")) or f == "mfma_util.h":
            h.update(open(os.path.join(d, f), "rb").read())
    for f in ("hipops/ops.py", "networks/unet_decoder.py", "networks/unet_encoder.py", "networks/blocks.py"):
        h.update(open(os.path.join(ROOT, "medical-image-editing_amd", f), "rb").read())
    return h.hexdigest()


def host_cores():
    """Usable host cores: min(affinity, cgroup CPU quota), capped at 16 (the CPU share of a 1-GPU box)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(size):
    """The oracle's training step on the host cores, a bounded sample of the same workload:
    batch 2 at `size`: one warm-up + 3 timed steps on all usable cores (`value`), 2 more timed steps on 8 threads
    (comparable with the reference's own 0.31 images/s measured on 8 cores in the build container); and BASELINE
    config 1 (32x32, batch 4), 3 timed steps."""
    from oracle import vqwnet_ref as O
    from networks import UNetEncoder, UNetDecoder
    cores = host_cores()

    def run(sz, batch, threads, steps, warm):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        enc = UNetEncoder(1, [16, 32, 64, 128, 256], 10, 0.999, 'torch', False, 1, True)
        dec = UNetDecoder(16, 1, [32, 64, 128, 256, 512], use_dropblock=False, dropped_skip_layers=[], use_pixel_shuffle=False)
        PE = {k: v.detach().clone().contiguous() for k, v in enc.state_dict().items()}
        PD = {k: v.detach().clone().contiguous() for k, v in dec.state_dict().items()}
        cfg = dict(dict_size=10, margin=0.5, border=0, momentum=0.999,
                   weights=dict(commit=1.0, cross=1.0, dist=1.0, reg=1.0, recon=1.0),
                   optim=dict(lr=1e-4, betas=(0.5, 0.999), weight_decay=0.0))
        tr = O.FirstStepTrainer(PE, PD, cfg)
        for s in range(warm):
            tr.step(*O.synthetic_slices(batch, sz, 1234))   # allocator, oneDNN primitives
        t0 = time.perf_counter()
        for s in range(steps):
            tr.step(*O.synthetic_slices(batch, sz, 1235 + s))
            print("[bench] cpu_baseline %dx%d batch %d, %d threads, step %d: %.1f s elapsed"
                  % (sz, sz, batch, threads, s, time.perf_counter() - t0), file=sys.stderr, flush=True)
        return batch * steps / (time.perf_counter() - t0)
    v_all = run(size, 2, cores, 3, 1)
    v_8 = run(size, 2, min(8, cores), 2, 0) if cores > 8 else v_all
    v_c1 = run(32, 4, cores, 3, 1)
    return dict(value=v_all, unit="images/sec", cores=cores, kind="port", value_8_threads=v_8, config1_32x32_batch4_images_per_sec=v_c1,
                sample="the oracle's first training step (oracle/vqwnet_ref.py, torch %s CPU, fp32): batch 2 at %dx%d, 3 timed steps after "
                       "1 warm-up on %d threads (value), 2 timed steps on 8 threads (value_8_threads); BASELINE config 1 (32x32, "
                       "batch 4), 3 timed steps on %d threads" % (torch.__version__, size, size, cores, cores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="source images per GPU per step (default: the config's batch_size)")
    ap.add_argument("--size", type=int, default=None, help="slice size (default: the config's image_size)")
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "baseline2_256x256_b32_1gpu.json"),
                    help="reference-style JSON config (configs/); --batch / --size override its dataset section")
    ap.add_argument("--serial-steps", type=int, default=5, help="steps of the serialised per-kernel timing pass")
    ap.add_argument("--dp-overlap", type=int, choices=(0, 1), default=None,
                    help="data-parallel gradient exchange: 1 = buckets all-reduced from inside the backward pass (overlapped with the "
                         "remaining backward kernels), 0 = after the backward pass; default: VQW_DP_OVERLAP or 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    if args.dp_overlap is not None:
        os.environ["VQW_DP_OVERLAP"] = str(args.dp_overlap)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # rehearsal switch (one-GPU boxes): VQW_BENCH_BACKEND=gloo puts every rank on cuda:0 and uses gloo instead of RCCL
    backend = os.environ.get("VQW_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # VQW_DP_FORCE=1 on one GPU: a process group of one rank over RCCL with every data-parallel collective issued (they are
    # identities): what the RCCL code path costs next to the kernels, measurable without a second GPU
    forced = world == 1 and os.environ.get("VQW_DP_FORCE", "0") == "1"
    if forced:
        os.environ.setdefault("MASTER_PORT", "29655")
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # tensors handed to a collective are kept alive by the work object instead of being recorded on RCCL's stream: a
        # recorded block cannot be reused by the caching allocator until that stream has passed it (pool growth = hipMalloc
        # stalls inside the step)
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from hipops import _lib
    from trainers import build_first_step_trainer
    from utils import load_json
    L = _lib.load()

    cfg = load_json(args.config)                           # the reference's config keys (trainers/base.py:164-278)
    if args.batch is None:
        args.batch = int(cfg.dataset.batch_size)
    if args.size is None:
        args.size = int(cfg.dataset.image_size)
    g = cfg.model.vqmodel
    torch.manual_seed(0)                                   # identical replicas on every rank
    # (VQW_DP_FORCE_GRADS=0 with VQW_DP_FORCE=1: only the statistics collectives are forced - a measurement aid)
    tr = build_first_step_trainer(cfg, device=dev, data_parallel=world > 1 or (forced and os.environ.get("VQW_DP_FORCE_GRADS", "1") != "0"))
    pool = [synthetic_batch(args.batch, args.size, 1234 + 1000 * rank + s, dev) for s in range(4)]

    # the dependency chain of the step runs on a high-priority stream, the off-chain weight gradients on the (normal
    # priority) side stream of hipops: the hardware dispatches chain kernels first and fills the gaps with wgrad
    prio = os.environ.get("VQW_BENCH_PRIORITY", "1") != "0"
    chain = torch.cuda.Stream(device=dev, priority=-1) if prio else torch.cuda.current_stream()

    def step(i):
        img, noise = pool[i % len(pool)]
        with torch.cuda.stream(chain):
            return tr.training_step({"image": img}, noise=noise)

    for i in range(args.warmup):
        step(i)                 # not synchronised one by one: the warm-up also fills the allocator's pool for the number
        if rank == 0:           # of steps the trainer keeps in flight
            print("[bench] warm-up step %d enqueued" % i, file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timing = not args.no_kernel_timing
    if timing:
        L.vqw_profile_families(0x1F)         # timed region: the five conv families only (see include/vqwnet_hip.h)
        _lib.check(L.vqw_profile_begin(), "vqw_profile_begin")
    from hipops import ops as _ops
    coll0 = (_ops.collective_calls, tr.reducer.launches if tr.reducer is not None else 0)
    ms0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    coll1 = (_ops.collective_calls, tr.reducer.launches if tr.reducer is not None else 0)
    if rank == 0:
        ms1 = torch.cuda.memory_stats(dev)
        print("[bench] %d timed steps: %.1f ms/step; device segments allocated inside the timed region: %d (+%.2f GB reserved)"
              % (args.steps, dt / args.steps * 1e3, ms1["segment.all.allocated"] - ms0["segment.all.allocated"],
                 (ms1["reserved_bytes.all.current"] - ms0["reserved_bytes.all.current"]) / 2**30), file=sys.stderr, flush=True)
    prof = (ctypes.c_double * 24)()
    if timing:
        _lib.check(L.vqw_profile_end(prof), "vqw_profile_end")
    # Second, untimed look at the same kernels WITHOUT concurrency: in the timed region the weight-gradient kernels run
    # on a side stream next to the chain kernels, which stretches every kernel's own duration.  Two extra steps with
    # the side stream off give the kernels' exclusive durations (reported as roofline.exclusive).
    prof_x = (ctypes.c_double * 24)()
    if timing and _ops.WGRAD_ASYNC:
        _ops.WGRAD_ASYNC = False
        cv, tr.concurrent_views = tr.concurrent_views, False
        step(0)
        torch.cuda.synchronize()
        L.vqw_profile_families(0x3F)         # serialised pass: the HBM-bound norm / element-wise family as well
        _lib.check(L.vqw_profile_begin(), "vqw_profile_begin")
        for i in range(args.serial_steps):
            step(1 + i)
        torch.cuda.synchronize()
        _lib.check(L.vqw_profile_end(prof_x), "vqw_profile_end")
        _ops.WGRAD_ASYNC = True
        tr.concurrent_views = cv
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total = float(out["total"].detach())
    assert total == total, "loss is NaN"

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        imgs = args.batch * world * args.steps / dt
        # (the Winograd family's FLOPs are the 4/9 of the direct form's that the matrix cores execute: its TFLOP/s is
        # hardware utilisation like the others'; x 2.25 = the rate in direct-form FLOPs)
        fam = ["conv_mfma_fwd_dgrad", "conv_mfma_wgrad", "conv_generic_fwd", "conv_generic_wgrad", "conv_winograd"]
        HBM_FAMILY = 5      # normalisation / element-wise entry points: bytes as launched, no FLOPs

        def families(pr, nsteps):
            out = {}
            for f, name in enumerate(fam):
                n, ms, fl, by = pr[4 * f], pr[4 * f + 1], pr[4 * f + 2], pr[4 * f + 3]
                if n > 0:
                    out[name] = dict(launches_per_step=n / nsteps, ms_per_step=ms / nsteps,
                                     tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                                     avg_launch_ms=ms / n, gflop_per_launch=fl / n / 1e9, bytes_per_launch=by / n)
            return out
        kern_c = families(prof, args.steps)          # timed region: three streams share the GPU
        kern_x = families(prof_x, args.serial_steps)  # serialised pass: each kernel alone on the GPU
        roofline = None
        primary = kern_x if kern_x else kern_c
        if primary:
            dom = max(primary, key=lambda k: primary[k]["ms_per_step"])
            ach = primary[dom]["tflops"]
            # HBM bytes per launch of that family: NOT measured by this run (PMC counters need rocprofv3 around the
            # process) but read from the committed PMC passes of this same command (tools/pmc_traffic.py); the file
            # names the digest of the kernel sources it was taken with, and a stale or missing file gives null
            traffic, traffic_source = None, "none: profiles/r04_hbm_traffic.json missing"
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r04_hbm_traffic.json")))
                if tj.get("csrc_digest") == csrc_digest():
                    # per API launch of the family (a call may be several kernel launches): bytes per step / calls per step
                    traffic = tj["families"][dom]["hbm_bytes_per_step"] / primary[dom]["launches_per_step"]
                    traffic_source = "profiles/r04_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, kernel sources %s)" % tj["csrc_digest"][:12]
                else:
                    traffic_source = "none: profiles/r04_hbm_traffic.json was taken with other kernel sources"
            except Exception:
                pass
            roofline = dict(bound="mfma", kernel=dom, achieved=ach, peak=PEAK_FP32_MFMA / 1e12, unit="TFLOP/s",
                            frac=ach / (PEAK_FP32_MFMA / 1e12), traffic=traffic, traffic_source=traffic_source,
                            algorithmic_bytes_per_launch=primary[dom]["bytes_per_launch"],
                            avg_launch_ms=primary[dom]["avg_launch_ms"],
                            measured=("HIP events on the launch stream around every launch of the family, over the steps run "
                                      "right after the timed region with the streams serialised (VQW_WGRAD_STREAM=0, "
                                      "VQW_CONCURRENT_VIEWS=0 equivalent): each kernel alone on the GPU"
                                      if kern_x else "HIP events over the timed region"),
                            kernels=primary)
            if dom == "conv_winograd":
                # this family's FLOPs are the 4/9 of the direct form's that its kernels execute: `achieved` / `frac` are
                # hardware utilisation; priced at the direct form's (the operator's algorithmic) FLOPs the rate is 2.25x
                roofline["algorithmic_direct_form"] = dict(
                    achieved=ach * 2.25, ratio_to_peak=ach * 2.25 / (PEAK_FP32_MFMA / 1e12),
                    note="the same launches priced at the operator's direct-form FLOPs (x 2.25): credit for work the Winograd "
                         "form avoids, NOT utilisation - the ratio can exceed 1")
                roofline["note"] = "Winograd F(2x2,3x3) kernels: achieved / frac count the FLOPs executed (4/9 of the direct form's)"
            if kern_x and dom in kern_c:
                roofline["timed_region_concurrent"] = dict(
                    achieved=kern_c[dom]["tflops"], frac=kern_c[dom]["tflops"] / (PEAK_FP32_MFMA / 1e12),
                    avg_launch_ms=kern_c[dom]["avg_launch_ms"], kernels=kern_c,
                    note="same events inside the timed region, where the two views' chains and the weight-gradient "
                         "stream run concurrently: a kernel's own duration then includes time its waves spend "
                         "sharing CUs with other kernels, so it under-states kernel quality; throughput (`value`) "
                         "is what the concurrency buys")
        # the HBM-bound block of the step (InstanceNorm / SPADE forward and backward, ResBlock tails, ReLU backward, adds,
        # pooling, Adam): the same HIP events, serialised pass; achieved = tensor passes as launched x 4 B / time
        roofline_hbm = None
        pr = prof_x if any(prof_x) else prof
        nst = args.serial_steps if any(prof_x) else args.steps
        if pr[4 * HBM_FAMILY] > 0:
            n, ms, by = pr[4 * HBM_FAMILY], pr[4 * HBM_FAMILY + 1], pr[4 * HBM_FAMILY + 3]
            roofline_hbm = dict(bound="hbm", family="norm_elementwise", launches_per_step=n / nst, ms_per_step=ms / nst,
                                achieved=by / (ms * 1e-3) / 1e9, peak=PEAK_HBM / 1e9, unit="GB/s", frac=by / (ms * 1e-3) / PEAK_HBM,
                                bytes_per_step=by / nst,
                                note="HIP events around the normalisation / element-wise entry points (vqw_inorm_*, vqw_spade_*, "
                                     "vqw_res_tail_*, vqw_relu_bwd, vqw_add, vqw_maxpool2_*, vqw_adam_step); bytes = tensor passes "
                                     "as launched; autograd's own at::add / fill kernels are outside these events")
        per_gpu = imgs / world
        scale = (args.size / 256.0) ** 2
        # FLOPs the kernels actually executed per step (collapsed up-sampled and Winograd-form convs at 4/9 of the reference's count), from the
        # same launch records: hardware utilisation; the algorithmic figure prices the step at the reference's conv FLOPs
        executed = sum(prof[4 * f + 2] for f in range(5)) / args.steps if timing else None
        rcfg = list(g.enc_filters) == [16, 32, 64, 128, 256] and list(g.dec_filters) == [32, 64, 128, 256, 512]
        line = {
            "metric": "images/sec (train step, 256x256 2D slices)", "value": imgs, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "VQ-W-Net first training step (2 views: encoder+VQ+decoder fwd/bwd + 2x Adam), "
                                   "filters enc %s dec %s, dict_size %d, %dx%d 1-ch synthetic slices (%s)"
                                   % (list(g.enc_filters), list(g.dec_filters), g.dict_size, args.size, args.size, os.path.basename(args.config)),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": "dp%d" % world},
            # whole step against the two rooflines.  "algorithmic" = the reference's conv FLOPs / bytes, i.e.
            # credit for work avoided (collapsed up-sampled and Winograd-form layers); "executed" = FLOPs the kernels ran = hardware utilisation
            # (the un-suffixed key is the EXECUTED fraction = utilisation; the algorithmic one prices Winograd-form and collapsed
            # layers at operator FLOPs their kernels legitimately do not execute (4/9, 1/4) and can therefore exceed 1)
            "step_fraction_of_fp32_mfma_roofline": executed / (ms_per_step * 1e-3) / PEAK_FP32_MFMA if executed else None,
            "step_fraction_of_fp32_mfma_roofline_executed": executed / (ms_per_step * 1e-3) / PEAK_FP32_MFMA if executed else None,
            "step_fraction_of_fp32_mfma_roofline_algorithmic": per_gpu * FLOP_PER_IMAGE * scale / PEAK_FP32_MFMA if rcfg else None,
            "step_fraction_of_fp32_mfma_roofline_algorithmic_note": "operator FLOPs of the reference's convolutions / peak: credit for "
                                                                    "work avoided, not utilisation; can exceed 1",
            "step_fraction_of_hbm_roofline": per_gpu * BYTES_PER_IMAGE * scale / PEAK_HBM if rcfg else None,
            "loss_total": total,
            # ranks of the RCCL process group the timed steps ran in (0: no process group - a plain single-GPU run), and the
            # collectives one rank issued per step: SyncBN / VQ statistics all-reduces, gradient-bucket all-reduces
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() and dist.get_backend() == "nccl" else 0,
            "collectives_per_step": {"statistics": (coll1[0] - coll0[0]) / args.steps, "gradient_buckets": (coll1[1] - coll0[1]) / args.steps},
            # how the gradient buckets are exchanged when there is more than one rank ("overlap": launched from inside the
            # backward pass as each bucket completes; "after_backward": in finish(), after the pass is enqueued); --dp-overlap
            "dp": (tr.reducer.describe() if tr.reducer is not None else
                   {"dp_schedule": "none (single rank, no reducer)", "would_be": "overlap" if os.environ.get("VQW_DP_OVERLAP", "0") != "0" else "after_backward"}),
            "roofline": roofline,
            "roofline_hbm": roofline_hbm,
        }
        # launches per step and the fraction of the step with a matrix-core kernel running: from the committed kernel trace of
        # this same command (tools/stream_timeline.py --json over a rocprofv3 --kernel-trace run), tied to the sources by digest
        try:
            tl = json.load(open(os.path.join(ROOT, "profiles", "r04_stream_timeline.json")))
            stale = tl.get("csrc_digest") != csrc_digest()
            line["launches_per_step"] = tl["launches_per_step"]
            line["mfma_active_fraction"] = tl["mfma_active_fraction"]
            line["timeline_source"] = "profiles/r04_stream_timeline.json (rocprofv3 --kernel-trace of this command)" + \
                (": taken with other kernel sources" if stale else "")
            # (the tracer slows the host to about the GPU's pace and the two views' forward passes stop overlapping: the fraction is a
            # lower bound for the untraced step, whose phase schedule is in profiles/r04_phase_events.txt)
            line["timeline_note"] = "traced step %.1f ms; lower bound for the untraced step (profiles/r04_phase_events.txt)" % tl.get("step_ms", float("nan"))
        except Exception:
            line["launches_per_step"] = line["mfma_active_fraction"] = None
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.size)
        print(json.dumps(line), flush=True)
    if world > 1 or forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
