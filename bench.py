#!/usr/bin/env python
"""bench.py — headline benchmark: 1920x1080 SIFT frames/s (+ 100k x 100k match Mpairs/s).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per
GPU with torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the env, backend "nccl" = RCCL).

A step = one pass of the hot path (ExtractSift: LowPass -> pyramid -> DoG -> extrema -> orientation ->
descriptors -> count read-back, mainSift.cpp:58-67 parameters) over one batch of FRAMES_PER_GPU
synthetic 1920x1080 frames already resident in HBM, followed — when N > 1 — by the RCCL gather of the
valid SiftPoint records to rank 0 (BASELINE config 4).  Per-GPU work is fixed as N grows ("weak").
Rank 0 prints ONE JSON line; `value` is whole-job frames/s.

torch is used for device memory, streams and torch.distributed only; all compute goes through the
C-ABI of libmisift.so (cudasift_amd.capi).  The oracle is used only for the `cpu_baseline` leg.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 1920, 1080
NUM_OCTAVES, INIT_BLUR, THRESH, MAX_PTS = 5, 1.0, 3.0, 32768
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32 matrix peak


def octave_pixels(w, h, n):
    out = []
    for _ in range(n):
        out.append(w * h)
        w //= 2
        h //= 2
    return out          # finest first


def algorithmic_bytes_per_frame():
    """fp32, each intermediate written once and read once."""
    N = octave_pixels(W, H, NUM_OCTAVES)
    lowpass = 8 * N[0]
    scaledown = sum(4 * N[i] + 4 * N[i + 1] for i in range(NUM_OCTAVES - 1))
    laplace = sum(32 * n for n in N)
    findpoints = sum(28 * n for n in N)
    return {"lowpass": lowpass, "scaledown": scaledown, "laplace": laplace, "detect": findpoints,
            "dog_scan": laplace + findpoints}


def gen_frames_torch(torch, nframes, first, device):
    """tests/synth.py recipe on the GPU (torch.fft): equal-energy octave bands of Gaussian noise."""
    import math
    g = torch.Generator(device=device)
    fy = torch.fft.fftfreq(H, device=device)[:, None]
    fx = torch.fft.rfftfreq(W, device=device)[None, :]
    r2 = fx * fx + fy * fy
    frames = torch.empty((nframes, H, W), dtype=torch.float32, device=device)
    for f in range(nframes):
        g.manual_seed(0x51F7 + first + f)
        acc = torch.zeros((H, W), dtype=torch.float32, device=device)
        for j in range(6):
            w = torch.randn((H, W), generator=g, device=device, dtype=torch.float32)
            sigma = 2.0 ** j
            gk = torch.exp(-2.0 * (math.pi ** 2) * (sigma ** 2) * r2)
            b = torch.fft.irfft2(torch.fft.rfft2(w) * gk, s=(H, W))
            acc += b / b.std()
        frames[f] = torch.clamp(128.0 + 26.0 * acc / acc.std(), 0.0, 255.0)
    return frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-gpu", type=int, default=64)
    ap.add_argument("--match-n", type=int, default=100000)
    ap.add_argument("--no-match", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=6)
    ap.add_argument("--unfused", action="store_true", help="separate laplace/detect kernels (DoG planes in HBM)")
    ap.add_argument("--selftest-dist", action="store_true", help="single GPU: run the RCCL count all-gather / barrier path with world_size 1")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-frame latency side measurement")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive (H2D + extract + D2H) side measurement")
    args = ap.parse_args()

    import numpy as np
    import torch                     # first: libmisift.so then binds to torch's HIP runtime
    import torch.distributed as dist
    from cudasift_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)
    elif args.selftest_dist:                                  # single GPU: exercise the RCCL path with one rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=0, world_size=1)

    B = args.frames_per_gpu
    stream = torch.cuda.current_stream()
    ctx = capi.Context(local_rank, stream.cuda_stream)
    ctx.set_options(quiet=1, fused=0 if args.unfused else 1)

    # ---------------- inputs resident in HBM before the timed region
    frames = gen_frames_torch(torch, B, rank * B, device)                   # [B,1080,1920], pitch 1920
    S = capi.scratch_floats(W, H, NUM_OCTAVES, False)
    scratch = torch.empty((B * S,), dtype=torch.float32, device=device)
    pts = torch.zeros((B * MAX_PTS * 576,), dtype=torch.uint8, device=device)
    counts = (C.c_int * B)()
    torch.cuda.synchronize()

    from cudasift_amd.dist import RecordGather

    # Software-pipelined step loop, the same for every N: batch k is extracted AND packed on the device
    # (misift_extract_batch_packed_async, nothing synchronises), then the host completes batch k-1: reads its
    # per-frame counts back and — with more than one GPU — gathers the packed SiftPoint records of all ranks on
    # rank 0 over RCCL/xGMI on a communication stream (BASELINE config 4), overlapping batch k's extraction.
    # Every batch's read-back/gather completes before the closing barrier: nothing is skipped, only overlapped.
    NSLOT, LAG = 3, 2          # batch k-2 is completed after batch k was queued: the GPU never waits for the host
    packed = [torch.empty((B * MAX_PTS * 576,), dtype=torch.uint8, device=device) for _ in range(NSLOT)]
    cnts = [torch.zeros((2 * B + 1,), dtype=torch.int32, device=device) for _ in range(NSLOT)]
    gather = RecordGather(dist, torch, rank, world, device, dst=0, nslots=NSLOT,
                          force_collectives=args.selftest_dist)
    torch.cuda.synchronize()

    def enqueue(k):
        slot = k % NSLOT
        fe = gather.free_event(slot)
        if fe is not None:
            torch.cuda.current_stream().wait_event(fe)          # slot's previous transfer has left the buffers
        capi.check(capi.lib().misift_extract_batch_packed_async(
            ctx.h, frames.data_ptr(), B, H * W, W, H, W, NUM_OCTAVES, INIT_BLUR, THRESH, 0.0, scratch.data_ptr(),
            pts.data_ptr() if args.unfused else None,      # merged-octave path writes the packed array directly
            MAX_PTS, cnts[slot].data_ptr(), cnts[slot][B:].data_ptr(), packed[slot].data_ptr()),
            "misift_extract_batch_packed_async")
        ev = torch.cuda.Event()
        ev.record()
        gather.post(slot, cnts[slot][:B], packed[slot], ev)

    trace = [] if os.environ.get("BENCH_TRACE") else None

    def run(nsteps):
        res = None
        for k in range(nsteps):
            enqueue(k)
            if trace is not None:
                trace.append(("enq", k, time.perf_counter()))
            if k >= LAG:
                res = gather.complete((k - LAG) % NSLOT)
                if trace is not None:
                    trace.append(("done", k - LAG, time.perf_counter()))
        for k in range(max(0, nsteps - LAG), nsteps):
            res = gather.complete(k % NSLOT)
            if trace is not None:
                trace.append(("done", k, time.perf_counter()))
        all_counts = res[0]
        if (all_counts < 0).any():
            raise RuntimeError("candidate list overflow in the bench workload")
        return all_counts

    def barrier():
        if world > 1 or args.selftest_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    all_counts = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if trace is not None and rank == 0:
        for kind, k, t in trace[-(3 * args.steps):]:
            if t >= t0:
                print("trace %-4s %3d %8.3f ms" % (kind, k, 1e3 * (t - t0)), file=sys.stderr)
        print("trace end %8.3f ms" % (1e3 * dt), file=sys.stderr)
    n = all_counts[rank if world > 1 else 0]
    kp_per_frame = float(np.mean(n))
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = 1e3 * dt / args.steps
    fps = world * B * args.steps / dt

    # ---------------- distribution of the synchronous step (median + p10/p90), same workload, one
    # misift_extract_batch call incl. its count read-back per sample
    dist_ms = None
    if rank == 0:
        ts = []
        for _ in range(40):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            capi.check(capi.lib().misift_extract_batch(ctx.h, frames.data_ptr(), B, H * W, W, H, W, NUM_OCTAVES,
                                                       INIT_BLUR, THRESH, 0.0, scratch.data_ptr(), pts.data_ptr(),
                                                       MAX_PTS, counts), "misift_extract_batch")
            ts.append(1e3 * (time.perf_counter() - t1))
        ts = np.sort(np.array(ts[8:]))
        dist_ms = {"p10": round(float(np.percentile(ts, 10)), 4), "p50": round(float(np.percentile(ts, 50)), 4),
                   "p90": round(float(np.percentile(ts, 90)), 4), "samples": int(len(ts)),
                   "note": "synchronous misift_extract_batch of the same 64-frame batch (no pipelining)"}

    # ---------------- per-kernel durations (HIP events on the launch stream) for the roofline
    ctx.profile_reset()
    ctx.profile_enable(True)
    psteps = max(2, min(args.steps, 5))
    for _ in range(psteps):
        capi.check(capi.lib().misift_extract_batch(ctx.h, frames.data_ptr(), B, H * W, W, H, W, NUM_OCTAVES,
                                                   INIT_BLUR, THRESH, 0.0, scratch.data_ptr(), pts.data_ptr(),
                                                   MAX_PTS, counts), "misift_extract_batch")
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    alg = algorithmic_bytes_per_frame()
    if "lowpass_down" in prof:
        # fused prefilter + first ScaleDown: its algorithmic bytes are the sum of the two reference kernels' figures
        # (LowPass 8*N0 + ScaleDown_1 4*N0 + 4*N1); the remaining ScaleDown launches cover levels 2..
        N = octave_pixels(W, H, NUM_OCTAVES)
        alg["lowpass_down"] = 8 * N[0] + 4 * N[0] + 4 * N[1]
        alg["scaledown"] = sum(4 * N[i] + 4 * N[i + 1] for i in range(1, NUM_OCTAVES - 1))
    kernels = {}
    for name, p in prof.items():
        per_step_ms = p["total_ms"] / psteps
        e = {"ms_per_step": round(per_step_ms, 4), "launches_per_step": p["calls"] // psteps}
        if name in alg:
            e["alg_GBps"] = round(alg[name] * B / (per_step_ms * 1e-3) / 1e9, 1)
        kernels[name] = e
    dom = max((k for k in kernels if k in alg), key=lambda k: kernels[k]["ms_per_step"])
    dom_ms = kernels[dom]["ms_per_step"]
    dom_launches = max(1, kernels[dom]["launches_per_step"])
    achieved = alg[dom] * B / (dom_ms * 1e-3) / 1e9
    traffic = None
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tj):                       # bytes per frame per kernel from the committed rocprofv3 PMC passes
        try:
            t = json.load(open(tj))
            if dom in t.get("bytes_per_frame", {}):
                traffic = int(t["bytes_per_frame"][dom] * B / dom_launches)
        except Exception:
            traffic = None
    roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_note": "HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json: "
                                "2*FETCH_SIZE + WRITE_SIZE, gfx950 correction), not collected live",
                "alg_bytes_per_launch": int(alg[dom] * B / dom_launches),
                "avg_launch_ms": round(dom_ms / dom_launches, 4),
                "pipeline_alg_GBps": round(197.2e6 * fps / world / 1e9, 1),
                "pipeline_frac": round(197.2e6 * fps / world / 1e9 / HBM_PEAK_GBS, 4),
                "note": "dog_scan fuses LaplaceMulti+FindPointsMulti: its algorithmic bytes (60 B/px, SURVEY 8d) never "
                        "reach HBM (see traffic), so achieved > peak is possible; the kernel itself is fp32-VALU-bound"}
    # the genuinely HBM-bound kernels, same definition (algorithmic bytes / summed launch time)
    hbm_kernels = {}
    tbytes = {}
    if os.path.exists(tj):
        try:
            tbytes = json.load(open(tj)).get("bytes_per_frame", {})
        except Exception:
            tbytes = {}
    split = kernels.get("dog_scan", {}).get("launches_per_step", 1) > 1
    for k in ("lowpass", "lowpass_down", "scaledown"):
        if k == "scaledown" and split:
            continue      # runs beside the fine-level scan on a second stream: its duration says nothing about HBM
        if k in kernels:
            a = alg[k] * B / (kernels[k]["ms_per_step"] * 1e-3) / 1e9
            hbm_kernels[k] = {"achieved": round(a, 1), "frac": round(a / HBM_PEAK_GBS, 4)}
            if k in tbytes:          # bytes actually moved (PMC): the fused prefilter never re-reads the finest level
                hbm_kernels[k]["traffic_GBps"] = round(tbytes[k] * B / (kernels[k]["ms_per_step"] * 1e-3) / 1e9, 1)
    roofline["hbm_bound_kernels"] = hbm_kernels
    # achievable-copy ceiling: device-to-device copy of 1 GiB, read + write bytes counted
    if rank == 0:
        a = torch.empty(1 << 28, dtype=torch.float32, device=device)
        b = torch.empty_like(a)
        for _ in range(2):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2.0 * a.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
        roofline["copy_ceiling_GBps"] = round(copy_gbs, 1)
        for k in hbm_kernels:
            if "traffic_GBps" in hbm_kernels[k]:
                hbm_kernels[k]["traffic_frac_of_copy_ceiling"] = round(hbm_kernels[k]["traffic_GBps"] / copy_gbs, 4)

    # ---------------- single-frame latencies (BASELINE configs 2 and 3; reported, never `value`)
    latency = None
    if rank == 0 and world == 1 and not args.no_latency:
        latency = {"note": "median wall time of one synchronous call, frame / records resident in HBM "
                           "(misift_extract incl. its count read-back; misift_match of 2 x ~2000 features)"}
        one = C.c_int(0)
        for (lw, lh), key in (((1920, 1080), "extract_1920x1080_ms"), ((1280, 960), "extract_1280x960_ms")):
            img = frames[0, :lh, :lw].contiguous()
            ts = []
            for i in range(60):
                t1 = time.perf_counter()
                capi.check(capi.lib().misift_extract(ctx.h, img.data_ptr(), lw, lh, lw, NUM_OCTAVES, INIT_BLUR, THRESH,
                                                     0.0, 0, scratch.data_ptr(), pts.data_ptr(), MAX_PTS, C.byref(one)),
                           "misift_extract")
                ts.append(time.perf_counter() - t1)
            latency[key] = round(1e3 * float(np.median(ts[10:])), 4)
            latency[key.replace("_ms", "_keypoints")] = int(one.value)
        npts = int(one.value) // 32 * 32
        if npts >= 64:
            a = pts[: npts * 576].clone()
            ts = []
            for i in range(40):
                t1 = time.perf_counter()
                capi.check(capi.lib().misift_match(ctx.h, a.data_ptr(), npts, pts.data_ptr(), npts), "misift_match")
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            latency["match_%dx%d_ms" % (npts, npts)] = round(1e3 * float(np.median(ts[10:])), 4)

    # ---------------- PCIe-inclusive side measurement (never `value`): pinned host frames -> H2D -> extract -> D2H
    pcie = None
    if rank == 0 and world == 1 and not args.no_pcie:
        # host-fed pipeline (misift_pipe_*): pinned host frames -> H2D | extraction | packed records -> D2H on three
        # streams, 3 batches in flight; fp32 frames (what the reference uploads) and 8-bit frames
        torch.cuda.synchronize()
        nb, nbatches = 16, 12
        pcie = {"batch_frames": nb, "batches": nbatches, "depth": 3,
                "note": "misift_pipe: pinned host frames uploaded, valid SiftPoint records packed and downloaded, "
                        "upload/compute/read-back overlapped; never `value`"}
        host_recs = capi.PinnedArray((nb * 4096,), capi.POINT_DTYPE)
        for key, dt in (("frames_per_s_u8", np.uint8), ("frames_per_s_f32", np.float32)):
            src = capi.PinnedArray((nb, H, W), dt)
            f = frames[:nb].round().clamp(0, 255)
            src.array[...] = f.cpu().numpy().astype(dt)
            pipe = capi.Pipe(ctx, W, H, nb, src_u8=(dt == np.uint8), num_octaves=NUM_OCTAVES, init_blur=INIT_BLUR,
                             thresh=THRESH, max_pts=MAX_PTS, depth=3)

            def run(k):
                tot = 0
                for i in range(k):
                    if pipe.pending() == 3:
                        tot += pipe.collect(host_recs.ptr, nb * 4096)[1]
                    pipe.submit(src.ptr, nb)
                while pipe.pending():
                    tot += pipe.collect(host_recs.ptr, nb * 4096)[1]
                return tot
            run(3)
            t0 = time.perf_counter()
            tot = run(nbatches)
            pdt = time.perf_counter() - t0
            pcie[key] = round(nb * nbatches / pdt, 1)
            pcie["records_per_frame"] = round(tot / (nb * nbatches), 1)
            pipe.close()
            src.free()
        host_recs.free()

    # ---------------- matcher: n x n x 128 brute force on fp32 MFMA, row-block split over ranks
    match = None
    if not args.no_match:
        nm = args.match_n // (32 * world) * (32 * world)
        rows = nm // world
        rec = np.dtype(capi.POINT_DTYPE)
        gm = torch.Generator(device=device)
        gm.manual_seed(12345 + rank)

        def make_set(n):
            t = torch.zeros((n, 144), dtype=torch.float32, device=device)
            d = torch.rand((n, 128), generator=gm, device=device, dtype=torch.float32)
            t[:, 16:] = d * (128.0 ** 0.5 / d.sum(dim=1, keepdim=True))       # match.cu:945-957 recipe
            return t
        set1 = make_set(nm) if world == 1 else None
        shard2 = make_set(rows)
        if world == 1:
            set2 = shard2
            my1 = set1
            row0 = 0
        else:
            my1 = make_set(rows)          # this rank's row block of set 1 (rows [rank*rows, ...))
            set2 = torch.empty((nm, 144), dtype=torch.float32, device=device)
            row0 = 0
        torch.cuda.synchronize()

        def mstep():
            if world > 1:
                dist.all_gather_into_tensor(set2, shard2)                     # set-2 descriptors over xGMI
            capi.check(capi.lib().misift_match_rows(ctx.h, my1.data_ptr(), row0, rows, set2.data_ptr(), nm),
                       "misift_match_rows")
        mstep()
        barrier()
        msteps = 3
        t0 = time.perf_counter()
        for _ in range(msteps):
            mstep()
        barrier()
        mdt = (time.perf_counter() - t0) / msteps
        if world > 1:
            tmax = torch.tensor([mdt], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            mdt = float(tmax.item())
        ctx.profile_reset()
        ctx.profile_enable(True)
        mstep()
        mp = ctx.profile_read()
        ctx.profile_enable(False)
        kms = mp.get("match_mfma", {"total_ms": 0.0})["total_ms"]
        flops = 2.0 * 128 * rows * nm
        match = {"metric": "match Mpairs/s", "value": round(nm * float(nm) / mdt / 1e6, 1), "n1": nm, "n2": nm,
                 "ms": round(mdt * 1e3, 3), "split": "row-block x%d" % world,
                 "roofline": {"kernel": "match_mfma", "bound": "mfma",
                              "achieved": round(flops / (kms * 1e-3) / 1e12, 2) if kms > 0 else None,
                              "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                              "frac": round(flops / (kms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4) if kms > 0 else None,
                              "kernel_ms": round(kms, 3)}}

    # ---------------- CPU baseline (rank 0, N = 1 only): the oracle port on a bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import pyoracle as orc
        host = frames[:args.cpu_frames].cpu().numpy()
        orc.extract(host[0], NUM_OCTAVES, INIT_BLUR, THRESH)      # warm-up (page-in, OpenMP pool)
        t0 = time.perf_counter()
        tot = 0
        for f in range(args.cpu_frames):
            _, nn, _ = orc.extract(host[f], NUM_OCTAVES, INIT_BLUR, THRESH)
            tot += nn
        cdt = time.perf_counter() - t0
        cpu = {"value": round(args.cpu_frames / cdt, 3), "unit": "frames/s", "cores": os.cpu_count(),
               "kind": "port",
               "sample": "%d of the same synthetic 1920x1080 frames, oracle/sift_oracle.c with OpenMP "
                         "(OpenCV cv::SIFT is not installed on this image)" % args.cpu_frames,
               "keypoints_per_frame": round(tot / args.cpu_frames, 1)}

        # matcher CPU baseline: the reference's OWN AVX2/OpenMP routine MatchC3 (match.cu:102-130, built from the
        # reference tree into oracle/_ref by oracle/build_ref.sh) on its own 16384 x 16384 problem
        L = orc.ref_lib(16384)
        if L is not None and match is not None:
            a = orc.aligned_f32(16384 * 128); b = orc.aligned_f32(16384 * 128)
            sc = orc.aligned_f32(16384); ix = np.zeros(16384, np.int32)
            L.ref_generate(a.ctypes.data, b.ctypes.data, 1)
            L.ref_match_c3(a.ctypes.data, b.ctypes.data, sc.ctypes.data, ix.ctypes.data)     # warm-up
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                L.ref_match_c3(a.ctypes.data, b.ctypes.data, sc.ctypes.data, ix.ctypes.data)
            mdt = (time.perf_counter() - t0) / reps
            match["cpu_baseline"] = {"value": round(16384.0 * 16384.0 / mdt / 1e6, 1), "unit": "Mpairs/s",
                                     "cores": os.cpu_count(), "kind": "reference",
                                     "sample": "reference MatchC3 (AVX2+FMA, OpenMP; argmax only, no runner-up) on "
                                               "16384 x 16384 x 128, its own generator"}

    if rank == 0:
        out = {"metric": "1920x1080 SIFT frames/sec", "value": round(fps, 1), "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": "batch of %d synthetic 1920x1080 frames per GPU (BASELINE config 4: 512 "
                                      "frames over 8 GPUs), ExtractSift 5 octaves initBlur 1.0 thresh 3.0 "
                                      "maxPts 32768, frames resident in HBM, count read-back%s" %
                                      (B, " + RCCL gather of SiftData to rank 0" if world > 1 else ""),
                          "frames_per_gpu": B, "path": "unfused" if args.unfused else "fused dog+detect",
                          "keypoints_per_frame": round(kp_per_frame, 1)},
               "roofline": roofline, "kernels": kernels, "match": match, "cpu_baseline": cpu, "pcie_inclusive": pcie, "single_frame": latency,
               "sync_step_ms": dist_ms,
               "kernels_note": "dog_scan runs as two launches per step (fine levels on the context stream, the coarse "
                               "ScaleDowns + coarse levels beside it on a second stream): their durations overlap, so "
                               "the per-kernel times add up to more than the step"}
        print(json.dumps(out))
    if world > 1 or args.selftest_dist:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
