#!/usr/bin/env python
"""bench.py — headline benchmark: 1920x1080 SIFT frames/s (+ 100k x 100k match Mpairs/s).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 either a launcher starts one rank per GPU
(torch.distributed.run: RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the env) or — the plain command — bench.py spawns the N
ranks itself (spawn_ranks) and refuses to run when fewer than N devices are visible.

A step = one pass of the hot path (ExtractSift: LowPass -> pyramid -> DoG -> extrema -> orientation ->
descriptors -> count read-back, mainSift.cpp:58-67 parameters) over one batch of FRAMES_PER_GPU synthetic
1920x1080 frames already resident in HBM.  Every rank holds 8 such batches of DISTINCT frames (512 frames = the
whole BASELINE config 4 job at N = 1) and rotates through them; with N > 1 the valid SiftPoint records of every
batch are gathered on ONE rank over RCCL/xGMI (misift_gather_*; the root rotates with the step by default, --gather-root), pipelined under the following batches.  Per-GPU
work is fixed as N grows ("weak").  Rank 0 prints ONE JSON line; `value` is whole-job frames/s.

torch is plumbing only (device memory, the frame generator, rendezvous/barrier): all compute AND the data-path
collectives go through the C-ABI of libmisift.so (cudasift_amd.capi).  The oracle is the checker of the
self-validation and the thing timed in the `cpu_baseline` leg — never part of the measured path.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 1920, 1080
NUM_OCTAVES, INIT_BLUR, THRESH, MAX_PTS = 5, 1.0, 3.0, 32768
NUM_BATCHES = 8                # distinct 64-frame batches per rank (8 x 64 = 512 frames, BASELINE config 4)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
VALU_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: peak FP32 (vector), 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32 matrix peak
ALG_BYTES_PER_FRAME = 197.2e6
# structural floor of a fused design: input 8.3 MB + pyramid written once 11.05 MB + read once by the
# scan 11.05 MB + records 1.2 MB
FLOOR_BYTES_PER_FRAME = 31.6e6

KERNEL_NAMES = {"lowpass_kernel": "lowpass", "lowpass_down_kernel": "lowpass_down", "scaledown_kernel": "scaledown",
                "scaledown_tail_kernel": "scaledown", "dog_scan_all_kernel": "dog_scan", "dog_scan_kernel": "dog_scan",
                "refine_all_kernel": "refine", "orient_all_kernel": "orient_all", "descr_all_kernel": "descr_all",
                "orient_all_gather_kernel": "orient_all", "descr_all_gather_kernel": "descr_all", "descr_big_kernel": "descr_all",
                "bin_detections_kernel": "bin_detections", "renumber_dups_kernel": "renumber_dups",
                "laplace_kernel": "laplace", "detect_kernel": "detect", "match_kernel": "match_mfma"}
import numpy as _np
RESULT_DTYPE_NP = _np.dtype([("score", "<f4"), ("ambiguity", "<f4"), ("match", "<i4")])     # misift_match_sharded's 12 B/row
GATHER_KERNELS = ("refine", "orient_all", "descr_all")      # scattered 8-byte reads, not wide streaming


def effective_cpus():
    """(cpus this process may use, logical cpus of the host, note).  A container's CFS quota (cgroup cpu.max) caps the
    CPU TIME it gets whatever os.cpu_count() says: on the GPU boxes of this pool 256 logical CPUs are visible and the
    quota is 16 — r02 ran one frame on each of the 256 and measured throttling (0.2 frames/s per 'core';
    profiles/r03_cpu_baseline_sweep.json: 9 frames/s per thread up to 16 threads, falling beyond)."""
    logical = os.cpu_count() or 1
    try:
        logical = min(logical, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota, note = None, "no CPU quota"
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())         # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None and quota < logical:
        return max(1, int(quota)), logical, "cgroup CPU quota %.1f of %d logical CPUs" % (quota, logical)
    return logical, logical, note


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


def dog_scan_flop_per_px():
    """fp32 operations the fused scan NEEDS per pixel of a pyramid level (kernels_dog.hip scan_strip), counted from
    the blur structure: it evaluates the 6 blurs (scales 1..6) whose 5 differences are the centre DoG planes.
      shared by the 6 blurs : 4 vertical pair sums r[-j] + r[+j]                         =   4
      per blur, vertical    : 1 mul + 4 fma (centre tap, then 4 symmetric pairs)         =   9
      per blur, horizontal  : 4 pair sums + 1 mul + 4 fma                                =  13
      5 DoG differences, 5 |v| folded into the running maximum (abs is a free modifier)  =  10
    = 4 + 6 * 22 + 10 = 146 flop/px (an fma counts 2).  The extremum tests run on < 1 % of the rows and are not
    counted; neither are DPP moves, selects or address arithmetic — they are overhead against this roof."""
    shared, per_blur, dog = 4, (1 + 2 * 4) + (4 + 1 + 2 * 4), 5 + 5
    return shared + 6 * per_blur + dog


def gen_frames_torch(torch, nframes, first, device, out=None):
    """tests/synth.py recipe on the GPU (torch.fft): equal-energy octave bands of Gaussian noise."""
    import math
    g = torch.Generator(device=device)
    fy = torch.fft.fftfreq(H, device=device)[:, None]
    fx = torch.fft.rfftfreq(W, device=device)[None, :]
    r2 = fx * fx + fy * fy
    gks = [torch.exp(-2.0 * (math.pi ** 2) * ((2.0 ** j) ** 2) * r2) for j in range(6)]
    frames = out if out is not None else torch.empty((nframes, H, W), dtype=torch.float32, device=device)
    for f in range(nframes):
        g.manual_seed(0x51F7 + first + f)
        acc = torch.zeros((H, W), dtype=torch.float32, device=device)
        for j in range(6):
            w = torch.randn((H, W), generator=g, device=device, dtype=torch.float32)
            b = torch.fft.irfft2(torch.fft.rfft2(w) * gks[j], s=(H, W))
            acc += b / b.std()
        frames[f] = torch.clamp(128.0 + 26.0 * acc / acc.std(), 0.0, 255.0)
    return frames


# ------------------------------------------------------------------------------------------------ PMC (live)
def pmc_child():
    """Run under `rocprofv3 --pmc ...` by collect_pmc(): 1 warm-up + 3 steps of the timed entry point on one batch."""
    import torch
    from cudasift_amd import capi
    device = torch.device("cuda", 0)
    B = int(os.environ.get("BENCH_PMC_FRAMES", "64"))
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_options(quiet=1)
    nsteps = int(os.environ.get("BENCH_CHILD_STEPS", "4"))
    pipelined = os.environ.get("BENCH_CHILD_PIPELINED", "0") == "1"      # trace pass: queued back to back like the timed loop
    ring = int(os.environ.get("BENCH_CHILD_RING", "1"))                  # tools/overlap_report.py: K batches in flight, like `value`
    if ring > 1:
        ctx.set_batches_in_flight(ring)
    nb = 4 if pipelined else 1       # distinct batches to rotate over: the 256 MB Infinity Cache must not serve the input
    frames = gen_frames_torch(torch, nb * B, 0, device)
    S = capi.scratch_floats(W, H, NUM_OCTAVES, False)
    scratch = [torch.empty((B * S,), dtype=torch.float32, device=device) for _ in range(ring)]
    packed = [torch.empty((B * MAX_PTS * 576,), dtype=torch.uint8, device=device) for _ in range(ring)]
    cnts = [torch.zeros((2 * B + 1,), dtype=torch.int32, device=device) for _ in range(ring)]
    for it in range(nsteps):
        r = it % ring
        capi.check(capi.lib().misift_extract_batch_packed_async(
            ctx.h, frames[(it % nb) * B].data_ptr(), B, H * W, W, H, W, NUM_OCTAVES, INIT_BLUR, THRESH, 0.0, scratch[r].data_ptr(), None,
            MAX_PTS, cnts[r].data_ptr(), cnts[r][B:].data_ptr(), packed[r].data_ptr()), "misift_extract_batch_packed_async")
        if not pipelined:
            torch.cuda.synchronize()
    ctx.sync()
    torch.cuda.synchronize()
    ctx.close()


def _parse_pmc_csv(path):
    import collections
    import csv
    tot, n = collections.defaultdict(float), collections.Counter()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
            if k in KERNEL_NAMES:
                tot[KERNEL_NAMES[k]] += float(r["Counter_Value"])
                n[KERNEL_NAMES[k]] += 1
    return tot, n


def collect_pmc(frames_per_launch, gather_read_factor, timeout_s=240):
    """Two separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass, MI355X_MICROARCH.md) over a
    4-step child run of the timed entry point.  Returns {kernel: {read, write, launches}} in BYTES PER STEP with the
    guide's gfx950 correction (FETCH_SIZE in KB counts half the bytes of wide coalesced reads: x2; the scattered
    8-byte gathers of the per-keypoint kernels use the factor calibrated by tools/pmc_calib), or (None, reason)."""
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    res = {}
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", BENCH_PMC_FRAMES=str(frames_per_launch))
    env.pop("RANK", None)
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child"]
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, p.returncode,
                                                                       p.stdout.decode(errors="replace")[-300:])
            tot, n = _parse_pmc_csv(files[0])
            steps = 4.0
            for k in tot:
                e = res.setdefault(k, {"read": 0.0, "write": 0.0, "launches": 0})
                if counter == "FETCH_SIZE":
                    factor = gather_read_factor if k in GATHER_KERNELS else 2.0
                    e["read"] = factor * tot[k] * 1024.0 / steps
                    e["launches"] = n[k] / steps
                else:
                    e["write"] = tot[k] * 1024.0 / steps
    except Exception as e:                                   # noqa: BLE001 — the bench line must still come out
        return None, "PMC collection failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res, None


def interval_union_ns(intervals):
    """Total length of the union of [start, end) intervals (ns): overlapping launches are counted once."""
    tot, cur_a, cur_b = 0, None, None
    for a, b in sorted(intervals):
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        elif b > cur_b:
            cur_b = b
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def collect_sq(frames_per_launch, timeout_s=240):
    """One more rocprofv3 pass over the same 4-step child run: SQ_INSTS_VALU (wavefront-level VALU instructions),
    SQ_ACTIVE_INST_VALU (quad-cycles a SIMD spends issuing them) and GRBM_GUI_ACTIVE (shader cycles of the dispatch), per
    kernel and STEP.  Returns ({kernel: {"insts", "active_quad_cycles", "gui_cycles"}}, None) or (None, reason)."""
    import collections
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="bench_sq_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", BENCH_PMC_FRAMES=str(frames_per_launch))
    env.pop("RANK", None)
    try:
        cmd = [exe, "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "-d", tmp, "-o", "p",
               "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__), "--pmc-child"]
        p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            return None, "rocprofv3 --pmc SQ_* failed (rc %d): %s" % (p.returncode, p.stdout.decode(errors="replace")[-300:])
        res = collections.defaultdict(lambda: {"insts": 0.0, "active_quad_cycles": 0.0, "gui_cycles": 0.0})
        key = {"SQ_INSTS_VALU": "insts", "SQ_ACTIVE_INST_VALU": "active_quad_cycles", "GRBM_GUI_ACTIVE": "gui_cycles"}
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
                if k in KERNEL_NAMES and r["Counter_Name"] in key:
                    res[KERNEL_NAMES[k]][key[r["Counter_Name"]]] += float(r["Counter_Value"]) / 4.0      # 4 steps
        # dispatch durations of the SAME pass (kernels run one at a time under --pmc): GRBM_GUI_ACTIVE / duration = shader clock
        for tf in glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True):
            with open(tf) as f:
                for r in csv.DictReader(f):
                    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
                    if k in KERNEL_NAMES and KERNEL_NAMES[k] in res:
                        res[KERNEL_NAMES[k]]["ms"] = res[KERNEL_NAMES[k]].get("ms", 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 / 4.0
        return dict(res), None
    except Exception as e:                                   # noqa: BLE001 — the bench line must still come out
        return None, "SQ counter pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def collect_trace(frames_per_launch, steps=30, skip=5, timeout_s=240, extra_env=None):
    """Kernel durations as rocprofv3 itself sees them: ONE `rocprofv3 --kernel-trace` pass (no counters) over a child run
    of `steps` back-to-back calls of the timed entry point; the first `skip` launches of every kernel are warm-up.
    Returns {kernel: {"ms_per_step": summed launch durations of a step, "launches_per_step": n}} or (None, reason).  (The library's own HIP-event
    pairs bracket every launch with two markers on the stream and read 6-17 % longer than the dispatch itself.)"""
    import collections
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="bench_trace_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", BENCH_PMC_FRAMES=str(frames_per_launch), BENCH_CHILD_STEPS=str(steps),
               BENCH_CHILD_PIPELINED="1")
    env.update(extra_env or {})
    env.pop("RANK", None)
    try:
        cmd = [exe, "--kernel-trace", "-d", tmp, "-o", "t", "--output-format", "csv", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child"]
        p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
        if p.returncode != 0 or not files:
            return None, "rocprofv3 --kernel-trace failed (rc %d): %s" % (p.returncode, p.stdout.decode(errors="replace")[-300:])
        per = collections.defaultdict(list)
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
                if k in KERNEL_NAMES:
                    per[KERNEL_NAMES[k]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        res = {}
        for k, v in per.items():
            v.sort()
            lps = max(1, int(round(len(v) / float(steps))))
            d = [(b - a) * 1e-6 for a, b in v[skip * lps:]]
            if d:
                res[k] = {"ms_per_step": sum(d) * lps / len(d), "launches_per_step": lps, "launches": len(d),
                          # wall time during which AT LEAST ONE launch of this kernel was running, per step: launches that
                          # run side by side (dog_scan: fine | coarse levels on two streams) count once
                          "union_ms_per_step": interval_union_ns(v[skip * lps:]) * 1e-6 * lps / len(d)}
        allv = sorted(iv for k, v in per.items() for iv in v[skip * max(1, int(round(len(v) / float(steps)))):])
        if allv:
            nst = max(1, steps - skip)
            res["_all"] = {"busy_union_ms_per_step": interval_union_ns(allv) * 1e-6 / nst,
                           "span_ms_per_step": (max(b for _, b in allv) - allv[0][0]) * 1e-6 / nst}
        return res, None
    except Exception as e:                                   # noqa: BLE001 — the bench line must still come out
        return None, "kernel trace failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ matcher leg
class CabiMatchOps:
    """Production ops of the matcher leg: device memory through torch, compute + collectives through the C-ABI
    (misift_match_rows at N = 1; misift_match_sharded = set-2 all-gather + MFMA sweep + 12 B/row result all-gather
    on RCCL at N > 1)."""

    def __init__(self, torch, capi, ctx, comm, device, rank, world):
        self.torch, self.capi, self.ctx, self.comm, self.device, self.rank, self.world = torch, capi, ctx, comm, device, rank, world

    def to_device(self, recs):
        import numpy as np
        return self.torch.from_numpy(recs.view(np.uint8).reshape(-1)).to(self.device)

    def empty(self, nbytes):
        return self.torch.zeros((max(nbytes, 16),), dtype=self.torch.uint8, device=self.device)

    def match_step(self, rows1, nrows, shard2, nshard, set2_all, results_all):
        capi = self.capi
        if self.comm is None:
            capi.check(capi.lib().misift_match_rows(self.ctx.h, rows1.data_ptr(), 0, nrows, shard2.data_ptr(), nshard),
                       "misift_match_rows")
        else:
            self.comm.match_sharded(rows1.data_ptr(), nrows, shard2.data_ptr(), nshard, set2_all.data_ptr(),
                                    results_all.data_ptr())

    def to_host(self, t, count, dtype):
        import numpy as np
        return t[: count * np.dtype(dtype).itemsize].cpu().numpy().view(dtype).copy()

    def barrier(self):
        if self.comm is not None:
            self.comm.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        import torch.distributed as dist
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def kernel_ms(self, fn):
        self.ctx.profile_reset()
        self.ctx.profile_enable(True)
        fn()
        mp = self.ctx.profile_read()
        self.ctx.profile_enable(False)
        return mp.get("match_mfma", {"total_ms": 0.0})["total_ms"]


def matcher_leg(ops, rank, world, nm, msteps=3, validate_rows=100, l2=False, point_dtype=None, result_dtype=None,
                oracle=None):
    """BASELINE config 5: nm x nm x 128 brute force, set 1 split into row blocks (one per rank), set 2 sharded the
    same way and replicated by one all-gather per step, the 12 B/row results all-gathered afterwards.  Descriptors
    follow match.cu:945-957 (uniform, scaled by sqrt(128)/sum) from mt19937(12345) — or unit-L2 when `l2`.
    `ops` supplies memory, the per-step call and the reductions (CabiMatchOps in production; the world-size-2 gloo
    test drives this same function with the oracle standing in for the GPU).  Returns the result dict (rank 0 view)."""
    import numpy as np
    from synth import descriptors_to_points, synth_descriptors
    nm = nm // (32 * world) * (32 * world)
    rows = nm // world
    # every rank generates the same two sets (0.3 s) and keeps its row block / shard
    set1 = descriptors_to_points(synth_descriptors(nm, 12345, l2), point_dtype)
    set2 = descriptors_to_points(synth_descriptors(nm, 12346, l2), point_dtype)
    b = rank * rows
    my1_host = set1[b:b + rows].copy()
    rows1 = ops.to_device(my1_host)
    shard2 = ops.to_device(set2[b:b + rows].copy())
    set2_all = ops.empty(576 * nm) if world > 1 else ops.to_device(set2)
    results_all = ops.empty(12 * nm) if world > 1 else None

    def mstep():
        if world > 1:
            ops.match_step(rows1, rows, shard2, rows, set2_all, results_all)
        else:
            ops.match_step(rows1, rows, set2_all, nm, set2_all, None)
    mstep()
    ops.barrier()
    t0 = time.perf_counter()
    for _ in range(msteps):
        mstep()
    ops.barrier()
    mdt = ops.max_over_ranks((time.perf_counter() - t0) / msteps)
    kms = ops.kernel_ms(mstep)
    flops = 2.0 * 128 * rows * nm
    out = {"metric": "match Mpairs/s", "value": round(nm * float(nm) / mdt / 1e6, 1), "n1": nm, "n2": nm,
           "ms": round(mdt * 1e3, 3), "split": "row-block x%d" % world,
           "descriptors": ("unit-L2" if l2 else "match.cu:945-957 recipe (scaled by sqrt(128)/sum)") + ", mt19937(12345)",
           "exchange": ("set-2 all-gather + 12 B/row result all-gather on RCCL (misift_match_sharded)" if world > 1
                        else "none (one GPU)"),
           "roofline": {"kernel": "match_mfma", "bound": "mfma",
                        "achieved": round(flops / (kms * 1e-3) / 1e12, 2) if kms > 0 else None,
                        "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(flops / (kms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4) if kms > 0 else None,
                        "kernel_ms": round(kms, 3)}}
    # ---- spot check against the oracle's sequential-FMA definition (orc_match_rows -> orc_dot128 chains): ~100 rows
    # of this rank's block, score / index / ambiguity bit for bit; with N > 1 also the all-gathered 12-byte results
    if oracle is not None and validate_rows > 0:
        got = ops.to_host(rows1, rows, point_dtype)
        pick = np.unique(np.linspace(0, rows - 1, validate_rows).astype(int))
        ref = my1_host[pick].copy()
        oracle.match_rows(ref, 0, len(pick), set2, nm)
        for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
            if not np.array_equal(ref[f], got[f][pick]):
                raise RuntimeError("matcher self-validation failed: field %s differs from the oracle" % f)
        if world > 1:
            res = ops.to_host(results_all, nm, result_dtype)
            for f in ("score", "ambiguity", "match"):
                if not np.array_equal(res[f][b + pick], ref[f]):
                    raise RuntimeError("matcher self-validation failed: gathered result field %s" % f)
        out["validated_rows"] = int(len(pick))
    return out


def np_clip0(a):
    import numpy as np
    return np.maximum(a, 0)


class StepPipeline:
    """The software-pipelined step loop of ONE rank, the same for every N: batch k is extracted AND packed on the device
    (misift_extract_batch_packed_async, nothing synchronises), then the host completes batch k-LAG: reads its per-frame
    counts back and — with a communicator — gathers the packed SiftPoint records of all ranks on rank 0
    (misift_gather_post / misift_gather_complete on the communicator's own stream), overlapping the extraction of the
    batches queued behind it.  Every batch's read-back/gather completes before run() returns: nothing is skipped, only
    overlapped."""

    def __init__(self, torch, capi, ctxs, ctx_streams, comm, rank, world, device, frames, B, NB, scratches, pts, unfused,
                 ring=1, root_mode="fixed"):
        self.torch, self.capi, self.ctxs, self.ctx_streams, self.comm = torch, capi, ctxs, ctx_streams, comm
        # root of the gather of batch k: 0 ("fixed", BASELINE config 4 as written) or k % world ("rotate": every rank's
        # xGMI ingress takes its turn — 7 senders x ~77 MB per step into ONE GPU is more than its links carry at today's
        # per-GPU rate).  May be switched between run() calls.
        self.root_mode = root_mode
        self.gather_s, self.gather_calls, self.wire_bytes = 0.0, 0, 0
        self.rank, self.world, self.frames, self.B, self.NB, self.scratches = rank, world, frames, B, NB, scratches
        self.pts, self.unfused = pts, unfused
        NCTX = self.NCTX = len(ctxs)
        self.RING = ring                    # batches in flight INSIDE every context (misift_ctx_set_batches_in_flight)
        self.INFLIGHT = NCTX * ring
        assert len(scratches) >= self.INFLIGHT, "one scratch arena per batch in flight"
        self.LAG = self.INFLIGHT + 1        # the host completes batch k-LAG: INFLIGHT batches stay queued on the GPU
        NSLOT = self.NSLOT = self.LAG + 1
        if comm is not None and NSLOT > 8:                  # MISIFT_GATHER_SLOTS
            raise ValueError("contexts x batches-in-flight = %d needs %d gather slots, the communicator has 8" % (self.INFLIGHT, NSLOT))
        # with a ring the results of a call are not ordered on the context stream: raw HIP events recorded behind each
        # batch (misift_ctx_record_batch) mark completion.  (No waiting side streams: a stream that waits and shares a
        # hardware queue with a pipeline holds that pipeline up.)
        REC_CAP = self.REC_CAP = MAX_PTS    # mainSift.cpp:58-67 capacity (32768 records per frame)
        self.packed = [torch.empty((B * REC_CAP * 576,), dtype=torch.uint8, device=device) for _ in range(NSLOT)]
        self.cnts = [torch.zeros((2 * B + 1,), dtype=torch.int32, device=device) for _ in range(NSLOT)]
        self.done_ev = [None] * NSLOT
        # every rank that can be a root holds a receive buffer (rotating root: all of them)
        self.recv = (torch.empty((world * B * REC_CAP * 576,), dtype=torch.uint8, device=device)
                     if (comm and (rank == 0 or world > 1)) else None)
        # normal priority: high-priority streams share ONE hardware queue with the contexts' coarse-level streams
        # (2 contexts: 49 k -> 54 k frames/s)
        self.rb_stream = torch.cuda.Stream(device=device)
        self.step_ev = []
        self.host_t = {"enqueue": 0.0, "complete": 0.0}
        self.trace_host = os.environ.get("BENCH_TRACE") == "1"     # developer aid: where the host spends the pipelined loop

    def enqueue(self, k):
        torch, capi, B = self.torch, self.capi, self.B
        slot = k % self.NSLOT
        b0 = (k % self.NB) * B
        ci = k % self.NCTX
        capi.check(capi.lib().misift_extract_batch_packed_async(
            self.ctxs[ci].h, self.frames[b0].data_ptr(), B, H * W, W, H, W, NUM_OCTAVES, INIT_BLUR, THRESH, 0.0,
            self.scratches[k % self.INFLIGHT].data_ptr(),
            self.pts.data_ptr() if self.unfused else None,      # merged-octave path writes the packed array directly
            self.REC_CAP, self.cnts[slot].data_ptr(), self.cnts[slot][B:].data_ptr(), self.packed[slot].data_ptr()),
            "misift_extract_batch_packed_async")
        if self.comm is not None:
            self.comm.gather_post(slot, self.cnts[slot].data_ptr(), B, self.packed[slot].data_ptr(), ctx=self.ctxs[ci])
        if self.RING > 1:
            e = capi.HipEvent()
            self.ctxs[ci].record_batch(e)
            self.done_ev[slot] = e
        else:
            if self.comm is None:
                ev = torch.cuda.Event()
                ev.record(self.ctx_streams[ci])
                self.done_ev[slot] = ev
            e = torch.cuda.Event(enable_timing=True)
            e.record(self.ctx_streams[ci])
        self.step_ev.append(e)

    def complete(self, k):
        torch, B = self.torch, self.B
        slot = k % self.NSLOT
        if self.comm is not None:
            root = (k % self.world) if self.root_mode == "rotate" else 0
            tg = time.perf_counter()
            c, _ = self.comm.gather_complete(slot, B, root, self.recv.data_ptr() if (self.recv is not None and root == self.rank) else None,
                                             self.world * B * self.REC_CAP)
            self.gather_s += time.perf_counter() - tg
            self.gather_calls += 1
            # bytes that crossed the links for this batch: every sender's valid records into the root, plus the per-frame
            # counts every rank gets from every other (the all-gather in front of the point-to-point messages)
            valid = np_clip0(c)
            self.wire_bytes += int(valid.sum() - valid[root].sum()) * 576 + (self.world - 1) * self.world * B * 4
            return c
        with torch.cuda.stream(self.rb_stream):             # count read-back beside the running extraction
            if self.RING > 1:
                self.done_ev[slot].stream_wait(self.rb_stream.cuda_stream)
            else:
                self.rb_stream.wait_event(self.done_ev[slot])
            c = self.cnts[slot][:B].cpu().numpy()
        return c[None, :]

    def run(self, k0, nsteps):
        last, LAG = None, self.LAG
        for k in range(k0, k0 + nsteps):
            ta = time.perf_counter()
            self.enqueue(k)
            tb = time.perf_counter()
            if k - k0 >= LAG:
                last = self.complete(k - LAG)
            if self.trace_host:
                self.host_t["enqueue"] += tb - ta
                self.host_t["complete"] += time.perf_counter() - tb
            if k - k0 >= LAG:
                if (last < 0).any():
                    raise RuntimeError("candidate list overflow in the bench workload")
        for k in range(max(k0, k0 + nsteps - LAG), k0 + nsteps):
            last = self.complete(k)
        if (last < 0).any():
            raise RuntimeError("candidate list overflow in the bench workload")
        return last


# ------------------------------------------------------------------------------------------------ emulated ranks
class LoopbackMatchOps(CabiMatchOps):
    """CabiMatchOps for ranks that are host THREADS of this process (loopback world): the two reductions that main()
    does through torch.distributed go through a threading.Barrier."""

    def __init__(self, torch, capi, ctx, comm, device, rank, world, shared):
        super().__init__(torch, capi, ctx, comm, device, rank, world)
        self.shared = shared

    def barrier(self):
        self.comm.barrier()
        self.torch.cuda.current_stream().synchronize()
        self.shared["bar"].wait()

    def max_over_ranks(self, x):
        self.shared["vals"][self.rank] = x
        self.shared["bar"].wait()
        m = max(self.shared["vals"])
        self.shared["bar"].wait()
        return m


def emulate_ranks(args):
    """`--emulate-ranks N`: the N-rank driver loops of BASELINE configs 4 and 5 executed END TO END on ONE GPU — N host
    threads, N contexts of device 0, N communicators of a misift_loopback_world (include/misift.h): the same
    StepPipeline (extract + pack, gather_post / gather_complete to rank 0) and the same matcher_leg (misift_match_sharded:
    set-2 all-gather, MFMA sweep, result all-gather) as a real N-GPU run, device-to-device copies instead of RCCL.
    FUNCTIONAL ONLY: the ranks share one GPU, the line it prints carries no rate and must never be read as scaling."""
    import threading
    import numpy as np
    import torch
    from cudasift_amd import capi
    N = args.emulate_ranks
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, NB = max(1, min(args.frames_per_gpu, 4)), 2
    steps, warm = max(3, min(args.steps, 12)), min(args.warmup, 2)
    root_last = ((warm + steps - 1) % N) if args.gather_root == "rotate" else 0      # root of the LAST batch's gather
    nm = min(args.match_n, 16384)
    frames_of = []
    for r in range(N):                       # every rank's own frames, generated as main() generates them
        f = torch.empty((NB * B, H, W), dtype=torch.float32, device=device)
        gen_frames_torch(torch, NB * B, r * NB * B, device, out=f)
        frames_of.append(f)
    torch.cuda.synchronize()
    lw = capi.LoopbackWorld(N)
    shared = {"bar": threading.Barrier(N), "vals": [0.0] * N}
    res, errs = [None] * N, []
    orc = None
    if not args.no_cpu:
        from oracle import pyoracle as orc

    def body(rank):
        try:
            stream = torch.cuda.Stream(device=device)
            with torch.cuda.stream(stream):
                ctx = capi.Context(0, stream.cuda_stream)
                ctx.set_options(quiet=1)
                comm = capi.Comm(ctx, N, rank, lw)
                S = capi.scratch_floats(W, H, NUM_OCTAVES, False)
                ring = max(1, args.batches_in_flight)          # the same in-context pipelining as the real run
                if ring > 1:
                    ctx.set_batches_in_flight(ring)
                scr = [torch.empty((B * S,), dtype=torch.float32, device=device) for _ in range(ring)]
                stream.synchronize()
                pl = StepPipeline(torch, capi, [ctx], [stream], comm, rank, N, device, frames_of[rank], B, NB, scr, None, False,
                                  ring=ring, root_mode=args.gather_root)
                pl.run(0, warm)
                comm.barrier()
                counts = pl.run(warm, steps)
                comm.barrier()
                stream.synchronize()
                out = {"counts": counts}
                if rank == root_last:        # the root holds every rank's records of the LAST batch, rank after rank
                    total = int(np.maximum(counts, 0).sum())
                    out["records"] = pl.recv[: total * 576].cpu().numpy().view(capi.POINT_DTYPE).copy()
                    out["last_batch"] = (warm + steps - 1) % NB
                if not args.no_match:
                    ops = LoopbackMatchOps(torch, capi, ctx, comm, device, rank, N, shared)
                    out["match"] = matcher_leg(ops, rank, N, nm, msteps=1, validate_rows=50, l2=False,
                                               point_dtype=capi.POINT_DTYPE, result_dtype=capi.RESULT_DTYPE, oracle=orc)
                res[rank] = out
                comm.close()
                ctx.close()
        except Exception as e:               # noqa: BLE001 — reported by the main thread
            import traceback
            errs.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
            try:
                shared["bar"].abort()
            except Exception:
                pass

    ts = [threading.Thread(target=body, args=(r,)) for r in range(N)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise RuntimeError("emulated ranks failed:\n" + "\n".join(errs))
    lw.close()
    counts = res[0]["counts"]
    for r in range(1, N):
        assert np.array_equal(res[r]["counts"], counts), "ranks disagree on the gathered counts"
    validated = None
    if orc is not None:                      # every rank's frames of the last batch, as gathered on rank 0, vs the oracle
        from util import compare_points
        recs, off, validated = res[root_last]["records"], 0, 0
        b0 = res[root_last]["last_batch"] * B
        for r in range(N):
            host = frames_of[r][b0:b0 + B].cpu().numpy()
            ref, nref, _ = orc.extract_batch(host, NUM_OCTAVES, INIT_BLUR, THRESH, max_pts=8192)
            for f in range(B):
                if int(nref[f]) != int(counts[r, f]):
                    raise RuntimeError("emulated rank %d frame %d: %d records, oracle %d" % (r, f, counts[r, f], nref[f]))
                compare_points(ref[f, :nref[f]], recs[off:off + nref[f]], "emulate_r%d_f%d" % (r, f))
                off += int(nref[f])
                validated += 1
        assert off == len(recs)
    m = res[0].get("match")
    out = {"mode": "emulate-ranks", "functional_only": True, "emulated_ranks": N, "n_gpus": 1, "frames_per_rank": B,
           "steps": steps, "warmup": warm, "batches_in_flight": max(1, args.batches_in_flight),
           "transport": "misift_loopback_world: N communicators / contexts / host threads on ONE device, device-to-device "
                        "copies through a shared rendezvous (no RCCL); same StepPipeline and matcher_leg as --gpus N",
           "gather_root": args.gather_root, "root_of_last_step": root_last,
           "gathered_frames_last_step": int(N * B), "validated_frames": validated,
           "keypoints_per_frame": round(float(np.mean(counts)), 1),
           "match": None if m is None else {"n1": m["n1"], "n2": m["n2"], "split": m["split"],
                                            "validated_rows_per_rank": m.get("validated_rows")},
           "note": "functional run of the N-rank code path on one GPU; carries no rate on purpose — it is not a scaling number"}
    C.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


# ------------------------------------------------------------------------------------------------ main
# ------------------------------------------------------------------------------------------------ self-spawned ranks
def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def visible_gpus():
    """Devices this process could use (no context is created)."""
    try:
        import torch
        return int(torch.cuda.device_count()) if torch.cuda.is_available() else 0
    except Exception:
        return 0


def spawn_ranks(ngpus, argv, need_gpus=True):
    """`python bench.py --gpus N` without a launcher: become the launcher.  N children of this same command, one per
    visible device, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment (exactly what
    torch.distributed.run sets), stdout / stderr inherited — rank 0's JSON line is the last line of stdout.  Fails LOUDLY
    (rc != 0, nothing measured) when fewer than N devices are visible: a run that silently measured one GPU would be
    recorded as an N-GPU number.  Returns the exit code."""
    import subprocess
    if need_gpus:
        have = visible_gpus()
        if have < ngpus:
            print("bench.py: --gpus %d but only %d GPU(s) visible to this process: refusing to run (an N-GPU number needs N "
                  "devices; use --emulate-ranks N for the functional single-GPU rehearsal)" % (ngpus, have), file=sys.stderr)
            return 3
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(ngpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                "BENCH_SELF_SPAWNED": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: the only kind the host driver supports
    procs = []
    for r in range(ngpus):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for pr in list(pending):
                code = pr.poll()
                if code is None:
                    continue
                pending.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in pending:              # one rank failed: the others would wait for it forever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    if rc:
        print("bench.py: a rank exited with code %d; no result" % rc, file=sys.stderr)
    return rc


# ------------------------------------------------------------------------------------------------ watchdog
class Watchdog:
    """Every rank stamps its progress (`stage(name)`); a daemon thread ends the process with rc 6 — after printing the
    rank and the stage it was stuck in — when nothing has been stamped for `limit_s` seconds.  Under a launcher (the
    driver's torch.distributed.run, or spawn_ranks below) one rank's exit takes the others down, and every stuck rank
    prints its own stage: a hang in rendezvous / ncclCommInitRank / a collective costs two minutes and says where, not the
    launcher's whole time-out.  BENCH_WATCHDOG_S sets the limit (default 120; 0 disables)."""

    def __init__(self, rank, limit_s):
        import threading
        self.rank, self.limit, self.name, self.t = rank, float(limit_s), "start", time.monotonic()
        self.history = []
        # every rank keeps its current stage in a file of a directory all ranks of the job share (keyed by MASTER_PORT):
        # whichever watchdog fires first prints the stage of EVERY rank, including the ones stuck inside a C call
        self.dir = None
        if os.environ.get("MASTER_PORT") and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            self.dir = os.path.join(tempfile.gettempdir(), "bench_stages_%s" % os.environ["MASTER_PORT"])
            try:
                os.makedirs(self.dir, exist_ok=True)
            except OSError:
                self.dir = None
        self._publish()
        if self.limit > 0:
            th = threading.Thread(target=self._watch, daemon=True)
            th.start()

    def _publish(self):
        if self.dir:
            try:
                with open(os.path.join(self.dir, "rank_%d" % self.rank), "w") as f:
                    f.write("%s\t%.3f\n" % (self.name, time.time()))
            except OSError:
                pass

    def stage(self, name):
        self.history.append((self.name, round(time.monotonic() - self.t, 3)))
        self.name, self.t = name, time.monotonic()
        self._publish()

    def _report(self, idle):
        lines = ["bench.py watchdog: rank %d silent for %.0f s in stage '%s' (stages before it: %s): giving up, rc 6"
                 % (self.rank, idle, self.name, ", ".join(n for n, _ in self.history[-6:]))]
        if self.dir:
            try:
                for fn in sorted(os.listdir(self.dir)):
                    name, ts = open(os.path.join(self.dir, fn)).read().strip().split("\t")
                    lines.append("bench.py watchdog:   %s: in stage '%s' for %.0f s" % (fn.replace("_", " "), name, time.time() - float(ts)))
            except Exception:                       # noqa: BLE001 — diagnostics only
                pass
        sys.stderr.write("\n".join(lines) + "\n")
        sys.stderr.flush()

    def _watch(self):
        while True:
            time.sleep(min(1.0, self.limit / 4.0))
            idle = time.monotonic() - self.t
            if idle > self.limit:
                try:
                    self._report(idle)
                finally:
                    os._exit(6)


def spawn_check():
    """`--spawn-check` (tests/test_bench_cpu.py): what a spawned rank sees, over gloo — no GPU needed."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    wd = Watchdog(rank, float(os.environ.get("BENCH_WATCHDOG_S", "120")))
    wd.stage("init_process_group")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    if os.environ.get("BENCH_SPAWN_CHECK_FAIL_RANK") == str(rank):
        sys.exit(7)                                            # the launcher must report this and stop the others
    if os.environ.get("BENCH_SPAWN_CHECK_HANG_RANK") == str(rank):
        wd.stage("pretend_collective")
        time.sleep(3600)                                       # the watchdog must end this rank, the launcher the others
    wd.stage("barrier")
    dist.barrier()
    if rank == 0:
        print(json.dumps({"spawn_check": True, "world": world, "sum_of_ranks_plus_1": int(t.item()),
                          "local_rank": int(os.environ["LOCAL_RANK"]), "master_addr": os.environ["MASTER_ADDR"]}), flush=True)
    dist.destroy_process_group()
    return 0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames-per-gpu", type=int, default=64)
    ap.add_argument("--batches", type=int, default=NUM_BATCHES, help="distinct batches per rank to rotate through")
    ap.add_argument("--contexts", type=int, default=1,
                    help="contexts (own stream, staging and scratch arena each) the steps rotate over = batches in flight on the "
                         "GPU.  4 measures ~8 %% more frames/s (profiles/r02_bench_contexts4.json) but every kernel's duration is "
                         "then stretched by its neighbours, so the per-kernel roofline is quoted on the default, 1")
    ap.add_argument("--batches-in-flight", type=int, default=0,
                    help="pipelines INSIDE the context (misift_ctx_set_batches_in_flight): consecutive batches overlap on the GPU "
                         "behind ONE context.  0 = auto = 2 (r05, three alternating runs each on one box, 100 timed steps: K = 2 "
                         "58.6 / 57.3 / 58.4 k frames/s, K = 4 56.8 / 58.0 / 55.6 k, K = 3 53.1 / 55.6 / 52.9 k, K = 1 51.2 k; with "
                         "4-5 kernels of different batches resident every small dependent kernel of a batch waits for a slot "
                         "behind the others' big ones — tools/overlap_report.py, profiles/r05_overlap_K4.txt).  The per-kernel "
                         "roofline durations always come from a K = 1 child run")
    ap.add_argument("--match-n", type=int, default=100000)
    ap.add_argument("--no-match", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg AND the oracle self-validation")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU leg (0 = one per core, capped by RAM)")
    ap.add_argument("--unfused", action="store_true", help="separate laplace/detect kernels (DoG planes in HBM)")
    ap.add_argument("--selftest-dist", action="store_true", help="single GPU: run the RCCL gather path with a 1-rank communicator")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-frame latency side measurement")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive (H2D + extract + D2H) side measurement")
    ap.add_argument("--no-skewed", action="store_true", help="skip the skewed-batch side measurement (8 busy frames of 64)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 PMC traffic passes")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="run the N-rank driver loops (gather of SiftData, sharded matcher) end to end on ONE GPU through the "
                         "loopback transport; functional only, prints no rate")
    ap.add_argument("--spawn-check", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--gather-root", choices=("rotate", "fixed"), default="rotate",
                    help="N > 1: root of the SiftData gather — 'rotate': step %% N (every rank's xGMI ingress takes its turn: "
                         "one fixed root needs 7 x 77 MB per 1.17 ms step = 460 GB/s into ONE GPU, DESIGN.md section 6); "
                         "'fixed': rank 0 (BASELINE config 4 as written).  The line reports the other mode in `gather`")
    ap.add_argument("--preroll-steps", type=int, default=60,
                    help="untimed steps before the warm-up: the shader clock needs ~40 ms of this load to settle (DESIGN.md "
                         "section 5), so that a 20-step timed window measures the same steady state as a 100-step one; the line "
                         "carries the same window WITHOUT the pre-roll beside it (`no_preroll`)")
    return ap.parse_args()


class Bench:
    """One rank's run of the benchmark, leg by leg (main() below is the sequence).  State shared between legs lives on the
    object; every leg that is only a side measurement says so in what it reports and never touches `value`."""

    # ------------------------------------------------------------------ setup
    def __init__(self, args):
        import numpy as np
        import torch                     # first: libmisift.so then binds to torch's HIP runtime
        import torch.distributed as dist
        from cudasift_amd import capi
        self.args, self.np, self.torch, self.dist, self.capi = args, np, torch, dist, capi
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.wd = Watchdog(self.rank, float(os.environ.get("BENCH_WATCHDOG_S", "120")))
        rank, world, local_rank = self.rank, self.world, self.local_rank
        if world != args.gpus:
            # a line that says n_gpus = WORLD_SIZE while the command said --gpus N would be mis-recorded: refuse
            if rank == 0:
                print("bench.py: --gpus %d but WORLD_SIZE=%s: refusing to run (launch N ranks for --gpus N, or run the plain "
                      "command and let bench.py spawn them)" % (args.gpus, os.environ.get("WORLD_SIZE")), file=sys.stderr)
            sys.exit(4)
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        if torch.cuda.device_count() <= local_rank:
            print("bench.py: rank %d has no device %d (%d visible)" % (rank, local_rank, torch.cuda.device_count()), file=sys.stderr)
            sys.exit(3)
        torch.cuda.set_device(local_rank)
        self.device = device = torch.device("cuda", local_rank)
        if world > 1:
            os.environ.setdefault("NCCL_DEBUG", "WARN")          # RCCL's own complaints go to stderr with the rank's stage
            self.wd.stage("init_process_group(nccl)")
            dist.init_process_group("nccl", rank=rank, world_size=world)       # rendezvous, barrier, time reduction only

        self.B, self.NB = args.frames_per_gpu, max(1, args.batches)
        self.stream = stream = torch.cuda.current_stream()
        # One context = one in-order pipeline (stream, counters, candidate lists, detection staging).  The steps rotate over
        # NCTX of them, each with its own scratch arena, so that NCTX batches are in flight: the HBM-bound front end of one
        # batch, the VALU-bound kernels of another and the launch tails of a third share the GPU.
        self.NCTX = NCTX = 1 if args.unfused else max(1, args.contexts)      # the dense path shares one record array
        self.wd.stage("misift_ctx_create")
        self.ctx = ctx = capi.Context(local_rank, stream.cuda_stream)
        self.ctx_streams = [stream] + [torch.cuda.Stream(device=device) for _ in range(NCTX - 1)]
        self.ctxs = [ctx] + [capi.Context(local_rank, st.cuda_stream) for st in self.ctx_streams[1:]]
        self.RING = RING = 1 if args.unfused else max(1, args.batches_in_flight)
        for c in self.ctxs:
            c.set_options(quiet=1, fused=0 if args.unfused else 1)
            if RING > 1:
                c.set_batches_in_flight(RING)

        # the data-path communicator lives behind the C-ABI (RCCL over xGMI): rank 0 makes the id, torch ships it
        self.comm = None
        if world > 1 or args.selftest_dist:
            self.wd.stage("misift_comm_create (ncclCommInitRank)")
            idt = torch.zeros((capi.COMM_ID_BYTES,), dtype=torch.uint8, device=device)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(idt, 0)
            self.comm = capi.Comm(ctx, world, rank, bytes(idt.cpu().numpy().tobytes()))
            if self.comm.size != world or self.comm.rank != rank:
                print("bench.py: communicator reports rank %d of %d, expected %d of %d" % (self.comm.rank, self.comm.size, rank, world),
                      file=sys.stderr)
                sys.exit(5)
        self.orc = None
        if not args.no_cpu and rank == 0:
            from oracle import pyoracle as orc
            self.orc = orc
        self.cfg2 = self.match = self.latency = self.pcie = self.skewed = self.gather_info = self.step_ms = None
        self.cpu = self.validated = self.no_preroll = None

    def barrier(self):
        if self.comm is not None:
            self.comm.barrier()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def packed_async(self, c, frames_ptr, scratch_ptr, cnt, packed):
        capi, B = self.capi, self.B
        capi.check(capi.lib().misift_extract_batch_packed_async(
            c.h, frames_ptr, B, H * W, W, H, W, NUM_OCTAVES, INIT_BLUR, THRESH, 0.0, scratch_ptr, None, self.pl.REC_CAP,
            cnt.data_ptr(), cnt[B:].data_ptr(), packed.data_ptr()), "misift_extract_batch_packed_async")

    # ------------------------------------------------------------------ matcher (BASELINE config 5)
    def leg_matcher(self):
        """Its own timed region (barrier on both sides, matcher_leg); the order of the two legs is free and does not change
        either number (measured both ways)."""
        args, capi, torch = self.args, self.capi, self.torch
        if args.no_match:
            return
        self.wd.stage("matcher 100k x 100k")
        ops = CabiMatchOps(torch, capi, self.ctx, self.comm if self.world > 1 else None, self.device, self.rank, self.world)
        self.match = matcher_leg(ops, self.rank, self.world, args.match_n, msteps=3, validate_rows=100, l2=False,
                                 point_dtype=capi.POINT_DTYPE, result_dtype=capi.RESULT_DTYPE, oracle=self.orc)
        self.wd.stage("matcher, unit-L2 variant")
        m2 = matcher_leg(ops, self.rank, self.world, args.match_n, msteps=2, validate_rows=0, l2=True,
                         point_dtype=capi.POINT_DTYPE, result_dtype=capi.RESULT_DTYPE, oracle=None)
        self.match["unit_l2_variant"] = {"value": m2["value"], "ms": m2["ms"], "frac": m2["roofline"]["frac"]}
        # the shape ONE rank of BASELINE config 5 runs: a 12 500-row block against all 100 k columns
        if self.world == 1 and args.match_n >= 100000:
            self.wd.stage("matcher, one rank's shard")
            self.match["rank_shard_12500x100000"] = self.matcher_shard(ops, 12512, args.match_n // 32 * 32)
        del ops
        torch.cuda.empty_cache()

    def matcher_shard(self, ops, n1, n2):
        from synth import descriptors_to_points, synth_descriptors
        capi = self.capi
        a = ops.to_device(descriptors_to_points(synth_descriptors(n1, 12345, False), capi.POINT_DTYPE))
        b = ops.to_device(descriptors_to_points(synth_descriptors(n2, 12346, False), capi.POINT_DTYPE))

        def step():
            capi.check(capi.lib().misift_match_rows(self.ctx.h, a.data_ptr(), 0, n1, b.data_ptr(), n2), "misift_match_rows")
        step()
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        self.torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        kms = ops.kernel_ms(step)
        flops = 2.0 * 128 * n1 * n2
        return {"n1": n1, "n2": n2, "ms": round(dt * 1e3, 3), "kernel_ms": round(kms, 3),
                "frac": round(flops / (kms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4) if kms > 0 else None,
                "note": "the row block one rank of the 8-GPU split sweeps (12 512 = 12 500 rounded to whole 32-row tiles); "
                        "kernel_ms from the library's event pair around match_kernel"}

    # ------------------------------------------------------------------ inputs + the step loop
    def setup_inputs(self):
        """Inputs resident in HBM before the timed region: NB batches of B distinct frames per rank."""
        torch, capi, args, device = self.torch, self.capi, self.args, self.device
        B, NB = self.B, self.NB
        self.wd.stage("allocate inputs")
        self.frames = torch.empty((NB * B, H, W), dtype=torch.float32, device=device)
        S = capi.scratch_floats(W, H, NUM_OCTAVES, False)
        self.scratches = [torch.empty((B * S,), dtype=torch.float32, device=device) for _ in range(self.NCTX * self.RING)]
        self.scratch = self.scratches[0]
        self.pts = torch.zeros((B * MAX_PTS * 576,), dtype=torch.uint8, device=device) if args.unfused else None
        torch.cuda.synchronize()
        # Software-pipelined step loop, the same for every N (class StepPipeline above; `--emulate-ranks` drives N of them
        # from N host threads of this process over the loopback transport)
        self.pl = StepPipeline(torch, capi, self.ctxs, self.ctx_streams, self.comm, self.rank, self.world, device, self.frames, B, NB,
                               self.scratches, self.pts, args.unfused, ring=self.RING,
                               root_mode=args.gather_root if self.world > 1 else "fixed")
        # The synthetic frames are generated last, after every allocation.
        self.wd.stage("generate frames")
        gen_frames_torch(torch, NB * B, self.rank * NB * B, device, out=self.frames)
        torch.cuda.synchronize()

    def timed_window(self, k0, warmup, steps):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; returns
        (seconds over all ranks' max, this rank's own seconds, counts of the last completed step, next k)."""
        pl, torch = self.pl, self.torch
        if warmup > 0:
            pl.run(k0, warmup)
        self.barrier()
        pl.step_ev.clear()
        pl.gather_s, pl.gather_calls, pl.wire_bytes = 0.0, 0, 0
        t0 = time.perf_counter()
        counts = pl.run(k0 + warmup, steps)
        own_dt = time.perf_counter() - t0          # this rank's own loop (before the closing barrier): per-rank rate
        self.barrier()
        dt = time.perf_counter() - t0
        if self.world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, own_dt, counts, k0 + warmup + steps

    def leg_timed_loop(self):
        """The contract's region.  A short run starts slow whatever precedes it (idle, the frame generator or the MFMA-bound
        matcher): per-launch durations (a 30-step run under rocprofv3 --kernel-trace, r03) show the VALU-bound kernels
        speeding up over the first ~30 steps (dog_scan 0.69 -> 0.57 ms) while the HBM-bound lowpass_down is flat — the shader
        clock ramps up over ~40 ms of this load.  So: (1) the driver's window as it stands, W + K steps from cold —
        reported as `no_preroll`; (2) an untimed pre-roll of the same steps, then W + K again — `value`."""
        args, np, torch = self.args, self.np, self.torch
        B, world = self.B, self.world
        PRE = self.PRE = max(0, args.preroll_steps)
        k = 0
        if PRE > 0:
            self.wd.stage("timed window without pre-roll")
            dt0, _, _, k = self.timed_window(0, args.warmup, args.steps)
            self.no_preroll = {"value": round(world * B * args.steps / dt0, 1), "ms_per_step": round(1e3 * dt0 / args.steps, 4),
                               "note": "the same W warm-up + K timed steps measured FIRST, straight after the frame generator, "
                                       "without the untimed pre-roll (the shader clock is still ramping: DESIGN.md section 5)"}
            self.wd.stage("pre-roll")
            self.pl.run(k, PRE)
            k += PRE
        self.wd.stage("timed window")
        dt, own_dt, all_counts, k_end = self.timed_window(k, args.warmup, args.steps)
        self.dt, self.last_k = dt, k_end - 1
        pl = self.pl
        if world > 1:
            own = torch.tensor([own_dt, pl.gather_s / max(1, args.steps), float(pl.wire_bytes) / max(1, args.steps)],
                               dtype=torch.float64, device=self.device)
            allr = [torch.zeros_like(own) for _ in range(world)]
            self.dist.all_gather(allr, own)
            allr = [t.cpu().numpy() for t in allr]
            self.gather_info = {
                "root": args.gather_root, "rccl_ranks": int(self.comm.size),
                "per_rank_frames_per_s": [round(B * args.steps / float(a[0]), 1) for a in allr],
                "gather_ms_per_step": round(1e3 * max(float(a[1]) for a in allr), 4),
                "gather_ms_per_step_note": "host time inside misift_gather_complete per step, max over ranks (the gather of "
                                           "batch k-LAG runs on the communicator's stream under the extraction of the "
                                           "batches behind it; what is NOT hidden shows up in ms_per_step)",
                "wire_MB_per_step": round(float(allr[0][2]) / 1e6, 3),
                "wire_note": "valid 576-byte records of the non-root ranks into the root + the per-frame counts "
                             "all-gather, per step; at ms_per_step this is %.1f GB/s summed over the links"
                             % (float(allr[0][2]) / 1e9 / (dt / args.steps))}
        self.kp_per_frame = float(np.mean(all_counts[self.rank if all_counts.shape[0] > 1 else 0]))
        self.ms_per_step = 1e3 * dt / args.steps
        self.fps = world * B * args.steps / dt
        if pl.trace_host and self.rank == 0:
            n = args.steps + args.warmup
            print("host time per step: enqueue %.3f ms, complete %.3f ms (incl. warm-up); step %.3f ms"
                  % (1e3 * pl.host_t["enqueue"] / n, 1e3 * pl.host_t["complete"] / n, self.ms_per_step), file=sys.stderr)
        # distribution of the pipelined loop's steps on the GPU timeline (median + p10/p90)
        NI = self.NCTX * self.RING
        if self.rank == 0 and len(pl.step_ev) >= NI + 2:
            # step k and step k+NI end on the same stream: that interval / NI is the loop's period seen from one context
            # (with a ring: the completion markers of batches k and k + batches-in-flight)
            ev = pl.step_ev
            d = np.array([ev[i].elapsed_time(ev[i + NI]) / NI for i in range(len(ev) - NI)])
            if os.environ.get("BENCH_STEP_DUMP"):
                print("step intervals (ms):", " ".join("%.3f" % v for v in d), file=sys.stderr)
            self.step_ms = {"p10": round(float(np.percentile(d, 10)), 4), "p50": round(float(np.percentile(d, 50)), 4),
                            "p90": round(float(np.percentile(d, 90)), 4), "samples": int(len(d)),
                            "note": "HIP-event interval between the ends of steps k and k+contexts (same stream) / contexts, "
                                    "over the timed, pipelined loop"}
        # the timed loop's LAST step, kept for the self-validation
        capi = self.capi
        last_slot, self.last_b0 = self.last_k % pl.NSLOT, (self.last_k % self.NB) * B
        last_cnt = pl.cnts[last_slot].cpu().numpy().copy()
        self.last_counts, self.last_offs = last_cnt[:B], last_cnt[B:]
        self.last_recs = pl.packed[last_slot][: int(self.last_offs[B]) * 576].cpu().numpy().view(capi.POINT_DTYPE).copy()

    def leg_other_root(self):
        """N > 1: the same loop with the OTHER gather root (reported beside the headline, never `value`)."""
        args, pl, torch = self.args, self.pl, self.torch
        if self.world == 1:
            return
        self.wd.stage("other gather root")
        other = "fixed" if args.gather_root == "rotate" else "rotate"
        n2 = max(pl.LAG + 2, min(args.steps, 40))
        pl.root_mode = other
        self.barrier()
        t1 = time.perf_counter()
        pl.run(self.last_k + 1, n2)
        self.barrier()
        dt2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(dt2, op=self.dist.ReduceOp.MAX)
        pl.root_mode = args.gather_root
        self.last_k += n2
        self.gather_info["other_root"] = {"root": other, "steps": n2, "value": round(self.world * self.B * n2 / float(dt2.item()), 1),
                                          "ms_per_step": round(1e3 * float(dt2.item()) / n2, 4)}
        pl.step_ev.clear()

    # ------------------------------------------------------------------ per-kernel durations
    def leg_kernel_events(self):
        """Per-kernel durations (HIP events on the launch stream) in the timed configuration: the same pipelined rotation
        (nothing synchronises between the steps, so the GPU stays at the clocks of the timed loop), every context recording
        its own kernels; with more than one context also every kernel alone on the GPU."""
        torch, capi, pl = self.torch, self.capi, self.pl
        B, NB, NCTX = self.B, self.NB, self.NCTX
        self.wd.stage("per-kernel event durations")
        if self.pts is None:
            self.pts = torch.zeros((B * MAX_PTS * 576,), dtype=torch.uint8, device=self.device)
        psteps = 10 * NCTX
        for c in self.ctxs:
            c.profile_reset()
            c.profile_enable(True)
        for k in range(psteps):
            pl.enqueue(self.last_k + 1 + k)
        torch.cuda.synchronize()
        prof = {}
        for c in self.ctxs:
            for name, p in c.profile_read().items():
                e = prof.setdefault(name, {"total_ms": 0.0, "calls": 0})
                e["total_ms"] += p["total_ms"]
                e["calls"] += p["calls"]
            c.profile_enable(False)
        pl.step_ev.clear()
        prof_alone, asteps = {}, 5
        if NCTX > 1:
            counts = (C.c_int * B)()
            self.ctx.profile_reset()
            self.ctx.profile_enable(True)
            for i in range(asteps):
                capi.check(capi.lib().misift_extract_batch(self.ctx.h, self.frames[(i % NB) * B].data_ptr(), B, H * W, W, H, W,
                                                           NUM_OCTAVES, INIT_BLUR, THRESH, 0.0, self.scratch.data_ptr(),
                                                           self.pts.data_ptr(), MAX_PTS, counts), "misift_extract_batch")
            prof_alone = self.ctx.profile_read()
            self.ctx.profile_enable(False)
        self.alg = alg = algorithmic_bytes_per_frame()
        self.N = N = octave_pixels(W, H, NUM_OCTAVES)
        if "lowpass_down" in prof:
            # fused prefilter + first ScaleDown: its algorithmic bytes are the sum of the two reference kernels' figures
            # (LowPass 8*N0 + ScaleDown_1 4*N0 + 4*N1); the remaining ScaleDown launches cover levels 2..
            alg["lowpass_down"] = 8 * N[0] + 4 * N[0] + 4 * N[1]
            alg["scaledown"] = sum(4 * N[i] + 4 * N[i + 1] for i in range(1, NUM_OCTAVES - 1))
        self.kernels = kernels = {}
        for name, p in prof.items():
            per_step_ms = p["total_ms"] / psteps
            e = {"ms_per_step": round(per_step_ms, 4), "launches_per_step": p["calls"] // psteps}
            if NCTX > 1 and name in prof_alone:
                e["alone_ms_per_step"] = round(prof_alone[name]["total_ms"] / asteps, 4)
            if name in alg:
                e["alg_GBps"] = round(alg[name] * B / (per_step_ms * 1e-3) / 1e9, 1)
            kernels[name] = e

    # ------------------------------------------------------------------ side measurements (never `value`)
    def leg_single_frame(self):
        """Single-frame latencies (BASELINE configs 2 and 3)."""
        args, np, capi, torch = self.args, self.np, self.capi, self.torch
        if not (self.rank == 0 and self.world == 1 and not args.no_latency):
            return
        self.wd.stage("single-frame latencies")
        ctx, frames, scratch, pts = self.ctx, self.frames, self.scratch, self.pts
        latency = {"note": "median wall time of one synchronous call, frame / records resident in HBM "
                           "(misift_extract incl. its count read-back and — the C-ABI's default since r05 — a full stream "
                           "synchronisation; misift_match of 2 x ~2000 features)"}
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
                ts.append(time.perf_counter() - t1)          # (the call itself returns with the results in place)
            torch.cuda.synchronize()
            latency["match_%dx%d_ms" % (npts, npts)] = round(1e3 * float(np.median(ts[10:])), 4)
        latency["chain_fallbacks"] = ctx.chain_fallbacks()     # 0 on a healthy device (bounded in-launch wait)

        # ... and the same calls made the way the reference's demo makes them: from C++, through the drop-in API
        # (build/single_call = tools/single_call.cpp: mainSift.cpp:58-81 on synthetic frames, 200 calls each) — no Python
        # or ctypes in the timed call.  `host=1` keeps the reference's host mirror of SiftData (ExtractSift then copies
        # the records back inside its timed region, like cudaSiftH.cu:139-140).
        exe = os.path.join(ROOT, "build", "single_call")
        if os.path.exists(exe):
            try:
                from synth import synth_frame
                dropin = {}
                with tempfile.TemporaryDirectory() as td:
                    for (lw, lh) in ((1920, 1080), (1280, 960)):
                        f0, f1 = os.path.join(td, "f0_%d.f32" % lw), os.path.join(td, "f1_%d.f32" % lw)
                        synth_frame(0, lw, lh).tofile(f0)
                        synth_frame(1, lw, lh).tofile(f1)
                        for host in (0, 1):
                            r = subprocess.run([exe, f0, f1, str(lw), str(lh), "200", "5", "3.0", str(host)], capture_output=True,
                                               text=True, timeout=300)
                            for ln in r.stdout.splitlines():
                                if ln.startswith("{"):
                                    e = json.loads(ln)
                                    kind = "extract" if e["what"].startswith("ExtractSift") else "match"
                                    key = "%s_%dx%d%s_ms" % (kind, lw, lh, "_with_host_mirror" if host else "")
                                    dropin[key] = e["p50_ms"]
                                    dropin[key.replace("_ms", "_points")] = e["points"]
                latency["cpp_caller_through_libcudasift"] = dropin
                latency["cpp_caller_note"] = ("p50 of 200 back-to-back calls from C++ through the drop-in API (tools/single_call.cpp; "
                                              "tests/synth.py frames 0 and 1), whose shim opts into the early return "
                                              "(misift_ctx_set_early_return); the figures above are the C-ABI's entry points called "
                                              "from Python through ctypes with the default full synchronisation")
            except Exception as e:          # noqa: BLE001 — a side measurement must not break the line
                latency["cpp_caller_through_libcudasift"] = "failed: %s" % e
        self.latency = latency

    def leg_skewed_batch(self):
        """A batch whose keypoints sit in 8 of its 64 frames (the other 56 are the same frames at 1/8 of the contrast: a few
        keypoints each).  The per-keypoint kernels deal their workgroups out in proportion to the frames' counts
        (MISIFT_BALANCE, default since r05); without that the busy frames' fixed share of the grid is the step's tail
        (r04: 1.8-2.1x).  Reported beside the uniform step, never `value`."""
        args, np, torch, pl = self.args, self.np, self.torch, self.pl
        if not (self.rank == 0 and self.world == 1 and not args.no_skewed and not args.unfused and self.B >= 16):
            return
        self.wd.stage("skewed batch")
        B = self.B
        nb = 2
        sk = torch.empty((nb * B, H, W), dtype=torch.float32, device=self.device)
        for b in range(nb):
            src = self.frames[b * B:(b + 1) * B]
            sk[b * B:(b + 1) * B] = 128.0 + (src - 128.0) * 0.125
            sk[b * B:b * B + B:8] = src[::8]                       # every 8th frame keeps its contrast: 8 busy frames of 64
        torch.cuda.synchronize()
        cnt, packed = pl.cnts[0], pl.packed[0]
        K = max(1, self.RING)
        n = 24

        def loop(nsteps):
            for i in range(nsteps):
                self.packed_async(self.ctx, sk[(i % nb) * B].data_ptr(), self.scratches[i % K].data_ptr(), cnt, packed)
        loop(6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(n)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        c = cnt[:B].cpu().numpy()
        busy = c[0::8]
        quiet = np.delete(c, np.arange(0, B, 8))
        self.skewed = {"ms_per_step": round(1e3 * dt, 4), "uniform_ms_per_step": round(self.ms_per_step, 4),
                       "ratio_to_uniform": round(1e3 * dt / self.ms_per_step, 3),
                       "keypoints_busy_frames": round(float(busy.mean()), 1), "keypoints_quiet_frames": round(float(quiet.mean()), 1),
                       "steps": n,
                       "note": "64 frames, 8 of them at full contrast, 56 at 1/8 (same images); %d batches in flight on the "
                               "headline's context; the step moves the same pixels as the uniform one and a quarter of its "
                               "keypoints — a ratio above 1 would be the load-balance cliff of r04's default" % K}

    def leg_pcie(self):
        """PCIe-inclusive side measurement: pinned host frames -> H2D -> extract -> D2H."""
        args, np, capi, torch = self.args, self.np, self.capi, self.torch
        if not (self.rank == 0 and self.world == 1 and not args.no_pcie):
            return
        self.wd.stage("PCIe-inclusive pipe")
        torch.cuda.synchronize()
        nb, nbatches = 16, 64
        pcie = {"batch_frames": nb, "batches": nbatches, "depth": 3,
                "note": "misift_pipe: pinned host frames uploaded, valid SiftPoint records packed and downloaded, "
                        "upload/compute/read-back overlapped; never `value`"}
        host_recs = capi.PinnedArray((nb * 4096,), capi.POINT_DTYPE)
        for key, dtp in (("frames_per_s_u8", np.uint8), ("frames_per_s_f32", np.float32)):
            src = capi.PinnedArray((nb, H, W), dtp)
            f = self.frames[:nb].round().clamp(0, 255)
            src.array[...] = f.cpu().numpy().astype(dtp)
            pipe = capi.Pipe(self.ctx, W, H, nb, src_u8=(dtp == np.uint8), num_octaves=NUM_OCTAVES, init_blur=INIT_BLUR,
                             thresh=THRESH, max_pts=MAX_PTS, depth=3)

            def prun(k):
                tot = 0
                for i in range(k):
                    if pipe.pending() == 3:
                        tot += pipe.collect(host_recs.ptr, nb * 4096)[1]
                    pipe.submit(src.ptr, nb)
                while pipe.pending():
                    tot += pipe.collect(host_recs.ptr, nb * 4096)[1]
                return tot
            prun(24)                                # warm-up: the GPU and the PCIe link have idled through the child passes
            best = None
            for _ in range(2):                      # short measurement on a shared host: best of two
                tp0 = time.perf_counter()
                tot = prun(nbatches)
                pdt = time.perf_counter() - tp0
                best = pdt if best is None else min(best, pdt)
                self.wd.stage("PCIe-inclusive pipe (%s)" % key)
            pcie[key] = round(nb * nbatches / best, 1)
            pcie["records_per_frame"] = round(tot / (nb * nbatches), 1)
            pipe.close()
            src.free()
        host_recs.free()
        self.pcie = pcie

    # ------------------------------------------------------------------ rocprofv3 child passes (rank 0, N = 1)
    def leg_counters_and_trace(self):
        """(the side measurements run BEFORE these: the child passes leave the GPU in a slower state for a while — the
        host-fed pipe read 17 k instead of 24 k frames/s right after them)"""
        args, torch, B = self.args, self.torch, self.B
        kernels, alg = self.kernels, self.alg
        one_gpu = self.rank == 0 and self.world == 1
        self.pmc, self.pmc_note = None, "not collected (N > 1 or --no-pmc)"
        self.calib = {"gather_read_factor": 2.0,
                      "source": "uncalibrated: the guide's x2 for wide coalesced reads applied to the gathers too"}
        cj = os.path.join(ROOT, "profiles", "r02_pmc_calibration.json")
        if os.path.exists(cj):
            try:
                cc = json.load(open(cj))
                self.calib = {"gather_read_factor": float(cc["gather_read_factor"]),
                              "source": "profiles/r02_pmc_calibration.json (tools/pmc_calib)"}
            except Exception:
                pass
        if one_gpu and not args.no_pmc:
            torch.cuda.synchronize()
            self.wd.stage("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes")
            self.pmc, self.pmc_note = collect_pmc(B, self.calib["gather_read_factor"])
            if self.pmc is not None:
                self.pmc_note = ("collected live in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) "
                                 "over 4 steps of the timed entry point; read = 2 x FETCH_SIZE KB (gfx950 correction for wide "
                                 "coalesced reads; gather kernels x%.2f, see pmc_calibration), write = WRITE_SIZE KB"
                                 % self.calib["gather_read_factor"])
                for k, e in self.pmc.items():
                    if k in kernels:
                        kernels[k]["traffic_MB_per_frame"] = round((e["read"] + e["write"]) / B / 1e6, 3)
                        kernels[k]["read_MB_per_frame"] = round(e["read"] / B / 1e6, 3)
        self.sq, self.sq_note = None, "not collected (N > 1 or --no-pmc)"
        if one_gpu and not args.no_pmc and not args.unfused:
            self.wd.stage("rocprofv3 --pmc SQ child pass")
            self.sq, self.sq_note = collect_sq(B)
        # Launch durations as rocprofv3 reports them (one --kernel-trace pass over a pipelined child run of the same entry
        # point).  The HIP-event pairs bracket every launch with two stream markers and read 6-17 % longer than the
        # dispatch; the roofline uses the dispatch durations (what `rocprofv3 --stats` of this command shows), the event
        # figure stays in the table as hip_event_ms_per_step.
        self.trace, self.trace_note = None, "not collected (N > 1, --no-pmc or --contexts > 1)"
        if one_gpu and self.NCTX == 1 and not args.no_pmc and not args.unfused:
            torch.cuda.synchronize()
            self.wd.stage("rocprofv3 --kernel-trace child pass")
            # the child run mirrors the timed region: the same number of warm-up and timed steps (the shader clock ramps up
            # over the first ~30 steps of this load, so a fixed short child run would quote longer durations)
            self.tsteps = max(10, min(args.steps, 200)) + max(0, min(args.warmup, 50))
            self.tskip = max(1, min(args.warmup, 50))
            self.trace, self.trace_note = collect_trace(B, steps=self.tsteps, skip=self.tskip)
            if self.trace is not None:
                self.trace_note = ("rocprofv3 --kernel-trace over a one-batch-at-a-time child run of %d back-to-back steps of the "
                                   "timed entry point (first %d dropped: the timed region's own warm-up and length), collected "
                                   "live in this run" % (self.tsteps, self.tskip))
                for k, e in self.trace.items():
                    if k in kernels:
                        kernels[k]["hip_event_ms_per_step"] = kernels[k]["ms_per_step"]
                        kernels[k]["ms_per_step"] = round(e["ms_per_step"], 4)
                        if e["launches_per_step"] > 1:
                            kernels[k]["union_ms_per_step"] = round(e["union_ms_per_step"], 4)
                        if k in alg:
                            kernels[k]["alg_GBps"] = round(alg[k] * B / (e["ms_per_step"] * 1e-3) / 1e9, 1)
        if self.trace is None and self.rank == 0 and not args.unfused and (self.RING > 1 or self.NCTX > 1):
            self.one_batch_at_a_time_events()

    def one_batch_at_a_time_events(self):
        """No trace (N > 1, --no-pmc, rocprofv3 unavailable): the event durations were taken with RING batches sharing the
        GPU and every launch stretched ~2x by its neighbours — the roofline would describe the mix, not the kernel.  Time
        the same entry point one batch at a time instead (a fresh in-order context, steps queued back to back, the
        library's own event pairs: 6-17 % above the dispatch durations, stated in duration_source)."""
        torch, capi, pl, kernels, alg, B, NB = self.torch, self.capi, self.pl, self.kernels, self.alg, self.B, self.NB
        self.wd.stage("one-batch-at-a-time event durations")
        try:
            ca = capi.Context(self.local_rank, self.stream.cuda_stream)
            ca.set_options(quiet=1)
            na = 12
            for i in range(na + 2):
                if i == 2:                        # two untimed steps first
                    torch.cuda.synchronize()
                    ca.profile_enable(True)
                self.packed_async(ca, self.frames[(i % NB) * B].data_ptr(), self.scratch.data_ptr(), pl.cnts[0], pl.packed[0])
            torch.cuda.synchronize()
            pa = ca.profile_read()
            ca.close()
            dom = "dog_scan"
            if dom in pa and pa[dom]["calls"] >= na:
                for k, e in pa.items():
                    if k in kernels and e["calls"] >= na:
                        kernels[k]["in_flight_ms_per_step"] = kernels[k]["ms_per_step"]
                        kernels[k]["ms_per_step"] = round(e["total_ms"] / na, 4)
                        kernels[k]["launches_per_step"] = e["calls"] // na
                        if k in alg:
                            kernels[k]["alg_GBps"] = round(alg[k] * B / (e["total_ms"] / na * 1e-3) / 1e9, 1)
                self.trace_note = ("HIP-event pairs of the library around every launch over %d back-to-back steps of the timed "
                                   "entry point on a fresh in-order context, one batch at a time (in_flight_ms_per_step: the same "
                                   "with %d batches sharing the GPU); rocprofv3 kernel trace %s"
                                   % (na, self.NCTX * self.RING, self.trace_note))
        except Exception as e:                               # noqa: BLE001 — the bench line must still come out
            self.trace_note = "%s; one-batch-at-a-time fallback failed: %r" % (self.trace_note, e)

    # ------------------------------------------------------------------ roofline
    def single_launch_scan(self, flops_step):
        """dog_scan timed as ONE launch (all levels, MISIFT_SPLIT_TAIL=0 context): what a stand-alone rocprof of the kernel
        shows — no overlap with the coarse ScaleDowns, so an upper bound of what the step sees."""
        torch, capi, pl, B, NB = self.torch, self.capi, self.pl, self.B, self.NB
        self.wd.stage("single-launch scan")
        c1 = capi.Context(self.local_rank, self.stream.cuda_stream)
        c1.set_knob("split_tail", 0)
        c1.set_options(quiet=1)
        c1.profile_enable(True)
        n1 = 8
        for i in range(n1):                       # queued back to back like the timed loop (no host sync in between)
            self.packed_async(c1, self.frames[(i % NB) * B].data_ptr(), self.scratch.data_ptr(), pl.cnts[0], pl.packed[0])
        torch.cuda.synchronize()
        p1 = c1.profile_read()
        c1.close()
        if "dog_scan" not in p1 or p1["dog_scan"]["calls"] != n1:
            return None
        ms1 = p1["dog_scan"]["total_ms"] / n1
        if self.trace is not None:                 # the same from dispatch durations (rocprofv3 --kernel-trace, MISIFT_SPLIT_TAIL=0 child)
            self.wd.stage("single-launch scan, kernel-trace child pass")
            t1, _ = collect_trace(B, steps=self.tsteps, skip=self.tskip, extra_env={"MISIFT_SPLIT_TAIL": "0", "MISIFT_TUNABLES": "1"})
            if t1 and "dog_scan" in t1 and t1["dog_scan"]["launches_per_step"] == 1:
                ms1 = t1["dog_scan"]["ms_per_step"]
        a1 = flops_step / (ms1 * 1e-3) / 1e12
        return {"ms": round(ms1, 4), "achieved": round(a1, 2), "frac": round(min(a1 / VALU_F32_PEAK_TF, 1.0), 4),
                "note": "all pyramid levels in one launch on one stream (no overlap with the coarse ScaleDowns)"}

    def leg_roofline(self):
        """The dominant kernel against the roof that bounds it, the HBM answer BASELINE's metric asks for, and the issue
        budget of the whole step.  ONE number means ONE thing:
          roofline.frac           dog_scan's flops over the wall time during which a dog_scan launch was RUNNING (the union
                                  of its two concurrent launches, from the rocprofv3 start / end stamps) / the fp32 vector peak
          roofline.summed         the same over the SUM of the two launch durations (double-counts the overlap: a lower bound)
          roofline.single_launch  the same with all levels in one launch (no overlap possible: what rocprof of the kernel alone shows)
          roofline.hbm            the whole step against the HBM peak: PMC bytes actually moved, the structural floor of a
                                  fused design, and their ratio — the fraction BASELINE's metric names."""
        kernels, alg, pmc = self.kernels, self.alg, self.pmc
        self.wd.stage("roofline")
        dom = "dog_scan" if "dog_scan" in kernels else max((k for k in kernels if k in alg), key=lambda k: kernels[k]["ms_per_step"])
        dom_ms = kernels[dom]["ms_per_step"]
        dom_launches = max(1, kernels[dom]["launches_per_step"])
        roofline = self._roofline_dominant(dom, dom_ms, dom_launches)
        roofline["hbm"] = self._roofline_hbm()        # the HBM answer (BASELINE: "frames/s ... as achieved fraction of HBM roofline")
        roofline["pipeline"] = roofline["hbm"]        # (r01-r04 name of the same block)
        roofline["issue"] = self._roofline_issue()    # issue budget of the step
        roofline["issue_model"] = {
            "cycles_per_wave64_instruction_per_simd": {"v_fma/mul/add/sub_f32, v_add_u32": 2.85, "v_pk_fma/mul/add_f32": 4.45,
                                                       "DPP moves, v_max3, cvt, floor, integer mul/shift-add, v_cndmask": 4.7,
                                                       "transcendentals": 8.5},
            "note": "tools/valu_rates, tools/scan_rates (profiles/r02_valu_rates.txt, r02_scan_rates.txt; whole-kernel times): "
                    "a SIMD is saturated from 2-3 resident wavefronts on, occupancy beyond that buys nothing.  One scan row is 96 "
                    "v_pk_fma + 24 v_pk_mul + 56 v_pk_add + 48 DPP moves + 20 v_sub + 12 v_max/v_max3 = ~1085 cycles for 35.1 kflop "
                    "executed (isolated: 1085 measured), i.e. 0.50 of 64 flop/cycle/SIMD is the ceiling of this instruction "
                    "stream before halo lanes, segment prologues and the extremum tests"}
        roofline["avg_launch_ms"] = round(dom_ms / dom_launches, 4)
        roofline["duration_source"] = self.trace_note
        roofline["launches_per_step"] = dom_launches
        roofline["traffic"] = int((pmc[dom]["read"] + pmc[dom]["write"]) / dom_launches) if pmc and dom in pmc else None
        roofline["traffic_note"] = self.pmc_note
        roofline["pmc_calibration"] = self.calib
        if self.trace and "_all" in self.trace:
            roofline["child_run_step"] = {"gpu_busy_union_ms": round(self.trace["_all"]["busy_union_ms_per_step"], 4),
                                          "span_ms": round(self.trace["_all"]["span_ms_per_step"], 4),
                                          "note": "the kernel-trace child run (one context, batches queued back to back): wall time "
                                                  "during which any extraction kernel ran, and first start to last end, per step"}
        roofline["hbm_bound_kernels"] = self._roofline_hbm_kernels()
        if self.rank == 0:
            roofline["copy_ceiling_GBps"] = self._copy_ceiling_gbps()
        self.roofline = roofline

    def _roofline_dominant(self, dom, dom_ms, dom_launches):
        """The dominant kernel against the roof that bounds it (dog_scan: fp32 vector peak; anything else: HBM)."""
        kernels, alg, N, B = self.kernels, self.alg, self.N, self.B
        if dom == "dog_scan":
            fpp = dog_scan_flop_per_px()
            flops_step = float(fpp) * sum(N) * B
            union_ms = kernels[dom].get("union_ms_per_step")
            basis_ms = union_ms if union_ms else dom_ms
            ach = flops_step / (basis_ms * 1e-3) / 1e12
            ach_sum = flops_step / (dom_ms * 1e-3) / 1e12
            roofline = {"kernel": dom, "bound": "valu", "achieved": round(ach, 2), "peak": VALU_F32_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(min(ach / VALU_F32_PEAK_TF, 1.0), 4),
                        "frac_basis": ("wall-time UNION of the kernel's %d concurrent launches per step (rocprofv3 start / end "
                                       "stamps): %.4f ms" % (dom_launches, basis_ms)) if union_ms else
                                      "summed launch durations (no kernel trace in this run: the launches' overlap is unknown)",
                        "union_ms_per_step": round(union_ms, 4) if union_ms else None,
                        "summed": {"ms_per_step": round(dom_ms, 4), "achieved": round(ach_sum, 2),
                                   "frac": round(min(ach_sum / VALU_F32_PEAK_TF, 1.0), 4),
                                   "note": "the two launches' durations added up although they run side by side: double-counts "
                                           "the overlap (r01-r04 quoted this one as `frac`)"},
                        "flop_per_px": fpp, "flop_per_launch": int(flops_step / dom_launches),
                        "flop_note": "6 blurs x (9 vertical + 13 horizontal) + 4 shared pair sums + 5 DoG + 5 |v|max; "
                                     "algorithmic flops only (halo lanes, DPP moves, selects and the extremum tests are overhead)",
                        "alg_GBps": kernels[dom].get("alg_GBps"),
                        "alg_note": "60 B/px of LaplaceMulti + FindPointsMulti traffic (SURVEY 8d) that the fused kernel never moves; "
                                    "not a fraction of anything"}
            if self.rank == 0:
                sl = self.single_launch_scan(flops_step)
                if sl:
                    roofline["single_launch"] = sl
            if "alone_ms_per_step" in kernels[dom]:      # only with --contexts > 1
                ams = kernels[dom]["alone_ms_per_step"]
                aa = flops_step / (ams * 1e-3) / 1e12
                roofline["alone"] = {"ms_per_step": ams, "achieved": round(aa, 2), "frac": round(min(aa / VALU_F32_PEAK_TF, 1.0), 4),
                                     "note": "one batch at a time on one context (nothing else on the GPU); launch durations summed"}
        else:
            ach = alg[dom] * B / (dom_ms * 1e-3) / 1e9
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(min(ach / HBM_PEAK_GBS, 1.0), 4)}
        return roofline

    def _roofline_hbm(self):
        """The whole step against the HBM peak: PMC bytes actually moved, the structural floor, their ratio."""
        pmc, B, world = self.pmc, self.B, self.world
        fps1 = self.fps / world
        hbm = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "floor_bytes_per_frame": int(FLOOR_BYTES_PER_FRAME),
               "floor_frac": round(FLOOR_BYTES_PER_FRAME * fps1 / 1e9 / HBM_PEAK_GBS, 4),
               "alg_bytes_per_frame": int(ALG_BYTES_PER_FRAME), "alg_GBps": round(ALG_BYTES_PER_FRAME * fps1 / 1e9, 1),
               "note": "traffic = PMC bytes (FETCH_SIZE x gfx950 correction + WRITE_SIZE) of all kernels of a step; floor = input + "
                       "pyramid written once + read once by the scan + records; each x frames/s / 8 TB/s.  The algorithmic "
                       "figure (197.2 MB/frame, SURVEY 8d) is mostly traffic the fused design never moves: a rate only"}
        if pmc:
            tb = sum(e["read"] + e["write"] for e in pmc.values()) / B
            hbm.update({"traffic_bytes_per_frame": int(tb), "achieved": round(tb * fps1 / 1e9, 1),
                        "traffic_frac": round(tb * fps1 / 1e9 / HBM_PEAK_GBS, 4), "frac": round(tb * fps1 / 1e9 / HBM_PEAK_GBS, 4),
                        "traffic_over_floor": round(tb / FLOOR_BYTES_PER_FRAME, 3)})
        return hbm

    def _roofline_issue(self):
        """VALU issue budget of the step from the SQ counter pass."""
        issue = {"note": "insts = SQ_INSTS_VALU (wavefront-level VALU instructions) of all kernels of a step; "
                         "active = SQ_ACTIVE_INST_VALU x 4 (shader cycles a SIMD spent issuing them, summed over SIMDs); "
                         "valu_active_frac_of_step = active / (1024 SIMDs x clock x ms_per_step) with the clock the counters' own "
                         "GRBM_GUI_ACTIVE / dispatch duration; issue_frac_of_step = insts x 4.1 cycles (the step's mix priced with "
                         "tools/valu_rates: 2.85 plain / 4.45 packed / 4.7 DPP, integer, select) / the same denominator",
                 "source": self.sq_note}
        if self.sq:
            issue["source"] = ("collected live in this run: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE "
                               "over 4 steps of the timed entry point")
            insts = sum(e["insts"] for e in self.sq.values())
            active = 4.0 * sum(e["active_quad_cycles"] for e in self.sq.values())
            # shader clock under these kernels: cycles the dispatch was active / its duration in the same counter pass, over the
            # big kernels (GRBM_GUI_ACTIVE may be reported per XCD and summed: a figure 8x too large is divided down)
            clk = clk_raw = None
            big = [k for k in ("dog_scan", "descr_all", "lowpass_down") if k in self.sq and self.sq[k].get("ms", 0) > 0]
            if big:
                clk_raw = sum(self.sq[k]["gui_cycles"] for k in big) / (sum(self.sq[k]["ms"] for k in big) * 1e-3)
                clk = clk_raw / 8.0 if clk_raw > 4.0e9 else clk_raw
            issue["clock_raw_GHz"] = round(clk_raw / 1e9, 3) if clk_raw else None
            clk_used = clk if clk and 1.0e9 < clk < 2.6e9 else 2.4e9
            simd_cycles = 1024.0 * clk_used * self.ms_per_step * 1e-3
            issue.update({"insts_per_step": int(insts), "active_simd_cycles_per_step": int(active),
                          "clock_GHz": round(clk_used / 1e9, 3),
                          "clock_source": "GRBM_GUI_ACTIVE / dispatch duration" if clk_used == clk else "nominal 2.4 GHz",
                          "valu_active_frac_of_step": round(active / simd_cycles, 4),
                          "issue_frac_of_step": round(insts * 4.1 / simd_cycles, 4),
                          "per_kernel_M_insts": {k: round(e["insts"] / 1e6, 1) for k, e in sorted(self.sq.items())}})
            # the per-keypoint kernels against what the reference's algorithm asks for per keypoint (r06)
            kp = max(1.0, self.kp_per_frame * self.B)
            per_kp = {k: round(self.sq[k]["insts"] / kp, 1) for k in ("orient_all", "descr_all", "refine") if k in self.sq}
            issue["per_keypoint"] = {
                "keypoints_per_step": int(kp), "wave_instructions_per_keypoint": per_kp,
                "floor_note": "one wavefront per keypoint.  descr_all: the reference's descriptor is 256 samples x 4 bilinear "
                              "fetches + 256 x 8 votes (cudaSiftD.cu:340-386) = 16 fetches per lane at ~27 VALU each (coordinate, "
                              "fract/floor, two 8-bit weight roundings, address, 4-texel blend) = 432, + 4 x (sqrt, FastAtan2, "
                              "vote split) ~ 120, + two 64-tap footprint sums per lane = 128 fmaf, + two wave reductions and the "
                              "normalisations ~ 60: ~740 per orientation against ~1040 measured per keypoint (1.1 orientations "
                              "per keypoint on these frames: ~0.87 of the measured count is the algorithm itself).  orient_all: "
                              "15 x 15 gradient samples (cudaSiftD.cu:972-1057) = 3.5 per lane x (4-texel gradient, exp weight, "
                              "atan2, bin) ~ 60 = 210, + 32-bin histogram smoothing, two peak searches and the parabola fits "
                              "~ 120: ~330 of the ~430 measured"}
        return issue

    def _roofline_hbm_kernels(self):
        """The genuinely HBM-bound kernels: algorithmic bytes AND bytes actually moved, each over the summed launch time."""
        kernels, alg, pmc, B = self.kernels, self.alg, self.pmc, self.B
        hbm_kernels = {}
        split = kernels.get("dog_scan", {}).get("launches_per_step", 1) > 1
        for k in ("lowpass", "lowpass_down", "scaledown"):
            if k == "scaledown" and split:
                continue      # runs beside the fine-level scan on a second stream: its duration says nothing about HBM
            if k in kernels:
                kms = kernels[k].get("alone_ms_per_step", kernels[k]["ms_per_step"])     # bandwidth of the kernel ALONE on the GPU
                a = alg[k] * B / (kms * 1e-3) / 1e9
                hbm_kernels[k] = {"ms_per_step": kms, "alg_GBps": round(a, 1), "alg_frac": round(a / HBM_PEAK_GBS, 4)}
                if pmc and k in pmc:
                    t = (pmc[k]["read"] + pmc[k]["write"]) / (kms * 1e-3) / 1e9
                    hbm_kernels[k]["traffic_GBps"] = round(t, 1)
                    hbm_kernels[k]["traffic_frac"] = round(t / HBM_PEAK_GBS, 4)
        return hbm_kernels

    def _copy_ceiling_gbps(self):
        """Achievable-copy ceiling: device-to-device copy of 1 GiB, read + write bytes counted."""
        torch = self.torch
        a = torch.empty(1 << 28, dtype=torch.float32, device=self.device)
        b = torch.empty_like(a)
        for _ in range(2):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        return round(2.0 * a.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)

    # ------------------------------------------------------------------ self-validation + CPU baseline
    def leg_validate_and_cpu(self):
        """Self-validation of the TIMED loop's last step against the oracle, and — N = 1 only — the CPU baseline: the
        oracle (`kind: port`) on the host's cores.  With N > 1 the other ranks wait in the closing (blocking) barrier."""
        args, np, orc = self.args, self.np, self.orc
        if not (self.rank == 0 and orc is not None):
            return
        from util import compare_points
        B, NB, world = self.B, self.NB, self.world
        self.wd.stage("oracle: validation + cpu_baseline")
        cores, logical_cpus, cpu_note = effective_cpus()
        ncpu = args.cpu_frames
        if ncpu <= 0:
            try:
                import psutil
                avail = psutil.virtual_memory().available
            except Exception:
                avail = 16 << 30
            # the validated frames (one batch) or four rounds of one frame per core, whichever is more
            ncpu = int(max(8, min(max(B, 4 * cores), NB * B, avail // 2 // (260 << 20))))
        ncpu = min(ncpu, NB * B)
        # frames of the last timed batch first (they are validated), then the following ones
        order = [(self.last_b0 + i) % (NB * B) for i in range(ncpu)]
        host = self.frames[order].cpu().numpy()
        outer = min(ncpu, cores)          # one frame per core the process may use (the knee of the sweep); no nested teams
        inner = 1
        orc.extract_batch(host[:min(ncpu, outer)], NUM_OCTAVES, INIT_BLUR, THRESH, max_pts=8192, outer_threads=outer,
                          inner_threads=inner)                                           # warm-up (work buffers of every thread)
        self.wd.stage("oracle: first timed pass")
        tc0 = time.perf_counter()
        ref, nref, cref = orc.extract_batch(host, NUM_OCTAVES, INIT_BLUR, THRESH, max_pts=8192, outer_threads=outer,
                                            inner_threads=inner)
        cdt = time.perf_counter() - tc0
        cpu_frames = ncpu
        while world == 1 and cdt < 10.0 and cpu_frames < 200 * ncpu:     # the contract asks for 10-30 s of CPU work
            self.wd.stage("oracle: timed pass")
            tc1 = time.perf_counter()
            orc.extract_batch(host, NUM_OCTAVES, INIT_BLUR, THRESH, max_pts=8192, outer_threads=outer, inner_threads=inner)
            cdt += time.perf_counter() - tc1
            cpu_frames += ncpu
        nval = min(ncpu, B)
        for f in range(nval):
            got = self.last_recs[self.last_offs[f]:self.last_offs[f + 1]]
            if int(nref[f]) != int(self.last_counts[f]) or len(got) != int(nref[f]):
                raise RuntimeError("self-validation failed: frame %d of the last timed step has %d records, oracle %d"
                                   % (f, int(self.last_counts[f]), int(nref[f])))
            compare_points(ref[f, :nref[f]], got, "bench_validate_f%d" % f)              # raises AssertionError on mismatch
        self.validated = nval
        if world != 1:
            return
        self.cpu = {"value": round(cpu_frames / cdt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
                    "threads": outer, "logical_cpus": logical_cpus, "cores_note": cpu_note,
                    "per_core": round(cpu_frames / cdt / outer, 3),
                    "sample": "%d extractions of %d of the same synthetic 1920x1080 frames (%.1f s), oracle/sift_oracle.c: one frame "
                              "per OpenMP thread on %d threads (OpenCV cv::SIFT is not installed on this image)"
                              % (cpu_frames, ncpu, cdt, outer),
                    "keypoints_per_frame": round(float(np.mean(nref)), 1)}
        # matcher CPU baseline: the reference's OWN AVX2/OpenMP routine MatchC3 (match.cu:102-130, built from the
        # reference tree into oracle/_ref by oracle/build_ref.sh) on its own 16384 x 16384 problem
        L = orc.ref_lib(16384)
        if L is not None and self.match is not None:
            self.wd.stage("reference MatchC3 on the host")
            a = orc.aligned_f32(16384 * 128); b = orc.aligned_f32(16384 * 128)
            sc = orc.aligned_f32(16384); ix = np.zeros(16384, np.int32)
            try:                                  # MatchC3's `#pragma omp parallel for` takes the default team: cap it at
                C.CDLL("libgomp.so.1").omp_set_num_threads(cores)      # the CPUs this process may actually use
            except Exception:
                pass
            L.ref_generate(a.ctypes.data, b.ctypes.data, 1)
            L.ref_match_c3(a.ctypes.data, b.ctypes.data, sc.ctypes.data, ix.ctypes.data)     # warm-up
            tm0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                L.ref_match_c3(a.ctypes.data, b.ctypes.data, sc.ctypes.data, ix.ctypes.data)
            mdt = (time.perf_counter() - tm0) / reps
            self.match["cpu_baseline"] = {"value": round(16384.0 * 16384.0 / mdt / 1e6, 1), "unit": "Mpairs/s",
                                          "cores": cores, "threads": cores, "kind": "reference",
                                          "sample": "reference MatchC3 (AVX2+FMA, OpenMP; argmax only, no runner-up) on "
                                                    "16384 x 16384 x 128, its own generator"}

    # ------------------------------------------------------------------ the line
    def emit(self):
        args, B, NB, world, comm = self.args, self.B, self.NB, self.world, self.comm
        if self.rank != 0:
            return
        validated = self.validated
        out = {"metric": "1920x1080 SIFT frames/sec", "value": round(self.fps, 1), "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(self.ms_per_step, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": "batches of %d synthetic 1920x1080 frames per GPU, %d distinct batches per GPU rotated "
                                      "(%d distinct frames per GPU; BASELINE config 4: 512 frames over 8 GPUs), ExtractSift "
                                      "5 octaves initBlur 1.0 thresh 3.0 maxPts 32768, frames resident in HBM, count read-back%s"
                                      % (B, NB, NB * B, (" + RCCL gather of SiftData (misift_gather_*), root = %s"
                                                         % ("step %% N" if args.gather_root == "rotate" and world > 1 else "rank 0"))
                                         if comm else ""),
                          "frames_per_gpu": B, "distinct_frames_per_gpu": NB * B, "contexts": self.NCTX,
                          "batches_in_flight": self.NCTX * self.RING,
                          "batches_in_flight_note": "misift_ctx_set_batches_in_flight(%d): pipelines behind ONE context; kernel "
                                                    "durations of the roofline come from a one-batch-at-a-time child run" % self.RING,
                          "contexts_note": "the steps rotate over this many misift contexts per GPU (own stream, staging and "
                                           "scratch arena each): that many batches are in flight on the GPU",
                          "path": "unfused" if args.unfused else "fused dog+detect",
                          "keypoints_per_frame": round(self.kp_per_frame, 1),
                          "timing_definition": "T2 of BASELINE.md section 2: device-resident end to end, LowPass ... descriptors + "
                                               "count read-back (T1 and T3 of the same run: `timing_definitions`)",
                          "preroll_steps": self.PRE,
                          "value_without_preroll": self.no_preroll["value"] if self.no_preroll else None},
               "timing_definitions": self.timing_definitions(),
               "config2_1280x960": self.cfg2,
               "no_preroll": self.no_preroll,
               "validated_frames": validated,
               "validation": ("frames 0..%d of the LAST timed step: counts and every record equal to the oracle (tests/util.py "
                              "compare_points: identical keypoint set; position, scale, orientation, sharpness, edgeness "
                              "bit-identical; every descriptor element within 1e-6; no outlier budget)" % (validated - 1))
               if validated else "skipped (--no-cpu)",
               "roofline": self.roofline, "kernels": self.kernels, "step_ms": self.step_ms, "match": self.match,
               "cpu_baseline": self.cpu, "pcie_inclusive": self.pcie, "single_frame": self.latency, "skewed_batch": self.skewed,
               "gather": self.gather_info, "rccl_ranks": int(comm.size) if comm else 0,
               "preroll_steps": self.PRE,
               "preroll_note": "untimed steps of the same loop before the W warm-up steps (clock settling, DESIGN.md section 5); "
                               "the timed region is exactly `steps` steps; `no_preroll` is the same window measured first, without them",
               "stages_s": {n: t for n, t in self.wd.history if t >= 0.5},
               "kernels_note": "ms_per_step = dispatch durations from the rocprofv3 kernel trace of a child run (hip_event_ms_per_step: "
                               "the library's event pairs over 10 more steps of the pipelined loop); dog_scan runs as two launches "
                               "per step (fine levels on the context stream, the coarse ScaleDowns + coarse levels beside it on a "
                               "second stream) whose durations overlap — union_ms_per_step is the wall time either was running.  "
                               "With --contexts > 1 kernels of several batches share the GPU, every duration is stretched by its "
                               "neighbours and alone_ms_per_step gives the same kernels with one batch at a time on one context"}
        C.CDLL(None).fflush(None)          # RCCL's version banner sits in the C stdio buffer: get it out first,
        print(json.dumps(out), flush=True)  # the JSON line is the LAST line of stdout

    def timing_definitions(self):
        """the three extraction timings, always side by side."""
        td = {"T2": {"frames_per_s": round(self.fps, 1), "what": "`value`: LowPass ... descriptors + count read-back, frames resident in HBM"},
              "T3": {"frames_per_s_f32": self.pcie.get("frames_per_s_f32") if isinstance(self.pcie, dict) else None,
                     "frames_per_s_u8": self.pcie.get("frames_per_s_u8") if isinstance(self.pcie, dict) else None,
                     "what": "T2 + H2D upload of the frames + D2H of the SiftPoint records (`pcie_inclusive`: the host-fed pipeline; "
                             "fp32 frames as the reference uploads them, and 8-bit frames converted in registers)"}}
        k = self.kernels or {}
        alone = {n: e.get("alone_ms_per_step", e.get("ms_per_step")) for n, e in k.items() if isinstance(e, dict)}
        if alone and "lowpass_down" in alone and all(v is not None for v in alone.values()):
            tot = sum(alone.values())
            lp_share = 8.0 / 13.0          # lowpass_down's algorithmic bytes: LowPass 8 B/px + first ScaleDown 5 B per input px
            t1_lo = tot - alone["lowpass_down"]
            t1_hi = tot - lp_share * alone["lowpass_down"]
            td["T1"] = {"ms_per_step_one_batch_at_a_time": [round(t1_lo, 4), round(t1_hi, 4)],
                        "frames_per_s": [round(self.world * self.B / t1_hi * 1e3, 1), round(self.world * self.B / t1_lo * 1e3, 1)],
                        "T2_same_basis_ms": round(tot, 4),
                        "what": "the reference's `timer1` scope (cudaSiftH.cu:113-117): everything but the initial LowPass.  Here "
                                "LowPass and the first ScaleDown are ONE HBM-bound launch (lowpass_down), so T1 is bracketed: summed "
                                "one-batch-at-a-time launch durations without lowpass_down (the first ScaleDown is missing: a lower "
                                "bound on the time), and the same plus the ScaleDown's 5/13 share of that launch's algorithmic bytes.  "
                                "Same basis as T2_same_basis_ms — launch durations summed although dog_scan's two launches and the "
                                "scaledowns overlap — not as `value` (which also overlaps batches)"}
        return td

    def leg_config2(self):
        """BASELINE config 2's shape through the batch path: 64 synthetic 1280x960 frames per step, reported with its fraction
        of the 117.3 MB/frame HBM roofline.  A side leg; never `value`."""
        args, torch = self.args, self.torch
        self.cfg2 = None
        if not (self.rank == 0 and self.world == 1 and not args.no_latency and not args.unfused):
            return
        self.wd.stage("config 2: 1280x960 batches")
        capi = self.capi
        w2, h2, B = 1280, 960, self.B
        try:
            fr = gen_frames_torch(torch, B, 7000, self.device)[:, :h2, :w2].contiguous()
            K = max(1, self.RING)
            S = capi.scratch_floats(w2, h2, NUM_OCTAVES, False)
            scr = [torch.empty(S * B, dtype=torch.float32, device=self.device) for _ in range(K)]
            cnt = torch.zeros(2 * B + 1, dtype=torch.int32, device=self.device)
            packed = torch.empty(576 * 8192 * B, dtype=torch.uint8, device=self.device)

            def loop(n):
                for i in range(n):
                    capi.check(capi.lib().misift_extract_batch_packed_async(
                        self.ctx.h, fr.data_ptr(), B, h2 * w2, w2, h2, w2, NUM_OCTAVES, INIT_BLUR, THRESH, 0.0, scr[i % K].data_ptr(),
                        None, 8192, cnt.data_ptr(), cnt.data_ptr() + 4 * B, packed.data_ptr()), "config2")
            loop(8)
            torch.cuda.synchronize()
            n = 24
            t0 = time.perf_counter()
            loop(n)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            fps = B / dt
            alg = 117.3e6
            self.cfg2 = {"frames_per_s": round(fps, 1), "ms_per_step": round(1e3 * dt, 4), "frames_per_step": B,
                         "keypoints_per_frame": round(float(cnt[:B].float().mean().item()), 1),
                         "alg_bytes_per_frame": int(alg), "alg_TBps": round(fps * alg / 1e12, 3),
                         "frac_of_117MB_roofline": round(fps * alg / (HBM_PEAK_GBS * 1e9), 4),
                         "note": "BASELINE config 2 (1280x960, 5 octaves) as 64-frame batches on the headline's context, %d batches in "
                                 "flight; crops of the synthetic 1920x1080 generator; 117.3 MB/frame = SURVEY 8d's algorithmic bytes, "
                                 "most of which the fused scan never moves (the fraction may exceed 1)" % K}
            del fr, scr, packed
        except Exception as e:                                   # noqa: BLE001 — a side leg must not cost the line
            self.cfg2 = {"error": repr(e)[:200]}

    def close(self):
        self.wd.stage("teardown")
        if self.comm is not None:
            self.comm.close()
        if self.world > 1:
            self.dist.destroy_process_group()
        for c in self.ctxs[1:]:
            c.close()
        self.ctx.close()


def main():
    args = parse_args()
    # `--gpus N` as a plain command (the driver's shape): spawn the N ranks ourselves
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.emulate_ranks and not args.pmc_child:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], need_gpus=not args.spawn_check))
    if args.spawn_check:
        sys.exit(spawn_check())
    if args.batches_in_flight <= 0:
        args.batches_in_flight = 2
    # HIP multiplexes a process's streams onto 4 hardware queues unless told otherwise, and a stream that shares a
    # queue with the extraction stream runs BEHIND the batches queued there: the gather's communication stream then
    # completes batch k-2 only after batch k, the host cannot run ahead and the pipeline loses its depth (1 rank through
    # the communicator: 41 k instead of 51 k frames/s; the host-fed pipe wanders between 16 k and 24 k).  Must be set
    # before the HIP runtime initialises, i.e. before torch is imported; a caller's own setting wins.
    if not os.environ.get("BENCH_NO_QUEUE_DEFAULT"):     # developer switch: leave HIP's default of 4 hardware queues
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if args.pmc_child:
        return pmc_child()
    if args.emulate_ranks > 0:
        return emulate_ranks(args)
    b = Bench(args)
    b.leg_matcher()               # BASELINE config 5 (its own timed region)
    b.setup_inputs()
    b.leg_timed_loop()            # `value` (+ `no_preroll`)
    b.leg_other_root()            # N > 1: the other gather root
    b.leg_kernel_events()
    b.leg_single_frame()          # BASELINE configs 2 and 3
    b.leg_skewed_batch()
    b.leg_config2()
    b.leg_pcie()
    b.leg_counters_and_trace()    # rocprofv3 child passes (rank 0, N = 1)
    b.leg_roofline()
    b.leg_validate_and_cpu()
    b.emit()
    b.close()


if __name__ == "__main__":
    main()
