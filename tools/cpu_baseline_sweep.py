#!/usr/bin/env python3
"""CPU baseline scaling sweep: oracle/sift_oracle.c (the port; OpenCV is not installed) on synthetic
1920x1080 frames, one frame per thread, for a range of thread counts, thread placements and process splits.
Every configuration runs in its own subprocess (OpenMP placement is fixed at library start-up).

  python tools/cpu_baseline_sweep.py            -> table on stdout + gpurun_out/r03_cpu_baseline_sweep.json
  python tools/cpu_baseline_sweep.py --worker T NFRAMES   (internal)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
FRAMES = "/tmp/cpu_sweep_frames.npy"


def worker(threads, nframes):
    import numpy as np
    from oracle import pyoracle as orc
    base = np.load(FRAMES, mmap_mode="r")
    imgs = np.stack([np.asarray(base[i % len(base)]) for i in range(nframes)])      # first touch in this process
    orc.extract_batch(imgs[:threads], 5, 1.0, 3.0, max_pts=8192, outer_threads=threads, inner_threads=1)   # warm-up
    t0 = time.perf_counter()
    _, n, _ = orc.extract_batch(imgs, 5, 1.0, 3.0, max_pts=8192, outer_threads=threads, inner_threads=1)
    dt = time.perf_counter() - t0
    print(json.dumps({"frames": nframes, "seconds": dt, "keypoints": float(np.mean(n))}))


def run_config(threads, procs, bind, rounds=2):
    env = dict(os.environ)
    if bind:
        env.update(OMP_PROC_BIND="close", OMP_PLACES="cores")
    else:
        env.pop("OMP_PROC_BIND", None)
        env.pop("OMP_PLACES", None)
    per = threads // procs
    nfr = per * rounds
    t0 = time.perf_counter()
    ps = []
    cores = os.cpu_count() or 1
    for p in range(procs):
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", str(per), str(nfr)]
        if procs > 1 and bind:                       # give every process its own block of cores
            lo = p * per % cores
            cmd = ["taskset", "-c", "%d-%d" % (lo, min(cores - 1, lo + per - 1))] + cmd
        ps.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    secs = []
    for p in ps:
        out = p.communicate()[0].strip().splitlines()
        secs.append(json.loads(out[-1])["seconds"] if out else float("nan"))
    wall = time.perf_counter() - t0
    work = max(secs)                                 # the timed region of the slowest process (start-up excluded)
    return {"threads": threads, "processes": procs, "bind_close_cores": bool(bind), "frames": nfr * procs,
            "seconds": round(work, 3), "frames_per_s": round(nfr * procs / work, 2),
            "frames_per_s_per_thread": round(nfr * procs / work / threads, 3), "wall_incl_startup_s": round(wall, 1)}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker(int(sys.argv[2]), int(sys.argv[3]))
    import numpy as np
    from synth import synth_frame
    cores = os.cpu_count() or 1
    if not os.path.exists(FRAMES):
        np.save(FRAMES, np.stack([synth_frame(f) for f in range(8)]))
    res = []
    for threads in [t for t in (8, 16, 32, 64, 128, 256) if t <= cores]:
        for procs, bind in ((1, False), (1, True)) + (((threads // 8, True),) if threads >= 32 else ()):
            r = run_config(threads, procs, bind)
            res.append(r)
            print(r, flush=True)
    best = max(res, key=lambda r: r["frames_per_s"])
    out = {"host_cores": cores, "workload": "oracle/sift_oracle.c, 1920x1080 synthetic frames, 5 octaves, thresh 3.0; one frame per "
           "thread, 2 frames per thread timed after a warm-up frame per thread", "configs": res, "best": best}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_cpu_baseline_sweep.json"), "w"), indent=1)
    print("best:", best)


if __name__ == "__main__":
    main()
