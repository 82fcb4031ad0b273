"""CPU checks of the measurement definitions in bench.py (no GPU): the algorithmic-byte model must be the one
SURVEY.md section 8(d) states, because roofline.achieved is computed from it."""
import importlib.util
import os

import numpy as np


def _bench():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    spec = importlib.util.spec_from_file_location("bench_module", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_survey_8d():
    b = _bench()
    N = b.octave_pixels(1920, 1080, 5)
    assert N == [2073600, 518400, 129600, 32400, 8040]                      # sum = 2 762 040 px
    alg = b.algorithmic_bytes_per_frame()
    assert alg["lowpass"] == 16588800                                        # 16.59 MB
    assert abs(alg["scaledown"] - 13.77e6) < 0.01e6                          # 13.77 MB
    assert abs(alg["laplace"] - 88.39e6) < 0.01e6 and abs(alg["detect"] - 77.34e6) < 0.01e6
    total = alg["lowpass"] + alg["scaledown"] + alg["laplace"] + alg["detect"] + 576 * 2000
    assert abs(total - 197.2e6) < 0.1e6                                      # the 197.2 MB/frame of SURVEY 8(d)
    assert alg["dog_scan"] == alg["laplace"] + alg["detect"]                 # the fused kernel is priced as both
    N2 = b.octave_pixels(1280, 960, 5)
    assert N2 == [1228800, 307200, 76800, 19200, 4800]


def test_default_arguments_finish_quickly():
    b = _bench()
    src = open(b.__file__).read()
    assert '"--gpus"' in src and '"--steps"' in src and '"--warmup"' in src    # the driver's contract


def test_dog_scan_flop_model():
    """roofline.frac of the dominant kernel is (146 flop/px x pixels) / time / 157.3 TF: the count is derived in
    code from the blur structure and must stay what DESIGN.md documents."""
    b = _bench()
    assert b.dog_scan_flop_per_px() == 146
    N = b.octave_pixels(1920, 1080, 5)
    assert abs(146 * sum(N) * 64 - 25.8e9) < 0.05e9                         # 25.8 GFLOP per 64-frame step
    assert b.FLOOR_BYTES_PER_FRAME < 54.2e6 < b.ALG_BYTES_PER_FRAME          # floor < measured r01 traffic < algorithmic


def test_kernel_trace_parsing(tmp_path, monkeypatch):
    """collect_trace(): per-kernel dispatch durations from a rocprofv3 kernel-trace CSV — warm-up launches dropped,
    two launches per step summed, unknown kernels ignored.  (rocprofv3 itself is faked: no GPU here.)"""
    import bench
    steps, skip = 6, 2
    rows = ["Kind,Agent_Id,Queue_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp"]
    t = 1000
    for s in range(steps):
        dur = 900000 if s < skip else 300000                 # warm-up launches are slow
        for name in ("void dog_scan_all_kernel<true>(float const*, ScanAllGeom)", "void dog_scan_all_kernel<true>(float const*, ScanAllGeom)",
                     "void descr_all_kernel<true, 4>(float const*)", "void at::native::elementwise_kernel<128>(int)"):
            rows.append('"KERNEL_DISPATCH",1,1,1,"%s",1,%d,%d' % (name, t, t + dur))
            t += dur + 50

    class P:
        returncode = 0
        stdout = b""

    def fake_run(cmd, **kw):
        d = cmd[cmd.index("-d") + 1]
        import os
        os.makedirs(os.path.join(d, "host"), exist_ok=True)
        open(os.path.join(d, "host", "t_kernel_trace.csv"), "w").write("\n".join(rows) + "\n")
        assert kw["env"]["BENCH_CHILD_PIPELINED"] == "1" and kw["env"]["BENCH_CHILD_STEPS"] == str(steps)
        return P()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.os.path, "exists", lambda p: True)
    res, note = bench.collect_trace(64, steps=steps, skip=skip)
    assert note is None
    assert res["dog_scan"]["launches_per_step"] == 2 and abs(res["dog_scan"]["ms_per_step"] - 0.6) < 1e-9
    assert res["descr_all"]["launches_per_step"] == 1 and abs(res["descr_all"]["ms_per_step"] - 0.3) < 1e-9
    assert set(res) == {"dog_scan", "descr_all", "_all"}
    # back-to-back launches never overlap here: the union is the sum
    assert abs(res["dog_scan"]["union_ms_per_step"] - 0.6) < 1e-9


def test_kernel_trace_union_of_overlapping_launches(tmp_path, monkeypatch):
    """roofline.frac is quoted against the wall time during which a dog_scan launch was RUNNING: the two launches of a step
    (fine levels | coarse levels on a second stream) overlap, and the union counts the overlap once.
    Canned trace: per step launch A = [t, t+400 us), launch B = [t+150, t+450 us) -> sum 0.7 ms, union 0.45 ms; a descr_all
    launch overlapping the next step's scan must not leak into dog_scan's union."""
    import bench
    assert bench.interval_union_ns([]) == 0
    assert bench.interval_union_ns([(0, 10), (5, 20), (30, 40), (32, 35), (40, 41)]) == 31
    steps, skip = 5, 1
    rows = ["Kind,Agent_Id,Queue_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp"]
    scan = "void dog_scan_all_kernel<1, false>(float const*, ScanAllGeom)"
    for s_ in range(steps):
        t = 10_000_000 + s_ * 1_000_000
        rows.append('"KERNEL_DISPATCH",1,1,1,"%s",1,%d,%d' % (scan, t, t + 400_000))
        rows.append('"KERNEL_DISPATCH",1,2,1,"%s",1,%d,%d' % (scan, t + 150_000, t + 450_000))
        rows.append('"KERNEL_DISPATCH",1,1,1,"void descr_all_kernel<true, 4, true>(float const*)",1,%d,%d' % (t + 500_000, t + 1_100_000))

    class P:
        returncode = 0
        stdout = b""

    def fake_run(cmd, **kw):
        d = cmd[cmd.index("-d") + 1]
        os.makedirs(os.path.join(d, "host"), exist_ok=True)
        open(os.path.join(d, "host", "t_kernel_trace.csv"), "w").write("\n".join(rows) + "\n")
        return P()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.os.path, "exists", lambda p: True)
    res, note = bench.collect_trace(64, steps=steps, skip=skip)
    assert note is None
    assert res["dog_scan"]["launches_per_step"] == 2
    assert abs(res["dog_scan"]["ms_per_step"] - 0.7) < 1e-9            # the sum double-counts the overlap ...
    assert abs(res["dog_scan"]["union_ms_per_step"] - 0.45) < 1e-9     # ... the union does not
    assert abs(res["descr_all"]["union_ms_per_step"] - 0.6) < 1e-9
    # all kernels together: scan [0, 450) + descr [500, 1100) of a 1000-us period, descr overlapping the next step's scan
    # (steps 1..4: 450 + 3 x 950 + 600 us of busy time)
    assert abs(res["_all"]["busy_union_ms_per_step"] - 3.9 / 4) < 1e-9


# ---- `python bench.py --gpus N` as a plain command spawns its own ranks (r04; VERDICT r03 "next" #2)
def _run_bench(args, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_n_spawns_n_ranks_over_localhost():
    """The launcher half of `--gpus N`: N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1, rendezvous
    works (gloo stands in for RCCL: --spawn-check needs no GPU), rank 0's JSON is the last line of stdout."""
    import json
    r = _run_bench(["--gpus", "3", "--spawn-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"spawn_check": True, "world": 3, "sum_of_ranks_plus_1": 6, "local_rank": 0, "master_addr": "127.0.0.1"}


def test_a_failing_rank_fails_the_whole_run():
    r = _run_bench(["--gpus", "2", "--spawn-check"], {"BENCH_SPAWN_CHECK_FAIL_RANK": "1"})
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert "a rank exited with code 7" in r.stderr and "spawn_check" not in r.stdout


def test_a_silent_rank_is_ended_by_the_watchdog():
    """A rank that hangs (in rendezvous, ncclCommInitRank, a collective ...) must not cost the launcher its whole time-out:
    its watchdog prints the rank and the stage it is stuck in and exits with rc 6, the launcher stops the others."""
    import time
    t0 = time.time()
    r = _run_bench(["--gpus", "2", "--spawn-check"], {"BENCH_SPAWN_CHECK_HANG_RANK": "1", "BENCH_WATCHDOG_S": "3"}, timeout=120)
    assert r.returncode == 6, (r.returncode, r.stderr[-2000:])
    # whichever rank's watchdog fires first (the hung one, or rank 0 waiting for it in the barrier) names every rank's stage
    assert "watchdog: rank" in r.stderr and "rank 1: in stage 'pretend_collective'" in r.stderr, r.stderr[-1500:]
    assert time.time() - t0 < 60
    assert "spawn_check" not in r.stdout


def test_gpus_n_refuses_to_run_on_fewer_devices():
    """`--gpus 2` where fewer than 2 devices are visible must fail loudly — r03's bench printed a warning and measured one
    GPU, which the driver would have recorded as a 2-GPU number."""
    b = _bench()
    have = b.visible_gpus()
    if have >= 2:
        import pytest
        pytest.skip("%d GPUs visible: `--gpus 2` is a real run here" % have)
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 3 and r.stdout.strip() == "", (r.returncode, r.stdout[-500:])
    assert "refusing to run" in r.stderr


def test_launcher_mismatch_is_refused():
    """WORLD_SIZE from a launcher that disagrees with --gpus: no silent fallback to either number."""
    r = _run_bench(["--gpus", "4", "--steps", "2", "--warmup", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 4 and "refusing to run" in r.stderr, (r.returncode, r.stderr[-500:])


def _torchrun(args, env_extra=None, timeout=300):
    """The driver's launcher shape: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ..."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b = _bench()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(b._free_port()), os.path.join(root, "bench.py")] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_under_the_drivers_launcher_the_ranks_are_not_spawned_twice():
    """`torch.distributed.run ... bench.py --gpus 2`: WORLD_SIZE comes from the launcher, bench.py must take its place as a
    rank (no self-spawn) and rank 0's JSON must be on stdout (gloo stands in for RCCL: --spawn-check needs no GPU)."""
    import json
    r = _torchrun(["--gpus", "2", "--spawn-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    assert json.loads(lines[0])["world"] == 2


def test_under_the_drivers_launcher_a_silent_rank_ends_the_job():
    """The same with a rank that hangs: its watchdog (or its peer's, waiting in the barrier) exits with rc 6, the launcher
    tears the job down — minutes before any driver-side time-out."""
    import time
    t0 = time.time()
    r = _torchrun(["--gpus", "2", "--spawn-check"], {"BENCH_SPAWN_CHECK_HANG_RANK": "1", "BENCH_WATCHDOG_S": "3"}, timeout=240)
    assert r.returncode != 0
    assert "watchdog: rank" in r.stderr and "pretend_collective" in r.stderr, r.stderr[-2000:]
    assert time.time() - t0 < 120
