import sys, json, subprocess
for w in (5, 30):
    out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", str(w), "--batches-in-flight", "1", "--no-pmc", "--no-match", "--no-cpu", "--no-pcie", "--no-latency"], capture_output=True, text=True, env=dict(__import__("os").environ, BENCH_STEP_DUMP="1")).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print("warmup", w, "fps", d["value"], "ms", d["ms_per_step"], d["step_ms"])
