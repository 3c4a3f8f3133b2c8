"""Noise-aware A/B of builds of libicar_hip.so on ONE lease (>= 5 alternating runs per arm, median and
spread, keep only if the medians differ by more than the spread):
    python profiles/micro/ab.py [-n 5] [--bench-args "..."] name=path/to/lib.so [name=path ...]
Each round runs every arm once, in rotating order; every run is a fresh `python bench.py --no-cpu-baseline --no-traffic-probe
--no-later-window` process with ICAR_HIP_LIB pointing at the arm.  Prints one line per arm: median [min .. max] of ms/step, the
advection's and the microphysics' event-timer averages, and appends everything to gpurun_out/ab_<tag>.jsonl."""
import json, os, statistics, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
args = sys.argv[1:]; n = 5; bench_args = ""; tag = "ab"
arms = []
while args:
    a = args.pop(0)
    if a == "-n": n = int(args.pop(0))
    elif a == "--bench-args": bench_args = args.pop(0)
    elif a == "--tag": tag = args.pop(0)
    else: k, v = a.split("=", 1); arms.append((k, os.path.abspath(v)))
res = {k: [] for k, _ in arms}
for r in range(n):
    order = arms[r % len(arms):] + arms[:r % len(arms)]
    for k, lib in order:
        env = dict(os.environ, ICAR_HIP_LIB=lib)
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-traffic-probe", "--no-later-window", "--steps", "20", "--warmup", "5"] + bench_args.split()
        try:
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]
            d = json.loads(out); rf = d["roofline"]
            row = {"ms": d["ms_per_step"], "advect": rf.get("avg_ms") or 0.0, "mp": rf.get("mp_ms_per_step") or 0.0, "setup": rf.get("setup_ms_per_step") or 0.0}
        except Exception as e:
            row = {"error": str(e)}
        res[k].append(row)
        print(f"round {r} {k}: {row}", flush=True)
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
with open(os.path.join(root, "gpurun_out", f"ab_{tag}.jsonl"), "a") as f:
    f.write(json.dumps({"arms": dict(arms), "bench_args": bench_args, "runs": res}) + "\n")
print(f"--- {n} alternating runs per arm, bench args: '{bench_args}'")
for k, _ in arms:
    good = [x for x in res[k] if "ms" in x]
    line = f"{k:12s}"
    for key in ("ms", "advect", "mp", "setup"):
        v = [x[key] for x in good]
        if v: line += f"  {key} {statistics.median(v):.4f} [{min(v):.4f} .. {max(v):.4f}]"
    print(line + f"  ({len(good)} runs)")
