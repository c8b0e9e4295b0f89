#!/usr/bin/env python3
"""Interleaved A/B of two (or more) builds of libagdiff_hip.so on ONE box (device clocks differ by ~10 % between boxes):
each lib is loaded through AGDIFF_LIB in its own bench.py process; prints the fused CFConv launch time and the step time.
  python tools/conv_ab.py [--reps 3] libA.so libB.so [-- bench args]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); args, extra = args[:i], args[i + 1:]
reps = 3
if args and args[0] == "--reps":
    reps = int(args[1]); args = args[2:]
res = {a: [] for a in args}
for _ in range(reps):
    for lib in args:
        env = dict(os.environ, AGDIFF_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "40", "--warmup", "5",
                              "--no-cpu-baseline", "--no-traj", "--no-extra"] + extra, env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            by = d["roofline"].get("launch_ms_by_kernel", {})
            res[lib].append((d["roofline"]["avg_launch_ms"], d["ms_per_step"], by))
        except Exception:
            print(lib, "FAILED", out.stderr[-600:])
for lib, v in res.items():
    if v:
        print("%-40s conv_ms %s   step_ms %s   %s" % (os.path.basename(lib), " ".join("%.4f" % a for a, _, _ in v),
                                                        " ".join("%.3f" % b for _, b, _ in v),
                                                        " ".join("%s %s" % (k, "/".join("%.4f" % x[2][k] for x in v)) for k in v[0][2])))
