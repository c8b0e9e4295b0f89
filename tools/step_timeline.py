#!/usr/bin/env python3
"""Kernel timeline of ONE denoising step from a rocprofv3 --kernel-trace run of bench.py (start, duration, stream, kernel),
plus the main stream's busy time and the gaps between its launches.
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $REPO/bench.py --steps 30 --warmup 5 \
        --no-cpu-baseline --no-extra --no-traj;  python3 tools/step_timeline.py /tmp/kt"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_sampler_front" in n] or [i for i, n in enumerate(names) if "k_langevin_update" in n]
a, b = idx[12], idx[13]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    print("%8.1f %8.1f  s%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Stream_Id"], r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]))
# gaps on the main stream inside that step
main = [r for r in rows[a:b + 1] if r["Stream_Id"] == rows[a]["Stream_Id"]]
gap = sum(max(0, int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) for x, y in zip(main[:-1], main[1:]))
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in main[:-1])
print("step %.1f us; main-stream kernels %.1f us; gaps between them %.1f us (%d launches)" % ((int(main[-1]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3, gap / 1e3, len(main) - 1))
