"""GPU busy fraction from a rocprofv3 kernel trace: python tools/busy.py <kernel_trace.csv> [last_n_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):] if len(sys.argv) > 2 else rows
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = iv[-1][1] - iv[0][0]
print("kernels %d  span %.3f ms  busy %.3f ms  idle %.1f %%  sum of kernel times %.3f ms" % (
    len(rows), span / 1e6, busy / 1e6, 100 * (1 - busy / span), sum(e - s for s, e in iv) / 1e6))
