import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_langevin_update" in n]
a, b = idx[12], idx[13]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    print("%8.1f %8.1f  s%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Stream_Id"], r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]))
