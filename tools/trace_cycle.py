"""Per-launch durations of one weak cycle from a rocprofv3 kernel trace (tools/trace_weak.sh)."""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |brklib::", "", n).split("(")[0][:28]
ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"])) for r in rows]
idx = [i for i, k in enumerate(ks) if "link_copy" in k[0]]
lo, hi = idx[-7], idx[-6]          # the bench appends 5 exchange-only launches after the timed loop
out = []
for i in range(lo, hi):
    k = ks[i]
    waves = k[3] // 64
    out.append(f"{k[0]:30s} waves={waves:8d} dur_us={(k[2]-k[1])/1e3:9.1f} GB/s_at_8KiB_per_wave={waves*8192/((k[2]-k[1])*1e-9)/1e9:8.0f} "
               f"gap_before_us={(k[1]-ks[i-1][2])/1e3:7.1f}")
out.append(f"cycle span us = {(ks[hi][1]-ks[lo][1])/1e3:.1f}")
open(sys.argv[1] + "/cycle.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
