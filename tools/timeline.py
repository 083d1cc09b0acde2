"""Print the kernel timeline of one steady-state bench step from a rocprofv3 --kernel-trace CSV (GPU box tool)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# steps are delimited by k_adam( launches
adam = [i for i, n in enumerate(names) if "k_adam(" in n]
if len(adam) < 3:
    print("not enough steps", len(adam))
    sys.exit(0)
k = int(sys.argv[2]) if len(sys.argv) > 2 else -3  # which step (index into the k_adam launches)
a, b = adam[k], adam[k + 1]
t0 = int(rows[a]["End_Timestamp"])
prev_end = t0
busy = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  gap %7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
                                                 r["Kernel_Name"][:90]))
    busy += e - s
    prev_end = max(prev_end, e)
print("step span %.1f us, kernel busy %.1f us, %d kernels" % ((prev_end - t0) / 1e3, busy / 1e3, b - a))
