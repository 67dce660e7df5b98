"""Timeline of the last steps of tools/dev/overlap_run.py from a rocprofv3 --kernel-trace CSV: start / end of every kernel relative
to the step kernel's first start, with the queue it ran on."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "npp_step_kernel" in r["Kernel_Name"]]
# steps = groups starting at a step kernel whose predecessor is not a step kernel
starts = [i for j, i in enumerate(idx) if j == 0 or idx[j - 1] != i - 1]
for s in starts[-4:-1]:
    e = starts[starts.index(s) + 1]
    t0 = int(rows[s]["Start_Timestamp"])
    print("---- step")
    for r in rows[s:e]:
        nm = r["Kernel_Name"].split("(")[0].replace("void npp::(anonymous namespace)::", "")[:28]
        print("%-28s q%-3s %8.1f %8.1f  grid %s" % (nm, r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "")))
    print("total %.1f" % ((int(rows[e]["Start_Timestamp"]) - t0) / 1e3))
