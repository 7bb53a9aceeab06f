"""Per-kernel summary of a rocprofv3 --pmc counter_collection.csv: mean of each counter per dispatch, grouped by kernel name
(+ grid size, so the level-1 and level-0 launches of one kernel stay apart), and the mean duration from the kernel trace.
usage: pmc_summary.py DIR [name-filter]"""
import csv, glob, os, sys, collections, re
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cc = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)       # the newest run's (gpurun merges into directories that may hold older files)
rows = list(csv.DictReader(open(cc)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void eppm::", "")
    key = (name, r["Grid_Size"])
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
kt = cc.replace("counter_collection.csv", "kernel_trace.csv")                          # the same process's trace
if os.path.exists(kt):
    for r in csv.DictReader(open(kt)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void eppm::", "")
        g = str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
        dur[(name, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ctrs = sorted({c for v in agg.values() for c in v})
w = csv.writer(sys.stdout, lineterminator="\n")
w.writerow(["kernel", "grid", "calls", "avg_us"] + ctrs)
for key in sorted(agg, key=lambda k: -sum(dur.get(k, [0]))):
    if flt and flt not in key[0]:
        continue
    v = agg[key]
    n = len(next(iter(v.values())))
    du = dur.get(key, [0])
    w.writerow([key[0], key[1], n, f"{sum(du)/max(1,len(du)):.1f}"] + [f"{sum(v[c])/len(v[c]):.4g}" if c in v else "" for c in ctrs])
