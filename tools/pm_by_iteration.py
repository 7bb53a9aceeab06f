"""Every PatchMatch launch of ONE run (the last complete one in a rocprofv3 --kernel-trace directory) in order: kernel, duration in us.
usage: pm_by_iteration.py DIR"""
import csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_pm_init_field" in r["Kernel_Name"]]
a = starts[-1]
it = -1
for r in rows[a:]:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("eppm::", "")
    if not n.startswith("k_pm_"):
        break
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{n[:70]:70s} {us:8.1f}")
