"""Counters of every dispatch of the kernels matching a filter, in launch order, with the duration from the kernel trace of the
same run (rocprofv3 --kernel-trace --pmc ...).  usage: pmc_by_dispatch.py DIR name-filter [first_n]"""
import csv, glob, sys, collections, re
d, flt = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    per.setdefault(int(r["Dispatch_Id"]), {"name": re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void eppm::", "")})[r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ctrs = sorted({c for v in per.values() for c in v if c != "name"})
print("dispatch kernel us " + " ".join(ctrs))
k = 0
for i in sorted(per):
    v = per[i]
    if flt in v["name"]:
        print(i, v["name"][:60], f"{dur.get(i, 0):.1f}", " ".join(f"{v.get(c, 0):.4g}" for c in ctrs))
        k += 1
        if k >= n: break
