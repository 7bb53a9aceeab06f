"""Copies the outputs of tools/gpu.sh round (gpurun_out/<tag>/) into profiles/<tag>_* as per-kernel summaries and writes
profiles/pmc_constants.json: PMC-derived constants per shape -- the dominant kernel (the candidate refine) per launch size and the
whole path per kernel group -- each valid only for the device sources it was measured on (sha256 inside; bench.py emits null
when they have changed since).  Usage: store_profiles.py r03_x"""
import csv, glob, hashlib, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
for f in glob.glob(f"{dst}/{tag}_*"):
    os.remove(f)
for name in ("bench_default.json", "bench_like_driver.json", "bench_streams3.json", "bench_under_rocprof.json", "bench_tol_under_rocprof.json", "bench_torchrun_2ranks_1gpu.json",
             "gpu_tests.txt", "tolerance_epe.json"):
    if os.path.exists(f"{src}/{name}") and os.path.getsize(f"{src}/{name}") > 0:
        shutil.copy(f"{src}/{name}", f"{dst}/{tag}_{name}")
for d in ("stats_default", "stats_single", "stats_batch8", "stats_tol_default", "stats_tol_batch8"):
    # the bench process's file: the one with kernel rows (child processes that launch no kernel leave empty or no stats files)
    f = [x for x in glob.glob(f"{src}/{d}/*/*kernel_stats.csv") if "k_" in open(x).read()]
    if f:
        # gpurun MERGES a call's outputs into gpurun_out/: a directory may still hold an earlier round's files (other process ids) -- the newest run counts
        newest = max(os.path.getmtime(x) for x in f)
        shutil.copy(max((x for x in f if newest - os.path.getmtime(x) < 300), key=os.path.getsize), f"{dst}/{tag}_{d}_kernel_stats.csv")
summ = {}
for d in sorted(os.listdir(src)):
    if d.startswith("pmc_") and glob.glob(f"{src}/{d}/*/*counter_collection.csv"):
        out = subprocess.run([sys.executable, f"{ROOT}/tools/pmc_summary.py", f"{src}/{d}"], capture_output=True, text=True, check=True).stdout
        open(f"{dst}/{tag}_{d}.csv", "w").write(out)
        summ[d] = list(csv.DictReader(out.splitlines()))

VALU_PEAK = 256 * 4 * 2.4e9 / 2
GROUPS = (("refine", ("k_c2f_refine", "k_c2f_select")), ("sweeps", ("k_pm_sweep", "k_pm_spec_all", "k_pm_seg_propagate")), ("search", ("k_pm_random_search",)),
          ("smoothing", ("k_flow_blf",)), ("weighted_median", ("k_wmf",)))


def group_of(kernel):
    for g, pref in GROUPS:
        if kernel.startswith(pref):
            return g
    return "other"


def pairs_of(rows, nb):
    """pairs the profiled command processed: k_pm_init_field runs once per launch sequence of nb pairs"""
    return sum(float(r["calls"]) for r in rows if "k_pm_init_field" in r["kernel"]) * nb


def per_pair(label, nb, kernels, counter, d_prefix):
    rows = summ[f"{d_prefix}_{label}"]
    return sum(float(r[counter]) * float(r["calls"]) for r in rows if r["kernel"].startswith(kernels)) / pairs_of(rows, nb)


all_srcs = sorted(os.path.relpath(f, ROOT) for f in glob.glob(f"{ROOT}/eppm_amd/csrc/*.hip") + glob.glob(f"{ROOT}/eppm_amd/csrc/*.cuh"))
sha_all = hashlib.sha256(b"".join(open(os.path.join(ROOT, f), "rb").read() for f in all_srcs)).hexdigest()
REF = ("k_c2f_refine", "k_c2f_select")


def have(prefix, label):
    return f"{prefix}_{label}" in summ


def lds_of(label, nb):
    """LDS-array cycles of the dominant kernel per pair (SQ_LDS_IDX_ACTIVE, of which SQ_LDS_BANK_CONFLICT are conflict cycles), summed over all CUs"""
    if not have("pmc_sq2", label):
        return None
    return {"lds_idx_active": per_pair(label, nb, REF, "SQ_LDS_IDX_ACTIVE", "pmc_sq2"), "lds_bank_conflict": per_pair(label, nb, REF, "SQ_LDS_BANK_CONFLICT", "pmc_sq2"),
            "us_under_pmc": per_pair(label, nb, REF, "avg_us", "pmc_sq2")}


def shapes_of(pre):
    shapes = {}
    for key, labels in (("1024x436_r9", (("single", 1), ("batch8", 8))), ("1920x1080_r9", (("hd", 1),)), ("3840x2160_r17", (("uhd17", 1),))):
        labels = tuple((pre + lb, nb) for lb, nb in labels)
        if not all(have("pmc_sq", lb) for lb, _ in labels):
            continue
        dominant = {}
        for lb, nb in labels:
            # per PAIR, summed over the kernel's level-1 and level-0 launches
            dominant[str(nb)] = {"valu_insts": per_pair(lb, nb, REF, "SQ_INSTS_VALU", "pmc_sq"),
                                 "fetch_size_kb": per_pair(lb, nb, REF, "FETCH_SIZE", "pmc_fetch") if have("pmc_fetch", lb) else None,
                                 "write_size_kb": per_pair(lb, nb, REF, "WRITE_SIZE", "pmc_write") if have("pmc_write", lb) else None,
                                 "us_under_pmc": per_pair(lb, nb, REF, "avg_us", "pmc_sq"), "lds": lds_of(lb, nb)}
        lb, nb = labels[-1]
        rows = summ[f"pmc_sq_{lb}"]
        np_ = pairs_of(rows, nb)
        by = {}
        for r in rows:
            g = by.setdefault(group_of(r["kernel"]), {"valu_insts_per_pair": 0.0, "kernel_us_per_pair": 0.0})
            g["valu_insts_per_pair"] += float(r["SQ_INSTS_VALU"]) * float(r["calls"]) / np_
            g["kernel_us_per_pair"] += float(r["avg_us"]) * float(r["calls"]) / np_
        for g in by.values():
            g["valu_issue_frac_under_pmc"] = g["valu_insts_per_pair"] / (g["kernel_us_per_pair"] * 1e-6) / VALU_PEAK if g["kernel_us_per_pair"] else None
        shapes[key] = {"source": f"profiles/{tag}_pmc_*{pre}*.csv (tools/gpu.sh round, tools/store_profiles.py)", "kernel_sources": all_srcs, "sources_sha256": sha_all,
                       "dominant": dominant,
                       "path": {"pairs_per_launch": nb, "valu_insts_per_pair": sum(g["valu_insts_per_pair"] for g in by.values()),
                                "kernel_us_per_pair_one_context": sum(g["kernel_us_per_pair"] for g in by.values()), "by_kernel_group": by}}
    return shapes


shapes, shapes_tol = shapes_of(""), shapes_of("tol_")
json.dump({"note": "FETCH_SIZE counts the 128-B requests of 16-B-per-lane loads at 64 B on gfx950 (MI355X_MICROARCH.md, HBM): traffic = 2*FETCH + WRITE; "
                   "dominant: the candidate refine per PAIR (level 1 + level 0) by pairs per launch; path: every kernel, one context",
           "shapes": shapes, "shapes_tol": shapes_tol}, open(f"{dst}/pmc_constants.json", "w"), indent=1)
print(json.dumps({lib: {k: {"dominant": v["dominant"], "path_insts": v["path"]["valu_insts_per_pair"],
                            "groups": {g: round(x["valu_issue_frac_under_pmc"] or 0, 3) for g, x in v["path"]["by_kernel_group"].items()}} for k, v in sh.items()}
                  for lib, sh in (("exact", shapes), ("tol", shapes_tol))}, indent=1))
