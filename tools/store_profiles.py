"""Copies the outputs of tools/gpu_round_end.sh (gpurun_out/<tag>/) into profiles/<tag>_* as per-kernel summaries and writes
profiles/pmc_constants.json: the PMC-derived per-pair constants of the dominant kernel, keyed by the sha256 of the kernel
sources they were measured on (bench.py emits null when the sources have changed since).  Usage: store_profiles.py r02_d"""
import csv, glob, hashlib, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
for f in glob.glob(f"{dst}/{tag}_*"):
    os.remove(f)
for name in ("bench_default.json", "bench_streams3.json", "bench_hd.json", "bench_4k_r17.json", "bench_under_rocprof.json", "bench_torchrun_2ranks_1gpu.json", "gpu_tests.txt"):
    if os.path.exists(f"{src}/{name}") and os.path.getsize(f"{src}/{name}") > 0:
        shutil.copy(f"{src}/{name}", f"{dst}/{tag}_{name}")
for d in ("stats_default", "stats_single"):
    f = glob.glob(f"{src}/{d}/*/*kernel_stats.csv")
    if f:
        shutil.copy(f[0], f"{dst}/{tag}_{d}_kernel_stats.csv")
summ = {}
for d in sorted(os.listdir(src)):
    if d.startswith("pmc_") and glob.glob(f"{src}/{d}/*/*counter_collection.csv"):
        out = subprocess.run([sys.executable, f"{ROOT}/tools/pmc_summary.py", f"{src}/{d}"], capture_output=True, text=True, check=True).stdout
        open(f"{dst}/{tag}_{d}.csv", "w").write(out)
        summ[d] = list(csv.DictReader(out.splitlines()))


def val(d, kernel, grid, counter):
    rows = [r for r in summ[d] if r["kernel"].startswith(kernel) and (grid is None or int(r["grid"]) == grid)]
    assert len(rows) == 1, (d, kernel, grid, [(r["kernel"], r["grid"]) for r in rows])
    return float(rows[0][counter])


W, H = 1024, 436
n0, n1 = W * H, (W // 2) * (H // 2)
tiles = lambda w, h: ((w + 15) // 16 + 0) * ((h + 15) // 16)
g_win = lambda w, h, n: ((tiles(w, h) + 7) // 8) * 8 * 512 * n            # threads of a k_c2f_refine_win launch
g_split4 = lambda w, h: ((tiles(w, h) + 7) // 8) * 8 * 4 * 256
per_pair = {}
for key, d_sfx, kern, grid, div in (("refine_win_L0", "single", "k_c2f_refine_win<9>", g_win(W, H, 1), 1),
                                    ("refine_split4_L1", "single", "k_c2f_refine_tiled<9, 4>", g_split4(W // 2, H // 2), 1),
                                    ("refine_win_L0_batch4", "batch4", "k_c2f_refine_win<9>", g_win(W, H, 4), 4),
                                    ("refine_win_L1_batch4", "batch4", "k_c2f_refine_win<9>", g_win(W // 2, H // 2, 4), 4)):
    per_pair[key] = {"valu_insts": val(f"pmc_sq_{d_sfx}", kern, grid, "SQ_INSTS_VALU") / div,
                     "fetch_size_kb": val(f"pmc_fetch_{d_sfx}", kern, grid, "FETCH_SIZE") / div,
                     "write_size_kb": val(f"pmc_write_{d_sfx}", kern, grid, "WRITE_SIZE") / div,
                     "avg_us_under_pmc": val(f"pmc_sq_{d_sfx}", kern, grid, "avg_us") / div}
srcs = ["eppm_amd/csrc/k_c2f.hip", "eppm_amd/csrc/eppm_device.cuh"]
sha = hashlib.sha256(b"".join(open(os.path.join(ROOT, f), "rb").read() for f in srcs)).hexdigest()
# the whole path: wave64 VALU instructions of EVERY kernel per pair, from the 4-pairs-per-launch SQ pass (k_pm_init_field runs
# once per batch: its call count gives the number of batches the profiled command processed); keyed by ALL device sources
all_srcs = sorted(os.path.relpath(f, ROOT) for f in glob.glob(f"{ROOT}/eppm_amd/csrc/*.hip") + glob.glob(f"{ROOT}/eppm_amd/csrc/*.cuh"))
sha_all = hashlib.sha256(b"".join(open(os.path.join(ROOT, f), "rb").read() for f in all_srcs)).hexdigest()
rows4 = summ["pmc_sq_batch4"]
batches = sum(float(r["calls"]) for r in rows4 if "k_pm_init_field" in r["kernel"])
path = {"sources_sha256": sha_all, "kernel_sources": all_srcs, "pairs_per_launch": 4,
        "valu_insts_per_pair": sum(float(r["SQ_INSTS_VALU"]) * float(r["calls"]) for r in rows4) / (batches * 4),
        "kernel_us_per_pair_one_context": sum(float(r["avg_us"]) * float(r["calls"]) for r in rows4) / (batches * 4)}
entry = {"width": W, "height": H, "patch_r": 9, "source": f"profiles/{tag}_pmc_*.csv (tools/gpu_round_end.sh, tools/store_profiles.py)",
         "kernel_sources": srcs, "per_pair": per_pair, "path": path,
         "note": "FETCH_SIZE counts the 128-B requests of 16-B-per-lane loads at 64 B on gfx950 (MI355X_MICROARCH.md, HBM): traffic = 2*FETCH + WRITE"}
json.dump({sha: entry}, open(f"{dst}/pmc_constants.json", "w"), indent=1)
print(json.dumps(per_pair, indent=1))
