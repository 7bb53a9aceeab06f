"""Copies the outputs of tools/gpu_round_end.sh (gpurun_out/) into profiles/<tag>_* and refreshes the constants and numbers
that are derived from them (bench.py PMC constants, profiles/README.md row, README.md headline).  Usage: store_profiles.py r01_g"""
import shutil, glob, os, csv, json, collections, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_g"
def latest(p): return sorted(glob.glob(p), key=os.path.getmtime)[-1]
for f in glob.glob(f'profiles/{tag}_*'): os.remove(f)
shutil.copy('gpurun_out/bench_default.json', f'profiles/{tag}_default_bench.json')
shutil.copy('gpurun_out/prof_default.json', f'profiles/{tag}_default_bench_under_rocprof.json')
for src, dst in (('gpurun_out/bench_hd.json', f'profiles/{tag}_bench_1920x1080.json'), ('gpurun_out/bench_4k.json', f'profiles/{tag}_bench_3840x2160_R17.json')):
    if os.path.exists(src): shutil.copy(src, dst)
shutil.copy(latest('gpurun_out/prof_default/*/*kernel_stats.csv'), f'profiles/{tag}_default_bench_kernel_stats.csv')
vals = {}
for c, src, dst in (('SQ_INSTS_VALU', 'gpurun_out/pmc_valu', f'profiles/{tag}_pmc_valu.csv'), ('FETCH_SIZE', 'gpurun_out/pmc_fetch', f'profiles/{tag}_pmc_fetch_size.csv'),
                    ('WRITE_SIZE', 'gpurun_out/pmc_write', f'profiles/{tag}_pmc_write_size.csv')):
    rows = list(csv.DictReader(open(latest(src + '/*/*counter_collection.csv'))))
    keep = [r for r in rows if 'c2f_refine_tiled' in r['Kernel_Name'] or 'k_flow_blf' in r['Kernel_Name'] or 'k_c2f_select' in r['Kernel_Name']]
    w = csv.DictWriter(open(dst, 'w', newline=''), fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
    d = collections.defaultdict(list)
    for r in keep:
        if r['Counter_Name'] == c and 'c2f_refine_tiled' in r['Kernel_Name']:
            d['L0' if '9, 0>' in r['Kernel_Name'] else 'L1'].append(float(r['Counter_Value']))
    vals[c] = {k: sum(v) / len(v) for k, v in d.items()}
f1, f0 = vals['FETCH_SIZE']['L1'], vals['FETCH_SIZE']['L0']; w1, w0 = vals['WRITE_SIZE']['L1'], vals['WRITE_SIZE']['L0']; v1, v0 = vals['SQ_INSTS_VALU']['L1'], vals['SQ_INSTS_VALU']['L0']
rows = list(csv.DictReader(open(f'profiles/{tag}_default_bench_kernel_stats.csv')))
k0 = [r for r in rows if 'c2f_refine_tiled<9, 0>' in r['Name']][0]; k1 = [r for r in rows if 'c2f_refine_tiled<9, ' in r['Name'] and '9, 0>' not in r['Name']][0]
d = json.load(open(f'profiles/{tag}_default_bench.json')); u = json.load(open(f'profiles/{tag}_default_bench_under_rocprof.json'))
hd = json.load(open(f'profiles/{tag}_bench_1920x1080.json')); k4 = json.load(open(f'profiles/{tag}_bench_3840x2160_R17.json'))
print(d['value'], d['ms_per_step'], d['latency_ms_per_pair'], hd['value'], hd['ms_per_step'], k4['ms_per_step'], k0['AverageNs'], k1['AverageNs'], u['roofline']['avg_launch_ms'])
s = open('bench.py').read()
s = re.sub(r"FETCH_SIZE [0-9.]+ / [0-9.]+ KB \(x2", "FETCH_SIZE %.1f / %.1f KB (x2" % (f1, f0), s)
s = re.sub(r"WRITE_SIZE [0-9.]+ / [0-9.]+ KB\n", "WRITE_SIZE %.1f / %.1f KB\n" % (w1, w0), s)
s = re.sub(r"TRAFFIC_BYTES_1024x436 = .*?\n", "TRAFFIC_BYTES_1024x436 = ((2 * %.1f + %.1f) + (2 * %.1f + %.1f)) / 2 * 1024\n" % (f1, w1, f0, w0), s)
s = re.sub(r"VALU_INSTS_1024x436 = \([0-9.e+]+ \+ [0-9.e+]+\) / 2", "VALU_INSTS_1024x436 = (%.4e + %.4e) / 2" % (v1, v0), s)
open('bench.py', 'w').write(s)
t = open('profiles/README.md').read()
t = re.sub(r"\| `%s_\*` \|.*?\n" % tag, "| `%s_*` | **final state of round 1**: default bench line (120 timed steps) %.1f Mflow-vectors/s (%.2f ms/step, latency %.2f ms/pair); kernel stats of the same command (`k_c2f_refine_tiled<9,0>` — level 0 — %.3f ms and `<9,4>` — level 1, split by affine pass — %.3f ms on average over %s launches each; the bench's own event average over the timed launches of that profiled run, mean of the two levels incl. `k_c2f_select`: %.3f ms); PMC passes (single stream, rows of `k_c2f_refine_tiled`, `k_c2f_select`, `k_flow_blf` only): SQ_INSTS_VALU %.3e / %.4e wave64 instructions, FETCH_SIZE %.0f / %.0f KB, WRITE_SIZE %.0f / %.0f KB per level-1 / level-0 launch; bench lines of the 1920×1080 pair (%.1f Mflow-vectors/s, %.2f ms/step) and of the 3840×2160 pair at patch radius 17 (%.1f ms/pair) | `tools/gpu_round_end.sh`, `tools/store_profiles.py` |\n" % (tag, d['value'], d['ms_per_step'], d['latency_ms_per_pair'], float(k0['AverageNs']) / 1e6, float(k1['AverageNs']) / 1e6, k0['Calls'], u['roofline']['avg_launch_ms'], v1, v0, f1, f0, w1, w0, hd['value'], hd['ms_per_step'], k4['ms_per_step']), t)
open('profiles/README.md', 'w').write(t)
t = open('README.md').read()
t = re.sub(r"# [0-9.]+ Mflow-vectors/s on one MI355X \(round 1\)", "# %.1f Mflow-vectors/s on one MI355X (round 1)" % d['value'], t)
open('README.md', 'w').write(t)
print("DESIGN.md numbers to check by hand: default %.2f ms/step, latency %.2f, %.1f Mvec/s; HD %.1f ms/step %.1f Mvec/s lat %.1f; 4K %.0f ms (pm %.0f, refine %.0f)" % (
    d['ms_per_step'], d['latency_ms_per_pair'], d['value'], hd['ms_per_step'], hd['value'], hd['latency_ms_per_pair'], k4['ms_per_step'], k4['stage_ms']['patchmatch'],
    k4['stage_ms']['c2f_refine_L1'] + k4['stage_ms']['c2f_refine_L0']))
