#!/usr/bin/env python3
"""bench.py -- EPPM hot path on MI355X: Mflow-vectors/s on 1024x436 Sintel-shape pairs.

A step = one pass of the hot path (set_data's device part: prefilter + pyramid + census, then
compute_flow: PatchMatch fwd/bwd at 1/4 res, L-R check, outlier removal, weighted median, hole fill,
two coarse-to-fine levels, final smoothing) over ONE synthetic 1024x436 pair whose RGBA planes are
already resident in HBM.  Steps are issued round robin over --inflight contexts (default 3), each on its own HIP stream, so the
quarter-resolution stages of one pair (latency bound: too few pixels to fill 256 CUs) overlap the
full-resolution stages of another; every step's work runs inside the timed region and the single-pair
latency is reported next to the throughput.  N > 1: one process per GPU, each rank its own pairs
(independent pairs, no data-path collective: SURVEY 8e); value = pairs of all ranks * W*H / max-over-ranks time.

Prints one JSON line (see the task contract) with `roofline` (dominant kernel: the level-0 plane-fit
candidate refine, algorithmic HBM bytes / HIP-event duration) and `cpu_baseline` (the CPU oracle on a
bounded sample, rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
W, H = 1024, 436               # BASELINE.json configs[1]
# algorithmic HBM bytes per pixel of one k_c2f_refine launch (the dominant kernel; it runs once at level 1 and
# once at level 0 per pair): reads flow 8 + img1 4 + img2 4 + census1 1 + census2 1, writes flow 8
REFINE_BYTES_PER_PX = 26
# HBM-side bytes per k_c2f_refine_tiled launch at 1024x436, mean of the level-1 and level-0 launches, from the PMC
# passes committed under profiles/r01_g_pmc_{fetch,write}_size.csv: FETCH_SIZE 3621.8 / 11486.4 KB (x2: gfx950 tallies
# the 128-B requests of 16-B-per-lane loads at 64 B, MI355X_MICROARCH.md section HBM) + WRITE_SIZE 15710.4 / 3488.0 KB
# (the level-1 launch is split by affine pass and writes 36 costs per pixel instead of the flow)
TRAFFIC_BYTES_1024x436 = ((2 * 3621.8 + 15710.4) + (2 * 11486.4 + 3488.0)) / 2 * 1024
# The kernel is bound by vector-ALU issue, not by HBM: SQ_INSTS_VALU per launch (wave64 instructions) from
# profiles/r01_g_pmc_valu.csv, level-1 / level-0 launch; peak = 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 op
VALU_INSTS_1024x436 = (3.1347e+08 + 1.2386e+09) / 2
VALU_PEAK_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--patch-r", type=int, default=9)
    ap.add_argument("--inflight", type=int, default=3,
                    help="pairs in flight per GPU: steps are issued round robin over this many contexts, each on its own HIP stream")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per k_c2f_refine launch from a separate rocprofv3 --pmc pass (profiles/)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    if os.environ.get("EPPM_BENCH_SHARE_GPU"):      # test hook: several ranks on one GPU (gloo only)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = args.dist_backend
        if backend == "nccl":                        # RCCL: used only for the barrier and the MAX of the wall time
            try:
                dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
            except Exception as e:                   # the data path needs no collective: gloo is as good for timing
                print(f"[bench] nccl init failed ({e}); falling back to gloo", file=sys.stderr)
                backend = "gloo"
        if backend == "gloo":
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    tdev = dev if (world > 1 and dist.get_backend() == "nccl") else torch.device("cpu")

    import eppm_amd
    from eppm_amd import synth
    w, h = args.width, args.height
    params = eppm_amd.Params(patch_r=args.patch_r)
    S = max(1, args.inflight)
    engs = []
    for _ in range(S):
        e = eppm_amd.EPPM(device=local_rank, params=params)
        e.init(h, w)
        engs.append(e)
    eng = engs[0]

    # synthetic pairs of this rank (one per context), as RGBA planes resident in HBM before the timed region
    def to_dev(img):
        rgba = np.zeros((h, w, 4), np.uint8)
        rgba[..., :3] = img
        return torch.from_numpy(rgba).to(dev)
    inputs = []
    for j in range(S):
        img1, img2, gu_j, gv_j = synth.make_pair(h, w, seed=1234 + rank * S + j)
        if j == 0:
            gu, gv = gu_j, gv_j
        inputs.append((to_dev(img1), to_dev(img2), torch.empty((h, w, 2), dtype=torch.float32, device=dev)))
    d_flow = inputs[0][2]
    pitch = w * 4

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i):
        e = engs[i % S]
        a, b, f = inputs[i % S]
        e.set_data_device(a.data_ptr(), b.data_ptr(), pitch)
        e.compute_flow_device(f.data_ptr())

    def sync_all():
        for e in engs:
            e.synchronize()

    for i in range(max(args.warmup, S)):
        step(i)
    sync_all()
    # single-pair latency (one context, one stream, nothing else in flight): reported beside the throughput
    lat = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(0)
        engs[0].synchronize()
        lat.append((time.perf_counter() - t1) * 1e3)
    latency_ms = float(np.median(lat))
    eng.enable_stage_timing(True)
    eng.stage_times(clear=True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stages = eng.stage_times(clear=True)
    eng.enable_stage_timing(False)

    # sanity: the flow is finite and close to the synthetic ground truth (not a parity check)
    flow = d_flow.cpu().numpy()
    epe_gt = float(np.sqrt((flow[..., 0] - gu) ** 2 + (flow[..., 1] - gv) ** 2).mean())

    if rank == 0:
        agg = {}
        for name, ms in stages:
            agg.setdefault(name, []).append(ms)
        stage_ms = {k: float(np.mean(v)) for k, v in agg.items()}
        # dominant kernel = k_c2f_refine_tiled; per launch = mean over its two launches per pair (levels 1 and 0),
        # which is what rocprofv3 --stats averages for that kernel name
        lv = eng.level_dims()
        dom_ms = (stage_ms.get("c2f_refine_L0", float("nan")) + stage_ms.get("c2f_refine_L1", float("nan"))) / 2
        alg_bytes = REFINE_BYTES_PER_PX * (lv[0][0] * lv[0][1] + lv[1][0] * lv[1][1]) / 2
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        out = {
            "metric": "Mflow-vectors/sec", "value": world * args.steps * w * h / dt / 1e6, "unit": "Mflow-vectors/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"single {w}x{h} Sintel-shape synthetic pair per step, full 3-level pyramid, patch_r={args.patch_r}, "
                                   f"default defs.h parameters; {world} rank(s), independent pairs",
                       "pairs_per_step_per_gpu": 1, "pairs_in_flight_per_gpu": S, "width": w, "height": h},
            "roofline": {"bound": "hbm", "kernel": "k_c2f_refine_tiled (mean of its level-1 and level-0 launches)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": args.traffic_bytes if args.traffic_bytes is not None else (TRAFFIC_BYTES_1024x436 if (w, h, args.patch_r) == (W, H, 9) else None),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms},
            "valu_roofline": ({"kernel": "k_c2f_refine_tiled", "wave64_valu_insts_per_launch": VALU_INSTS_1024x436,
                               "achieved_insts_per_s": VALU_INSTS_1024x436 / (dom_ms * 1e-3), "peak_insts_per_s": VALU_PEAK_WAVE_INSTS_PER_S,
                               "frac": VALU_INSTS_1024x436 / (dom_ms * 1e-3) / VALU_PEAK_WAVE_INSTS_PER_S}
                              if (w, h, args.patch_r) == (W, H, 9) else None),
            "latency_ms_per_pair": latency_ms,
            "stage_ms": stage_ms,
            "epe_vs_synthetic_gt": epe_gt,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, h)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(w, h):
    """The CPU oracle (oracle/, a port of the reference's kernel semantics: the reference has no CPU path)
    timed on this host on a bounded sample: the workload's pair once (about 25 s on 8 cores, 6 s on 128)."""
    from oracle import oracle as O
    from eppm_amd import synth
    sw, sh = w, h
    a, b, _, _ = synth.make_pair(sh, sw, seed=1234)
    O.compute_flow(a[:32, :32].copy(), b[:32, :32].copy())      # build + warm
    t0 = time.perf_counter()
    O.compute_flow(a, b)
    dt = time.perf_counter() - t0
    return {"value": sw * sh / dt / 1e6, "unit": "Mflow-vectors/s", "cores": O.num_threads(), "kind": "port",
            "sample": f"1 pair {sw}x{sh} (the workload's own pair, seed 1234), whole path once, {dt:.1f} s, OpenMP oracle"}


if __name__ == "__main__":
    main()
