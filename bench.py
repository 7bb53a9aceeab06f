#!/usr/bin/env python3
"""bench.py -- EPPM hot path on MI355X: Mflow-vectors/s (`value`, `ms_per_step`) on 1024x436 Sintel-shape pairs, and the single-pair
latency beside it (`latency_ms_per_pair`: one pair, one stream, nothing else in flight -- the "ms/frame-pair" half of BASELINE.json's metric).

A step = one pass of the hot path (set_data's device part: prefilter + pyramid + census, then
compute_flow: PatchMatch fwd/bwd at 1/4 res, L-R check, outlier removal, weighted median, hole fill,
two coarse-to-fine levels, final smoothing) over ONE synthetic 1024x436 pair whose RGBA planes are
already resident in HBM; the float2 flow stays in HBM.  Consecutive steps are issued in groups of --batch (default 8) to
batch contexts (eppm_create_batch: every kernel launch covers the group's pairs -- the quarter-resolution stages of ONE pair
are too few pixels to fill 256 CUs), round robin over --inflight such contexts (default 3), each on its own HIP
stream, so that the tail of one launch overlaps the next context's work; every step's work runs inside the timed
region; `--batch 1` issues one context per step.  The timed window of exactly --steps steps is run --repeats (5) times,
each bracketed by barrier + synchronize; the median is reported, min / max beside it.

The line proves its own correctness: the pairs of the timed region are pairs (rank * P + j) mod 64 of BASELINE configs[2]'s 64 pairs
(seeds 1234 + index), whose CPU-oracle flows are committed as sha256 (tests/golden/MANIFEST_config3.json); after the last repeat
every flow the timed region wrote is hashed and compared: `timed_region_verified: {ok, of}` at every N.

N > 1 (`--gpus N`): one process per GPU, each rank its own pairs (independent pairs, no data-path
collective: SURVEY 8e); value = pairs of all ranks * W*H / max-over-ranks time.  When RANK is not in the
environment (plain `python bench.py --gpus N`) this process only spawns the N rank processes -- before
importing torch or touching HIP -- relays rank 0's JSON line and exits non-zero if any rank failed; under
`torch.distributed.run` (RANK set) it is a rank itself.  At N > 1 (or with `--verify-config3`) every rank also runs its share of
configs[2]'s 64 pairs (pair i -> rank i mod N) through the host boundary and checks each flow against the same hashes
(`config3_verified`; a pair whose synthetic images differ on this host counts as unverified, never as OK).

Prints ONE JSON line: the contract fields, `timed_region_verified`, `roofline` (dominant kernel = the candidate refine; its launch
duration by HIP events from a one-context pass of the same launches; bound "valu" with the HBM form beside it, from
profiles/pmc_constants.json when that was measured on these device sources), `path_valu_roofline` (the whole path against the VALU
issue peak) and, on rank 0 at N = 1: `latency_ms_per_pair`, `stage_ms` and `valu_roofline` from a single-stream pass, `other_configs`
(BASELINE configs[3] 1920x1080 and configs[4] 3840x2160 at patch radius 17: value, ms per pair, fraction of the VALU floor, flow
checked against the committed oracle hash), `host_boundary` (PCIe-inclusive rates of the reference API's own window -- host RGB in,
host u/v out -- synchronous, batched and pipelined; tools/host_boundary.py), `cold_ms` (init + compute_flow, the window
main.cpp:63-66 times), `config3` (8 distinct pairs per GPU: BASELINE.json configs[2]) and `cpu_baseline` (the CPU oracle on a bounded
sample).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
W, H = 1024, 436               # BASELINE.json configs[1]
# algorithmic HBM bytes per pixel of one k_c2f_refine launch (the dominant kernel; it runs once at level 1 and
# once at level 0 per pair): reads flow 8 + img1 4 + img2 4 + census1 1 + census2 1, writes flow 8
REFINE_BYTES_PER_PX = 26
VALU_PEAK_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2      # 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_constants.json")


def parse_args(known_only=False):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--patch-r", type=int, default=9)
    ap.add_argument("--propagation", type=int, default=0, choices=[0, 1, 2],
                    help="0: the reference's live segmented sweeps (baoSegPropagate; the parity default); 1: jump flood (baoJumpPropagate, disabled in the "
                         "reference); 2: 4-neighbour propagation (10 x baoParallelPropagate, disabled).  Information only: 1 and 2 compute other flows")
    ap.add_argument("--inflight", type=int, default=3,
                    help="pairs in flight per GPU: steps are issued round robin over this many contexts, each on its own HIP stream")
    ap.add_argument("--batch", type=int, default=8,
                    help="pairs per launch sequence: > 1 groups consecutive steps into batch contexts (eppm_create_batch) whose every kernel launch "
                         "covers the whole group; --inflight such contexts are kept in flight")
    ap.add_argument("--pairs-per-gpu", type=int, default=8, help="size of the config-3 leg (distinct pairs per GPU)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed window of exactly --steps steps is run this many times back to back; the median is reported (min/max beside it)")
    ap.add_argument("--verify-config3", dest="verify_config3", action="store_true", default=None,
                    help="BASELINE configs[2] with a correctness bit: this rank's share of the 64 pairs (pair i -> rank i mod N, seeds 1234+i) through the "
                         "host boundary, every flow checked against tests/golden/MANIFEST_config3.json.  Default: on when --gpus > 1")
    ap.add_argument("--no-verify-config3", dest="verify_config3", action="store_false")
    ap.add_argument("--library", default="tol" if os.environ.get("EPPM_HIP_VARIANT") == "tol" else "exact", choices=["exact", "tol"],
                    help="exact: libeppm_hip.so, bit-identical to the oracle (the default, and what `value` always is); tol: the tolerance library "
                         "libeppm_hip_tol.so (not bit-identical, <= 1e-3 px EPE on the bundled pair) -- the run the default line embeds as `tolerance_mode`")
    ap.add_argument("--no-tolerance-mode", action="store_true", help="skip the tolerance_mode leg (a child run of this script with --library tol)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip host_boundary / cold / config3 / single-stream / other-config legs (profiling runs)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the 1920x1080 and 3840x2160 R=17 legs (other_configs)")
    return ap.parse_known_args()[0] if known_only else ap.parse_args()


def parse_args_known():
    return parse_args(known_only=True)


# ---------------------------------------------------------------------------------------------------
# N > 1 without a launcher: the parent spawns the ranks (it never imports torch nor touches the GPU)
# ---------------------------------------------------------------------------------------------------
def spawn_ranks(args, script=None):
    """script: the rank program (default: this file); tests substitute a stub."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    line = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if line:
        print(line[-1], flush=True)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad or not line:
        print(f"[bench] ranks failed: {bad}" if bad else "[bench] rank 0 printed no JSON line", file=sys.stderr)
        sys.exit(1)


def _sha(files):
    return hashlib.sha256(b"".join(open(os.path.join(ROOT, f), "rb").read() for f in files)).hexdigest()


LIBRARY = "exact"          # which library this process computes with (worker() sets it from --library)


def pmc_constants(w, h, patch_r):
    """PMC-derived constants for this shape and library (profiles/pmc_constants.json, written by tools/store_profiles.py from separate
    rocprofv3 --pmc passes): valid only for the device sources they were measured on (sha256 inside the entry); None otherwise."""
    try:
        e = json.load(open(PMC_FILE))["shapes" if LIBRARY == "exact" else "shapes_" + LIBRARY].get(f"{w}x{h}_r{patch_r}")
        if e and _sha(e["kernel_sources"]) == e["sources_sha256"]:
            return e
    except Exception:
        pass
    return None


def path_valu_roofline(pmc, nb, s_per_step):
    """The WHOLE path against the bound that limits it: wave64 VALU instructions of every kernel per pair (SQ_INSTS_VALU summed
    over all dispatches of a one-context counter pass) / the nominal issue peak = the time per pair below which this
    instruction stream cannot run."""
    try:
        p = pmc["path"]
        floor_ms = p["valu_insts_per_pair"] / VALU_PEAK_WAVE_INSTS_PER_S * 1e3
        return {"bound": "valu", "scope": f"every kernel of one pair (set_data device part + compute_flow), counters taken at {p['pairs_per_launch']} pair(s) per launch",
                "wave64_valu_insts_per_pair": p["valu_insts_per_pair"], "peak_insts_per_s": VALU_PEAK_WAVE_INSTS_PER_S,
                "floor_ms_per_pair": floor_ms, "ms_per_step": s_per_step * 1e3, "frac": floor_ms / (s_per_step * 1e3),
                "by_kernel_group": p.get("by_kernel_group"),
                "note": "nominal peak = 256 CUs x 4 SIMD x 2.4 GHz / 2 cycles per wave64 instruction; the clock under this load is "
                        "2.2-2.3 GHz and about a tenth of the instructions are half rate", "source": pmc.get("source")}
    except Exception:
        return None


def dominant_pmc(pmc, nb):
    """per-launch PMC record {valu_insts, fetch_size_kb, write_size_kb} of the dominant kernel (sum over its level-1 and level-0
    launches, per PAIR) for launches of nb pairs, or None"""
    try:
        return pmc["dominant"][str(nb)]
    except Exception:
        return None


def agree_on_group(dist, rank, probe):
    """The data path needs no collective; the process group only carries the barrier and one MAX.  The gloo world group (already
    initialised) is the control plane; `probe` (None, or a callable that builds the RCCL group and proves it with an all-reduce)
    runs on top of it, and the ranks AGREE -- a MIN over gloo -- before any of them uses the result: a rank whose RCCL fails can
    never sit in a gloo barrier while the others wait in an RCCL one.  Returns (control group, group for barrier / MAX)."""
    import torch
    ctl = dist.group.WORLD
    grp, ok = None, 0
    if probe is not None:
        try:
            grp = probe()
            ok = 1
        except Exception as e:
            print(f"[bench] rank {rank}: nccl unusable ({e})", file=sys.stderr)
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctl)
    if not int(flag.item()):
        if probe is not None and rank == 0:
            print("[bench] RCCL not usable on every rank; barrier and MAX go over gloo", file=sys.stderr)
        grp = ctl
    return ctl, grp


def worker(args):
    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before the HIP runtime starts: RCCL needs dmabuf IPC on this pool
    # stdout carries the ONE JSON line and nothing else: gloo and RCCL print connection banners on stdout (every rank's, merged under
    # torch.distributed.run), so file descriptor 1 points at stderr for the life of the rank and the line is written to the saved one
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    verify3 = (world > 1) if args.verify_config3 is None else bool(args.verify_config3)
    other_cfgs = world == 1 and not args.no_extras and not args.no_other_configs and (args.width, args.height, args.patch_r, args.propagation) == (W, H, 9, 0)
    S = max(1, args.inflight)
    NB = max(1, args.batch)
    NC3 = 0 if args.no_extras else args.pairs_per_gpu        # pairs of the config-3 leg
    NP = max(S * max(1, args.batch), NC3)
    plan = InputPlan(args, rank, world, NP, verify3, other_cfgs)
    plan.generate()                 # worker processes, before this process starts the GPU runtime
    global LIBRARY
    LIBRARY = args.library
    import eppm_amd
    eppm_amd.select_library("" if args.library == "exact" else args.library)      # before the first call into the library
    import torch
    import torch.distributed as dist
    if os.environ.get("EPPM_BENCH_SHARE_GPU"):      # test hook: several ranks on one GPU (gloo only)
        local_rank = 0
    ndev = torch.cuda.device_count()
    if ndev and local_rank >= ndev:                 # a launcher that hands every rank one visible device (HIP_VISIBLE_DEVICES per rank)
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    host_cpus = os.sched_getaffinity(0)            # restored before the CPU baseline
    numa = bind_to_gpu_numa(local_rank)            # this rank's host thread(s) next to its GPU
    ctl = None                      # gloo group: control plane (backend agreement); `grp` carries the barrier and the MAX of the wall time
    grp = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

        def probe_rccl():
            import datetime
            g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120), device_id=dev)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe, group=g)          # communicators are created lazily: fail here, not inside the timed region
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError(f"all_reduce probe returned {probe.item()} for {world} ranks")
            return g
        ctl, grp = agree_on_group(dist, rank, probe_rccl if args.dist_backend == "nccl" else None)
    tdev = dev if (world > 1 and grp is not ctl) else torch.device("cpu")

    import eppm_amd
    from eppm_amd import synth
    w, h = args.width, args.height
    params = eppm_amd.Params(patch_r=args.patch_r, propagation=args.propagation)
    engs = []
    for _ in range(S):
        e = eppm_amd.EPPM(device=local_rank, params=params)
        e.init(h, w)
        engs.append(e)
    eng = engs[0]
    bengs = [eppm_amd.EPPMBatch(h, w, NB, device=local_rank, params=params) for _ in range(S)] if NB > 1 else []

    # synthetic pairs of this rank, as RGBA planes resident in HBM before any timed region
    def to_dev(img):
        rgba = np.zeros((h, w, 4), np.uint8)
        rgba[..., :3] = img
        return torch.from_numpy(rgba).to(dev)
    host_pairs, inputs = [], []
    for j in range(NP):
        img1, img2, gu_j, gv_j = plan.timed_pair(j)
        if j == 0:
            gu, gv = gu_j, gv_j
        host_pairs.append((img1, img2))
        inputs.append((to_dev(img1), to_dev(img2), torch.empty((h, w, 2), dtype=torch.float32, device=dev)))
    d_flow = inputs[0][2]
    pitch = w * 4

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group=grp)
        torch.cuda.synchronize()

    def all_max(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
            return float(t.item())
        return x

    def step(i, npairs=S):
        e = engs[i % S]
        a, b, f = inputs[i % npairs]
        e.set_data_device(a.data_ptr(), b.data_ptr(), pitch)
        e.compute_flow_device(f.data_ptr())

    def run_steps(first, count, npairs):
        """Issue steps first .. first+count-1 (step i works on input pair i % npairs): one context per step, or -- with
        --batch B -- groups of B consecutive steps per batch context (the last group may be partial)."""
        if NB == 1:
            for i in range(first, first + count):
                step(i, npairs)
            return
        g = 0
        for i0 in range(first, first + count, NB):
            idx = [i % npairs for i in range(i0, min(i0 + NB, first + count))]
            e = bengs[g % S]
            g += 1
            e.set_data_device([inputs[j][0].data_ptr() for j in idx], [inputs[j][1].data_ptr() for j in idx], pitch)
            e.compute_flow_device([inputs[j][2].data_ptr() for j in idx])

    def sync_all():
        for e in engs + bengs:
            e.synchronize()

    run_steps(0, max(args.warmup, S * NB), S * NB)
    sync_all()

    # ---- the timed region: exactly --steps steps, bracketed by barrier + synchronize; run --repeats times, median reported ----
    tengs = bengs if NB > 1 else engs
    for e in tengs:                            # HIP events around the dominant kernel only, from a per-context pool
        e.enable_stage_timing(2)
        e.stage_times(clear=True)
    dts = []
    for _ in range(max(1, args.repeats)):
        # outside the timed window: every output buffer is poisoned, so that the hashes below can only match flows THIS repeat wrote
        # (the warm-up and the earlier repeats wrote the same flows into the same buffers)
        for x in inputs:
            x[2].fill_(float("nan"))
        barrier()
        t0 = time.perf_counter()
        run_steps(0, args.steps, S * NB)
        sync_all()
        barrier()
        dts.append(all_max(time.perf_counter() - t0))
    dt = float(np.median(dts))
    # every flow the timed region wrote (step i -> buffer i mod S*NB) against the committed oracle hash of its pair
    tv = plan.verify_timed([(j, inputs[j][2]) for j in sorted({i % (S * NB) for i in range(args.steps)})], host_pairs)
    if world > 1:
        t = torch.tensor(tv, dtype=torch.int64)
        dist.all_reduce(t, group=ctl)
        tv = [int(x) for x in t]
    bindings = [numa]
    if world > 1:
        bindings = [None] * world
        dist.all_gather_object(bindings, numa, group=ctl)
    dom_timed = []
    for e in tengs:
        dom_timed += e.stage_times(clear=True)
        e.enable_stage_timing(0)
    # the dominant kernel's launch duration WITHOUT other contexts' kernels interleaved between the event pair: the same
    # launches (NB pairs per launch) issued to ONE context with nothing else in flight
    te = tengs[0]
    te.enable_stage_timing(2)
    te.stage_times(clear=True)
    for r in range(6):
        if NB > 1:
            idx = list(range(NB))
            te.set_data_device([inputs[j][0].data_ptr() for j in idx], [inputs[j][1].data_ptr() for j in idx], pitch)
            te.compute_flow_device([inputs[j][2].data_ptr() for j in idx])
        else:
            a, b, f = inputs[0]
            te.set_data_device(a.data_ptr(), b.data_ptr(), pitch)
            te.compute_flow_device(f.data_ptr())
        te.synchronize()
    dom = te.stage_times(clear=True)
    te.enable_stage_timing(0)

    # sanity: the flow is finite and close to the synthetic ground truth (not a parity check)
    flow = d_flow.cpu().numpy()
    epe_gt = float(np.sqrt((flow[..., 0] - gu) ** 2 + (flow[..., 1] - gv) ** 2).mean())

    extras = {}
    if not args.no_extras:
        # ---- config 3 (BASELINE.json configs[2]): --pairs-per-gpu DISTINCT pairs per GPU, same issue scheme ----
        # (a) one context per pair, S contexts in flight on S streams; (b) ONE batch context: every launch covers the NP pairs
        def timed(fn, reps=3):
            fn()
            sync_all()
            ts = []
            for _ in range(reps):
                barrier()
                t0 = time.perf_counter()
                fn()
                sync_all()
                barrier()
                ts.append(all_max(time.perf_counter() - t0))
            return float(np.median(ts))
        dt3 = timed(lambda: [step(i, NC3) for i in range(NC3)])
        cb = eppm_amd.EPPMBatch(h, w, NC3, device=local_rank, params=params)

        def batch_pass():
            cb.set_data_device([x[0].data_ptr() for x in inputs[:NC3]], [x[1].data_ptr() for x in inputs[:NC3]], pitch)
            cb.compute_flow_device([x[2].data_ptr() for x in inputs[:NC3]])
            cb.synchronize()
        dt3b = timed(batch_pass)
        cb.enable_stage_timing(1)
        cb.stage_times(clear=True)
        batch_pass()
        bst = {}
        for name, ms in cb.stage_times(clear=True):
            bst[name] = bst.get(name, 0.0) + ms / NC3
        cb.close()
        extras["config3"] = {"workload": f"{NC3} distinct {w}x{h} pairs per GPU x {world} GPU(s), one pass (median of 3)",
                             "pairs": NC3 * world, "unit": "Mflow-vectors/s",
                             "streams": {"value": world * NC3 * w * h / dt3 / 1e6, "ms_per_pair": dt3 / NC3 * 1e3, "contexts_in_flight": S},
                             "batch": {"value": world * NC3 * w * h / dt3b / 1e6, "ms_per_pair": dt3b / NC3 * 1e3, "pairs_per_launch": NC3,
                                       "stage_ms_per_pair": bst}}

    if verify3:
        extras["config3_verified"] = verify_config3(args, rank, world, local_rank, params, dist, ctl, plan)

    if rank == 0:
        def per_launch(records):
            agg = {}
            for name, ms in records:
                agg.setdefault(name, []).append(ms)
            # per launch = mean over the kernel's two launches per group (levels 1 and 0): what rocprofv3 --stats averages for it
            return (float(np.mean(agg["c2f_refine_L0"])) + float(np.mean(agg["c2f_refine_L1"]))) / 2
        lv = eng.level_dims()
        dom_ms = per_launch(dom)                                   # one context, nothing else in flight
        dom_ms_timed = per_launch(dom_timed)                       # inside the timed region: event pairs also span other contexts' kernels
        alg_bytes1 = REFINE_BYTES_PER_PX * (lv[0][0] * lv[0][1] + lv[1][0] * lv[1][1]) / 2      # one pair, per launch
        alg_bytes = alg_bytes1 * NB
        hbm = {"bound": "hbm", "achieved": alg_bytes / (dom_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "algorithmic_bytes_per_launch": alg_bytes}
        hbm["frac"] = hbm["achieved"] / HBM_PEAK_GBS
        pmc = pmc_constants(w, h, args.patch_r)
        dp = dominant_pmc(pmc, NB)
        kern = ("k_c2f_refine_win / _win4 (candidate refine, bao_pmflow_kernel.cu:2005-2041); per launch = mean of its level-1 and level-0 launches of "
                f"{NB} pair(s), HIP events on the context's stream, ONE context with nothing else in flight (same launches as the timed region)")
        if dp:
            insts = dp["valu_insts"] / 2 * NB                      # wave64 VALU instructions per launch
            roof = {"bound": "valu", "kernel": kern, "achieved": insts / (dom_ms * 1e-3) / 1e9, "peak": VALU_PEAK_WAVE_INSTS_PER_S / 1e9,
                    "unit": "G wave64-inst/s", "frac": insts / (dom_ms * 1e-3) / VALU_PEAK_WAVE_INSTS_PER_S,
                    "traffic": (2 * (dp.get("fetch_size_kb") or 0) + (dp.get("write_size_kb") or 0)) * 1024 / 2 * NB, "traffic_unit": "HBM-side bytes per launch",
                    "wave64_valu_insts_per_launch": insts, "hbm": dict(hbm, traffic=(2 * (dp.get("fetch_size_kb") or 0) + (dp.get("write_size_kb") or 0)) * 1024 / 2 * NB),
                    "source": pmc.get("source"),
                    "note": "the path has no dense contraction and is not HBM bound (3 600 patch samples x ~46 VALU instructions per pixel against 26 "
                            "algorithmic bytes): the bound that applies is vector-ALU issue, peak = 256 CUs x 4 SIMD x 2.4 GHz / 2 cycles per wave64 "
                            "instruction; the HBM form the contract names is in `hbm`; `traffic` = HBM bytes per launch from separate --pmc passes "
                            "(2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction)"}
            if dp.get("fetch_size_kb") is None or dp.get("write_size_kb") is None:
                roof["traffic"] = roof["hbm"]["traffic"] = None
            if dp.get("lds"):
                # the LDS array beside the VALU (separate --pmc pass): cycles in which it serves this kernel, per CU and launch, against the launch's duration
                cyc = dp["lds"]["lds_idx_active"] / 2 * NB / 256
                roof["lds"] = {"lds_array_cycles_per_cu_per_launch": cyc, "bank_conflict_frac_of_them": dp["lds"]["lds_bank_conflict"] / max(1.0, dp["lds"]["lds_idx_active"]),
                               "busy_frac_at_2.4GHz": cyc / (dom_ms * 1e-3 * 2.4e9),
                               "note": "SQ_LDS_IDX_ACTIVE summed over the CUs / 256; 1.0 = the LDS array of every CU busy every cycle of the launch"}
        else:
            roof = dict(hbm, kernel=kern, traffic=None,
                        note="profiles/pmc_constants.json has no entry measured on these device sources: VALU instruction count and HBM traffic unknown")
        roof.update({"avg_launch_ms": dom_ms, "avg_launch_ms_timed_region": dom_ms_timed, "pairs_per_launch": NB,
                     "dominant_kernel_ms_per_step": 2 * dom_ms / NB})
        vals = sorted(world * args.steps * w * h / d / 1e6 for d in dts)
        out = {
            "metric": "Mflow-vectors/sec", "value": world * args.steps * w * h / dt / 1e6, "unit": "Mflow-vectors/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": len(dts), "value_min": vals[0], "value_max": vals[-1], "ms_per_step_by_repeat": [d / args.steps * 1e3 for d in dts],
            "config": {"workload": f"single {w}x{h} Sintel-shape synthetic pair per step, full 3-level pyramid, patch_r={args.patch_r}, "
                                   + ("" if args.propagation == 0 else f"PROPAGATION MODE {args.propagation} (not the reference's live scheme: other flows), ") +
                                   
                                   f"default defs.h parameters; {world} rank(s), independent pairs; inputs: RGBA planes resident in HBM before the "
                                   "timed region; outputs: interleaved float2 flow left in HBM (the PCIe-inclusive window of the reference API -- host "
                                   "RGB in, host u/v out -- is `host_boundary`)",
                       "inputs": "device-resident RGBA", "outputs": "device-resident float2 flow",
                       "pairs_per_step_per_gpu": 1, "pairs_in_flight_per_gpu": S * NB, "pairs_per_launch": NB, "contexts_in_flight": S,
                       "width": w, "height": h,
                       "host_binding": {"what": "each rank's host threads bound to the CPUs of its GPU's NUMA node (eppm_bind_thread_to_device; "
                                                "numa_node -1 = topology not visible, unbound)", "ranks": bindings}},
            "timed_region_verified": {"ok": tv[0], "of": tv[1], "inputs_differ_on_this_host": tv[2], "all_ok": tv[0] == tv[1] and tv[1] > 0,
                                      "what": "sha256 of every flow the last repeat of the timed region left in HBM (all ranks) == the CPU oracle's flow of that "
                                              "pair, tests/golden/MANIFEST_config3.json" if plan.man3 else plan.why_unverifiable},
            "roofline": roof,
            "epe_vs_synthetic_gt": epe_gt,
            "epe_vs_synthetic_gt_note": "flow QUALITY of the algorithm on the synthetic pair, not parity (the oracle's own flow has exactly this error): band-limited "
                                        "noise with two motion layers is hard for EPPM -- 43 % of the quarter-resolution matches fail the left-right check and 63 % are "
                                        "holes after the outlier vote, against 26 % / 33 % on the natural pair of other_configs.natural_1024x436 and 24 % / 27 % on the bundled pair (DESIGN.md section 5)",
        }
        out.update(extras)
        out["path_valu_roofline"] = path_valu_roofline(pmc, NB, dt / args.steps)
        if world == 1 and not args.no_extras:
            out.update(single_stream_legs(args, eng, inputs, pitch, pmc, alg_bytes1))
            if args.library == "exact":
                out["host_boundary"] = host_boundary(args, engs, host_pairs)
                if isinstance(out["host_boundary"], dict) and "pipelined" in out["host_boundary"]:
                    out["host_boundary"]["pipelined_over_value"] = out["host_boundary"]["pipelined"] / out["value"]
            out["cold_ms"] = cold_window(args, local_rank, params, host_pairs[0])
            if other_cfgs:
                for e in engs + bengs:          # the 4K context needs no room, but the timings should not share the chip with idle-but-resident contexts' streams
                    e.synchronize()
                if not args.no_cpu_baseline:
                    unbind_all_threads(host_cpus)          # the oracle legs get the job's CPUs, not the GPU's NUMA node
                out["other_configs"] = other_configs(plan, local_rank, dev, with_cpu=not args.no_cpu_baseline)
            if args.library == "exact" and not args.no_tolerance_mode:
                for e in engs + bengs:
                    e.synchronize()
                out["tolerance_mode"] = tolerance_mode(args)
        out["library"] = "exact (libeppm_hip.so: every flow bit-identical to the CPU oracle)" if args.library == "exact" else "tol (libeppm_hip_tol.so: NOT bit-identical)"
        if world == 1 and not args.no_cpu_baseline:
            unbind_all_threads(host_cpus)          # the CPU baseline gets the whole host, not the GPU's NUMA node
            out["cpu_baseline"] = cpu_baseline(w, h)
        data = (json.dumps(out) + "\n").encode()
        while data:
            data = data[os.write(json_fd, data):]
    if world > 1:
        dist.destroy_process_group()


def unbind_all_threads(cpus):
    """Give EVERY thread of this process the CPU set `cpus`: the threads created while the rank was bound to its GPU's NUMA node (torch's
    intra-op pool, an OpenMP pool, gloo, the HIP runtime's) inherited the narrow mask, and sched_setaffinity(0, ..) alone widens only the
    caller -- an oracle run on a pool thread would otherwise be timed on the node's CPUs."""
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for t in tids:
        try:
            os.sched_setaffinity(t, cpus)
        except OSError:
            pass                    # a thread that has exited meanwhile
    os.sched_setaffinity(0, cpus)


def bind_to_gpu_numa(device):
    """One process per GPU: bind this rank's host threads to the CPUs of the NUMA node its GPU's PCIe slot belongs to
    (eppm_bind_thread_to_device: hipDeviceGetPCIBusId + sysfs).  Recorded in config.host_binding; a host that does not expose the
    topology is left unbound."""
    import ctypes as C
    import eppm_amd
    info = {"pci": None, "numa_node": -1, "cpus_bound": 0}
    try:
        L = eppm_amd.lib()
        buf = C.create_string_buffer(32)
        if L.eppm_device_pci_bus_id(device, buf, C.c_size_t(32)) == 0:
            info["pci"] = buf.value.decode()
        node, n = C.c_int(-1), C.c_int(0)
        if L.eppm_bind_thread_to_device(device, C.byref(node), C.byref(n)) == 0:
            info["numa_node"], info["cpus_bound"] = node.value, n.value
    except Exception as ex:
        info["error"] = str(ex)[:120]
    return info


def verify_config3(args, rank, world, local_rank, params, dist, ctl, plan):
    """BASELINE configs[2] with a correctness bit: this rank's share of the 64 pairs (pair i -> rank i mod world: eppm_amd/shard.py;
    seeds 1234 + i) goes through the host boundary of a batch context (eppm_batch_set_images / eppm_batch_compute) and the
    sha256 of every flow is compared with tests/golden/MANIFEST_config3.json (the CPU oracle's flows, computed once in the build
    container).  Rank 0 reports how many of the 64 verified; a mismatch is named on stderr.  A pair whose synthetic images do not
    hash to the manifest's (another numpy / libm than the build container's) is UNVERIFIED: `all_ok` needs all 64 verified."""
    import hashlib as H
    import torch
    import eppm_amd
    man = plan.man3
    if man is None:
        return {"error": plan.why_unverifiable}
    w, h = man["w"], man["h"]
    mine = plan.share3
    B = eppm_amd.EPPMBatch(h, w, 8, device=local_rank, params=params)
    ok = bad_inputs = 0
    t0 = time.perf_counter()
    for g in range(0, len(mine), 8):
        idx = mine[g:g + 8]
        pairs = [plan.pair3(i)[:2] for i in idx]
        B.set_data(pairs)
        for i, (a, b), (u, v) in zip(idx, pairs, B.compute_flow()):
            rec = man["pairs"][str(i)]
            if H.sha256(a.tobytes()).hexdigest() != rec["img1_sha256"] or H.sha256(b.tobytes()).hexdigest() != rec["img2_sha256"]:
                bad_inputs += 1          # numpy / libm of this host generates other images than the build container's: cannot be verified
            elif H.sha256(u.tobytes() + v.tobytes()).hexdigest() == rec["flow_sha256"]:
                ok += 1
            else:
                print(f"[bench] rank {rank}: config-3 pair {i} (seed {man['seed0'] + i}): flow differs from the golden", file=sys.stderr)
    dt = time.perf_counter() - t0
    B.close()
    t = torch.tensor([ok, bad_inputs, len(mine)], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(t, group=ctl)
    n_ok, n_bad, n = int(t[0]), int(t[1]), int(t[2])
    return {"verified_pairs": n_ok, "pairs": n, "unverifiable_inputs_differ_on_this_host": n_bad, "mismatches": n - n_ok - n_bad,
            "all_ok": n_ok == n and n == man["n_pairs"], "state": "verified" if n_ok == n else ("mismatch" if n - n_ok - n_bad else "unverifiable"),
            "seconds_rank0": dt}


class InputPlan:
    """Which synthetic pairs this rank needs, generated up front by worker processes (eppm_amd.synth.make_pairs_parallel):
      * the timed region's pairs: indices (rank * NP + j) mod 64 of BASELINE configs[2]'s 64 pairs (seed 1234 + index), so that every
        flow of the timed region has a committed oracle hash (tests/golden/MANIFEST_config3.json) at any number of ranks;
      * with config-3 verification: this rank's share i = rank mod world of the 64 pairs;
      * rank 0 at N = 1: the 1920x1080 and 3840x2160 pairs of BASELINE configs[3], [4] (tests/golden/MANIFEST_large.json)."""

    def __init__(self, args, rank, world, NP, verify3, other):
        from eppm_amd import shard
        self.h, self.w, self.rank, self.NP = args.height, args.width, rank, NP
        self.man3, self.large, self.why_unverifiable = None, None, None
        try:
            self.man3 = json.load(open(os.path.join(ROOT, "tests", "golden", "MANIFEST_config3.json")))
            self.large = json.load(open(os.path.join(ROOT, "tests", "golden", "MANIFEST_large.json")))
        except Exception as e:
            self.why_unverifiable = f"no golden manifest: {e}"
        if self.man3 and (args.width, args.height, args.patch_r, args.propagation) != (self.man3["w"], self.man3["h"], 9, 0):
            self.man3, self.why_unverifiable = None, "the committed oracle hashes are for 1024x436 at patch_r 9, default parameters"
        n64 = self.man3["n_pairs"] if self.man3 else 64
        self.seed0 = self.man3["seed0"] if self.man3 else 1234
        self.timed_idx = [(rank * NP + j) % n64 for j in range(NP)]
        self.share3 = shard.pairs_for_rank(n64, rank, world) if (verify3 and self.man3) else []
        self.other = ["hd_1234", "uhd_r17_1234"] if (other and self.large) else []
        self.natural = [k for k in ("bundled_640x480", "natural_1024x436") if other and self.large and k in self.large]
        self._pairs = {}

    def _jobs(self):
        jobs = [(self.h, self.w, self.seed0 + i, 20.0) for i in dict.fromkeys(self.timed_idx + self.share3)]
        jobs += [(self.large[k]["h"], self.large[k]["w"], self.large[k]["seed"], self.large[k]["max_flow"]) for k in self.other]
        return jobs

    def generate(self):
        from eppm_amd import synth
        jobs = self._jobs()
        world = int(os.environ.get("WORLD_SIZE", "1"))
        for j, p in zip(jobs, synth.make_pairs_parallel(jobs, workers=max(1, min(32, (os.cpu_count() or 1) // (2 * world))))):
            self._pairs[j] = p

    def timed_pair(self, j):
        return self._pairs[(self.h, self.w, self.seed0 + self.timed_idx[j], 20.0)]

    def pair3(self, i):
        return self._pairs[(self.h, self.w, self.seed0 + i, 20.0)]

    def large_pair(self, name):
        r = self.large[name]
        if "seed" not in r:                       # a natural pair: the reference's bundled frames, or the Sintel-shape pair made from them
            from eppm_amd import synth
            return synth.bundled_pair() if name.startswith("bundled") else synth.natural_pair(r["h"], r["w"])
        return self._pairs[(r["h"], r["w"], r["seed"], r["max_flow"])]

    def verify_timed(self, flows, host_pairs):
        """flows: [(j, device tensor (h, w, 2) float32)] -> [ok, of, inputs_differ]"""
        if self.man3 is None:
            return [0, 0, 0]
        ok = bad = 0
        for j, f in flows:
            rec = self.man3["pairs"][str(self.timed_idx[j])]
            a, b = host_pairs[j]
            if hashlib.sha256(a.tobytes()).hexdigest() != rec["img1_sha256"] or hashlib.sha256(b.tobytes()).hexdigest() != rec["img2_sha256"]:
                bad += 1
                continue
            uv = f.cpu().numpy()
            if hashlib.sha256(uv[..., 0].tobytes() + uv[..., 1].tobytes()).hexdigest() == rec["flow_sha256"]:
                ok += 1
            else:
                print(f"[bench] rank {self.rank}: timed-region pair {self.timed_idx[j]} (buffer {j}): flow differs from the golden", file=sys.stderr)
        return [ok, len(flows), bad]


def other_configs(plan, device, dev, with_cpu=True):
    """The other BASELINE configurations in the driver's line, each with device-resident inputs, the window definition of `value` at a
    few steps, and the flow of every context checked against the committed CPU-oracle hash (tests/golden/MANIFEST_large.json):
      * `bundled`: configs[0], the reference's frame10/frame11 (640x480, the default three levels): throughput (3 contexts), the latency
        of one pair, `cold_ms` = init + set_data + compute_flow on a fresh object -- the window main.cpp:63-66 times --, per-stage device
        times, a `levels=1` run for information (BASELINE says "single scale"; the reference has no such mode), and the
        CPU oracle's time on the same pair;
      * `natural_1024x436`: a NATURAL pair of the headline shape (the bundled frames x1.6, cropped; eppm_amd/synth.py: natural_pair): the
        same fields -- what the synthetic noise pair of `value` says about real images;
      * `hd`: configs[3] 1920x1080 (3 contexts);  `uhd_r17`: configs[4] 3840x2160 at patch radius 17 (2 contexts), each with the fraction
        of the VALU-issue floor of that shape (profiles/pmc_constants.json).
    With the tolerance library a flow that differs from the hash is not an error: `verified.state` then says "differs" and the end-point
    error against the oracle is in tolerance_mode.epe_vs_oracle_px."""
    import numpy as np
    import torch
    import eppm_amd
    out = {}
    cases = [(n, k, c, st) for n, k, c, st in (("bundled_640x480", "bundled", 3, 48), ("natural_1024x436", "natural_1024x436", 3, 36)) if n in plan.natural]
    cases += [(n, k, c, st) for n, k, c, st in (("hd_1234", "hd", 3, 12), ("uhd_r17_1234", "uhd_r17", 2, 4)) if n in plan.other]
    for name, key, nctx, steps in cases:
        try:
            rec = plan.large[name]
            h, w, R = rec["h"], rec["w"], rec["patch_r"]
            natural = "seed" not in rec
            a, b = plan.large_pair(name)[:2]
            inputs_ok = hashlib.sha256(a.tobytes()).hexdigest() == rec["img1_sha256"] and hashlib.sha256(b.tobytes()).hexdigest() == rec["img2_sha256"]
            prm = eppm_amd.Params(patch_r=R)
            engs = []
            for _ in range(nctx):
                e = eppm_amd.EPPM(device=device, params=prm)
                e.init(h, w)
                engs.append(e)

            def to_dev(img):
                rgba = np.zeros((h, w, 4), np.uint8)
                rgba[..., :3] = img
                return torch.from_numpy(rgba).to(dev)
            da, db = to_dev(a), to_dev(b)
            flows = [torch.empty((h, w, 2), dtype=torch.float32, device=dev) for _ in range(nctx)]

            def run(n):
                for i in range(n):
                    e = engs[i % nctx]
                    e.set_data_device(da.data_ptr(), db.data_ptr(), w * 4)
                    e.compute_flow_device(flows[i % nctx].data_ptr())
                for e in engs:
                    e.synchronize()
            run(nctx)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(steps)
                ts.append((time.perf_counter() - t0) / steps)
            dt = float(np.median(ts))
            ok = 0
            for f in flows:
                uv = f.cpu().numpy()
                ok += int(hashlib.sha256(uv[..., 0].tobytes() + uv[..., 1].tobytes()).hexdigest() == rec["flow_sha256"])
            extra = {}
            if natural:
                # one pair on an idle GPU: latency and per-stage device times (the window main.cpp:63-66 times minus the allocation)
                e0 = engs[0]
                lat = []
                for _ in range(7):
                    e0.synchronize()
                    t1 = time.perf_counter()
                    e0.set_data_device(da.data_ptr(), db.data_ptr(), w * 4)
                    e0.compute_flow_device(flows[0].data_ptr())
                    e0.synchronize()
                    lat.append((time.perf_counter() - t1) * 1e3)
                e0.enable_stage_timing(1)
                e0.stage_times(clear=True)
                for _ in range(6):
                    e0.set_data_device(da.data_ptr(), db.data_ptr(), w * 4)
                    e0.compute_flow_device(flows[0].data_ptr())
                    e0.synchronize()
                agg = {}
                for sname, ms in e0.stage_times(clear=True):
                    agg.setdefault(sname, []).append(ms)
                e0.enable_stage_timing(0)
                extra = {"latency_ms_per_pair": float(np.median(lat)), "stage_ms": {k: float(np.mean(v)) for k, v in agg.items()}}
            for e in engs:
                e.close()
            if natural:
                cold = []
                for _ in range(3):                 # the reference's own window: a fresh object, host images in, host planes out
                    t1 = time.perf_counter()
                    ec = eppm_amd.EPPM(device=device, params=prm)
                    ec.init(a, b, h, w)
                    ec.compute_flow()
                    cold.append((time.perf_counter() - t1) * 1e3)
                    ec.close()
                extra["cold_ms"] = float(np.median(cold))
                if name.startswith("bundled"):     # BASELINE configs[0] says "single scale": a levels=1 run, for information (no oracle hash)
                    e1 = eppm_amd.EPPM(device=device, params=eppm_amd.Params(patch_r=R, levels=1))
                    e1.init(a, b, h, w)
                    e1.compute_flow()
                    t1 = time.perf_counter()
                    for _ in range(3):
                        e1.compute_flow()
                    extra["levels_1_ms_per_pair"] = (time.perf_counter() - t1) / 3 * 1e3
                    e1.close()
                if with_cpu:
                    extra["cpu_oracle"] = cpu_oracle_pair(a, b)
            del da, db, flows
            pv = path_valu_roofline(pmc_constants(w, h, R), 1, dt)
            state = "verified" if (inputs_ok and ok == nctx) else ("unverifiable: inputs differ on this host" if not inputs_ok else
                                                                   ("mismatch" if LIBRARY == "exact" else "differs (tolerance library: see tolerance_mode.epe_vs_oracle_px)"))
            what = (f"the reference's bundled frame10/frame11 ({w}x{h})" if name.startswith("bundled") else
                    f"natural {w}x{h} pair: the bundled frames scaled x{w / 640:g} and centre-cropped (eppm_amd/synth.py: natural_pair)") if natural else \
                   f"single {w}x{h} synthetic pair per step (seed {rec['seed']}, |flow| <= {rec['max_flow']:g})"
            out[key] = {"workload": f"{what}, full 3-level pyramid, patch_r={R}; "
                                    f"device-resident RGBA in, float2 flow left in HBM; {nctx} single-pair contexts in flight, {steps} steps, median of 3 windows",
                        "value": w * h / dt / 1e6, "unit": "Mflow-vectors/s", "ms_per_step": dt * 1e3, "contexts_in_flight": nctx,
                        "path_valu_roofline": {"frac": pv["frac"], "floor_ms_per_pair": pv["floor_ms_per_pair"]} if pv else None,
                        "verified": {"ok": ok if inputs_ok else 0, "of": nctx, "state": state,
                                     "against": f"tests/golden/MANIFEST_large.json[{name}].flow_sha256 (the CPU oracle's flow)"}}
            out[key].update(extra)
        except Exception as ex:                  # a reported extra, never a reason to lose the line
            out[key] = {"error": str(ex)[:300]}
    return out


def cpu_oracle_pair(a, b, runs=2):
    """The CPU oracle on one pair, on at most 16 of this job's CPUs (its lockstep sweeps are fastest there): ms per pair, best of `runs`."""
    from oracle import oracle as O
    n_all = O.num_threads()
    n = min(16, len(os.sched_getaffinity(0)))
    O.set_num_threads(n)
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        O.compute_flow(a, b)
        ts.append((time.perf_counter() - t0) * 1e3)
    O.set_num_threads(n_all)
    h, w = a.shape[:2]
    return {"ms_per_pair": min(ts), "value": w * h / min(ts) / 1e3, "unit": "Mflow-vectors/s", "cores": n, "kind": "port",
            "sample": f"the same pair, whole path, best of {runs} runs, OpenMP oracle on {n} threads"}


def tolerance_mode(args):
    """The tolerance library (libeppm_hip_tol.so) beside `value`, never instead of it: the same window, steps and issue scheme by a child
    run of this script with --library tol (one library per process), its single-pair latency, stage times, roofline of its dominant kernel
    and the other configurations; plus its end-point error against the exact library = the CPU oracle (tools/tolerance_epe.py: the bundled
    pair in both directions -- north_star's 1e-3 px case -- and BASELINE configs[1], [3], [4])."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "EPPM_HIP_VARIANT"):
        env.pop(k, None)
    out = {"library": "eppm_amd/lib/libeppm_hip_tol.so (make -C eppm_amd/csrc; -DEPPM_TOL: integer-domain tables / one hardware exp2 in the patch "
                      "term, fused sums in one canonical chunked order (DESIGN.md section 9.2), column-parity target planes; everything else -- prepare, census, random field, left-right "
                      "check, outlier vote, weighted median, hole filling, flow smoothing -- is the exact library's code)",
           "parity": "NOT bit-identical; north_star's bar: mean EPE <= 1e-3 px against the oracle on frame10/frame11 (asserted by -m gpu tests)"}
    try:
        cmd = [sys.executable, os.path.abspath(__file__), "--library", "tol", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--batch", str(args.batch), "--inflight", str(args.inflight), "--repeats", str(args.repeats), "--no-cpu-baseline", "--no-tolerance-mode"]
        if args.no_other_configs:
            cmd.append("--no-other-configs")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        for k in ("value", "unit", "ms_per_step", "value_min", "value_max", "steps", "warmup", "latency_ms_per_pair", "stage_ms", "roofline",
                  "path_valu_roofline", "valu_roofline", "other_configs", "epe_vs_synthetic_gt"):
            if k in d:
                out[k] = d[k]
        tv = d.get("timed_region_verified", {})
        out["timed_region_bit_identical"] = {"ok": tv.get("ok"), "of": tv.get("of"),
                                             "what": "flows of the timed region whose sha256 EQUALS the oracle's (informational: the library is not required to be bit-identical)"}
    except Exception as ex:
        out["error"] = str(ex)[:300]
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tolerance_epe.py"), "--all", "--no-fuzz"], env=env, capture_output=True, text=True, timeout=900)
        e = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        out["epe_vs_oracle_px"] = {"tolerance_px_on_the_bundled_pair": 1e-3, "against": e["against"],
                                   "cases": {k: {"mean": c["epe_mean_px"], "max": c["epe_max_px"], "frac_over_1px": c["frac_over_1px"], "pixels_differing": c["pixels_differing"]}
                                             for k, c in e["cases"].items()}}
    except Exception as ex:
        out["epe_vs_oracle_px"] = {"error": str(ex)[:300]}
    return out


def single_stream_legs(args, eng, inputs, pitch, pmc, alg_bytes):
    """One context, one stream, nothing else in flight: per-pair latency, per-stage device times and the dominant
    kernel against the bound that limits it (vector-ALU issue), free of the contention of the throughput window."""
    import numpy as np
    a, b, f = inputs[0]
    lat = []
    for _ in range(7):
        eng.synchronize()
        t1 = time.perf_counter()
        eng.set_data_device(a.data_ptr(), b.data_ptr(), pitch)
        eng.compute_flow_device(f.data_ptr())
        eng.synchronize()
        lat.append((time.perf_counter() - t1) * 1e3)
    eng.enable_stage_timing(1)
    eng.stage_times(clear=True)
    for _ in range(10):
        eng.set_data_device(a.data_ptr(), b.data_ptr(), pitch)
        eng.compute_flow_device(f.data_ptr())
        eng.synchronize()
    st = eng.stage_times(clear=True)
    eng.enable_stage_timing(0)
    agg = {}
    for name, ms in st:
        agg.setdefault(name, []).append(ms)
    stage_ms = {k: float(np.mean(v)) for k, v in agg.items()}
    out = {"latency_ms_per_pair": float(np.median(lat)), "stage_ms": stage_ms}
    dom1 = (stage_ms["c2f_refine_L0"] + stage_ms["c2f_refine_L1"]) / 2
    out["roofline_single_stream"] = {"achieved": alg_bytes / (dom1 * 1e-3) / 1e9, "unit": "GB/s", "avg_launch_ms": dom1}
    dp = dominant_pmc(pmc, 1)
    if dp:
        v = dp["valu_insts"] / 2
        out["valu_roofline"] = {"bound": "valu", "kernel": "the candidate refine at one pair per launch, single stream (mean of its level-1 and level-0 launches)",
                                "wave64_valu_insts_per_launch": v, "achieved_insts_per_s": v / (dom1 * 1e-3), "peak_insts_per_s": VALU_PEAK_WAVE_INSTS_PER_S,
                                "frac": v / (dom1 * 1e-3) / VALU_PEAK_WAVE_INSTS_PER_S, "avg_launch_ms": dom1, "source": pmc.get("source")}
    else:
        out["valu_roofline"] = None
    return out


def host_boundary(args, engs, host_pairs):
    """Through the host-pointer boundary (eppm_set_images + eppm_compute: RGB->RGBA, H2D, path, D2H, planar copy-out):
    synchronous on one context, and pipelined by one host thread over 2-4 contexts (the best is reported).  Measured by
    tools/host_boundary.py in a child process (a process without this harness's other contexts, streams and torch runtime).
    PCIe-inclusive: never `value`."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_boundary.py"), "--json", str(args.width), str(args.height)],
                             env=env, capture_output=True, text=True, timeout=600)
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as ex:
        return {"error": str(ex)[:200]}


def cold_window(args, device, params, pair):
    """init + compute_flow on a fresh object, the window main.cpp:63-66 times (allocation included); median of 3."""
    import numpy as np
    import eppm_amd
    w, h = args.width, args.height
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        e = eppm_amd.EPPM(device=device, params=params)
        e.init(pair[0], pair[1], h, w)
        e.compute_flow()
        ts.append((time.perf_counter() - t) * 1e3)
        e.close()
    return float(np.median(ts))


def cpu_baseline(w, h):
    """The CPU oracle (oracle/, a port of the reference's kernel semantics: the reference has no CPU path) timed on this host's cores,
    on a bounded sample of the workload:
      * `value` = `whole_host`: what the BOX computes -- floor(physical cores / 16) oracle processes side by side, each on 16 OpenMP
        threads bound to 16 physical cores of its own, each computing a DISTINCT pair of the workload (seeds 1234 + i), all started
        together; three such rounds, the median round reported; `cores` = the threads actually busy.  (One process cannot use the box:
        the oracle's lockstep sweeps synchronise every step, 16-32 threads are fastest, 128 take twice and 256 nine times as long,
        tools/orc_threads.py.)
      * `single_pair_16_threads`: the latency of ONE pair on 16 threads, median of 3 runs (the figure of rounds 1-4);
      * `single_thread`: one thread on the pair's centre 256x128 crop."""
    from oracle import oracle as O
    from eppm_amd import synth
    a, b, _, _ = synth.make_pair_cached(h, w, seed=1234)
    O.compute_flow(a[:32, :32].copy(), b[:32, :32].copy())      # build + warm
    n_all = O.num_threads()
    ncpu = len(os.sched_getaffinity(0))
    n = min(16, ncpu)
    O.set_num_threads(n)
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.compute_flow(a, b)
        runs.append(time.perf_counter() - t0)
    dt = float(sorted(runs)[len(runs) // 2])
    qh, qw = min(h, 128), min(w, 256)
    y0, x0 = (h - qh) // 2, (w - qw) // 2
    qa, qb = a[y0:y0 + qh, x0:x0 + qw].copy(), b[y0:y0 + qh, x0:x0 + qw].copy()
    O.set_num_threads(1)
    t0 = time.perf_counter()
    O.compute_flow(qa, qb)
    dt1 = time.perf_counter() - t0
    O.set_num_threads(n_all)
    single = {"value": w * h / dt / 1e6, "unit": "Mflow-vectors/s", "cores": n, "ms_per_pair": dt * 1e3,
              "sample": f"1 pair {w}x{h} (the workload's own pair, seed 1234), whole path, median of {len(runs)} runs = {dt:.2f} s (min {min(runs):.2f}, max "
                        f"{max(runs):.2f}), OpenMP oracle on {n} of {ncpu} hardware threads"}
    out = dict(single, kind="port")
    out["single_pair_16_threads"] = single
    out["single_thread"] = {"value": qw * qh / dt1 / 1e6, "unit": "Mflow-vectors/s", "cores": 1,
                            "sample": f"centre {qw}x{qh} crop of the same pair, whole path once, {dt1:.1f} s, one thread"}
    cores, quota = physical_cores(os.sched_getaffinity(0)), cpu_quota()
    usable = len(cores) if quota is None else max(1, min(len(cores), int(quota)))
    if usable // n <= 1:
        # everything this job may use IS one n-thread process: the runs above are the whole-host figure (no second set of runs)
        wh = {"value": single["value"], "unit": "Mflow-vectors/s", "cores": n, "kind": "port", "processes": 1, "threads_per_process": n,
              "round_s": runs, "hardware_threads": ncpu, "physical_cores": len(cores), "smt_siblings_used": False, "cgroup_cpu_quota": quota,
              "usable_cpus": usable,
              "sample": single["sample"] + (f"; the job's cgroup allows {quota:g} CPUs of time (cpu.max) on this {ncpu}-thread host, so one {n}-thread "
                                            "process is everything it may use (more busy threads are throttled: profiles/r05x_b_cpu_layouts.txt)" if quota is not None else "")}
    else:
        wh = cpu_whole_host(w, h, n)
    out["whole_host"] = wh
    if "value" in wh:          # the throughput figure of the box is the baseline of a throughput metric
        out.update({"value": wh["value"], "cores": wh["cores"], "sample": wh["sample"]})
    return out


def physical_cores(cpus):
    """One hardware thread per physical core among `cpus` (sysfs thread_siblings_list; every CPU its own core when sysfs does not say)."""
    seen, out = set(), []
    for c in sorted(cpus):
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return out


def cpu_quota():
    """CPUs of time the job's cgroup may use (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), or None when unlimited.  The GPU boxes of this
    pool show 256 hardware threads and give a job 16 CPUs ("1600000 100000"): more busy threads than that are throttled, not run."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / p if q > 0 else None
    except Exception:
        return None


def cpu_whole_host(w, h, threads_per_proc, rounds=3, smt=False, env_extra=None, budget_s=30.0, ignore_quota=False):
    """One worker process (this file, --cpu-worker) per `threads_per_proc` CPUs the job may really use -- physical cores (one hardware
    thread each), capped by the cgroup's CPU quota: on the GPU boxes of this pool cpu.max gives a job 16 CPUs of a 256-thread host, and
    8 x 16 or 16 x 16 busy threads are throttled to that (24-84 s per pair against 2.5 s for one process alone:
    profiles/r05x_b_cpu_layouts.txt; ignore_quota=True reproduces it) -- each bound to its own cores, each holding its own pair; a round = every worker computes its pair once, all started by one "go"; the round's wall time runs from
    the go to the last answer.  Bounded: rounds x one pair per worker, and no further round once budget_s is spent."""
    allowed = sorted(os.sched_getaffinity(0))
    cpus = allowed if smt else physical_cores(allowed)
    quota = cpu_quota()
    usable = len(cpus) if (quota is None or ignore_quota) else max(1, min(len(cpus), int(quota)))
    nproc = max(1, usable // threads_per_proc)
    threads_per_proc = min(threads_per_proc, usable)
    procs = []
    try:
        from oracle import oracle as O
        O.lib()                              # built here, once: the workers only load it
        for i in range(nproc):
            mine = cpus[i * threads_per_proc:(i + 1) * threads_per_proc]
            env = dict(os.environ, OMP_NUM_THREADS=str(len(mine)), OMP_PROC_BIND="false")
            env.update(env_extra or {})
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(h), str(w), str(1234 + i),
                                           ",".join(map(str, mine))], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("a CPU-baseline worker did not start")
        walls, per_pair = [], []
        t_all = time.perf_counter()
        for _ in range(rounds):
            t0 = time.perf_counter()
            for p in procs:
                p.stdin.write("go\n")
                p.stdin.flush()
            per_pair += [float(p.stdout.readline()) for p in procs]
            walls.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all + walls[-1] > budget_s:
                break
        for p in procs:
            p.stdin.write("quit\n")
            p.stdin.flush()
        wall = float(sorted(walls)[len(walls) // 2])
        busy = nproc * threads_per_proc
        return {"value": nproc * w * h / wall / 1e6, "unit": "Mflow-vectors/s", "cores": busy, "kind": "port",
                "processes": nproc, "threads_per_process": threads_per_proc, "round_s": walls, "pair_s_min_max": [min(per_pair), max(per_pair)],
                "hardware_threads": len(allowed), "physical_cores": len(physical_cores(allowed)), "smt_siblings_used": bool(smt),
                "cgroup_cpu_quota": quota, "usable_cpus": usable,
                "sample": f"{nproc} oracle process(es) x {threads_per_proc} OpenMP threads = {busy} threads on {busy} "
                          f"{'hardware threads' if smt else 'physical cores (one hardware thread per core)'} of a host with {len(allowed)} hardware threads"
                          f"{'' if quota is None else f' whose cgroup gives this job {quota:g} CPUs of time (cpu.max)'}, each "
                          f"process bound to its own CPUs and computing a distinct {w}x{h} pair (seeds 1234..{1233 + nproc}), whole path; {len(walls)} round(s) "
                          f"of one pair per process, started together; median round {wall:.2f} s"}
    except Exception as ex:                  # a reported extra, never a reason to lose the line
        return {"error": str(ex)[:300]}
    finally:
        for p in procs:
            try:
                p.stdin.close()
                p.wait(timeout=20)
            except Exception:
                p.kill()


def cpu_worker(argv):
    """--cpu-worker H W SEED CPULIST: one process of cpu_whole_host.  Binds itself, loads its pair and the oracle, says "ready", then
    computes the pair once per "go" line and answers with the seconds it took."""
    h, w, seed = int(argv[0]), int(argv[1]), int(argv[2])
    cpus = [int(c) for c in argv[3].split(",") if c]
    if cpus:
        os.sched_setaffinity(0, cpus)
    from oracle import oracle as O
    from eppm_amd import synth
    a, b, _, _ = synth.make_pair_cached(h, w, seed=seed)
    O.set_num_threads(max(1, len(cpus)))
    O.compute_flow(a[:32, :32].copy(), b[:32, :32].copy())
    print("ready", flush=True)
    for line in sys.stdin:
        if line.strip() != "go":
            break
        t0 = time.perf_counter()
        O.compute_flow(a, b)
        print(time.perf_counter() - t0, flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        cpu_worker(sys.argv[2:])
        return
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args)
        return
    worker(args)


if __name__ == "__main__":
    main()
