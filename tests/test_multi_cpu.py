"""The N > 1 path on CPU: world_size-2 gloo processes shard independent pairs with no data-path collective
(SURVEY 8e); only the timing protocol of bench.py (barrier + MAX over ranks) uses torch.distributed."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT
from eppm_amd.shard import pairs_for_rank


def test_round_robin_partition():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 4, 8):
            parts = [pairs_for_rank(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert pairs_for_rank(64, 3, 8) == list(range(3, 64, 8))


def test_pipelined_runner_bookkeeping():
    """run_pairs_pipelined on fake engines: every pair comes back once, with the result of ITS inputs, and an engine is
    never given a new pair before its previous one was collected."""
    from eppm_amd.shard import run_pairs_pipelined

    class Fake:
        def __init__(self):
            self.cur, self.pending, self.log = None, False, []
        def set_data(self, a, b):
            assert not self.pending, "set_data while a flow is pending"
            self.cur = (a, b)
        def compute_flow_begin(self):
            self.pending = True
        def compute_flow_end(self):
            assert self.pending
            self.pending = False
            return (self.cur[0] * 10, self.cur[1] * 10)

    pairs = [(i, -i) for i in range(11)]
    for k in (1, 2, 3, 5, 16):
        out = run_pairs_pipelined([Fake() for _ in range(k)], pairs, [0, 2, 3, 4, 7, 8, 9, 10])
        assert sorted(out) == [0, 2, 3, 4, 7, 8, 9, 10]
        assert all(out[i] == (10 * i, -10 * i) for i in out)


def test_batched_runner_bookkeeping():
    """run_pairs_batched on a fake batch engine: groups of npairs, a smaller last group, every pair's result under its own index."""
    from eppm_amd.shard import run_pairs_batched

    class FakeBatch:
        npairs = 4

        def __init__(self):
            self.cur, self.sizes = None, []

        def set_data(self, pairs):
            assert 1 <= len(pairs) <= self.npairs
            self.cur = list(pairs)
            self.sizes.append(len(pairs))

        def compute_flow(self):
            return [(a * 10, b * 10) for a, b in self.cur]

    pairs = [(i, -i) for i in range(23)]
    for world in (1, 2, 8):
        seen = {}
        for r in range(world):
            fb = FakeBatch()
            mine = pairs_for_rank(len(pairs), r, world)
            out = run_pairs_batched(fb, pairs, mine)
            assert sorted(out) == mine and all(out[i] == (10 * i, -10 * i) for i in out)
            assert fb.sizes == [4] * (len(mine) // 4) + ([len(mine) % 4] if len(mine) % 4 else [])
            seen.update(out)
        assert sorted(seen) == list(range(len(pairs)))


WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from eppm_amd.shard import pairs_for_rank
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = pairs_for_rank(13, rank, world)
    # the data path needs nothing from the other rank; the checks below are test-only
    got = [None] * world
    dist.all_gather_object(got, mine)
    assert sorted(i for p in got for i in p) == list(range(13))
    dist.barrier()
    t = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)      # bench.py: max over ranks of the local time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert abs(t.item() - 0.010 * world) < 1e-12
    vectors = sum(len(p) for p in got) * 1024 * 436
    if rank == 0:
        print("OK", vectors / t.item() / 1e6)
    dist.destroy_process_group()
""")


def test_two_gloo_ranks_shard_pairs(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e
    assert any(line.startswith("OK") for line in outs[0][0].splitlines())


STUB = textwrap.dedent("""
    import json, os, sys
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    dist.destroy_process_group()
    if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        sys.exit(3)
    if rank == 0:
        print("noise before the line")
        print(json.dumps({"n_gpus": world, "argv": sys.argv[1:]}))
""")


def test_bench_gpus_n_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher: the parent spawns N rank processes with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, relays rank 0's JSON line, fails when a rank fails, and never imports torch itself
    (it must not touch the GPU: ADVICE r1).  The rank program is replaced by a stub (no GPU here); the real ranks run
    in the GPU suite (test_bench_two_ranks_share_one_gpu)."""
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    drv = textwrap.dedent(f"""
        import json, sys
        sys.path.insert(0, {ROOT!r})
        import bench
        sys.argv = ["bench.py", "--gpus", "2", "--steps", "3"] + sys.argv[1:]
        args = bench.parse_args_known()
        bench.spawn_ranks(args, script={str(stub)!r})
        assert "torch" not in sys.modules, "the spawning parent imported torch"
    """)
    ok = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr
    line = [ln for ln in ok.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1 and '"n_gpus": 2' in line[0] and "--steps" in line[0], ok.stdout
    bad = subprocess.run([sys.executable, "-c", drv, "--fail-rank", "1"], capture_output=True, text=True, timeout=300)
    assert bad.returncode == 1 and "ranks failed" in bad.stderr, (bad.returncode, bad.stderr)


def test_bench_gpus_8_spawns_eight_ranks(tmp_path):
    """The same with --gpus 8 (BASELINE configs[2]'s rank count): eight rank processes rendezvous on the port the parent picked, one line
    comes back.  (The real ranks, eight of them on one GPU: tests/test_configs_gpu.py::test_bench_eight_ranks_share_one_gpu.)"""
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    drv = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        sys.argv = ["bench.py", "--gpus", "8", "--steps", "16"]
        bench.spawn_ranks(bench.parse_args_known(), script={str(stub)!r})
    """)
    ok = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=600)
    assert ok.returncode == 0, ok.stderr[-2000:]
    line = [ln for ln in ok.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1 and '"n_gpus": 8' in line[0], ok.stdout


def test_cpu_whole_host_baseline_uses_every_cpu_once():
    """bench.cpu_whole_host: one oracle process per `threads_per_proc` CPUs, each bound to its own CPUs, each with a distinct pair, rounds
    started together; `cores` is what was busy.  Small pairs here; the GPU box runs it at 1024x436 with 16 threads per process."""
    import bench
    ncpu = len(bench.physical_cores(os.sched_getaffinity(0)))
    tpp = max(1, ncpu // 2)
    r = bench.cpu_whole_host(96, 64, tpp, rounds=2)
    assert "error" not in r, r
    # what the job may use is its cgroup's CPU quota when there is one (the pool's GPU boxes: 16 CPUs of a 128-core host -> one process of
    # 16 threads), every physical core otherwise (this container)
    usable = min(ncpu, int(bench.cpu_quota() or ncpu))
    procs, threads = max(1, usable // tpp), min(tpp, usable)
    assert r["physical_cores"] == ncpu and r["usable_cpus"] == usable and r["processes"] == procs and r["threads_per_process"] == threads
    assert r["cores"] == procs * threads and r["kind"] == "port"
    assert r["value"] > 0 and len(r["round_s"]) == 2 and r["value"] == r["processes"] * 96 * 64 / sorted(r["round_s"])[1] / 1e6


def test_ranks_agree_on_the_backend_when_the_probe_fails_on_one_rank(tmp_path):
    """bench.agree_on_group: two gloo ranks, the "RCCL" probe succeeds on rank 0 and raises on rank 1 (a partial failure): BOTH must
    end up on the gloo group -- decided by a MIN over the gloo control group -- and a barrier + MAX over the agreed group completes.
    With a probe that succeeds everywhere both take the probed group; without a probe both take gloo."""
    stub = tmp_path / "rank.py"
    stub.write_text(textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        import bench
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        out = {{}}
        def good():
            return dist.new_group(backend="gloo")
        for name, probe in (("good", good), ("none", None)):
            ctl, grp = bench.agree_on_group(dist, rank, probe)
            out[name] = grp is ctl
            dist.barrier(group=grp)
            t = torch.tensor([float(rank)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
            assert t.item() == world - 1
        # the partial failure: rank 0's probe succeeds, rank 1's raises
        def partial_safe():
            if rank == 1:
                raise RuntimeError("simulated RCCL failure on rank 1")
            return object()                                 # stands for a group only rank 0 believes in
        ctl, grp = bench.agree_on_group(dist, rank, partial_safe)
        out["partial"] = grp is ctl
        dist.barrier(group=grp)
        print(json.dumps(out), flush=True)
        dist.destroy_process_group()
    """))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(stub)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    import json
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, se[-2000:]
        d = json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1])
        assert d == {"good": False, "none": True, "partial": True}, (r, d)
    assert "simulated RCCL failure" in outs[1][1] and "barrier and MAX go over gloo" in outs[0][1]


def test_bench_input_plan_keeps_every_rank_inside_the_64_goldens():
    """bench.InputPlan: the pairs a rank times are pairs (rank * P + j) mod 64 of BASELINE configs[2]'s 64 pairs, so that every flow of the
    timed region has a committed oracle hash at any number of ranks (with 24 pairs in flight per rank, rank 3 of 8 would otherwise start
    at pair 72); its config-3 share is pair i -> rank i mod N; another size or radius has no goldens and says so."""
    import argparse
    import bench
    mk = lambda **kw: argparse.Namespace(**dict(dict(width=1024, height=436, patch_r=9, propagation=0), **kw))   # noqa: E731
    seen = set()
    for rank in range(8):
        p = bench.InputPlan(mk(), rank, 8, 24, True, False)
        assert p.man3 is not None and len(p.timed_idx) == 24 and all(0 <= i < 64 for i in p.timed_idx)
        assert p.timed_idx[0] == (rank * 24) % 64
        assert p.share3 == list(range(rank, 64, 8))
        seen |= set(p.share3)
        jobs = p._jobs()
        assert len(jobs) == len(set(jobs)) and all(j[:2] == (436, 1024) and 1234 <= j[2] < 1298 for j in jobs)
    assert seen == set(range(64))
    p = bench.InputPlan(mk(), 0, 1, 24, False, True)
    assert p.other == ["hd_1234", "uhd_r17_1234"] and (1080, 1920, 1234, 40.0) in p._jobs() and (2160, 3840, 1234, 60.0) in p._jobs()
    for odd in (dict(width=640, height=480), dict(patch_r=17), dict(propagation=1)):
        q = bench.InputPlan(mk(**odd), 0, 1, 24, True, False)
        assert q.man3 is None and q.why_unverifiable and q.share3 == [] and q.verify_timed([], []) == [0, 0, 0]


def test_synth_cache_and_worker_processes_reproduce_the_generator(tmp_path, monkeypatch):
    """eppm_amd.synth.make_pairs_parallel (worker processes filling a file cache) returns exactly what make_pair generates, from a cold and
    from a warm cache; the goldens are tied to the generator's exact output."""
    import numpy as np
    from eppm_amd import synth
    monkeypatch.setenv("EPPM_SYNTH_CACHE", str(tmp_path / "cache"))
    jobs = [(48, 64, 5, 6.0), (40, 72, 6, 4.0), (48, 64, 5, 6.0)]
    want = [synth.make_pair(*j[:2], seed=j[2], max_flow=j[3]) for j in jobs]
    for attempt in range(2):
        got = synth.make_pairs_parallel(jobs, workers=2)
        for g, w_ in zip(got, want):
            assert all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(g, w_)), attempt
    assert len(list((tmp_path / "cache").iterdir())) == 2
