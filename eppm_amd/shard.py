"""Pair-level sharding across the GPUs of one node: independent image pairs,
pair i -> rank i mod world, no exchange step and therefore no collective on the data path."""


def pairs_for_rank(n_pairs, rank, world):
    """Indices of the pairs rank `rank` of `world` processes computes (round robin)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    return list(range(rank, n_pairs, world))


def run_pairs(engine, pairs, indices):
    """Run `engine` (an initialised eppm_amd.EPPM) over pairs[i] for i in indices; returns {i: (u, v)}."""
    out = {}
    for i in indices:
        a, b = pairs[i]
        engine.set_data(a, b)
        out[i] = engine.compute_flow()
    return out


def run_pairs_pipelined(engines, pairs, indices):
    """The same through several initialised engines of ONE GPU used round robin by one host thread: while engine k
    computes pair i, the host stages pair i+1 into engine k+1 and collects pair i-1 -- the PCIe copies and the host-side
    staging of one pair overlap the kernels of the others.  Returns {i: (u, v)}."""
    if not engines:
        raise ValueError("no engines")
    out, busy = {}, [None] * len(engines)
    for n, i in enumerate(indices):
        k = n % len(engines)
        if busy[k] is not None:
            out[busy[k]] = engines[k].compute_flow_end()
        a, b = pairs[i]
        engines[k].set_data(a, b)
        engines[k].compute_flow_begin()
        busy[k] = i
    for k, i in enumerate(busy):
        if i is not None:
            out[i] = engines[k].compute_flow_end()
    return out


def run_pairs_batched(batch_engine, pairs, indices):
    """The same through ONE batch engine (eppm_amd.EPPMBatch): the rank's pairs go through the context `npairs` at a time,
    every kernel launch covering the whole group (the last group may be smaller).  Returns {i: (u, v)}."""
    out = {}
    idx = list(indices)
    n = batch_engine.npairs
    for g in range(0, len(idx), n):
        group = idx[g:g + n]
        batch_engine.set_data([pairs[i] for i in group])
        for i, uv in zip(group, batch_engine.compute_flow()):
            out[i] = uv
    return out
