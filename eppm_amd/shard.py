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
