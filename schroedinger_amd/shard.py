"""Picture-level sharding of a Dirac stream across GPUs.

The pixel path has no cross-picture data flow except through reference pictures, so
the unit of distribution is a *reference chain*: a connected component of the
"picture A predicts from picture B" graph (a closed GOP, or a single intra / VC-2
low-delay picture).  Every chain is decoded on one device; chains are spread over the
ranks longest-first.  No collective is needed on the data path: the only exchange is
the host handing each rank its own pictures (schro_decoder_push per rank).

This mirrors what the reference's scheduler would need per device
(schro_decoder_async_schedule, schrodecoder.c:1546-1682: a render stage may only run
where the picture's references live).
"""
from collections import defaultdict


def reference_chains(pictures):
    """pictures: iterable of (picture_number, [reference picture numbers]).
    Returns a list of chains, each a sorted list of picture numbers; chains are ordered
    by their first picture."""
    parent = {}

    def find(x):
        parent.setdefault(x, x)
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for num, refs in pictures:
        find(num)
        for r in refs:
            ra, rb = find(num), find(r)
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
    groups = defaultdict(list)
    for num in parent:
        groups[find(num)].append(num)
    return [sorted(v) for _, v in sorted(groups.items())]


def assign_chains(chains, world_size, cost=len):
    """Longest-processing-time-first assignment of chains to ranks.
    Returns (rank_of_chain list, per-rank load list).  Deterministic."""
    order = sorted(range(len(chains)), key=lambda i: (-cost(chains[i]), i))
    load = [0] * world_size
    owner = [0] * len(chains)
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += cost(chains[i])
    return owner, load


def pictures_for_rank(pictures, rank, world_size):
    """The picture numbers rank `rank` decodes, in coded order."""
    pictures = list(pictures)
    chains = reference_chains(pictures)
    owner, _ = assign_chains(chains, world_size)
    mine = set()
    for c, r in zip(chains, owner):
        if r == rank:
            mine.update(c)
    return [num for num, _ in pictures if num in mine]


def batch_slice(n_items, rank, world_size):
    """Independent pictures (intra-only, low-delay, or a bench batch): contiguous,
    balanced slices.  Returns range(start, stop)."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))
