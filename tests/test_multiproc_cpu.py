"""The N>1 path on CPU: world_size-2 gloo. Interval sharding has no data-path collective; the only exchange is
the summary all-reduce. Each rank calls its shard (here with the oracle as the stand-in compute, since there
is no GPU in this container) and the reduced summary must equal the unsharded run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pisces_amd import _abi, shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_loci, depth, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pisces_amd import synth
    from tests import orc
    p = synth.make_pileup(n_loci, depth, seed=77)           # every rank sees the same position-sorted input
    tiles = p.tiles.numpy().view(_abi.TILE_DTYPE)
    lo, hi = shard.tile_range(p.n_tiles, rank, world)        # contiguous tile range owned by this rank
    pos, tup = synth.observations_of(p)
    own_lo = int(tiles[lo]["start_position"]) if lo < hi else 0
    own_hi = int(tiles[hi - 1]["start_position"] + tiles[hi - 1]["n_loci"]) if lo < hi else 0
    m = (pos >= own_lo) & (pos < own_hi)
    out, nloci = orc.run_observations(pos[m], tup[m], p.ref.numpy(), own_lo, max(own_hi - own_lo, 1), _abi.default_config())
    summary = torch.tensor([len(out), nloci, int(m.sum()), hi - lo], dtype=torch.int64)
    summary = shard.reduce_summary(summary)                  # the per-chromosome summary reduce
    gathered = [None] * world
    dist.all_gather_object(gathered, out.tobytes())
    if rank == 0:
        q.put((summary.tolist(), b"".join(gathered)))
    dist.destroy_process_group()


def test_interval_sharding_two_ranks_matches_single():
    from pisces_amd import synth
    from tests import orc
    n_loci, depth = 700, 30
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_loci, depth, q)) for r in range(2)]
    for p in procs:
        p.start()
    summary, blob = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    p = synth.make_pileup(n_loci, depth, seed=77)
    pos, tup = synth.observations_of(p)
    exp, nloci = orc.run_observations(pos, tup, p.ref.numpy(), p.region_start, n_loci, _abi.default_config())
    assert summary == [len(exp), nloci, len(pos), p.n_tiles]
    assert blob == exp.tobytes()      # shard-order concatenation == genomic order == the unsharded result


def test_tile_range_partition_properties():
    for n in (0, 1, 7, 64, 1563):
        for w in (1, 2, 3, 4, 8):
            ranges = [shard.tile_range(n, r, w) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
