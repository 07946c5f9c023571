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


def _worker_intervals(rank, world, port, total, depth, q):
    """bench.py's N > 1 path on CPU: ONE interval set partitioned by shard.partition_intervals, every rank making only its own locus
    range of the global pileup and calling it (oracle as the stand-in compute), then the summary all-reduce."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pisces_amd import synth
    from tests import orc
    origin = synth.READ_LEN + 1
    n_amp = -(-total // synth.READ_LEN)
    intervals = [(origin + a * synth.READ_LEN, origin + min((a + 1) * synth.READ_LEN, total) - 1) for a in range(n_amp)]
    lo, hi, _ = shard.partition_intervals(intervals, world, block_size=1000)[rank]
    lo, hi = max(lo, intervals[0][0]), min(hi, intervals[-1][1])
    p = synth.make_pileup(hi - lo + 1, depth, seed=78, first_locus=lo - origin, total_loci=total)
    pos, tup = synth.observations_of(p)
    ref = p.ref.numpy()
    ref_full = np.full(p.ref_start - 1 + len(ref), ord("N"), dtype=np.uint8)
    ref_full[p.ref_start - 1:] = ref
    out, nloci = orc.run_observations(pos, tup, ref_full, p.region_start, p.n_loci, _abi.default_config())
    summary = shard.reduce_summary(torch.tensor([len(out), nloci, len(pos), p.n_tiles], dtype=torch.int64))
    gathered = [None] * world
    dist.all_gather_object(gathered, out.tobytes())
    if rank == 0:
        q.put((summary.tolist(), b"".join(gathered)))
    dist.destroy_process_group()


def test_interval_set_partitioned_across_two_ranks_matches_single():
    from pisces_amd import synth
    from tests import orc
    total, depth = 3300, 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_intervals, args=(r, 2, port, total, depth, q)) for r in range(2)]
    for p in procs:
        p.start()
    summary, blob = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    p = synth.make_pileup(total, depth, seed=78)
    pos, tup = synth.observations_of(p)
    exp, nloci = orc.run_observations(pos, tup, p.ref.numpy(), p.region_start, total, _abi.default_config())
    assert summary[:3] == [len(exp), nloci, len(pos)]
    assert blob == exp.tobytes()      # rank-order concatenation of the shards == the unsharded result


def test_tile_range_partition_properties():
    for n in (0, 1, 7, 64, 1563):
        for w in (1, 2, 3, 4, 8):
            ranges = [shard.tile_range(n, r, w) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_interval_partition_is_block_aligned_balanced_and_complete():
    ivs = [(150, 420), (980, 1310), (2500, 2790), (5001, 5600), (9100, 9200), (12000, 14050)]
    for world in (1, 2, 3, 4, 8):
        parts = shard.partition_intervals(ivs, world)
        assert len(parts) == world
        owned = [p for p in parts if p[1] >= p[0]]
        assert all(lo % 1000 == 1 and hi % 1000 == 0 for lo, hi, _ in owned)               # cuts on the block grid
        assert all(a[1] < b[0] for a, b in zip(owned, owned[1:]))                           # contiguous, ordered, disjoint
        covered = sorted(iv for _, _, clipped in owned for iv in clipped)
        loci = lambda L: sum(b - a + 1 for a, b in L)
        assert loci(covered) == loci(ivs) and covered[0][0] == ivs[0][0] and covered[-1][1] == ivs[-1][1]
        if world <= 4:
            sizes = [loci(c) for _, _, c in owned]
            assert max(sizes) <= 2 * (loci(ivs) / len(owned)) + 1000                          # within a block of balance
    heavy = shard.partition_intervals([(1, 1000), (1001, 2000), (2001, 3000)], 2, weights=[10, 1, 1])
    assert heavy[0][1] == 1000 and heavy[1][0] == 1001                                       # depth-weighted: the deep block stands alone


def test_shards_with_halo_reads_reproduce_the_single_run():
    """SURVEY 8e end to end on the CPU: reads near a cut are handed to both shards, each shard calls only the loci it owns, and the
    rank-order concatenation is the unsharded result; every read is counted once (by the shard that owns its start)."""
    from tests import orc
    rng = np.random.default_rng(12)
    ref = rng.choice(list(b"ACGT"), 4000).astype(np.uint8)
    reads = []
    for i in range(6000):
        start = int(rng.integers(1, 3850))
        seq = ref[start - 1: start - 1 + 120].copy()
        for k in range(120):
            if rng.random() < 0.01:
                seq[k] = rng.choice(list(b"ACGT"))
        reads.append({"pos": start, "cigar": [("M", 120)], "seq": bytes(seq).decode(), "quals": [37] * 120, "reverse": bool(i % 2)})
    reads.sort(key=lambda r: r["pos"])
    cfg = _abi.default_config()
    whole, _ = orc.run_reads(_abi.ReadBatch(reads), ref, 1, 4000, cfg)
    starts = np.array([r["pos"] for r in reads])
    ends = starts + 119
    parts = shard.partition_intervals([(1, 4000)], 3)
    got, counted = [], 0
    for lo, hi, clipped in parts:
        idx, owner = shard.reads_for_shard(starts, ends, lo, hi, halo=130)
        counted += int(owner.sum())
        recs, _ = orc.run_reads(_abi.ReadBatch([reads[i] for i in idx]), ref, lo, hi - lo + 1, cfg)
        got.append(recs)
    assert counted == len(reads)
    assert np.concatenate(got).tobytes() == whole.tobytes()


# ---- bench.py --gpus N starts its own ranks ----

def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class _Args:
    def __init__(self, gpus):
        self.gpus = gpus


def test_bench_launcher_refuses_more_ranks_than_devices(monkeypatch, capsys):
    """`python bench.py --gpus 2` on a box with one device exits non-zero with a message that names the count, instead of running one rank."""
    bench = _bench_module()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("PISCES_BENCH_ONE_DEVICE", raising=False)
    started = []
    rc = bench.launch_ranks(_Args(2), ["--gpus", "2"], device_count=lambda: 1, run=lambda cmd, env: started.append(cmd) or 0)
    assert rc not in (None, 0) and not started
    err = capsys.readouterr().err
    assert "--gpus 2" in err and "pisces_hip_device_count() = 1" in err


def test_bench_launcher_starts_one_rank_per_gpu(monkeypatch):
    """With enough devices (or PISCES_BENCH_ONE_DEVICE=1) the command line is re-run under torch.distributed.run, N processes on the loopback."""
    bench = _bench_module()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    started = []

    def run(cmd, env):
        started.append((cmd, env))
        return 0

    assert bench.launch_ranks(_Args(4), ["--gpus", "4", "--steps", "5"], device_count=lambda: 8, run=run) == 0
    cmd, env = started[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("PISCES_BENCH_ONE_DEVICE", "1")
    assert bench.launch_ranks(_Args(2), ["--gpus", "2"], device_count=lambda: 1, run=run) == 0 and len(started) == 2


def test_bench_launcher_leaves_a_launched_rank_and_n1_alone(monkeypatch, capsys):
    """N = 1 runs in this process as before; under a launcher (WORLD_SIZE set) the rank runs; a launcher that disagrees with --gpus is an error."""
    bench = _bench_module()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    boom = lambda *a, **k: (_ for _ in ()).throw(AssertionError("nothing may be started"))
    assert bench.launch_ranks(_Args(1), [], device_count=boom, run=boom) is None
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert bench.launch_ranks(_Args(2), ["--gpus", "2"], device_count=boom, run=boom) is None
    assert bench.launch_ranks(_Args(4), ["--gpus", "4"], device_count=boom, run=boom) == 2
    assert "WORLD_SIZE=2" in capsys.readouterr().err


def _spawned_rank_script():
    return ("import os, sys, torch.distributed as dist\n"
            "dist.init_process_group('gloo')\n"
            "import torch\n"
            "t = torch.tensor([dist.get_rank() + 1]); dist.all_reduce(t)\n"
            "assert int(t) == sum(range(1, dist.get_world_size() + 1))\n"
            "open(os.path.join(sys.argv[1], 'rank%d' % dist.get_rank()), 'w').write(str(int(t)))\n")


def test_bench_launcher_command_brings_up_a_gloo_world_of_two(tmp_path, monkeypatch):
    """The launcher's own command form (torch.distributed.run on 127.0.0.1 with a free port) brings up two ranks that find each other:
    run here with a stand-in script in place of bench.py (whose ranks need a GPU each)."""
    import subprocess
    bench = _bench_module()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    script = tmp_path / "rank.py"
    script.write_text(_spawned_rank_script())

    def run(cmd, env):
        i = cmd.index(os.path.abspath(bench.__file__))
        return subprocess.call(cmd[:i] + [str(script), str(tmp_path)], env=env, timeout=240)

    assert bench.launch_ranks(_Args(2), ["--gpus", "2"], device_count=lambda: 2, run=run) == 0
    assert (tmp_path / "rank0").read_text() == "3" and (tmp_path / "rank1").read_text() == "3"
