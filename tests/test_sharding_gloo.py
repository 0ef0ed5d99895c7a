"""world_size-2 (and 3) CPU runs of the multi-GPU path over gloo: shard ranges, the single
all-gather of bitset words, and coverage re-summed from the gathered bits must reproduce the
unsharded result.  (The per-shard bits come from the CPU oracle here; on the GPU box they come
from pcr_amplify_device -- the collective and the host arithmetic are the code under test.)"""
import os
import random
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pack_bits(b):
    """bool[n] -> int64 words (bit i%64 of word i//64)."""
    n = b.size
    padded = np.zeros(((n + 63) // 64) * 64, dtype=np.uint8)
    padded[:n] = b
    return np.packbits(padded, bitorder="little").view(np.uint64).astype(np.int64)


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_lib import Oracle
    from pcramp_amd import api, shard
    from testdata import family_targets, sample_pair
    orc = Oracle()
    rng = random.Random(77)
    seqs = family_targets(rng, 5, 40, 300, div=0.04)          # 200 targets -> 4 bitset words
    rng2 = random.Random(78)
    lens = [len(s) for s in seqs]
    weights = np.array([1.0 + 0.123 * (i % 7) for i in range(len(seqs))], np.float32)
    pairs = []
    while len(pairs) < 6:
        p = sample_pair(rng2, rng2.choice(seqs), amplicon=(80, 200))
        if p:
            pairs.append((orc.centered_word(p[0]), orc.centered_word(p[1])))
    ranges = shard.shard_ranges(lens, world)
    lo, hi = ranges[rank]
    # this rank's shard
    s = orc.session(target_threshold=0.9)
    for i in range(lo, hi):
        s.add_target(seqs[i], float(weights[i]))
    s.select(pairs)
    fr = np.stack([pack_bits((s.target_match(p, orient=True)[1] & 1) != 0) for p in pairs]) if hi > lo else np.zeros((len(pairs), 0), np.int64)
    rf = np.stack([pack_bits((s.target_match(p, orient=True)[1] & 2) != 0) for p in pairs]) if hi > lo else np.zeros((len(pairs), 0), np.int64)
    local = torch.from_numpy(np.stack([fr, rf]))
    full = shard.gather_bitsets(local, ranges)                  # [2, P, ceil(n/64)] on every rank
    full = full.numpy().view(np.uint64)
    # the unsharded truth
    t = orc.session(target_threshold=0.9)
    for i, q in enumerate(seqs):
        t.add_target(q, float(weights[i]))
    t.select(pairs)
    ok = True
    for k, p in enumerate(pairs):
        bits, ori = t.target_match(p, orient=True)
        got = api.bits_to_bool(full[0, k] | full[1, k], len(seqs))
        ok &= bool((got == bits.astype(bool)).all())
        cov = api.coverage_from_bits(full[0, k], full[1, k], weights)
        t2 = orc.session(target_threshold=0.9, search_multiplier=1.0)
        for i, q in enumerate(seqs):
            t2.add_target(q, float(weights[i]))
        t2.select(pairs, threshold=np.float32(0.9) * np.float32(0.9))
        ok &= (cov == np.float32(t2.target_coverage(p)))
    with open(os.path.join(tmp, "ok%d" % rank), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def _bench_worker(rank, world, port, tmp):
    """The N>1 path of bench.py with its own helpers: a synth.GlobalSet cut by shard.shard_ranges, every rank screening
    its block (the CPU oracle stands in for pcr_screen_device), the [world, K, 2, P, wmax] all-gather layout,
    bench.assemble_gathered and bench.check_against_unsharded against the unsharded screen."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle_lib import Oracle
    from pcramp_amd import shard, synth
    orc = Oracle()
    gs = synth.GlobalSet("C1", world, scale=1.3)               # world x 130 targets of 1 kb: blocks are not multiples of 64
    pairs = gs.pairs()
    ranges = shard.shard_ranges(gs.lengths, world)
    lo, hi = ranges[rank]
    wmax = max((b - a + 63) // 64 for a, b in ranges)
    nb = (gs.L + 1) // 2

    def screen(packed, n):
        s = orc.session()
        for i in range(n):
            s.add_target_packed(packed[i * nb:(i + 1) * nb], gs.L)
        s.select(pairs)
        ori = [s.target_match(p, orient=True)[1] for p in pairs]
        cov = np.array([s.target_coverage(p) for p in pairs], np.float32)
        return np.stack([(o & 1) != 0 for o in ori]), np.stack([(o & 2) != 0 for o in ori]), cov
    packed, _, _ = gs.members(lo, hi)
    fr, rf, _ = screen(packed, hi - lo)
    K = 2
    local = torch.zeros((K, 2, len(pairs), wmax), dtype=torch.int64)
    for k in range(len(pairs)):
        local[0, 0, k, :(hi - lo + 63) // 64] = torch.from_numpy(pack_bits(fr[k]))
        local[0, 1, k, :(hi - lo + 63) // 64] = torch.from_numpy(pack_bits(rf[k]))
    gathered = torch.zeros((world, K, 2, len(pairs), wmax), dtype=torch.int64)
    dist.all_gather_into_tensor(gathered.view(-1), local.view(-1))
    full = bench.assemble_gathered(gathered[:, 0], ranges)
    ok = True
    if rank == 0:
        upacked, _, _ = gs.members(0, gs.T)
        ufr, urf, ucov = screen(upacked, gs.T)
        try:
            n_set = bench.check_against_unsharded(full, gs.T, list(range(gs.T)), ufr, urf, ucov)
            ok = n_set > 0
            # the sampled form (C4): first and last member of every block
            ids = sorted(set([a for a, b in ranges if b > a] + [b - 1 for a, b in ranges if b > a]))
            bench.check_against_unsharded(full, gs.T, ids, ufr[:, ids], urf[:, ids], None)
            # and it does notice a wrong bit
            bad = full.copy()
            bad[0, 0, 0] ^= np.uint64(1)
            try:
                bench.check_against_unsharded(bad, gs.T, list(range(gs.T)), ufr, urf, ucov)
                ok = False
            except AssertionError:
                pass
        except AssertionError:
            ok = False
    with open(os.path.join(tmp, "ok%d" % rank), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_sharding_helpers(tmp_path, world):
    port = 31500 + random.randint(0, 2000)
    mp.spawn(_bench_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "ok%d" % r)).read() == "1"


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_unsharded(tmp_path, world):
    port = 29500 + random.randint(0, 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "ok%d" % r)).read() == "1"


def test_shard_ranges_properties():
    from pcramp_amd import shard
    rng = np.random.RandomState(1)
    for n in (0, 1, 63, 64, 65, 1000, 5000):
        lens = rng.randint(100, 10000, size=n)
        for world in (1, 2, 4, 8):
            r = shard.shard_ranges(lens, world)
            assert len(r) == world and r[0][0] == 0 and r[-1][1] == n
            for (a, b), (c, d) in zip(r, r[1:]):
                assert b == c and a <= b
            for lo, hi in r:
                assert lo % 64 == 0 or lo == n
