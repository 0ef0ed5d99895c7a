"""Sizes the r02 suite only reached inside bench.py's timed blocks, now with their results checked:

* C5 at its FULL size (100 000 targets x 10 kb, IUPAC primers) on one GPU;
* the reference's default trial batch -- 1 000 assays per design iteration (pcramp.h:32, main.cpp:644-676) = 4 000
  orientations, sixteen-odd launches of the second form of the seed scan -- on a C5 shard;
* pcr_optimize_batch with >= 128 assays (host worker threads, is_valid cache growth) and with its task lists shrunk
  so that both overflow branches of k_pair_moves_batch run.

Each is tied to the CPU oracle the way tests/test_gpu_configs.py does it: a sample of the same workload equals the
oracle bit for bit, and the full-size result restricted to the sample equals the sample screened alone."""
import os
import random

import numpy as np
import pytest

from pcramp_amd import api, synth, words as W
from test_gpu_configs import _screener, _sample, _target_pass, _fused, _check_targets_against_oracle, _oracle_session

pytestmark = pytest.mark.gpu


def test_c5_full_size(oracle):
    """BASELINE.json configs[4] unsharded: 100 000 x 10 kb = eight blocks of the C5 shard workload (synth.GlobalSet, what
    bench.py --config C5 cuts over the ranks), block 0's 50 IUPAC primer pairs."""
    gs = synth.GlobalSet("C5_shard", 8)
    wl0 = gs.block(0)
    pairs = gs.pairs()
    T = gs.T
    assert (T, gs.L) == (100000, 10000)
    packed, off, lens = gs.members(0, T)
    wl = dict(packed=packed, byte_offsets=off, lengths=lens, pairs=pairs, T=T, L=gs.L)
    a, b = _screener(None), _screener(2)
    try:
        a.load_sequences(packed, off, lens)
        n3, fr3, rf3, cov3 = _target_pass(a, wl)
        b.load_sequences(packed, off, lens)
        n2, fr2, rf2, cov2 = _target_pass(b, wl)
        assert n3 == n2 and n3 > 5000                                    # seed scan == bit-sliced scan
        assert np.array_equal(fr3, fr2) and np.array_equal(rf3, rf2) and np.array_equal(cov3, cov2)
        assert np.array_equal(cov3, (fr3 | rf3).sum(axis=1).astype(np.float32)) and (fr3 | rf3).any()
        got = _fused(a, wl, T)                                           # fused asynchronous pass == separate calls
        assert np.array_equal(got[0], fr3) and np.array_equal(got[1], rf3)
        # the oracle tie on block 0's sample (global index = block-0 index)
        idx = _sample(wl0)
        _check_targets_against_oracle(oracle, wl0, idx, fr3, rf3)
        # evaluation is independent per target: block 5 screened alone gives its targets' bits
        lo, hi = 5 * gs.T_block, 6 * gs.T_block
        b.load_sequences(*gs.members(lo, hi))
        _, frs, rfs, _ = _target_pass(b, wl)
        assert np.array_equal(frs, fr3[:, lo:hi]) and np.array_equal(rfs, rf3[:, lo:hi])
    finally:
        a.close()
        b.close()


def test_trial_batch_of_1000_assays(oracle):
    """select_words + find_target_match for the 1 000 trial assays of one design iteration on a C5 shard: the seed scan in
    groups of whole orientations == the bit-sliced scan (word DB entry for entry, bits), and == the oracle on a sample."""
    wl = synth.workload("C5_shard")
    T = wl["T"]
    thr = float(np.float32(1.0) * np.float32(0.9))
    a, b = _screener(None), _screener(2)
    try:
        a.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
        trial, _, info = a.random_assays(2025, 1000)                      # the sampler's trial assays (main.cpp:538-550)
        assert len(trial) == 1000 and len(set(trial)) > 900
        n3 = a.select_words(trial, thr, 18)
        ea = a.entries()
        _, fr3, rf3, cov3 = a.amplify(trial, 1.0, 1.0, 80, 200, False)
        b.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
        n2 = b.select_words(trial, thr, 18)
        assert n3 == n2 and n3 > 10000
        assert ea == b.entries()
        _, fr2, rf2, cov2 = b.amplify(trial, 1.0, 1.0, 80, 200, False)
        assert np.array_equal(fr3, fr2) and np.array_equal(rf3, rf2) and np.array_equal(cov3, cov2)
        hit = (fr3 | rf3).any(axis=1)
        assert hit.sum() >= 990                                           # a sampled assay amplifies the target it was cut from
        # the same batch once more (the per-oligo seed cache is warm now): idempotent
        assert a.select_words(trial, thr, 18) == n3 and a.entries() == ea
        # sample: the targets the first assays were cut from + family mates; the oracle screens all 1 000 assays on it
        idx = []
        for i in info[:4]:
            fam0 = int(i["sequence"]) // wl["family"] * wl["family"]
            for t in [int(i["sequence"])] + list(range(fam0, min(fam0 + 6, T))):
                if t not in idx:
                    idx.append(t)
        idx = sorted(idx)[:28]
        so = _oracle_session(oracle, wl, idx, wl["L"])
        n_o = so.select(trial)
        d = api.Screener(0)
        try:
            d.load_sequences(*synth.subset(wl, idx, wl["L"]))
            assert d.select_words(trial, thr, 18) == n_o
            assert d.entries() == so.db_entries()
            _, fr, rf, cov = d.amplify(trial, 1.0, 1.0, 80, 200, False)
            set_bits = 0
            for k, p in enumerate(trial):
                ob = so.target_match(p).astype(bool)
                assert ((fr[k] | rf[k]) == ob).all(), k
                set_bits += int(ob.sum())
            assert set_bits >= 8
            assert np.array_equal(fr3[:, idx], fr) and np.array_equal(rf3[:, idx], rf)
        finally:
            d.close()
    finally:
        a.close()
        b.close()


def _opt_case(oracle, n_base=8):
    from testdata import family_targets, sample_pair, mutate, rand_seq
    rng = random.Random(90210)
    seqs = family_targets(rng, 4, 10, 700, div=0.05)
    bgs = [mutate(rng, s, 0.12) for s in seqs[::3]] + [rand_seq(rng, 600) for _ in range(4)]
    txt = []
    while len(txt) < n_base:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            txt.append(p)
    # IUPAC primers (a 2-fold code at two positions) and damaged primers the search repairs
    iupac = {"A": "R", "C": "Y", "G": "R", "T": "Y"}
    for f, r in list(txt[:4]):
        k = rng.randrange(2, len(f) - 2)
        txt.append((f[:k] + iupac[f[k]] + f[k + 1:], r[:3] + iupac[r[3]] + r[4:]))
    for f, r in list(txt[:4]):
        txt.append((mutate(rng, f, 0.1), mutate(rng, r, 0.1)))
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in txt]
    tw = [1.0 + 0.25 * (i % 5) for i in range(len(seqs))]
    return seqs, tw, bgs, pairs


@pytest.mark.parametrize("case", [dict(degen=16), dict(degen=4, target_threshold=0.9, use_taq_mama=1)])
def test_optimize_batch_of_many_assays(oracle, case):
    """>= 128 assays in one pcr_optimize_batch call (host worker threads with their list joins, the is_valid cache's growth,
    the many-round expansion map) == one call per assay, and == the oracle for the distinct assays; then the same batch with
    the task lists of k_pair_moves_batch shrunk (PCRAMP_DEBUG_OPT_TASKS) so that the workgroup-list and the global-list
    overflow branches both run."""
    from oracle_lib import optimize as oracle_optimize
    from pcramp_amd import moves
    case = dict(case)
    sess = {k: case.pop(k) for k in ("target_threshold", "use_taq_mama") if k in case}
    o = dict(target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200, use_taq_mama=0,
             pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0, min_primer=18, optimize_5=1, optimize_3=1)
    o.update(sess)
    seqs, tw, bgs, pairs = _opt_case(oracle)
    batch = (pairs * 12)[:len(pairs) * 12]
    rnd = random.Random(5)
    rnd.shuffle(batch)
    assert len(batch) >= 128
    to, bo = oracle.session(**o), oracle.session(**o)
    for s, w in zip(seqs, tw):
        to.add_target(s, w)
    for s in bgs:
        bo.add_target(s, 1.0)
    to.select(pairs)
    bthr = float(np.float32(0.8) * np.float32(0.9))
    bo.select(pairs, threshold=bthr, min_len_override=16)
    thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
    kw = dict(target_threshold=o["target_threshold"], use_taq_mama=bool(o["use_taq_mama"]), **case)

    def run(env):
        old = os.environ.get("PCRAMP_DEBUG_OPT_TASKS")
        if env is None:
            os.environ.pop("PCRAMP_DEBUG_OPT_TASKS", None)
        else:
            os.environ["PCRAMP_DEBUG_OPT_TASKS"] = env
        try:
            d = api.Screener(0)
        finally:
            if old is None:
                os.environ.pop("PCRAMP_DEBUG_OPT_TASKS", None)
            else:
                os.environ["PCRAMP_DEBUG_OPT_TASKS"] = old
        try:
            d.load_texts(seqs, tw, which=api.TARGET)
            d.load_texts(bgs, [1.0] * len(bgs), which=api.BACKGROUND)
            d.select_words(pairs, thr, 18, True, True, which=api.TARGET)
            d.select_words(pairs, bthr, 16, True, True, which=api.BACKGROUND)
            single = {}
            for p in pairs:
                single[p] = moves.optimize(d, p, **kw)
            bp, bs, it = moves.optimize_batch(d, batch, **kw)
            return single, bp, bs, it
        finally:
            d.close()

    single, bp, bs, it = run(None)
    changed = 0
    for p in pairs:                                                      # one call per assay == the oracle
        po, so_ = oracle_optimize(oracle, to, bo, p, **case)
        assert single[p][0] == po and tuple(float(x) for x in single[p][1]) == so_
        changed += po != p
    assert changed >= 4
    for k, p in enumerate(batch):                                        # the batch == one call per assay
        assert bp[k] == single[p][0] and tuple(bs[k]) == tuple(single[p][1]), k
    assert len(set(it)) > 1
    # task lists of 3 per workgroup and 40 in all: nearly every task takes an in-place branch
    single2, bp2, bs2, it2 = run("3,40")
    assert bp2 == bp and [tuple(s) for s in bs2] == [tuple(s) for s in bs] and it2 == it
    assert all(single2[p][0] == single[p][0] for p in pairs)
