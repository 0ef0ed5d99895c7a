"""BASELINE.json's full C2 size (10 000 targets x 10 kb, 50 pairs): properties that do not need the CPU
oracle (which would take hours here) -- the seed scan and the bit-sliced scan build the same word DB and
amplification bits, the fused asynchronous pass equals the separate calls, evaluation is independent per
target (a shard screened alone gives the same bits for its targets), a pass is idempotent, and the
coverage checksum is the weighted popcount of the bits; the oracle itself screens a sample of the same workload, to which
the full-size bits are tied.  Run on the GPU box with `-m gpu`."""
import os

import numpy as np
import pytest

from pcramp_amd import api, synth

pytestmark = pytest.mark.gpu


def _screener(scan=None, seed_form=None, tables=None):
    """scan: PCRAMP_SCAN (2 = bit-sliced scan for everything); seed_form=1: the first form of the seed scan (k_seed)
    instead of the second (k_seed2); tables="host": the first form's tables built on the host instead of by k_seed_tables."""
    old = {k: os.environ.get(k) for k in ("PCRAMP_SCAN", "PCRAMP_SEED", "PCRAMP_SEED_TABLES")}
    for k, v in (("PCRAMP_SCAN", scan), ("PCRAMP_SEED", seed_form), ("PCRAMP_SEED_TABLES", tables)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    try:
        return api.Screener(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def c2():
    return synth.workload("C2", 0, 1.0)


def _screen(d, wl, lo=0, hi=None, thr_t=1.0, mult=0.9):
    hi = wl["T"] if hi is None else hi
    nb = (wl["L"] + 1) // 2
    d.load_sequences(wl["packed"][lo * nb:hi * nb], wl["byte_offsets"][lo:hi] - wl["byte_offsets"][lo], wl["lengths"][lo:hi])
    thr = float(np.float32(thr_t) * np.float32(mult))
    n = d.select_words(wl["pairs"], thr, 18)
    bits, fr, rf, cov = d.amplify(wl["pairs"], thr_t, thr_t, 80, 200, False)
    return n, fr, rf, cov


def test_full_c2_properties(c2, oracle):
    import torch
    from test_gpu_configs import _sample, _check_targets_against_oracle
    a, b, s1, s1h = _screener(None), _screener(2), _screener(None, seed_form=1), _screener(None, seed_form=1, tables="host")
    try:
        n3, fr3, rf3, cov3 = _screen(a, c2)
        n2, fr2, rf2, cov2 = _screen(b, c2)
        # seed scan == bit-sliced scan at full size
        assert n3 == n2 and n3 > 1000
        assert np.array_equal(fr3, fr2) and np.array_equal(rf3, rf2) and np.array_equal(cov3, cov2)
        # second form of the seed scan (9-gram seeds, tables built in LDS) == first form (8-gram seeds, host-built tables),
        # word DB entry for entry
        n1, fr1, rf1, cov1 = _screen(s1, c2)
        assert n1 == n3 and np.array_equal(fr1, fr3) and np.array_equal(rf1, rf3)
        assert s1.entries() == a.entries()
        s1.close()
        n1h, fr1h, rf1h, _ = _screen(s1h, c2)                    # ... == first form with tables built on the host
        assert n1h == n3 and s1h.entries() == a.entries()
        s1h.close()
        assert fr3.any() or rf3.any()
        # the tie to the CPU oracle (as for C3 / C4 / C5 in test_gpu_configs.py): a sample of the same workload -- the targets the
        # first pairs were cut from plus family mates -- equals the oracle bit for bit (word DB, bits, coverage), and the full-size
        # bits restricted to the sample equal the sample screened alone
        _check_targets_against_oracle(oracle, c2, _sample(c2), fr3, rf3)
        # idempotence
        n3b, fr3b, rf3b, cov3b = _screen(a, c2)
        assert n3b == n3 and np.array_equal(fr3b, fr3) and np.array_equal(rf3b, rf3) and np.array_equal(cov3b, cov3)
        # coverage == weighted popcount of the union of both orientations (all weights 1 here)
        assert np.array_equal(cov3, (fr3 | rf3).sum(axis=1).astype(np.float32))
        # fused asynchronous pass == separate calls
        words = int(a.bitset_words())
        P = len(c2["pairs"])
        out = torch.full((2, P, words), -1, dtype=torch.int64, device="cuda:0")
        thr = float(np.float32(1.0) * np.float32(0.9))
        a.screen_device(c2["pairs"], thr, out[0].data_ptr(), out[1].data_ptr(), 1.0, 1.0, 80, 200, False)
        a.synchronize()
        torch.cuda.synchronize()
        w = out.cpu().numpy().view(np.uint64)
        got = [np.stack([api.bits_to_bool(w[k, i], c2["T"]) for i in range(P)]) for k in range(2)]
        assert np.array_equal(got[0], fr3) and np.array_equal(got[1], rf3)
        # independence per target: a shard screened alone (same pairs) amplifies the same targets.  The word DB
        # couples the pairs of a batch, not the targets (select_words runs per sequence), so the bits are equal.
        lo, hi = 2560, 5120
        _, frs, rfs, _ = _screen(b, c2, lo, hi)
        assert np.array_equal(frs, fr3[:, lo:hi]) and np.array_equal(rfs, rf3[:, lo:hi])
    finally:
        a.close()
        b.close()
        s1.close()
        s1h.close()


@pytest.mark.parametrize("thr_t", [0.9, 0.85])
def test_full_c2_low_threshold(c2, thr_t):
    """select threshold 0.81 / 0.765 (4-5 / 5-6 mismatching slots allowed): dense seed tables (padded budget-1 blocks, budget-2
    blocks; the densest orientations handed to the bit-sliced scan) == bit-sliced scan alone, word DB entry for entry."""
    a, b, h = _screener(None), _screener(2), _screener(None, tables="host")
    try:
        hi = 2560
        n3, fr3, rf3, cov3 = _screen(a, c2, 0, hi, thr_t)
        n2, fr2, rf2, cov2 = _screen(b, c2, 0, hi, thr_t)
        assert n3 == n2 and n3 > 1000
        assert a.entries() == b.entries()
        assert np.array_equal(fr3, fr2) and np.array_equal(rf3, rf2) and np.array_equal(cov3, cov2)
        nh, frh, rfh, covh = _screen(h, c2, 0, hi, thr_t)          # tables built on the host == built by k_seed_tables
        assert nh == n3 and h.entries() == a.entries()
    finally:
        a.close()
        b.close()
        h.close()


def test_megabase_sequences_and_table_overflow():
    """(a) Genome-sized sequences (C4's shape, 3 Mb each): seed scan == bit-sliced scan, sites planted near the
    far end are found.  (b) More seed codes than the table admits (hundreds of pairs with 5'/3' shifts): the
    pass falls back to the bit-sliced scan for everything and still equals it."""
    rs = np.random.RandomState(123)
    L = 3_000_000
    from pcramp_amd import words as W
    codes = [2 ** rs.randint(0, 4, size=L).astype(np.uint8) for _ in range(3)]
    codes.append(codes[0].copy())
    flip = rs.randint(0, L, size=L // 50)
    codes[3][flip] = 2 ** rs.randint(0, 4, size=flip.size).astype(np.uint8)     # ~1.5 % divergent copy
    packed = np.concatenate([W.pack_codes(c) for c in codes])
    nb = (L + 1) // 2
    off = np.arange(4, dtype=np.uint64) * np.uint64(nb)
    lens = np.full(4, L, dtype=np.uint64)
    comp = {1: 8, 2: 4, 4: 2, 8: 1}
    pairs = []
    for pos in (100, 1_000_000, 2_999_700, L - 200, 1234567):
        f = codes[0][pos:pos + 21]
        r = np.array([comp[int(v)] for v in codes[0][pos + 120:pos + 142][::-1]], dtype=np.uint8)
        pairs.append((W.centered_word(f), W.centered_word(r)))
    a, b = _screener(None), _screener(2)
    try:
        res = []
        for d in (a, b):
            d.load_sequences(packed, off, lens)
            n = d.select_words(pairs, float(np.float32(1.0) * np.float32(0.9)), 18)
            bits, fr, rf, cov = d.amplify(pairs, 1.0, 1.0, 80, 200, False)
            res.append((n, d.entries(), fr, rf))
        assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
        assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
        assert res[0][2][:, 0].all()                       # every planted pair amplifies sequence 0 in F(+)/R(-) orientation
        # (b) table overflow
        wl = synth.workload("C2", 0, 0.02)
        many = []
        for k in range(900):
            i = rs.randint(0, wl["T"])
            c = synth.sequence_codes(wl["packed"], wl["byte_offsets"], wl["lengths"], i)
            p = rs.randint(0, wl["L"] - 200)
            many.append((W.centered_word(c[p:p + rs.randint(18, 26)]),
                         W.centered_word(np.array([comp[int(v)] for v in c[p + 120:p + 120 + rs.randint(18, 26)][::-1]], dtype=np.uint8))))
        out = []
        for d in (a, b):
            d.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
            n = d.select_words(many, float(np.float32(1.0) * np.float32(0.9)), 18, True, True)
            out.append((n, d.entries()))
        assert out[0] == out[1] and out[0][0] > 0
    finally:
        a.close()
        b.close()
