"""Nearest-neighbour thermodynamics on the GPU against the CPU oracle (itself pinned bit-exact to
the compiled reference): Tm / dH / dS of the perfect-match duplex, hairpin Tm, homo- and
heterodimer Tm, and the three filters.  north_star tolerance: 1e-6 (degC, kcal/mol); the tests
also count how many values are bit-identical."""
import random

import numpy as np
import pytest

from pcramp_amd import api
from testdata import rand_seq, mutate, revcomp

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def dev():
    s = api.Screener(0)
    yield s
    s.close()


def hairpin_prone(rng, n):
    stem = rand_seq(rng, rng.randint(4, 8))
    loop = rand_seq(rng, rng.randint(3, 7))
    s = rand_seq(rng, rng.randint(0, 4)) + stem + loop + mutate(rng, revcomp(stem), 0.1) + rand_seq(rng, 8)
    return s[:n] if len(s) >= 12 else s + rand_seq(rng, 12)


def close(a, b):
    return abs(float(a) - float(b)) <= TOL * max(1.0, abs(float(b)))


@pytest.mark.parametrize("salt,strand", [(0.05, 9e-7), (0.2, 4.5e-7), (0.01, 2e-8)])
def test_oligo_thermo_matches_oracle(dev, oracle, salt, strand):
    rng = random.Random(61)
    seqs = []
    for it in range(1500):
        n = rng.randint(12, 32)
        m = rng.random()
        if m < 0.6:
            seqs.append(rand_seq(rng, n))
        elif m < 0.8:
            seqs.append(hairpin_prone(rng, n)[:32])
        else:
            seqs.append((rand_seq(rng, rng.randint(1, 4)) * 32)[:n])
    words = [oracle.centered_word(s) for s in seqs]
    res = dev.is_valid(words, True, salt=salt, primer_strand=strand)
    exact = tot = 0
    for s, r in zip(seqs, res):
        o = oracle.thermo_full(s, salt, strand)
        got = [r["tm"], r["dH"], r["dS"], r["dG"], r["hairpin_tm"], r["homodimer_tm"]]
        want = [o[0], o[1], o[2], o[3], o[4], o[7]]
        for g, w in zip(got, want):
            assert close(g, w), (s, got, want)
            exact += int(np.float32(g) == np.float32(w)); tot += 1
        valid = (50.0 <= o[0] <= 75.0) and not (o[4] > 40.0) and not (o[7] > 40.0)
        assert r["valid"] == valid, s
    assert exact == tot, "%d of %d values bit-identical" % (exact, tot)


def test_is_valid_degenerate(dev, oracle):
    rng = random.Random(62)
    words, kws = [], []
    for it in range(200):
        w = oracle.centered_word(rand_seq(rng, rng.randint(18, 25), p_degen=0.08))
        if oracle.word_degeneracy(w) <= 16:
            words.append(w)
    for chk in (True, False):
        res = dev.is_valid(words, chk, tm_min=45.0, tm_max=70.0, max_hairpin=35.0, max_dimer=35.0)
        npass = 0
        for w, r in zip(words, res):
            want = oracle.is_valid(w, tm_min=45.0, tm_max=70.0, max_hairpin=35.0, max_dimer=35.0, check_homo_dimer=chk)
            assert int(r["valid"]) == want
            assert r["n"] == int(oracle.word_degeneracy(w))
            npass += want
        assert 5 < npass < len(words) - 5


def test_dimer_and_multiplex(dev, oracle):
    rng = random.Random(63)
    pairs = []
    for it in range(150):
        f = rand_seq(rng, rng.randint(18, 25), p_degen=0.04)
        if it % 3 == 0:   # a reverse primer that partly pairs with the forward one
            core = mutate(rng, revcomp(f[2:14].replace("N", "A")), 0.1)
            r = (rand_seq(rng, 5) + core + rand_seq(rng, 6))[:25]
        else:
            r = rand_seq(rng, rng.randint(18, 25), p_degen=0.04)
        fw, rw = oracle.centered_word(f), oracle.centered_word(r)
        if oracle.word_degeneracy(fw) * oracle.word_degeneracy(rw) <= 64:
            pairs.append((fw, rw))
    tm = dev.max_dimer_tm(pairs)
    nz = 0
    for p, g in zip(pairs, tm):
        w = oracle.max_dimer_tm(p)
        assert close(g, w) and np.float32(g) == np.float32(w)
        nz += w > 0
    assert nz > 20
    a, b = pairs[:60], pairs[60:120]
    for md in (10.0, 25.0, 40.0):
        ok = dev.multiplex_compatible(a, b, max_dimer=md)
        for x, y, g in zip(a, b, ok):
            assert int(g) == oracle.multiplex_compatible(x, y, max_dimer=md)


def test_known_values(dev, oracle):
    """SURVEY.md 8c, captured from the compiled reference."""
    r = dev.is_valid([oracle.centered_word("AGAAGGCTCGCCAAAATAAACG"), oracle.centered_word("GCGCGCAAAAGCGCGC")])
    assert abs(float(r[0]["tm"]) - 59.139496) < 1e-4 and abs(float(r[0]["dH"]) + 172.899994) < 1e-4
    assert abs(float(r[1]["hairpin_tm"]) - 67.163818) < 1e-3 and abs(float(r[1]["homodimer_tm"]) - 40.364777) < 1e-3


def test_bad_oligo_rejected(dev, oracle):
    with pytest.raises(api.PcrError):
        dev.is_valid([(0, 0)])


def test_results_do_not_depend_on_the_batch(dev, oracle):
    """One kernel form (a wave per job) serves every batch size; a big call differs from a small one in how the jobs reach it
    (1 200 oligos: job records in the mapped input buffer, homodimer halves mirrored by the kernel; 24 000 oligos: 48 000
    wave jobs, several per resident block): the same oligos give the same bits, and a sample of them equals the oracle."""
    rng = random.Random(88)
    seqs = []
    for _ in range(1200):
        n = rng.randint(12, 32)
        seqs.append(rand_seq(rng, n) if rng.random() < 0.7 else hairpin_prone(rng, n)[:32])
    words = [oracle.centered_word(s) for s in seqs]
    small = dev.is_valid(words, True)
    big = dev.is_valid(words * 20, True)
    for k in range(20):
        assert big[k * 1200:(k + 1) * 1200] == small
    for s, r in list(zip(seqs, small))[::25]:
        o = oracle.thermo_full(s, 0.05, 9e-7)
        assert [r["tm"], r["dH"], r["dS"], r["hairpin_tm"], r["homodimer_tm"]] == [o[0], o[1], o[2], o[4], o[7]], s
    pairs = [(words[i], words[(7 * i + 3) % 1200]) for i in range(1200)]
    d_small = dev.max_dimer_tm(pairs)
    d_big = dev.max_dimer_tm(pairs * 20)
    assert np.array_equal(np.tile(d_small, 20), d_big)
    for (a, b), d in list(zip(pairs, d_small))[::40]:
        assert np.float32(oracle.max_dimer_tm((a, b))) == d
