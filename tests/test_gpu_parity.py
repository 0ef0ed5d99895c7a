"""Parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs.  Bit-exact: word DB (select_words), amplification bits per orientation,
coverage floats.  Run on the GPU box with `-m gpu`.
"""
import random

import numpy as np
import pytest

from pcramp_amd import api, words as W, synth
from testdata import rand_seq, family_targets, sample_pair, revcomp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    s = api.Screener(0)
    yield s
    s.close()


def build_case(rng, oracle, n_fam=3, per_fam=6, L=700, n_pairs=12, extra=True, p_degen=0.0, primer_degen=0):
    seqs = family_targets(rng, n_fam, per_fam, L, div=0.04, p_degen=p_degen)
    if extra:
        seqs.append(rand_seq(rng, 40))
        seqs.append(rand_seq(rng, 20))
        seqs.append(rand_seq(rng, 333, p_degen=0.03, p_n=0.02))
        s = list(seqs[0]); s[350] = "-"; s[351] = "-"; seqs.append("".join(s))
        seqs.append("A" * 100 + rand_seq(rng, 200) + "N" * 40 + rand_seq(rng, 100))
    pairs_txt = []
    while len(pairs_txt) < n_pairs:
        p = sample_pair(rng, rng.choice(seqs[:n_fam * per_fam]))
        if p:
            f, r = list(p[0]), list(p[1])
            for o in (f, r):
                for _ in range(primer_degen):
                    k = rng.randrange(len(o))
                    if o[k] in "ACGT":
                        o[k] = rng.choice("MRSWYKN")
            pairs_txt.append(("".join(f), "".join(r)))
    s0 = seqs[1]
    pairs_txt.append((s0[0:20], revcomp(s0[100:120])))                                   # 5' end of the sequence
    pairs_txt.append((s0[len(s0) - 150:len(s0) - 130], revcomp(s0[len(s0) - 21:])))      # 3' end
    pairs_txt.append((s0[3:22], revcomp(s0[len(s0) - 25:len(s0) - 2])))                  # amplicon too long
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in pairs_txt]
    weights = [1.0 + 0.37 * (i % 5) for i in range(len(seqs))]
    return seqs, weights, pairs


def check(dev, oracle, seqs, weights, pairs, inactive=(), splits=(), **opts):
    o = dict(target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200, use_taq_mama=0,
             pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0, min_primer=18, optimize_5=0, optimize_3=0)
    o.update(opts)
    so = oracle.session(**o)
    for s, w in zip(seqs, weights):
        so.add_target(s, w)
    d = api.Screener(0, pack_max_degen=o["pack_max_degen"], pack_min_gc=o["pack_min_gc"], pack_max_gc=o["pack_max_gc"]) \
        if (o["pack_max_degen"], o["pack_min_gc"], o["pack_max_gc"]) != (256, 0.0, 1.0) else dev
    d.load_texts(seqs, weights)
    active = np.ones(len(seqs), np.uint8)
    for i in inactive:
        so.set_active(i, False)
        active[i] = 0
    d.set_active(active)
    for (i, pos) in splits:
        so.split(i, pos)
        d.split(i, pos)
    thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
    n_o = so.select(pairs)
    n_d = d.select_words(pairs, thr, o["min_primer"], o["optimize_5"], o["optimize_3"])
    assert d.entries() == so.db_entries()
    assert n_d == n_o
    bits, fr, rf, _ = d.amplify(pairs, o["target_threshold"], o["target_threshold"], o["amp_min"], o["amp_max"],
                                o["use_taq_mama"])
    cov = d.compute_coverage(pairs, o["target_threshold"], o["search_multiplier"], o["amp_min"], o["amp_max"],
                             o["use_taq_mama"])
    n_set = 0
    for k, p in enumerate(pairs):
        ob, oo = so.target_match(p, orient=True)
        assert (bits[k] == ob.astype(bool)).all(), k
        assert (fr[k] == ((oo & 1) != 0)).all(), k
        assert (rf[k] == ((oo & 2) != 0)).all(), k
        assert cov[k] == np.float32(so.target_coverage(p)), k
        n_set += int(ob.sum())
    if d is not dev:
        d.close()
    return n_set


@pytest.mark.parametrize("opts", [
    dict(),
    dict(target_threshold=0.9),
    dict(target_threshold=0.8, use_taq_mama=1),
    dict(target_threshold=0.9, optimize_5=1, optimize_3=1),
    dict(target_threshold=0.85, amp_min=60, amp_max=300, use_taq_mama=1),
    dict(target_threshold=0.7, search_multiplier=0.8, min_primer=16),
    dict(target_threshold=0.9, pack_max_degen=16),
    dict(target_threshold=0.9, pack_min_gc=0.35, pack_max_gc=0.65),
])
def test_select_and_amplify_match_oracle(dev, oracle, opts):
    rng = random.Random(21)
    seqs, weights, pairs = build_case(rng, oracle)
    n_set = check(dev, oracle, seqs, weights, pairs, **opts)
    assert n_set > 0


def test_degenerate_targets_and_primers(dev, oracle):
    rng = random.Random(22)
    seqs, weights, pairs = build_case(rng, oracle, p_degen=0.02, primer_degen=3)
    assert check(dev, oracle, seqs, weights, pairs, target_threshold=0.9) > 0


def test_inactive_and_split(dev, oracle):
    rng = random.Random(23)
    seqs, weights, pairs = build_case(rng, oracle)
    splits = [(0, 300), (4, 10), (4, 580), (5, 0), (6, len(seqs[6]) - 1), (7, 31), (7, 32), (7, 33)]
    check(dev, oracle, seqs, weights, pairs, inactive=(2, 7, 9), splits=splits, target_threshold=0.9)


def test_ragged_and_empty(dev, oracle):
    rng = random.Random(24)
    # sequences shorter than a word, exactly a word, odd/even lengths, all-N
    seqs = [rand_seq(rng, n) for n in (1, 17, 18, 19, 31, 32, 33, 63, 64, 65)] + ["N" * 80, "ACGT" * 30]
    weights = [1.0] * len(seqs)
    pairs_txt = [(seqs[8][0:18], revcomp(seqs[8][40:64])), (seqs[9][1:20], revcomp(seqs[9][45:65])),
                 ("ACGTACGTACGTACGTAC", revcomp("GTACGTACGTACGTACGTA"))]
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in pairs_txt]
    check(dev, oracle, seqs, weights, pairs, target_threshold=0.9, amp_min=30, amp_max=120)
    # no pairs / amplify with zero pairs
    dev.load_texts(seqs, weights)
    assert dev.select_words([], 0.9) == 0
    b, fr, rf, cov = dev.amplify([], 1.0, 1.0)
    assert b.shape[0] == 0


def test_call_order_error(oracle):
    d = api.Screener(0)
    d.load_texts(["ACGT" * 20])
    with pytest.raises(api.PcrError):
        d.amplify([(oracle.centered_word("ACGTACGTACGTACGTAC"), oracle.centered_word("ACGTACGTACGTACGTAC"))], 1.0, 1.0)
    d.close()


def test_medium_synthetic_matches_oracle(dev, oracle):
    """A C1-shaped workload from the bench generator (100 x 1 kb, 5 pairs) plus a denser pair set."""
    wl = synth.workload("C1")
    so = oracle.session(target_threshold=0.9)
    for i in range(wl["T"]):
        nb = (wl["L"] + 1) // 2
        o = int(wl["byte_offsets"][i])
        so.add_target_packed(wl["packed"][o:o + nb], wl["L"])
    pairs = synth.make_pairs(wl["packed"], wl["byte_offsets"], wl["lengths"], 24, 77)
    dev.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
    thr = float(np.float32(0.9) * np.float32(0.9))
    assert dev.select_words(pairs, thr) == so.select(pairs)
    assert dev.entries() == so.db_entries()
    bits = dev.find_target_match(pairs, 0.9)
    cov = dev.compute_coverage(pairs, 0.9, 0.9)
    tot = 0
    for k, p in enumerate(pairs):
        ob = so.target_match(p).astype(bool)
        assert (bits[k] == ob).all()
        assert cov[k] == np.float32(so.target_coverage(p))
        tot += int(ob.sum())
    assert tot > 24
