"""Smith-Waterman family on the GPU against the CPU oracle: single alignments (SeqOverlap lanes),
find_background_match and find_multiplex_background_match.  Bit-exact scores, coordinates, bits."""
import random

import numpy as np
import pytest

from pcramp_amd import api, words as W
from testdata import rand_seq, family_targets, sample_pair, mutate, revcomp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    s = api.Screener(0)
    yield s
    s.close()


def test_sw_lanes_match_oracle(dev, oracle):
    rng = random.Random(31)
    qs, ts = [], []
    for it in range(3000):
        qtxt = rand_seq(rng, rng.randint(1, 32), p_degen=0.05)
        mode = rng.random()
        if mode < 0.5:
            core = mutate(rng, qtxt, 0.12)
            if rng.random() < 0.4 and len(core) > 6:
                k = rng.randrange(2, len(core) - 2)
                core = core[:k] + (rand_seq(rng, 1) if rng.random() < 0.5 else "") + core[k + (rng.random() < 0.5):]
            core = core[:32]
            left = rng.randint(0, 32 - len(core))
            ttxt = (rand_seq(rng, left) + core + rand_seq(rng, 32))[:rng.randint(min(32, left + len(core)), 32)]
        elif mode < 0.6:
            qtxt = "A" * rng.randint(10, 20)
            ttxt = "A" * rng.randint(10, 32)
        elif mode < 0.65:
            qtxt, ttxt = "A" * rng.randint(1, 10), "C" * rng.randint(1, 10)      # no match at all
        else:
            ttxt = rand_seq(rng, rng.randint(1, 32), p_degen=0.05)
        q, t = oracle.word(qtxt), oracle.word(ttxt)
        for _ in range(rng.randint(0, 32 - len(qtxt))):
            q = oracle.word_shift_right(q)
        for _ in range(rng.randint(0, 32 - len(ttxt))):
            t = oracle.word_shift_right(t)
        qs.append(q); ts.append(t)
    got = dev.sw_align_words(qs, ts)
    nz = 0
    for k in range(len(qs)):
        want = oracle.sw_align_words(qs[k], ts[k])
        assert got[k][0] == want.score, k
        assert got[k][7] == want.valid, k
        if want.valid:
            assert got[k][:7] == want.tup(), k
            nz += 1
    assert nz > 2000


def test_sw_known_answers(dev, oracle):
    q = oracle.word("ACGTACGTACGTACGTAC")
    r = dev.sw_align_words([q, oracle.word("A" * 18)], [oracle.word("TTTTACGTACGTACGTACGTACTTTTTTTTTT"), oracle.word("A" * 32)])
    assert r[0][:5] == (36, 0, 17, 4, 21)          # SURVEY.md 3.4, captured from the compiled reference
    assert (r[1][0], r[1][3], r[1][4]) == (36, 14, 31)


@pytest.mark.parametrize("kw", [dict(bg_threshold=0.8, bg_multiplier=0.9, use_taq_mama=0),
                                dict(bg_threshold=0.45, bg_multiplier=0.9, use_taq_mama=1),
                                dict(bg_threshold=0.4, bg_multiplier=0.8, use_taq_mama=0, amp_max=400),
                                dict(bg_threshold=0.35, bg_multiplier=1.0, use_taq_mama=1)])
def test_background_match(dev, oracle, kw):
    rng = random.Random(41)
    roots = family_targets(rng, 3, 1, 700, div=0.0)
    seqs = [mutate(rng, r, 0.05) for r in roots for _ in range(5)]
    pairs_txt = []
    while len(pairs_txt) < 20:
        p = sample_pair(rng, rng.choice(roots))
        if p:
            pairs_txt.append(p)
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in pairs_txt]
    thr = float(np.float32(kw["bg_threshold"]) * np.float32(kw["bg_multiplier"]))
    so = oracle.session()
    for s in seqs:
        so.add_target(s)
    min_len = int(18 * 0.9)
    dev.load_texts(seqs, which=api.BACKGROUND)
    assert dev.select_words(pairs, thr, min_len, which=api.BACKGROUND) == so.select(pairs, threshold=thr, min_len_override=min_len)
    # default = the reference bit for bit (odd-indexed amplicons at or beyond the sequence count are never scored,
    # background_match.cpp:122); evaluate_all = every candidate amplicon
    bits = dev.find_background_match(pairs, kw["bg_threshold"], kw["bg_multiplier"], 0, kw.get("amp_max", 2000),
                                     kw["use_taq_mama"])
    bits_all = dev.find_background_match(pairs, kw["bg_threshold"], kw["bg_multiplier"], 0, kw.get("amp_max", 2000),
                                         kw["use_taq_mama"], evaluate_all=True)
    hits = differ = 0
    for k, p in enumerate(pairs):
        ob, _ = so.background_match(p, emulate_index_bug=1, **kw)
        oa, _ = so.background_match(p, emulate_index_bug=0, **kw)
        assert (bits[k] == ob.astype(bool)).all(), k
        assert (bits_all[k] == oa.astype(bool)).all(), k
        hits += int(ob.sum())
        differ += int((ob != oa).any())
    assert hits > 0 or kw["bg_threshold"] > 0.5
    assert differ > 0 or kw["bg_threshold"] > 0.5        # the two modes are told apart on this input


@pytest.mark.parametrize("taq", [0, 1])
def test_multiplex_match(dev, oracle, taq):
    rng = random.Random(43)
    base = rand_seq(rng, 400)
    seqs = [base[40:220], mutate(rng, base[40:220], 0.1), rand_seq(rng, 150), base[60:210], rand_seq(rng, 33),
            mutate(rng, base[30:230], 0.2), revcomp(base[40:220]), rand_seq(rng, 5), base[50:70]]
    pairs = [(oracle.centered_word(base[50:70]), oracle.centered_word(revcomp(base[180:202]))),
             (oracle.centered_word(base[100:125]), oracle.centered_word(revcomp(base[300:318])))]
    so = oracle.session()
    for s in seqs:
        so.add_target(s)
    dev.load_texts(seqs, which=api.BACKGROUND)
    tot = 0
    for thr in (0.6, 0.8, 0.95):
        bits = dev.find_multiplex_background_match(pairs, thr, taq)
        for k, p in enumerate(pairs):
            ob = so.multiplex_match(p, thr, taq)
            assert (bits[k] == ob.astype(bool)).all()
            tot += int(ob.sum())
    assert tot > 4
