"""Pins the CPU restatement (oracle/pcr_oracle.cpp) against the REAL reference, compiled from
/root/reference by oracle/Makefile behind oracle/ref_harness.cpp.  Runs only where that build
exists (this container); the committed goldens in tests/golden/ carry the same pins to the GPU box.
"""
import random

import numpy as np
import pytest

from testdata import rand_seq, family_targets, sample_pair, mutate, revcomp

pytestmark = pytest.mark.usefixtures("reference")


def rand_word(rng, lib, lo=1, hi=32, p_degen=0.2):
    n = rng.randint(lo, hi)
    return lib.word(rand_seq(rng, n, p_degen=p_degen))


def test_word_primitives(oracle, reference):
    rng = random.Random(1)
    for it in range(3000):
        s = rand_seq(rng, rng.randint(1, 32), p_degen=0.3)
        a_o, a_r = oracle.word(s), reference.word(s)
        assert a_o == a_r
        # push the word around so that it sits at arbitrary slots
        for _ in range(rng.randint(0, 8)):
            if rng.random() < 0.5:
                a_o, a_r = oracle.word_shift_right(a_o), reference.word_shift_right(a_r)
            else:
                a_o, a_r = oracle.word_shift_left(a_o), reference.word_shift_left(a_r)
        assert a_o == a_r
        b = reference.word(rand_seq(rng, rng.randint(1, 32), p_degen=0.3))
        assert oracle.word_and(a_o, b) == reference.word_and(a_r, b)
        assert oracle.word_size(a_o) == reference.word_size(a_r)
        assert oracle.word_start(a_o) == reference.word_start(a_r)
        assert oracle.word_stop(a_o) == reference.word_stop(a_r)
        assert oracle.word_degeneracy(a_o) == reference.word_degeneracy(a_r)
        if oracle.word_size(a_o) > 0:
            assert oracle.word_center(a_o) == reference.word_center(a_r)
            assert oracle.word_complement(a_o) == reference.word_complement(a_r)


def test_word_expansion_order(oracle, reference):
    rng = random.Random(2)
    for it in range(300):
        s = rand_seq(rng, rng.randint(4, 25), p_degen=0.15)
        w = reference.centered_word(s)
        if reference.word_degeneracy(w) > 512:
            continue
        assert oracle.word_expand(w) == reference.word_expand(w)


def test_taq_mama_table(oracle, reference):
    for p1 in range(16):
        for p2 in range(16):
            for t1 in range(16):
                for t2 in range(16):
                    assert oracle.taq_mama(p1, p2, t1, t2) == reference.taq_mama(p1, p2, t1, t2)


PACK_CASES = [
    # (length, p_degen, p_n, eos positions, min_len, degen_thr, min_gc, max_gc)
    (23, 0.0, 0.0, [], 18, 256, 0.0, 1.0),
    (31, 0.0, 0.0, [], 18, 256, 0.0, 1.0),
    (32, 0.0, 0.0, [], 18, 256, 0.0, 1.0),
    (33, 0.0, 0.0, [], 18, 256, 0.0, 1.0),
    (34, 0.0, 0.0, [], 18, 256, 0.0, 1.0),
    (35, 0.0, 0.0, [], 18, 256, 0.0, 1.0),
    (77, 0.0, 0.0, [], 18, 256, 0.0, 1.0),
    (200, 0.05, 0.02, [], 18, 256, 0.0, 1.0),
    (201, 0.05, 0.02, [], 16, 256, 0.0, 1.0),
    (300, 0.0, 0.0, [100], 18, 256, 0.0, 1.0),
    (301, 0.0, 0.0, [100, 101, 102, 103], 18, 256, 0.0, 1.0),
    (300, 0.0, 0.0, [5, 150, 170, 299], 18, 256, 0.0, 1.0),
    (300, 0.0, 0.0, [0, 1, 40], 16, 256, 0.0, 1.0),
    (257, 0.1, 0.05, [64, 65, 128], 18, 16, 0.0, 1.0),
    (400, 0.0, 0.0, [], 18, 256, 0.3, 0.7),
    (401, 0.02, 0.0, [200], 18, 256, 0.4, 0.6),
    (150, 0.0, 0.3, [], 18, 256, 0.0, 1.0),
    (90, 0.0, 0.0, [44, 45], 10, 256, 0.0, 1.0),
]


@pytest.mark.parametrize("ci", range(len(PACK_CASES)))
def test_pack(oracle, reference, ci):
    L, pd, pn, eos, min_len, degen_thr, min_gc, max_gc = PACK_CASES[ci]
    rng = random.Random(1000 + ci)
    for rep in range(3):
        s = list(rand_seq(rng, L, p_degen=pd, p_n=pn))
        for e in eos:
            s[e] = "-"
        s = "".join(s)
        a = oracle.pack(s, 7, degen_thr, min_gc, max_gc, min_len)
        b = reference.pack(s, 7, degen_thr, min_gc, max_gc, min_len)
        assert a == b


def _sessions(oracle, reference, seqs, weights=None, **opts):
    so, sr = oracle.session(**opts), reference.session(**opts)
    for i, s in enumerate(seqs):
        w = 1.0 if weights is None else weights[i]
        so.add_target(s, w)
        sr.add_target(s, w)
    return so, sr


SELECT_CASES = [
    dict(),
    dict(target_threshold=0.9, search_multiplier=0.9),
    dict(target_threshold=0.8, search_multiplier=0.9, use_taq_mama=1),
    dict(target_threshold=0.9, optimize_5=1, optimize_3=1),
    dict(target_threshold=0.85, amp_min=60, amp_max=300, use_taq_mama=1),
]


@pytest.mark.parametrize("opts", SELECT_CASES)
def test_select_and_amplify(oracle, reference, opts):
    rng = random.Random(11 + len(opts))
    seqs = family_targets(rng, 3, 6, 700, div=0.04)
    # a few ragged ones: short, with N runs, with EOS splits
    seqs.append(rand_seq(rng, 40))
    seqs.append(rand_seq(rng, 333, p_degen=0.03, p_n=0.02))
    s = list(seqs[0]); s[350] = "-"; seqs.append("".join(s))
    weights = [1.0 + 0.37 * (i % 5) for i in range(len(seqs))]
    pairs_txt = []
    while len(pairs_txt) < 12:
        p = sample_pair(rng, rng.choice(seqs[:18]))
        if p:
            pairs_txt.append(p)
    # primers that sit at the very ends of a sequence (partial-word territory)
    s0 = seqs[1]
    pairs_txt.append((s0[0:20], __import__("testdata").revcomp(s0[100:120])))
    pairs_txt.append((s0[len(s0) - 150:len(s0) - 130], __import__("testdata").revcomp(s0[len(s0) - 21:])))
    pairs = [(reference.centered_word(f), reference.centered_word(r)) for f, r in pairs_txt]
    so, sr = _sessions(oracle, reference, seqs, weights, **opts)
    assert so.select(pairs) == sr.select(pairs)
    assert so.db_entries() == sr.db_entries()
    for p in pairs:
        assert (so.target_match(p) == sr.target_match(p)).all()
        assert so.target_coverage(p) == sr.target_coverage(p)


def test_inactive_and_split(oracle, reference):
    rng = random.Random(5)
    seqs = family_targets(rng, 2, 5, 600, div=0.03)
    pairs_txt = [sample_pair(rng, seqs[i % len(seqs)]) for i in range(8)]
    pairs = [(reference.centered_word(f), reference.centered_word(r)) for f, r in pairs_txt]
    so, sr = _sessions(oracle, reference, seqs, target_threshold=0.9)
    for s in (so, sr):
        s.set_active(2, False)
        s.set_active(7, False)
        s.split(0, 300)
        s.split(4, 10)
        s.split(4, 580)
    assert so.select(pairs) == sr.select(pairs)
    assert so.db_entries() == sr.db_entries()
    for p in pairs:
        assert (so.target_match(p) == sr.target_match(p)).all()
        assert so.target_coverage(p) == sr.target_coverage(p)


# ------------------------------------------------------------------------------------ Smith-Waterman
def test_sw_lanes(oracle, reference):
    """8-lane SeqOverlap calls (as background_match.cpp drives them) vs the scalar restatement."""
    rng = random.Random(31)
    nz = 0
    for it in range(400):
        qs, ts = [], []
        for lane in range(8):
            qtxt = rand_seq(rng, rng.randint(8, 32), p_degen=0.05)
            mode = rng.random()
            if mode < 0.5:       # a noisy copy of the query embedded in the target
                core = mutate(rng, qtxt, 0.12)
                if rng.random() < 0.4 and len(core) > 6:   # indel
                    k = rng.randrange(2, len(core) - 2)
                    core = core[:k] + (rand_seq(rng, 1) if rng.random() < 0.5 else "") + core[k + (rng.random() < 0.5):]
                core = core[:32]
                left = rng.randint(0, 32 - len(core))
                ttxt = (rand_seq(rng, left) + core + rand_seq(rng, 32))[:rng.randint(min(32, left + len(core)), 32)]
            elif mode < 0.6:
                qtxt = "A" * rng.randint(10, 20)
                ttxt = "A" * rng.randint(10, 32)
            else:
                ttxt = rand_seq(rng, rng.randint(8, 32), p_degen=0.05)
            q = reference.word(qtxt)
            t = reference.word(ttxt)
            for _ in range(rng.randint(0, 32 - len(qtxt))):
                q = reference.word_shift_right(q)
            for _ in range(rng.randint(0, 32 - len(ttxt))):
                t = reference.word_shift_right(t)
            qs.append(q); ts.append(t)
        want = reference.sw_align_words8(qs, ts)
        for lane in range(8):
            got = oracle.sw_align_words(qs[lane], ts[lane])
            assert got.score == want[lane][0]
            if got.score > 0:
                assert got.tup() == want[lane]
                nz += 1
    assert nz > 1000


def test_sw_known_answers(oracle):
    """samples captured from the compiled reference."""
    from pcramp_amd import words as W
    q = W.codes_from_text("ACGTACGTACGTACGTAC")
    r = oracle.sw_align_codes(q, W.codes_from_text("TTTTACGTACGTACGTACGTACTTTTTTTTTT"))
    assert (r.score, r.q_start, r.q_stop, r.t_start, r.t_stop) == (36, 0, 17, 4, 21)
    r = oracle.sw_align_codes(W.codes_from_text("A" * 18), W.codes_from_text("A" * 32))
    assert (r.score, r.t_start, r.t_stop) == (36, 14, 31)


# find_background_match multiplies a same-strand score with an opposite-strand one (lanes 0x3, 1x2,
# background_match.cpp:82-83), so realistic thresholds (0.8) rarely fire; lower ones exercise the bit.
BG_CASES = [dict(bg_threshold=0.8, bg_multiplier=0.9, use_taq_mama=0), dict(bg_threshold=0.45, bg_multiplier=0.9, use_taq_mama=1),
            dict(bg_threshold=0.4, bg_multiplier=0.8, use_taq_mama=0, amp_max=400), dict(bg_threshold=0.35, bg_multiplier=1.0, use_taq_mama=1)]


@pytest.mark.parametrize("kw", BG_CASES)
def test_background_match(oracle, reference, kw):
    rng = random.Random(41)
    roots = family_targets(rng, 3, 1, 700, div=0.0)
    seqs = []
    for r in roots:
        for _ in range(5):
            seqs.append(mutate(rng, r, 0.05))      # backgrounds: diverged relatives
    pairs_txt = []
    while len(pairs_txt) < 20:
        p = sample_pair(rng, rng.choice(roots))
        if p:
            pairs_txt.append(p)
    pairs = [(reference.centered_word(f), reference.centered_word(r)) for f, r in pairs_txt]
    thr = np.float32(kw["bg_threshold"]) * np.float32(kw["bg_multiplier"])
    so, sr = _sessions(oracle, reference, seqs)
    min_len = int(18 * 0.9)
    assert so.select(pairs, threshold=thr, min_len_override=min_len) == sr.select(pairs, threshold=thr, min_len_override=min_len)
    compared = hits = 0
    for p in pairs:
        rb, n_amp = sr.background_match(p, **kw)
        ob, _ = so.background_match(p, emulate_index_bug=1, **kw)
        if rb is None:
            continue        # the reference's odd-count out-of-bounds case (background_match.cpp:122)
        compared += 1
        hits += int(ob.sum())
        assert (ob == rb).all()
    assert compared >= 5 and (hits > 0 or kw["bg_threshold"] > 0.5)


@pytest.mark.parametrize("taq", [0, 1])
def test_multiplex_match(oracle, reference, taq):
    rng = random.Random(43)
    base = rand_seq(rng, 400)
    f, r = base[50:70], revcomp(base[180:202])
    seqs = [base[40:220], mutate(rng, base[40:220], 0.1), rand_seq(rng, 150), base[60:210], rand_seq(rng, 33),
            mutate(rng, base[30:230], 0.2), revcomp(base[40:220])]
    pair = (reference.centered_word(f), reference.centered_word(r))
    so, sr = _sessions(oracle, reference, seqs)
    for thr in (0.6, 0.8, 0.95):
        a = so.multiplex_match(pair, thr, taq)
        b = sr.multiplex_match(pair, thr, taq)
        assert (a == b).all()
    assert so.multiplex_match(pair, 0.8, taq).sum() >= 2


# ------------------------------------------------------------------------------------ thermodynamics
def test_thermo_known_values(oracle, reference):
    """SURVEY.md section 8c samples (salt 0.05, strand 9e-7)."""
    r = reference.thermo_full("AGAAGGCTCGCCAAAATAAACG", 0.05, 9e-7)
    assert abs(r[0] - 59.139496) < 1e-4 and abs(r[1] + 172.899994) < 1e-4
    o = oracle.thermo_full("AGAAGGCTCGCCAAAATAAACG", 0.05, 9e-7)
    assert (o == r).all()
    r = reference.thermo_full("GCGCGCAAAAGCGCGC", 0.05, 9e-7)
    assert abs(r[4] - 67.163818) < 1e-3 and abs(r[7] - 40.364777) < 1e-3
    assert (oracle.thermo_full("GCGCGCAAAAGCGCGC", 0.05, 9e-7) == r).all()


def hairpin_prone(rng, n):
    stem = rand_seq(rng, rng.randint(4, 8))
    loop = rand_seq(rng, rng.randint(3, 7))
    s = rand_seq(rng, rng.randint(0, 4)) + stem + loop + mutate(rng, revcomp(stem), 0.1) + rand_seq(rng, 8)
    return s[:n] if len(s) >= 12 else s + rand_seq(rng, 12)


def test_thermo_random_oligos(oracle, reference):
    rng = random.Random(51)
    worst = 0.0
    for it in range(3000):
        n = rng.randint(12, 32)
        mode = rng.random()
        if mode < 0.6:
            s = rand_seq(rng, n)
        elif mode < 0.8:
            s = hairpin_prone(rng, n)
        else:       # low complexity / self-complementary
            unit = rand_seq(rng, rng.randint(1, 4))
            s = (unit * 32)[:n]
        salt = rng.choice([0.05, 0.05, 0.01, 0.2, 1.0])
        strand = rng.choice([9e-7, 9e-7, 4.5e-7, 1e-6, 2e-8])
        r = reference.thermo_full(s, salt, strand)
        o = oracle.thermo_full(s, salt, strand)
        assert (o == r).all(), (s, salt, strand, o, r)


def test_heterodimer_random(oracle, reference):
    rng = random.Random(52)
    for it in range(1500):
        a = rand_seq(rng, rng.randint(12, 30))
        mode = rng.random()
        if mode < 0.4:
            b = rand_seq(rng, rng.randint(12, 30))
        elif mode < 0.8:     # partially complementary: dimers with mismatches, bulges, dangling ends
            core = mutate(rng, revcomp(a[rng.randint(0, 5):rng.randint(len(a) - 5, len(a))]), 0.15)
            if rng.random() < 0.4 and len(core) > 6:
                k = rng.randrange(2, len(core) - 2)
                core = core[:k] + rand_seq(rng, rng.randint(0, 2)) + core[k + rng.randint(0, 2):]
            b = (rand_seq(rng, rng.randint(0, 4)) + core + rand_seq(rng, rng.randint(0, 4)))[:32]
        else:
            b = a
        sa, sb = rng.choice([(9e-7, 9e-7), (9e-7, 4.5e-7), (1e-7, 9e-7)])
        r = reference.heterodimer_full(a, b, 0.05, sa, sb)
        o = oracle.heterodimer_full(a, b, 0.05, sa, sb)
        assert (o == r).all(), (a, b, o, r)


def test_is_valid_and_dimer_filters(oracle, reference):
    rng = random.Random(53)
    n_pass = 0
    for it in range(300):
        s = rand_seq(rng, rng.randint(18, 25), p_degen=0.08 if it % 3 == 0 else 0.0)
        w = reference.centered_word(s)
        if reference.word_degeneracy(w) > 16:
            continue
        kw = dict(tm_min=rng.choice([45.0, 50.0]), tm_max=rng.choice([65.0, 75.0]), max_hairpin=rng.choice([30.0, 40.0]),
                  max_dimer=rng.choice([30.0, 40.0]), check_homo_dimer=bool(it & 1))
        a, b = oracle.is_valid(w, **kw), reference.is_valid(w, **kw)
        assert a == b, (s, kw)
        n_pass += a
    assert 10 < n_pass < 290
    for it in range(120):
        f = reference.centered_word(rand_seq(rng, rng.randint(18, 25), p_degen=0.05))
        r = reference.centered_word(rand_seq(rng, rng.randint(18, 25), p_degen=0.05))
        if reference.word_degeneracy(f) * reference.word_degeneracy(r) > 64:
            continue
        assert oracle.max_dimer_tm((f, r)) == reference.max_dimer_tm((f, r))
        f2 = reference.centered_word(rand_seq(rng, rng.randint(18, 25)))
        r2 = reference.centered_word(revcomp(rand_seq(rng, 6) + "ACGTTGCAAT" + rand_seq(rng, 5)))
        for md in (10.0, 25.0, 40.0):
            assert oracle.multiplex_compatible((f, r), (f2, r2), max_dimer=md) == reference.multiplex_compatible((f, r), (f2, r2), max_dimer=md)


@pytest.mark.parametrize("opts", [dict(), dict(target_threshold=0.9), dict(target_threshold=0.85, use_taq_mama=1)])
def test_move_coverage(oracle, reference, opts):
    """The coverage of every single-edit variant of an oligo, evaluated as optimize_pcr.cpp does (base pair's
    candidate amplicons, identity table of the edited oligo recomputed): oracle == compiled reference."""
    from pcramp_amd import words as W
    from testdata import move_variants
    rng = random.Random(77 + len(opts))
    seqs = family_targets(rng, 3, 8, 700, div=0.05)
    weights = [1.0 + 0.21 * (i % 7) for i in range(len(seqs))]
    pairs_txt = []
    while len(pairs_txt) < 6:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            pairs_txt.append(p)
    pairs = [(reference.centered_word(f), reference.centered_word(r)) for f, r in pairs_txt]
    so, sr = _sessions(oracle, reference, seqs, weights, **opts)
    assert so.select(pairs) == sr.select(pairs)
    n_checked = n_nonzero = 0
    for p in pairs:
        for side in (0, 1):
            var = []
            for kind in ("inc", "trim5", "trim3", "grow5", "grow3"):
                var += move_variants(W, p[side], kind)
            # second-order edits so that 'dec' has something to remove
            var += [v2 for v in var[:6] for v2 in move_variants(W, v, "dec")]
            co = so.move_coverage(p, side, var)
            cr = sr.move_coverage(p, side, var)
            assert np.array_equal(co, cr)
            n_checked += len(var)
            n_nonzero += int(np.count_nonzero(co))
    assert n_checked > 500 and n_nonzero > 50


@pytest.mark.parametrize("case", [dict(), dict(degen=16), dict(target_threshold=0.9, degen=4, use_taq_mama=1),
                                  dict(degen=8, tm_min=-100.0, tm_max=200.0, max_hairpin=500.0)])
def test_optimization_move(oracle, reference, case):
    """Every local-search move of optimize_pcr.cpp for both oligos, run by the reference's own
    optimization_move() and by the oracle: returned trial word and Score identical (targets and backgrounds,
    non-multiplex)."""
    from oracle_lib import optimization_move
    from testdata import mutate
    case = dict(case)
    sess = {k: case.pop(k) for k in ("target_threshold", "use_taq_mama") if k in case}
    rng = random.Random(313 + len(case))
    seqs = family_targets(rng, 3, 8, 600, div=0.06)
    bgs = [mutate(rng, s, 0.12) for s in seqs[::4]] + [rand_seq(rng, 500) for _ in range(3)]
    pairs_txt = []
    while len(pairs_txt) < 6:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            pairs_txt.append(p)
    # some degenerate starting oligos so that -degeneracy has trials
    deg = []
    for f, r in pairs_txt[:3]:
        f = list(f); f[rng.randrange(3, len(f) - 3)] = rng.choice("RYKM"); deg.append(("".join(f), r))
    pairs = [(reference.centered_word(f), reference.centered_word(r)) for f, r in pairs_txt + deg]
    to, tr = _sessions(oracle, reference, seqs, [1.0 + 0.3 * (i % 4) for i in range(len(seqs))], **sess)
    bo, br = _sessions(oracle, reference, bgs, None, **sess)
    assert to.select(pairs) == tr.select(pairs)
    bthr = float(np.float32(0.8) * np.float32(0.9))
    assert bo.select(pairs, threshold=bthr, min_len_override=16) == br.select(pairs, threshold=bthr, min_len_override=16)
    n_nonempty = 0
    for p in pairs:
        for side in (0, 1):
            for move in range(6):
                ro = optimization_move(oracle, to, bo, p, move, side, **case)
                rr = optimization_move(reference, tr, br, p, move, side, **case)
                assert ro == rr, (move, side, ro, rr)
                n_nonempty += ro[0] != (0, 0)
    assert n_nonempty > 10


@pytest.mark.parametrize("case", [dict(), dict(degen=16), dict(target_threshold=0.9, degen=4, use_taq_mama=1),
                                  dict(degen=8, tm_min=-100.0, tm_max=200.0, max_hairpin=500.0),
                                  dict(degen=64, target_threshold=0.85, tm_min=40.0, tm_max=80.0)])
def test_optimize_loop(oracle, reference, case):
    """The reference's optimize() (greedy local search, optimize.cpp:14-207) against the oracle's restatement:
    final assay and Score identical, for assays sampled from the targets and for deliberately bad ones."""
    from oracle_lib import optimize
    from testdata import mutate
    case = dict(case)
    sess = {k: case.pop(k) for k in ("target_threshold", "use_taq_mama") if k in case}
    rng = random.Random(4242 + len(case))
    seqs = family_targets(rng, 3, 8, 600, div=0.06)
    bgs = [mutate(rng, s, 0.12) for s in seqs[::4]] + [rand_seq(rng, 500) for _ in range(3)]
    pairs_txt = []
    while len(pairs_txt) < 8:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            pairs_txt.append(p)
    # primers with a few wrong bases: the search has something to repair
    for f, r in list(pairs_txt[:4]):
        pairs_txt.append((mutate(rng, f, 0.1), mutate(rng, r, 0.1)))
    pairs = [(reference.centered_word(f), reference.centered_word(r)) for f, r in pairs_txt]
    to, tr = _sessions(oracle, reference, seqs, [1.0 + 0.3 * (i % 4) for i in range(len(seqs))], optimize_5=1, optimize_3=1, **sess)
    bo, br = _sessions(oracle, reference, bgs, None, optimize_5=1, optimize_3=1, **sess)
    assert to.select(pairs) == tr.select(pairs)
    bthr = float(np.float32(0.8) * np.float32(0.9))
    assert bo.select(pairs, threshold=bthr, min_len_override=16) == br.select(pairs, threshold=bthr, min_len_override=16)
    changed = 0
    for p in pairs:
        ro = optimize(oracle, to, bo, p, **case)
        rr = optimize(reference, tr, br, p, **case)
        assert ro == rr, (p, ro, rr)
        changed += ro[0] != p
    assert changed > 0


@pytest.mark.parametrize("case", [dict(), dict(max_degen=8.0, tm_min=40.0, tm_max=80.0), dict(tm_min=57.0, tm_max=61.0, max_hairpin=30.0),
                                  dict(primer_min=20, primer_max=32, amp_min=70, amp_max=90, tm_min=55.0, tm_max=90.0),
                                  dict(primer_min=18, primer_max=22, amp_min=38, amp_max=60, tm_min=40.0, tm_max=80.0)])
def test_random_assays(oracle, reference, case):
    """The reference's PCR::random_assay (one NucCruc, one running rand_r state, as main.cpp:544-550 at one
    thread) against the oracle's restatement: same assays, same state afterwards, over many seeds."""
    from oracle_lib import random_assays, rand_r
    rng = random.Random(31 + len(case))
    if case.get("amp_min") == 38:
        seqs = [rand_seq(rng, n) for n in (40, 43, 52, 66, 90)]
    else:
        seqs = family_targets(rng, 3, 6, 800, div=0.05) + [rand_seq(rng, 301)]
        q = list(seqs[3])
        for k in range(200, 380, 7):
            q[k] = rng.choice("RYKMSWBN")
        seqs[3] = "".join(q)
    so, sr = oracle.session(), reference.session()
    for i, q in enumerate(seqs):
        so.add_target(q, 1.0, i != 2)
        sr.add_target(q, 1.0, i != 2)
    if len(seqs) > 6:
        for i, pos in ((1, 420), (6, 90), (6, 600)):
            so.split(i, pos)
            sr.split(i, pos)
    for seed in range(60):
        assert rand_r(oracle, seed * 2654435761 % 2**32) == rand_r(reference, seed * 2654435761 % 2**32)
        assert random_assays(oracle, so, seed, 8, **case) == random_assays(reference, sr, seed, 8, **case), seed


def test_random_assay_errors(oracle, reference):
    """Where the reference throws, so does the oracle (same text)."""
    from oracle_lib import random_assays
    for lib in (oracle, reference):
        s = lib.session()
        s.add_target("ACGT" * 10, 1.0, True)
        with pytest.raises(RuntimeError, match="sequence length is too small"):
            random_assays(lib, s, 1, 1)
        s = lib.session()
        s.add_target("ACGT" * 100, 1.0, False)
        with pytest.raises(RuntimeError, match="No active sequences"):
            random_assays(lib, s, 1, 1)
        s = lib.session()
        s.add_target("A" * 400, 1.0, True)                             # nothing passes the Tm filter
        with pytest.raises(RuntimeError, match="Unable to generate"):
            random_assays(lib, s, 1, 1)


def test_max_overlap_and_oligo_overlap(oracle, reference):
    """Word::max_overlap (word.h:38-91) and PCR::compute_oligo_overlap (pcr_assay.cpp:736-754)."""
    rng = random.Random(99)
    words = []
    for _ in range(120):
        w = reference.word(rand_seq(rng, rng.randint(12, 32), p_degen=0.2))
        for _ in range(rng.randint(0, 6)):
            w = reference.word_shift_right(w)
        words.append(w)
    stem = rand_seq(rng, 32)
    words += [reference.centered_word(stem[k:k + 20]) for k in range(10)] + [reference.centered_word(stem[:20])]
    for a in words[::3]:
        for b in words:
            assert oracle.max_overlap(a, b) == reference.max_overlap(a, b)
    for _ in range(200):
        assay = (rng.choice(words), rng.choice(words))
        pool = [(rng.choice(words), rng.choice(words)) for _ in range(rng.randint(0, 8))]
        assert oracle.oligo_overlap(assay, pool) == reference.oligo_overlap(assay, pool)


@pytest.mark.parametrize("taq", [0, 1])
def test_multiplex_background_coverage(oracle, reference, taq):
    """collect_multiplex_background_candidates + update_identity + compute_multiplex_background_coverage (the
    reference's own, over a DB packed like main.cpp:989-1001) against the oracle: key count and the coverage of
    every move variant of both oligos."""
    from testdata import multiplex_case, move_variants
    from pcramp_amd import words as W
    rng = random.Random(640 + taq)
    amps, pairs = multiplex_case(rng, W, reference)
    so, sr = oracle.session(min_primer=18), reference.session(min_primer=18)
    for a in amps:
        so.add_target(a, 1.0)
        sr.add_target(a, 1.0)
    nonzero = 0
    for p in pairs:
        for side in (0, 1):
            var = [p[side]]
            for kind in ("inc", "dec", "trim5", "trim3", "grow5", "grow3"):
                var += move_variants(W, p[side], kind)
            for thr in (0.8, 0.65):
                co, ko = so.multiplex_coverage(p, side, var, thr, taq)
                cr, kr = sr.multiplex_coverage(p, side, var, thr, taq)
                assert ko == kr and ko > 0
                assert np.array_equal(co, cr), (p, side, thr)
                nonzero += int(np.count_nonzero(co))
    assert nonzero > 100


@pytest.mark.parametrize("case", [dict(), dict(degen=8), dict(degen=16, target_threshold=0.9, use_taq_mama=1),
                                  dict(degen=4, tm_min=-100.0, tm_max=200.0, max_hairpin=500.0)])
def test_optimize_loop_multiplex(oracle, reference, case):
    """optimize() with opt.use_multiplex (multiplex background keys, oligo reuse term, the '< 0' coverage bound,
    increase_degeneracy's carried overlap): the reference's own loop against the oracle's restatement."""
    from oracle_lib import optimize_multiplex
    from testdata import multiplex_design_case
    case = dict(case)
    sess = {k: case.pop(k) for k in ("target_threshold", "use_taq_mama") if k in case}
    rng = random.Random(77 + len(case) + 10 * len(sess))
    seqs, bgs, amps, pool, cands = multiplex_design_case(rng, reference)
    to, tr = _sessions(oracle, reference, seqs, [1.0 + 0.25 * (i % 3) for i in range(len(seqs))], optimize_5=1, optimize_3=1, **sess)
    bo, br = _sessions(oracle, reference, bgs, None, optimize_5=1, optimize_3=1, **sess)
    ao, ar = _sessions(oracle, reference, amps, None, **sess)
    allp = cands + pool
    assert to.select(allp) == tr.select(allp)
    bthr = float(np.float32(0.8) * np.float32(0.9))
    assert bo.select(allp, threshold=bthr, min_len_override=16) == br.select(allp, threshold=bthr, min_len_override=16)
    changed = with_overlap = 0
    for p in cands:
        for pl in (pool, []):
            ro = optimize_multiplex(oracle, to, bo, ao, pl, p, **case)
            rr = optimize_multiplex(reference, tr, br, ar, pl, p, **case)
            assert ro == rr, (p, ro, rr)
            changed += ro[0] != p
            with_overlap += ro[1][2] > 0
    assert changed > 0 and with_overlap > 0


@pytest.mark.parametrize("case", [dict(degen=8), dict(degen=16, target_threshold=0.9, use_taq_mama=1)])
def test_optimization_move_multiplex(oracle, reference, case):
    """Every move x both oligos with opt.use_multiplex: trial word, Score and base Score (reference's own
    optimization_move() against the oracle)."""
    from oracle_lib import optimization_move_multiplex
    from testdata import multiplex_design_case
    case = dict(case)
    sess = {k: case.pop(k) for k in ("target_threshold", "use_taq_mama") if k in case}
    rng = random.Random(515 + len(sess))
    seqs, bgs, amps, pool, cands = multiplex_design_case(rng, reference)
    to, tr = _sessions(oracle, reference, seqs, None, optimize_5=1, optimize_3=1, **sess)
    bo, br = _sessions(oracle, reference, bgs, None, optimize_5=1, optimize_3=1, **sess)
    ao, ar = _sessions(oracle, reference, amps, None, **sess)
    allp = cands + pool
    to.select(allp); tr.select(allp)
    bthr = float(np.float32(0.8) * np.float32(0.9))
    bo.select(allp, threshold=bthr, min_len_override=16); br.select(allp, threshold=bthr, min_len_override=16)
    found = 0
    for p in cands:
        for side in (0, 1):
            for move in range(6):
                ro = optimization_move_multiplex(oracle, to, bo, ao, pool, p, move, side, **case)
                rr = optimization_move_multiplex(reference, tr, br, ar, pool, p, move, side, **case)
                assert ro == rr, (p, side, move, ro, rr)
                found += ro[0] != (0, 0)
    assert found > 20


@pytest.mark.parametrize("opts", [dict(), dict(target_threshold=0.9), dict(target_threshold=0.85, amp_min=40, amp_max=400)])
def test_collect_unique_amplicons(oracle, reference, opts):
    """PCR::collect_unique_amplicons (pcr_assay.cpp:756-813): AmpliconBounds in discovery order and the unique
    amplicon stretches, with EOS splits, IUPAC bases and inactive sequences in the targets."""
    o = dict(target_threshold=1.0, amp_min=80, amp_max=200)
    o.update(opts)
    rng = random.Random(2718 + len(opts))
    seqs = family_targets(rng, 3, 7, 700, div=0.05)
    q = list(seqs[2])
    for k in range(100, 600, 23):
        q[k] = rng.choice("RYKMSWN")
    seqs[2] = "".join(q)
    pairs_txt = []
    while len(pairs_txt) < 8:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            pairs_txt.append(p)
    pairs = [(reference.centered_word(f), reference.centered_word(r)) for f, r in pairs_txt]
    so, sr = _sessions(oracle, reference, seqs, None, **o)
    for i, pos in ((1, 350), (5, 120), (9, 500)):
        so.split(i, pos); sr.split(i, pos)
    so.set_active(4, False); sr.set_active(4, False)
    assert so.select(pairs) == sr.select(pairs)
    n_b = n_a = 0
    for p in pairs:
        bo, ao = so.collect_amplicons(p, o["target_threshold"], o["amp_min"], o["amp_max"])
        br, ar = sr.collect_amplicons(p, o["target_threshold"], o["amp_min"], o["amp_max"])
        assert bo == br
        assert ao == ar
        n_b += len(bo); n_a += len(ao)
    assert n_b >= 5 and n_a >= 3, (n_b, n_a)


@pytest.mark.parametrize("case", [dict(degen=16), dict(degen=64, target_threshold=0.9, tm_min=40.0, tm_max=80.0, max_hairpin=50.0),
                                  dict(degen=8, target_threshold=0.85, use_taq_mama=1, tm_min=40.0, tm_max=80.0),
                                  dict(degen=64, max_dimer=20.0, tm_min=-100.0, tm_max=200.0, max_hairpin=500.0),
                                  dict(degen=64, max_dimer=12.0, tm_min=-100.0, tm_max=200.0, max_hairpin=500.0, seed=9103)])
def test_make_degenerate(oracle, reference, case):
    """The reference's make_degenerate (optimize.cpp:356-398 -> PCR::maximize_degeneracy, pcr_assay.cpp:111-230: the top-down
    start of the local search) against the oracle's restatement: resulting assay and return value.  Families 8 % apart so
    that the unions with the matched keys add bases; the last two cases set max_dimer low enough for the greedy heterodimer
    reduction (and its failure exit) to run."""
    from oracle_lib import make_degenerate
    case = dict(case)
    sess = {k: case.pop(k) for k in ("target_threshold", "use_taq_mama") if k in case}
    rng = random.Random(case.pop("seed", 9100 + len(case) + 7 * len(sess)))
    seqs = family_targets(rng, 3, 12, 500, div=0.08)
    pairs_txt = []
    while len(pairs_txt) < 14:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            pairs_txt.append(p)
    pairs = [(reference.centered_word(f), reference.centered_word(r)) for f, r in pairs_txt]
    to, tr = _sessions(oracle, reference, seqs, [1.0 + 0.3 * (i % 4) for i in range(len(seqs))], **sess)
    assert to.select(pairs) == tr.select(pairs)
    changed = reduced = 0
    for p in pairs:
        ro = make_degenerate(oracle, to, p, **case)
        rr = make_degenerate(reference, tr, p, **case)
        assert ro == rr, (p, ro, rr)
        changed += ro[0] != p
        if "max_dimer" in case:                     # did the heterodimer reduction change the outcome?
            loose = dict(case, max_dimer=40.0)
            reduced += make_degenerate(oracle, to, p, **loose) != ro
    assert changed >= 3
    if "max_dimer" in case:
        assert reduced >= 1
