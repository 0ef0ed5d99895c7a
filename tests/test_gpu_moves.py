"""pcr_move_coverage (local-search move evaluation, optimize_pcr.cpp) against the oracle: coverage floats
and per-orientation bits of every single-edit variant, bit-exact.  Run on the GPU box with `-m gpu`."""
import random

import numpy as np
import pytest

from pcramp_amd import api, words as W
from testdata import family_targets, sample_pair, move_variants, rand_seq, mutate, revcomp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("opts", [dict(), dict(target_threshold=0.9), dict(target_threshold=0.85, use_taq_mama=1)])
def test_move_coverage_matches_oracle(oracle, opts):
    o = dict(target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200, use_taq_mama=0,
             pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0, min_primer=18, optimize_5=0, optimize_3=0)
    o.update(opts)
    rng = random.Random(177 + len(opts))
    seqs = family_targets(rng, 4, 10, 900, div=0.05)
    weights = [1.0 + 0.21 * (i % 7) for i in range(len(seqs))]
    pairs_txt = []
    while len(pairs_txt) < 8:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            pairs_txt.append(p)
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in pairs_txt]
    so = oracle.session(**o)
    for s, w in zip(seqs, weights):
        so.add_target(s, w)
    so.select(pairs)
    d = api.Screener(0)
    try:
        d.load_texts(seqs, weights)
        thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
        d.select_words(pairs, thr, o["min_primer"])
        n_checked = n_nonzero = 0
        for p in pairs:
            for side in (0, 1):
                var = []
                for kind in ("inc", "trim5", "trim3", "grow5", "grow3"):
                    var += move_variants(W, p[side], kind)
                var += [v2 for v in var[:6] for v2 in move_variants(W, v, "dec")]
                co, oo = so.move_coverage(p, side, var, orient=True)
                cd, fr, rf = d.move_coverage(p, side, var, o["target_threshold"], o["search_multiplier"], o["amp_min"],
                                             o["amp_max"], bool(o["use_taq_mama"]))
                assert np.array_equal(cd, co)
                assert np.array_equal(fr, (oo & 1) != 0)
                assert np.array_equal(rf, (oo & 2) != 0)
                n_checked += len(var)
                n_nonzero += int(np.count_nonzero(co))
        assert n_checked > 500 and n_nonzero > 50
    finally:
        d.close()


def test_move_coverage_with_splits_and_inactive_sequences(oracle):
    """EOS inside the candidate amplicons (has_split, sequence.cpp:304), record padding ('-'), inactive
    sequences; wide amplicon range so that many partners are swept per site."""
    o = dict(target_threshold=0.85, search_multiplier=0.9, amp_min=40, amp_max=600, use_taq_mama=0,
             pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0, min_primer=18, optimize_5=0, optimize_3=0)
    rng = random.Random(4711)
    seqs = family_targets(rng, 3, 8, 900, div=0.06)
    pairs_txt = []
    while len(pairs_txt) < 6:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            pairs_txt.append(p)
    seqs[3] = seqs[3][:450] + "----" + seqs[3][454:]                  # multi-record padding
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in pairs_txt]
    so = oracle.session(**o)
    for s in seqs:
        so.add_target(s, 1.0)
    d = api.Screener(0)
    try:
        d.load_texts(seqs, [1.0] * len(seqs))
        active = [i % 5 != 2 for i in range(len(seqs))]
        d.set_active(active)
        for i, a in enumerate(active):
            so.set_active(i, a)
        for i in range(0, len(seqs), 3):                               # a split somewhere inside every third sequence
            pos = 150 + 29 * i
            so.split(i, pos)
            d.split(i, pos)
        thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
        so.select(pairs)
        d.select_words(pairs, thr, o["min_primer"])
        nonzero = zero = 0
        for p in pairs:
            for side in (0, 1):
                var = [p[side]]
                for kind in ("inc", "trim5", "trim3", "grow5", "grow3"):
                    var += move_variants(W, p[side], kind)
                co, oo = so.move_coverage(p, side, var, orient=True)
                cd, fr, rf = d.move_coverage(p, side, var, o["target_threshold"], o["search_multiplier"], o["amp_min"],
                                             o["amp_max"], False)
                assert np.array_equal(cd, co)
                assert np.array_equal(fr, (oo & 1) != 0)
                assert np.array_equal(rf, (oo & 2) != 0)
                nonzero += int(np.count_nonzero(co))
                zero += int(np.count_nonzero(co == 0))
        assert nonzero > 50, (nonzero, zero)
    finally:
        d.close()


def test_move_coverage_of_the_base_word_is_compute_coverage(oracle):
    """The unedited oligo as its own 'variant' gives compute_coverage of the pair."""
    rng = random.Random(5)
    seqs = family_targets(rng, 3, 8, 800, div=0.04)
    pairs_txt = []
    while len(pairs_txt) < 5:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            pairs_txt.append(p)
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in pairs_txt]
    d = api.Screener(0)
    try:
        d.load_texts(seqs, [1.0] * len(seqs))
        thr = float(np.float32(1.0) * np.float32(0.9))
        d.select_words(pairs, thr, 18)
        cov = d.compute_coverage(pairs, 1.0, 0.9, 80, 200, False)
        for k, p in enumerate(pairs):
            for side in (0, 1):
                c, _, _ = d.move_coverage(p, side, [p[side]], 1.0, 0.9, 80, 200, False)
                assert c[0] == cov[k]
    finally:
        d.close()


@pytest.mark.parametrize("case", [dict(), dict(degen=16), dict(target_threshold=0.9, degen=4, use_taq_mama=1),
                                  dict(degen=8, tm_min=-100.0, tm_max=200.0, max_hairpin=500.0)])
def test_optimization_move_matches_oracle(oracle, case):
    """The six local-search moves for both oligos: host trial generation + device is_valid + device coverage +
    Score logic (pcramp_amd.moves) return the same trial word and Score as the oracle (itself pinned to the
    reference's own optimization_move() in test_oracle_vs_reference)."""
    from oracle_lib import optimization_move as oracle_move
    from pcramp_amd import moves
    from testdata import mutate, rand_seq
    case = dict(case)
    sess = {k: case.pop(k) for k in ("target_threshold", "use_taq_mama") if k in case}
    o = dict(target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200, use_taq_mama=0,
             pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0, min_primer=18, optimize_5=0, optimize_3=0)
    o.update(sess)
    rng = random.Random(313 + len(case))
    seqs = family_targets(rng, 3, 8, 600, div=0.06)
    bgs = [mutate(rng, s, 0.12) for s in seqs[::4]] + [rand_seq(rng, 500) for _ in range(3)]
    pairs_txt = []
    while len(pairs_txt) < 6:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            pairs_txt.append(p)
    deg = []
    for f, r in pairs_txt[:3]:
        f = list(f); f[rng.randrange(3, len(f) - 3)] = rng.choice("RYKM"); deg.append(("".join(f), r))
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in pairs_txt + deg]
    tw = [1.0 + 0.3 * (i % 4) for i in range(len(seqs))]
    to, bo = oracle.session(**o), oracle.session(**o)
    for s, w in zip(seqs, tw):
        to.add_target(s, w)
    for s in bgs:
        bo.add_target(s, 1.0)
    to.select(pairs)
    bthr = float(np.float32(0.8) * np.float32(0.9))
    bo.select(pairs, threshold=bthr, min_len_override=16)
    d = api.Screener(0)
    try:
        d.load_texts(seqs, tw, which=api.TARGET)
        d.load_texts(bgs, [1.0] * len(bgs), which=api.BACKGROUND)
        thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
        d.select_words(pairs, thr, 18, which=api.TARGET)
        d.select_words(pairs, bthr, 16, which=api.BACKGROUND)
        n_nonempty = 0
        for p in pairs:
            for side in (0, 1):
                for move in range(6):
                    wo, so_, base = oracle_move(oracle, to, bo, p, move, side, **case)
                    wd, sd = moves.optimization_move(d, p, move, side, target_threshold=o["target_threshold"],
                                                     use_taq_mama=bool(o["use_taq_mama"]), **case)
                    assert wd == wo, (move, side)
                    assert tuple(float(x) for x in sd) == so_, (move, side)
                    n_nonempty += wo != (0, 0)
        assert n_nonempty > 10
    finally:
        d.close()


@pytest.mark.parametrize("case", [dict(), dict(degen=16), dict(target_threshold=0.9, degen=4, use_taq_mama=1),
                                  dict(degen=64, target_threshold=0.85, tm_min=40.0, tm_max=80.0)])
def test_optimize_loop_matches_oracle(oracle, case):
    """optimize() (the greedy local search) over the device primitives == the oracle's restatement (pinned to the
    reference's optimize() in test_oracle_vs_reference): final assay and Score."""
    from oracle_lib import optimize as oracle_optimize
    from pcramp_amd import moves
    from testdata import mutate, rand_seq
    case = dict(case)
    sess = {k: case.pop(k) for k in ("target_threshold", "use_taq_mama") if k in case}
    o = dict(target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200, use_taq_mama=0,
             pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0, min_primer=18, optimize_5=1, optimize_3=1)
    o.update(sess)
    rng = random.Random(4242 + len(case))
    seqs = family_targets(rng, 3, 8, 600, div=0.06)
    bgs = [mutate(rng, s, 0.12) for s in seqs[::4]] + [rand_seq(rng, 500) for _ in range(3)]
    pairs_txt = []
    while len(pairs_txt) < 6:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            pairs_txt.append(p)
    for f, r in list(pairs_txt[:3]):
        pairs_txt.append((mutate(rng, f, 0.1), mutate(rng, r, 0.1)))
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in pairs_txt]
    tw = [1.0 + 0.3 * (i % 4) for i in range(len(seqs))]
    to, bo = oracle.session(**o), oracle.session(**o)
    for s, w in zip(seqs, tw):
        to.add_target(s, w)
    for s in bgs:
        bo.add_target(s, 1.0)
    to.select(pairs)
    bthr = float(np.float32(0.8) * np.float32(0.9))
    bo.select(pairs, threshold=bthr, min_len_override=16)
    d = api.Screener(0)
    try:
        d.load_texts(seqs, tw, which=api.TARGET)
        d.load_texts(bgs, [1.0] * len(bgs), which=api.BACKGROUND)
        thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
        d.select_words(pairs, thr, 18, True, True, which=api.TARGET)
        d.select_words(pairs, bthr, 16, True, True, which=api.BACKGROUND)
        changed = 0
        want = []
        for p in pairs:
            po, so_ = oracle_optimize(oracle, to, bo, p, **case)
            pd, sd = moves.optimize(d, p, target_threshold=o["target_threshold"], use_taq_mama=bool(o["use_taq_mama"]), **case)
            assert pd == po
            assert tuple(float(x) for x in sd) == so_
            changed += po != p
            want.append((po, so_))
        assert changed > 0
        # the same assays as ONE batch (pcr_optimize_batch: lockstep iterations, one thermodynamics launch and one coverage
        # pass per set and iteration for all of them) -- assays finish after different numbers of iterations
        bp, bs, it = moves.optimize_batch(d, pairs * 3, target_threshold=o["target_threshold"], use_taq_mama=bool(o["use_taq_mama"]), **case)
        for k in range(len(pairs) * 3):
            assert bp[k] == want[k % len(pairs)][0] and tuple(float(x) for x in bs[k]) == want[k % len(pairs)][1], k
        assert len(set(it)) > 1
    finally:
        d.close()


@pytest.mark.parametrize("taq", [0, 1])
def test_multiplex_coverage_matches_oracle(oracle, taq):
    """The multiplex background term: unique keys of the accepted amplicons (host window model) and the
    distinct-key count per trial word, against the oracle (itself == the compiled reference)."""
    from testdata import multiplex_case
    rng = random.Random(3300 + taq)
    amps, pairs = multiplex_case(rng, W, oracle, n_amp=30)
    so = oracle.session(min_primer=18)
    for a in amps:
        so.add_target(a, 1.0)
    d = api.Screener(0)
    try:
        assert d.multiplex_coverage(pairs[0], 0, [pairs[0][0]]).tolist() == [0.0]   # no keys loaded yet
        nk = d.multiplex_load(amps, 18)
        nonzero = 0
        for p in pairs:
            for side in (0, 1):
                var = [p[side]]
                for kind in ("inc", "dec", "trim5", "trim3", "grow5", "grow3"):
                    var += move_variants(W, p[side], kind)
                for thr in (0.8, 0.65, 1.0):
                    co, ko = so.multiplex_coverage(p, side, var, thr, taq)
                    assert ko == nk
                    cd = d.multiplex_coverage(p, side, var, thr, bool(taq))
                    assert np.array_equal(cd, co), (p, side, thr)
                    nonzero += int(np.count_nonzero(co))
        assert nonzero > 200
        assert d.multiplex_load([], 18) == 0                           # an empty pool: coverage 0 (pcr_assay.cpp:306)
        assert d.multiplex_coverage(pairs[0], 0, [pairs[0][0]]).tolist() == [0.0]
    finally:
        d.close()


@pytest.mark.parametrize("case", [dict(degen=8), dict(degen=16, use_taq_mama=1)])
def test_optimization_move_multiplex_matches_oracle(oracle, case):
    """Every move x both oligos with opt.use_multiplex (multiplex background term, reuse term, '< 0' bound):
    trial word and Score identical to the oracle (itself == the reference's own optimization_move())."""
    from pcramp_amd import moves
    from oracle_lib import optimization_move_multiplex, DEFAULT_MOVE_OPTIONS
    from testdata import multiplex_design_case
    case = dict(case)
    taq = case.pop("use_taq_mama", 0)
    rng = random.Random(929 + taq)
    seqs, bgs, amps, pool, cands = multiplex_design_case(rng, oracle)
    o = dict(target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200, use_taq_mama=taq, pack_max_degen=256,
             pack_min_gc=0.0, pack_max_gc=1.0, min_primer=18, optimize_5=1, optimize_3=1)
    ts, bs, ams = oracle.session(**o), oracle.session(**o), oracle.session(use_taq_mama=taq)
    for q in seqs:
        ts.add_target(q, 1.0)
    for q in bgs:
        bs.add_target(q, 1.0)
    for q in amps:
        ams.add_target(q, 1.0)
    allp = cands + pool
    bthr = float(np.float32(0.8) * np.float32(0.9))
    ts.select(allp)
    bs.select(allp, threshold=bthr, min_len_override=16)
    d = api.Screener(0)
    try:
        d.load_texts(seqs, [1.0] * len(seqs))
        d.load_texts(bgs, [1.0] * len(bgs), which=api.BACKGROUND)
        d.multiplex_load(amps, 18)
        d.select_words(allp, float(np.float32(1.0) * np.float32(0.9)), 18, True, True)
        d.select_words(allp, bthr, 16, True, True, which=api.BACKGROUND)
        mo = dict(DEFAULT_MOVE_OPTIONS)
        mo.update(case)
        found = 0
        for p in cands:
            for side in (0, 1):
                for move in range(6):
                    want = optimization_move_multiplex(oracle, ts, bs, ams, pool, p, move, side, **case)
                    w, sc = moves.optimization_move(d, p, move, side, pool=pool, use_taq_mama=bool(taq), **mo)
                    assert w == want[0], (p, side, move)
                    assert tuple(float(x) for x in sc) == want[1], (p, side, move)
                    found += w != (0, 0)
        assert found > 20
    finally:
        d.close()


@pytest.mark.parametrize("opts", [dict(), dict(target_threshold=0.9), dict(target_threshold=0.85, amp_min=40, amp_max=400)])
def test_collect_amplicons_matches_oracle(oracle, opts):
    """pcr_collect_amplicons (PCR::collect_unique_amplicons, pcr_assay.cpp:756-813): the AmpliconBounds and the
    unique amplicon stretches of an assay, against the oracle (== the compiled reference), with EOS splits,
    IUPAC bases and an inactive sequence; then the round trip the multiplex loop makes with them: load the
    amplicons as the multiplex background, split the targets at begin / middle / end."""
    o = dict(target_threshold=1.0, amp_min=80, amp_max=200)
    o.update(opts)
    rng = random.Random(31415 + len(opts))
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
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in pairs_txt]
    so = oracle.session(**o)
    for s in seqs:
        so.add_target(s, 1.0)
    d = api.Screener(0)
    try:
        d.load_texts(seqs, [1.0] * len(seqs))
        for i, pos in ((1, 350), (5, 120), (9, 500)):
            so.split(i, pos)
            d.split(i, pos)
            seqs[i] = seqs[i][:pos] + "-" + seqs[i][pos + 1:]
        so.set_active(4, False)
        d.set_active([i != 4 for i in range(len(seqs))])
        so.select(pairs)
        d.select_words(pairs, float(np.float32(o["target_threshold"]) * np.float32(0.9)), 18)
        codes = [W.codes_from_text(s) for s in seqs]
        n_b = n_a = 0
        for p in pairs:
            bo, ao = so.collect_amplicons(p, o["target_threshold"], o["amp_min"], o["amp_max"])
            rec = d.collect_amplicons(p, o["target_threshold"], o["amp_min"], o["amp_max"], cap=2)   # forces the grow-and-retry path
            assert sorted((r["sequence"], r["begin"], r["end"]) for r in rec) == sorted(bo)
            got = sorted({tuple(int(x) for x in codes[r["sequence"]][r["inner_start"]:r["inner_start"] + r["inner_length"]]) for r in rec})
            assert got == sorted(set(ao))
            assert all(0 not in a for a in got)
            n_b += len(bo)
            n_a += len(ao)
            if rec:                                                    # main.cpp:989-1017 with these amplicons
                texts = ["".join(W.text_from_codes(np.array(a, np.uint8))) for a in got]
                assert d.multiplex_load(texts, 18) > 0
        assert n_b >= 5 and n_a >= 3, (n_b, n_a)
    finally:
        d.close()


def test_move_coverage_big_buckets(oracle):
    """Hundreds of DB entries per sequence (low select threshold, every slot shift, repetitive sequences): the
    bucket capacity grows past 256 slots and pcr_move_coverage takes its workgroup-per-sequence form
    (k_pair_moves_seq: compact partner list in LDS).  Wide amplicon range, EOS splits, both sides."""
    o = dict(target_threshold=0.8, search_multiplier=0.8, amp_min=0, amp_max=1500, use_taq_mama=1,
             pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0, min_primer=16, optimize_5=1, optimize_3=1)
    rng = random.Random(8086)
    unit = rand_seq(rng, 60)
    seqs = []
    for i in range(10):
        s = "".join(mutate(rng, unit, 0.08) for _ in range(40))       # 2 400 bases of diverged repeats
        seqs.append(s)
    f, r = unit[5:27], revcomp(unit[30:52])
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)), (oracle.centered_word(mutate(rng, f, 0.1)), oracle.centered_word(r))]
    so = oracle.session(**o)
    for s in seqs:
        so.add_target(s, 1.0)
    d = api.Screener(0)
    try:
        d.load_texts(seqs, [1.0] * len(seqs))
        for i, pos in ((2, 700), (2, 1500), (7, 300)):
            so.split(i, pos)
            d.split(i, pos)
        thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
        n = so.select(pairs)
        assert d.select_words(pairs, thr, o["min_primer"], True, True) == n
        assert n > 256 * len(seqs) // 2                                # enough entries per sequence for big buckets
        for p in pairs:
            for side in (0, 1):
                var = [p[side]]
                for kind in ("inc", "trim5", "trim3", "grow5", "grow3"):
                    var += move_variants(W, p[side], kind)
                co, oo = so.move_coverage(p, side, var, orient=True)
                cd, fr, rf = d.move_coverage(p, side, var, o["target_threshold"], o["search_multiplier"], o["amp_min"],
                                             o["amp_max"], True)
                assert np.array_equal(cd, co)
                assert np.array_equal(fr, (oo & 1) != 0)
                assert np.array_equal(rf, (oo & 2) != 0)
                assert np.count_nonzero(co) > 0
    finally:
        d.close()
