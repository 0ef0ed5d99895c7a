"""The CPU oracle against the committed golden vectors (tests/golden/*.json, captured from the
compiled reference by oracle/make_golden.py).  Runs anywhere, including the GPU box."""
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(G, name + ".json")) as f:
        return json.load(f)


def w(h):
    return (int(h[0], 16), int(h[1], 16))


def test_words(oracle):
    g = load("words")
    for c in g["pairs"]:
        a, b = w(c["a"]), w(c["b"])
        assert oracle.word_and(a, b) == c["and"]
        assert oracle.word_size(a) == c["size"]
        assert oracle.word_start(a) == c["start"]
        assert oracle.word_stop(a) == c["stop"]
        assert oracle.word_degeneracy(a) == c["degeneracy"]
        if "center" in c:
            assert oracle.word_center(a) == w(c["center"])
            assert oracle.word_complement(a) == w(c["complement"])
    for e in g["expansions"]:
        assert oracle.word_expand(w(e["word"])) == [w(x) for x in e["expansion"]]
    for p1, p2, t1, t2, v in g["taq_mama"]:
        assert oracle.taq_mama(p1, p2, t1, t2) == np.float32(v)


def test_pack(oracle):
    for c in load("pack")["cases"]:
        got = oracle.pack(c["seq"], c["index"], c["degen_thr"], c["min_gc"], c["max_gc"], c["min_len"])
        want = sorted((int(a, 16), int(b, 16), loc, idx, st) for a, b, loc, idx, st in c["entries"])
        assert got == want


def screen_cases():
    return load("screen")["cases"]


@pytest.mark.parametrize("ci", range(5))
def test_screen(oracle, ci):
    c = screen_cases()[ci]
    s = oracle.session(**c["options"])
    for q, wt in zip(c["seqs"], c["weights"]):
        s.add_target(q, wt)
    for i in c["inactive"]:
        s.set_active(i, False)
    for i, pos in c["splits"]:
        s.split(i, pos)
    pairs = [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pairs"]]
    assert s.select(pairs) == c["n_entries"]
    assert s.db_entries() == sorted((int(a, 16), int(b, 16), loc, idx, st) for a, b, loc, idx, st in c["db"])
    for k, p in enumerate(pairs):
        assert s.target_match(p).tolist() == c["bits"][k]
        assert s.target_coverage(p) == np.float32(c["coverage"][k])


def test_sw(oracle):
    for c in load("sw")["cases"]:
        r = oracle.sw_align_words(w(c["q"]), w(c["t"]))
        assert r.score == c["score"]
        if c["rest"] is not None:
            assert list(r.tup()[1:]) == c["rest"]


def test_thermo(oracle):
    g = load("thermo")
    for c in g["oligos"]:
        o = oracle.thermo_full(c["seq"], c["salt"], c["strand"])
        assert (o == np.array(c["out"], np.float32)).all(), c["seq"]
    for c in g["hetero"]:
        o = oracle.heterodimer_full(c["a"], c["b"], 0.05, c["sa"], c["sb"])
        assert (o == np.array(c["out"], np.float32)).all()
    for c in g["is_valid"]:
        assert oracle.is_valid(w(c["word"]), **c["kw"]) == c["valid"]
    for c in g["dimers"]:
        p = (w(c["pair"][:2]), w(c["pair"][2:]))
        q = (w(c["other"][:2]), w(c["other"][2:]))
        assert oracle.max_dimer_tm(p) == np.float32(c["max_dimer_tm"])
        for md, v in c["compatible"].items():
            assert oracle.multiplex_compatible(p, q, max_dimer=float(md)) == v


@pytest.mark.parametrize("ci", range(3))
def test_moves(oracle, ci):
    """Complete local-search moves (optimize_pcr.cpp via optimization_move()): the reference's returned trial
    word and Score for six moves x both oligos x five assays."""
    from oracle_lib import optimization_move, optimize
    c = load("moves")["cases"][ci]
    ts, bs = oracle.session(**c["options"]), oracle.session(**c["options"])
    for q, wt in zip(c["seqs"], c["weights"]):
        ts.add_target(q, wt)
    for q in c["backgrounds"]:
        bs.add_target(q, 1.0)
    pairs = [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pairs"]]
    ts.select(pairs)
    bs.select(pairs, threshold=c["bg_select_threshold"], min_len_override=c["bg_min_len"])
    for pi, side, move, wh, sc, base in c["moves"]:
        got = optimization_move(oracle, ts, bs, pairs[pi], move, side, **c["move_options"])
        assert got[0] == (int(wh[0], 16), int(wh[1], 16))
        assert got[1] == tuple(float(np.float32(x)) for x in sc)
        assert got[2] == tuple(float(np.float32(x)) for x in base)
    for pi, bp, sc in c["optimize"]:                                   # the whole optimize() loop
        got = optimize(oracle, ts, bs, pairs[pi], **c["move_options"])
        assert got[0] == ((int(bp[0], 16), int(bp[1], 16)), (int(bp[2], 16), int(bp[3], 16)))
        assert got[1] == tuple(float(np.float32(x)) for x in sc)


@pytest.mark.parametrize("ci", range(5))
def test_make_degenerate(oracle, ci):
    """make_degenerate (optimize.cpp:356-398 -> PCR::maximize_degeneracy): the reference's resulting assay and return value
    for 14 assays per case (tests/golden/degenerate.json), incl. the greedy heterodimer reduction and its failure exit."""
    from oracle_lib import make_degenerate
    c = load("degenerate")["cases"][ci]
    ts = oracle.session(**c["options"])
    for q, wt in zip(c["seqs"], c["weights"]):
        ts.add_target(q, wt)
    pairs = [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pairs"]]
    ts.select(pairs)
    changed = 0
    for p, (wh, ok) in zip(pairs, c["degenerate"]):
        got = make_degenerate(oracle, ts, p, max_dimer=c["max_dimer"], **c["move_options"])
        assert got[0] == ((int(wh[0], 16), int(wh[1], 16)), (int(wh[2], 16), int(wh[3], 16)))
        assert got[1] == bool(ok)
        changed += got[0] != p
    assert changed >= 3


def _sampler_session(lib, c):
    sess = lib.session()
    for q, a in zip(c["seqs"], c["active"]):
        sess.add_target(q, 1.0, a)
    for i, pos in c["splits"]:
        sess.split(i, pos)
    return sess


def test_rand_r(oracle):
    """glibc rand_r as the reference's libc produced it (golden) and as this host's libc does."""
    import ctypes
    from oracle_lib import rand_r
    for seed, vals, after in load("sampler")["rand_r"]:
        s = seed
        for v in vals:
            got, s = rand_r(oracle, s)
            assert got == v
        assert s == after
    libc = ctypes.CDLL("libc.so.6")
    for seed in (0, 5, 123456789, 0x80000000, 0xfffffffe):
        st = ctypes.c_uint(seed)
        v = libc.rand_r(ctypes.byref(st))
        assert rand_r(oracle, seed) == (v, st.value)


@pytest.mark.parametrize("ci", range(4))
def test_sampler(oracle, ci):
    """PCR::random_assay on a running rand_r state (the reference's own output): assays and final state."""
    from oracle_lib import random_assays
    c = load("sampler")["cases"][ci]
    sess = _sampler_session(oracle, c)
    for seed, pairs, after in c["runs"]:
        got, s = random_assays(oracle, sess, seed, len(pairs), **c["sampler_options"])
        assert got == [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in pairs], seed
        assert s == after, seed


def _hw(h):
    return (int(h[0], 16), int(h[1], 16))


def test_overlap(oracle):
    """Word::max_overlap / PCR::compute_oligo_overlap as the reference computed them."""
    g = load("overlap")
    words = [_hw(h) for h in g["words"]]
    for i, j, v in g["max_overlap"]:
        assert np.float32(oracle.max_overlap(words[i], words[j])) == np.float32(v), (i, j)
    for c in g["oligo_overlap"]:
        a = c["assay"]
        pool = [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pool"]]
        assert np.float32(oracle.oligo_overlap((_hw(a[:2]), _hw(a[2:])), pool)) == np.float32(c["overlap"])


@pytest.mark.parametrize("ci", range(2))
def test_multiplex_coverage(oracle, ci):
    c = load("multiplex")["cases"][ci]
    sess = oracle.session(min_primer=c["min_primer"])
    for a in c["amplicons"]:
        sess.add_target(a, 1.0)
    pairs = [(_hw(p[:2]), _hw(p[2:])) for p in c["pairs"]]
    for pi, side, thr, var, cov, nk in c["rows"]:
        got, k = sess.multiplex_coverage(pairs[pi], side, [_hw(v) for v in var], thr, c["use_taq_mama"])
        assert k == nk
        assert np.array_equal(got, np.array(cov, np.float32)), (pi, side, thr)


@pytest.mark.parametrize("ci", range(3))
def test_multiplex_optimize(oracle, ci):
    """optimize() with opt.use_multiplex: the reference's final assay and Score."""
    from oracle_lib import optimize_multiplex
    c = load("multiplex_optimize")["cases"][ci]
    sess = {k: c["options"][k] for k in ("target_threshold", "use_taq_mama")}
    ts, bs, ams = oracle.session(**c["options"]), oracle.session(**c["options"]), oracle.session(**sess)
    for q, wt in zip(c["seqs"], c["weights"]):
        ts.add_target(q, wt)
    for q in c["backgrounds"]:
        bs.add_target(q, 1.0)
    for q in c["amplicons"]:
        ams.add_target(q, 1.0)
    pw = lambda p: (_hw(p[:2]), _hw(p[2:]))
    pool, cands = [pw(p) for p in c["pool"]], [pw(p) for p in c["candidates"]]
    ts.select(cands + pool)
    bs.select(cands + pool, threshold=c["bg_select_threshold"], min_len_override=c["bg_min_len"])
    for pi, use_pool, bp, sc in c["optimize"]:
        got = optimize_multiplex(oracle, ts, bs, ams, pool if use_pool else [], cands[pi], **c["move_options"])
        assert got[0] == pw(bp), (pi, use_pool)
        assert got[1] == tuple(float(np.float32(x)) for x in sc), (pi, use_pool)


@pytest.mark.parametrize("ci", range(3))
def test_collect_amplicons(oracle, ci):
    c = load("amplicons")["cases"][ci]
    o = c["options"]
    sess = oracle.session(**o)
    for q in c["seqs"]:
        sess.add_target(q, 1.0)
    for i, pos in c["splits"]:
        sess.split(i, pos)
    for i in c["inactive"]:
        sess.set_active(i, False)
    pairs = [(_hw(p[:2]), _hw(p[2:])) for p in c["pairs"]]
    sess.select(pairs)
    for p, row in zip(pairs, c["rows"]):
        b, a = sess.collect_amplicons(p, o["target_threshold"], o["amp_min"], o["amp_max"])
        assert [list(x) for x in b] == row["bounds"]
        assert ["".join("%x" % v for v in t) for t in a] == row["amplicons"]


def _pairs(c):
    return [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pairs"]]


@pytest.mark.parametrize("ci", range(9))
def test_background_match(oracle, ci):
    """find_background_match of the compiled reference (tests/golden/background.json): amplicon counts below,
    equal to and above the number of sequences; the oracle in its reference-identical mode."""
    c = load("background")["cases"][ci]
    seqs = c["seqs"] + [c["pad"]] * c["n_pad"]
    s = oracle.session()
    for q in seqs:
        s.add_target(q)
    pairs = _pairs(c)
    assert s.select(pairs, threshold=c["select_threshold"], min_len_override=c["min_len"]) == c["n_entries"]
    for pi, n_amp, want in c["rows"]:
        got, _ = s.background_match(pairs[pi], emulate_index_bug=1, **c["kw"])
        assert np.nonzero(got)[0].tolist() == want, (ci, pi, n_amp)


def test_background_golden_covers_the_regimes():
    cases = load("background")["cases"]
    tot = {k: sum(c["regimes"][k] for c in cases) for k in ("below", "equal", "above")}
    assert tot["below"] >= 5 and tot["equal"] >= 2 and tot["above"] >= 50
    assert sum(len(r[2]) for c in cases for r in c["rows"]) > 200


@pytest.mark.parametrize("ci", range(2))
def test_multiplex_match(oracle, ci):
    c = load("multiplex_match")["cases"][ci]
    s = oracle.session()
    for q in c["seqs"]:
        s.add_target(q)
    pairs = _pairs(c)
    hits = 0
    for pi, thr, want in c["rows"]:
        assert s.multiplex_match(pairs[pi], thr, c["use_taq_mama"]).tolist() == want
        hits += sum(want)
    assert hits > 10
