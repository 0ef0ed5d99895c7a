"""The HIP path against the committed golden vectors captured from the compiled reference
(tests/golden/screen.json): word DB, amplification bits, coverage -- bit-exact."""
import json
import os

import numpy as np
import pytest

from pcramp_amd import api

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("ci", range(5))
def test_screen_golden(ci):
    with open(os.path.join(G, "screen.json")) as f:
        c = json.load(f)["cases"][ci]
    o = c["options"]
    d = api.Screener(0, pack_max_degen=o["pack_max_degen"], pack_min_gc=o["pack_min_gc"], pack_max_gc=o["pack_max_gc"])
    d.load_texts(c["seqs"], c["weights"])
    active = np.ones(len(c["seqs"]), np.uint8)
    for i in c["inactive"]:
        active[i] = 0
    d.set_active(active)
    for i, pos in c["splits"]:
        d.split(i, pos)
    pairs = [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pairs"]]
    thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
    n = d.select_words(pairs, thr, o["min_primer"], o["optimize_5"], o["optimize_3"])
    assert n == c["n_entries"]
    assert d.entries() == sorted((int(a, 16), int(b, 16), loc, idx, st) for a, b, loc, idx, st in c["db"])
    bits = d.find_target_match(pairs, o["target_threshold"], o["amp_min"], o["amp_max"], o["use_taq_mama"])
    cov = d.compute_coverage(pairs, o["target_threshold"], o["search_multiplier"], o["amp_min"], o["amp_max"], o["use_taq_mama"])
    for k in range(len(pairs)):
        assert bits[k].astype(int).tolist() == c["bits"][k]
        assert cov[k] == np.float32(c["coverage"][k])
    d.close()


def _w(h):
    return (int(h[0], 16), int(h[1], 16))


def test_sw_golden():
    with open(os.path.join(G, "sw.json")) as f:
        cases = json.load(f)["cases"]
    d = api.Screener(0)
    got = d.sw_align_words([_w(c["q"]) for c in cases], [_w(c["t"]) for c in cases])
    for c, r in zip(cases, got):
        assert r[0] == c["score"]
        if c["rest"] is not None:
            assert list(r[1:7]) == c["rest"]
    d.close()


def test_thermo_golden():
    with open(os.path.join(G, "thermo.json")) as f:
        g = json.load(f)
    d = api.Screener(0)
    from pcramp_amd import words as W
    by = {}
    for c in g["oligos"]:
        by.setdefault((c["salt"], c["strand"]), []).append(c)
    for (salt, strand), cs in by.items():
        res = d.is_valid([W.centered_word(W.codes_from_text(c["seq"])) for c in cs], True, salt=salt, primer_strand=strand)
        for c, r in zip(cs, res):
            o = np.array(c["out"], np.float32)
            # north_star asks for 1e-6 degC / kcal; the device reproduces the reference's float operations in order, so
            # the comparison is bit equality (Tm, dH, dS, dG at 37 C, hairpin Tm, homodimer Tm)
            for got, want in ((r["tm"], o[0]), (r["dH"], o[1]), (r["dS"], o[2]), (r["dG"], o[3]), (r["hairpin_tm"], o[4]), (r["homodimer_tm"], o[7])):
                assert np.float32(got) == np.float32(want), c["seq"]
    for c in g["is_valid"]:
        kw = dict(c["kw"])
        chk = kw.pop("check_homo_dimer")
        assert int(d.is_valid([_w(c["word"])], chk, **kw)[0]["valid"]) == c["valid"]
    pairs = [(_w(c["pair"][:2]), _w(c["pair"][2:])) for c in g["dimers"]]
    others = [(_w(c["other"][:2]), _w(c["other"][2:])) for c in g["dimers"]]
    tm = d.max_dimer_tm(pairs)
    for c, t in zip(g["dimers"], tm):
        assert np.float32(t) == np.float32(c["max_dimer_tm"])
    for md in ("10.0", "25.0", "40.0"):
        ok = d.multiplex_compatible(pairs, others, max_dimer=float(md))
        for c, o in zip(g["dimers"], ok):
            assert int(o) == c["compatible"][md]
    d.close()


@pytest.mark.parametrize("ci", range(3))
def test_moves_golden(ci):
    """Complete local-search moves (tests/golden/moves.json: the reference's own optimization_move()) through
    pcramp_amd.moves: host trial generation, device is_valid, device move coverage, Score logic."""
    from pcramp_amd import moves
    with open(os.path.join(G, "moves.json")) as f:
        c = json.load(f)["cases"][ci]
    o, mo = c["options"], c["move_options"]
    d = api.Screener(0)
    try:
        d.load_texts(c["seqs"], c["weights"], which=api.TARGET)
        d.load_texts(c["backgrounds"], [1.0] * len(c["backgrounds"]), which=api.BACKGROUND)
        pairs = [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pairs"]]
        thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
        d.select_words(pairs, thr, o["min_primer"], o["optimize_5"], o["optimize_3"], which=api.TARGET)
        d.select_words(pairs, float(np.float32(c["bg_select_threshold"])), c["bg_min_len"], o["optimize_5"], o["optimize_3"],
                       which=api.BACKGROUND)
        for pi, side, move, wh, sc, base in c["moves"]:
            w, s = moves.optimization_move(d, pairs[pi], move, side, target_threshold=o["target_threshold"],
                                           search_multiplier=o["search_multiplier"], amp_min=o["amp_min"], amp_max=o["amp_max"],
                                           use_taq_mama=bool(o["use_taq_mama"]), **mo)
            assert w == (int(wh[0], 16), int(wh[1], 16)), (pi, side, move)
            assert tuple(float(x) for x in s) == tuple(float(np.float32(x)) for x in sc), (pi, side, move)
        for pi, bp, sc in c["optimize"]:                               # the whole optimize() loop
            got, s = moves.optimize(d, pairs[pi], target_threshold=o["target_threshold"], search_multiplier=o["search_multiplier"],
                                    amp_min=o["amp_min"], amp_max=o["amp_max"], use_taq_mama=bool(o["use_taq_mama"]), **mo)
            assert got == ((int(bp[0], 16), int(bp[1], 16)), (int(bp[2], 16), int(bp[3], 16))), pi
            assert tuple(float(x) for x in s) == tuple(float(np.float32(x)) for x in sc), pi
    finally:
        d.close()


@pytest.mark.parametrize("ci", range(5))
def test_make_degenerate_golden(ci, oracle):
    """pcr_make_degenerate (the top-down start of the local search, --optimize.top-down) against tests/golden/degenerate.json --
    the reference's own make_degenerate (optimize.cpp:356-398 -> PCR::maximize_degeneracy): the resulting assays and return
    values of a batch of 14 assays in one call, incl. the greedy heterodimer reduction and its failure exit; and the same
    batch against the oracle with the assays in another order (the batch shares its thermodynamics rounds)."""
    from pcramp_amd import moves
    from oracle_lib import make_degenerate as oracle_make_degenerate
    with open(os.path.join(G, "degenerate.json")) as f:
        c = json.load(f)["cases"][ci]
    o, mo = c["options"], c["move_options"]
    pairs = [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pairs"]]
    want = [(((int(w[0], 16), int(w[1], 16)), (int(w[2], 16), int(w[3], 16))), bool(ok)) for w, ok in c["degenerate"]]
    kw = dict(target_threshold=o["target_threshold"], search_multiplier=o["search_multiplier"], amp_min=o["amp_min"], amp_max=o["amp_max"],
              use_taq_mama=bool(o["use_taq_mama"]), **mo)
    d = api.Screener(0)
    try:
        d.load_texts(c["seqs"], c["weights"], which=api.TARGET)
        thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
        d.select_words(pairs, thr, o["min_primer"], o["optimize_5"], o["optimize_3"], which=api.TARGET)
        got, ok = moves.make_degenerate(d, pairs, max_dimer=c["max_dimer"], **kw)
        for k in range(len(pairs)):
            assert (got[k], ok[k]) == want[k], k
        assert sum(g != p for g, p in zip(got, pairs)) >= 3
        # one assay per call, and the batch reversed: the same answers
        for k in (0, 5, len(pairs) - 1):
            g1, o1 = moves.make_degenerate(d, [pairs[k]], max_dimer=c["max_dimer"], **kw)
            assert (g1[0], o1[0]) == want[k]
        gr, okr = moves.make_degenerate(d, pairs[::-1], max_dimer=c["max_dimer"], **kw)
        assert list(zip(gr, okr)) == want[::-1]
        # the oracle beside it (it is pinned to the reference by the same fixture and by test_oracle_vs_reference)
        ts = oracle.session(**o)
        for q, wt in zip(c["seqs"], c["weights"]):
            ts.add_target(q, wt)
        ts.select(pairs)
        for k in (1, 7):
            assert oracle_make_degenerate(oracle, ts, pairs[k], max_dimer=c["max_dimer"], **mo) == want[k]
    finally:
        d.close()


@pytest.mark.parametrize("ci", range(4))
def test_sampler_golden(ci):
    """pcr_random_assays against tests/golden/sampler.json (the reference's PCR::random_assay on a running
    rand_r state): the same assays and the same state afterwards."""
    with open(os.path.join(G, "sampler.json")) as f:
        c = json.load(f)["cases"][ci]
    d = api.Screener(0)
    try:
        d.load_texts(c["seqs"], [1.0] * len(c["seqs"]))
        d.set_active([bool(a) for a in c["active"]])
        for i, pos in c["splits"]:
            d.split(i, pos)
        for seed, pairs, after in c["runs"]:
            got, s, _ = d.random_assays(seed, len(pairs), **c["sampler_options"])
            assert got == [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in pairs], seed
            assert s == after, seed
    finally:
        d.close()


@pytest.mark.parametrize("ci", range(2))
def test_multiplex_golden(ci):
    """pcr_multiplex_load + pcr_multiplex_coverage against tests/golden/multiplex.json (the reference's
    compute_multiplex_background_coverage over a DB packed like main.cpp:989-1001)."""
    with open(os.path.join(G, "multiplex.json")) as f:
        c = json.load(f)["cases"][ci]
    hw = lambda h: (int(h[0], 16), int(h[1], 16))
    d = api.Screener(0)
    try:
        nk = d.multiplex_load(c["amplicons"], c["min_primer"])
        pairs = [(hw(p[:2]), hw(p[2:])) for p in c["pairs"]]
        for pi, side, thr, var, cov, want_keys in c["rows"]:
            assert nk == want_keys
            got = d.multiplex_coverage(pairs[pi], side, [hw(v) for v in var], thr, bool(c["use_taq_mama"]))
            assert np.array_equal(got, np.array(cov, np.float32)), (pi, side, thr)
    finally:
        d.close()


@pytest.mark.parametrize("ci", range(3))
def test_multiplex_optimize_golden(ci):
    """pcramp_amd.moves.optimize with opt.use_multiplex against the reference's own optimize()
    (tests/golden/multiplex_optimize.json): final assay and Score incl. the oligo-reuse term."""
    from pcramp_amd import moves
    with open(os.path.join(G, "multiplex_optimize.json")) as f:
        c = json.load(f)["cases"][ci]
    o, mo = c["options"], c["move_options"]
    pw = lambda p: ((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16)))
    pool, cands = [pw(p) for p in c["pool"]], [pw(p) for p in c["candidates"]]
    d = api.Screener(0)
    try:
        d.load_texts(c["seqs"], c["weights"], which=api.TARGET)
        d.load_texts(c["backgrounds"], [1.0] * len(c["backgrounds"]), which=api.BACKGROUND)
        d.multiplex_load(c["amplicons"], o["min_primer"])
        thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
        d.select_words(cands + pool, thr, o["min_primer"], o["optimize_5"], o["optimize_3"], which=api.TARGET)
        d.select_words(cands + pool, float(np.float32(c["bg_select_threshold"])), c["bg_min_len"], o["optimize_5"], o["optimize_3"],
                       which=api.BACKGROUND)
        for pi, use_pool, bp, sc in c["optimize"]:
            got, s = moves.optimize(d, cands[pi], pool=(pool if use_pool else []), target_threshold=o["target_threshold"],
                                    search_multiplier=o["search_multiplier"], amp_min=o["amp_min"], amp_max=o["amp_max"],
                                    use_taq_mama=bool(o["use_taq_mama"]), **mo)
            assert got == pw(bp), (pi, use_pool)
            assert tuple(float(x) for x in s) == tuple(float(np.float32(x)) for x in sc), (pi, use_pool)
    finally:
        d.close()


@pytest.mark.parametrize("ci", range(3))
def test_collect_amplicons_golden(ci):
    """pcr_collect_amplicons against tests/golden/amplicons.json (the reference's collect_unique_amplicons)."""
    from pcramp_amd import words as W
    with open(os.path.join(G, "amplicons.json")) as f:
        c = json.load(f)["cases"][ci]
    o = c["options"]
    hw = lambda h: (int(h[0], 16), int(h[1], 16))
    d = api.Screener(0)
    try:
        seqs = list(c["seqs"])
        d.load_texts(seqs, [1.0] * len(seqs))
        for i, pos in c["splits"]:
            d.split(i, pos)
            seqs[i] = seqs[i][:pos] + "-" + seqs[i][pos + 1:]
        d.set_active([i not in c["inactive"] for i in range(len(seqs))])
        pairs = [(hw(p[:2]), hw(p[2:])) for p in c["pairs"]]
        thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
        d.select_words(pairs, thr, o["min_primer"])
        codes = [W.codes_from_text(s) for s in seqs]
        for p, row in zip(pairs, c["rows"]):
            rec = d.collect_amplicons(p, o["target_threshold"], o["amp_min"], o["amp_max"])
            assert sorted([r["sequence"], r["begin"], r["end"]] for r in rec) == sorted(row["bounds"])
            got = {"".join("%x" % int(v) for v in codes[r["sequence"]][r["inner_start"]:r["inner_start"] + r["inner_length"]]) for r in rec}
            assert got == set(row["amplicons"])
    finally:
        d.close()


@pytest.mark.parametrize("ci", range(9))
def test_background_golden(ci):
    """pcr_background_match in its default (reference-identical) mode against the compiled reference's own
    find_background_match (tests/golden/background.json): candidate amplicon counts below, equal to and above the
    number of sequences, where background_match.cpp:122 drops the odd-indexed amplicon of every couple."""
    with open(os.path.join(G, "background.json")) as f:
        c = json.load(f)["cases"][ci]
    seqs = c["seqs"] + [c["pad"]] * c["n_pad"]
    kw = c["kw"]
    d = api.Screener(0)
    try:
        d.load_texts(seqs, which=api.BACKGROUND)
        pairs = [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pairs"]]
        assert d.select_words(pairs, c["select_threshold"], c["min_len"], which=api.BACKGROUND) == c["n_entries"]
        bits = d.find_background_match(pairs, kw["bg_threshold"], kw["bg_multiplier"], 0, kw.get("amp_max", 2000), kw["use_taq_mama"])
        for pi, n_amp, want in c["rows"]:
            assert np.nonzero(bits[pi])[0].tolist() == want, (ci, pi, n_amp)
    finally:
        d.close()


@pytest.mark.parametrize("ci", range(2))
def test_multiplex_match_golden(ci):
    with open(os.path.join(G, "multiplex_match.json")) as f:
        c = json.load(f)["cases"][ci]
    d = api.Screener(0)
    try:
        d.load_texts(c["seqs"], which=api.BACKGROUND)
        pairs = [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pairs"]]
        for thr in sorted(set(r[1] for r in c["rows"])):
            bits = d.find_multiplex_background_match(pairs, thr, c["use_taq_mama"])
            for pi, t, want in c["rows"]:
                if t == thr:
                    assert bits[pi].astype(int).tolist() == want, (ci, pi, thr)
    finally:
        d.close()
