"""pcr_multiplex_screen: the multiplex compatibility filter of the trial loop (main.cpp:744-803) for all trial assays in one
call, against the same three quantities composed from the oracle's own pieces (each pinned to the compiled reference in
test_oracle_vs_reference.py: multiplex_compatible, find_multiplex_background_match, collect_unique_amplicons).
Run with `-m gpu`."""
import random

import numpy as np
import pytest

from pcramp_amd import api, words as W
from testdata import multiplex_design_case, revcomp, sample_pair

pytestmark = pytest.mark.gpu


def _text(word):
    return W.text_from_codes(np.array([c for c in W.slots_from_word(word) if c], dtype=np.uint8))


def _expected(oracle, ts, ams, pool, trials, thr_t, bg_thr, taq):
    comp, mcov, pcov = [], [], []
    for t in trials:
        ok = all(bool(oracle.multiplex_compatible(p, t)) for p in pool)
        comp.append(ok)
        if not ok:
            mcov.append(0.0)
            pcov.append(0.0)
            continue
        mcov.append(float(np.float32(ams.multiplex_match(t, bg_thr, taq).sum())) if ams.n else 0.0)   # weights 1
        _, amps = ts.collect_amplicons(t, thr_t, 80, 200)
        if not amps or not pool:
            pcov.append(0.0)
            continue
        loc = oracle.session(use_taq_mama=taq)
        for a in amps:
            loc.add_target(W.text_from_codes(np.array(a, dtype=np.uint8)), 1.0)
        union = np.zeros(len(amps), dtype=bool)
        for p in pool:
            union |= loc.multiplex_match(p, bg_thr, taq).astype(bool)
        pcov.append(float(union.sum()))
    return comp, mcov, pcov


@pytest.mark.parametrize("seed,taq", [(1, 0), (2, 0), (3, 1)])
def test_multiplex_screen_matches_oracle_pieces(oracle, seed, taq):
    rng = random.Random(7700 + seed)
    seqs, bgs, amps, pool, cands = multiplex_design_case(rng, oracle)
    # a trial that cannot share a tube with the pool (its F is the reverse complement of a pooled primer) ...
    cands.append((oracle.centered_word(revcomp(_text(pool[0][0]))), cands[0][1]))
    # ... and trials whose own amplicon holds a pooled assay's whole footprint (the existing primers bind the proposed amplicon)
    for pf, pr in pool:
        ftxt, rtxt = _text(pf), _text(pr)
        for t in seqs:
            i, j = t.find(ftxt), t.find(revcomp(rtxt))
            if i >= 25 and 0 < j - i < 130 and j + len(rtxt) + 25 <= len(t):
                cands.append((oracle.centered_word(t[i - 24:i - 4]), oracle.centered_word(revcomp(t[j + len(rtxt) + 4:j + len(rtxt) + 24]))))
                break
    thr_t, mult, bg_thr = 0.9, 0.9, 0.8
    o = dict(target_threshold=thr_t, search_multiplier=mult, amp_min=80, amp_max=200, use_taq_mama=taq, pack_max_degen=256,
             pack_min_gc=0.0, pack_max_gc=1.0, min_primer=18, optimize_5=0, optimize_3=0)
    ts = oracle.session(**o)
    for q in seqs:
        ts.add_target(q, 1.0)
    ams = oracle.session(use_taq_mama=taq)
    for a in amps:
        ams.add_target(a, 1.0)
    every = cands + pool
    ts.select(every)
    thr = float(np.float32(thr_t) * np.float32(mult))
    d = api.Screener(0)
    try:
        d.load_texts(seqs, [1.0] * len(seqs))
        d.load_texts(amps, [1.0] * len(amps), which=api.MULTIPLEX)
        d.select_words(every, thr, 18)
        comp, mcov, pcov = d.multiplex_screen(cands, pool, background_threshold=bg_thr, use_taq_mama=bool(taq), target_threshold=thr_t)
        wc, wm, wp = _expected(oracle, ts, ams, pool, cands, thr_t, bg_thr, taq)
        assert list(comp) == wc
        assert [float(x) for x in mcov] == wm
        assert [float(x) for x in pcov] == wp
        assert not all(wc) and any(wc)                          # the incompatible trial is caught, others pass
        assert any(x > 0 for x in wm) and any(x > 0 for x in wp), (wm, wp)
        # detail mask: covers only where asked, compatibility for all
        mask = np.array([k % 2 for k in range(len(cands))], dtype=np.uint8)
        comp2, mcov2, pcov2 = d.multiplex_screen(cands, pool, background_threshold=bg_thr, use_taq_mama=bool(taq), target_threshold=thr_t, detail=mask)
        assert list(comp2) == wc
        assert [float(x) for x in mcov2] == [m if k % 2 else 0.0 for k, m in enumerate(wm)]
        assert [float(x) for x in pcov2] == [p if k % 2 else 0.0 for k, p in enumerate(wp)]
        # empty pool: everything compatible, no pool term
        comp3, mcov3, pcov3 = d.multiplex_screen(cands, [], background_threshold=bg_thr, use_taq_mama=bool(taq), target_threshold=thr_t)
        assert all(comp3) and not pcov3.any()
        # the target and multiplex sets are untouched by the internal scratch set
        assert d.num_sequences() == len(seqs) and d.num_sequences(api.MULTIPLEX) == len(amps)
    finally:
        d.close()


def test_unknown_set_is_rejected():
    d = api.Screener(0)
    try:
        with pytest.raises(RuntimeError):
            d.load_texts(["ACGT" * 20], [1.0], which=3)        # the scratch slot is not the caller's
        with pytest.raises(RuntimeError):
            d.load_texts(["ACGT" * 20], [1.0], which=7)
        assert d.num_sequences(9) == 0
    finally:
        d.close()
