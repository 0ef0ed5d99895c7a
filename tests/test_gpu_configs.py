"""BASELINE.json's configs C3, C4 (one of its eight shards) and C5 (one shard) at their full single-GPU size.

The CPU oracle cannot run these sizes (it manages ~2 400 evaluations a second), so each config is tied to it
in two steps: (1) on a sample of <= 64 sequences of the SAME workload -- the targets the first primer pairs were cut
from plus family mates -- the device equals the oracle bit for bit; (2) the full-size device result restricted to the
sample's sequences equals the sample screened alone (evaluation is independent per sequence: select_words runs per
sequence, pcr_assay.cpp:12-69 pairs sites of one sequence), so the full-size result is the oracle's for those
sequences.  On top, size-independent properties: seed scan == bit-sliced scan, fused asynchronous pass == separate
calls, coverage == weighted popcount, idempotence."""
import os

import numpy as np
import pytest

from pcramp_amd import api, synth, words as W

pytestmark = pytest.mark.gpu


def _screener(scan=None):
    old = os.environ.get("PCRAMP_SCAN")
    if scan is None:
        os.environ.pop("PCRAMP_SCAN", None)
    else:
        os.environ["PCRAMP_SCAN"] = str(scan)
    try:
        return api.Screener(0)
    finally:
        if old is None:
            os.environ.pop("PCRAMP_SCAN", None)
        else:
            os.environ["PCRAMP_SCAN"] = old


def _sample(wl, n_pairs=6, mates=8, cap=64):
    """Targets the first pairs were cut from, plus the first members of their families."""
    idx = []
    for t, _, _ in wl["origins"][:n_pairs]:
        fam0 = t // wl["family"] * wl["family"]
        for i in [t] + list(range(fam0, min(fam0 + mates, wl["T"]))):
            if i not in idx:
                idx.append(i)
    return sorted(idx)[:cap]


def _oracle_session(oracle, wl_set, idx, L, **opts):
    s = oracle.session(**opts)
    nb = (L + 1) // 2
    for i in idx:
        o = int(wl_set["byte_offsets"][i])
        s.add_target_packed(wl_set["packed"][o:o + nb], L)
    return s


def _target_pass(d, wl, thr_t=1.0, mult=0.9):
    thr = float(np.float32(thr_t) * np.float32(mult))
    n = d.select_words(wl["pairs"], thr, 18)
    _, fr, rf, cov = d.amplify(wl["pairs"], thr_t, thr_t, 80, 200, False)
    return n, fr, rf, cov


def _fused(d, wl, T, thr_t=1.0, mult=0.9):
    import torch
    words = int(d.bitset_words())
    P = len(wl["pairs"])
    out = torch.full((2, P, words), -1, dtype=torch.int64, device="cuda:0")
    thr = float(np.float32(thr_t) * np.float32(mult))
    d.screen_device(wl["pairs"], thr, out[0].data_ptr(), out[1].data_ptr(), thr_t, thr_t, 80, 200, False)
    d.synchronize()
    torch.cuda.synchronize()
    w = out.cpu().numpy().view(np.uint64)
    return [np.stack([api.bits_to_bool(w[k, i], T) for i in range(P)]) for k in range(2)]


def _check_targets_against_oracle(oracle, wl, idx, fr_full, rf_full, cov_kind="unit"):
    """(1) sample alone == oracle (word DB, bits, coverage); (2) full-size bits restricted to the sample == sample alone."""
    so = _oracle_session(oracle, wl, idx, wl["L"])
    n_o = so.select(wl["pairs"])
    d = api.Screener(0)
    try:
        d.load_sequences(*synth.subset(wl, idx, wl["L"]))
        n, fr, rf, cov = _target_pass(d, wl)
        assert n == n_o and d.entries() == so.db_entries()
        hits = 0
        for k, p in enumerate(wl["pairs"]):
            ob = so.target_match(p).astype(bool)
            assert ((fr[k] | rf[k]) == ob).all(), k
            assert cov[k] == np.float32(so.target_coverage(p)), k
            hits += int(ob.sum())
        assert hits >= 6                                   # the sample holds the origins of the first pairs
        assert np.array_equal(fr_full[:, idx], fr) and np.array_equal(rf_full[:, idx], rf)
    finally:
        d.close()
    return so


def test_c3_targets_and_backgrounds(oracle):
    """C3: 50 000 gene targets x 2 kb + 10 000 backgrounds: the target pass, then the background path --
    select_words on the backgrounds at background_threshold x multiplier = 0.72 with min length 0.9 x 18
    (main.cpp:595) and find_background_match."""
    wl = synth.workload("C3")
    bg = wl["background"]
    T, B = wl["T"], bg["B"]
    assert (T, wl["L"], B, bg["L"]) == (50000, 2000, 10000, 2000)
    a, b = _screener(None), _screener(2)
    try:
        res = []
        for d in (a, b):
            d.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
            d.load_sequences(bg["packed"], bg["byte_offsets"], bg["lengths"], which=api.BACKGROUND)
            res.append(_target_pass(d, wl))
        (n3, fr3, rf3, cov3), (n2, fr2, rf2, cov2) = res
        assert n3 == n2 and n3 > 1000                                   # seed scan == bit-sliced scan
        assert np.array_equal(fr3, fr2) and np.array_equal(rf3, rf2) and np.array_equal(cov3, cov2)
        assert np.array_equal(cov3, (fr3 | rf3).sum(axis=1).astype(np.float32))
        got = _fused(a, wl, T)
        assert np.array_equal(got[0], fr3) and np.array_equal(got[1], rf3)
        idx = _sample(wl)
        _check_targets_against_oracle(oracle, wl, idx, fr3, rf3)

        # ---- the background path
        bthr = float(np.float32(0.8) * np.float32(0.9))
        min_len = int(18 * 0.9)
        nb = [d.select_words(wl["pairs"], bthr, min_len, which=api.BACKGROUND) for d in (a, b)]
        assert nb[0] == nb[1] and nb[0] > 1000
        ref_mode = [d.find_background_match(wl["pairs"], 0.8, 0.9, 0, 2000, False) for d in (a, b)]
        all_mode = a.find_background_match(wl["pairs"], 0.8, 0.9, 0, 2000, False, evaluate_all=True)
        assert np.array_equal(ref_mode[0], ref_mode[1])
        assert not (ref_mode[0] & ~all_mode).any()                      # dropping amplicons can only clear bits
        assert np.array_equal(a.find_background_match(wl["pairs"], 0.8, 0.9, 0, 2000, False), ref_mode[0])   # idempotent
        # a lower final threshold so that bits are set (the reference pairs F with the F-site key but (R) with the
        # R-site key, background_match.cpp:82: products stay small), same candidate amplicons
        low_all = a.find_background_match(wl["pairs"], 0.45, 1.6, 0, 2000, False, evaluate_all=True)
        low_ref = a.find_background_match(wl["pairs"], 0.45, 1.6, 0, 2000, False)
        assert low_all.any() and not (low_ref & ~low_all).any()
        # sample of the backgrounds: those derived from the roots of the first pairs' families
        n_roots = (T + wl["family"] - 1) // wl["family"]
        fams = []
        for t, _, _ in wl["origins"][:8]:
            if t // wl["family"] not in fams:
                fams.append(t // wl["family"])
        bidx = sorted(i for i in range(B) if int(bg["root_of"][i]) in fams)[:64]
        assert len(bidx) >= 16 and n_roots == 1000
        so = _oracle_session(oracle, bg, bidx, bg["L"])
        n_o = so.select(wl["pairs"], threshold=bthr, min_len_override=min_len)
        d = api.Screener(0)
        try:
            d.load_sequences(*synth.subset(bg, bidx, bg["L"]), which=api.BACKGROUND)
            assert d.select_words(wl["pairs"], bthr, min_len, which=api.BACKGROUND) == n_o
            assert d.entries(which=api.BACKGROUND) == so.db_entries()
            for (t_bg, mult) in ((0.8, 0.9), (0.45, 1.6)):
                s_ref = d.find_background_match(wl["pairs"], t_bg, mult, 0, 2000, False)
                s_all = d.find_background_match(wl["pairs"], t_bg, mult, 0, 2000, False, evaluate_all=True)
                full_all = all_mode if t_bg == 0.8 else low_all
                for k, p in enumerate(wl["pairs"]):
                    o1, _ = so.background_match(p, bg_threshold=t_bg, bg_multiplier=mult, emulate_index_bug=1)
                    o0, _ = so.background_match(p, bg_threshold=t_bg, bg_multiplier=mult, emulate_index_bug=0)
                    assert (s_ref[k] == o1.astype(bool)).all(), (t_bg, k)          # reference-identical mode
                    assert (s_all[k] == o0.astype(bool)).all(), (t_bg, k)
                # scoring every amplicon is independent per background sequence: full size == sample == oracle
                assert np.array_equal(full_all[:, bidx], s_all)
            assert low_all[:, bidx].any()
        finally:
            d.close()
    finally:
        a.close()
        b.close()


def test_c4_shard(oracle):
    """C4: one of the eight shards of 5 000 genomes x 5 Mb = 625 x 5 Mb (1.56 GB packed, 3.1 G window positions)."""
    wl = synth.workload("C4_shard")
    T = wl["T"]
    assert (T, wl["L"]) == (625, 5000000)
    a, b = _screener(None), _screener(2)
    try:
        a.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
        n3, fr3, rf3, cov3 = _target_pass(a, wl)
        b.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
        n2, fr2, rf2, cov2 = _target_pass(b, wl)
        assert n3 == n2 and n3 > 100
        assert np.array_equal(fr3, fr2) and np.array_equal(rf3, rf2) and np.array_equal(cov3, cov2)
        assert np.array_equal(cov3, (fr3 | rf3).sum(axis=1).astype(np.float32)) and (fr3 | rf3).any()
        b.close()
        got = _fused(a, wl, T)
        assert np.array_equal(got[0], fr3) and np.array_equal(got[1], rf3)
        n3b, fr3b, rf3b, _ = _target_pass(a, wl)                          # idempotent
        assert n3b == n3 and np.array_equal(fr3b, fr3) and np.array_equal(rf3b, rf3)
        # the oracle scans two whole genomes (the origins of the first two pairs) with all 50 pairs
        idx = sorted({wl["origins"][0][0], wl["origins"][1][0]})
        so = _oracle_session(oracle, wl, idx, wl["L"])
        n_o = so.select(wl["pairs"])
        d = api.Screener(0)
        try:
            d.load_sequences(*synth.subset(wl, idx, wl["L"]))
            n, fr, rf, cov = _target_pass(d, wl)
            assert n == n_o and d.entries() == so.db_entries()
            for k, p in enumerate(wl["pairs"]):
                assert ((fr[k] | rf[k]) == so.target_match(p).astype(bool)).all(), k
            assert np.array_equal(fr3[:, idx], fr) and np.array_equal(rf3[:, idx], rf)
            assert (fr | rf).sum() >= 2
        finally:
            d.close()
    finally:
        a.close()
        b.close()


def test_c5_shard_iupac_and_local_search(oracle):
    """C5: one shard of 100 000 viral targets = 12 500 x 10 kb, IUPAC-degenerate primers (2-fold codes at up to three
    positions per primer), then one iteration of the optimize_pcr local search for the first pairs: every trial word of
    every move of both oligos, is_valid on the device, coverage of all surviving trials over the shard."""
    from pcramp_amd import moves
    wl = synth.workload("C5_shard")
    T = wl["T"]
    assert (T, wl["L"]) == (12500, 10000)
    assert sum(W.word_degeneracy(f) > 1 or W.word_degeneracy(r) > 1 for f, r in wl["pairs"]) >= 40
    a, b = _screener(None), _screener(2)
    try:
        res = []
        for d in (a, b):
            d.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
            res.append(_target_pass(d, wl))
        (n3, fr3, rf3, cov3), (n2, fr2, rf2, cov2) = res
        assert n3 == n2 and n3 > 1000
        assert np.array_equal(fr3, fr2) and np.array_equal(rf3, rf2) and np.array_equal(cov3, cov2)
        assert np.array_equal(cov3, (fr3 | rf3).sum(axis=1).astype(np.float32))
        got = _fused(a, wl, T)
        assert np.array_equal(got[0], fr3) and np.array_equal(got[1], rf3)
        idx = _sample(wl)
        so = _check_targets_against_oracle(oracle, wl, idx, fr3, rf3)

        # ---- one local-search iteration (optimize.cpp:120-140) at shard size, degeneracy bound 16
        a.select_words(wl["pairs"], float(np.float32(1.0) * np.float32(0.9)), 18)
        s = api.Screener(0)
        try:
            s.load_sequences(*synth.subset(wl, idx, wl["L"]))
            s.select_words(wl["pairs"], float(np.float32(1.0) * np.float32(0.9)), 18)
            checked = 0
            for pi in range(4):
                pair = wl["pairs"][pi]
                for side in (0, 1):
                    trials = []
                    for mv in moves.DEFAULT_MOVES:
                        trials += api.host_move_trials(pair[side], mv, 16, 18, 25)
                    ok = a.is_valid(trials, check_homo_dimer=False, tm_min=40.0, tm_max=80.0, flags=True)
                    live = [t for t, v in zip(trials, ok) if v]
                    if len(live) < 2:                     # a hairpin-prone primer: no trial passes is_valid
                        continue
                    cov_f, fr_f, rf_f = a.move_coverage(pair, side, live)
                    cov_s, fr_s, rf_s = s.move_coverage(pair, side, live)
                    cov_o, ori_o = so.move_coverage(pair, side, live, orient=True)
                    # shard == popcount of its bits; sample == oracle; shard restricted to the sample == sample
                    assert np.array_equal(cov_f, (fr_f | rf_f).sum(axis=1).astype(np.float32))
                    assert np.array_equal(cov_s, cov_o)
                    assert np.array_equal((fr_s | rf_s), ori_o.astype(bool))
                    assert np.array_equal(fr_f[:, idx], fr_s) and np.array_equal(rf_f[:, idx], rf_s)
                    checked += len(live)
                # the assembled iteration picks the same winner on the sample as the oracle's optimization_move loop would:
                # covered by test_gpu_moves on small sets; here the whole optimize() runs at shard size and must terminate
                best, score = moves.optimize(a, pair, degen=16, have_background=False, tm_min=40.0, tm_max=80.0)
                base = moves.base_score(a, pair, have_background=False)
                assert float(score[0]) >= float(base[0])
            assert checked > 100
        finally:
            s.close()
    finally:
        a.close()
        b.close()
