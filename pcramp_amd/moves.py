"""Local-search moves of the reference's optimiser (optimize_pcr.cpp, optimize.cpp:61-140) over the
device primitives: trial generation on the host (pcr_host_move_trials), PCR::is_valid on the device
(pcr_thermo, no dimer check), coverage of every surviving trial in one call per sequence set
(pcr_move_coverage), then the reference's coverage-bound shortcut and Score comparison.

optimize() also runs with opt.use_multiplex (`pool` = the assays designed so far, the multiplex background
keys loaded with Screener.multiplex_load): the multiplex background coverage joins the background term
(pcr_multiplex_coverage), Score.oligo_overlap is the oligo-reuse term (pcr_host_pool_overlaps), the coverage
bound turns from `<= 0` to `< 0`.  The word DBs of the target and background sets must have been built for
the current trial assays (Screener.select_words on each set), as optimize() is called inside main.cpp's
per-iteration DB build.
"""
import numpy as np

from . import api
from . import words as W

INCREASE_DEGENERACY, DECREASE_DEGENERACY, TRIM5, TRIM3, GROW5, GROW3 = range(6)

EMPTY_SCORE = (np.float32(-1.0e6), np.float32(1.0e6), np.float32(0.0))     # Score(), pcramp.h:176-179
REUSE_BONUS = np.float32(10.0)                                              # MULTIPLEX_OLIGO_REUSE_BONUS, assay.h:19


def _reuse(v):
    return REUSE_BONUS if np.float32(v) == np.float32(1.0) else np.float32(v)


def _accuracy(sc):
    return np.float32(sc[0]) - np.float32(sc[1])                            # Score::accuracy, pcramp.h:205


def score_gt(a, b):
    """Score::operator> (pcramp.h:190-197)."""
    if _accuracy(a) == _accuracy(b):
        return np.float32(a[2]) > np.float32(b[2])
    return _accuracy(a) > _accuracy(b)


def base_score(scr, pair, target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200, use_taq_mama=False,
               bg_threshold=0.8, bg_multiplier=0.9, bg_amp_min=0, bg_amp_max=2000, have_background=True):
    """(target_coverage, background_coverage) of the unmodified assay, optimize.cpp:72-76."""
    tc, _, _ = scr.move_coverage(pair, 0, [pair[0]], target_threshold, search_multiplier, amp_min, amp_max, use_taq_mama,
                                 which=api.TARGET, bits=False)
    bc = np.float32(0.0)
    if have_background:
        b, _, _ = scr.move_coverage(pair, 0, [pair[0]], bg_threshold, bg_multiplier, bg_amp_min, bg_amp_max, use_taq_mama,
                                    which=api.BACKGROUND, bits=False)
        bc = b[0]
    return np.float32(tc[0]), np.float32(bc)


def optimization_move(scr, pair, move, side, score_threshold=None, degen=1, primer_min=18, primer_max=25, salt=0.05,
                      primer_strand=9.0e-7, tm_min=50.0, tm_max=70.0, max_hairpin=40.0, target_threshold=1.0,
                      search_multiplier=0.9, amp_min=80, amp_max=200, use_taq_mama=False, bg_threshold=0.8,
                      bg_multiplier=0.9, bg_amp_min=0, bg_amp_max=2000, have_background=True, pool=None):
    """optimization_move (optimize.cpp:303-352) for one oligo of `pair` (side 0 = F, 1 = R).  pool is not None =
    opt.use_multiplex (score_threshold then carries the multiplex terms: see multiplex_base_score).

    -> (word, (target_coverage, background_coverage, oligo_overlap)); the empty word (0, 0) and Score()
    if no trial survives, as the reference's move functions return."""
    kw = dict(target_threshold=target_threshold, search_multiplier=search_multiplier, amp_min=amp_min, amp_max=amp_max,
              use_taq_mama=use_taq_mama, bg_threshold=bg_threshold, bg_multiplier=bg_multiplier, bg_amp_min=bg_amp_min,
              bg_amp_max=bg_amp_max, have_background=have_background)
    if score_threshold is None:
        score_threshold = multiplex_base_score(scr, pair, pool, **kw) if pool is not None else base_score(scr, pair, **kw)
    trials = api.host_move_trials(pair[side], move, degen, primer_min, primer_max)
    best_w, best = (0, 0), EMPTY_SCORE
    if not trials:
        return best_w, best
    ok = scr.is_valid(trials, check_homo_dimer=False, salt=salt, primer_strand=primer_strand, tm_min=tm_min, tm_max=tm_max,
                      max_hairpin=max_hairpin, max_dimer=0.0, flags=True)
    live = [t for t, v in zip(trials, ok) if v]
    if not live:
        return best_w, best
    tcov, _, _ = scr.move_coverage(pair, side, live, target_threshold, search_multiplier, amp_min, amp_max, use_taq_mama,
                                   which=api.TARGET, bits=False)
    if have_background:
        bcov, _, _ = scr.move_coverage(pair, side, live, bg_threshold, bg_multiplier, bg_amp_min, bg_amp_max, use_taq_mama,
                                       which=api.BACKGROUND, bits=False)
    else:
        bcov = np.zeros(len(live), np.float32)
    mcov = pov = np.zeros(len(live), np.float32)
    partial = None
    if pool is not None:
        mcov = scr.multiplex_coverage(pair, side, live, bg_threshold, use_taq_mama)
        pov = api.host_pool_overlaps(live, pool)
        partial = _reuse(api.host_pool_overlaps([pair[1 - side]], pool)[0])
    rows = [(t, True, np.float32(tc), np.float32(bc), np.float32(mc), np.float32(po))
            for t, tc, bc, mc, po in zip(live, tcov, bcov, mcov, pov)]
    return _decide_move(rows, score_threshold, move, partial)


def multiplex_base_score(scr, pair, pool, **kw):
    """Score of the unmodified assay with opt.use_multiplex (optimize.cpp:72-97): (tc, bc + multiplex coverage,
    oligo_overlap)."""
    tc, bc = base_score(scr, pair, **kw)
    mc = scr.multiplex_coverage(pair, 0, [pair[0]], kw.get("bg_threshold", 0.8), kw.get("use_taq_mama", False))[0]
    return tc, np.float32(np.float32(bc) + np.float32(mc)), api.host_oligo_overlap(pair, pool)


def score_lt(a, b):
    """Score::operator< (pcramp.h:181-188)."""
    if _accuracy(a) == _accuracy(b):
        return np.float32(a[2]) < np.float32(b[2])
    return _accuracy(a) < _accuracy(b)


def score_eq(a, b):
    """Score::operator== (pcramp.h:199-203)."""
    return _accuracy(a) == _accuracy(b) and np.float32(a[2]) == np.float32(b[2])


DEFAULT_MOVES = (INCREASE_DEGENERACY, DECREASE_DEGENERACY, TRIM5, GROW5, TRIM3, GROW3)      # main.cpp:82-95


def _evaluate_iteration(scr, approx, move_list, degen=1, primer_min=18, primer_max=25, salt=0.05, primer_strand=9.0e-7,
                        tm_min=50.0, tm_max=70.0, max_hairpin=40.0, target_threshold=1.0, search_multiplier=0.9, amp_min=80,
                        amp_max=200, use_taq_mama=False, bg_threshold=0.8, bg_multiplier=0.9, bg_amp_min=0, bg_amp_max=2000,
                        have_background=True, pool=None):
    """Everything one optimize() iteration needs from the device, in five calls instead of three per move: the
    trial words of every move of both oligos, their is_valid flags, their target and background coverage.
    Coverage does not depend on the running score threshold, so the moves can be decided on the host afterwards
    in the reference's order.  With `pool` (use_multiplex) also the multiplex background coverage and the largest
    overlap with a pooled oligo of every trial.  -> {(side, move): [(word, valid, tc, bc, mc, pov), ...]}"""
    per = {}
    flat = {0: [], 1: []}
    for side in (0, 1):
        for mv in move_list:
            tr = api.host_move_trials(approx[side], mv, degen, primer_min, primer_max)
            per[(side, mv)] = (len(flat[side]), len(tr))
            flat[side] += tr
    allw = flat[0] + flat[1]
    ok = [bool(v) for v in scr.is_valid(allw, check_homo_dimer=False, salt=salt, primer_strand=primer_strand, tm_min=tm_min,
                                        tm_max=tm_max, max_hairpin=max_hairpin, max_dimer=0.0, flags=True)] if allw else []
    out = {}
    base = 0
    for side in (0, 1):
        words = flat[side]
        valid = ok[base:base + len(words)]
        base += len(words)
        live = [w for w, v in zip(words, valid) if v]
        tcov = bcov = []
        if live:
            tcov, _, _ = scr.move_coverage(approx, side, live, target_threshold, search_multiplier, amp_min, amp_max, use_taq_mama,
                                           which=api.TARGET, bits=False)
            if have_background:
                bcov, _, _ = scr.move_coverage(approx, side, live, bg_threshold, bg_multiplier, bg_amp_min, bg_amp_max,
                                               use_taq_mama, which=api.BACKGROUND, bits=False)
            else:
                bcov = np.zeros(len(live), np.float32)
        mcov = pov = np.zeros(len(live), np.float32)
        if live and pool is not None:
            mcov = scr.multiplex_coverage(approx, side, live, bg_threshold, use_taq_mama)
            pov = api.host_pool_overlaps(live, pool)
        it = iter(zip(tcov, bcov, mcov, pov))
        rows = []
        for w, v in zip(words, valid):
            if v:
                tc, bc, mc, po = next(it)
                rows.append((w, True, np.float32(tc), np.float32(bc), np.float32(mc), np.float32(po)))
            else:
                rows.append((w, False, None, None, None, None))
        for mv in move_list:
            lo, n = per[(side, mv)]
            out[(side, mv)] = rows[lo:lo + n]
    return out


def _decide_move(rows, score_threshold, move=None, partial=None):
    """One move function's loop over its trials (optimize_pcr.cpp): is_valid gate, coverage-bound shortcut
    (:95-109), Score comparison.  rows: [(word, valid, tc, bc, mc, pov)].  partial is not None = use_multiplex:
    the reuse term of the oligo that is not edited (:27-53); the bound is then `< 0`, the multiplex coverage
    joins the background term and the trial's reuse term is added -- increase_degeneracy alone never resets
    trial_score.oligo_overlap between its trials (:133-145), so there the maximum starts from the previous
    trial's total."""
    best_w, best = (0, 0), EMPTY_SCORE
    carried = np.float32(0.0)
    for w, valid, tc, bc, mc, pov in rows:
        if not valid:
            continue
        bound = np.float32(np.float32(tc) + np.float32(score_threshold[1])) - np.float32(score_threshold[0])
        if (bound < 0.0) if partial is not None else (bound <= 0.0):
            continue
        ov = np.float32(0.0)
        if partial is not None:
            bc = np.float32(np.float32(bc) + np.float32(mc))
            ov = max(np.float32(pov), carried) if move == INCREASE_DEGENERACY else np.float32(pov)
            ov = np.float32(_reuse(ov) + np.float32(partial))
            carried = ov
        trial = (np.float32(tc), np.float32(bc), ov)
        if score_gt(trial, best):
            best, best_w = trial, w
    return best_w, best


def optimize(scr, pair, move_list=DEFAULT_MOVES, **opts):
    """optimize() (optimize.cpp:14-207): greedy local search over both oligos; `pool=[...]` (the assays designed so
    far; may be empty) switches opt.use_multiplex on.

    Per iteration: score of the current assay (collect + update + compute coverage, :61-79), every move of
    every oligo against the running best (`local_score` doubles as the moves' score threshold, :126-130; ties
    go to the lower degeneracy, :133-135), the winner re-centred and installed (:152-154); stops when nothing
    improves, when the score drops, or when an assay repeats (:196-202).  The device is asked once per
    iteration for all trial words of all moves (`_evaluate_iteration`).  -> (best pair, Score)."""
    pool = opts.get("pool")                                            # not None = opt.use_multiplex
    if pool is not None:
        pool = [(tuple(int(x) for x in f), tuple(int(x) for x in r)) for f, r in pool]
        opts = dict(opts, pool=pool)
    cov_kw = {k: opts[k] for k in ("target_threshold", "search_multiplier", "amp_min", "amp_max", "use_taq_mama",
                                   "bg_threshold", "bg_multiplier", "bg_amp_min", "bg_amp_max", "have_background") if k in opts}
    best = (tuple(int(x) for x in pair[0]), tuple(int(x) for x in pair[1]))
    approx = best
    best_score = EMPTY_SCORE
    previous = {approx}
    while True:
        if pool is not None:                                           # optimize.cpp:79-97
            approx_score = multiplex_base_score(scr, approx, pool, **cov_kw)
        else:
            approx_score = base_score(scr, approx, **cov_kw) + (np.float32(0.0),)
        if score_lt(approx_score, best_score):
            break
        best_score, best = approx_score, approx
        evaluated = _evaluate_iteration(scr, approx, move_list, **opts)
        local_seq, local_oligo, local_score, improved = (0, 0), None, approx_score, False
        for side in (0, 1):
            partial = None
            if pool is not None:                                       # the other oligo's reuse term, e.g. optimize_pcr.cpp:27-53
                partial = _reuse(api.host_pool_overlaps([approx[1 - side]], pool)[0])
            for mv in move_list:
                w, sc = _decide_move(evaluated[(side, mv)], local_score, mv, partial)
                if score_gt(sc, local_score) or (score_eq(sc, local_score) and W.word_degeneracy(w) < W.word_degeneracy(local_seq)):
                    local_score, local_seq, local_oligo, improved = sc, w, side, True
        if not improved:
            break
        approx_score = local_score
        centred = W.center_word(local_seq)
        approx = (centred, approx[1]) if local_oligo == 0 else (approx[0], centred)
        if approx in previous:
            break
        previous.add(approx)
    return best, best_score
