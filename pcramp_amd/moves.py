"""Thin binding of the local search behind the C-ABI (include/pcramp_hip.h: pcr_optimize_batch, pcr_optimization_move).

The optimiser itself -- trial generation, is_valid, coverage of every trial, the reference's Score arithmetic, move order
and greedy loop (optimize.cpp:14-207, 303-352; optimize_pcr.cpp:8-989; pcramp.h:158-208) -- lives in
pcramp_amd/csrc/pcr_optimize.inc; this module only packs arguments.  With `pool` (the assays designed so far; may be
empty) opt.use_multiplex is on: the multiplex background keys must have been loaded with Screener.multiplex_load.  The
word DBs of the target and background sets must have been built for the assays passed in (Screener.select_words on
each set), as optimize() is called inside main.cpp's per-iteration DB build.
"""
import ctypes as C

import numpy as np

from . import api
from . import words as W

INCREASE_DEGENERACY, DECREASE_DEGENERACY, TRIM5, TRIM3, GROW5, GROW3 = range(6)
DEFAULT_MOVES = (INCREASE_DEGENERACY, DECREASE_DEGENERACY, TRIM5, GROW5, TRIM3, GROW3)      # main.cpp:82-95
EMPTY_SCORE = (np.float32(-1.0e6), np.float32(1.0e6), np.float32(0.0))                      # Score(), pcramp.h:176-179


def _accuracy(sc):
    return np.float32(sc[0]) - np.float32(sc[1])                            # Score::accuracy, pcramp.h:205


def score_gt(a, b):
    """Score::operator> (pcramp.h:190-197)."""
    if _accuracy(a) == _accuracy(b):
        return np.float32(a[2]) > np.float32(b[2])
    return _accuracy(a) > _accuracy(b)


class OptimizeArgs(C.Structure):
    _fields_ = [("max_degen", C.c_double), ("primer_min", C.c_int32), ("primer_max", C.c_int32), ("thermo", api.ThermoArgs),
                ("target", api.AmplifyArgs), ("background", api.AmplifyArgs), ("have_background", C.c_int32),
                ("use_multiplex", C.c_int32), ("multiplex_threshold", C.c_float), ("n_moves", C.c_int32), ("moves", C.c_int32 * 8)]


def _args(move_list=DEFAULT_MOVES, degen=1, primer_min=18, primer_max=25, salt=0.05, primer_strand=9.0e-7, tm_min=50.0, tm_max=70.0,
          max_hairpin=40.0, target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200, use_taq_mama=False,
          bg_threshold=0.8, bg_multiplier=0.9, bg_amp_min=0, bg_amp_max=2000, have_background=True, pool=None):
    """The Options fields optimize() reads, in the reference's float arithmetic (threshold x multiplier formed in float)."""
    ct = float(np.float32(target_threshold) * np.float32(search_multiplier))
    cb = float(np.float32(bg_threshold) * np.float32(bg_multiplier))
    a = OptimizeArgs()
    a.max_degen, a.primer_min, a.primer_max = float(degen), int(primer_min), int(primer_max)
    a.thermo = api.ThermoArgs(salt, primer_strand, tm_min, tm_max, max_hairpin, 0.0)
    a.target = api.AmplifyArgs(ct, target_threshold, amp_min, amp_max, int(use_taq_mama))
    a.background = api.AmplifyArgs(cb, bg_threshold, bg_amp_min, bg_amp_max, int(use_taq_mama))
    a.have_background, a.use_multiplex, a.multiplex_threshold = int(bool(have_background)), int(pool is not None), bg_threshold
    a.n_moves = len(move_list)
    for k, m in enumerate(move_list):
        a.moves[k] = int(m)
    return a


def _pool_array(pool):
    return W.pairs_array(list(pool)) if pool else np.zeros((1, 4), np.uint64)


def optimize_batch(scr, pairs, move_list=DEFAULT_MOVES, **opts):
    """optimize() for a batch of assays in lockstep (main.cpp:697-887 over the trial assays) -> ([best pair], [Score], [iterations])."""
    L = scr.L
    a = _args(move_list, **opts)
    pool = opts.get("pool")
    pa = W.pairs_array(list(pairs))
    n = pa.shape[0]
    pp = _pool_array(pool)
    best = np.zeros((max(n, 1), 4), np.uint64)
    score = np.zeros((max(n, 1), 3), np.float32)
    iters = np.zeros(max(n, 1), np.uint32)
    fn = L.pcr_optimize_batch
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(OptimizeArgs), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    scr._check(fn(scr.h, pa.ctypes.data, n, C.byref(a), pp.ctypes.data if pool else None, len(pool) if pool else 0,
                  best.ctypes.data, score.ctypes.data, iters.ctypes.data))
    out = [((int(r[0]), int(r[1])), (int(r[2]), int(r[3]))) for r in best[:n]]
    return out, [tuple(np.float32(x) for x in s) for s in score[:n]], [int(i) for i in iters[:n]]


def make_degenerate(scr, pairs, max_dimer=40.0, **opts):
    """make_degenerate (optimize.cpp:356-398 -> PCR::maximize_degeneracy, pcr_assay.cpp:111-230) for a batch of trial assays:
    the top-down start of the local search (--optimize.top-down) -> ([assay], [valid])."""
    a = _args(DEFAULT_MOVES, **opts)
    a.thermo.max_dimer = float(max_dimer)
    pa = W.pairs_array(list(pairs)).copy()
    n = pa.shape[0]
    ok = np.zeros(max(n, 1), np.uint8)
    fn = scr.L.pcr_make_degenerate
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(OptimizeArgs), C.c_void_p]
    scr._check(fn(scr.h, pa.ctypes.data, n, C.byref(a), ok.ctypes.data))
    out = [((int(r[0]), int(r[1])), (int(r[2]), int(r[3]))) for r in pa[:n]]
    return out, [bool(v) for v in ok[:n]]


def optimize(scr, pair, move_list=DEFAULT_MOVES, **opts):
    """optimize() (optimize.cpp:14-207) for one assay -> (best pair, Score)."""
    best, score, _ = optimize_batch(scr, [pair], move_list, **opts)
    return best[0], score[0]


def optimization_move(scr, pair, move, side, score_threshold=None, **opts):
    """optimization_move (optimize.cpp:303-352) for one oligo of `pair` (side 0 = F, 1 = R) -> (word, Score); the empty
    word (0, 0) and Score() if no trial survives.  score_threshold=None: the unmodified assay's own Score."""
    L = scr.L
    a = _args((move,), **opts)
    pool = opts.get("pool")
    pa = W.pairs_array([pair])
    pp = _pool_array(pool)
    word = np.zeros(2, np.uint64)
    score = np.zeros(3, np.float32)
    thr = None if score_threshold is None else np.array([float(x) for x in score_threshold], np.float32)
    fn = L.pcr_optimization_move
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(OptimizeArgs), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p]
    scr._check(fn(scr.h, pa.ctypes.data, int(move), int(side), C.byref(a), pp.ctypes.data if pool else None, len(pool) if pool else 0,
                  thr.ctypes.data if thr is not None else None, word.ctypes.data, score.ctypes.data, None))
    return (int(word[0]), int(word[1])), tuple(np.float32(x) for x in score)


def base_score(scr, pair, **opts):
    """Score of the unmodified assay (optimize.cpp:72-97) -> (target_coverage, background_coverage[, oligo_overlap with pool])."""
    a = _args((TRIM5,), **opts)
    pool = opts.get("pool")
    pa = W.pairs_array([pair])
    pp = _pool_array(pool)
    word = np.zeros(2, np.uint64)
    score = np.zeros(3, np.float32)
    base = np.zeros(3, np.float32)
    fn = scr.L.pcr_optimization_move
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(OptimizeArgs), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p]
    scr._check(fn(scr.h, pa.ctypes.data, TRIM5, 0, C.byref(a), pp.ctypes.data if pool else None, len(pool) if pool else 0,
                  None, word.ctypes.data, score.ctypes.data, base.ctypes.data))
    if pool is not None:
        return np.float32(base[0]), np.float32(base[1]), np.float32(base[2])
    return np.float32(base[0]), np.float32(base[1])
