"""Two rounds of the reference's multiplex design iteration (main.cpp:471-1130, one thread) stitched from the
C-ABI pieces, with the oracle doing the same steps beside it: sample trial assays on the running rand_r
state, build the word DBs for them, optimize() every trial with the multiplex terms, keep the best, find the
targets it detects, collect its amplicons, add them to the multiplex background, split the targets at the
amplicon ends and deactivate the detected targets.  Every intermediate result must agree -- the point is the
state carried from step to step (splits, active flags, key table, pool).  Run with `-m gpu`."""
import random

import numpy as np
import pytest

from pcramp_amd import api, moves, words as W
from testdata import family_targets, mutate, rand_seq

pytestmark = pytest.mark.gpu


def test_two_multiplex_design_rounds(oracle):
    from oracle_lib import random_assays, optimize_multiplex, rand_r, DEFAULT_MOVE_OPTIONS
    rng = random.Random(20261004)
    seqs = family_targets(rng, 3, 6, 900, div=0.05)
    bgs = [mutate(rng, s, 0.15) for s in seqs[::5]] + [rand_seq(rng, 600)]
    o = dict(target_threshold=0.9, search_multiplier=0.9, amp_min=80, amp_max=200, use_taq_mama=0, pack_max_degen=256,
             pack_min_gc=0.0, pack_max_gc=1.0, min_primer=18, optimize_5=1, optimize_3=1)
    mo = dict(DEFAULT_MOVE_OPTIONS, degen=4)
    ts, bs = oracle.session(**o), oracle.session(**o)
    for q in seqs:
        ts.add_target(q, 1.0)
    for q in bgs:
        bs.add_target(q, 1.0)
    d = api.Screener(0)
    try:
        d.load_texts(seqs, [1.0] * len(seqs))
        d.load_texts(bgs, [1.0] * len(bgs), which=api.BACKGROUND)
        texts = list(seqs)
        active = [True] * len(seqs)
        pool, amplicon_texts = [], []
        global_seed = 4242
        thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
        bthr = float(np.float32(0.8) * np.float32(0.9))
        for rnd in range(2):
            local, g2 = api.host_rand_r(global_seed)                   # main.cpp:541-542
            assert (local, g2) == rand_r(oracle, global_seed)
            global_seed = g2
            trials, _, _ = d.random_assays(local, 5)                   # :544-550
            want_trials, _ = random_assays(oracle, ts, local, 5)
            assert trials == want_trials
            every = trials + pool
            assert d.select_words(every, thr, 18, True, True) == ts.select(every)                       # :644-691
            d.select_words(every, bthr, 16, True, True, which=api.BACKGROUND)                           # :579-615
            bs.select(every, threshold=bthr, min_len_override=16)
            ams = oracle.session(use_taq_mama=0)                       # the multiplex background so far
            for a in amplicon_texts:
                ams.add_target(a, 1.0)
            d.multiplex_load(amplicon_texts, 18)
            best, best_score = None, None
            for t in trials:                                           # :700-735
                got, sc = moves.optimize(d, t, pool=pool, target_threshold=o["target_threshold"], search_multiplier=0.9,
                                         amp_min=80, amp_max=200, use_taq_mama=False, **mo)
                want = optimize_multiplex(oracle, ts, bs, ams, pool, t, **{k: v for k, v in mo.items()})
                assert got == want[0] and tuple(float(x) for x in sc) == want[1], (rnd, t)
                if best is None or moves.score_gt(sc, best_score):
                    best, best_score = got, sc
            bits = d.find_target_match([best], o["target_threshold"], 80, 200, False)[0]                # :898
            assert np.array_equal(bits, ts.target_match(best).astype(bool))
            rec = d.collect_amplicons(best, o["target_threshold"], 80, 200)                             # :920
            bo, ao = ts.collect_amplicons(best, o["target_threshold"], 80, 200)
            assert sorted((r["sequence"], r["begin"], r["end"]) for r in rec) == sorted(bo)
            codes = [W.codes_from_text(s) for s in texts]
            new_amps = sorted({W.text_from_codes(codes[r["sequence"]][r["inner_start"]:r["inner_start"] + r["inner_length"]])
                               for r in rec})
            assert [tuple(int(x) for x in W.codes_from_text(a)) for a in new_amps] == sorted(set(ao))
            amplicon_texts += new_amps                                 # :989-1001
            for r in rec:                                              # :1008-1017
                for pos in (r["begin"], (r["begin"] + r["end"]) // 2, r["end"]):
                    d.split(r["sequence"], pos)
                    ts.split(r["sequence"], pos)
                    texts[r["sequence"]] = texts[r["sequence"]][:pos] + "-" + texts[r["sequence"]][pos + 1:]
            for i, hit in enumerate(bits):                             # :1105-1120
                if hit:
                    active[i] = False
                    ts.set_active(i, False)
            d.set_active(active)
            pool.append(best)
        assert len(pool) == 2 and amplicon_texts and not all(active)
    finally:
        d.close()
