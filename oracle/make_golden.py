#!/usr/bin/env python3
"""Generates tests/golden/*.json from the REAL reference (oracle/_ref/libpcramp_ref.so, built
from /root/reference by oracle/Makefile).  Inputs are seeded random data made here; outputs are
what the reference computes for them.  Only inputs + outputs are stored -- no reference source.

    python oracle/make_golden.py          # rewrites tests/golden/

The fixtures travel to the GPU box (which has no /root/reference) and pin both the CPU oracle
(tests/test_oracle_golden.py) and the HIP path (tests/test_gpu_golden.py).
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from oracle_lib import Reference, build_reference  # noqa: E402
from testdata import rand_seq, family_targets, sample_pair, revcomp  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def hexw(w):
    return ["%016x" % w[0], "%016x" % w[1]]


def words_golden(ref):
    rng = random.Random(101)
    cases = []
    for _ in range(400):
        a = ref.word(rand_seq(rng, rng.randint(1, 32), p_degen=0.3))
        for _ in range(rng.randint(0, 8)):
            a = ref.word_shift_right(a) if rng.random() < 0.5 else ref.word_shift_left(a)
        b = ref.word(rand_seq(rng, rng.randint(1, 32), p_degen=0.3))
        for _ in range(rng.randint(0, 6)):
            b = ref.word_shift_right(b)
        c = {"a": hexw(a), "b": hexw(b), "and": ref.word_and(a, b), "size": ref.word_size(a),
             "start": ref.word_start(a), "stop": ref.word_stop(a), "degeneracy": ref.word_degeneracy(a)}
        if ref.word_size(a) > 0:
            c["center"] = hexw(ref.word_center(a))
            c["complement"] = hexw(ref.word_complement(a))
        cases.append(c)
    exp = []
    for _ in range(40):
        w = ref.centered_word(rand_seq(rng, rng.randint(6, 25), p_degen=0.15))
        if ref.word_degeneracy(w) <= 64:
            exp.append({"word": hexw(w), "expansion": [hexw(x) for x in ref.word_expand(w)]})
    taq = [[p1, p2, t1, t2, ref.taq_mama(p1, p2, t1, t2)] for p1 in (1, 2, 4, 8, 15) for p2 in (1, 2, 4, 8, 3)
           for t1 in (1, 2, 4, 8, 0) for t2 in (1, 2, 4, 8, 5)]
    return {"pairs": cases, "expansions": exp, "taq_mama": taq}


def pack_golden(ref):
    rng = random.Random(202)
    cases = []
    specs = [(23, 0, 0, [], 18, 256, 0.0, 1.0), (32, 0, 0, [], 18, 256, 0.0, 1.0), (33, 0, 0, [], 18, 256, 0.0, 1.0),
             (34, 0, 0, [], 18, 256, 0.0, 1.0), (35, 0, 0, [], 18, 256, 0.0, 1.0), (77, 0, 0, [], 18, 256, 0.0, 1.0),
             (120, 0.05, 0.03, [], 16, 256, 0.0, 1.0), (150, 0, 0, [60], 18, 256, 0.0, 1.0),
             (151, 0, 0, [60, 61, 62], 18, 256, 0.0, 1.0), (140, 0.1, 0.05, [50, 90], 18, 16, 0.0, 1.0),
             (160, 0, 0, [], 18, 256, 0.3, 0.7), (161, 0.02, 0, [80], 18, 256, 0.4, 0.6), (90, 0, 0, [0, 1, 89], 10, 256, 0.0, 1.0)]
    for (L, pd, pn, eos, min_len, thr, gmin, gmax) in specs:
        s = list(rand_seq(rng, L, p_degen=pd, p_n=pn))
        for e in eos:
            s[e] = "-"
        s = "".join(s)
        ent = ref.pack(s, 3, thr, gmin, gmax, min_len)
        cases.append({"seq": s, "index": 3, "degen_thr": thr, "min_gc": gmin, "max_gc": gmax, "min_len": min_len,
                      "entries": [["%016x" % a, "%016x" % b, loc, idx, st] for (a, b, loc, idx, st) in ent]})
    return {"cases": cases}


def screen_golden(ref):
    """select_words + find_target_match + compute_coverage on small families."""
    cases = []
    for ci, opts in enumerate([dict(), dict(target_threshold=0.9), dict(target_threshold=0.8, use_taq_mama=1),
                               dict(target_threshold=0.9, optimize_5=1, optimize_3=1),
                               dict(target_threshold=0.85, amp_min=60, amp_max=300)]):
        rng = random.Random(303 + ci)
        seqs = family_targets(rng, 2, 5, 420, div=0.04)
        seqs.append(rand_seq(rng, 40))
        seqs.append(rand_seq(rng, 211, p_degen=0.03, p_n=0.02))
        s = list(seqs[0]); s[200] = "-"; seqs.append("".join(s))
        weights = [1.0 + 0.37 * (i % 5) for i in range(len(seqs))]
        pairs_txt = []
        while len(pairs_txt) < 8:
            p = sample_pair(rng, rng.choice(seqs[:10]))
            if p:
                pairs_txt.append(p)
        s0 = seqs[1]
        pairs_txt.append((s0[0:20], revcomp(s0[100:120])))
        pairs_txt.append((s0[len(s0) - 150:len(s0) - 130], revcomp(s0[len(s0) - 21:])))
        pairs = [(ref.centered_word(f), ref.centered_word(r)) for f, r in pairs_txt]
        ses = ref.session(**opts)
        for q, w in zip(seqs, weights):
            ses.add_target(q, w)
        inactive = [3] if ci == 1 else []
        splits = [[2, 100]] if ci == 1 else []
        for i in inactive:
            ses.set_active(i, False)
        for i, pos in splits:
            ses.split(i, pos)
        n = ses.select(pairs)
        db = ses.db_entries()
        bits = [ses.target_match(p).tolist() for p in pairs]
        cov = [float(ses.target_coverage(p)) for p in pairs]
        cases.append({"options": ses.opts, "seqs": seqs, "weights": weights, "inactive": inactive, "splits": splits,
                      "pairs": [hexw(f) + hexw(r) for f, r in pairs], "n_entries": n,
                      "db": [["%016x" % a, "%016x" % b, loc, idx, st] for (a, b, loc, idx, st) in db],
                      "bits": bits, "coverage": cov})
    return {"cases": cases}


def sw_golden(ref):
    """8-lane SeqOverlap calls (seq_overlap.cpp:347) on word pairs; lanes with score 0 keep only the score."""
    rng = random.Random(404)
    from testdata import mutate
    cases = []
    for it in range(40):
        qs, ts = [], []
        for lane in range(8):
            qtxt = rand_seq(rng, rng.randint(6, 32), p_degen=0.05)
            m = rng.random()
            if m < 0.55:
                core = mutate(rng, qtxt, 0.12)
                if rng.random() < 0.4 and len(core) > 6:
                    k = rng.randrange(2, len(core) - 2)
                    core = core[:k] + (rand_seq(rng, 1) if rng.random() < 0.5 else "") + core[k + (rng.random() < 0.5):]
                core = core[:32]
                left = rng.randint(0, 32 - len(core))
                ttxt = (rand_seq(rng, left) + core + rand_seq(rng, 32))[:rng.randint(min(32, left + len(core)), 32)]
            elif m < 0.65:
                qtxt, ttxt = "A" * rng.randint(10, 20), "A" * rng.randint(10, 32)
            else:
                ttxt = rand_seq(rng, rng.randint(6, 32), p_degen=0.05)
            q, t = ref.word(qtxt), ref.word(ttxt)
            for _ in range(rng.randint(0, 32 - len(qtxt))):
                q = ref.word_shift_right(q)
            for _ in range(rng.randint(0, 32 - len(ttxt))):
                t = ref.word_shift_right(t)
            qs.append(q); ts.append(t)
        res = ref.sw_align_words8(qs, ts)
        for lane in range(8):
            r = res[lane]
            cases.append({"q": hexw(qs[lane]), "t": hexw(ts[lane]), "score": r[0],
                          "rest": list(r[1:]) if r[0] > 0 else None})
    return {"cases": cases}


def thermo_golden(ref):
    """NucCruc values (deterministic harness form, see ref_harness.cpp: rings pre-filled with E)."""
    rng = random.Random(505)
    from testdata import mutate
    oligos = []
    for it in range(400):
        n = rng.randint(12, 32)
        m = rng.random()
        if m < 0.6:
            s = rand_seq(rng, n)
        elif m < 0.8:
            stem = rand_seq(rng, rng.randint(4, 8))
            s = (rand_seq(rng, rng.randint(0, 4)) + stem + rand_seq(rng, rng.randint(3, 7)) + mutate(rng, revcomp(stem), 0.1) + rand_seq(rng, 8))[:32]
            if len(s) < 12:
                s += rand_seq(rng, 12)
        else:
            s = (rand_seq(rng, rng.randint(1, 4)) * 32)[:n]
        salt = rng.choice([0.05, 0.05, 0.01, 0.2])
        strand = rng.choice([9e-7, 9e-7, 4.5e-7, 2e-8])
        oligos.append({"seq": s, "salt": salt, "strand": strand, "out": [float(x) for x in ref.thermo_full(s, salt, strand)]})
    hetero = []
    for it in range(200):
        a = rand_seq(rng, rng.randint(12, 30))
        if rng.random() < 0.5:
            b = rand_seq(rng, rng.randint(12, 30))
        else:
            core = mutate(rng, revcomp(a[rng.randint(0, 5):rng.randint(len(a) - 5, len(a))]), 0.15)
            b = (rand_seq(rng, rng.randint(0, 4)) + core + rand_seq(rng, rng.randint(0, 4)))[:32]
        sa, sb = rng.choice([(9e-7, 9e-7), (9e-7, 4.5e-7), (1e-7, 9e-7)])
        hetero.append({"a": a, "b": b, "sa": sa, "sb": sb, "out": [float(x) for x in ref.heterodimer_full(a, b, 0.05, sa, sb)]})
    filt = []
    for it in range(120):
        w = ref.centered_word(rand_seq(rng, rng.randint(18, 25), p_degen=0.06 if it % 3 == 0 else 0.0))
        if ref.word_degeneracy(w) > 16:
            continue
        kw = dict(tm_min=rng.choice([45.0, 50.0]), tm_max=rng.choice([65.0, 75.0]), max_hairpin=rng.choice([30.0, 40.0]),
                  max_dimer=rng.choice([30.0, 40.0]), check_homo_dimer=bool(it & 1))
        filt.append({"word": hexw(w), "kw": kw, "valid": ref.is_valid(w, **kw)})
    dimers = []
    for it in range(60):
        f = ref.centered_word(rand_seq(rng, rng.randint(18, 25), p_degen=0.04))
        r = ref.centered_word(rand_seq(rng, rng.randint(18, 25), p_degen=0.04))
        if ref.word_degeneracy(f) * ref.word_degeneracy(r) > 64:
            continue
        f2 = ref.centered_word(rand_seq(rng, rng.randint(18, 25)))
        r2 = ref.centered_word(revcomp(rand_seq(rng, 6) + "ACGTTGCAAT" + rand_seq(rng, 5)))
        dimers.append({"pair": hexw(f) + hexw(r), "max_dimer_tm": float(ref.max_dimer_tm((f, r))), "other": hexw(f2) + hexw(r2),
                       "compatible": {str(md): ref.multiplex_compatible((f, r), (f2, r2), max_dimer=md) for md in (10.0, 25.0, 40.0)}})
    return {"oligos": oligos, "hetero": hetero, "is_valid": filt, "dimers": dimers}


def moves_golden(ref):
    """The six local-search moves of optimize_pcr.cpp run by the reference's optimization_move() for both
    oligos of a few assays (targets + backgrounds, non-multiplex): returned trial word and Score."""
    from oracle_lib import optimization_move, optimize, DEFAULT_MOVE_OPTIONS
    from testdata import mutate
    cases = []
    for ci, case in enumerate([dict(), dict(degen=16), dict(target_threshold=0.9, degen=4, use_taq_mama=1)]):
        case = dict(case)
        sess = {k: case.pop(k) for k in ("target_threshold", "use_taq_mama") if k in case}
        rng = random.Random(909 + ci)
        seqs = family_targets(rng, 2, 6, 420, div=0.06)
        bgs = [mutate(rng, q, 0.12) for q in seqs[::3]] + [rand_seq(rng, 300)]
        weights = [1.0 + 0.3 * (i % 4) for i in range(len(seqs))]
        pairs_txt = []
        while len(pairs_txt) < 4:
            p = sample_pair(rng, rng.choice(seqs))
            if p:
                pairs_txt.append(p)
        f, r = pairs_txt[0]
        f = list(f); f[rng.randrange(3, len(f) - 3)] = rng.choice("RYKM")
        pairs_txt.append(("".join(f), r))
        pairs = [(ref.centered_word(a), ref.centered_word(b)) for a, b in pairs_txt]
        # two damaged assays: the full optimize() loop has something to repair
        for a0, b0 in pairs_txt[1:3]:
            pairs_txt.append((mutate(rng, a0, 0.1), mutate(rng, b0, 0.1)))
        pairs = [(ref.centered_word(a), ref.centered_word(b)) for a, b in pairs_txt]
        sess = dict(sess, optimize_5=1, optimize_3=1)
        ts, bs = ref.session(**sess), ref.session(**sess)
        for q, wt in zip(seqs, weights):
            ts.add_target(q, wt)
        for q in bgs:
            bs.add_target(q, 1.0)
        ts.select(pairs)
        bs.select(pairs, threshold=0.8 * 0.9, min_len_override=16)
        opt_out = []
        for pi, p in enumerate(pairs):
            bp, sc = optimize(ref, ts, bs, p, **case)
            opt_out.append([pi, hexw(bp[0]) + hexw(bp[1]), list(sc)])
        out = []
        for pi, p in enumerate(pairs):
            for side in (0, 1):
                for move in range(6):
                    w, sc, base = optimization_move(ref, ts, bs, p, move, side, **case)
                    out.append([pi, side, move, hexw(w), list(sc), list(base)])
        mo = dict(DEFAULT_MOVE_OPTIONS); mo.update(case)
        cases.append({"options": ts.opts, "move_options": mo, "seqs": seqs, "weights": weights, "backgrounds": bgs,
                      "bg_select_threshold": 0.8 * 0.9, "bg_min_len": 16,
                      "pairs": [hexw(a) + hexw(b) for a, b in pairs], "moves": out, "optimize": opt_out})
    return {"cases": cases}


def degenerate_golden(ref):
    """make_degenerate (optimize.cpp:356-398 -> PCR::maximize_degeneracy: the top-down start of the local search) run by the
    reference itself: resulting assay and return value, incl. cases whose max_dimer is low enough for the greedy heterodimer
    reduction to run (and once to fail)."""
    from oracle_lib import make_degenerate, DEFAULT_MOVE_OPTIONS
    cases = []
    for ci, case in enumerate([dict(degen=16), dict(degen=64, target_threshold=0.9, tm_min=40.0, tm_max=80.0, max_hairpin=50.0),
                               dict(degen=8, target_threshold=0.85, use_taq_mama=1, tm_min=40.0, tm_max=80.0),
                               dict(degen=64, max_dimer=20.0, tm_min=-100.0, tm_max=200.0, max_hairpin=500.0),
                               dict(degen=64, max_dimer=12.0, tm_min=-100.0, tm_max=200.0, max_hairpin=500.0, seed=9103)]):
        case = dict(case)
        sess = {k: case.pop(k) for k in ("target_threshold", "use_taq_mama") if k in case}
        rng = random.Random(case.pop("seed", 9100 + len(case) + 7 * len(sess)))
        max_dimer = case.pop("max_dimer", 40.0)
        seqs = family_targets(rng, 3, 12, 500, div=0.08)
        weights = [1.0 + 0.3 * (i % 4) for i in range(len(seqs))]
        pairs_txt = []
        while len(pairs_txt) < 14:
            p = sample_pair(rng, rng.choice(seqs))
            if p:
                pairs_txt.append(p)
        pairs = [(ref.centered_word(a), ref.centered_word(b)) for a, b in pairs_txt]
        ts = ref.session(**sess)
        for q, wt in zip(seqs, weights):
            ts.add_target(q, wt)
        ts.select(pairs)
        out = []
        for p in pairs:
            got, ok = make_degenerate(ref, ts, p, max_dimer=max_dimer, **case)
            out.append([hexw(got[0]) + hexw(got[1]), int(ok)])
        mo = dict(DEFAULT_MOVE_OPTIONS); mo.update(case)
        cases.append({"options": ts.opts, "move_options": mo, "max_dimer": max_dimer, "seqs": seqs, "weights": weights,
                      "pairs": [hexw(a) + hexw(b) for a, b in pairs], "degenerate": out})
    return {"cases": cases}


def sampler_inputs(ci):
    """Targets of the sampler cases: families with an IUPAC stretch, an EOS split, an inactive record and, for
    case 3, records barely longer than the amplicon."""
    rng = random.Random(3100 + ci)
    if ci == 3:
        seqs = [rand_seq(rng, n) for n in (41, 44, 47, 60, 75)]
        return seqs, [True] * len(seqs), []
    seqs = family_targets(rng, 3, 5, 700, div=0.05) + [rand_seq(rng, 333)]
    q = list(seqs[2])
    for k in range(300, 420, 9):
        q[k] = rng.choice("RYKMSWN")
    seqs[2] = "".join(q)
    active = [i != 4 for i in range(len(seqs))]
    return seqs, active, [(1, 350), (7, 100)]


SAMPLER_CASES = [dict(), dict(max_degen=4.0, tm_min=45.0, tm_max=75.0), dict(tm_min=58.0, tm_max=60.0, max_hairpin=25.0, max_dimer=20.0),
                 dict(primer_min=18, primer_max=22, amp_min=38, amp_max=60, tm_min=40.0, tm_max=80.0)]


def sampler_golden(ref):
    """PCR::random_assay as the one-thread sampling loop of main.cpp:544-550 runs it: n trials per seed on one
    running rand_r state; the assays and the state afterwards.  Plus raw rand_r values."""
    from oracle_lib import random_assays, rand_r, DEFAULT_SAMPLER_OPTIONS
    stream = []
    for seed in (0, 1, 42, 0x7fffffff, 0xdeadbeef, 0xffffffff):
        s, vals = seed, []
        for _ in range(6):
            v, s = rand_r(ref, s)
            vals.append(v)
        stream.append([seed, vals, s])
    cases = []
    for ci, case in enumerate(SAMPLER_CASES):
        seqs, active, splits = sampler_inputs(ci)
        sess = ref.session()
        for q, a in zip(seqs, active):
            sess.add_target(q, 1.0, a)
        for i, pos in splits:
            sess.split(i, pos)
        runs = []
        for seed in (1, 2, 3, 77, 4242, 0x9e3779b9):
            pairs, after = random_assays(ref, sess, seed, 6, **case)
            runs.append([seed, [hexw(f) + hexw(r) for f, r in pairs], after])
        so = dict(DEFAULT_SAMPLER_OPTIONS); so.update(case)
        cases.append({"sampler_options": so, "seqs": seqs, "active": active, "splits": splits, "runs": runs})
    return {"rand_r": stream, "cases": cases}


def overlap_golden(ref):
    """Word::max_overlap and PCR::compute_oligo_overlap (the oligo-reuse term of the multiplex Score)."""
    rng = random.Random(5150)
    words = []
    for _ in range(60):
        w = ref.word(rand_seq(rng, rng.randint(15, 32), p_degen=0.15))
        for _ in range(rng.randint(0, 5)):
            w = ref.word_shift_right(w)
        words.append(w)
    base = rand_seq(rng, 30)
    for k in range(8):                                                 # related words: shared stretches, identical copies
        words.append(ref.centered_word(base[k:k + 18 + k]))
    words.append(words[-1])
    pairs = [[i, j, ref.max_overlap(words[i], words[j])] for i in range(len(words)) for j in range(0, len(words), 3)]
    assays = []
    for _ in range(40):
        a = (rng.choice(words), rng.choice(words))
        pool = [(rng.choice(words), rng.choice(words)) for _ in range(rng.randint(0, 6))]
        assays.append({"assay": hexw(a[0]) + hexw(a[1]), "pool": [hexw(f) + hexw(r) for f, r in pool],
                       "overlap": ref.oligo_overlap(a, pool)})
    return {"words": [hexw(w) for w in words], "max_overlap": pairs, "oligo_overlap": assays}


def multiplex_golden(ref):
    """Multiplex background coverage (pcr_assay.cpp:71-102, :304-336) over amplicon-like sequences packed as
    main.cpp:989-1001 packs accepted amplicons: key count and the coverage of every move variant."""
    from testdata import multiplex_case, move_variants
    from pcramp_amd import words as W
    cases = []
    for taq in (0, 1):
        rng = random.Random(7300 + taq)
        amps, pairs = multiplex_case(rng, W, ref, n_amp=8)
        sess = ref.session(min_primer=18)
        for a in amps:
            sess.add_target(a, 1.0)
        rows = []
        for pi, p in enumerate(pairs[:6]):
            for side in (0, 1):
                var = [p[side]]
                for kind in ("inc", "dec", "trim5", "trim3", "grow5", "grow3"):
                    var += move_variants(W, p[side], kind)
                for thr in (0.8, 0.65):
                    cov, nk = sess.multiplex_coverage(p, side, var, thr, taq)
                    rows.append([pi, side, thr, [hexw(v) for v in var], [float(c) for c in cov], nk])
        cases.append({"use_taq_mama": taq, "amplicons": amps, "min_primer": 18, "pairs": [hexw(f) + hexw(r) for f, r in pairs[:6]],
                      "rows": rows})
    return {"cases": cases}


def multiplex_optimize_golden(ref):
    """optimize() with opt.use_multiplex run by the reference: final assay and Score for candidate assays against
    targets + backgrounds + the amplicons / pool of the assays designed so far."""
    from oracle_lib import optimize_multiplex, DEFAULT_MOVE_OPTIONS
    from testdata import multiplex_design_case
    cases = []
    for ci, case in enumerate([dict(), dict(degen=8), dict(degen=16, target_threshold=0.9, use_taq_mama=1)]):
        case = dict(case)
        sess = {k: case.pop(k) for k in ("target_threshold", "use_taq_mama") if k in case}
        rng = random.Random(8800 + ci)
        seqs, bgs, amps, pool, cands = multiplex_design_case(rng, ref)
        weights = [1.0 + 0.25 * (i % 3) for i in range(len(seqs))]
        sopt = dict(sess, optimize_5=1, optimize_3=1)
        ts, bs, ams = ref.session(**sopt), ref.session(**sopt), ref.session(**sess)
        for q, wt in zip(seqs, weights):
            ts.add_target(q, wt)
        for q in bgs:
            bs.add_target(q, 1.0)
        for q in amps:
            ams.add_target(q, 1.0)
        allp = cands + pool
        ts.select(allp)
        bs.select(allp, threshold=0.8 * 0.9, min_len_override=16)
        out = []
        for pi, p in enumerate(cands):
            for use_pool in (1, 0):
                bp, sc = optimize_multiplex(ref, ts, bs, ams, pool if use_pool else [], p, **case)
                out.append([pi, use_pool, hexw(bp[0]) + hexw(bp[1]), list(sc)])
        mo = dict(DEFAULT_MOVE_OPTIONS); mo.update(case)
        cases.append({"options": ts.opts, "move_options": mo, "seqs": seqs, "weights": weights, "backgrounds": bgs,
                      "amplicons": amps, "bg_select_threshold": 0.8 * 0.9, "bg_min_len": 16,
                      "pool": [hexw(a) + hexw(b) for a, b in pool], "candidates": [hexw(a) + hexw(b) for a, b in cands],
                      "optimize": out})
    return {"cases": cases}


def amplicons_golden(ref):
    """PCR::collect_unique_amplicons: AmpliconBounds and unique amplicon stretches (as nibbles) per assay."""
    cases = []
    for ci, opts in enumerate([dict(), dict(target_threshold=0.9), dict(target_threshold=0.85, amp_min=40, amp_max=400)]):
        o = dict(target_threshold=1.0, amp_min=80, amp_max=200)
        o.update(opts)
        rng = random.Random(1618 + ci)
        seqs = family_targets(rng, 3, 6, 600, div=0.05)
        q = list(seqs[2])
        for k in range(100, 500, 23):
            q[k] = rng.choice("RYKMSWN")
        seqs[2] = "".join(q)
        pairs_txt = []
        while len(pairs_txt) < 6:
            p = sample_pair(rng, rng.choice(seqs))
            if p:
                pairs_txt.append(p)
        pairs = [(ref.centered_word(f), ref.centered_word(r)) for f, r in pairs_txt]
        sess = ref.session(**o)
        for q in seqs:
            sess.add_target(q, 1.0)
        splits = [(1, 350), (5, 120), (9, 410)]
        for i, pos in splits:
            sess.split(i, pos)
        sess.set_active(4, False)
        sess.select(pairs)
        rows = []
        for p in pairs:
            b, a = sess.collect_amplicons(p, o["target_threshold"], o["amp_min"], o["amp_max"])
            rows.append({"bounds": [list(x) for x in b], "amplicons": ["".join("%x" % v for v in t) for t in a]})
        cases.append({"options": sess.opts, "seqs": seqs, "splits": splits, "inactive": [4],
                      "pairs": [hexw(f) + hexw(r) for f, r in pairs], "rows": rows})
    return {"cases": cases}


def background_golden(ref):
    """PCR::find_background_match run by the reference on a small background set, with the candidate amplicon
    count of every pair: below, equal to and above the number of sequences (background_match.cpp:122 drops the
    odd-indexed amplicon of a couple once its index reaches num_seq), TaqMAMA on and off.  Pairs whose amplicon
    count is odd and below num_seq are left out: the reference reads past its deque there (undefined)."""
    import numpy as np
    from testdata import mutate
    PAD = "N" * 40
    cases = []
    for ci, kw in enumerate([dict(bg_threshold=0.8, bg_multiplier=0.9, use_taq_mama=0),
                             dict(bg_threshold=0.75, bg_multiplier=0.9, use_taq_mama=1),
                             dict(bg_threshold=0.7, bg_multiplier=0.9, use_taq_mama=0, amp_max=400),
                             dict(bg_threshold=0.45, bg_multiplier=0.9, use_taq_mama=1),
                             dict(bg_threshold=0.4, bg_multiplier=0.8, use_taq_mama=0, amp_max=400),
                             dict(bg_threshold=0.35, bg_multiplier=1.0, use_taq_mama=1),
                             dict(bg_threshold=0.5, bg_multiplier=0.9, use_taq_mama=0, n_extra="equal"),
                             dict(bg_threshold=0.5, bg_multiplier=0.9, use_taq_mama=1, n_extra="equal+1"),
                             dict(bg_threshold=0.5, bg_multiplier=0.9, use_taq_mama=1, n_extra="equal-1")]):
        kw = dict(kw)
        n_extra = kw.pop("n_extra", None)
        rng = random.Random(1200 + ci)
        roots = family_targets(rng, 3, 1, 700, div=0.0)
        seqs = [mutate(rng, r, 0.05) for r in roots for _ in range(5)]
        pairs_txt = []
        while len(pairs_txt) < 16:
            p = sample_pair(rng, rng.choice(roots))
            if p:
                pairs_txt.append(p)
        pairs = [(ref.centered_word(f), ref.centered_word(r)) for f, r in pairs_txt]
        thr = float(np.float32(kw["bg_threshold"]) * np.float32(kw["bg_multiplier"]))
        min_len = int(18 * 0.9)

        def run(seqs, n_pad=0):
            ses = ref.session()
            for q in seqs + [PAD] * n_pad:
                ses.add_target(q, 1.0)
            n = ses.select(pairs, threshold=thr, min_len_override=min_len)
            rows = []
            for pi, p in enumerate(pairs):
                b, n_amp = ses.background_match(p, **kw)
                rows.append(None if b is None else [pi, int(n_amp), [int(i) for i in np.nonzero(b)[0]]])
            return n, rows
        n_entries, rows = run(seqs)
        n_pad = 0
        if n_extra:
            # pad the set with records that yield no word at all (all-N windows exceed pack_max_degen) until the
            # sequence count meets an even amplicon count of a pair with hits: index == num_seq exactly on the boundary
            cand = sorted(r[1] for r in rows if r is not None and r[1] > len(seqs) and r[1] % 2 == 0 and r[2]) or \
                sorted(r[1] for r in rows if r is not None and r[1] > len(seqs) and r[1] % 2 == 0)
            assert cand, "no pair with more amplicons than sequences"
            want = cand[0] + {"equal": 0, "equal+1": 1, "equal-1": -1}[n_extra]
            n_pad = want - len(seqs)
            n_entries, rows = run(seqs, n_pad)
        rows = [r for r in rows if r is not None]
        n_seq = len(seqs) + n_pad
        regimes = {"below": sum(r[1] < n_seq for r in rows), "equal": sum(r[1] == n_seq for r in rows), "above": sum(r[1] > n_seq for r in rows)}
        cases.append({"kw": kw, "seqs": seqs, "n_pad": n_pad, "pad": PAD, "select_threshold": thr, "min_len": min_len, "n_entries": n_entries,
                      "pairs": [hexw(f) + hexw(r) for f, r in pairs], "rows": rows, "regimes": regimes})
    return {"cases": cases}


def multiplex_match_golden(ref):
    """PCR::find_multiplex_background_match: F, (F), R, (R) against whole amplicon sequences."""
    from testdata import mutate
    cases = []
    for taq in (0, 1):
        rng = random.Random(1300 + taq)
        base = rand_seq(rng, 500)
        seqs = [base[40:220], mutate(rng, base[40:220], 0.1), rand_seq(rng, 150), base[60:210], rand_seq(rng, 33),
                mutate(rng, base[30:230], 0.2), revcomp(base[40:220]), rand_seq(rng, 5), base[50:70], base[250:470],
                mutate(rng, base[250:470], 0.07), rand_seq(rng, 64, p_degen=0.1)]
        pairs_txt = [(base[50:70], revcomp(base[180:202])), (base[100:125], revcomp(base[300:318])),
                     (base[260:280], revcomp(base[400:424])), (mutate(rng, base[262:284], 0.1), revcomp(base[395:415]))]
        f = list(pairs_txt[2][0]); f[7] = "R"; f[12] = "Y"
        pairs_txt.append(("".join(f), pairs_txt[2][1]))
        pairs = [(ref.centered_word(a), ref.centered_word(b)) for a, b in pairs_txt]
        ses = ref.session()
        for q in seqs:
            ses.add_target(q, 1.0)
        rows = []
        for thr in (0.6, 0.8, 0.95):
            for pi, p in enumerate(pairs):
                rows.append([pi, thr, [int(x) for x in ses.multiplex_match(p, thr, taq)]])
        cases.append({"use_taq_mama": taq, "seqs": seqs, "pairs": [hexw(a) + hexw(b) for a, b in pairs], "rows": rows})
    return {"cases": cases}


def writers_golden(ref):
    """Scope row f-4.  (a) PCR::write / write_json in their four forms for assays and pools with reused oligos
    (through the harness); (b) whole output files of the reference PROGRAM (oracle/_ref/pcramp = main.cpp linked with
    the same objects), one rank, one thread, fixed seed, on toy FASTA files: text and JSON."""
    import subprocess
    import tempfile
    rng = random.Random(4711)
    oligos = []
    for it in range(40):
        f = ref.centered_word(rand_seq(rng, rng.randint(18, 25), p_degen=0.08 if it % 2 else 0.0))
        r = ref.centered_word(rand_seq(rng, rng.randint(18, 25), p_degen=0.08 if it % 3 == 0 else 0.0))
        pool = [(ref.centered_word(rand_seq(rng, rng.randint(18, 25))), ref.centered_word(rand_seq(rng, rng.randint(18, 25))))
                for _ in range(rng.randint(0, 4))]
        if it % 4 == 1:
            pool.append((r, pool[0][0] if pool else f))          # the reverse oligo is reused
        if it % 4 == 2:
            pool.insert(0, (f, f))                               # the forward oligo is reused
        if it % 8 == 3:
            pool.append((ref.word_shift_right(f), r))            # same oligos, one of them shifted inside the word
        forms = {}
        for json_ in (0, 1):
            for with_pool in (0, 1):
                forms["%d%d" % (json_, with_pool)] = ref.format_oligos((f, r), pool, json_, with_pool).decode("latin-1")
        oligos.append({"assay": hexw(f) + hexw(r), "pool": [hexw(a) + hexw(b) for a, b in pool], "forms": forms})

    from testdata import mutate
    exe = os.path.join(ROOT, "oracle", "_ref", "pcramp")
    runs = []
    with tempfile.TemporaryDirectory() as tmp:
        libdir = os.path.join(tmp, "lib")
        os.makedirs(libdir)
        for so in ("libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0"):      # not the whole conda lib dir: its libstdc++ is older
            os.symlink(os.path.join("/opt/conda/lib", so), os.path.join(libdir, so))
        env = dict(os.environ, LD_LIBRARY_PATH=libdir, OMP_NUM_THREADS="1")
        specs = [dict(n_fam=3, per=4, L=600, n_bg=2, args=["--count", "3", "--trial", "40", "--seed", "42"]),
                 dict(n_fam=3, per=4, L=600, n_bg=2, args=["--count", "3", "--trial", "40", "--seed", "42", "--o.json"]),
                 dict(n_fam=1, per=4, L=500, n_bg=0, args=["--count", "3", "--trial", "60", "--seed", "7", "-d", "8"]),
                 dict(n_fam=1, per=4, L=500, n_bg=0, args=["--count", "3", "--trial", "60", "--seed", "7", "-d", "8", "--o.json"]),
                 dict(n_fam=2, per=3, L=451, n_bg=3, args=["--count", "6", "--trial", "30", "--seed", "99", "--target.threshold", "0.9"]),
                 dict(n_fam=2, per=3, L=451, n_bg=3, args=["--count", "6", "--trial", "30", "--seed", "99", "--target.threshold", "0.9", "--o.json"])]
        for si, sp in enumerate(specs):
            r2 = random.Random(6000 + si // 2)
            roots = [rand_seq(r2, sp["L"] + 7 * k) for k in range(sp["n_fam"])]
            targets = [(">target_%d family %d" % (k * sp["per"] + j, k), mutate(r2, roots[k], 0.03)) for k in range(sp["n_fam"]) for j in range(sp["per"])]
            bgs = [(">bg_%d" % i, mutate(r2, roots[i % len(roots)], 0.12)) for i in range(sp["n_bg"])]
            with open(os.path.join(tmp, "t.fa"), "w") as f:
                f.write("".join("%s\n%s\n" % (d, q) for d, q in targets))
            argv = ["pcramp", "-t", "t.fa", "-o", "out.txt", "--thread", "1"] + sp["args"]
            if bgs:
                with open(os.path.join(tmp, "b.fa"), "w") as f:
                    f.write("".join("%s\n%s\n" % (d, q) for d, q in bgs))
                argv += ["-b", "b.fa"]
            subprocess.check_call([exe] + argv[1:], cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            out = open(os.path.join(tmp, "out.txt"), "rb").read().decode("latin-1")
            # argv[0] as the program saw it
            out = out.replace(exe, "pcramp")
            runs.append({"argv": argv, "seed": int(sp["args"][sp["args"].index("--seed") + 1]), "json": int("--o.json" in sp["args"]),
                         "targets": [[d, len(q)] for d, q in targets], "backgrounds": [[d, len(q)] for d, q in bgs], "output": out})
    return {"oligos": oligos, "runs": runs}


def program_golden(ref):
    """Scope row f-7: more whole runs of the reference PROGRAM (as writers_golden's, oracle/_ref/pcramp, one rank, one thread) with the
    switches those do not use: the top-down start, the 5' / 3' moves, TaqMAMA, a relaxed background gate on close backgrounds.
    tests/test_gpu_design_program.py replays the command lines through pcr_design."""
    import subprocess
    import tempfile
    from testdata import mutate
    exe = os.path.join(ROOT, "oracle", "_ref", "pcramp")
    specs = [dict(n_fam=2, per=4, L=520, n_bg=2, div=0.04, bg_div=0.12, args=["--count", "4", "--trial", "30", "--seed", "11", "-d", "8", "--optimize.top-down"]),
             dict(n_fam=2, per=4, L=520, n_bg=2, div=0.04, bg_div=0.12, args=["--count", "4", "--trial", "30", "--seed", "11", "-d", "8", "--optimize.top-down", "--o.json"]),
             dict(n_fam=2, per=3, L=480, n_bg=2, div=0.05, bg_div=0.10, args=["--count", "4", "--trial", "25", "--seed", "5", "-d", "4", "--optimize.5", "--optimize.3",
                                                                             "--target.threshold", "0.9"]),
             dict(n_fam=3, per=3, L=450, n_bg=3, div=0.03, bg_div=0.04, args=["--count", "5", "--trial", "30", "--seed", "23", "--background.cover", "2",
                                                                             "--background.threshold", "0.75"]),
             dict(n_fam=2, per=3, L=450, n_bg=1, div=0.05, bg_div=0.08, args=["--count", "3", "--trial", "25", "--seed", "3", "-d", "2", "--primer.taq-mama",
                                                                             "--target.threshold", "0.85", "--optimize.3"]),
             dict(n_fam=1, per=5, L=700, n_bg=0, div=0.02, bg_div=0.0, args=["--count", "9", "--trial", "20", "--seed", "77"]),
             # backgrounds that ARE the targets' roots and a generous background gate (the reference still reports no cross-reaction: see
             # tests/test_writers.py's note on the B- branches)
             dict(n_fam=2, per=4, L=500, n_bg=2, div=0.03, bg_div=0.0, args=["--count", "4", "--trial", "30", "--seed", "19", "--background.cover", "4"]),
             dict(n_fam=2, per=4, L=500, n_bg=2, div=0.03, bg_div=0.0, args=["--count", "4", "--trial", "30", "--seed", "19", "--background.cover", "4", "--o.json"])]
    # ... and sixteen runs with inputs and switches drawn at random (seeded): the point is the bookkeeping between iterations under
    # option combinations nobody picked by hand
    rs = random.Random(31337)
    for k in range(16):
        args = ["--count", str(rs.randint(3, 7)), "--trial", str(rs.randint(15, 40)), "--seed", str(rs.randint(1, 10 ** 6))]
        if rs.random() < 0.6:
            args += ["-d", str(rs.choice([2, 4, 8, 16]))]
            if rs.random() < 0.4:
                args += ["--optimize.top-down"]
        if rs.random() < 0.35:
            args += ["--optimize.5"]
        if rs.random() < 0.35:
            args += ["--optimize.3"]
        if rs.random() < 0.3:
            args += ["--primer.taq-mama"]
        if rs.random() < 0.6:
            args += ["--target.threshold", rs.choice(["0.85", "0.9", "0.95"])]
        if rs.random() < 0.3:
            args += ["--background.threshold", rs.choice(["0.7", "0.75", "0.85"])]
        if rs.random() < 0.3:
            args += ["--background.cover", rs.choice(["1", "2"])]
        if rs.random() < 0.25:
            args += ["--target.amplicon.min", "60", "--target.amplicon.max", str(rs.choice([150, 250]))]
        if rs.random() < 0.5:
            args += ["--o.json"]
        specs.append(dict(n_fam=rs.randint(1, 3), per=rs.randint(3, 6), L=rs.randint(400, 900), n_bg=rs.choice([0, 1, 2, 3]),
                          div=rs.choice([0.02, 0.03, 0.05]), bg_div=rs.choice([0.05, 0.1, 0.15]), args=args))
    runs = []
    with tempfile.TemporaryDirectory() as tmp:
        libdir = os.path.join(tmp, "lib")
        os.makedirs(libdir)
        for so in ("libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0"):
            os.symlink(os.path.join("/opt/conda/lib", so), os.path.join(libdir, so))
        env = dict(os.environ, LD_LIBRARY_PATH=libdir, OMP_NUM_THREADS="1")
        for si, sp in enumerate(specs):
            input_seed = 7100 + (si // 2 if si < 8 else si)
            r2 = random.Random(input_seed)
            roots = [rand_seq(r2, sp["L"] + 7 * k) for k in range(sp["n_fam"])]
            targets = [(">target_%d family %d" % (k * sp["per"] + j, k), mutate(r2, roots[k], sp["div"])) for k in range(sp["n_fam"]) for j in range(sp["per"])]
            bgs = [(">bg_%d" % i, mutate(r2, roots[i % len(roots)], sp["bg_div"])) for i in range(sp["n_bg"])]
            with open(os.path.join(tmp, "t.fa"), "w") as f:
                f.write("".join("%s\n%s\n" % (d, q) for d, q in targets))
            argv = ["pcramp", "-t", "t.fa", "-o", "out.txt", "--thread", "1"] + sp["args"]
            if bgs:
                with open(os.path.join(tmp, "b.fa"), "w") as f:
                    f.write("".join("%s\n%s\n" % (d, q) for d, q in bgs))
                argv += ["-b", "b.fa"]
            pr = subprocess.run([exe] + argv[1:], cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            # a `throw "..."` inside the reference's OpenMP regions (the sampler that finds no valid assay on targets cut up by earlier
            # amplicons, say) ends the program through std::terminate, its buffered output file lost: recorded as such
            aborted = pr.returncode != 0
            out = "" if aborted else open(os.path.join(tmp, "out.txt"), "rb").read().decode("latin-1").replace(exe, "pcramp")
            runs.append({"argv": argv, "seed": int(sp["args"][sp["args"].index("--seed") + 1]), "json": int("--o.json" in sp["args"]),
                         "input_seed": input_seed, "spec": {k: sp[k] for k in ("n_fam", "per", "L", "n_bg", "div", "bg_div")},
                         "targets": [[d, len(q)] for d, q in targets], "backgrounds": [[d, len(q)] for d, q in bgs], "output": out,
                         "aborted": int(aborted), "stderr_tail": pr.stderr.decode("latin-1")[-160:] if aborted else ""})
            print("program run", si, argv[7:], "->", out.count("ASSAY.") + out.count('"forward primer"'), "assays,", out.count("\nB-"), "background lines")
    return {"runs": runs}


def main():
    build_reference()
    ref = Reference()
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    for name, fn in (("words", words_golden), ("pack", pack_golden), ("screen", screen_golden), ("sw", sw_golden),
                     ("thermo", thermo_golden), ("moves", moves_golden), ("sampler", sampler_golden),
                     ("overlap", overlap_golden), ("multiplex", multiplex_golden),
                     ("multiplex_optimize", multiplex_optimize_golden), ("amplicons", amplicons_golden),
                     ("background", background_golden), ("multiplex_match", multiplex_match_golden), ("writers", writers_golden),
                     ("degenerate", degenerate_golden), ("program", program_golden)):
        if only and name not in only:
            continue
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(fn(ref), f, separators=(",", ":"))
        print("wrote", name, os.path.getsize(os.path.join(OUT, name + ".json")), "bytes")


if __name__ == "__main__":
    main()
