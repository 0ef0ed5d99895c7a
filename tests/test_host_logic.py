"""CPU tests of the product's host half (through the C-ABI's host-only helpers): the window
model must reproduce Sequence::pack exactly -- irregular words UNION (valid regular windows,
both strands) == the oracle's pack() output -- and the candidate list must match select_words'.
"""
import random

import numpy as np
import pytest

from pcramp_amd import api, words as W
from testdata import rand_seq, revcomp


def window_entries(seq_txt, valid, index=0):
    """What the GPU implies for regular windows: (word, loc, strand) for both strands."""
    out = []
    codes = W.codes_from_text(seq_txt)
    for p in np.nonzero(valid)[0]:
        win = codes[p:p + 32]
        plus = W.word_from_slots(win)
        minus = W.word_from_slots(W.revcomp_codes(win))
        out.append((plus[0], plus[1], int(p), index, 1))          # sequence.cpp:184
        out.append((minus[0], minus[1], int(p) + 31, index, 2))   # sequence.cpp:190
    return out


CASES = [
    (23, 0.0, 0.0, [], 18, 256, 0.0, 1.0), (31, 0.0, 0.0, [], 18, 256, 0.0, 1.0),
    (32, 0.0, 0.0, [], 18, 256, 0.0, 1.0), (33, 0.0, 0.0, [], 18, 256, 0.0, 1.0),
    (34, 0.0, 0.0, [], 18, 256, 0.0, 1.0), (35, 0.0, 0.0, [], 18, 256, 0.0, 1.0),
    (64, 0.0, 0.0, [], 18, 256, 0.0, 1.0), (65, 0.0, 0.0, [], 18, 256, 0.0, 1.0),
    (77, 0.0, 0.0, [], 18, 256, 0.0, 1.0), (200, 0.05, 0.02, [], 18, 256, 0.0, 1.0),
    (201, 0.05, 0.02, [], 16, 256, 0.0, 1.0), (300, 0.0, 0.0, [100], 18, 256, 0.0, 1.0),
    (301, 0.0, 0.0, [100, 101, 102, 103], 18, 256, 0.0, 1.0), (300, 0.0, 0.0, [5, 150, 170, 299], 18, 256, 0.0, 1.0),
    (300, 0.0, 0.0, [0, 1, 40], 16, 256, 0.0, 1.0), (257, 0.1, 0.05, [64, 65, 128], 18, 16, 0.0, 1.0),
    (400, 0.0, 0.0, [], 18, 256, 0.3, 0.7), (401, 0.02, 0.0, [200], 18, 256, 0.4, 0.6),
    (150, 0.0, 0.3, [], 18, 256, 0.0, 1.0), (90, 0.0, 0.0, [44, 45], 10, 256, 0.0, 1.0),
    (1200, 0.01, 0.01, [31, 32, 63, 64, 95, 600, 633, 1199], 18, 256, 0.0, 1.0),
    (1201, 0.0, 0.0, [1200], 18, 256, 0.0, 1.0), (500, 0.0, 0.0, [468], 18, 256, 0.0, 1.0),
    (500, 0.0, 0.0, [467, 499], 18, 256, 0.0, 1.0), (3, 0.0, 0.0, [], 1, 256, 0.0, 1.0),
]


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_window_model_equals_pack(oracle, ci):
    L, pd, pn, eos, min_len, degen_thr, min_gc, max_gc = CASES[ci]
    rng = random.Random(500 + ci)
    for rep in range(4):
        s = list(rand_seq(rng, L, p_degen=pd, p_n=pn))
        for e in eos:
            s[e] = "-"
        s = "".join(s)
        packed = W.pack_codes(W.codes_from_text(s))
        irr = api.host_irregular_words(packed, L, min_len, degen_thr, min_gc, max_gc)
        valid = api.host_window_valid(packed, L, degen_thr, min_gc, max_gc)
        got = sorted([(a, b, loc, 0, st) for (a, b, loc, st) in irr] + window_entries(s, valid))
        want = oracle.pack(s, 0, degen_thr, min_gc, max_gc, min_len)
        assert got == want


def test_irregular_count_is_small():
    rng = random.Random(9)
    s = rand_seq(rng, 5000)
    packed = W.pack_codes(W.codes_from_text(s))
    irr = api.host_irregular_words(packed, len(s), 18)
    assert len(irr) < 80   # head + tail partial words only


@pytest.mark.parametrize("opt5,opt3", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_candidates_match_oracle_shifts(oracle, opt5, opt3):
    rng = random.Random(3)
    pairs = []
    for _ in range(10):
        f = oracle.centered_word(rand_seq(rng, rng.randint(18, 25), p_degen=0.1))
        r = oracle.centered_word(rand_seq(rng, rng.randint(18, 25)))
        pairs.append((f, r))
    thr = 0.9
    words, floors = api.host_candidates(pairs, opt5, opt3, thr)
    want = []
    for f, r in pairs:
        for o in (f, r):
            want.append(o)
            cs, ce = oracle.word_start(o), oracle.word_stop(o)
            if opt5 and cs > 0:
                t = o
                for _ in range(cs):
                    t = oracle.word_shift_left(t)
                    want.append(t)
            if opt3 and ce < 31:
                t = o
                for _ in range(ce, 31):
                    t = oracle.word_shift_right(t)
                    want.append(t)
    assert words == want
    for w, fl in zip(words, floors):
        assert int(fl) == int(np.float32(oracle.word_size(w)) * np.float32(thr))


def test_centered_word_matches_oracle(oracle):
    rng = random.Random(4)
    for n in range(1, 33):
        s = rand_seq(rng, n, p_degen=0.2)
        assert W.centered_word(W.codes_from_text(s)) == oracle.centered_word(s)
        assert W.word_text(oracle.centered_word(s)) == s


def test_coverage_from_bits_order():
    # compute_coverage visits {F(+),R(-)} amplicons first, then {R(+),F(-)}: the double sum
    # must follow that order, not plain ascending index.
    w = np.array([1e8, 1.0, 1e-8, 3.0], np.float32)
    fr = np.array([0b1000], np.uint64)
    rf = np.array([0b0111], np.uint64)
    exp = np.float32(np.float64(w[3]) + np.float64(w[0]) + np.float64(w[1]) + np.float64(w[2]))
    assert api.coverage_from_bits(fr, rf, w) == exp
    assert api.weighted_coverage(np.array([0b1111], np.uint64), w) == np.float32(
        np.float64(w[0]) + np.float64(w[1]) + np.float64(w[2]) + np.float64(w[3]))


def test_abi_exports_every_declared_symbol():
    import re, os
    L = api.load_library()
    hdr = open(os.path.join(os.path.dirname(api.library_path()), "..", "include", "pcramp_hip.h")).read()
    declared = set(re.findall(r"\b(pcr_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"pcr_ctx"}
    assert declared == set(api.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.PcrError):
        api.Screener(0)


# ---------------------------------------------------------------------------- seed scan filter
def _match_count(slots_oligo, window_codes):
    """Word::operator& (word.h): slots whose base sets intersect."""
    return sum(1 for k in range(32) if slots_oligo[k] & window_codes[k])


@pytest.mark.parametrize("seed", range(6))
def test_seed_filter_is_a_necessary_condition(seed):
    """Every 32-base ACGT window that reaches the floor must contain one of the orientation's
    seeds at its offset (the pigeonhole argument the seed scan relies on); random windows planted
    with up to k mismatches, plus the exact site, are all caught."""
    rng = random.Random(4200 + seed)
    two_bit = {1: 0, 2: 1, 4: 2, 8: 3}
    n_seedable = 0
    for _ in range(60):
        size = rng.randint(18, 32)
        codes = [rng.choice([1, 2, 4, 8]) for _ in range(size)]
        for _ in range(rng.choice([0, 0, 1, 2, 3])):           # IUPAC positions in the oligo
            codes[rng.randrange(size)] = rng.choice([3, 5, 6, 9, 10, 12, 7, 11, 13, 14, 15])
        word = W.centered_word(np.array(codes, dtype=np.uint8))
        slots = [int(v) for v in W.slots_from_word(word)]
        thr = rng.choice([0.95, 0.9, 0.9, 0.85, 0.8, 0.7])
        floor = int(np.float32(size) * np.float32(thr))
        seeds = api.host_orientation_seeds(word, floor)
        if seeds is None:
            continue
        n_seedable += 1
        occ = [k for k in range(32) if slots[k]]
        k_max = size - floor
        for trial in range(40):
            win = [rng.choice([1, 2, 4, 8]) for _ in range(32)]
            for k in occ:                                       # plant a site ...
                win[k] = rng.choice([b for b in (1, 2, 4, 8) if slots[k] & b])
            for k in rng.sample(occ, rng.randint(0, k_max)):    # ... with at most k mismatching slots
                bad = [b for b in (1, 2, 4, 8) if not (slots[k] & b)]
                if bad:
                    win[k] = rng.choice(bad)
            assert _match_count(slots, win) >= floor
            hit = False
            for code, q, off in seeds:
                if all(two_bit[win[off + j]] == ((code >> (2 * j)) & 3) for j in range(q)):
                    hit = True
                    break
            assert hit, (size, floor, seeds, win)
    assert n_seedable > 10


def test_seed_filter_declines_low_thresholds():
    word = W.centered_word(W.codes_from_text("ACGTACGTACGTACGTACGT"))
    assert api.host_orientation_seeds(word, 10) is None        # k = 10: no block structure fits 20 bases
    assert api.host_orientation_seeds(word, 0) is None         # floor 0: every window matches
    s = api.host_orientation_seeds(word, 18)                   # k = 2: three exact blocks (7,7,6 -> 4+4+16 codes) beat 1 + 25
    assert s is not None and len(s) == 24 and all(q == 8 and off + 8 <= 32 for _, q, off in s)
    s = api.host_orientation_seeds(word, 20)                   # exact match required: one 8-gram
    assert s is not None and len(s) == 1


def test_seed_filter_structures_at_low_thresholds():
    """floor = int(size * 0.81) (--target-threshold 0.9 x the 0.9 search multiplier): k = 4 or 5 mismatching slots; every primer
    length has a structure with padded budget-1 blocks or a budget-2 block (code counts for plain oligos, 8-grams)."""
    expect = {18: 277 + 25,            # 8 slots budget 2 + 8 slots budget 1 (3 + 2 >= 5)
              19: 88 + 88 + 64,        # 5 exact + 7 + 7 with budget 1
              20: 64 + 88 + 25,        # 5 exact + 7 + 8
              21: 25 + 25 + 64,        # 5 exact + 8 + 8
              22: 88 + 88 + 25,        # k = 5: 7 + 7 + 8 with budget 1
              23: 88 + 25 + 25, 24: 75, 25: 75}
    for size, n in expect.items():
        word = W.centered_word(W.codes_from_text(("ACGTTGCA" * 4)[:size]))
        s = api.host_orientation_seeds(word, int(np.float32(size) * np.float32(0.81)))
        assert s is not None and len(s) == n, (size, None if s is None else len(s))
        assert all(q == 8 and off + 8 <= 32 for _, q, off in s)


def test_center_and_degeneracy_helpers_match_oracle(oracle):
    """words.center_word / word_degeneracy (used by the optimize() loop) against the oracle's Word::center / degeneracy."""
    rng = random.Random(9)
    for _ in range(400):
        n = rng.randint(1, 32)
        start = rng.randint(0, 32 - n)
        slots = [0] * 32
        for k in range(start, start + n):
            slots[k] = rng.choice([1, 2, 4, 8, 3, 5, 15, 7])
        w = W.word_from_slots(slots)
        assert W.center_word(w) == oracle.word_center(w)
        assert W.word_degeneracy(w) == oracle.word_degeneracy(w)
    assert W.center_word((0, 0)) == (0, 0)


def test_host_rand_r_is_glibc_rand_r():
    """pcr_host_rand_r against this host's libc and the values the reference's libc produced (golden)."""
    import ctypes
    import json
    import os
    libc = ctypes.CDLL("libc.so.6")
    for seed in (0, 1, 99, 0x7fffffff, 0x80000001, 0xffffffff):
        st = ctypes.c_uint(seed)
        s = seed
        for _ in range(5):
            v = libc.rand_r(ctypes.byref(st))
            got, s = api.host_rand_r(s)
            assert (got, s) == (v, st.value)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler.json")) as f:
        for seed, vals, after in json.load(f)["rand_r"]:
            s = seed
            for v in vals:
                got, s = api.host_rand_r(s)
                assert got == v
            assert s == after


def test_host_overlap_helpers(oracle):
    """pcr_host_max_overlap / pcr_host_oligo_overlap (bit-plane form) against the oracle's DP restatement and the
    reference's own values (golden)."""
    import json
    import os
    import random
    from testdata import rand_seq
    rng = random.Random(17)
    words = []
    for _ in range(150):
        w = oracle.word(rand_seq(rng, rng.randint(1, 32), p_degen=0.25))
        for _ in range(rng.randint(0, 8)):
            w = oracle.word_shift_right(w)
        words.append(w)
    words += [oracle.word("ACGTACGTACGTACGTACGTACGTACGTACGT"), oracle.word("A"), words[3]]
    for a in words[::2]:
        for b in words:
            assert api.host_max_overlap(a, b) == np.float32(oracle.max_overlap(a, b)), (a, b)
    for _ in range(150):
        assay = (rng.choice(words), rng.choice(words))
        pool = [(rng.choice(words), rng.choice(words)) for _ in range(rng.randint(0, 7))]
        assert api.host_oligo_overlap(assay, pool) == np.float32(oracle.oligo_overlap(assay, pool))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "overlap.json")) as f:
        g = json.load(f)
    gw = [(int(h[0], 16), int(h[1], 16)) for h in g["words"]]
    for i, j, v in g["max_overlap"]:
        assert api.host_max_overlap(gw[i], gw[j]) == np.float32(v)
    for c in g["oligo_overlap"]:
        a = c["assay"]
        pool = [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pool"]]
        assert api.host_oligo_overlap(((int(a[0], 16), int(a[1], 16)), (int(a[2], 16), int(a[3], 16))), pool) == np.float32(c["overlap"])


def test_design_options_from_argv():
    """pcramp_amd.design maps the reference's command-line switches (options.cpp:161-214) onto pcr_design_args; unknown ones raise."""
    from pcramp_amd import design
    o = design.options_from_argv(["pcramp", "-t", "t.fa", "-o", "out.txt", "--thread", "1", "--count", "3", "--trial", "40", "--seed", "42",
                                  "-d", "8", "--optimize.top-down", "--optimize.3", "--target.threshold", "0.9", "--o.json", "-b", "b.fa"])
    assert (o["num_assay"], o["num_trial"], o["seed"], o["max_degen"]) == (3, 40, 42, 8.0)
    assert o["top_down_search"] == 1 and o["optimize_3"] == 1 and o["optimize_5"] == 0 and o["json"] == 1
    assert o["target_threshold"] == 0.9 and o["background_threshold"] == design.DEFAULTS["background_threshold"]
    assert design.DEFAULTS["use_multiplex"] == 1 and design.DEFAULTS["num_trial"] == 1000          # options.cpp:72, pcramp.h:32
    with pytest.raises(ValueError):
        design.options_from_argv(["pcramp", "--no-such-switch"])
    import ctypes as C
    assert C.sizeof(design.DesignArgs) == 112                                                      # pcr_design_args, include/pcramp_hip.h
