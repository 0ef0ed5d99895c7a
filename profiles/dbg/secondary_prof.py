"""The secondary kernels bench.py prices (its `secondary` block) for a kernel-level profile: run under
rocprofv3 --kernel-trace --stats, or under a --pmc pass, on the GPU box:
  * Smith-Waterman lanes (pcr_sw_align_words: k_sw_words), 200 000 lanes of an 18-25-mer against a 32-slot word
  * thermodynamics (pcr_thermo: thermo::k_thermo_wave), is_valid with homodimer of 20 000 oligos
  * C3's background path: select_words at 0.72 on 10 000 backgrounds (k_scan2) + find_background_match (k_bg_emit, k_sw_bg, k_bg_score)
  * pcr_optimize_batch: 256 sampler assays on the C2 targets + 2 000 backgrounds (k_pair_moves_lds, k_match_t, k_thermo_wave ...)
usage: secondary_prof.py [sw] [thermo] [c3bg] [opt]   (default: all)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcramp_amd import api, synth, moves, words as W  # noqa: E402

what = set(sys.argv[1:]) or {"sw", "thermo", "c3bg", "opt"}
rs = np.random.RandomState(1)


def rand_word(k):
    return W.centered_word(2 ** rs.randint(0, 4, size=k).astype(np.uint8))


s = api.Screener(0)
if "sw" in what:
    n = 200_000
    q = [rand_word(rs.randint(18, 26)) for _ in range(2000)]
    t = [rand_word(32) for _ in range(2000)]
    qa = np.array([[q[i % 2000][0], q[i % 2000][1]] for i in range(n)], dtype=np.uint64)
    ta = np.array([[t[(7 * i) % 2000][0], t[(7 * i) % 2000][1]] for i in range(n)], dtype=np.uint64)
    res = (api.SwResult * n)()
    s._check(s.L.pcr_sw_align_words(s.h, qa.ctypes.data, ta.ctypes.data, 1000, res))
    t0 = time.perf_counter()
    for _ in range(3):
        s._check(s.L.pcr_sw_align_words(s.h, qa.ctypes.data, ta.ctypes.data, n, res))
    print("sw: %.3f ms per call of %d lanes" % ((time.perf_counter() - t0) / 3 * 1e3, n))
if "thermo" in what:
    m = 20_000
    ol = np.array([list(rand_word(rs.randint(18, 26))) for _ in range(m)], dtype=np.uint64)
    resb = (api.ThermoResult * m)()
    args = s._targs(0.05, 9e-7, 50.0, 75.0, 40.0, 40.0)
    s._check(s.L.pcr_thermo(s.h, ol.ctypes.data, m, 1, C.byref(args), resb))
    t0 = time.perf_counter()
    for _ in range(5):
        s._check(s.L.pcr_thermo(s.h, ol.ctypes.data, m, 1, C.byref(args), resb))
    print("thermo: %.3f ms per call of %d oligos" % ((time.perf_counter() - t0) / 5 * 1e3, m))
if "c3bg" in what:
    c3 = synth.workload("C3")
    bg = c3["background"]
    s.load_sequences(bg["packed"], bg["byte_offsets"], bg["lengths"], which=api.BACKGROUND)
    bthr = float(np.float32(0.8) * np.float32(0.9))
    p3 = c3["pairs"]
    s.select_words(p3, bthr, 16, which=api.BACKGROUND, count=False)
    s.find_background_match(p3, 0.8, 0.9, 0, 2000, False)
    t0 = time.perf_counter()
    for _ in range(10):
        s.select_words(p3, bthr, 16, which=api.BACKGROUND, count=False)
    s.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        s.find_background_match(p3, 0.8, 0.9, 0, 2000, False)
    print("c3 backgrounds: select_words %.3f ms, find_background_match %.3f ms" % ((t1 - t0) / 10 * 1e3, (time.perf_counter() - t1) / 10 * 1e3))
if "opt" in what:
    wl = synth.workload("C2")
    nbg = 2000
    bsel = slice(0, int(wl["byte_offsets"][nbg]))
    s.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"], which=api.TARGET)
    s.load_sequences(wl["packed"][bsel], wl["byte_offsets"][:nbg], wl["lengths"][:nbg], which=api.BACKGROUND)
    thr = float(np.float32(1.0) * np.float32(0.9))
    bthr = float(np.float32(0.8) * np.float32(0.9))
    kw = dict(degen=16, target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200)
    trial, _, _ = s.random_assays(2024, 256)
    s.select_words(trial, thr, 18, count=False)
    s.select_words(trial, bthr, 16, which=api.BACKGROUND, count=False)
    moves.optimize_batch(s, trial, **kw)
    t0 = time.perf_counter()
    _, _, it = moves.optimize_batch(s, trial, **kw)
    print("optimize_batch: 256 assays %.1f ms, <= %d iterations" % ((time.perf_counter() - t0) * 1e3, max(it)))
s.close()
