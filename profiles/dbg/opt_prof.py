"""pcr_optimize_batch under a profiler: opt_prof.py single | batch N | both N  (256 sampler assays on the C2 targets + 2 000 backgrounds is
what bench.py's secondary.optimize_batch times).  profiles/run_r03_profiles.sh passes `opt` and `optsq` run it under rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pcramp_amd import api, synth, moves, words as W
wl = synth.workload("C2")
s = api.Screener(0)
nbg = 2000
bsel = slice(0, int(wl["byte_offsets"][nbg]))
s.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"], which=api.TARGET)
s.load_sequences(wl["packed"][bsel], wl["byte_offsets"][:nbg], wl["lengths"][:nbg], which=api.BACKGROUND)
thr = float(np.float32(1.0) * np.float32(0.9)); bthr = float(np.float32(0.8) * np.float32(0.9))
kw = dict(degen=16, target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200)
mode = sys.argv[1]
if mode in ("single", "both"):
    s.select_words(wl["pairs"], thr, 18, True, True, count=False)
    s.select_words(wl["pairs"], bthr, 16, True, True, which=api.BACKGROUND, count=False)
    moves.optimize(s, wl["pairs"][0], **kw)
    t0 = time.perf_counter()
    for pp in wl["pairs"][:8]:
        moves.optimize(s, pp, **kw)
    print("single ms/assay", (time.perf_counter() - t0) / 8 * 1e3)
if mode in ("batch", "both"):
    n = int(sys.argv[2])
    if len(sys.argv) > 3:
        s.load_sequences(wl["packed"][bsel], wl["byte_offsets"][:nbg], wl["lengths"][:nbg], which=api.BACKGROUND)
    trial, _, _ = s.random_assays(2024, n)
    s.select_words(trial, thr, 18, count=False)
    s.select_words(trial, bthr, 16, which=api.BACKGROUND, count=False)
    s.synchronize()
    t0 = time.perf_counter()
    s.select_words(trial, bthr, 16, which=api.BACKGROUND, count=False)
    s.synchronize()
    print("background select_words for", n, "assays: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    moves.optimize_batch(s, trial, **kw)                 # first call sizes the buffers
    t0 = time.perf_counter()
    _, _, it = moves.optimize_batch(s, trial, **kw)
    dt = time.perf_counter() - t0
    print("batch", n, "ms total", dt * 1e3, "ms/assay", dt / n * 1e3, "iters", max(it))
s.close()
