#!/usr/bin/env python3
"""Secondary figures SURVEY.md section 8(d) asks for, through the C-ABI (host buffers in, host buffers
out, so PCIe and launch overhead are included -- end-to-end per call, not kernel-only):
  * Smith-Waterman (pcr_sw_align_words, SeqOverlap lanes): GCUPS, cells = |q| x |t| per lane
  * thermodynamics (pcr_thermo = PCR::is_valid incl. hairpin + homodimer): oligos/s
  * thermodynamics, small batch (the latency the local search and the sampler see): seconds per call of 64 oligos
  * local-search move evaluation (pcr_move_coverage): trial words/s at C2 scale
  * the optimize() loop of the local search (pcramp_amd.moves.optimize) on C2 targets + 2 000 backgrounds: seconds per
    assay
  * random assay sampler (pcr_random_assays): trials/s on the C2 targets (the CPU oracle's time for the same 1 000
    trials is printed by tests/test_gpu_sampler.py::test_sampler_at_c2_scale -- checkers live under tests/)
Prints one JSON object.  python profiles/bench_kernels.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pcramp_amd import api, synth, words as W  # noqa: E402


def main():
    rs = np.random.RandomState(1)
    scr = api.Screener(0)
    out = {}
    # ---- SW lanes: 32-slot word against 32-slot word (the shape find_background_match uses per candidate)
    n = 200_000
    def rand_word(k):
        return W.centered_word(2 ** rs.randint(0, 4, size=k).astype(np.uint8))
    q = [rand_word(rs.randint(18, 26)) for _ in range(2000)]
    t = [rand_word(32) for _ in range(2000)]
    qs = [q[i % 2000] for i in range(n)]
    ts = [t[(7 * i) % 2000] for i in range(n)]
    scr.sw_align_words(qs[:1000], ts[:1000])
    qa = np.array([[w[0], w[1]] for w in qs], dtype=np.uint64)
    ta = np.array([[w[0], w[1]] for w in ts], dtype=np.uint64)
    res = (api.SwResult * n)()
    t0 = time.perf_counter()
    for _ in range(3):
        scr._check(scr.L.pcr_sw_align_words(scr.h, qa.ctypes.data, ta.ctypes.data, n, res))
    dt = (time.perf_counter() - t0) / 3
    cells = sum(sum(1 for v in W.slots_from_word(a) if v) for a in q) / 2000.0 * 32
    out["sw"] = {"lanes": n, "seconds_per_call": dt, "lanes_per_s": n / dt, "GCUPS": n * cells / dt / 1e9,
                 "cells_per_lane": cells}
    # ---- thermo: is_valid with hairpin + homodimer
    m = 20_000
    ol = [rand_word(rs.randint(18, 26)) for _ in range(m)]
    scr.is_valid(ol[:256], True)
    t0 = time.perf_counter()
    scr.is_valid(ol, True)
    dt = time.perf_counter() - t0
    out["thermo"] = {"oligos": m, "seconds_per_call": dt, "oligos_per_s": m / dt}
    t0 = time.perf_counter()
    for k in range(50):
        scr.is_valid(ol[64 * k:64 * k + 64], True)
    out["thermo_small_batch"] = {"oligos": 64, "seconds_per_call": (time.perf_counter() - t0) / 50}
    # ---- move evaluation at C2 scale: all +degeneracy trials of one oligo (~60) in one call
    wl = synth.workload("C2", 0, 1.0)
    scr.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
    thr = float(np.float32(1.0) * np.float32(0.9))
    scr.select_words(wl["pairs"], thr, 18, True, True)
    p = wl["pairs"][0]
    trials = api.host_move_trials(p[0], 0, 16, 18, 25)
    scr.move_coverage(p, 0, trials)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        scr.move_coverage(p, 0, trials)
    dt = (time.perf_counter() - t0) / reps
    out["move_coverage"] = {"trials_per_call": len(trials), "targets": wl["T"], "seconds_per_call": dt,
                            "trial_x_target_evals_per_s": len(trials) * wl["T"] / dt}
    # ---- the whole optimize() loop for a few of the trial assays (targets + a background set)
    try:
        from pcramp_amd import moves
        nb = 2000
        bsel = slice(0, int(wl["byte_offsets"][nb]))
        scr.load_sequences(wl["packed"][bsel], wl["byte_offsets"][:nb], wl["lengths"][:nb], which=api.BACKGROUND)
        bthr = float(np.float32(0.8) * np.float32(0.9))
        scr.select_words(wl["pairs"], bthr, 16, True, True, which=api.BACKGROUND)
        kw = dict(degen=16, target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200)
        moves.optimize(scr, wl["pairs"][0], **kw)
        t0 = time.perf_counter()
        res = [moves.optimize(scr, pp, **kw) for pp in wl["pairs"][:8]]
        dt = (time.perf_counter() - t0) / 8
        out["optimize_loop"] = {"assays": 8, "targets": wl["T"], "backgrounds": nb, "seconds_per_assay": dt}
    except Exception as e:                                             # noqa: BLE001
        out["optimize_loop"] = {"error": str(e)}
    # ---- sampler: 1000 trial assays on one running rand_r state (main.cpp:544-550 at one thread)
    scr.random_assays(1, 20)
    t0 = time.perf_counter()
    pairs, _, info = scr.random_assays(7, 1000)
    dt = time.perf_counter() - t0
    out["sampler"] = {"trials": 1000, "seconds": dt, "trials_per_s": 1000 / dt,
                      "mean_attempts": sum(i["assay_iterations"] for i in info) / 1000.0}
    scr.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
