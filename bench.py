#!/usr/bin/env python3
"""bench.py -- primer-pair x target amplification evaluations / second on MI355X.

One step = one pass of the hot path over one batch of synthetic input, inputs already resident in HBM: the
per-iteration index build (Sequence::pack + select_words for every target, reference main.cpp:644-691) followed
by the amplicon screen of every primer pair against every target (PCR::find_target_match, pcr_assay.cpp:544).

Workload (default): BASELINE.json configs[1] ("C2": 10 000 viral targets x 10 kb, 50 primer pairs) PER GPU.  The
timed loop rotates through three distinct target sets of that shape (420 MB resident, more than the 256 MB Infinity
Cache), so a pass streams its targets from HBM instead of re-reading what the previous pass left on the die.

--gpus N (default workload): weak scaling.  Every target set is ONE global set of N x 10 000 targets cut by
pcramp_amd.shard.shard_ranges (contiguous blocks, boundaries multiples of 64 sequences); rank r loads its block; the
pairs are replicated; the only exchange is one all-gather of the [2, P, words] orientation bitsets per pass (batched,
pipelined).  --config C4 | C5: strong scaling of BASELINE.json configs[3] / configs[4] -- ONE fixed target set
(5 000 x 5 Mb, 100 000 x 10 kb) sharded over the ranks.  Before the clock starts rank 0 screens the unsharded set (or,
for C4, one member of every rank's block) on its own GPU and asserts that the gathered bits and
pcr_coverage_from_bits equal it (--no-verify skips that).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, HBM)
VALU_CLOCK_GHZ = 2.4        # MI355X engine clock (same guide); 256 CUs x 4 SIMDs, a wave64 VALU op holds its SIMD 4 clocks
N_SIMD = 1024
STRONG = {"C4": ("C4_shard", 8), "C5": ("C5_shard", 8)}       # name -> (block config, blocks of the fixed global set)
PROFILE_ROUND = "r03"


# ---------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(wl, thr_t, mult, budget_s=12.0):
    """The reference's own CPU path (oracle/_ref, kind "reference"; our restatement, "port", where it is absent) timed
    on a bounded sample of the SAME workload, at one thread, at 16 and at ALL host threads of this box (the best leg is `value`)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    host = len(os.sched_getaffinity(0))
    cores = min(16, host)
    os.environ.setdefault("OMP_NUM_THREADS", str(host))
    from oracle_lib import Oracle, Reference
    from pcramp_amd import words as W
    kind = "reference" if Reference.available() else "port"
    lib = Reference() if kind == "reference" else Oracle()
    omp = None
    if kind == "reference":
        try:
            omp = ctypes.CDLL("libgomp.so.1")
        except OSError:
            omp = None
    nb = (wl["L"] + 1) // 2
    texts = {}

    def text(i):
        if i not in texts:
            o = int(wl["byte_offsets"][i])
            texts[i] = W.text_from_codes(W.unpack_codes(wl["packed"][o:o + nb], wl["L"]))
        return texts[i]

    def run(n_t, threads):
        if omp is not None:
            omp.omp_set_num_threads(int(threads))
        s = lib.session(target_threshold=thr_t, search_multiplier=mult)
        for i in range(n_t):
            s.add_target(text(i))
        t0 = time.perf_counter()
        s.select(wl["pairs"])                 # main.cpp:644-676: targets one after the other, OpenMP inside select_words
        for p in wl["pairs"]:
            s.target_match(p)
        return time.perf_counter() - t0

    out = {}
    legs = [1]
    if omp is not None and cores > 1:
        legs.append(cores)
        if host > cores:
            legs.append(host)                  # every host thread of the box (the reference parallelises inside select_words only)
    for threads in legs:
        # calibrate on 16 targets (4 for the all-threads leg: a GPU box grants its process a CPU share of 16 cores while
        # sched_getaffinity lists every hardware thread, so that leg can be heavily oversubscribed), then size the sample to the
        # budget (the first targets of the workload); a leg whose calibration is already slower than a finished leg stays there
        n_cal = min(16 if threads <= cores else 4, wl["T"])
        dt = run(n_cal, threads)
        n_big = int(max(n_cal, min(wl["T"], 0.5 * budget_s / max(dt / n_cal, 1e-9))))
        slower = any(n_cal * len(wl["pairs"]) / dt < 0.5 * v[0] for v in out.values())
        if n_big > n_cal and not slower:
            dt = run(n_big, threads)
            n_cal = n_big
        out[threads] = (n_cal * len(wl["pairs"]) / dt, n_cal, dt)
    best = max(out, key=lambda k: out[k][0])
    res = {"value": out[best][0], "unit": "evals/s", "cores": best, "kind": kind,
           "sample": "the first %d of the %d targets (%d bases each) x %d pairs: select_words + find_target_match in %.1f s; "
                     "the reference walks the targets serially and parallelises inside select_words (main.cpp:644-676)"
                     % (out[best][1], wl["T"], wl["L"], len(wl["pairs"]), out[best][2]),
           "host_threads_available": host, "one_thread_evals_per_s": out[1][0]}
    res["legs"] = {str(k): {"evals_per_s": v[0], "targets": v[1], "seconds": v[2]} for k, v in out.items()}
    if cores in out and cores != 1:
        res["sixteen_threads_evals_per_s"] = out[cores][0]
    if host in out and host != 1:
        res["all_threads_evals_per_s"] = out[host][0]
        res["threads_used_all"] = host
        res["speedup_all_threads"] = out[host][0] / out[1][0]
    return res


# ---------------------------------------------------------------------------------------------- helpers
def load_profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:                                                  # noqa: BLE001
        return None


def bits_of(t):
    """int64 tensor/array of bitset words -> uint64 numpy."""
    a = t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)
    return np.ascontiguousarray(a).view(np.uint64)


def assemble_gathered(gathered0, ranges):
    """The all-gathered [world, 2, P, wmax] words of one pass -> [2, P, total words] of the global set: rank r owns
    ceil((hi_r - lo_r) / 64) words and its block starts at a multiple of 64, so the words concatenate."""
    g = bits_of(gathered0)
    words = [(hi - lo + 63) // 64 for lo, hi in ranges]
    return np.concatenate([g[r][:, :, :words[r]] for r in range(len(ranges))], axis=-1)


def check_against_unsharded(full, n_total, ids, fr, rf, cov, weights=None):
    """full: assemble_gathered(); fr / rf: bool [P, len(ids)] of the unsharded screen for the listed sequences; cov: its
    coverage floats or None.  Raises on any difference; -> number of amplification calls set."""
    from pcramp_amd import api
    n_set = 0
    for k in range(full.shape[1]):
        got_fr = api.bits_to_bool(full[0, k], n_total)[ids]
        got_rf = api.bits_to_bool(full[1, k], n_total)[ids]
        if not (np.array_equal(got_fr, fr[k]) and np.array_equal(got_rf, rf[k])):
            raise AssertionError("gathered bits of pair %d differ from the unsharded screen" % k)
        if cov is not None:
            w = np.ones(n_total, np.float32) if weights is None else np.asarray(weights, np.float32)
            if api.coverage_from_bits(full[0, k], full[1, k], w) != cov[k]:
                raise AssertionError("pcr_coverage_from_bits of the gathered bits differs from the unsharded coverage (pair %d)" % k)
        n_set += int((got_fr | got_rf).sum())
    return n_set


def verify_first_pass(api, synth, gs, ranges, pa, thr, local_rank, gathered0, stream, sample_only):
    """Rank 0: the gathered [world, 2, P, wmax] words of the first pass against the unsharded set on ONE GPU.
    sample_only: screen the first and last member of every rank's block instead of the whole set (evaluation is
    independent per target)."""
    select_thr, thr_t = thr
    full = assemble_gathered(gathered0, ranges)
    chk = api.Screener(local_rank, stream=stream)
    try:
        if sample_only:
            ids = sorted(set([lo for lo, hi in ranges if hi > lo] + [hi - 1 for lo, hi in ranges if hi > lo]))
            packs = [gs.members(i, i + 1) for i in ids]
            packed = np.concatenate([p[0] for p in packs])
            nbytes = packs[0][0].size
            chk.load_sequences(packed, np.arange(len(ids), dtype=np.uint64) * np.uint64(nbytes), np.full(len(ids), gs.L, np.uint64))
        else:
            ids = list(range(gs.T))
            chk.load_sequences(*gs.members(0, gs.T))
        chk.select_words(pa, select_thr, 18, count=False)
        _, fr, rf, cov = chk.amplify(pa, thr_t, thr_t, 80, 200, False)
        try:
            n_set = check_against_unsharded(full, gs.T, ids, fr, rf, None if sample_only else cov)
        except AssertionError as e:
            raise SystemExit("bench.py: " + str(e))
        return {"checked_targets": len(ids), "mode": "sample (first and last member of every rank's block)" if sample_only else "whole set",
                "amplification_calls_set": n_set}
    finally:
        chk.close()


def per_set_streams_figure(api, sets, lo, hi, dev, pa, thr, steps=3000, warmup=60):
    """The same passes with one HIP stream per target set: the tail of one pass (k_post) and the staging of the next (k_stage) run
    beside another set's scan instead of between two scans.  Kernel durations then include waiting for CUs, so this figure
    carries no roofline block; the headline keeps one stream."""
    import torch
    select_thr, thr_t = thr
    dev_t = torch.device("cuda", dev)
    streams = [torch.cuda.Stream(device=dev_t) for _ in sets]
    scrs = []
    try:
        for gs, st in zip(sets, streams):
            s = api.Screener(dev, stream=st.cuda_stream)
            s.load_sequences(*gs.members(lo, hi))
            scrs.append(s)
        P = pa.shape[0]
        words = int(scrs[0].bitset_words())
        outs = [torch.zeros((2, P, words), dtype=torch.int64, device=dev_t) for _ in scrs]
        torch.cuda.synchronize()

        calls = [s.screen_call(pa, select_thr, o[0].data_ptr(), o[1].data_ptr(), thr_t, thr_t, 80, 200, False, 18, False, False) for s, o in zip(scrs, outs)]

        def run(n):
            for i in range(n):
                calls[i % len(calls)]()
            for s in scrs:
                s.synchronize()
            torch.cuda.synchronize()
        run(warmup)
        t0 = time.perf_counter()
        run(steps)
        dt = time.perf_counter() - t0
        T = int(scrs[0].num_sequences())
        return {"streams": len(scrs), "steps": steps, "ms_per_step": dt / steps * 1e3, "evals_per_s": float(P) * T * steps / dt,
                "note": "one stream per target set (three handles): passes of different sets overlap on the GPU; every pass complete before the clock stops"}
    finally:
        for s in scrs:
            s.close()


def secondary_figures(api, synth, W, dev, stream, wl_single, scr0, pa, thr):
    """SURVEY 8(d)'s mandatory secondary figures, in the same run: each is timed through the C-ABI; kernel-only times come
    from HIP events on the launch stream (pcr_profile_read_kernel).  Failures are recorded, never fatal."""
    import torch
    out = {}
    select_thr, thr_t = thr
    rs = np.random.RandomState(1)

    def rand_word(k):
        return W.centered_word(2 ** rs.randint(0, 4, size=k).astype(np.uint8))
    s = api.Screener(dev, stream=stream)
    try:
        # ---- the headline step WITHOUT rotation: one target set re-screened (140 MB resident, fits the Infinity Cache)
        try:
            words = int(scr0.bitset_words())
            buf = torch.zeros((2, pa.shape[0], words), dtype=torch.int64, device="cuda:%d" % dev)
            call0 = scr0.screen_call(pa, select_thr, buf[0].data_ptr(), buf[1].data_ptr(), thr_t, thr_t, 80, 200, False)
            for _ in range(20):
                call0()
            scr0.synchronize()
            n = 2000
            t0 = time.perf_counter()
            for _ in range(n):
                call0()
            scr0.synchronize()
            out["single_target_set"] = {"ms_per_step": (time.perf_counter() - t0) / n * 1e3,
                                        "note": "the same 10 000 targets every pass (cache-resident); the headline rotates three sets"}
        except Exception as e:                                         # noqa: BLE001
            out["single_target_set"] = {"error": str(e)}
        # ---- Smith-Waterman lanes: 18-25-mer query word x 32-slot template word (find_background_match's lane shape)
        try:
            n = 200_000
            q = [rand_word(rs.randint(18, 26)) for _ in range(2000)]
            t = [rand_word(32) for _ in range(2000)]
            qa = np.array([[q[i % 2000][0], q[i % 2000][1]] for i in range(n)], dtype=np.uint64)
            ta = np.array([[t[(7 * i) % 2000][0], t[(7 * i) % 2000][1]] for i in range(n)], dtype=np.uint64)
            res = (api.SwResult * n)()
            s._check(s.L.pcr_sw_align_words(s.h, qa.ctypes.data, ta.ctypes.data, 1000, res))
            s.profile(1)
            s.profile_read_kernel(1)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                s._check(s.L.pcr_sw_align_words(s.h, qa.ctypes.data, ta.ctypes.data, n, res))
            dt = (time.perf_counter() - t0) / reps
            kms, kn = s.profile_read_kernel(1)              # all launches of the three calls (a call is cut into chunks)
            s.profile(0)
            cells = sum(sum(1 for v in W.slots_from_word(a) if v) for a in q) / 2000.0 * 32
            qlens = [sum(1 for v in W.slots_from_word(a) if v) for a in q]
            # lane utilisation of the anti-diagonal sweep: |q| x |t| cells in (|q| + |t| - 1) steps of 32 lanes
            util = float(np.mean([ql * 32.0 / (32.0 * (ql + 32 - 1)) for ql in qlens]))
            sec_pmc = load_profile_json("%s_secondary_pmc.json" % PROFILE_ROUND) or {}
            sw_roof = {"bound": "valu", "lane_utilisation": util,
                       "note": "k_sw_words carries start coordinates with every cell (five packed shuffles per anti-diagonal step); a |q| x 32 matrix keeps "
                               "|q| x 32 / (32 x (|q| + 31)) of its lane-steps busy"}
            hitk = [v for k, v in sec_pmc.items() if k == "k_sw_words" or k.startswith("k_sw_words")]
            if hitk and hitk[0].get("SQ_INSTS_VALU") and kms > 0:
                # the counter pass profiled the same 200 000-lane call (profiles/dbg/secondary_prof.py): instructions per launch = per chunk of 65 536 lanes
                insts = hitk[0]["SQ_INSTS_VALU"]
                k_s = (kms / max(kn, 1)) / 1e3
                sw_roof.update({"valu_wave_instructions_per_launch": insts, "valu_issue_time_us": insts * 4.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) * 1e6,
                                "frac": insts * 4.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) / k_s if k_s > 0 else None,
                                "frac_at_2clk": insts * 2.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) / k_s if k_s > 0 else None,
                                "source": "profiles/%s_secondary_pmc.json (static)" % PROFILE_ROUND,
                                "model": "SQ_INSTS_VALU x 4 clk (one wave's issue rate; x 2 clk = the SIMD-32's: frac_at_2clk) / (1024 SIMDs x 2.4 GHz) / mean "
                                         "kernel time per launch; a value above 1 at 4 clk means the SIMDs overlap their waves' instructions, i.e. the kernel "
                                         "is issue-bound"})
            out["smith_waterman"] = {"lanes": n, "cells_per_lane": cells, "abi_GCUPS": n * cells / dt / 1e9, "roofline": sw_roof,
                                     "kernel_GCUPS": n * cells / (kms / reps / 1e3) / 1e9 if kms > 0 else None,
                                     "abi_ms_per_call": dt * 1e3, "kernel_ms_per_call": kms / reps, "kernel_launches_per_call": kn / reps,
                                     "note": "ABI: host words in, host results out, per call; kernel: HIP events around its launches"}
        except Exception as e:                                         # noqa: BLE001
            out["smith_waterman"] = {"error": str(e)}
        # ---- thermodynamics: PCR::is_valid (duplex Tm + hairpin + homodimer) of 20 000 oligos in one call
        try:
            m = 20_000
            ol = np.array([list(rand_word(rs.randint(18, 26))) for _ in range(m)], dtype=np.uint64)
            resb = (api.ThermoResult * m)()
            args = s._targs(0.05, 9e-7, 50.0, 75.0, 40.0, 40.0)
            import ctypes as C
            s._check(s.L.pcr_thermo(s.h, ol.ctypes.data, 256, 1, C.byref(args), resb))
            t0 = time.perf_counter()
            s._check(s.L.pcr_thermo(s.h, ol.ctypes.data, m, 1, C.byref(args), resb))       # first call of this size: sizes the buffers
            dt_first = time.perf_counter() - t0
            s.profile(1)
            s.profile_read_kernel(2)
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                s._check(s.L.pcr_thermo(s.h, ol.ctypes.data, m, 1, C.byref(args), resb))
            dt = (time.perf_counter() - t0) / reps
            kms, kn = s.profile_read_kernel(2)
            s.profile(0)
            th_roof = {"bound": "valu", "note": "one job per wave (is_valid = duplex Tm + hairpin + homodimer = 2 wave jobs per oligo): the DP fill runs the wave "
                                                "along anti-diagonals, the trace-backs 8 lanes wide"}
            sec_pmc = load_profile_json("%s_secondary_pmc.json" % PROFILE_ROUND) or {}
            hitk = [v for k, v in sec_pmc.items() if "k_thermo_wave" in k]
            if hitk and hitk[0].get("SQ_INSTS_VALU") and kms > 0:
                insts = hitk[0]["SQ_INSTS_VALU"]
                k_s = (kms / max(kn, 1)) / 1e3
                th_roof.update({"valu_wave_instructions_per_launch": insts, "valu_issue_time_us": insts * 4.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) * 1e6,
                                "frac": insts * 4.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) / k_s if k_s > 0 else None,
                                "frac_at_2clk": insts * 2.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) / k_s if k_s > 0 else None,
                                "wave_cycles_parked_on_waitcnt": (hitk[0].get("SQ_WAIT_ANY") / hitk[0]["SQ_WAVE_CYCLES"]) if hitk[0].get("SQ_WAVE_CYCLES") else None,
                                "source": "profiles/%s_secondary_pmc.json (static)" % PROFILE_ROUND,
                                "model": "SQ_INSTS_VALU x 4 clk / (1024 SIMDs x 2.4 GHz) / kernel time"})
            out["thermodynamics"] = {"oligos": m, "abi_oligos_per_s": m / dt, "abi_ms_per_call": dt * 1e3, "abi_ms_first_call": dt_first * 1e3, "roofline": th_roof,
                                     "kernel_ms": kms / reps, "kernel_launches": int(kn) // reps,
                                     "note": "is_valid with the homodimer test; mean of %d calls after one call of the same size (which allocates the buffers)" % reps}
        except Exception as e:                                         # noqa: BLE001
            out["thermodynamics"] = {"error": str(e)}
        # ---- C3's background path: select_words on 10 000 backgrounds at 0.8 x 0.9, then find_background_match
        try:
            c3 = synth.workload("C3")
            bg = c3["background"]
            s.load_sequences(bg["packed"], bg["byte_offsets"], bg["lengths"], which=api.BACKGROUND)
            p3 = W.pairs_array(c3["pairs"])
            bthr = float(np.float32(0.8) * np.float32(0.9))
            s.select_words(p3, bthr, 16, which=api.BACKGROUND, count=False)
            s.find_background_match(p3, 0.8, 0.9, 0, 2000, False)
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                s.select_words(p3, bthr, 16, which=api.BACKGROUND, count=False)
            s.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                s.find_background_match(p3, 0.8, 0.9, 0, 2000, False)
            t2 = time.perf_counter()
            # the scan launch of that select_words (k_scan2 at k = 7: 0.72 has no seedable structure), HIP events on the launch stream
            s.profile(1)
            s.profile_read()
            for _ in range(reps):
                s.select_words(p3, bthr, 16, which=api.BACKGROUND, count=False)
            kms, kn = s.profile_read()
            s.profile(0)
            k_s = kms / max(kn, 1) / 1e3
            bg_bytes = float(np.asarray(bg["packed"]).nbytes) + 2 * len(c3["pairs"]) * 16
            bg_roof = {"bound": "valu", "kernel": "k_scan2<7,..> (bit-sliced oligo x window scan of the background set at 0.8 x 0.9)",
                       "kernel_ms": kms / max(kn, 1), "launches": int(kn), "algorithmic_bytes_per_launch": bg_bytes,
                       "hbm_achieved_GBps": bg_bytes / k_s / 1e9 if k_s > 0 else None, "hbm_frac": bg_bytes / k_s / 1e9 / HBM_PEAK_GBPS if k_s > 0 else None,
                       "note": "every window of every background against all 200 orientations in bit planes: the packed backgrounds are read once "
                               "(HBM fraction tiny), the time is VALU instructions"}
            sec_pmc = load_profile_json("%s_secondary_pmc.json" % PROFILE_ROUND) or {}
            hitk = [v for k, v in sec_pmc.items() if k.startswith("k_scan2<7")]
            if hitk and hitk[0].get("SQ_INSTS_VALU") and k_s > 0:
                insts = hitk[0]["SQ_INSTS_VALU"]
                bg_roof.update({"valu_wave_instructions_per_launch": insts, "valu_issue_time_us": insts * 4.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) * 1e6,
                                "frac": insts * 4.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) / k_s, "frac_at_2clk": insts * 2.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) / k_s,
                                "wave_cycles_parked_on_waitcnt": (hitk[0].get("SQ_WAIT_ANY") / hitk[0]["SQ_WAVE_CYCLES"]) if hitk[0].get("SQ_WAVE_CYCLES") else None,
                                "source": "profiles/%s_secondary_pmc.json (static)" % PROFILE_ROUND,
                                "model": "SQ_INSTS_VALU x 4 clk / (1024 SIMDs x 2.4 GHz) / kernel time"})
            out["c3_background"] = {"backgrounds": int(bg["B"]), "pairs": len(c3["pairs"]), "select_words_ms": (t1 - t0) / reps * 1e3,
                                    "find_background_match_ms": (t2 - t1) / reps * 1e3,
                                    "background_evals_per_s": bg["B"] * len(c3["pairs"]) / ((t2 - t0) / reps), "roofline": bg_roof}
        except Exception as e:                                         # noqa: BLE001
            out["c3_background"] = {"error": str(e)}
        # ---- the reference PROGRAM's design loop (pcr_design = main.cpp:471-1130 behind the ABI) on the C2 targets with the reference's
        # default options (1 000 trial assays per iteration, -d 1, multiplex on), three iterations, no backgrounds
        try:
            from pcramp_amd import design
            wl = wl_single
            dz = api.Screener(dev, stream=stream)                      # a handle of its own: the call changes active flags and splits targets
            try:
                dz.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"], which=api.TARGET)
                T = int(wl["T"])
                defl = [">t%d" % i for i in range(T)]
                t0 = time.perf_counter()
                text, pool = design.design(dz, defl, [int(x) for x in wl["lengths"]], argv=["pcramp", "--count", "3", "--seed", "42"],
                                           num_assay=3, num_trial=1000, seed=42)
                dt = time.perf_counter() - t0
            finally:
                dz.close()
            out["design_loop"] = {"targets": T, "trial_assays_per_iteration": 1000, "iterations": 3, "assays_accepted": len(pool),
                                  "ms_per_iteration": dt / 3 * 1e3, "output_bytes": len(text),
                                  "evals_per_s": 3 * 1000 * T / dt,
                                  "note": "whole design iterations: sampler, select_words for 1 000 assays, optimize(), multiplex filter, "
                                          "find_target_match, amplicon DB, EOS splits, output text; evals = trial assays x targets per iteration"}
        except Exception as e:                                         # noqa: BLE001
            out["design_loop"] = {"error": str(e)}
        # ---- the optimize() local search on the C2 targets + 2 000 backgrounds: one assay per call, and the trial assays
        # of a design iteration as ONE batch (pcr_optimize_batch; main.cpp:697-887 runs optimize() for num_trial assays)
        try:
            from pcramp_amd import moves
            wl = wl_single
            nbg = 2000
            bsel = slice(0, int(wl["byte_offsets"][nbg]))
            s.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"], which=api.TARGET)
            s.load_sequences(wl["packed"][bsel], wl["byte_offsets"][:nbg], wl["lengths"][:nbg], which=api.BACKGROUND)
            bthr = float(np.float32(0.8) * np.float32(0.9))
            s.select_words(wl["pairs"], select_thr, 18, True, True, count=False)
            s.select_words(wl["pairs"], bthr, 16, True, True, which=api.BACKGROUND, count=False)
            kw = dict(degen=16, target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200)
            moves.optimize(s, wl["pairs"][0], **kw)
            t0 = time.perf_counter()
            for pp in wl["pairs"][:8]:
                moves.optimize(s, pp, **kw)
            out["optimize"] = {"assays": 8, "targets": wl["T"], "backgrounds": nbg, "ms_per_assay": (time.perf_counter() - t0) / 8 * 1e3,
                               "note": "one pcr_optimize_batch call per assay"}
            n_trial = 256
            trial, _, _ = s.random_assays(2024, n_trial)              # the sampler's trial assays (main.cpp:538-550)
            s.select_words(trial, select_thr, 18, count=False)
            s.select_words(trial, bthr, 16, which=api.BACKGROUND, count=False)
            t0 = time.perf_counter()
            moves.optimize_batch(s, trial, **kw)                      # first call of this size: sizes the buffers
            dt_first = time.perf_counter() - t0
            t0 = time.perf_counter()
            _, _, iters = moves.optimize_batch(s, trial, **kw)
            dt = time.perf_counter() - t0
            # the kernels of that call, from the committed rocprofv3 passes of profiles/dbg/opt_prof.py (trace + counters; static: a profiler
            # cannot run inside this process): time per launch, VALU issue share at 4 clk per wave instruction, share of the wave cycles parked
            opt_roof = None
            try:
                pj = load_profile_json("%s_optimize_batch_pmc.json" % PROFILE_ROUND) or {}
                rows = {}
                with open(os.path.join(ROOT, "profiles", "%s_optimize_batch_stats.md" % PROFILE_ROUND)) as f:
                    for line in f:
                        c = [x.strip() for x in line.strip().strip("|").split("|")]
                        if len(c) >= 4 and c[0] not in ("kernel", "---") and not c[0].startswith("-"):
                            try:
                                rows[c[0]] = (int(c[1]), float(c[2]))
                            except ValueError:
                                pass
                opt_roof = {"bound": "valu / LDS and memory latency (see parked share)", "source": "profiles/%s_optimize_batch_{stats.md,pmc.json} (static)" % PROFILE_ROUND,
                            "model": "SQ_INSTS_VALU x 4 clk / (1024 SIMDs x 2.4 GHz) / mean launch time", "kernels": {}}
                for k in ("k_pair_moves_lds<true>", "k_pair_moves_tasks", "k_match_t", "thermo::k_thermo_wave", "k_cov_from_bits"):
                    if k in rows and k in pj and pj[k].get("SQ_INSTS_VALU"):
                        us = rows[k][1]
                        insts = pj[k]["SQ_INSTS_VALU"]
                        opt_roof["kernels"][k] = {"launches_profiled": rows[k][0], "us_per_launch": us,
                                                  "valu_frac": insts * 4.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9) / (us * 1e-6),
                                                  "wave_cycles_parked_on_waitcnt": (pj[k].get("SQ_WAIT_ANY") / pj[k]["SQ_WAVE_CYCLES"]) if pj[k].get("SQ_WAVE_CYCLES") else None}
            except Exception:                                          # noqa: BLE001
                opt_roof = None
            out["optimize_batch"] = {"assays": n_trial, "targets": wl["T"], "backgrounds": nbg, "ms_per_assay": dt / n_trial * 1e3, "roofline": opt_roof,
                                     "ms_total": dt * 1e3, "ms_total_first_call": dt_first * 1e3, "optimiser_iterations_max": max(iters), "optimiser_iterations_mean": sum(iters) / len(iters),
                                     "note": "all trial assays in lockstep: one thermodynamics launch and one move-coverage pass per set per iteration"}
        except Exception as e:                                         # noqa: BLE001
            out.setdefault("optimize", {"error": str(e)})
            out["optimize_batch"] = {"error": str(e)}
        # ---- C5's named path at its per-GPU shard size: 1 000 trial assays of a design iteration (the sampler's), optimize()
        # with degenerate (IUPAC) trial primers allowed, 12 500 targets x 10 kb, no backgrounds
        try:
            from pcramp_amd import moves
            c5 = synth.workload("C5_shard")
            s.load_sequences(c5["packed"], c5["byte_offsets"], c5["lengths"], which=api.TARGET)
            n_trial = 1000
            trial, _, _ = s.random_assays(2025, n_trial)
            t0 = time.perf_counter()
            s.select_words(trial, select_thr, 18, count=False)
            s.synchronize()
            dt_sel_first = time.perf_counter() - t0                     # first call of this size: best[] (100 MB) allocated and cleared, seeds derived
            reps_sel = 5
            t0 = time.perf_counter()
            for _ in range(reps_sel):
                s.select_words(trial, select_thr, 18, count=False)
            s.synchronize()
            dt_sel = (time.perf_counter() - t0) / reps_sel
            # the same DB build + find_target_match for the whole trial batch (what a design iteration issues, main.cpp:644-676,898)
            t0 = time.perf_counter()
            for _ in range(reps_sel):
                s.select_words(trial, select_thr, 18, count=False)
                s.amplify(trial, 1.0, 1.0, 80, 200, False)
            dt_screen = (time.perf_counter() - t0) / reps_sel
            kw = dict(degen=16, target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200, have_background=False)
            t0 = time.perf_counter()
            moves.optimize_batch(s, trial, **kw)                      # first call of this size: sizes the buffers
            dt_first = time.perf_counter() - t0
            t0 = time.perf_counter()
            best, scores, iters = moves.optimize_batch(s, trial, **kw)
            dt = time.perf_counter() - t0
            n_degen = sum(1 for f, r in best if W.word_degeneracy(f) > 1 or W.word_degeneracy(r) > 1)
            out["optimize_batch_c5_shard"] = {"assays": n_trial, "targets": int(c5["T"]), "target_len": int(c5["L"]), "select_words_ms": dt_sel * 1e3,
                                              "select_words_ms_first_call": dt_sel_first * 1e3, "select_words_evals_per_s": n_trial * int(c5["T"]) / dt_sel,
                                              "select_plus_find_target_match_ms": dt_screen * 1e3, "screen_evals_per_s": n_trial * int(c5["T"]) / dt_screen,
                                              "ms_per_assay": dt / n_trial * 1e3, "ms_total": dt * 1e3, "ms_total_first_call": dt_first * 1e3,
                                              "optimiser_iterations_max": max(iters), "optimiser_iterations_mean": sum(iters) / len(iters),
                                              "assays_ending_degenerate": n_degen}
        except Exception as e:                                         # noqa: BLE001
            out["optimize_batch_c5_shard"] = {"error": str(e)}
    finally:
        s.close()
    return out


# ---------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8000, help="timed passes (default: about one second)")
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--config", default="C2", help="C2 (default; weak scaling: 10 000 targets per GPU), C1, C3, C4_shard, C5_shard "
                                                   "(per-GPU shapes), or C4 / C5 (ONE fixed set sharded over the ranks: strong scaling)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the target count (debug only)")
    ap.add_argument("--rotate", type=int, default=3, help="distinct target sets the timed loop rotates through (per-GPU shapes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="N>1: skip the sharded == unsharded check of the first pass")
    ap.add_argument("--random-primers", action="store_true", help="diagnostic: primers unrelated to the targets (no hits)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--gather-every", type=int, default=32, help="N>1: passes per all-gather (their bitsets travel in one collective)")
    ap.add_argument("--exchange", choices=["torch", "abi"], default="torch",
                    help="N>1: the all-gather through torch.distributed (async, on RCCL's stream) or through the C-ABI's pcr_exchange_bits "
                         "(ncclAllGather enqueued on the handle's stream; what a C++ host would call)")
    ap.add_argument("--optimize-shifts", action="store_true",
                    help="diagnostic: --optimize.5/--optimize.3 of the reference (every 5'/3' slot shift of every oligo is a candidate)")
    ap.add_argument("--separate-calls", action="store_true",
                    help="pcr_select_words + pcr_amplify_device per step (host wait between them) instead of pcr_screen_device")
    ap.add_argument("--target-threshold", type=float, default=1.0, help="--target.threshold of the reference (pcramp.h:39 default 1.0)")
    ap.add_argument("--search-multiplier", type=float, default=0.9, help="target search multiplier of the reference (pcramp.h:51 default 0.9)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pcramp_amd import api, shard, synth, words as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PCRAMP_BENCH_REHEARSAL=1: exercise the N>1 code path on a box with fewer GPUs than ranks -- gloo collectives on
    # host copies, ranks share the visible GPUs.  The number it prints is not a measurement.
    rehearsal = os.environ.get("PCRAMP_BENCH_REHEARSAL") == "1"
    use_dist = world > 1 or os.environ.get("PCRAMP_BENCH_FORCE_DIST") == "1"   # FORCE_DIST: run the collective path with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if rehearsal:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev_t = torch.device("cuda", local_rank)
    if use_dist:
        # RCCL (version banner) and gloo (peer list) write to file descriptor 1 when the communicator comes up: send that to
        # stderr so that the JSON line is the only thing on stdout
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
            warm = torch.zeros(1, device="cpu" if rehearsal else dev_t)
            dist.all_reduce(warm)                      # the communicator is created by the first collective
            if not rehearsal:
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    thr_t, mult = args.target_threshold, args.search_multiplier
    select_thr = float(np.float32(thr_t) * np.float32(mult))

    # ---- the global target set(s) and this rank's block
    strong = args.config in STRONG
    if strong:
        block_cfg, n_blocks = STRONG[args.config]
        sets = [synth.GlobalSet(block_cfg, n_blocks, scale=args.scale)]
    else:
        block_cfg = args.config
        R = max(1, args.rotate)
        sets = [synth.GlobalSet(block_cfg, world, scale=args.scale, seed_base=j * world) for j in range(R)]
    pairs = sets[0].pairs() if rank == 0 else None
    if args.random_primers and rank == 0:
        rs = np.random.RandomState(7)
        pairs = [(W.centered_word(2 ** rs.randint(0, 4, size=rs.randint(18, 26))),
                  W.centered_word(2 ** rs.randint(0, 4, size=rs.randint(18, 26)))) for _ in range(len(pairs))]
    if world > 1:
        obj = [pairs]
        dist.broadcast_object_list(obj, src=0)            # the reference broadcasts its trial assays the same way
        pairs = obj[0]
    pa = W.pairs_array(pairs)
    P = len(pairs)
    L = sets[0].L
    ranges = shard.shard_ranges(sets[0].lengths, world)   # every set has the same shape -> the same cut
    lo, hi = ranges[rank]
    T_local, T_total = hi - lo, sets[0].T
    wmax = max((b - a + 63) // 64 for a, b in ranges)

    # one explicit stream for the kernels AND the collectives (the null stream is not ordered against a non-blocking one)
    stream_obj = torch.cuda.Stream(device=dev_t)
    stream = stream_obj.cuda_stream
    scrs = []
    wl_single = None
    for j, gs in enumerate(sets):
        s = api.Screener(local_rank, stream=stream)
        s.load_sequences(*gs.members(lo, hi))
        scrs.append(s)
    if world == 1 and not strong:
        wl_single = sets[0].block(0)
    NS = len(scrs)
    comm = None
    if use_dist and args.exchange == "abi" and not rehearsal:
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)                                     # (RCCL's banner goes to file descriptor 1)
        try:
            obj = [api.Screener.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(obj, src=0)        # the 128 bytes travel however the host likes (the reference: MPI_Bcast)
            comm = scrs[0].comm_init_rank(obj[0], world, rank)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    # The exchange is batched and pipelined: K passes write their bitsets into the K slices of one buffer, one all-gather
    # ships the batch (a torch collective costs the host ~100 us, a pass ~130 us of GPU time), and it runs on RCCL's
    # stream while the next batch computes into the other buffer.
    NBUF = 2
    K = max(1, args.gather_every) if use_dist else 1
    with torch.cuda.stream(stream_obj):
        local = [torch.zeros((K, 2, P, wmax), dtype=torch.int64, device=dev_t) for _ in range(NBUF)]
        gathered = [torch.zeros((world, K, 2, P, wmax), dtype=torch.int64, device="cpu" if rehearsal else dev_t) for _ in range(NBUF)] if use_dist else None
    stream_obj.synchronize()
    works = [None] * NBUF
    step_no = [0]
    ptrs = [[(t[k, 0].data_ptr(), t[k, 1].data_ptr()) for k in range(K)] for t in local]

    def ship(b):
        # The bitsets of a pass are final only once its counters have been inspected (a bucket overflow replays the pass
        # into the same buffers, include/pcramp_hip.h pcr_screen_device): drain every handle before the collective reads them.
        for s in scrs:
            s.synchronize()
        with torch.cuda.stream(stream_obj):
            if rehearsal:
                dist.all_gather_into_tensor(gathered[b].view(-1), local[b].cpu().view(-1))
            elif comm is not None:
                scrs[0].exchange_bits(comm, local[b].data_ptr(), local[b].numel(), gathered[b].data_ptr())
            else:
                works[b] = dist.all_gather_into_tensor(gathered[b].view(-1), local[b].view(-1), async_op=True)

    # (the pass's call with its ctypes arguments converted once per (target set, buffer slice): what remains per step is the call itself)
    calls = {}

    def step():
        k = step_no[0] % K
        b = (step_no[0] // K) % NBUF
        si = step_no[0] % NS
        scr = scrs[si]
        step_no[0] += 1
        if k == 0 and works[b] is not None:
            works[b].wait()          # the gather that still reads this buffer (two batches ago) is ordered before the new pass
            works[b] = None
        p_fr, p_rf = ptrs[b][k]
        if args.separate_calls:
            scr.select_words(pa, select_thr, 18, args.optimize_shifts, args.optimize_shifts, count=False)
            scr.amplify_device(pa, p_fr, p_rf, thr_t, thr_t, 80, 200, False)
        else:
            # one optimiser iteration's DB build + find_target_match, enqueued without a host wait
            call = calls.get((si, b, k))
            if call is None:
                call = calls[(si, b, k)] = scr.screen_call(pa, select_thr, p_fr, p_rf, thr_t, thr_t, 80, 200, False, 18, args.optimize_shifts, args.optimize_shifts)
            call()
        if use_dist and k == K - 1:
            # the path's only exchange: every rank's [K, 2, P, words] orientation bitsets (31 KB per pass at C2)
            ship(b)

    def drain_works():
        if use_dist and step_no[0] % K != 0:      # a partial batch is still unsent
            ship((step_no[0] // K) % NBUF)
            step_no[0] += K - step_no[0] % K
        for i in range(NBUF):
            if works[i] is not None:
                works[i].wait()
                works[i] = None

    def sync_all():
        for s in scrs:
            s.synchronize()
        drain_works()
        stream_obj.synchronize()
        torch.cuda.synchronize()

    # ---- sharded == unsharded, first pass of the first set, before the clock (rank 0 holds the unsharded set on its own GPU)
    verify = None
    if use_dist and not args.no_verify:
        step_no[0] = 0
        step()                                    # pass 0 -> slice 0 of buffer 0 (target set 0)
        ship(0)
        if works[0] is not None:
            works[0].wait()
            works[0] = None
        stream_obj.synchronize()
        torch.cuda.synchronize()
        if rank == 0:
            g0 = gathered[0][:, 0]                # [world, 2, P, wmax]
            verify = verify_first_pass(api, synth, sets[0], ranges, pa, (select_thr, thr_t), local_rank, g0, stream,
                                       sample_only=(sets[0].L >= 1000000))
        dist.barrier()
        step_no[0] = 0

    for _ in range(args.warmup):
        step()
    sync_all()
    for s in scrs:
        s.profile(4)        # HIP events bracket the scan of every 4th pass: an event between two kernels costs a ~6 us queue bubble
        s.profile_read(reset=True)
    step_no[0] = 0

    def timed_block():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; the max over ranks."""
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync_all()             # inspects the counters of the passes still in flight (replays on bucket overflow), flushes the last batch
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([d], dtype=torch.float64, device="cpu" if rehearsal else dev_t)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        return d

    # A short block (the driver's --steps 20 is 1.5 ms) is a noisy sample and mostly warm-up (three rotating handles, the lean
    # pass starts with a handle's second pass): the block of `steps` steps is repeated until MIN_TIMED_S have been timed and the
    # MEDIAN block is reported.  Every block is bracketed as the contract says; the count is derived from the first block's
    # max-over-ranks time, so all ranks run the same number.
    MIN_TIMED_S, MAX_BLOCKS = 0.3, 400
    block_dts = [timed_block()]
    n_blocks = int(min(MAX_BLOCKS, max(1, np.ceil(MIN_TIMED_S / max(block_dts[0], 1e-6)))))
    for _ in range(n_blocks - 1):
        block_dts.append(timed_block())
    dt = float(np.median(block_dts))
    scan_ms = scan_launches = 0
    for s in scrs:
        a, b = s.profile_read(reset=True)
        scan_ms += a
        scan_launches += b
        s.profile(False)

    last = bits_of(local[0][0])              # the first pass of the first batch buffer
    n_set = int(np.unpackbits((last[0] | last[1]).view(np.uint8)).sum())
    # per-GPU work fixed (weak): every rank screens its block of every pass; strong: the blocks add up to the fixed set
    evals_total = float(P) * T_total * args.steps
    value = evals_total / dt

    if rank == 0:
        kern_s = (scan_ms / 1e3) / max(scan_launches, 1)
        # Algorithmic bytes.  SURVEY.md 8(d) charges ceil(L/2) + 32 + 1/8 bytes per EVALUATION, i.e. the packed target once
        # per pair; the scan reads every target once per pass for all pairs, so the bytes a launch has to move at least once
        # are the packed targets + the oligos + the result bits.  `achieved` / `frac` price THOSE (a fraction of the HBM roof
        # that cannot exceed 1); the per-pair figure is kept beside them under its own name.
        once = T_local * ((L + 1) // 2) + P * 32 + P * T_local / 8.0
        per_pair = float(P) * T_local * ((L + 1) // 2 + 32 + 0.125)
        achieved = once / kern_s / 1e9 if kern_s > 0 else 0.0
        # which kernel is the scan: the second form of the seed filter at the reference's default thresholds; below that (select
        # threshold 0.81: 4-5 mismatching slots, ~160 seed codes per orientation) the first form with tables built on the device
        default_regime = select_thr >= 0.85
        # the third form (k_seed3: the seeds looked up in the set's 9-gram position index) where it applies -- the default; PCRAMP_SEED3=0 or a
        # set it cannot serve keeps the second form (k_seed2: a table probe per position)
        third = default_regime and not args.optimize_shifts and os.environ.get("PCRAMP_SEED3", "1") != "0" and os.environ.get("PCRAMP_IRR_INDEX", "1") != "0" and os.environ.get("PCRAMP_S2DBG", "0") in ("", "0")
        kname = ("k_seed3" if third else "k_seed2") if (default_regime and not args.optimize_shifts) else "k_seed"
        klabel = (("k_seed3 (oligo x window match scan, third form: the pass's 9-gram seeds looked up in the targets' position index, every entry "
                   "carrying the 64 bases around it)" if third else
                   "k_seed2 (seed-filter oligo x window match scan, second form: 9-gram seeds, tables built in LDS)") if (default_regime and not args.optimize_shifts) else
                  "k_seed<true> (seed-filter oligo x window match scan, first form: dense 8-gram tables built by k_seed_tables)")
        suffix = "" if default_regime else "_thr081"
        comparable = args.config == "C2" and args.scale == 1.0 and not args.random_primers and not args.optimize_shifts
        is_kernel = lambda k: k == kname or k.startswith(kname + "<")    # noqa: E731
        traffic = valu = None
        tj = load_profile_json("%s_hbm_traffic%s.json" % (PROFILE_ROUND, suffix)) if comparable else None
        if tj:
            hit = [v for k, v in tj.items() if is_kernel(k)]
            traffic = hit[0] if hit else None
        vj = load_profile_json("%s_valu_pmc%s.json" % (PROFILE_ROUND, suffix)) if comparable else None
        if vj:
            hit = [v for k, v in vj.items() if is_kernel(k)]
            if hit and hit[0].get("SQ_INSTS_VALU"):
                insts = hit[0]["SQ_INSTS_VALU"]
                busy_s = insts * 4.0 / (N_SIMD * VALU_CLOCK_GHZ * 1e9)
                valu = {"wave_instructions_per_launch": insts, "issue_time_us": busy_s * 1e6,
                        "frac": (busy_s / kern_s) if kern_s > 0 else None,
                        "frac_at_2clk": (busy_s / 2 / kern_s) if kern_s > 0 else None,
                        "lds_instructions_per_launch": hit[0].get("SQ_INSTS_LDS"),
                        "lds_bank_conflict_cycles_per_launch": hit[0].get("SQ_LDS_BANK_CONFLICT"),
                        "source": "profiles/%s_valu_pmc%s.json (static: rocprofv3 --pmc pass of this command, committed)" % (PROFILE_ROUND, suffix),
                        "model": "SQ_INSTS_VALU x 4 clk / (1024 SIMDs x 2.4 GHz) / kernel time: 4 clk is what ONE wave's stream sustains per instruction "
                                 "(MI355X_MICROARCH.md, vector-instruction issue cost); a SIMD-32 with several ready waves retires a wave64 instruction in 2 "
                                 "(frac_at_2clk), so the truth lies between; wave_cycles (counters) says where the waves' time goes"}
        # where the wave cycles of the scan kernel go, from the committed counter pass (SQ_WAVE_CYCLES = ACTIVE_INST_ANY + WAIT_INST_ANY
        # + WAIT_ANY, quad-cycles summed over waves): issuing, stalled at issue (pipe busy), parked on s_waitcnt / barriers
        wj = load_profile_json("%s_wave_cycles_pmc%s.json" % (PROFILE_ROUND, suffix)) if comparable else None
        if wj and valu is not None:
            hit = [v for k, v in wj.items() if is_kernel(k)]
            if hit and hit[0].get("SQ_WAVE_CYCLES"):
                h = hit[0]
                wc = h["SQ_WAVE_CYCLES"]
                valu["wave_cycles"] = {k2: (h.get(k1) / wc if h.get(k1) is not None else None) for k1, k2 in (
                    ("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_ACTIVE_INST_VALU", "issuing_valu"), ("SQ_ACTIVE_INST_LDS", "issuing_lds"),
                    ("SQ_WAIT_INST_ANY", "stalled_at_issue"), ("SQ_WAIT_INST_LDS", "stalled_at_issue_lds"), ("SQ_WAIT_ANY", "parked_on_waitcnt"))}
                valu["wave_cycles"]["source"] = "profiles/%s_wave_cycles_pmc%s.json (static)" % (PROFILE_ROUND, suffix)
                valu["wave_cycles"]["note"] = ("fractions of SQ_WAVE_CYCLES; 4 waves per SIMD, so issuing_valu x 4 = the share of the SIMD's time a VALU "
                                               "instruction of one of its waves is in flight (one quad-cycle per wave64 instruction)")
        hbm_frac = achieved / HBM_PEAK_GBPS
        traffic_frac = (traffic / kern_s / 1e9 / HBM_PEAK_GBPS) if (traffic and kern_s > 0) else None
        bound = "valu" if (valu and valu["frac"] is not None and valu["frac"] > max(hbm_frac, traffic_frac or 0.0)) else "hbm"
        shape = ("%s: ONE set of %d targets x %d bases sharded over %d GPU(s)" % (args.config, T_total, L, world)) if strong else \
                ("%s: %d targets x %d bases per GPU (%d in all), %d target sets rotated" % (args.config, sets[0].T_block, L, T_total, NS))
        out = {
            "metric": "primer-pair x target amplification evals/sec",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic" + (" (REHEARSAL: gloo, shared GPU -- not a measurement)" if rehearsal else ""),
            "config": {"workload": "%s, %d primer pairs (18-25 nt), select_words thr %.2f + find_target_match thr %.2f, amplicon 80-200"
                                   % (shape, P, select_thr, thr_t),
                       "targets_this_gpu": T_local, "targets_total": T_total, "target_len": L, "pairs": P,
                       "sharding": "shard_ranges: contiguous target blocks, boundaries multiples of 64, x%d" % world,
                       "exchange": ((("pcr_exchange_bits (ncclAllGather behind the C-ABI, on the handle's stream)" if comm is not None else "all_gather_into_tensor, pipelined")
                                     + " of [%d passes, 2, P, words] u64 per rank; handles drained before each gather" % K) if use_dist else "none"),
                       "sharded_equals_unsharded": verify,
                       "timed_region_s": dt, "timed_blocks": len(block_dts), "timed_total_s": float(sum(block_dts)),
                       "block_ms_min_median_max": [min(block_dts) * 1e3, dt * 1e3, max(block_dts) * 1e3],
                       "timing": "median over the timed blocks of `steps` steps each (a block is repeated until %.1f s have been timed)" % MIN_TIMED_S,
                       "staging": scrs[0].staging_mode() + (" (fused pass without a staging launch from a handle's second consecutive pass on)" if scrs[0].staging_mode() == "lean" else ""),
                       "amplification_calls_set_rank0": n_set},
            "roofline": {"bound": bound,
                         "kernel": klabel,
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_frac, "frac_once_per_pass": hbm_frac,
                         "note": "achieved = bytes one launch must read at least once (packed targets + oligos + result bits) / kernel time "
                                 "(HIP events on the launch stream, every 4th pass); k_seed3 does not read the targets at all but the runs of its seeds in "
                                 "the position index (traffic), so `achieved` is an algorithmic rate, not the kernel's own memory traffic",
                         "algorithmic_bytes_per_launch": once,
                         "algorithmic_per_pair": {"bytes_per_launch": per_pair, "achieved": per_pair / kern_s / 1e9 if kern_s > 0 else None,
                                                  "frac": per_pair / kern_s / 1e9 / HBM_PEAK_GBPS if kern_s > 0 else None,
                                                  "note": "SURVEY 8(d): ceil(L/2)+32+1/8 bytes per evaluation, the target re-charged per pair; the scan "
                                                          "reads a target once for all pairs, so this is not a fraction of any roof"},
                         "valu": valu,
                         "traffic": traffic,
                         "traffic_source": ("profiles/%s_hbm_traffic%s.json (static: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                            "command, FETCH_SIZE x2 for gfx950; committed)" % (PROFILE_ROUND, suffix)) if traffic else None,
                         "traffic_GBps": (traffic / kern_s / 1e9) if (traffic and kern_s > 0) else None,
                         "traffic_frac": traffic_frac,
                         "kernel_ms": kern_s * 1e3, "launches": int(scan_launches)},
        }
        if world == 1 and not strong and not args.no_secondary:
            try:
                out["secondary"] = secondary_figures(api, synth, W, local_rank, stream, wl_single, scrs[0], pa, (select_thr, thr_t))
            except Exception as e:                                     # noqa: BLE001
                out["secondary"] = {"error": str(e)}
            try:
                out["secondary"]["per_set_streams"] = per_set_streams_figure(api, sets, lo, hi, local_rank, pa, (select_thr, thr_t))
            except Exception as e:                                     # noqa: BLE001
                out["secondary"]["per_set_streams"] = {"error": str(e)}
        if world == 1 and not strong and not args.no_cpu_baseline:
            for s in scrs:
                s.close()
            wl = dict(wl_single)
            wl["pairs"] = pairs
            out["cpu_baseline"] = cpu_baseline(wl, thr_t, mult, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
