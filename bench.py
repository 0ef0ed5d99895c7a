#!/usr/bin/env python3
"""bench.py -- primer-pair x target amplification evaluations / second on MI355X.

One step = one pass of the hot path over one batch of synthetic input, inputs already resident
in HBM: the per-iteration index build (Sequence::pack + select_words for every target,
reference main.cpp:644-691) followed by the amplicon screen of every primer pair against every
target (PCR::find_target_match, pcr_assay.cpp:544).  Workload at N=1: BASELINE.json configs[1]
("C2": 10 000 viral targets x 10 kb, 50 primer pairs).  With --gpus N each rank owns its own
10 000-target shard (weak scaling; targets shard with no data-path collective except the one
all-gather of the per-target coverage bitsets per pass).

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

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(wl, thr_t, mult, budget_s=15.0):
    """Time the CPU path on a bounded sample of the SAME workload (rank 0, N=1 only).
    Prefers the real reference (oracle/_ref, kind "reference"), else our restatement ("port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = min(16, len(os.sched_getaffinity(0)))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    from oracle_lib import Oracle, Reference
    from pcramp_amd import words as W
    kind = "reference" if Reference.available() else "port"
    lib = Reference() if kind == "reference" else Oracle()
    if kind == "port":
        cores = 1
    nb = (wl["L"] + 1) // 2

    def run(n_t):
        s = lib.session(target_threshold=thr_t, search_multiplier=mult)
        for i in range(n_t):
            o = int(wl["byte_offsets"][i])
            codes = W.unpack_codes(wl["packed"][o:o + nb], wl["L"])
            s.add_target(W.text_from_codes(codes))
        t0 = time.perf_counter()
        s.select(wl["pairs"])
        for p in wl["pairs"]:
            s.target_match(p)
        return time.perf_counter() - t0

    n_t = min(4, wl["T"])
    dt = run(n_t)
    per_target = dt / n_t
    n_big = int(max(n_t, min(wl["T"], budget_s / max(per_target, 1e-9))))
    if n_big > n_t:
        dt = run(n_big)
        n_t = n_big
    evals = n_t * len(wl["pairs"])
    return {"value": evals / dt, "unit": "evals/s", "cores": cores, "kind": kind,
            "sample": "%d of the %d targets (%d bases each) x %d pairs: select_words + find_target_match, %.1f s"
                      % (n_t, wl["T"], wl["L"], len(wl["pairs"]), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the target count (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--random-primers", action="store_true", help="diagnostic: primers unrelated to the targets (no hits)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--gather-every", type=int, default=16,
                    help="N>1: passes per all-gather (their bitsets travel in one collective)")
    ap.add_argument("--optimize-shifts", action="store_true",
                    help="diagnostic: --optimize.5/--optimize.3 of the reference (every 5'/3' slot shift of every oligo is a candidate)")
    ap.add_argument("--separate-calls", action="store_true",
                    help="pcr_select_words + pcr_amplify_device per step (host wait between them) instead of pcr_screen_device")
    ap.add_argument("--target-threshold", type=float, default=1.0,
                    help="--target.threshold of the reference (pcramp.h:39 default 1.0)")
    ap.add_argument("--search-multiplier", type=float, default=0.9,
                    help="target search multiplier of the reference (pcramp.h:51 default 0.9)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pcramp_amd import api, synth, words as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PCRAMP_BENCH_REHEARSAL=1: exercise the N>1 code path on a box with fewer GPUs than ranks -- gloo
    # collectives on host copies, ranks share the visible GPUs.  The number it prints is not a measurement.
    rehearsal = os.environ.get("PCRAMP_BENCH_REHEARSAL") == "1"
    use_dist = world > 1 or os.environ.get("PCRAMP_BENCH_FORCE_DIST") == "1"   # FORCE_DIST: run the collective path with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if rehearsal:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev_t = torch.device("cuda", local_rank)

    thr_t, mult = args.target_threshold, args.search_multiplier
    select_thr = float(np.float32(thr_t) * np.float32(mult))

    # every rank owns a different shard of the same family structure; the primer pairs are
    # replicated (rank 0's), as the reference broadcasts its trial assays
    wl = synth.workload(args.config, seed_offset=rank, scale=args.scale)
    pairs = wl["pairs"]
    if args.random_primers:
        rs = np.random.RandomState(7)
        pairs = [(W.centered_word(2 ** rs.randint(0, 4, size=rs.randint(18, 26))),
                  W.centered_word(2 ** rs.randint(0, 4, size=rs.randint(18, 26)))) for _ in range(len(pairs))]
        wl["pairs"] = pairs
    if world > 1:
        obj = [pairs if rank == 0 else None]
        dist.broadcast_object_list(obj, src=0)
        pairs = obj[0]
        wl["pairs"] = pairs
    pa = W.pairs_array(pairs)
    T, L, P = wl["T"], wl["L"], len(pairs)

    stream = torch.cuda.current_stream().cuda_stream
    scr = api.Screener(local_rank, stream=stream)
    scr.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
    words = int(scr.bitset_words())
    # The exchange is batched and pipelined: K passes write their bitsets into the K slices of one buffer, one
    # all-gather ships the batch (fewer, larger collectives: a torch collective costs the host ~100 us, a pass
    # 150 us of GPU time), and it runs on RCCL's stream while the next batch computes into the other buffer.
    NBUF = 2
    K = max(1, args.gather_every) if use_dist else 1
    local = [torch.zeros((K, 2, P, words), dtype=torch.int64, device=dev_t) for _ in range(NBUF)]
    gathered = [torch.zeros((world, K, 2, P, words), dtype=torch.int64, device="cpu" if rehearsal else dev_t) for _ in range(NBUF)] if use_dist else None
    works = [None] * NBUF
    step_no = [0]

    host_t = [0.0, 0.0]          # PCRAMP_TIMING=1: wall time inside the two ABI calls (diagnostic)
    timing = os.environ.get("PCRAMP_TIMING") == "1"
    ptrs = [[(t[k, 0].data_ptr(), t[k, 1].data_ptr()) for k in range(K)] for t in local]

    def ship(b):
        if rehearsal:
            dist.all_gather_into_tensor(gathered[b].view(-1), local[b].cpu().view(-1))
        else:
            works[b] = dist.all_gather_into_tensor(gathered[b].view(-1), local[b].view(-1), async_op=True)

    def step():
        k = step_no[0] % K
        b = (step_no[0] // K) % NBUF
        step_no[0] += 1
        if k == 0 and works[b] is not None:
            works[b].wait()          # the gather that still reads this buffer (two batches ago) is ordered before the new pass
            works[b] = None
        p_fr, p_rf = ptrs[b][k]
        if timing:
            t_a = time.perf_counter()
        if args.separate_calls:
            scr.select_words(pa, select_thr, 18, args.optimize_shifts, args.optimize_shifts, count=False)
        if timing:
            t_b = time.perf_counter()
        if args.separate_calls:
            scr.amplify_device(pa, p_fr, p_rf, thr_t, thr_t, 80, 200, False)
        else:
            # one optimiser iteration's DB build + find_target_match, enqueued without a host wait
            scr.screen_device(pa, select_thr, p_fr, p_rf, thr_t, thr_t, 80, 200, False, 18, args.optimize_shifts, args.optimize_shifts)
        if timing:
            t_c = time.perf_counter()
            host_t[0] += t_b - t_a
            host_t[1] += t_c - t_b
        if use_dist and k == K - 1:
            # the path's only exchange: every rank's [K, 2, P, words] orientation bitsets (31 KB per pass at C2),
            # ordered behind the screens by torch (the collective waits for the current stream)
            ship(b)

    def drain_works():
        if use_dist and step_no[0] % K != 0:      # a partial batch is still unsent
            ship((step_no[0] // K) % NBUF)
            step_no[0] += K - step_no[0] % K
        for i in range(NBUF):
            if works[i] is not None:
                works[i].wait()
                works[i] = None

    for _ in range(args.warmup):
        step()
    scr.synchronize()
    drain_works()
    torch.cuda.synchronize()
    scr.profile(4)        # HIP events bracket the scan of every 4th pass: an event between two kernels costs a ~6 us queue bubble
    scr.profile_read(reset=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    scr.synchronize()          # inspects the counters of the passes still in flight (replays on bucket overflow)
    drain_works()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    scan_ms, scan_launches = scr.profile_read(reset=True)
    scr.profile(False)
    if timing and rank == 0:
        n_st = args.steps + args.warmup
        sys.stderr.write("[bench] host us/step inside select_words %.1f, amplify_device %.1f; step %.1f\n"
                         % (host_t[0] / n_st * 1e6, host_t[1] / n_st * 1e6, dt / args.steps * 1e6))

    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev_t)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    last = local[0][0]               # the first pass of the first batch buffer (every pass screens the same pairs)
    n_set = int(sum(bin(int(x) & 0xFFFFFFFFFFFFFFFF).count("1") for x in (last[0] | last[1]).flatten().tolist()))
    evals_total = float(P) * T * world * args.steps
    value = evals_total / dt

    if rank == 0:
        # dominant kernel: the oligo x window match scan.  Algorithmic bytes per evaluation
        # (SURVEY.md section 8d): ceil(L/2) packed target bytes + 2 x 16-byte oligos + 1 result bit.
        b_eval = (L + 1) // 2 + 32 + 0.125
        evals_per_launch = float(P) * T
        kern_s = (scan_ms / 1e3) / max(scan_launches, 1)
        achieved = evals_per_launch * b_eval / kern_s / 1e9 if kern_s > 0 else 0.0
        traffic = None
        try:   # HBM bytes per launch of the scan kernel from the committed PMC passes (profiles/summarize.py)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
            key = "k_seed" if select_thr >= 0.85 else "k_scan2"      # which scan the thresholds select (match scan)
            traffic = ([v for k, v in tj.items() if k.startswith(key)][0]
                       if args.config == "C2" and args.scale == 1.0 and not args.random_primers else None)
        except Exception:
            traffic = None
        out = {
            "metric": "primer-pair x target amplification evals/sec",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic" + (" (REHEARSAL: gloo, shared GPU -- not a measurement)" if rehearsal else ""),
            "config": {"workload": "%s: %d targets x %d bases per GPU, %d primer pairs (18-25 nt), "
                                   "select_words thr %.2f + find_target_match thr %.2f, amplicon 80-200"
                                   % (args.config, T, L, P, select_thr, thr_t),
                       "targets_per_gpu": T, "target_len": L, "pairs": P, "sharding": "targets x%d" % world,
                       "exchange": ("all_gather_into_tensor of [%d passes, 2, P, words] u64 per rank, pipelined" % K) if use_dist else "none",
                       "amplification_calls_set_rank0": n_set},
            "roofline": {"bound": "hbm",
                         "kernel": ("k_seed (seed-filter oligo x window match scan)" if select_thr >= 0.85
                                    else "k_scan2 (bit-sliced oligo x window match scan)"),
                         "note": "achieved = SURVEY 8(d) algorithmic bytes (target re-read per pair) / scan time; the scan reads "
                                 "each target once per pass for all pairs, so frac may exceed 1 -- see traffic for measured HBM bytes",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_GBps": (traffic / kern_s / 1e9) if (traffic and kern_s > 0) else None,
                         "traffic_frac": (traffic / kern_s / 1e9 / HBM_PEAK_GBPS) if (traffic and kern_s > 0) else None,
                         "kernel_ms": kern_s * 1e3, "launches": int(scan_launches),
                         "algorithmic_bytes_per_launch": evals_per_launch * b_eval},
        }
        if world == 1 and not args.no_cpu_baseline:
            scr.close()
            out["cpu_baseline"] = cpu_baseline(wl, thr_t, mult, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
