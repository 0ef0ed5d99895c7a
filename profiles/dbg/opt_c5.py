# phase timers of pcr_optimize_batch on C5's shard (GPU box): PCRAMP_TIMING=1 python profiles/dbg/opt_c5.py [n_trial]
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from pcramp_amd import api, synth, moves
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
c5 = synth.workload("C5_shard")
s = api.Screener(0)
s.load_sequences(c5["packed"], c5["byte_offsets"], c5["lengths"])
thr = float(np.float32(1.0) * np.float32(0.9))
trial, _, _ = s.random_assays(2025, n)
s.select_words(trial, thr, 18, count=False)
kw = dict(degen=16, target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200, have_background=False)
moves.optimize_batch(s, trial, **kw)
t0 = time.perf_counter()
_, _, it = moves.optimize_batch(s, trial, **kw)
print("batch", n, "ms", (time.perf_counter() - t0) * 1e3, "iters max", max(it), "mean", sum(it) / len(it))
s.close()
