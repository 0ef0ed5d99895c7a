# soak of the lean fused pass (GPU box): many passes over rotating target sets and changing primer batches, the bits of every
# 500th pass compared with the synchronous path on a second handle that runs the bit-sliced scan (PCRAMP_SCAN=2)
import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pcramp_amd import api, synth, words as W
n_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
sets = [synth.workload("C2", k, 0.2) for k in range(3)]          # 2 000 targets each
devs, refs = [], []
for wl in sets:
    d = api.Screener(0); d.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"]); devs.append(d)
    os.environ["PCRAMP_SCAN"] = "2"; r = api.Screener(0); os.environ.pop("PCRAMP_SCAN")     # the checker counts every window (bit-sliced scan), no seeds, no index
    r.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"]); refs.append(r)
batches = [sets[k]["pairs"][i:i + 10] for k in range(3) for i in (0, 10, 20, 30, 40)]
thr = float(np.float32(1.0) * np.float32(0.9))
words = int(devs[0].bitset_words())
outs = [torch.zeros((2, 10, words), dtype=torch.int64, device="cuda:0") for _ in range(4)]
bad = 0
t0 = time.perf_counter()
for i in range(n_pass):
    k = i % 3
    b = batches[(i * 7) % len(batches)]
    o = outs[i % 4]
    devs[k].screen_device(b, thr, o[0].data_ptr(), o[1].data_ptr(), 1.0, 1.0, 80, 200, False)
    if i % 500 == 499:
        devs[k].synchronize(); torch.cuda.synchronize()
        w = o.cpu().numpy().view(np.uint64)
        refs[k].select_words(b, thr, 18)
        _, fr, rf, _ = refs[k].amplify(b, 1.0, 1.0, 80, 200, False)
        T = sets[k]["T"]
        got_fr = np.stack([api.bits_to_bool(w[0, j], T) for j in range(10)]); got_rf = np.stack([api.bits_to_bool(w[1, j], T) for j in range(10)])
        if not (np.array_equal(got_fr, np.array(fr)) and np.array_equal(got_rf, np.array(rf))): bad += 1
for d in devs: d.synchronize()
torch.cuda.synchronize()
print("passes", n_pass, "checked", n_pass // 500, "mismatches", bad, "time %.2f s" % (time.perf_counter() - t0))
for d in devs + refs: d.close()
sys.exit(1 if bad else 0)
