"""C3's background path for a kernel-level profile (run under rocprofv3 --kernel-trace --stats on the GPU box):
select_words on the 10 000 backgrounds at 0.8 x 0.9 = 0.72 (min length 16) + find_background_match, a few repetitions."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcramp_amd import api, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
c3 = synth.workload("C3")
bg = c3["background"]
s = api.Screener(0)
s.load_sequences(bg["packed"], bg["byte_offsets"], bg["lengths"], which=api.BACKGROUND)
bthr = float(np.float32(0.8) * np.float32(0.9))
p3 = c3["pairs"]
s.select_words(p3, bthr, 16, which=api.BACKGROUND, count=False)
s.find_background_match(p3, 0.8, 0.9, 0, 2000, False)
s.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    s.select_words(p3, bthr, 16, which=api.BACKGROUND, count=False)
s.synchronize()
t1 = time.perf_counter()
for _ in range(reps):
    s.find_background_match(p3, 0.8, 0.9, 0, 2000, False)
t2 = time.perf_counter()
print("select_words %.3f ms   find_background_match %.3f ms" % ((t1 - t0) / reps * 1e3, (t2 - t1) / reps * 1e3))
s.close()
