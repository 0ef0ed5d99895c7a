"""pcr_design on the C2 targets with the reference's default options (1 000 trial assays per iteration): time per design iteration.
PCRAMP_TIMING=1 prints the phases of every iteration.  usage: design_c2.py [iterations] [trials]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcramp_amd import api, synth, design
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_trial = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
wl = synth.workload("C2")
d = api.Screener(0)
d.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
T = int(wl["T"])
t0 = time.perf_counter()
text, pool = design.design(d, [">t%d" % i for i in range(T)], [int(x) for x in wl["lengths"]], argv=["pcramp"], num_assay=n_it, num_trial=n_trial, seed=42)
dt = time.perf_counter() - t0
print("design: %d iterations x %d trials on %d targets: %.1f ms per iteration, %d assays, %d bytes" % (n_it, n_trial, T, dt / n_it * 1e3, len(pool), len(text)))
d.close()
