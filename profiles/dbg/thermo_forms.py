import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, ctypes as C
from pcramp_amd import api, words as W
rs = np.random.RandomState(1)
def rand_word(k): return W.centered_word(2 ** rs.randint(0, 4, size=k).astype(np.uint8))
for n in (20000, 100000, 400000):
    ol = np.array([list(rand_word(rs.randint(18, 26))) for _ in range(n)], dtype=np.uint64)
    for form in ("wave", "lane"):
        os.environ["PCRAMP_THERMO_WAVE_MAX"] = "100000000" if form == "wave" else "0"
        s = api.Screener(0)
        res = (api.ThermoResult * n)()
        args = s._targs(0.05, 9e-7, 50.0, 75.0, 40.0, 40.0)
        s._check(s.L.pcr_thermo(s.h, ol.ctypes.data, 256, 1, C.byref(args), res))
        s.profile(1); s.profile_read_kernel(2)
        t0 = time.perf_counter()
        s._check(s.L.pcr_thermo(s.h, ol.ctypes.data, n, 1, C.byref(args), res))
        dt = time.perf_counter() - t0
        kms, kn = s.profile_read_kernel(2)
        v = np.frombuffer(res, dtype=np.uint32).reshape(-1, 8)[:, 0].sum()
        print(n, form, "abi ms %.2f kernel ms %.2f launches %d valid %d" % (dt * 1e3, kms, kn, v))
        s.close()
