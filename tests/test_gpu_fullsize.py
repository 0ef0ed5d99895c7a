"""BASELINE.json's full C2 size (10 000 targets x 10 kb, 50 pairs): properties that do not need the CPU
oracle (which would take hours here) -- the seed scan and the bit-sliced scan build the same word DB and
amplification bits, the fused asynchronous pass equals the separate calls, evaluation is independent per
target (a shard screened alone gives the same bits for its targets), a pass is idempotent, and the
coverage checksum is the weighted popcount of the bits.  Run on the GPU box with `-m gpu`."""
import os

import numpy as np
import pytest

from pcramp_amd import api, synth

pytestmark = pytest.mark.gpu


def _screener(scan=None):
    old = os.environ.get("PCRAMP_SCAN")
    if scan is None:
        os.environ.pop("PCRAMP_SCAN", None)
    else:
        os.environ["PCRAMP_SCAN"] = str(scan)
    try:
        return api.Screener(0)
    finally:
        if old is None:
            os.environ.pop("PCRAMP_SCAN", None)
        else:
            os.environ["PCRAMP_SCAN"] = old


@pytest.fixture(scope="module")
def c2():
    return synth.workload("C2", 0, 1.0)


def _screen(d, wl, lo=0, hi=None, thr_t=1.0, mult=0.9):
    hi = wl["T"] if hi is None else hi
    nb = (wl["L"] + 1) // 2
    d.load_sequences(wl["packed"][lo * nb:hi * nb], wl["byte_offsets"][lo:hi] - wl["byte_offsets"][lo], wl["lengths"][lo:hi])
    thr = float(np.float32(thr_t) * np.float32(mult))
    n = d.select_words(wl["pairs"], thr, 18)
    bits, fr, rf, cov = d.amplify(wl["pairs"], thr_t, thr_t, 80, 200, False)
    return n, fr, rf, cov


def test_full_c2_properties(c2):
    import torch
    a, b = _screener(None), _screener(2)
    try:
        n3, fr3, rf3, cov3 = _screen(a, c2)
        n2, fr2, rf2, cov2 = _screen(b, c2)
        # seed scan == bit-sliced scan at full size
        assert n3 == n2 and n3 > 1000
        assert np.array_equal(fr3, fr2) and np.array_equal(rf3, rf2) and np.array_equal(cov3, cov2)
        assert fr3.any() or rf3.any()
        # idempotence
        n3b, fr3b, rf3b, cov3b = _screen(a, c2)
        assert n3b == n3 and np.array_equal(fr3b, fr3) and np.array_equal(rf3b, rf3) and np.array_equal(cov3b, cov3)
        # coverage == weighted popcount of the union of both orientations (all weights 1 here)
        assert np.array_equal(cov3, (fr3 | rf3).sum(axis=1).astype(np.float32))
        # fused asynchronous pass == separate calls
        words = int(a.bitset_words())
        P = len(c2["pairs"])
        out = torch.full((2, P, words), -1, dtype=torch.int64, device="cuda:0")
        thr = float(np.float32(1.0) * np.float32(0.9))
        a.screen_device(c2["pairs"], thr, out[0].data_ptr(), out[1].data_ptr(), 1.0, 1.0, 80, 200, False)
        a.synchronize()
        torch.cuda.synchronize()
        w = out.cpu().numpy().view(np.uint64)
        got = [np.stack([api.bits_to_bool(w[k, i], c2["T"]) for i in range(P)]) for k in range(2)]
        assert np.array_equal(got[0], fr3) and np.array_equal(got[1], rf3)
        # independence per target: a shard screened alone (same pairs) amplifies the same targets.  The word DB
        # couples the pairs of a batch, not the targets (select_words runs per sequence), so the bits are equal.
        lo, hi = 2560, 5120
        _, frs, rfs, _ = _screen(b, c2, lo, hi)
        assert np.array_equal(frs, fr3[:, lo:hi]) and np.array_equal(rfs, rf3[:, lo:hi])
    finally:
        a.close()
        b.close()
