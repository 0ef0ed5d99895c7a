"""pcr_exchange_bits: the path's one exchange step behind the C-ABI (RCCL all-gather on the handle's stream; it replaces the
reference's gather + broadcast of the winner's BitSets, main.cpp:1421-1601).  A one-GPU box can only form a one-rank
communicator (RCCL refuses two ranks on one device), so this checks the plumbing on hardware -- RCCL bound at run time, the
collective enqueued in stream order behind a fused pass, gathered words == local words, coverage from the gathered words ==
compute_coverage -- while the sharding arithmetic of N > 1 (rank-major blocks of whole words, re-summed coverage) is covered
by tests/test_sharding_gloo.py on CPU."""
import numpy as np
import pytest

from pcramp_amd import api, synth

pytestmark = pytest.mark.gpu


def test_exchange_bits_single_rank():
    import torch
    wl = synth.workload("C1")
    pairs = wl["pairs"]
    stream = torch.cuda.Stream(device="cuda:0")
    d = api.Screener(0, stream=stream.cuda_stream)
    try:
        d.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
        words = int(d.bitset_words())
        P = len(pairs)
        comm = d.comm_init_rank(api.Screener.comm_unique_id(), 1, 0)
        assert d.L.pcr_comm_world(comm) == 1 and d.L.pcr_comm_rank(comm) == 0
        assert d.L.pcr_comm_library()                               # which librccl was bound
        with torch.cuda.stream(stream):
            local = torch.full((2, P, words), -1, dtype=torch.int64, device="cuda:0")
            full = torch.full((1, 2, P, words), -1, dtype=torch.int64, device="cuda:0")
        stream.synchronize()
        thr = float(np.float32(1.0) * np.float32(0.9))
        for _ in range(3):                                          # the third pass is a lean one
            d.screen_device(pairs, thr, local[0].data_ptr(), local[1].data_ptr(), 1.0, 1.0, 80, 200, False)
            d.exchange_bits(comm, local.data_ptr(), 2 * P * words, full.data_ptr())
        d.synchronize()
        stream.synchronize()
        lw, fw = local.cpu().numpy().view(np.uint64), full.cpu().numpy().view(np.uint64)[0]
        assert np.array_equal(lw, fw)
        bits = d.find_target_match(pairs, 1.0)
        got = np.stack([api.bits_to_bool(fw[0, i] | fw[1, i], wl["T"]) for i in range(P)])
        assert np.array_equal(got, bits) and bits.any()
        w = np.ones(wl["T"], dtype=np.float32)
        cov = d.compute_coverage(pairs, 1.0, 1.0)
        for i in range(P):
            assert api.coverage_from_bits(fw[0, i], fw[1, i], w) == cov[i]
        d.comm_destroy(comm)
    finally:
        d.close()
