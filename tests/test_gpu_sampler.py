"""pcr_random_assays (PCR::random_assay, pcr_assay.cpp:580-734, on a running glibc rand_r state) against the
oracle: the same assays in the same order and the same state afterwards.  Run on the GPU box with `-m gpu`."""
import random

import pytest

from pcramp_amd import api, words as W
from testdata import family_targets, rand_seq, revcomp

pytestmark = pytest.mark.gpu


def _load(oracle, seqs, active, splits):
    so = oracle.session()
    for q, a in zip(seqs, active):
        so.add_target(q, 1.0, a)
    d = api.Screener(0)
    d.load_texts(seqs, [1.0] * len(seqs))
    d.set_active(active)
    for i, pos in splits:
        so.split(i, pos)
        d.split(i, pos)
    return so, d


@pytest.mark.parametrize("case", [dict(), dict(max_degen=8.0, tm_min=40.0, tm_max=80.0),
                                  dict(tm_min=57.5, tm_max=60.5, max_hairpin=30.0, max_dimer=25.0),
                                  dict(primer_min=20, primer_max=32, amp_min=70, amp_max=90, tm_min=55.0, tm_max=90.0),
                                  dict(primer_min=18, primer_max=22, amp_min=38, amp_max=60, tm_min=40.0, tm_max=80.0)])
def test_random_assays_match_oracle(oracle, case):
    from oracle_lib import random_assays
    rng = random.Random(811 + len(case))
    if case.get("amp_min") == 38:
        seqs = [rand_seq(rng, n) for n in (40, 43, 52, 66, 90)]
        active, splits = [True] * 5, []
    else:
        seqs = family_targets(rng, 4, 6, 1000, div=0.05) + [rand_seq(rng, 257)]
        q = list(seqs[5])
        for k in range(100, 500, 11):
            q[k] = rng.choice("RYKMSWBDN")
        seqs[5] = "".join(q)
        active = [i not in (2, 9) for i in range(len(seqs))]
        splits = [(1, 500), (7, 80), (7, 640), (12, 333)]
    so, d = _load(oracle, seqs, active, splits)
    try:
        state = 20240 + len(case)
        for _ in range(6):                                             # the state runs on from call to call
            want, after = random_assays(oracle, so, state, 25, **case)
            got, s, info = d.random_assays(state, 25, **case)
            assert got == want
            assert s == after
            for (f, r), i in zip(got, info):                           # the reported origin spells the assay
                assert active[i["sequence"]]
                t = seqs[i["sequence"]]
                fl, rl = len(W.word_text(f)), len(W.word_text(r))
                assert W.word_text(f) == t[i["f_start"]:i["f_start"] + fl]
                end = i["f_start"] + i["amplicon_length"]
                assert W.word_text(r) == revcomp(t[end - rl:end])
                assert 1 <= i["assay_iterations"] <= 100 and 1 <= i["sequence_iterations"] <= 100
            state = s
    finally:
        d.close()


def test_random_assays_thread_seed_protocol(oracle):
    """main.cpp:538-550 at one thread: local_seed = rand_r(&global_seed), then num_trial assays on it."""
    from oracle_lib import random_assays, rand_r
    rng = random.Random(3)
    seqs = family_targets(rng, 2, 5, 600, div=0.04)
    so, d = _load(oracle, seqs, [True] * len(seqs), [])
    try:
        g = 1234567
        for _ in range(3):                                             # three design iterations
            local, g2 = api.host_rand_r(g)
            assert (local, g2) == rand_r(oracle, g)
            got, _, _ = d.random_assays(local, 40)
            want, _ = random_assays(oracle, so, local, 40)
            assert got == want
            g = g2
    finally:
        d.close()


def test_random_assay_errors():
    d = api.Screener(0)
    try:
        d.load_texts(["ACGT" * 10], [1.0])
        with pytest.raises(api.PcrError, match="sequence length is too small"):
            d.random_assays(1, 1)
        d.load_texts(["ACGT" * 100], [1.0])
        d.set_active([False])
        with pytest.raises(api.PcrError, match="No active sequences"):
            d.random_assays(1, 1)
        d.load_texts(["A" * 400], [1.0])
        with pytest.raises(api.PcrError, match="Unable to generate"):
            d.random_assays(1, 1)
    finally:
        d.close()


def test_sampler_at_c2_scale(oracle, capsys):
    """1 000 trials on the 10 000 x 10 kb bench targets: identical to the oracle; both times are printed
    (`pytest -s`) -- the figures DESIGN.md quotes."""
    import time
    from oracle_lib import random_assays
    from pcramp_amd import synth
    wl = synth.workload("C2", 0, 1.0)
    so = oracle.session()
    for i in range(wl["T"]):
        lo = int(wl["byte_offsets"][i])
        so.add_target_packed(wl["packed"][lo:lo + (int(wl["lengths"][i]) + 1) // 2], int(wl["lengths"][i]))
    d = api.Screener(0)
    try:
        d.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"])
        d.random_assays(1, 20)
        t0 = time.perf_counter()
        got, s, _ = d.random_assays(7, 1000)
        t_dev = time.perf_counter() - t0
        t0 = time.perf_counter()
        want, after = random_assays(oracle, so, 7, 1000)
        t_cpu = time.perf_counter() - t0
        assert got == want and s == after
        with capsys.disabled():
            print("\n[sampler, C2 targets, 1000 trials] device path %.3f s, CPU oracle %.3f s" % (t_dev, t_cpu))
    finally:
        d.close()
