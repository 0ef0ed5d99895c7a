"""The seed scan (default match scan of pcr_select_words) against the bit-sliced scan and the
oracle: tile borders, many candidates, duplicated primers, IUPAC on either side, dense hits.
The word DB must be identical entry for entry.  Run on the GPU box with `-m gpu`."""
import os
import random

import numpy as np
import pytest

from pcramp_amd import api, words as W
from testdata import rand_seq, revcomp, mutate

pytestmark = pytest.mark.gpu


def _screener(scan):
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
def both():
    a, b = _screener(None), _screener(2)
    yield a, b
    a.close()
    b.close()


def _entries(dev, seqs, pairs, thr, opt5=0, opt3=0, min_len=18):
    dev.load_texts(seqs, [1.0] * len(seqs))
    n = dev.select_words(pairs, thr, min_len, opt5, opt3)
    e = dev.entries()
    assert len(e) == n
    return e


def _entries_only(dev, pairs, thr, min_len=18):
    n = dev.select_words(pairs, thr, min_len, 0, 0)
    e = dev.entries()
    assert len(e) == n
    return e


def _oracle_entries(oracle, seqs, pairs, thr_t, mult, opt5=0, opt3=0):
    so = oracle.session(target_threshold=thr_t, search_multiplier=mult, amp_min=80, amp_max=200, use_taq_mama=0,
                        pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0, min_primer=18, optimize_5=opt5, optimize_3=opt3)
    for s in seqs:
        so.add_target(s, 1.0)
    so.select(pairs)
    return so.db_entries()


def _border_case(rng, oracle):
    """Primer sites planted so that windows start just before / on / after every kind of tile border."""
    seqs, pairs = [], []
    for L in (1023, 1024, 1025, 1055, 1056, 1057, 2047, 2048, 2080, 3100, 5000):
        s = list(rand_seq(rng, L))
        seqs.append("".join(s))
    base = seqs[-1]
    for start in (0, 1, 990, 1000, 1003, 1010, 1020, 1023, 1024, 1025, 1040, 2030, 2047, 2048, 2050, 4960, 4975):
        ln = rng.randint(18, 25)
        if start + ln + 150 > len(base):
            f = base[start - 150:start - 150 + 20]
            r = revcomp(base[start:start + ln])
        else:
            f = base[start:start + ln]
            r = revcomp(base[start + 100:start + 100 + rng.randint(18, 25)])
        pairs.append((oracle.centered_word(f), oracle.centered_word(r)))
    # sites at the 3' end of every sequence: the last regular window (ends 6 bases before the end for a
    # centred 20-mer) and the very end (irregular words); lengths chosen so that the last window start
    # falls just before / on / after a tile border
    for sq in seqs[:-1]:
        L = len(sq)
        pairs.append((oracle.centered_word(sq[L - 150:L - 130]), oracle.centered_word(revcomp(sq[L - 26:L - 6]))))
        pairs.append((oracle.centered_word(sq[L - 26:L - 6]), oracle.centered_word(revcomp(sq[L - 20:]))))
        pairs.append((oracle.centered_word(sq[L - 140:L - 118]), oracle.centered_word(revcomp(sq[L - 27:L - 5]))))
    # near-copies of the long sequence: sites with 1-3 substitutions
    for _ in range(4):
        seqs.append(mutate(rng, base, 0.03))
    return seqs, pairs


@pytest.mark.parametrize("thr_t,mult", [(1.0, 0.9), (0.95, 0.9), (1.0, 0.95), (0.9, 0.9), (0.85, 0.9), (0.8, 0.9)])
def test_tile_borders(both, oracle, thr_t, mult):
    rng = random.Random(991)
    seqs, pairs = _border_case(rng, oracle)
    thr = float(np.float32(thr_t) * np.float32(mult))
    e3 = _entries(both[0], seqs, pairs, thr)
    e2 = _entries(both[1], seqs, pairs, thr)
    assert e3 == e2
    assert e3 == _oracle_entries(oracle, seqs, pairs, thr_t, mult)
    assert len(e3) > 0


def test_many_candidates_and_duplicates(both, oracle):
    """5'/3' shift candidates (> 256 candidates: planes read from global memory instead of LDS) and the same
    primer listed several times (several seeds per q-gram code)."""
    rng = random.Random(77)
    root = rand_seq(rng, 2600)
    seqs = [root] + [mutate(rng, root, 0.03) for _ in range(9)] + [rand_seq(rng, 1500) for _ in range(3)]
    pairs = []
    for i in range(24):
        a = rng.randrange(0, 2300)
        f = root[a:a + rng.randint(18, 25)]
        r = revcomp(root[a + 120:a + 120 + rng.randint(18, 25)])
        pairs.append((oracle.centered_word(f), oracle.centered_word(r)))
    pairs = pairs + pairs[:6] + pairs[:3]
    thr = float(np.float32(1.0) * np.float32(0.9))
    n_cand = len(api.host_candidates(pairs, True, True, thr)[1])
    assert n_cand > 256
    e3 = _entries(both[0], seqs, pairs, thr, 1, 1)
    e2 = _entries(both[1], seqs, pairs, thr, 1, 1)
    assert e3 == e2
    assert e3 == _oracle_entries(oracle, seqs, pairs, 1.0, 0.9, 1, 1)


@pytest.mark.parametrize("n_pairs,lens", [(128, (18, 19)), (300, (18, 25))])
def test_dense_seed_lists_are_demoted(both, oracle, n_pairs, lens):
    """Select threshold 0.81 with hundreds of orientations: the seed lists outgrow the tables (128 pairs of 18-19-mers: 256
    orientations x ~270 codes > 40 960 seeds for k_seed_tables, the longest lists go to the bit-sliced scan; 300 pairs = 1 200
    orientations: host-built tables, > 32 768 distinct codes, a quarter of the orientations demoted per round) -- the word DB
    must not depend on who scans what."""
    rng = random.Random(4100 + n_pairs)
    root = rand_seq(rng, 2400)
    seqs = [root] + [mutate(rng, root, 0.04) for _ in range(5)] + [rand_seq(rng, 1200) for _ in range(2)]
    pairs = []
    for i in range(n_pairs):
        a = rng.randrange(0, 2100)
        f = root[a:a + rng.randint(*lens)]
        r = revcomp(root[a + 110:a + 110 + rng.randint(*lens)])
        pairs.append((oracle.centered_word(f), oracle.centered_word(r)))
    thr = float(np.float32(0.9) * np.float32(0.9))
    e3 = _entries(both[0], seqs, pairs, thr)
    e2 = _entries(both[1], seqs, pairs, thr)
    assert e3 == e2 and len(e3) > 100
    if n_pairs == 128:
        assert e3 == _oracle_entries(oracle, seqs, pairs, 0.9, 0.9)


def test_iupac_both_sides(both, oracle):
    rng = random.Random(5)
    root = rand_seq(rng, 3000)
    seqs = [root]
    for i in range(6):
        s = list(mutate(rng, root, 0.02))
        for _ in range(i * 3):                                   # tiles with and without IUPAC target codes
            s[rng.randrange(len(s))] = rng.choice("RYKMSWN")
        seqs.append("".join(s))
    pairs = []
    for i in range(16):
        a = rng.randrange(0, 2700)
        f, r = list(root[a:a + rng.randint(18, 25)]), list(revcomp(root[a + 110:a + 110 + rng.randint(18, 25)]))
        for o in (f, r):
            for _ in range(rng.choice([0, 1, 2, 4])):
                o[rng.randrange(len(o))] = rng.choice("RYKMSWBDHVN")
        pairs.append((oracle.centered_word("".join(f)), oracle.centered_word("".join(r))))
    for thr_t in (1.0, 0.95):
        thr = float(np.float32(thr_t) * np.float32(0.9))
        e3 = _entries(both[0], seqs, pairs, thr)
        e2 = _entries(both[1], seqs, pairs, thr)
        assert e3 == e2
        assert e3 == _oracle_entries(oracle, seqs, pairs, thr_t, 0.9)


def test_iupac_primers_in_two_seed_groups(both, oracle):
    """60 primer pairs (240 orientations) with 2-4 two-fold IUPAC positions each (C5's kind): their 9-gram seeds (> 12 288) do not fit one launch of the
    second form, which then takes the pass in two groups of whole orientations; targets with IUPAC tiles and irregular words
    holding IUPAC codes (those meet every candidate once, in the first launch)."""
    rng = random.Random(505)
    root = rand_seq(rng, 2600)
    seqs = [root]
    for i in range(5):
        t = list(mutate(rng, root, 0.03))
        for _ in range(2 * i):
            t[rng.randrange(len(t))] = rng.choice("RYKMSWN")
        if i == 4:
            t[5] = "R"                                             # an IUPAC code inside the head partial words
            t[len(t) - 7] = "Y"
        seqs.append("".join(t))
    seqs.append(rand_seq(rng, 900))
    pairs = []
    for i in range(60):
        a = rng.randrange(0, 2300)
        f, r = list(root[a:a + rng.randint(18, 25)]), list(revcomp(root[a + 110:a + 110 + rng.randint(18, 25)]))
        for o in (f, r):
            for _ in range(rng.randint(2, 4)):
                o[rng.randrange(len(o))] = rng.choice("RYKMSW")
        pairs.append((oracle.centered_word("".join(f)), oracle.centered_word("".join(r))))
    thr = float(np.float32(1.0) * np.float32(0.9))
    n9 = 0
    for f, r in pairs:
        for w in (f, r):
            floor = int(np.float32(sum(1 for c in W.slots_from_word(w) if c)) * np.float32(thr))
            sd = api.host_orientation_seeds(w, floor)
            n9 += 2 * len(sd) if sd is not None else 0            # (8-gram lists; the 9-gram ones are longer)
    assert n9 > 12288
    e3 = _entries(both[0], seqs, pairs, thr)
    e2 = _entries(both[1], seqs, pairs, thr)
    assert e3 == e2 and len(e3) > 50
    assert e3 == _oracle_entries(oracle, seqs, pairs, 1.0, 0.9)


def test_dense_hits_low_complexity(both, oracle):
    """Low-complexity targets: almost every position is a seed hit and very many windows tie at the maximum
    (bucket growth, multi-seed codes, verification in every lane)."""
    rng = random.Random(12)
    seqs = ["A" * 700 + rand_seq(rng, 300) + "AC" * 300, "ACGT" * 400, "A" * 40 + "C" + "A" * 500, rand_seq(rng, 1200)]
    txt = [("A" * 20, "T" * 20), ("AC" * 10, "GT" * 10), ("ACGT" * 5, "ACGT" * 5), ("A" * 19 + "C", "T" * 18)]
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in txt]
    thr = float(np.float32(1.0) * np.float32(0.9))
    e3 = _entries(both[0], seqs, pairs, thr)
    e2 = _entries(both[1], seqs, pairs, thr)
    assert e3 == e2
    assert e3 == _oracle_entries(oracle, seqs, pairs, 1.0, 0.9)
    assert len(e3) > 1000


def test_bench_shape_sample(both):
    """A slice of the bench workload (families of 50 at 3 %): seed scan == bit-sliced scan at full length."""
    from pcramp_amd import synth
    wl = synth.workload("C2", 0, 0.03)
    thr = float(np.float32(1.0) * np.float32(0.9))
    out = []
    for d in both:
        d.load_sequences(wl["packed"], wl["byte_offsets"], wl["lengths"], np.ones(wl["T"], np.float32))
        n = d.select_words(wl["pairs"], thr, 18)
        out.append((n, d.entries()))
    assert out[0] == out[1]
    assert out[0][0] > 0


# ---------------------------------------------------------------------------- asynchronous fused entry
def _to_bool(t, n):
    """[2, P, words] int64 device tensor -> two [P, n] bool arrays."""
    a = t.cpu().numpy().view(np.uint64)
    return [np.stack([api.bits_to_bool(a[k, i], n) for i in range(a.shape[1])]) for k in range(2)]


def _bits_sync(dev, pairs, thr, thr_t):
    dev.select_words(pairs, thr, 18)
    _, fr, rf, _ = dev.amplify(pairs, thr_t, thr_t, 80, 200, False)
    return np.array(fr), np.array(rf)


def test_screen_device_equals_separate_calls(both, oracle):
    """pcr_screen_device (select + amplify enqueued without a host wait, several passes in flight) leaves the
    same orientation bitsets as pcr_select_words + pcr_amplify, pass by pass."""
    import torch
    rng = random.Random(31)
    root = rand_seq(rng, 3000)
    seqs = [root] + [mutate(rng, root, 0.03) for _ in range(40)] + [rand_seq(rng, 2000) for _ in range(5)]
    batches = []
    for b in range(5):
        pairs = []
        for i in range(10):
            a = rng.randrange(0, 2600)
            f = root[a:a + rng.randint(18, 25)]
            r = revcomp(root[a + 100:a + 100 + rng.randint(18, 25)])
            pairs.append((oracle.centered_word(f), oracle.centered_word(r)))
        batches.append(pairs)
    dev = both[0]
    dev.load_texts(seqs, [1.0] * len(seqs))
    thr_t = 1.0
    thr = float(np.float32(thr_t) * np.float32(0.9))
    want = [_bits_sync(dev, p, thr, thr_t) for p in batches]
    assert sum(int(np.count_nonzero(w[0])) + int(np.count_nonzero(w[1])) for w in want) > 0
    words = int(dev.bitset_words())
    outs = [torch.full((2, len(p), words), -1, dtype=torch.int64, device="cuda:0") for p in batches]
    for p, o in zip(batches, outs):
        dev.screen_device(p, thr, o[0].data_ptr(), o[1].data_ptr(), thr_t, thr_t, 80, 200, False)
    dev.synchronize()
    torch.cuda.synchronize()
    for w, o in zip(want, outs):
        got = _to_bool(o, len(seqs))
        assert np.array_equal(got[0], w[0])
        assert np.array_equal(got[1], w[1])


def _screener_env(**env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return api.Screener(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("thr_t", [1.0, 0.95])
def test_three_forms_of_the_seed_pass_agree(oracle, thr_t):
    """The default pass (third form: position index), the scanning form (PCRAMP_SEED3=0: k_seed2 with the
    inverse index of the irregular words) and the scanning form with the irregular words probed in chunks
    (PCRAMP_IRR_INDEX=0) give the same word DB entry for entry, on the border case (primer sites at every
    kind of tile border and at the sequence ends) and after EOS splits -- and all equal the oracle."""
    import torch
    rng = random.Random(4242)
    seqs, pairs = _border_case(rng, oracle)
    thr = float(np.float32(thr_t) * np.float32(0.9))
    devs = [api.Screener(0), _screener_env(PCRAMP_SEED3=0), _screener_env(PCRAMP_IRR_INDEX=0), _screener_env(PCRAMP_SCAN=2)]
    try:
        want = _oracle_entries(oracle, seqs, pairs, thr_t, 0.9)
        got = [_entries(d, seqs, pairs, thr) for d in devs]
        assert len(want) > 0
        for g in got:
            assert g == want
        # the fused pass (where the third form is the default) after two EOS splits: same bitsets from all three
        outs = []
        active = np.array([(i % 3) != 1 for i in range(len(seqs))], dtype=np.uint8)     # (the third form keeps the flag in its per-block words)
        for d in devs:
            d.split(len(seqs) - 1, 2500)
            d.split(3, 1030)
            d.set_active(active)
            words = int(d.bitset_words())
            o = torch.full((2, len(pairs), words), -1, dtype=torch.int64, device="cuda:0")
            for _ in range(3):      # the lean pass starts with the second consecutive fused pass
                d.screen_device(pairs, thr, o[0].data_ptr(), o[1].data_ptr(), thr_t, thr_t, 80, 200, False)
            d.synchronize()
            torch.cuda.synchronize()
            outs.append(o.cpu().numpy().copy())
        for o in outs[1:]:
            assert np.array_equal(outs[0], o)
        e = [d.entries() for d in devs]
        assert e[0] == e[1] == e[2] == e[3] and len(e[0]) > 0
        # ... and all of them active again: the flag is taken back
        for d in devs:
            d.set_active(np.ones(len(seqs), np.uint8))
        assert [_entries_only(d, pairs, thr) for d in devs[1:]] == [_entries_only(devs[0], pairs, thr)] * 3
    finally:
        for d in devs:
            d.close()


def _random_case(rng, oracle):
    """A small random screen: a few sequence families (some with IUPAC codes, some with EOS inside), primers cut from them
    (some mutated, some with IUPAC positions, some unrelated), a select threshold between 0.81 and 1."""
    seqs = []
    for _ in range(rng.randint(1, 4)):
        L = rng.choice((40, 90, 333, 1024, 1500, 2600))
        root = rand_seq(rng, L, p_degen=rng.choice((0.0, 0.0, 0.002)), p_n=rng.choice((0.0, 0.0, 0.001)))
        seqs.append(root)
        for _ in range(rng.randint(0, 6)):
            m = mutate(rng, root, rng.choice((0.0, 0.01, 0.04)))
            if rng.random() < 0.15 and L > 200:
                cut = rng.randrange(50, L - 50)
                m = m[:cut] + "-" + m[cut + 1:]
            seqs.append(m[:rng.randint(max(33, L - 40), L)] if rng.random() < 0.3 else m)
    pairs = []
    plain = [q for q in seqs if len(q) >= 260]
    for _ in range(rng.randint(1, 30)):
        def oligo():
            n = rng.randint(18, 25)
            if plain and rng.random() < 0.85:
                q = rng.choice(plain)
                a = rng.randrange(0, len(q) - n)
                o = "".join(c if c in "ACGT" else "A" for c in q[a:a + n])
                if rng.random() < 0.3:
                    o = mutate(rng, o, 0.06)
                if rng.random() < 0.25:
                    o = list(o)
                    for _k in range(rng.randint(1, 3)):
                        o[rng.randrange(n)] = rng.choice("MRSWYKN")
                    o = "".join(o)
                return o if rng.random() < 0.5 else revcomp(o)
            return rand_seq(rng, n)
        pairs.append((oracle.centered_word(oligo()), oracle.centered_word(oligo())))
    thr_t = rng.choice((1.0, 1.0, 0.95, 0.9))
    return seqs, pairs, float(np.float32(thr_t) * np.float32(0.9))


def test_random_screens_against_the_bitsliced_scan(both, oracle):
    """Differential test of the default pass (third form where it applies, the second and first forms elsewhere) against the
    bit-sliced scan over random small screens, with random inactive sequences and EOS splits (PCRAMP_DIFF_CASES=n for a longer run)."""
    rng = random.Random(20261005)
    n_cases = int(os.environ.get("PCRAMP_DIFF_CASES", "200"))
    nonempty = 0
    for case in range(n_cases):
        seqs, pairs, thr = _random_case(rng, oracle)
        e = [_entries(d, seqs, pairs, thr) for d in both]
        assert e[0] == e[1], "case %d" % case
        nonempty += bool(e[0])
        if rng.random() < 0.5:
            act = np.array([rng.random() < 0.7 for _ in seqs], dtype=np.uint8)
            cut = len(seqs[0]) > 200 and rng.random() < 0.5
            for d in both:
                d.set_active(act)
                if cut:
                    d.split(0, 100)
            e = [_entries_only(d, pairs, thr) for d in both]
            assert e[0] == e[1], "case %d (inactive sequences / split)" % case
    assert nonempty > n_cases // 2


def test_lean_passes_leave_a_consistent_state(oracle):
    """From the second fused pass over a set on, the pass is 'lean': no staging launch (tables written into device memory by the
    host, control block left clean by the previous pass's tail, result bitsets cleared by the scan).  Batches that hit different
    families alternate, so sequences with entries in one pass have none in the next; after every pass the word DB, the bits
    of a synchronous amplify over that DB (touched list rebuilt from the segment ends) and the amplicons equal what the
    synchronous path gives for the same batch."""
    import torch
    rng = random.Random(77031)
    roots = [rand_seq(rng, 2400) for _ in range(3)]
    seqs = []
    for r in roots:
        seqs += [r] + [mutate(rng, r, 0.03) for _ in range(12)]
    seqs += [rand_seq(rng, 1500) for _ in range(4)]
    def batch(root, k):
        out = []
        for i in range(k):
            a = rng.randrange(0, 2000)
            out.append((oracle.centered_word(root[a:a + rng.randint(18, 25)]), oracle.centered_word(revcomp(root[a + 100:a + 100 + rng.randint(18, 25)]))))
        return out
    # (even batch sizes: with 43 sequences a pair's bitset is 8 bytes, and the fused tail wants 16-byte multiples; the odd batch
    # at the end takes the unfused route in between, after which the next pass is not lean either)
    batches = [batch(roots[0], 6), batch(roots[1], 4), batch(roots[2], 8), batch(roots[0], 4) + batch(roots[2], 4), batch(roots[1], 2), batch(roots[1], 3)]
    thr_t = 1.0
    thr = float(np.float32(thr_t) * np.float32(0.9))
    ref = _screener(None)                       # synchronous path, for the expected state of every batch
    dev = _screener(None)
    try:
        ref.load_texts(seqs, [1.0] * len(seqs))
        dev.load_texts(seqs, [1.0] * len(seqs))
        words = int(dev.bitset_words())
        for k, p in enumerate(batches * 2):     # twelve passes: the first and those around an odd batch are not lean, the others are
            want_fr, want_rf = _bits_sync(ref, p, thr, thr_t)
            want_entries = ref.entries()
            o = torch.full((2, len(p), words), -1, dtype=torch.int64, device="cuda:0")
            dev.screen_device(p, thr, o[0].data_ptr(), o[1].data_ptr(), thr_t, thr_t, 80, 200, False)
            dev.synchronize()
            torch.cuda.synchronize()
            got = _to_bool(o, len(seqs))
            assert np.array_equal(got[0], want_fr) and np.array_equal(got[1], want_rf), k
            assert dev.entries() == want_entries, k                        # no entry of an earlier pass survives
            _, fr, rf, _ = dev.amplify(p, thr_t, thr_t, 80, 200, False)    # touched list from the segment ends
            assert np.array_equal(np.array(fr), want_fr) and np.array_equal(np.array(rf), want_rf), k
            if k % 3 == 1:
                a1 = dev.collect_amplicons(p[0], thr_t, 80, 200)
                a2 = ref.collect_amplicons(p[0], thr_t, 80, 200)
                assert a1 == a2, k
        assert any(len(b) for b in batches)
    finally:
        ref.close()
        dev.close()


def test_screen_device_replays_after_bucket_overflow(oracle):
    """A pass whose per-sequence buckets overflow is detected at synchronize() and replayed: same bits as the
    synchronous path (which retries inside pcr_select_words)."""
    import torch
    rng = random.Random(12)
    seqs = ["A" * 300 + rand_seq(rng, 300) + "AC" * 200 + rand_seq(rng, 100) + "T" * 300, rand_seq(rng, 1200)]
    s1 = seqs[1]
    txt = [("A" * 20, "A" * 20), ("AC" * 10, "GT" * 10), (s1[100:120], revcomp(s1[220:240]))]
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in txt]
    thr = float(np.float32(1.0) * np.float32(0.9))
    a = _screener(None)
    try:
        a.load_texts(seqs, [1.0, 1.0])
        want = _bits_sync(a, pairs, thr, 1.0)
        assert len(a.entries()) > 64                     # more than the initial bucket size in one sequence
        a.load_texts(seqs, [1.0, 1.0])                   # resets the bucket size
        words = int(a.bitset_words())
        o1 = torch.full((2, len(pairs), words), -1, dtype=torch.int64, device="cuda:0")
        o2 = torch.full((2, len(pairs), words), -1, dtype=torch.int64, device="cuda:0")
        a.screen_device(pairs, thr, o1[0].data_ptr(), o1[1].data_ptr(), 1.0, 1.0, 80, 200, False)
        a.screen_device(pairs[2:], thr, o2[0].data_ptr(), o2[1].data_ptr(), 1.0, 1.0, 80, 200, False)
        a.synchronize()
        torch.cuda.synchronize()
        got = _to_bool(o1, len(seqs))
        assert np.array_equal(got[0], want[0])
        assert np.array_equal(got[1], want[1])
        assert want[0].any() or want[1].any()
        got2 = _to_bool(o2[:, :1], len(seqs))
        assert np.array_equal(got2[0][0], want[0][2])
        assert np.array_equal(got2[1][0], want[1][2])
        # the buckets have grown past the 64 slots the one-launch tail handles: the next asynchronous pass takes
        # the k_touched / k_finalize / k_match / k_pair route with the touched count read on the device
        o3 = torch.full((2, len(pairs), words), -1, dtype=torch.int64, device="cuda:0")
        a.screen_device(pairs, thr, o3[0].data_ptr(), o3[1].data_ptr(), 1.0, 1.0, 80, 200, False)
        a.synchronize()
        torch.cuda.synchronize()
        got3 = _to_bool(o3, len(seqs))
        assert np.array_equal(got3[0], want[0]) and np.array_equal(got3[1], want[1])
        assert len(a.entries()) > 64
    finally:
        a.close()


def test_epoch_wrap_with_bucket_growth_retries(oracle):
    """best[] is tagged (pass epoch << 8 | count) and never cleared between passes.  A select that enters two passes
    before the 24-bit epoch limit and needs bucket-growth retries (one epoch each) must clear the table when the
    counter wraps instead of dropping every hit: same DB as the oracle before, across and after the wrap."""
    rng = random.Random(12)
    seqs = ["A" * 300 + rand_seq(rng, 300) + "AC" * 200 + rand_seq(rng, 100) + "T" * 300, rand_seq(rng, 1200)]
    s1 = seqs[1]
    txt = [("A" * 20, "A" * 20), ("AC" * 10, "GT" * 10), (s1[100:120], revcomp(s1[220:240]))]
    pairs = [(oracle.centered_word(f), oracle.centered_word(r)) for f, r in txt]
    thr = float(np.float32(1.0) * np.float32(0.9))
    so = oracle.session()
    for q in seqs:
        so.add_target(q)
    n_o = so.select(pairs)
    want = so.db_entries()
    os.environ["PCRAMP_DEBUG_EPOCH"] = str((1 << 24) - 3)
    try:
        a = _screener(None)
    finally:
        os.environ.pop("PCRAMP_DEBUG_EPOCH", None)
    try:
        for rep in range(4):
            a.load_texts(seqs, [1.0, 1.0])               # resets the bucket size: every select retries with larger buckets
            assert a.select_words(pairs, thr, 18) == n_o and n_o > 64, rep
            assert a.entries() == want, rep
    finally:
        a.close()


def test_screen_device_with_shift_candidates(oracle):
    """--optimize.5/--optimize.3 (every slot shift of every oligo is a candidate): many words per site, so the
    per-sequence buckets outgrow 64 slots and the one-launch tail runs in its 128- / 256-slot form.  The fused
    asynchronous passes equal the synchronous calls, pass after pass."""
    import torch
    rng = random.Random(2024)
    root = rand_seq(rng, 4000)
    seqs = [root] + [mutate(rng, root, 0.02) for _ in range(30)] + [rand_seq(rng, 1500) for _ in range(4)]
    pairs = []
    for i in range(12):
        a0 = rng.randrange(0, 3600)
        f = root[a0:a0 + rng.randint(18, 25)]
        r = revcomp(root[a0 + 110:a0 + 110 + rng.randint(18, 25)])
        pairs.append((oracle.centered_word(f), oracle.centered_word(r)))
    thr = float(np.float32(1.0) * np.float32(0.9))
    d = _screener(None)
    try:
        d.load_texts(seqs, [1.0] * len(seqs))
        n = d.select_words(pairs, thr, 18, True, True)
        per_seq = {}
        for e in d.entries():
            per_seq[e[3]] = per_seq.get(e[3], 0) + 1
        assert max(per_seq.values()) > 64                      # buckets must have grown
        _, fr, rf, _ = d.amplify(pairs, 1.0, 1.0, 80, 200, False)
        assert fr.any() or rf.any()
        assert d.entries() == _oracle_entries(oracle, seqs, pairs, 1.0, 0.9, 1, 1)
        words = int(d.bitset_words())
        for rep in range(3):                                   # asynchronous passes with the grown buckets
            o = torch.full((2, len(pairs), words), -1, dtype=torch.int64, device="cuda:0")
            d.screen_device(pairs, thr, o[0].data_ptr(), o[1].data_ptr(), 1.0, 1.0, 80, 200, False, 18, True, True)
            d.synchronize()
            torch.cuda.synchronize()
            got = _to_bool(o, len(seqs))
            assert np.array_equal(got[0], fr) and np.array_equal(got[1], rf)
            assert len(d.entries()) == n
    finally:
        d.close()
