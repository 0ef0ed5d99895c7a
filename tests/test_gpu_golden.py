"""The HIP path against the committed golden vectors captured from the compiled reference
(tests/golden/screen.json): word DB, amplification bits, coverage -- bit-exact."""
import json
import os

import numpy as np
import pytest

from pcramp_amd import api

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("ci", range(5))
def test_screen_golden(ci):
    with open(os.path.join(G, "screen.json")) as f:
        c = json.load(f)["cases"][ci]
    o = c["options"]
    d = api.Screener(0, pack_max_degen=o["pack_max_degen"], pack_min_gc=o["pack_min_gc"], pack_max_gc=o["pack_max_gc"])
    d.load_texts(c["seqs"], c["weights"])
    active = np.ones(len(c["seqs"]), np.uint8)
    for i in c["inactive"]:
        active[i] = 0
    d.set_active(active)
    for i, pos in c["splits"]:
        d.split(i, pos)
    pairs = [((int(p[0], 16), int(p[1], 16)), (int(p[2], 16), int(p[3], 16))) for p in c["pairs"]]
    thr = float(np.float32(o["target_threshold"]) * np.float32(o["search_multiplier"]))
    n = d.select_words(pairs, thr, o["min_primer"], o["optimize_5"], o["optimize_3"])
    assert n == c["n_entries"]
    assert d.entries() == sorted((int(a, 16), int(b, 16), loc, idx, st) for a, b, loc, idx, st in c["db"])
    bits = d.find_target_match(pairs, o["target_threshold"], o["amp_min"], o["amp_max"], o["use_taq_mama"])
    cov = d.compute_coverage(pairs, o["target_threshold"], o["search_multiplier"], o["amp_min"], o["amp_max"], o["use_taq_mama"])
    for k in range(len(pairs)):
        assert bits[k].astype(int).tolist() == c["bits"][k]
        assert cov[k] == np.float32(c["coverage"][k])
    d.close()
