"""Scope row f-7: the reference PROGRAM replayed.  tests/golden/writers.json holds whole output files of `pcramp` (main.cpp
linked unchanged; one rank, one thread, fixed seed; oracle/make_golden.py) together with the command lines that made them; the
FASTA inputs are regenerated from the same seeded generator.  pcr_design runs the design loop (main.cpp:471-1130) over the C-ABI
on those inputs -- sampler on the running rand_r seed, word DBs, local search with the multiplex terms, the compatibility /
multiplex / background gates in trial order, best-assay bookkeeping, amplicon database, EOS splits, active flags -- and must
produce the same bytes, text and JSON.  Run with `-m gpu`."""
import json
import os
import random

import pytest

from pcramp_amd import api, design
from testdata import mutate, rand_seq

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# the input specs of oracle/make_golden.py::writers_golden and ::program_golden, by run index
SPECS = [dict(n_fam=3, per=4, L=600, n_bg=2), dict(n_fam=3, per=4, L=600, n_bg=2), dict(n_fam=1, per=4, L=500, n_bg=0),
         dict(n_fam=1, per=4, L=500, n_bg=0), dict(n_fam=2, per=3, L=451, n_bg=3), dict(n_fam=2, per=3, L=451, n_bg=3)]


def inputs(si, sp):
    r2 = random.Random(6000 + si // 2)
    roots = [rand_seq(r2, sp["L"] + 7 * k) for k in range(sp["n_fam"])]
    targets = [(">target_%d family %d" % (k * sp["per"] + j, k), mutate(r2, roots[k], 0.03)) for k in range(sp["n_fam"]) for j in range(sp["per"])]
    bgs = [(">bg_%d" % i, mutate(r2, roots[i % len(roots)], 0.12)) for i in range(sp["n_bg"])]
    return targets, bgs


def replay(run, targets, bgs):
    assert [[d, len(q)] for d, q in targets] == run["targets"] and [[d, len(q)] for d, q in bgs] == run["backgrounds"]
    o = design.options_from_argv(run["argv"])
    assert o["seed"] == run["seed"] and o["json"] == run["json"]
    d = api.Screener(0)
    try:
        d.load_texts([q for _, q in targets], [1.0] * len(targets))
        if bgs:
            d.load_texts([q for _, q in bgs], [1.0] * len(bgs), which=api.BACKGROUND)
        text, pool = design.design(d, [x for x, _ in targets], [len(q) for _, q in targets], [x for x, _ in bgs], [len(q) for _, q in bgs],
                                   argv=run["argv"], **o)
    finally:
        d.close()
    return text.decode("latin-1"), pool


@pytest.mark.parametrize("ri", range(6))
def test_reference_program_runs(ri):
    with open(os.path.join(G, "writers.json")) as f:
        run = json.load(f)["runs"][ri]
    targets, bgs = inputs(ri, SPECS[ri])
    got, pool = replay(run, targets, bgs)
    assert got == run["output"]
    assert len(pool) >= 2


def test_more_program_runs():
    """Runs with the switches the first six do not use (top-down start, 5' / 3' moves, a relaxed background gate): program.json."""
    path = os.path.join(G, "program.json")
    with open(path) as f:
        runs = json.load(f)["runs"]
    assert len(runs) >= 20
    n_aborted = 0
    for run in runs:
        r2 = random.Random(run["input_seed"])
        sp = run["spec"]
        roots = [rand_seq(r2, sp["L"] + 7 * k) for k in range(sp["n_fam"])]
        targets = [(">target_%d family %d" % (k * sp["per"] + j, k), mutate(r2, roots[k], sp["div"])) for k in range(sp["n_fam"]) for j in range(sp["per"])]
        bgs = [(">bg_%d" % i, mutate(r2, roots[i % len(roots)], sp["bg_div"])) for i in range(sp["n_bg"])]
        if run.get("aborted"):
            # the reference threw (uncaught inside an OpenMP region: the program dies, its output file is lost): pcr_design reports an error
            with pytest.raises(api.PcrError):
                replay(run, targets, bgs)
            n_aborted += 1
            continue
        got, pool = replay(run, targets, bgs)
        assert got == run["output"], run["argv"]
    assert n_aborted <= 3
