"""Scope row f-4: the assay-list writers (pcr_format_*; host-only entry points of the C-ABI) against bytes the
reference wrote (tests/golden/writers.json, captured by oracle/make_golden.py):

* PCR::write / write_json (assay.h:288-375) in their four forms, called through the harness on assays whose pools
  reuse an oligo (lower case in text, "recycled":True in JSON);
* whole output files of the reference PROGRAM (main.cpp linked unchanged), text and JSON: each file is cut into header,
  iteration headings, assay records and footer; every piece is regenerated from its parsed content (oligos, scores,
  matched deflines, the driver's own id / active-target bookkeeping, main.cpp:468-502,1105-1120) and must give the same
  bytes, so the concatenation is the file.

Not covered by any reference run: "B-" / "+B-" lines -- with the reference's own optimiser no accepted assay of the toy
inputs cross-reacts with a background (lowering --background.threshold makes it reject every assay); those branches
are pinned by reading main.cpp:1046-1052,1081-1107,1192-1200,1235-1259 only."""
import json
import os
import re

import numpy as np
import pytest

from pcramp_amd import api, words as W

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RULE = "#" * 91 + "\n"


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(G, "writers.json")) as f:
        return json.load(f)


def _pair(h):
    return ((int(h[0], 16), int(h[1], 16)), (int(h[2], 16), int(h[3], 16)))


def _word(text):
    return W.centered_word(W.codes_from_text(text.upper()))


def test_oligo_forms(golden):
    reused = 0
    for c in golden["oligos"]:
        a, pool = _pair(c["assay"]), [_pair(p) for p in c["pool"]]
        for json_ in (0, 1):
            for with_pool in (0, 1):
                want = c["forms"]["%d%d" % (json_, with_pool)].encode("latin-1")
                assert api.format_oligos(a, pool, json=json_, use_multiplex=with_pool) == want
        reused += int(c["forms"]["01"] != c["forms"]["00"])
    assert reused >= 10


def _regenerate(run):
    """-> the reference output file rebuilt piece by piece with the writers."""
    out = run["output"]
    tdef, tlen = [t[0] for t in run["targets"]], [t[1] for t in run["targets"]]
    bdef, blen = [b[0] for b in run["backgrounds"]], [b[1] for b in run["backgrounds"]]
    js = bool(run["json"])
    w = api.AssayWriter(tdef, tlen, bdef, blen, json=js, use_multiplex=True)
    pieces = [w.header(run["argv"], run["seed"]).decode("latin-1")]
    assert out.startswith(pieces[0])
    pos = len(pieces[0])
    nt, nb = len(tdef), len(bdef)
    active = np.ones(nt, bool)
    total_bg = np.zeros(nb, bool)
    major, minor, it = 1, 1, 0
    pool = []
    while True:
        it += 1
        remaining, n_major, n_minor = int(active.sum()), major, minor
        if remaining == 0:                                   # main.cpp:488-502: everything detected -> next major id
            remaining, n_major, n_minor = nt, major + 1, 1
        head = w.iteration(it, n_major, n_minor, remaining).decode("latin-1")
        if not out.startswith(head, pos):
            break                                            # the design loop ended before this iteration (main.cpp:1126-1129)
        if not active.any():
            active[:] = True
        major, minor = n_major, n_minor
        pieces.append(head)
        pos += len(head)
        rest = out[pos:]
        if js:
            m = re.match(r'\t\t\t"forward primer":\{\n\t\t\t\t"sequence":"([A-Z]+)",.*?\t\t\t"reverse primer":\{\n\t\t\t\t"sequence":"([A-Z]+)",'
                         r'.*?\t\t\t"target matches":\[\n(.*?)\n\t\t\t\],\n\t\t\t"background matches":\[(.*?)\]\n\t\t\}', rest, re.S)
            if not m:
                break                                        # heading without a record: the iteration found no assay
            f, r = m.group(1), m.group(2)
            tm = re.findall(r'"(.*?)"', m.group(3))
            bm = re.findall(r'"(.*?)"', m.group(4))
            tc = bc = 0.0
        else:
            m = re.match(r"# Assay (\d+)\.(\d+) has target coverage score = (\S+) \((\S+)% of active\) and background coverage score = (\S+) "
                         r"\((\S+)% of active\)\nASSAY\.\d+\.\d+\t(\S+)\t(\S+)\tD\(F\)=\S+;D\(R\)=\S+\n((?:T-.*\n)*)((?:B-.*\n)*)", rest)
            if not m:
                break
            assert (int(m.group(1)), int(m.group(2))) == (major, minor)
            tc, bc, f, r = float(m.group(3)), float(m.group(5)), m.group(7), m.group(8)
            tm = [x[2:] for x in m.group(9).splitlines()]
            bm = [x[2:] for x in m.group(10).splitlines()]
        tmatch = np.array([d in tm for d in tdef], bool)
        bmatch = np.array([d in bm for d in bdef], bool)
        assert tmatch.sum() == len(tm) and bmatch.sum() == len(bm)
        pair = (_word(f), _word(r))
        # weights are 1: the norms are the counts of active sequences (main.cpp:603,649)
        rec = w.assay(pair, major, minor, tc, bc, float(remaining), float(nb), nb, tmatch, bmatch, pool).decode("latin-1")
        pieces.append(rec)
        pos += len(rec)
        active &= ~tmatch
        total_bg |= bmatch
        pool.append(pair)
    pieces.append(w.footer(active, total_bg).decode("latin-1"))
    return "".join(pieces), len(pool)


@pytest.mark.parametrize("ri", range(6))
def test_whole_output_files(golden, ri):
    run = golden["runs"][ri]
    got, n_assays = _regenerate(run)
    assert got == run["output"]
    assert n_assays >= 2


def test_background_branches_are_exercised():
    """The cross-reaction branches no reference run reaches (see module docstring): well-formed, and the JSON footer
    leaves its array open exactly as main.cpp:1235-1259 does."""
    w = api.AssayWriter([">t0", ">t1"], [100, 120], [">b0", ">b1", ">b2"], [90, 95, 99], json=False)
    pair = (_word("ACGTACGTACGTACGTAC"), _word("TTGACCATGCATGCATGCA"))
    rec = w.assay(pair, 1, 1, 2.0, 1.0, 2.0, 3.0, 3, [1, 1], [0, 1, 0]).decode()
    assert rec == ("# Assay 1.1 has target coverage score = 2 (100% of active) and background coverage score = 1 (33.3333% of active)\n"
                   "ASSAY.1.1\tACGTACGTACGTACGTAC\tTTGACCATGCATGCATGCA\tD(F)=1;D(R)=1\nT->t0\nT->t1\nB->b1\n")
    foot = w.footer([0, 0], [0, 1, 1]).decode()
    assert foot == RULE + "# Detected all targets\n" + RULE + "# Cross reacted with a total of 2 background sequences\n+B->b1\n+B->b2\n"
    wj = api.AssayWriter([">t0", ">t1"], [100, 120], [">b0", ">b1", ">b2"], [90, 95, 99], json=True)
    rec = wj.assay(pair, 1, 1, 2.0, 1.0, 2.0, 3.0, 3, [1, 0], [1, 0, 1]).decode()
    assert rec.endswith('\t\t\t"target matches":[\n\t\t\t\t">t0"\n\t\t\t],\n\t\t\t"background matches":[\n\t\t\t\t">b0",\n\t\t\t\t">b2"\n\t\t\t]\n\t\t}')
    foot = wj.footer([0, 1], [1, 0, 0]).decode()
    assert foot == '\n\t],\n\t"unmatched targets":[\t\t">t1"\n\t],\n\t"total number of background matches":1,\n\t"background matches":[\n\t\t">b0"\n}\n'
