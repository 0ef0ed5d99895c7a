"""Small seeded inputs shared by the parity tests (plain numpy / random; no reference data)."""
import random

IUPAC2 = "MRSWYK"
IUPAC3 = "VHDB"


def rand_seq(rng, n, p_degen=0.0, p_n=0.0):
    out = []
    for _ in range(n):
        x = rng.random()
        if x < p_n:
            out.append("N")
        elif x < p_n + p_degen:
            out.append(rng.choice(IUPAC2 + IUPAC3))
        else:
            out.append(rng.choice("ACGT"))
    return "".join(out)


def mutate(rng, s, rate):
    out = list(s)
    for i, c in enumerate(out):
        if c in "ACGT" and rng.random() < rate:
            out[i] = rng.choice([b for b in "ACGT" if b != c])
    return "".join(out)


COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "M": "K", "K": "M", "R": "Y", "Y": "R", "S": "S", "W": "W",
        "V": "B", "B": "V", "H": "D", "D": "H", "N": "N", "-": "-"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def family_targets(rng, n_fam, per_fam, L, div=0.03, p_degen=0.0):
    seqs = []
    for _ in range(n_fam):
        root = rand_seq(rng, L, p_degen=p_degen)
        for _ in range(per_fam):
            seqs.append(mutate(rng, root, div))
    return seqs


def sample_pair(rng, seq, primer=(18, 25), amplicon=(80, 200)):
    """(F text, R text) cut from `seq`, R reverse-complemented; None if the pick hits an EOS."""
    L = len(seq)
    for _ in range(100):
        fl = rng.randint(*primer)
        rl = rng.randint(*primer)
        amp = rng.randint(*amplicon)
        if amp > L or amp < fl + rl:
            continue
        fs = rng.randint(0, L - amp)
        f = seq[fs:fs + fl]
        r = revcomp(seq[fs + amp - rl:fs + amp])
        if "-" in f or "-" in r:
            continue
        return f, r
    return None


def move_variants(W, word, kind, rng=None, context=None):
    """Single-edit variants of an oligo word in the spirit of the reference's local-search moves
    (optimize_pcr.cpp): 'inc' = add one base bit at one occupied slot (increase_degeneracy :57-70),
    'dec' = remove one base bit from a degenerate slot (decrease_degeneracy), 'trim5'/'trim3' = drop the
    first / last occupied slot, 'grow5'/'grow3' = occupy the next slot towards 5'/3' with one of the
    four bases.  Words are returned as the reference leaves them before re-centring (slots unmoved).
    W = pcramp_amd.words module."""
    slots = [int(v) for v in W.slots_from_word(word)]
    occ = [k for k in range(32) if slots[k]]
    first, last = occ[0], occ[-1]
    out = []
    if kind == "inc":
        for k in occ:
            for b in (1, 2, 4, 8):
                if not (slots[k] & b):
                    t = list(slots); t[k] |= b; out.append(W.word_from_slots(t))
    elif kind == "dec":
        for k in occ:
            if bin(slots[k]).count("1") > 1:
                for b in (1, 2, 4, 8):
                    if slots[k] & b:
                        t = list(slots); t[k] &= ~b; out.append(W.word_from_slots(t))
    elif kind == "trim5":
        t = list(slots); t[first] = 0; out.append(W.word_from_slots(t))
    elif kind == "trim3":
        t = list(slots); t[last] = 0; out.append(W.word_from_slots(t))
    elif kind == "grow5" and first > 0:
        for b in (1, 2, 4, 8):
            t = list(slots); t[first - 1] = b; out.append(W.word_from_slots(t))
    elif kind == "grow3" and last < 31:
        for b in (1, 2, 4, 8):
            t = list(slots); t[last + 1] = b; out.append(W.word_from_slots(t))
    return out


def multiplex_case(rng, words_mod, lib, n_amp=12):
    """Amplicon-like sequences (accepted assays' amplicons) and assays whose primers overlap them: primers cut
    from the amplicons (either strand) with a few wrong bases, so that keys are matched at ~0.8 by one or both
    oligos.  -> (amplicons, [(F, R)] as words)."""
    amps = []
    for i in range(n_amp):
        a = rand_seq(rng, rng.randint(45, 260))
        if i % 4 == 1:
            k = rng.randrange(5, len(a) - 5)
            a = a[:k] + rng.choice("RYKMSWN") + a[k + 1:]
        if i % 5 == 3:
            k = rng.randrange(20, len(a) - 20)
            a = a[:k] + "-" + a[k + 1:]
        amps.append(a)
    amps.append(amps[0][10:])                                          # shares most of its words with another amplicon
    pairs = []
    for _ in range(10):
        def primer():
            a = rng.choice(amps).replace("-", "A")
            n = rng.randint(18, 25)
            k = rng.randrange(0, len(a) - n)
            p = mutate(rng, a[k:k + n], rng.choice((0.0, 0.05, 0.12)))
            return revcomp(p) if rng.random() < 0.5 else p
        f, r = primer(), primer() if rng.random() < 0.7 else rand_seq(rng, 20)
        pairs.append((lib.centered_word(f), lib.centered_word(r)))
    return amps, pairs


def multiplex_design_case(rng, lib, seed_shift=0):
    """A small multiplex design state: target families, backgrounds, assays designed so far (the pool) with the
    amplicons they produced (the multiplex background), and candidate assays -- some reuse a pooled oligo, some
    sit on a pooled amplicon, some were damaged so that the search has something to repair."""
    seqs = family_targets(rng, 3, 6, 500, div=0.06)
    bgs = [mutate(rng, s, 0.12) for s in seqs[::4]] + [rand_seq(rng, 400)]
    pool_txt, amps = [], []
    while len(pool_txt) < 3:
        t = rng.choice(seqs)
        p = sample_pair(rng, t)
        if not p:
            continue
        f, r = p
        i, j = t.find(f), t.find(revcomp(r))
        if i < 0 or j < 0 or j <= i:
            continue
        pool_txt.append(p)
        amps.append(t[i + len(f) - 5:j + 5])                           # primers trimmed, 5 bases of padding (pcr_assay.cpp:489-497)
    cands = []
    while len(cands) < 4:
        p = sample_pair(rng, rng.choice(seqs))
        if p:
            cands.append(p)
    cands.append((pool_txt[0][0], cands[0][1]))                        # reuses a pooled forward primer
    cands.append((cands[1][0], pool_txt[1][0]))                        # reverse oligo identical to a pooled oligo
    a = amps[0]
    cands.append((a[8:28], revcomp(a[len(a) - 30:len(a) - 10])))       # sits on an accepted amplicon
    cands.append((mutate(rng, cands[2][0], 0.1), mutate(rng, cands[2][1], 0.1)))
    w = lambda x: lib.centered_word(x)
    return seqs, bgs, amps, [(w(f), w(r)) for f, r in pool_txt], [(w(f), w(r)) for f, r in cands]
