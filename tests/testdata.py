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
