"""Seeded synthetic workloads with the shapes of BASELINE.json's configs.

Targets come in families: ceil(T/50) random roots, every member a copy of its root with 3 %
substitutions, so that primers sampled from one target hit many.  Primers are 18-25 nt windows
of random targets (F) and reverse complements of a downstream window (R), 80-200 bases apart,
stored centred like the reference's trial assays (pcr_assay.cpp:646,688,719).  Nothing is read
from disk; everything is a function of (config, seed).
"""
import numpy as np

from . import words as W

BASE_SEED = 20240901

CONFIGS = {
    # name: (T, L, P, n_background, L_background)
    "C1": dict(T=100, L=1000, P=5),
    "C2": dict(T=10000, L=10000, P=50),
    "C3": dict(T=50000, L=2000, P=50, B=10000, LB=2000),
    "C4_shard": dict(T=625, L=5000000, P=50),            # one of the 8 shards of C4 (5 000 genomes x 5 Mb); 1.56 GB packed
    "C5_shard": dict(T=12500, L=10000, P=50, degenerate=3),
}

_ACGT = np.array([1, 2, 4, 8], dtype=np.uint8)
_LOG2 = np.array([0, 0, 1, 0, 2, 0, 0, 0, 3], dtype=np.uint8)


def _rng(seed):
    return np.random.Generator(np.random.MT19937(seed))


def make_sequences(T, L, seed, family=50, divergence=0.03, chunk=256, n_roots=None, root_of=None, members=None):
    """-> (packed uint8 [sum ceil(L/2)], byte_offsets uint64[n], lengths uint64[n]) for the n = T sequences of the set,
    or for the listed `members` of it only.
    The roots are the FIRST draw of the seed's stream, so another call with the same (seed, L, n_roots) derives
    its members from the same roots (the backgrounds of a config: `root_of` maps member -> root).
    Genome-sized sets (L >= 1 Mb) give every member its own stream, so a rank can build exactly its block."""
    rng = _rng(seed)
    n_fam = (T + family - 1) // family if n_roots is None else n_roots
    roots = _ACGT[rng.integers(0, 4, size=(n_fam, L), dtype=np.uint8)]
    nb = (L + 1) // 2
    fam_of = np.arange(T) // family if root_of is None else np.asarray(root_of)
    if root_of is not None:
        rng = _rng(seed ^ 0xBAC6)                           # the members' own stream (the targets keep the seed's)
    if L >= 1000000:
        ids = np.arange(T) if members is None else np.asarray(list(members), dtype=np.int64)
        packed = np.empty(len(ids) * nb, dtype=np.uint8)
        n_mut = int(round(L * divergence))
        for k, m in enumerate(ids):
            # draw the substituted positions instead of a coin per base (a position drawn twice is substituted once)
            r = _rng((seed * 1000003 + 17 + int(m)) & 0x7FFFFFFFFFFF)
            c = roots[fam_of[int(m)]].copy()
            pos = r.integers(0, L, size=n_mut)
            rot = r.integers(1, 4, size=n_mut, dtype=np.uint8)
            c[pos] = _ACGT[(_LOG2[c[pos]] + rot) & 3]
            if L & 1:
                c = np.concatenate([c, np.zeros(1, np.uint8)])
            packed[k * nb:(k + 1) * nb] = (c[0::2] << 4) | c[1::2]
        n = len(ids)
        return packed, (np.arange(n, dtype=np.uint64) * np.uint64(nb)), np.full(n, L, dtype=np.uint64)
    packed = np.empty(T * nb, dtype=np.uint8)
    chunk = max(1, min(chunk, (32 << 20) // max(L, 1)))     # bound the temporaries (~20 B per base of a chunk)
    for lo in range(0, T, chunk):
        hi = min(T, lo + chunk)
        c = roots[fam_of[lo:hi]].copy()
        mut = rng.random(c.shape, dtype=np.float32) < divergence
        # a substitution always changes the base: rotate the one-hot code by 1..3 positions
        rot = rng.integers(1, 4, size=c.shape, dtype=np.uint8)
        idx = np.log2(c).astype(np.uint8)
        c = np.where(mut, _ACGT[(idx + rot) & 3], c)
        if L & 1:
            c = np.concatenate([c, np.zeros((hi - lo, 1), np.uint8)], axis=1)
        packed[lo * nb:hi * nb] = ((c[:, 0::2] << 4) | c[:, 1::2]).reshape(-1)
    if members is not None:
        ids = [int(m) for m in members]
        packed = np.concatenate([packed[m * nb:(m + 1) * nb] for m in ids]) if ids else np.zeros(0, np.uint8)
        T = len(ids)
    byte_offsets = (np.arange(T, dtype=np.uint64) * np.uint64(nb))
    lengths = np.full(T, L, dtype=np.uint64)
    return packed, byte_offsets, lengths


def sequence_codes(packed, byte_offsets, lengths, i):
    nb = (int(lengths[i]) + 1) // 2
    o = int(byte_offsets[i])
    return W.unpack_codes(packed[o:o + nb], int(lengths[i]))


def make_pairs(packed, byte_offsets, lengths, P, seed, primer=(18, 25), amplicon=(80, 200), degenerate=0, origins_out=None):
    """P primer pairs sampled from the sequences; `degenerate` > 0 widens that many positions
    per primer to a 2-fold IUPAC code (degeneracy <= 2**degenerate)."""
    rng = _rng(seed ^ 0x5EED)
    T = len(lengths)
    pairs = []
    origins = [] if origins_out is None else origins_out
    while len(pairs) < P:
        t = int(rng.integers(0, T))
        L = int(lengths[t])
        fl = int(rng.integers(primer[0], primer[1] + 1))
        rl = int(rng.integers(primer[0], primer[1] + 1))
        amp = int(rng.integers(amplicon[0], amplicon[1] + 1))
        if L < amp or amp < fl + rl:
            continue
        fs = int(rng.integers(0, L - amp + 1))
        codes = sequence_codes(packed, byte_offsets, lengths, t)
        f = codes[fs:fs + fl].copy()
        r = W.revcomp_codes(codes[fs + amp - rl:fs + amp]).copy()
        if (f == 0).any() or (r == 0).any():
            continue
        for o in (f, r):
            for _ in range(degenerate):
                k = int(rng.integers(0, o.size))
                o[k] |= _ACGT[int(rng.integers(0, 4))]
        pairs.append((W.centered_word(f), W.centered_word(r)))
        origins.append((t, fs, amp))
    return pairs


def workload(name, seed_offset=0, scale=1.0, family=50):
    """-> dict(packed, byte_offsets, lengths, pairs, origins, T, L, P [, background = dict(packed, byte_offsets,
    lengths, B, L, root_of)]).  `scale` < 1 shrinks T and B (CPU baselines).  origins[k] = (target, start, amplicon
    length) the k-th pair was cut from.  Backgrounds (C3): B sequences, member i = root (i mod n_roots) of the TARGET
    families with 15 % substitutions."""
    cfg = dict(CONFIGS[name])
    T = max(1, int(round(cfg["T"] * scale)))
    seed = BASE_SEED + {"C1": 1, "C2": 2, "C3": 3, "C4_shard": 4, "C5_shard": 5}[name] + 1000 * seed_offset
    packed, off, lens = make_sequences(T, cfg["L"], seed, family=family)
    origins = []
    pairs = make_pairs(packed, off, lens, cfg["P"], seed, degenerate=cfg.get("degenerate", 0), origins_out=origins)
    out = dict(packed=packed, byte_offsets=off, lengths=lens, pairs=pairs, origins=origins, T=T, L=cfg["L"], P=cfg["P"], name=name,
               family=family)
    if cfg.get("B"):
        if cfg["LB"] != cfg["L"]:
            raise ValueError("backgrounds are mutated target roots: LB must equal L")
        B = max(1, int(round(cfg["B"] * scale)))
        n_roots = (T + family - 1) // family
        root_of = np.arange(B) % n_roots
        bp, bo, bl = make_sequences(B, cfg["LB"], seed, family=family, divergence=0.15, n_roots=n_roots, root_of=root_of)
        out["background"] = dict(packed=bp, byte_offsets=bo, lengths=bl, B=B, L=cfg["LB"], root_of=root_of)
    return out


def subset(wl_set, idx, L):
    """(packed, byte_offsets, lengths) of the listed sequences of a workload's target or background set."""
    nb = (L + 1) // 2
    idx = [int(i) for i in idx]
    packed = np.concatenate([wl_set["packed"][int(wl_set["byte_offsets"][i]):int(wl_set["byte_offsets"][i]) + nb] for i in idx])
    return packed, np.arange(len(idx), dtype=np.uint64) * np.uint64(nb), np.full(len(idx), L, dtype=np.uint64)


class GlobalSet:
    """ONE target set made of `n_blocks` blocks of a config (block b = workload(config, seed_offset=b): the same family
    structure, different roots), addressed by global sequence index -- what the ranks of a multi-GPU run shard
    (pcramp_amd.shard.shard_ranges) and what a single GPU screens unsharded for the cross-check.  The primer pairs
    are block 0's.  Blocks are generated on demand and the two most recent are kept."""

    def __init__(self, config, n_blocks, scale=1.0, seed_base=0):
        self.config, self.n_blocks, self.scale, self.seed_base = config, int(n_blocks), scale, int(seed_base)
        cfg = CONFIGS[config]
        self.T_block = max(1, int(round(cfg["T"] * scale)))
        self.L = cfg["L"]
        self.T = self.T_block * self.n_blocks
        self.lengths = np.full(self.T, self.L, dtype=np.uint64)
        self._cache = {}
        self._pairs = None

    def _seed(self, b):
        return BASE_SEED + {"C1": 1, "C2": 2, "C3": 3, "C4_shard": 4, "C5_shard": 5}[self.config] + 1000 * (self.seed_base + b)

    def block(self, b):
        if b not in self._cache:
            if len(self._cache) >= 2:
                self._cache.pop(next(iter(self._cache)))
            self._cache[b] = workload(self.config, seed_offset=self.seed_base + b, scale=self.scale)
        return self._cache[b]

    def pairs(self):
        if self._pairs is None:
            self._pairs = self.block(0)["pairs"]
        return self._pairs

    def members(self, lo, hi):
        """(packed, byte_offsets, lengths) of global sequences [lo, hi)."""
        nb = (self.L + 1) // 2
        parts = []
        g = lo
        while g < hi:
            b, i = divmod(g, self.T_block)
            n = min(hi - g, self.T_block - i)
            if self.L >= 1000000:                         # genome-sized: build exactly these members
                cfg = CONFIGS[self.config]
                p, _, _ = make_sequences(self.T_block, self.L, self._seed(b), members=range(i, i + n))
                if b == 0 and self._pairs is None and i == 0 and n == self.T_block:
                    pass
                parts.append(p)
            else:
                blk = self.block(b)
                parts.append(blk["packed"][i * nb:(i + n) * nb])
            g += n
        packed = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        n = hi - lo
        return packed, np.arange(n, dtype=np.uint64) * np.uint64(nb), np.full(n, self.L, dtype=np.uint64)
