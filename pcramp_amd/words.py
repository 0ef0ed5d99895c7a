"""The reference's sequence / oligo encodings, host side (numpy only).

* bases: 4-bit IUPAC codes A=1 C=2 G=4 T=8, unions for ambiguity codes, EOS ('-') = 0
  (reference base_table.h:11-76).
* sequences: two codes per byte, HIGH nibble first (reference sequence.h:223-228).
* oligos: ``Word`` = 32 slots in two u64, slot k at bits (15 - k%16)*4 of w[k//16]
  (reference word.cpp:11-16); assay oligos are stored centred (Word::center, word.h:392-418).
"""
import numpy as np

_CODE = {'A': 1, 'C': 2, 'G': 4, 'T': 8, 'U': 8, 'M': 3, 'R': 5, 'S': 6, 'V': 7, 'W': 9, 'Y': 10,
         'H': 11, 'K': 12, 'D': 13, 'B': 14, 'N': 15, 'I': 15, 'X': 15, '-': 0}
_LUT = np.full(256, 255, dtype=np.uint8)
for _k, _v in _CODE.items():
    _LUT[ord(_k)] = _v
    _LUT[ord(_k.lower())] = _v
_LETTER = np.frombuffer(b"-ACMGRSVTWYHKDBN", dtype=np.uint8)
_COMP = np.array([((v & 1) << 3) | ((v & 8) >> 3) | ((v & 2) << 1) | ((v & 4) >> 1) for v in range(16)], dtype=np.uint8)


def codes_from_text(seq):
    """IUPAC text -> uint8 array of 4-bit codes (raises on an illegal symbol, as base_to_bits does)."""
    a = _LUT[np.frombuffer(seq.encode("ascii"), dtype=np.uint8)]
    if (a == 255).any():
        raise ValueError("illegal base symbol")
    return a


def text_from_codes(codes):
    return _LETTER[np.asarray(codes, dtype=np.uint8)].tobytes().decode("ascii")


def pack_codes(codes):
    """4-bit codes -> packed bytes, high nibble first; an odd tail is padded with EOS."""
    c = np.asarray(codes, dtype=np.uint8)
    if c.size & 1:
        c = np.concatenate([c, np.zeros(1, dtype=np.uint8)])
    return ((c[0::2] << 4) | c[1::2]).astype(np.uint8)


def unpack_codes(packed, length):
    p = np.asarray(packed, dtype=np.uint8)
    out = np.empty(2 * p.size, dtype=np.uint8)
    out[0::2] = p >> 4
    out[1::2] = p & 0xF
    return out[:length]


def revcomp_codes(codes):
    return _COMP[np.asarray(codes, dtype=np.uint8)][::-1]


def word_from_slots(slots):
    """32 slot codes -> (w0, w1)."""
    w = [0, 0]
    for k in range(32):
        w[k >> 4] |= int(slots[k]) << ((15 - (k & 15)) * 4)
    return (w[0], w[1])


def slots_from_word(word):
    return np.array([(int(word[k >> 4]) >> ((15 - (k & 15)) * 4)) & 0xF for k in range(32)], dtype=np.uint8)


def centered_word(codes):
    """An n-base oligo (n <= 32) as the reference stores assay oligos: left-aligned then
    Word::center(), which puts the first base at slot (33 - n)//2 (word.h:392-418)."""
    c = np.asarray(codes, dtype=np.uint8)
    n = c.size
    if n == 0 or n > 32:
        raise ValueError("oligo length must be in [1, 32]")
    start = (33 - n) // 2
    slots = np.zeros(32, dtype=np.uint8)
    slots[start:start + n] = c
    return word_from_slots(slots)


def center_word(word):
    """Word::center() (word.h:392-418): delta = ((32 - stop) - start)/2 truncated toward zero; shift right by
    delta if positive, left by -delta otherwise."""
    s = [int(v) for v in slots_from_word(word)]
    occ = [k for k in range(32) if s[k]]
    if not occ:
        return (int(word[0]), int(word[1]))
    left, right = occ[0], 32 - occ[-1]
    delta = int((right - left) / 2)            # C++ int division truncates toward zero
    out = [0] * 32
    for k in occ:
        if 0 <= k + delta < 32:
            out[k + delta] = s[k]
    return word_from_slots(out)


def word_degeneracy(word):
    """Word::degeneracy() (word.h:97-138): product of the slot multiplicities (1.0 for the empty word)."""
    d = 1.0
    for v in slots_from_word(word):
        n = bin(int(v)).count("1")
        if n:
            d *= n
    return d


def word_text(word):
    s = slots_from_word(word)
    nz = np.nonzero(s)[0]
    if nz.size == 0:
        return ""
    return text_from_codes(s[nz[0]:nz[-1] + 1])


def pairs_array(pairs):
    """[(F, R), ...] with F, R = (w0, w1) -> uint64 array [n, 4] laid out as pcr_pair[]."""
    a = np.zeros((len(pairs), 4), dtype=np.uint64)
    for i, (f, r) in enumerate(pairs):
        a[i] = (f[0], f[1], r[0], r[1])
    return a
