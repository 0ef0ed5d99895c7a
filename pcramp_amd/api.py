"""Host-side mirror of the reference call sites, over the C-ABI in include/pcramp_hip.h.

``Screener`` owns one GPU.  Its methods are named after what they replace in the reference:

=====================  =======================================================================
Screener method        reference call site
=====================  =======================================================================
load_sequences         deque<Sequence> after parse_fasta (main.cpp:257-344)
set_active             Sequence::active(bool) (main.cpp:1105-1120)
split                  Sequence::split_sequence (sequence.h:228; main.cpp:1008-1017)
select_words           Sequence::pack + select_words per active sequence (main.cpp:579-615, 644-691)
entries                the word DB `target_db` / `background_db`
amplify                PCR::find_target_match + PCR::compute_coverage (pcr_assay.cpp:544, 271)
find_target_match      PCR::find_target_match with Options' thresholds (main.cpp:898)
compute_coverage       optimize()'s collect/update/compute_coverage (optimize.cpp:61-74)
=====================  =======================================================================

Errors surface as ``PcrError`` (the reference throws ``const char*``, main.cpp:1269).
"""
import ctypes as C
import os

import numpy as np

from . import words as W

TARGET, BACKGROUND, MULTIPLEX = 0, 1, 2       # pcr_set: target_seq, background_seq, multiplex_background_seq (accepted amplicons)


class PcrError(RuntimeError):
    pass


class _Params(C.Structure):
    _fields_ = [("pack_max_degen", C.c_uint32), ("pack_min_gc", C.c_float), ("pack_max_gc", C.c_float)]


class _Entry(C.Structure):
    _fields_ = [("w", C.c_uint64 * 2), ("loc", C.c_int32), ("index", C.c_uint32), ("strand", C.c_uint32),
                ("pad", C.c_uint32)]


class SwResult(C.Structure):
    _fields_ = [("score", C.c_int16), ("q_start", C.c_int16), ("q_stop", C.c_int16), ("t_start", C.c_int16),
                ("t_stop", C.c_int16), ("last1", C.c_uint8), ("last2", C.c_uint8), ("valid", C.c_uint8), ("pad", C.c_uint8)]


class BackgroundArgs(C.Structure):
    _fields_ = [("collect_threshold", C.c_float), ("background_threshold", C.c_float), ("amp_min", C.c_int32),
                ("amp_max", C.c_int32), ("use_taq_mama", C.c_int32), ("evaluate_all_amplicons", C.c_int32)]


class ThermoArgs(C.Structure):
    _fields_ = [("salt", C.c_float), ("primer_strand", C.c_float), ("tm_min", C.c_float), ("tm_max", C.c_float),
                ("max_hairpin", C.c_float), ("max_dimer", C.c_float)]


class Amplicon(C.Structure):
    _fields_ = [("sequence", C.c_uint32), ("begin", C.c_int32), ("end", C.c_int32), ("inner_start", C.c_int32),
                ("inner_length", C.c_int32), ("orientation", C.c_uint32)]


class SamplerArgs(C.Structure):
    _fields_ = [("primer_min", C.c_int32), ("primer_max", C.c_int32), ("amp_min", C.c_int32), ("amp_max", C.c_int32),
                ("max_degen", C.c_double)]


class SampleInfo(C.Structure):
    _fields_ = [("sequence", C.c_uint32), ("f_start", C.c_int32), ("amplicon_length", C.c_int32),
                ("sequence_iterations", C.c_uint32), ("assay_iterations", C.c_uint32)]


class MultiplexScreenArgs(C.Structure):
    _fields_ = [("thermo", ThermoArgs), ("background_threshold", C.c_float), ("use_taq_mama", C.c_int32),
                ("target_threshold", C.c_float), ("amp_min", C.c_int32), ("amp_max", C.c_int32)]


class ThermoResult(C.Structure):
    _fields_ = [("valid", C.c_uint32), ("n_expansions", C.c_uint32), ("tm", C.c_float), ("dH", C.c_float), ("dS", C.c_float),
                ("hairpin_tm", C.c_float), ("homodimer_tm", C.c_float), ("dG", C.c_float)]


class Output(C.Structure):
    _fields_ = [("json", C.c_int32), ("use_multiplex", C.c_int32), ("n_target", C.c_uint32), ("n_background", C.c_uint32),
                ("target_deflines", C.POINTER(C.c_char_p)), ("background_deflines", C.POINTER(C.c_char_p)),
                ("target_lengths", C.c_void_p), ("background_lengths", C.c_void_p)]


class AssayRecord(C.Structure):
    _fields_ = [("major_id", C.c_uint32), ("minor_id", C.c_uint32), ("assay", C.c_uint64 * 4),
                ("target_coverage", C.c_float), ("background_coverage", C.c_float),
                ("active_target_norm", C.c_float), ("active_background_norm", C.c_float),
                ("num_active_background", C.c_uint32), ("target_match", C.c_void_p), ("background_match", C.c_void_p)]


class AmplifyArgs(C.Structure):
    _fields_ = [("collect_threshold", C.c_float), ("ident_threshold", C.c_float), ("amp_min", C.c_int32),
                ("amp_max", C.c_int32), ("use_taq_mama", C.c_int32)]


def library_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpcramp_hip.so")


_LIB = None

# every symbol include/pcramp_hip.h declares
ABI_SYMBOLS = [
    "pcr_last_error", "pcr_create", "pcr_destroy", "pcr_load_sequences", "pcr_set_active", "pcr_split", "pcr_split_many",
    "pcr_select_words", "pcr_get_entries", "pcr_amplify", "pcr_amplify_device", "pcr_screen_device", "pcr_move_coverage", "pcr_coverage_from_bits",
    "pcr_weighted_coverage", "pcr_num_sequences", "pcr_bitset_words", "pcr_profile_enable", "pcr_profile_read", "pcr_profile_read_kernel",
    "pcr_synchronize", "pcr_host_irregular_words", "pcr_host_window_valid", "pcr_host_candidates",
    "pcr_host_orientation_seeds", "pcr_host_move_trials",
    "pcr_sw_align_words", "pcr_background_match", "pcr_multiplex_match",
    "pcr_thermo", "pcr_dimer", "pcr_multiplex_compatible", "pcr_multiplex_screen",
    "pcr_random_assays", "pcr_host_rand_r", "pcr_host_max_overlap", "pcr_host_oligo_overlap", "pcr_host_pool_overlaps",
    "pcr_multiplex_load", "pcr_multiplex_coverage", "pcr_collect_amplicons",
    "pcr_format_oligos", "pcr_format_header", "pcr_format_iteration", "pcr_format_assay", "pcr_format_footer",
    "pcr_optimize_batch", "pcr_optimization_move", "pcr_make_degenerate", "pcr_staging_mode",
    "pcr_design", "pcr_design_output",
    "pcr_comm_unique_id", "pcr_comm_init_rank", "pcr_comm_world", "pcr_comm_rank", "pcr_exchange_bits", "pcr_comm_destroy", "pcr_comm_library",
]


def load_library():
    """dlopen the in-tree C-ABI library.  Fails loudly if it was not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise PcrError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(or make -C pcramp_amd/csrc); there is no CPU fallback" % path)
    # PyTorch-ROCm bundles its own HIP runtime: whichever libamdhip64 is loaded first serves the whole process, and
    # a process in which this library came first cannot initialise torch.cuda afterwards ("No HIP GPUs are
    # available").  Callers use torch for device buffers and RCCL, so let it load first when it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    L.pcr_last_error.restype = C.c_char_p
    L.pcr_create.restype = C.c_void_p
    L.pcr_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(_Params)]
    L.pcr_destroy.argtypes = [C.c_void_p]
    L.pcr_load_sequences.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.pcr_set_active.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.pcr_split.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint64]
    L.pcr_select_words.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_float,
                                   C.c_uint32, C.POINTER(C.c_uint64)]
    L.pcr_get_entries.restype = C.c_int64
    L.pcr_get_entries.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    L.pcr_amplify.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(AmplifyArgs), C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcr_amplify_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(AmplifyArgs),
                                     C.c_void_p, C.c_void_p]
    L.pcr_coverage_from_bits.restype = C.c_float
    L.pcr_coverage_from_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.pcr_weighted_coverage.restype = C.c_float
    L.pcr_weighted_coverage.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.pcr_num_sequences.restype = C.c_uint32
    L.pcr_num_sequences.argtypes = [C.c_void_p, C.c_int]
    L.pcr_bitset_words.restype = C.c_uint64
    L.pcr_bitset_words.argtypes = [C.c_void_p, C.c_int]
    L.pcr_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.pcr_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
    L.pcr_profile_read_kernel.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
    L.pcr_synchronize.argtypes = [C.c_void_p]
    L.pcr_staging_mode.argtypes = [C.c_void_p]
    L.pcr_comm_unique_id.argtypes = [C.c_void_p]
    L.pcr_comm_init_rank.restype = C.c_void_p
    L.pcr_comm_init_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.pcr_comm_world.argtypes = [C.c_void_p]
    L.pcr_comm_rank.argtypes = [C.c_void_p]
    L.pcr_exchange_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.pcr_comm_destroy.argtypes = [C.c_void_p]
    L.pcr_comm_destroy.restype = None
    L.pcr_comm_library.restype = C.c_char_p
    L.pcr_sw_align_words.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.pcr_background_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(BackgroundArgs), C.c_void_p]
    L.pcr_multiplex_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_void_p]
    L.pcr_thermo.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(ThermoArgs), C.c_void_p]
    L.pcr_dimer.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(ThermoArgs), C.c_void_p]
    L.pcr_multiplex_compatible.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(ThermoArgs), C.c_void_p]
    L.pcr_random_assays.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(SamplerArgs),
                                    C.POINTER(ThermoArgs), C.c_void_p, C.c_void_p]
    L.pcr_collect_amplicons.restype = C.c_int64
    L.pcr_collect_amplicons.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64]
    L.pcr_multiplex_load.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.pcr_multiplex_coverage.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_void_p]
    L.pcr_host_pool_overlaps.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    L.pcr_host_max_overlap.restype = C.c_float
    L.pcr_host_max_overlap.argtypes = [C.c_void_p, C.c_void_p]
    L.pcr_host_oligo_overlap.restype = C.c_float
    L.pcr_host_oligo_overlap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.pcr_host_rand_r.restype = C.c_uint32
    L.pcr_host_rand_r.argtypes = [C.POINTER(C.c_uint32)]
    L.pcr_host_irregular_words.restype = C.c_int64
    L.pcr_host_irregular_words.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_Params), C.c_uint32, C.c_void_p, C.c_uint64]
    L.pcr_host_window_valid.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_Params), C.c_void_p]
    L.pcr_host_candidates.restype = C.c_int64
    L.pcr_host_candidates.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_uint64]
    L.pcr_screen_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_float, C.c_uint32,
                                    C.POINTER(AmplifyArgs), C.c_void_p, C.c_void_p]
    L.pcr_move_coverage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(AmplifyArgs),
                                    C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcr_host_move_trials.restype = C.c_int64
    L.pcr_host_move_trials.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_uint64]
    L.pcr_host_orientation_seeds.restype = C.c_int64
    L.pcr_host_orientation_seeds.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    for fn in ("pcr_format_oligos", "pcr_format_header", "pcr_format_iteration", "pcr_format_assay", "pcr_format_footer"):
        getattr(L, fn).restype = C.c_int64
    L.pcr_format_oligos.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_char_p, C.c_uint64]
    L.pcr_format_header.argtypes = [C.POINTER(Output), C.c_int, C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.c_uint64]
    L.pcr_format_iteration.argtypes = [C.POINTER(Output), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64]
    L.pcr_format_assay.argtypes = [C.POINTER(Output), C.POINTER(AssayRecord), C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint64]
    L.pcr_format_footer.argtypes = [C.POINTER(Output), C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64]
    _LIB = L
    return L


def _err(L):
    return (L.pcr_last_error() or b"").decode()


# ---------------------------------------------------------------------------- host-only helpers
def host_irregular_words(packed, length, min_oligo_length=18, pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0):
    L = load_library()
    p = _Params(pack_max_degen, pack_min_gc, pack_max_gc)
    buf = np.ascontiguousarray(packed, dtype=np.uint8)
    n = L.pcr_host_irregular_words(buf.ctypes.data, length, C.byref(p), min_oligo_length, None, 0)
    if n < 0:
        raise PcrError(_err(L))
    out = (_Entry * max(int(n), 1))()
    L.pcr_host_irregular_words(buf.ctypes.data, length, C.byref(p), min_oligo_length, out, n)
    return [(e.w[0], e.w[1], e.loc, e.strand) for e in out[:n]]


def host_window_valid(packed, length, pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0):
    L = load_library()
    p = _Params(pack_max_degen, pack_min_gc, pack_max_gc)
    buf = np.ascontiguousarray(packed, dtype=np.uint8)
    out = np.zeros(max(length, 1), dtype=np.uint8)
    if L.pcr_host_window_valid(buf.ctypes.data, length, C.byref(p), out.ctypes.data) != 0:
        raise PcrError(_err(L))
    return out[:length]


def host_max_overlap(a, b):
    """Word::max_overlap (word.h:38-91) -> float32."""
    L = load_library()
    x = np.array([int(a[0]), int(a[1])], dtype=np.uint64)
    y = np.array([int(b[0]), int(b[1])], dtype=np.uint64)
    return np.float32(L.pcr_host_max_overlap(x.ctypes.data, y.ctypes.data))


def host_pool_overlaps(words, pool):
    """Largest Word::max_overlap of every word with any oligo of the pooled assays -> float32[n]."""
    L = load_library()
    a = np.array([[int(w[0]), int(w[1])] for w in words], dtype=np.uint64).reshape(-1, 2)
    out = np.zeros(max(a.shape[0], 1), np.float32)
    p = W.pairs_array(pool) if len(pool) else np.zeros((0, 4), np.uint64)
    if L.pcr_host_pool_overlaps(a.ctypes.data, a.shape[0], p.ctypes.data if len(pool) else None, len(pool), out.ctypes.data) != 0:
        raise PcrError(_err(L))
    return out[:a.shape[0]]


def host_oligo_overlap(assay, pool):
    """PCR::compute_oligo_overlap (pcr_assay.cpp:736-754): assay (F, R) against a pool of (F, R) -> float32."""
    L = load_library()
    a = W.pairs_array([assay])
    p = W.pairs_array(pool) if len(pool) else np.zeros((0, 4), np.uint64)
    return np.float32(L.pcr_host_oligo_overlap(a.ctypes.data, p.ctypes.data if len(pool) else None, len(pool)))


def host_rand_r(seed):
    """glibc rand_r (sample.cpp:12) -> (value, next seed)."""
    L = load_library()
    s = C.c_uint32(int(seed))
    v = L.pcr_host_rand_r(C.byref(s))
    return int(v), int(s.value)


def host_move_trials(word, move, max_degen=1, primer_min=18, primer_max=25):
    """Trial words of one optimize_pcr.cpp move (0 +degen, 1 -degen, 2 trim5, 3 trim3, 4 grow5, 5 grow3)."""
    L = load_library()
    w = np.array([int(word[0]), int(word[1])], dtype=np.uint64)
    out = np.zeros((512, 2), dtype=np.uint64)
    n = L.pcr_host_move_trials(w.ctypes.data, int(move), float(max_degen), int(primer_min), int(primer_max), out.ctypes.data, 512)
    if n < 0:
        raise PcrError(_err(L))
    return [(int(out[i, 0]), int(out[i, 1])) for i in range(int(n))]


def host_orientation_seeds(word, floor):
    """Seeds (code, q, off) of one oligo word as the seed scan would use them, or None if unseedable."""
    L = load_library()
    w = np.array([int(word[0]), int(word[1])], dtype=np.uint64)
    n = L.pcr_host_orientation_seeds(w.ctypes.data, int(floor), None, None, None, 0)
    if n < 0:
        return None
    codes = np.zeros(max(int(n), 1), np.uint32)
    q = np.zeros(max(int(n), 1), np.uint8)
    off = np.zeros(max(int(n), 1), np.uint8)
    L.pcr_host_orientation_seeds(w.ctypes.data, int(floor), codes.ctypes.data, q.ctypes.data, off.ctypes.data, n)
    return [(int(codes[i]), int(q[i]), int(off[i])) for i in range(int(n))]


def host_candidates(pairs, optimize_5=False, optimize_3=False, threshold=0.9):
    L = load_library()
    a = W.pairs_array(pairs)
    n = L.pcr_host_candidates(a.ctypes.data, len(pairs), int(optimize_5), int(optimize_3), threshold, None, None, 0)
    words = np.zeros((max(int(n), 1), 2), dtype=np.uint64)
    floors = np.zeros(max(int(n), 1), dtype=np.uint32)
    L.pcr_host_candidates(a.ctypes.data, len(pairs), int(optimize_5), int(optimize_3), threshold,
                          words.ctypes.data, floors.ctypes.data, n)
    return [(int(words[i, 0]), int(words[i, 1])) for i in range(n)], floors[:n].copy()


def coverage_from_bits(bits_fr, bits_rf, weights):
    """PCR::compute_coverage's weight sum from (gathered) orientation bitsets."""
    L = load_library()
    a = np.ascontiguousarray(bits_fr, dtype=np.uint64)
    b = np.ascontiguousarray(bits_rf, dtype=np.uint64)
    w = np.ascontiguousarray(weights, dtype=np.float32)
    return L.pcr_coverage_from_bits(a.ctypes.data, b.ctypes.data, w.ctypes.data, w.size)


def weighted_coverage(bits, weights):
    L = load_library()
    a = np.ascontiguousarray(bits, dtype=np.uint64)
    w = np.ascontiguousarray(weights, dtype=np.float32)
    return L.pcr_weighted_coverage(a.ctypes.data, w.ctypes.data, w.size)


def bits_to_bool(words, n):
    """u64 bitset words (bit i%64 of word i//64) -> bool array of n."""
    w = np.ascontiguousarray(words, dtype=np.uint64)
    b = np.unpackbits(w.view(np.uint8), bitorder="little")
    return b[:n].astype(bool)


# ---------------------------------------------------------------------------- assay-list writers (host only)
def bool_to_bits(flags):
    """bool[n] -> u64 BitSet words (bit i%64 of word i//64)."""
    f = np.asarray(flags).astype(np.uint8)
    pad = (-f.size) % 64
    return np.packbits(np.concatenate([f, np.zeros(pad, np.uint8)]), bitorder="little").view(np.uint64).copy()


class AssayWriter:
    """The reference's output file, piece by piece (main.cpp:131-163, 440-519, 950-1264; assay.h:288-375).
    Every method returns the bytes the reference writes at that point."""

    def __init__(self, target_deflines, target_lengths, background_deflines=(), background_lengths=(), json=False,
                 use_multiplex=True):
        self.L = load_library()
        self._keep = []
        self.nt, self.nb = len(target_deflines), len(background_deflines)

        def strs(v):
            a = (C.c_char_p * max(len(v), 1))(*[x.encode() for x in v])
            self._keep.append(a)
            return C.cast(a, C.POINTER(C.c_char_p))

        def lens(v):
            a = np.ascontiguousarray(list(v) or [0], dtype=np.uint64)
            self._keep.append(a)
            return a.ctypes.data
        self.o = Output(int(json), int(use_multiplex), self.nt, self.nb, strs(target_deflines), strs(background_deflines),
                        lens(target_lengths), lens(background_lengths))

    def _call(self, fn, *args):
        n = fn(*args, None, 0)
        if n < 0:
            raise PcrError(_err(self.L))
        buf = C.create_string_buffer(int(n) + 1)
        fn(*args, buf, n + 1)
        return buf.raw[:n]

    def header(self, argv, seed):
        a = (C.c_char_p * len(argv))(*[x.encode() for x in argv])
        return self._call(self.L.pcr_format_header, C.byref(self.o), len(argv), a, int(seed))

    def iteration(self, assay_iteration, major_id, minor_id, targets_remaining):
        return self._call(self.L.pcr_format_iteration, C.byref(self.o), assay_iteration, major_id, minor_id, targets_remaining)

    def assay(self, pair, major_id, minor_id, target_coverage, background_coverage, active_target_norm, active_background_norm,
              num_active_background, target_match, background_match, pool=()):
        tm, bm = bool_to_bits(target_match), bool_to_bits(background_match) if self.nb else None
        r = AssayRecord(major_id, minor_id, (C.c_uint64 * 4)(pair[0][0], pair[0][1], pair[1][0], pair[1][1]), target_coverage,
                        background_coverage, active_target_norm, active_background_norm, num_active_background,
                        tm.ctypes.data, bm.ctypes.data if bm is not None else None)
        p = W.pairs_array(list(pool)) if len(pool) else None
        return self._call(self.L.pcr_format_assay, C.byref(self.o), C.byref(r), p.ctypes.data if p is not None else None, len(pool))

    def footer(self, target_active, total_background):
        a = np.ascontiguousarray(np.asarray(target_active).astype(np.uint8))
        b = bool_to_bits(total_background) if self.nb else None
        return self._call(self.L.pcr_format_footer, C.byref(self.o), a.ctypes.data, b.ctypes.data if b is not None else None)


def format_oligos(pair, pool=(), json=False, use_multiplex=True):
    """PCR::write / PCR::write_json (assay.h:288-375) -> bytes."""
    L = load_library()
    a = W.pairs_array([pair])
    p = W.pairs_array(list(pool)) if len(pool) else None
    args = (a.ctypes.data, p.ctypes.data if p is not None else None, len(pool), int(json), int(use_multiplex))
    n = L.pcr_format_oligos(*args, None, 0)
    if n < 0:
        raise PcrError(_err(L))
    buf = C.create_string_buffer(int(n) + 1)
    L.pcr_format_oligos(*args, buf, n + 1)
    return buf.raw[:n]


# ---------------------------------------------------------------------------- the device handle
class Screener:
    """One GPU's share of the sequence sets plus the kernels that evaluate primer pairs on it."""

    def __init__(self, device=0, stream=None, pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0):
        self.L = load_library()
        p = _Params(pack_max_degen, pack_min_gc, pack_max_gc)
        self.h = self.L.pcr_create(device, C.c_void_p(stream) if stream else None, C.byref(p))
        if not self.h:
            raise PcrError(_err(self.L))
        self.weights = {TARGET: np.zeros(0, np.float32), BACKGROUND: np.zeros(0, np.float32)}

    def close(self):
        if getattr(self, "h", None):
            self.L.pcr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise PcrError(_err(self.L))

    def load_sequences(self, packed, byte_offsets, lengths, weights=None, which=TARGET):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        bo = np.ascontiguousarray(byte_offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        n = ln.size
        w = np.ones(n, np.float32) if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
        self.weights[which] = w
        self._check(self.L.pcr_load_sequences(self.h, which, packed.ctypes.data, bo.ctypes.data, ln.ctypes.data,
                                              w.ctypes.data, n))

    def load_texts(self, seqs, weights=None, which=TARGET):
        """Convenience for tests: IUPAC strings ('-' = EOS)."""
        packs = [W.pack_codes(W.codes_from_text(s)) for s in seqs]
        lengths = [len(s) for s in seqs]
        off = np.zeros(len(seqs), dtype=np.uint64)
        tot = 0
        for i, p in enumerate(packs):
            off[i] = tot
            tot += p.size
        packed = np.concatenate(packs) if packs else np.zeros(0, np.uint8)
        self.load_sequences(packed, off, lengths, weights, which)

    def num_sequences(self, which=TARGET):
        return self.L.pcr_num_sequences(self.h, which)

    def bitset_words(self, which=TARGET):
        return self.L.pcr_bitset_words(self.h, which)

    def set_active(self, active, which=TARGET):
        a = np.ascontiguousarray(np.asarray(active).astype(np.uint8))
        self._check(self.L.pcr_set_active(self.h, which, a.ctypes.data))

    def split(self, seq, pos, which=TARGET):
        self._check(self.L.pcr_split(self.h, which, seq, pos))

    def select_words(self, pairs, threshold, min_oligo_length=18, optimize_5=False, optimize_3=False, which=TARGET,
                     count=True):
        """count=False skips the (host-side) DB size computation and returns None."""
        a = pairs if isinstance(pairs, np.ndarray) else W.pairs_array(pairs)
        n = C.c_uint64(0)
        self._check(self.L.pcr_select_words(self.h, which, a.ctypes.data, a.shape[0], int(optimize_5), int(optimize_3),
                                            threshold, min_oligo_length, C.byref(n) if count else None))
        return n.value if count else None

    def entries(self, which=TARGET):
        n = self.L.pcr_get_entries(self.h, which, None, 0)
        if n < 0:
            raise PcrError(_err(self.L))
        buf = (_Entry * max(int(n), 1))()
        self.L.pcr_get_entries(self.h, which, buf, n)
        return sorted((e.w[0], e.w[1], e.loc, e.index, e.strand) for e in buf[:n])

    def amplify(self, pairs, collect_threshold, ident_threshold, amp_min=80, amp_max=200, use_taq_mama=False,
                which=TARGET):
        """-> (bits[n_pairs, n] bool, bits_fr, bits_rf, coverage[n_pairs] float32)."""
        a = pairs if isinstance(pairs, np.ndarray) else W.pairs_array(pairs)
        P = a.shape[0]
        nw = int(self.bitset_words(which))
        n = self.num_sequences(which)
        bits = np.zeros((P, nw), np.uint64)
        fr = np.zeros((P, nw), np.uint64)
        rf = np.zeros((P, nw), np.uint64)
        cov = np.zeros(P, np.float32)
        args = AmplifyArgs(collect_threshold, ident_threshold, amp_min, amp_max, int(use_taq_mama))
        self._check(self.L.pcr_amplify(self.h, which, a.ctypes.data, P, C.byref(args), bits.ctypes.data,
                                       fr.ctypes.data, rf.ctypes.data, cov.ctypes.data))
        tb = lambda x: np.stack([bits_to_bool(x[i], n) for i in range(P)]) if P else np.zeros((0, n), bool)
        return tb(bits), tb(fr), tb(rf), cov

    def amplify_device(self, pairs, d_fr_ptr, d_rf_ptr, collect_threshold, ident_threshold, amp_min=80, amp_max=200,
                       use_taq_mama=False, which=TARGET):
        a = pairs if isinstance(pairs, np.ndarray) else W.pairs_array(pairs)
        args = AmplifyArgs(collect_threshold, ident_threshold, amp_min, amp_max, int(use_taq_mama))
        self._check(self.L.pcr_amplify_device(self.h, which, a.ctypes.data, a.shape[0], C.byref(args),
                                              C.c_void_p(d_fr_ptr), C.c_void_p(d_rf_ptr)))

    def screen_device(self, pairs, select_threshold, d_fr_ptr, d_rf_ptr, collect_threshold, ident_threshold,
                      amp_min=80, amp_max=200, use_taq_mama=False, min_oligo_length=18, optimize_5=False,
                      optimize_3=False, which=TARGET):
        """select_words + amplify_device for one batch, enqueued without a host wait; the device
        buffers are final after synchronize()."""
        a = pairs if isinstance(pairs, np.ndarray) else W.pairs_array(pairs)
        args = AmplifyArgs(collect_threshold, ident_threshold, amp_min, amp_max, int(use_taq_mama))
        self._check(self.L.pcr_screen_device(self.h, which, a.ctypes.data, a.shape[0], int(optimize_5), int(optimize_3),
                                             select_threshold, min_oligo_length, C.byref(args),
                                             C.c_void_p(d_fr_ptr), C.c_void_p(d_rf_ptr)))

    def screen_call(self, pairs, select_threshold, d_fr_ptr, d_rf_ptr, collect_threshold, ident_threshold,
                    amp_min=80, amp_max=200, use_taq_mama=False, min_oligo_length=18, optimize_5=False,
                    optimize_3=False, which=TARGET):
        """screen_device() with its arguments converted once: returns a function of no arguments that enqueues the
        pass.  For loops that issue the same call again and again from Python (bench.py): the ctypes conversions of
        one call cost about as much host time as the library needs to plan a pass; a C or C++ caller has no such cost."""
        a = np.ascontiguousarray(pairs if isinstance(pairs, np.ndarray) else W.pairs_array(pairs))
        args = AmplifyArgs(collect_threshold, ident_threshold, amp_min, amp_max, int(use_taq_mama))
        fn, check = self.L.pcr_screen_device, self._check
        cargs = (C.c_void_p(self.h if isinstance(self.h, int) else self.h.value), C.c_int(which), C.c_void_p(a.ctypes.data), C.c_uint32(a.shape[0]),
                 C.c_int(int(optimize_5)), C.c_int(int(optimize_3)), C.c_float(select_threshold), C.c_uint32(min_oligo_length),
                 C.byref(args), C.c_void_p(d_fr_ptr), C.c_void_p(d_rf_ptr))

        def call(_keep=(a, args)):
            rc = fn(*cargs)
            if rc:
                check(rc)
        return call

    def move_coverage(self, base_pair, side, variants, target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200,
                      use_taq_mama=False, which=TARGET, bits=True):
        """optimize_pcr.cpp move evaluation: every variant of one oligo (side 0 = F, 1 = R) over the base pair's
        candidate amplicons -> (coverage float32[n_variants], bits_fr bool[n_variants, n], bits_rf);
        bits=False skips the per-sequence bitsets (None, None)."""
        a = W.pairs_array([base_pair])
        v = np.array([[int(w[0]), int(w[1])] for w in variants], dtype=np.uint64).reshape(-1, 2)
        V = v.shape[0]
        cov = np.zeros(max(V, 1), np.float32)
        ct = float(np.float32(target_threshold) * np.float32(search_multiplier))
        args = AmplifyArgs(ct, target_threshold, amp_min, amp_max, int(use_taq_mama))
        if not bits:
            self._check(self.L.pcr_move_coverage(self.h, which, a.ctypes.data, int(side), v.ctypes.data, V, C.byref(args),
                                                 None, None, cov.ctypes.data))
            return cov[:V], None, None
        nw = int(self.bitset_words(which))
        n = self.num_sequences(which)
        fr = np.zeros((max(V, 1), nw), np.uint64)
        rf = np.zeros((max(V, 1), nw), np.uint64)
        self._check(self.L.pcr_move_coverage(self.h, which, a.ctypes.data, int(side), v.ctypes.data, V, C.byref(args),
                                             fr.ctypes.data, rf.ctypes.data, cov.ctypes.data))
        tb = lambda x: np.stack([bits_to_bool(x[i], n) for i in range(V)]) if V else np.zeros((0, n), bool)
        return cov[:V], tb(fr), tb(rf)

    # -- the two reference evaluations, with Options-style arguments
    def find_target_match(self, pairs, target_threshold=1.0, amp_min=80, amp_max=200, use_taq_mama=False, which=TARGET):
        """PCR::find_target_match (pcr_assay.cpp:544): collect at target_threshold, test at target_threshold."""
        return self.amplify(pairs, target_threshold, target_threshold, amp_min, amp_max, use_taq_mama, which)[0]

    def compute_coverage(self, pairs, target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200,
                         use_taq_mama=False, which=TARGET):
        """optimize.cpp:61-74: collect at threshold*multiplier (float product), test at threshold."""
        ct = float(np.float32(target_threshold) * np.float32(search_multiplier))
        return self.amplify(pairs, ct, target_threshold, amp_min, amp_max, use_taq_mama, which)[3]

    # -- Smith-Waterman family (SO::SeqOverlap + background_match.cpp)
    def sw_align_words(self, queries, templates):
        """SeqOverlap lanes: query word i vs template word i -> structured array of pcr_sw_result."""
        q = np.array([[w[0], w[1]] for w in queries], dtype=np.uint64).reshape(-1, 2)
        t = np.array([[w[0], w[1]] for w in templates], dtype=np.uint64).reshape(-1, 2)
        out = (SwResult * max(len(queries), 1))()
        self._check(self.L.pcr_sw_align_words(self.h, q.ctypes.data, t.ctypes.data, len(queries), out))
        return [(r.score, r.q_start, r.q_stop, r.t_start, r.t_stop, r.last1, r.last2, r.valid) for r in out[:len(queries)]]

    def find_background_match(self, pairs, background_threshold=0.8, search_multiplier=0.9, amp_min=0, amp_max=2000,
                              use_taq_mama=False, which=BACKGROUND, evaluate_all=False):
        """PCR::find_background_match (background_match.cpp:7) -> bool [n_pairs, n].  evaluate_all=False is the
        reference bit for bit (the odd-indexed amplicon of a couple is dropped once its index reaches the number
        of sequences, background_match.cpp:122); True scores every candidate amplicon."""
        a = pairs if isinstance(pairs, np.ndarray) else W.pairs_array(pairs)
        P = a.shape[0]
        nw, n = int(self.bitset_words(which)), self.num_sequences(which)
        bits = np.zeros((P, nw), np.uint64)
        ct = float(np.float32(background_threshold) * np.float32(search_multiplier))
        args = BackgroundArgs(ct, background_threshold, amp_min, amp_max, int(use_taq_mama), int(evaluate_all))
        self._check(self.L.pcr_background_match(self.h, which, a.ctypes.data, P, C.byref(args), bits.ctypes.data))
        return np.stack([bits_to_bool(bits[i], n) for i in range(P)]) if P else np.zeros((0, n), bool)

    def find_multiplex_background_match(self, pairs, background_threshold=0.8, use_taq_mama=False, which=BACKGROUND):
        """PCR::find_multiplex_background_match (background_match.cpp:168) -> bool [n_pairs, n]."""
        a = pairs if isinstance(pairs, np.ndarray) else W.pairs_array(pairs)
        P = a.shape[0]
        nw, n = int(self.bitset_words(which)), self.num_sequences(which)
        bits = np.zeros((P, nw), np.uint64)
        self._check(self.L.pcr_multiplex_match(self.h, which, a.ctypes.data, P, background_threshold, int(use_taq_mama),
                                               bits.ctypes.data))
        return np.stack([bits_to_bool(bits[i], n) for i in range(P)]) if P else np.zeros((0, n), bool)

    # -- nearest-neighbour thermodynamics (NucCruc + valid_pcr.cpp / pcr_assay.cpp:232,815)
    @staticmethod
    def _targs(salt, primer_strand, tm_min, tm_max, max_hairpin, max_dimer):
        return ThermoArgs(salt, primer_strand, tm_min, tm_max, max_hairpin, max_dimer)

    def is_valid(self, oligos, check_homo_dimer=True, salt=0.05, primer_strand=9e-7, tm_min=50.0, tm_max=75.0,
                 max_hairpin=40.0, max_dimer=40.0, flags=False):
        """PCR::is_valid for a batch of oligo words -> list of ThermoResult-like dicts (flags=True: only the
        pass/fail bools, as an array)."""
        a = np.array([[w[0], w[1]] for w in oligos], dtype=np.uint64).reshape(-1, 2)
        out = (ThermoResult * max(len(oligos), 1))()
        args = self._targs(salt, primer_strand, tm_min, tm_max, max_hairpin, max_dimer)
        self._check(self.L.pcr_thermo(self.h, a.ctypes.data, len(oligos), int(check_homo_dimer), C.byref(args), out))
        if flags:
            return np.frombuffer(out, dtype=np.uint32).reshape(-1, 8)[:len(oligos), 0] != 0
        return [dict(valid=bool(r.valid), n=r.n_expansions, tm=np.float32(r.tm), dH=np.float32(r.dH), dS=np.float32(r.dS),
                     hairpin_tm=np.float32(r.hairpin_tm), homodimer_tm=np.float32(r.homodimer_tm), dG=np.float32(r.dG)) for r in out[:len(oligos)]]

    def collect_amplicons(self, pair, threshold=1.0, amp_min=80, amp_max=200, which=TARGET, cap=4096):
        """PCR::collect_unique_amplicons: -> [dict(sequence, begin, end, inner_start, inner_length, orientation)]
        sorted by (orientation, sequence, begin, end)."""
        a = W.pairs_array([pair])
        while True:
            buf = (Amplicon * cap)()
            n = self.L.pcr_collect_amplicons(self.h, which, a.ctypes.data, float(threshold), int(amp_min), int(amp_max), buf, cap)
            if n < 0:
                raise PcrError(_err(self.L))
            if n <= cap:
                return [dict(sequence=r.sequence, begin=r.begin, end=r.end, inner_start=r.inner_start,
                             inner_length=r.inner_length, orientation=r.orientation) for r in buf[:n]]
            cap = int(n)

    def multiplex_load(self, seqs, min_oligo_length=18):
        """The multiplex background keys (main.cpp:989-1001) from amplicon texts -> number of unique keys."""
        packs = [W.pack_codes(W.codes_from_text(s)) for s in seqs]
        off = np.zeros(max(len(seqs), 1), dtype=np.uint64)
        tot = 0
        for i, p in enumerate(packs):
            off[i] = tot
            tot += len(p)
        flat = np.concatenate(packs) if packs else np.zeros(1, np.uint8)
        ln = np.array([len(s) for s in seqs] or [0], dtype=np.uint64)
        nk = C.c_uint64(0)
        self._check(self.L.pcr_multiplex_load(self.h, flat.ctypes.data, off.ctypes.data, ln.ctypes.data, len(seqs),
                                              int(min_oligo_length), C.byref(nk)))
        return int(nk.value)

    def multiplex_coverage(self, base_pair, side, variants, background_threshold=0.8, use_taq_mama=False):
        """compute_multiplex_background_coverage for every trial word of one oligo -> float32[n_variants]."""
        a = W.pairs_array([base_pair])
        v = np.array([[int(w[0]), int(w[1])] for w in variants], dtype=np.uint64).reshape(-1, 2)
        cov = np.zeros(max(v.shape[0], 1), np.float32)
        self._check(self.L.pcr_multiplex_coverage(self.h, a.ctypes.data, int(side), v.ctypes.data, v.shape[0],
                                                  float(background_threshold), int(use_taq_mama), cov.ctypes.data))
        return cov[:v.shape[0]]

    def random_assays(self, seed, n_trials, primer_min=18, primer_max=25, amp_min=80, amp_max=200, max_degen=1.0, salt=0.05,
                      primer_strand=9e-7, tm_min=50.0, tm_max=70.0, max_hairpin=40.0, max_dimer=40.0, which=TARGET):
        """PCR::random_assay x n_trials on one running rand_r seed (main.cpp:544-550 at one thread)
        -> ([(F, R)], seed afterwards, [info dict])."""
        sa = SamplerArgs(primer_min, primer_max, amp_min, amp_max, max_degen)
        ta = self._targs(salt, primer_strand, tm_min, tm_max, max_hairpin, max_dimer)
        out = np.zeros((max(n_trials, 1), 4), dtype=np.uint64)
        info = (SampleInfo * max(n_trials, 1))()
        s = C.c_uint32(int(seed))
        self._check(self.L.pcr_random_assays(self.h, which, C.byref(s), n_trials, C.byref(sa), C.byref(ta), out.ctypes.data, info))
        pairs = [((int(r[0]), int(r[1])), (int(r[2]), int(r[3]))) for r in out[:n_trials]]
        infos = [dict(sequence=i.sequence, f_start=i.f_start, amplicon_length=i.amplicon_length,
                      sequence_iterations=i.sequence_iterations, assay_iterations=i.assay_iterations) for i in info[:n_trials]]
        return pairs, int(s.value), infos

    def max_dimer_tm(self, pairs, salt=0.05, primer_strand=9e-7):
        """PCR::max_dimer_tm for a batch of pairs -> float32 array."""
        a = pairs if isinstance(pairs, np.ndarray) else W.pairs_array(pairs)
        out = np.zeros(max(a.shape[0], 1), np.float32)
        args = self._targs(salt, primer_strand, 0.0, 0.0, 0.0, 0.0)
        self._check(self.L.pcr_dimer(self.h, a.ctypes.data, a.shape[0], C.byref(args), out.ctypes.data))
        return out[:a.shape[0]]

    def multiplex_compatible(self, assays_a, assays_b, salt=0.05, primer_strand=9e-7, max_dimer=40.0):
        """PCR::multiplex_compatible: a[i] (this) against b[i] (argument) -> bool array."""
        a, b = W.pairs_array(assays_a), W.pairs_array(assays_b)
        out = np.zeros(max(a.shape[0], 1), np.uint8)
        args = self._targs(salt, primer_strand, 0.0, 0.0, 0.0, max_dimer)
        self._check(self.L.pcr_multiplex_compatible(self.h, a.ctypes.data, b.ctypes.data, a.shape[0], C.byref(args), out.ctypes.data))
        return out[:a.shape[0]].astype(bool)

    def multiplex_screen(self, trial, pool, salt=0.05, primer_strand=9e-7, max_dimer=40.0, background_threshold=0.8, use_taq_mama=False,
                         target_threshold=1.0, amp_min=80, amp_max=200, detail=None):
        """The multiplex compatibility filter of the trial loop (main.cpp:744-803) for all trial assays ->
        (compatible bool[n], multiplex_cover float32[n], pool_cover float32[n]).  The accepted amplicons must be loaded as the
        MULTIPLEX set (sequences) and the target word DB selected for the trial batch."""
        t = W.pairs_array(list(trial))
        n = t.shape[0]
        pp = W.pairs_array(list(pool)) if pool else np.zeros((1, 4), np.uint64)
        a = MultiplexScreenArgs()
        a.thermo = self._targs(salt, primer_strand, 0.0, 0.0, 0.0, max_dimer)
        a.background_threshold, a.use_taq_mama = background_threshold, int(bool(use_taq_mama))
        a.target_threshold, a.amp_min, a.amp_max = target_threshold, int(amp_min), int(amp_max)
        ok = np.zeros(max(n, 1), np.uint8)
        mc = np.zeros(max(n, 1), np.float32)
        pc = np.zeros(max(n, 1), np.float32)
        d = None if detail is None else np.ascontiguousarray(detail, dtype=np.uint8)
        fn = self.L.pcr_multiplex_screen
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(MultiplexScreenArgs), C.c_void_p, C.c_void_p,
                       C.c_void_p, C.c_void_p]
        self._check(fn(self.h, t.ctypes.data, n, pp.ctypes.data if pool else None, len(pool) if pool else 0, C.byref(a),
                       d.ctypes.data if d is not None else None, ok.ctypes.data, mc.ctypes.data, pc.ctypes.data))
        return ok[:n].astype(bool), mc[:n], pc[:n]

    def profile(self, on=True):
        self._check(self.L.pcr_profile_enable(self.h, int(on)))

    def profile_read(self, reset=True):
        ms = C.c_double(0.0)
        n = C.c_uint64(0)
        self._check(self.L.pcr_profile_read(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def profile_read_kernel(self, kernel, reset=True):
        """kernel: 0 = match scan, 1 = Smith-Waterman, 2 = thermodynamics -> (ms, launches) since the last reset."""
        ms = C.c_double(0.0)
        n = C.c_uint64(0)
        self._check(self.L.pcr_profile_read_kernel(self.h, int(kernel), C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    # ---- the bitset exchange behind the ABI (RCCL all-gather on this handle's stream; include/pcramp_hip.h)
    @staticmethod
    def comm_unique_id():
        """Rank 0: the 128 opaque bytes every rank passes to comm_init_rank."""
        L = load_library()
        buf = (C.c_uint8 * 128)()
        if L.pcr_comm_unique_id(buf) != 0:
            raise PcrError(_err(L))
        return bytes(buf)

    def comm_init_rank(self, unique_id, world, rank):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        h = self.L.pcr_comm_init_rank(self.h, buf, int(world), int(rank))
        if not h:
            raise PcrError(_err(self.L))
        return h

    def exchange_bits(self, comm, d_local_ptr, words_per_rank, d_full_ptr):
        """ncclAllGather of words_per_rank u64 per rank into d_full [world, words_per_rank] (device pointers), in stream order."""
        self._check(self.L.pcr_exchange_bits(self.h, comm, int(d_local_ptr), int(words_per_rank), int(d_full_ptr)))

    def comm_destroy(self, comm):
        self.L.pcr_comm_destroy(comm)

    def staging_mode(self):
        """'lean' (the CPU stores the per-pass tables straight into device memory, no staging launch) or 'k_stage'."""
        return "lean" if self.L.pcr_staging_mode(self.h) else "k_stage"

    def synchronize(self):
        self._check(self.L.pcr_synchronize(self.h))
