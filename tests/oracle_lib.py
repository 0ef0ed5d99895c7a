"""ctypes bindings for the two CPU checkers (TEST INFRASTRUCTURE, never used by the product):

* ``Oracle``    -- oracle/liboracle.so, our scalar C++ restatement of the reference path.
* ``Reference`` -- oracle/_ref/libpcramp_ref.so, the real reference compiled from /root/reference
                   behind oracle/ref_harness.cpp (only exists where oracle/Makefile could build it).

Both expose the same Python surface so the tests can run one against the other.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class Entry(C.Structure):
    _fields_ = [("w", C.c_uint64 * 2), ("loc", C.c_int32), ("index", C.c_uint32),
                ("strand", C.c_uint32), ("pad", C.c_uint32)]


class SWResult(C.Structure):
    _fields_ = [("score", C.c_int16), ("q_start", C.c_int16), ("q_stop", C.c_int16), ("t_start", C.c_int16),
                ("t_stop", C.c_int16), ("last1", C.c_uint8), ("last2", C.c_uint8), ("valid", C.c_uint8), ("pad", C.c_uint8)]

    def tup(self):
        return (self.score, self.q_start, self.q_stop, self.t_start, self.t_stop, self.last1, self.last2)


class OrcOptions(C.Structure):
    _fields_ = [("target_threshold", C.c_float), ("search_multiplier", C.c_float),
                ("amp_min", C.c_int32), ("amp_max", C.c_int32), ("use_taq_mama", C.c_int32),
                ("pack_max_degen", C.c_uint32), ("pack_min_gc", C.c_float), ("pack_max_gc", C.c_float),
                ("min_primer", C.c_int32), ("optimize_5", C.c_int32), ("optimize_3", C.c_int32)]


DEFAULT_OPTIONS = dict(target_threshold=1.0, search_multiplier=0.9, amp_min=80, amp_max=200,
                       use_taq_mama=0, pack_max_degen=256, pack_min_gc=0.0, pack_max_gc=1.0,
                       min_primer=18, optimize_5=0, optimize_3=0)

U64x2 = C.c_uint64 * 2


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def build_reference():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


def _w(x):
    return U64x2(int(x[0]), int(x[1]))


class _Lib:
    prefix = None
    path = None

    def __init__(self):
        self.lib = C.CDLL(self.path)
        L, p = self.lib, self.prefix
        f = lambda n: getattr(L, p + n)
        f("word_and").restype = C.c_uint
        f("word_size").restype = C.c_uint
        f("word_degeneracy").restype = C.c_double
        f("taq_mama").restype = C.c_float
        f("taq_mama").argtypes = [C.c_uint] * 4
        f("pack").restype = C.c_long
        f("pack").argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_float, C.c_float, C.c_uint,
                              C.POINTER(Entry), C.c_long]
        f("session_create").restype = C.c_void_p
        f("session_destroy").argtypes = [C.c_void_p]
        f("session_error").restype = C.c_char_p
        f("session_error").argtypes = [C.c_void_p]
        f("session_add_target").argtypes = [C.c_void_p, C.c_char_p, C.c_float, C.c_int]
        f("session_set_active").argtypes = [C.c_void_p, C.c_uint, C.c_int]
        f("session_split").argtypes = [C.c_void_p, C.c_uint, C.c_uint]
        f("session_select").restype = C.c_long
        f("session_select").argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_float, C.c_int]
        f("session_db_entries").restype = C.c_long
        f("session_db_entries").argtypes = [C.c_void_p, C.POINTER(Entry), C.c_long]
        f("session_target_coverage").restype = C.c_float
        f("session_target_coverage").argtypes = [C.c_void_p, C.c_void_p]
        self._f = f
        f("thermo_full").argtypes = [C.c_char_p, C.c_float, C.c_float, C.c_void_p]
        f("heterodimer_full").argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
        f("is_valid").argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]
        f("max_dimer_tm").restype = C.c_float
        f("max_dimer_tm").argtypes = [C.c_void_p, C.c_float, C.c_float]
        f("multiplex_compatible").argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float]

    # ---- words
    def word(self, s):
        out = U64x2()
        if self._f("word_from_string")(s.encode(), out) != 0:
            raise ValueError("bad word " + s)
        return (out[0], out[1])

    def word_and(self, a, b):
        return self._f("word_and")(_w(a), _w(b))

    def word_size(self, a):
        return self._f("word_size")(_w(a))

    def word_start(self, a):
        return self._f("word_start")(_w(a))

    def word_stop(self, a):
        return self._f("word_stop")(_w(a))

    def word_degeneracy(self, a):
        return self._f("word_degeneracy")(_w(a))

    def word_center(self, a):
        x = _w(a)
        self._f("word_center")(x)
        return (x[0], x[1])

    def word_complement(self, a):
        o = U64x2()
        self._f("word_complement")(_w(a), o)
        return (o[0], o[1])

    def word_shift_left(self, a):
        x = _w(a)
        self._f("word_shift_left")(x)
        return (x[0], x[1])

    def word_shift_right(self, a):
        x = _w(a)
        self._f("word_shift_right")(x)
        return (x[0], x[1])

    def word_expand(self, a, cap=4096):
        buf = (C.c_uint64 * (2 * cap))()
        n = self._f("word_expand")(_w(a), buf, cap)
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(min(n, cap))]

    def centered_word(self, s):
        return self.word_center(self.word(s))

    def max_overlap(self, a, b):
        fn = self._f("word_max_overlap")
        fn.restype = C.c_float
        return fn(_w(a), _w(b))

    def oligo_overlap(self, assay, pool):
        fn = getattr(self.lib, self.prefix + "oligo_overlap")
        fn.restype = C.c_float
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        a = pairs_array([assay])
        p = pairs_array(pool) if len(pool) else np.zeros((1, 4), dtype=np.uint64)
        return fn(a.ctypes.data, p.ctypes.data, len(pool))

    def format_oligos(self, assay, pool=(), json=False, with_pool=True):
        """Reference only: PCR::write / write_json (assay.h:288-375) -> bytes."""
        fn = self.lib.ref_format_oligos
        fn.restype = C.c_long
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_char_p, C.c_long]
        a = pairs_array([assay])
        p = pairs_array(list(pool)) if len(pool) else np.zeros((1, 4), dtype=np.uint64)
        buf = C.create_string_buffer(4096)
        n = fn(a.ctypes.data, p.ctypes.data, len(pool), int(json), int(with_pool), buf, 4096)
        assert 0 <= n <= 4096
        return buf.raw[:n]

    def taq_mama(self, p1, p2, t1, t2):
        return self._f("taq_mama")(p1, p2, t1, t2)

    # ---- thermodynamics
    def thermo_full(self, seq, salt=0.05, strand=9e-7):
        out = np.zeros(10, np.float32)
        rc = self._f("thermo_full")(seq.encode(), salt, strand, out.ctypes.data)
        assert rc == 0, rc
        return out

    def heterodimer_full(self, a, b, salt=0.05, strand_a=9e-7, strand_b=9e-7):
        out = np.zeros(3, np.float32)
        rc = self._f("heterodimer_full")(a.encode(), b.encode(), salt, strand_a, strand_b, out.ctypes.data)
        assert rc == 0, rc
        return out

    def is_valid(self, word, salt=0.05, primer_strand=9e-7, tm_min=50.0, tm_max=75.0, max_hairpin=40.0, max_dimer=40.0,
                 check_homo_dimer=True):
        w = _w(word)
        return self._f("is_valid")(w, salt, primer_strand, tm_min, tm_max, max_hairpin, max_dimer, int(check_homo_dimer))

    def max_dimer_tm(self, pair, salt=0.05, primer_strand=9e-7):
        a = pairs_array([pair])
        return self._f("max_dimer_tm")(a.ctypes.data, salt, primer_strand)

    def multiplex_compatible(self, a, b, salt=0.05, primer_strand=9e-7, max_dimer=40.0):
        x, y = pairs_array([a]), pairs_array([b])
        return self._f("multiplex_compatible")(x.ctypes.data, y.ctypes.data, salt, primer_strand, max_dimer)

    # ---- pack
    def pack(self, seq, index=0, degen_thr=256, min_gc=0.0, max_gc=1.0, min_len=18):
        cap = 2 * len(seq) + 256
        buf = (Entry * cap)()
        n = self._f("pack")(seq.encode(), index, degen_thr, min_gc, max_gc, min_len, buf, cap)
        if n < 0:
            raise RuntimeError("pack failed")
        assert n <= cap
        return sorted((e.w[0], e.w[1], e.loc, e.index, e.strand) for e in buf[:n])

    # ---- session
    def session(self, **opts):
        return Session(self, **opts)


class Oracle(_Lib):
    prefix = "orc_"
    path = os.path.join(ORACLE_DIR, "liboracle.so")

    def __init__(self):
        if not os.path.exists(self.path):
            build_oracle()
        super().__init__()
        self.lib.orc_session_target_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.orc_session_add_target_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_int]
        self.lib.orc_weighted_coverage.restype = C.c_float
        self.lib.orc_weighted_coverage.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.orc_sw_align.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(SWResult)]
        self.lib.orc_sw_align_words.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(SWResult)]
        self.lib.orc_session_background_match.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int,
                                                          C.c_int, C.c_int, C.c_int, C.c_void_p]
        self.lib.orc_session_multiplex_match.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]

    def sw_align_codes(self, q, t):
        """q, t: uint8 arrays of 4-bit codes -> SWResult."""
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        r = SWResult()
        self.lib.orc_sw_align(q.ctypes.data, q.size, t.ctypes.data, t.size, C.byref(r))
        return r

    def sw_align_words(self, q, t):
        r = SWResult()
        self.lib.orc_sw_align_words(_w(q), _w(t), C.byref(r))
        return r


class Reference(_Lib):
    prefix = "ref_"
    path = os.path.join(ORACLE_DIR, "_ref", "libpcramp_ref.so")

    def __init__(self):
        super().__init__()
        self.lib.ref_session_target_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.ref_session_set_options.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int,
                                                     C.c_int, C.c_uint, C.c_float, C.c_float, C.c_int,
                                                     C.c_int, C.c_int]
        self.lib.ref_session_background_match.restype = C.c_long
        self.lib.ref_session_background_match.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int,
                                                          C.c_int, C.c_int, C.c_void_p]
        self.lib.ref_session_multiplex_match.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        self.lib.ref_sw_align_words.argtypes = [C.c_void_p] * 6

    def sw_align_words8(self, queries, targets):
        """One 8-lane SeqOverlap call: lists of 8 query words / 8 target words -> list of 8 tuples
        (score, q_start, q_stop, t_start, t_stop, last1, last2)."""
        q = np.array([[w[0], w[1]] for w in queries], dtype=np.uint64)
        t = np.array([[w[0], w[1]] for w in targets], dtype=np.uint64)
        score = np.zeros(8, np.int16)
        qr = np.zeros(16, np.int32)
        tr = np.zeros(16, np.int32)
        l2 = np.zeros(16, np.uint8)
        rc = self.lib.ref_sw_align_words(q.ctypes.data, t.ctypes.data, score.ctypes.data, qr.ctypes.data,
                                         tr.ctypes.data, l2.ctypes.data)
        assert rc == 0
        return [(int(score[i]), int(qr[2 * i]), int(qr[2 * i + 1]), int(tr[2 * i]), int(tr[2 * i + 1]),
                 int(l2[2 * i]), int(l2[2 * i + 1])) for i in range(8)]

    @classmethod
    def available(cls):
        return os.path.exists(cls.path)


class MoveOptions(C.Structure):
    _fields_ = [("degen", C.c_int), ("primer_min", C.c_int), ("primer_max", C.c_int),
                ("salt", C.c_float), ("primer_strand", C.c_float), ("tm_min", C.c_float), ("tm_max", C.c_float),
                ("max_hairpin", C.c_float), ("bg_threshold", C.c_float), ("bg_multiplier", C.c_float),
                ("bg_amp_min", C.c_int), ("bg_amp_max", C.c_int)]


DEFAULT_MOVE_OPTIONS = dict(degen=1, primer_min=18, primer_max=25, salt=0.05, primer_strand=9.0e-7, tm_min=50.0,
                            tm_max=70.0, max_hairpin=40.0, bg_threshold=0.8, bg_multiplier=0.9, bg_amp_min=0,
                            bg_amp_max=2000)


def optimization_move(lib, target_session, background_session, pair, move, side, **mo):
    """One complete local-search move (optimize_pcr.cpp) -> ((w0, w1), (tc, bc, overlap), (base tc, base bc))."""
    o = dict(DEFAULT_MOVE_OPTIONS)
    o.update(mo)
    opts = MoveOptions(**o)
    a = pairs_array([pair])
    w = np.zeros(2, dtype=np.uint64)
    sc = np.zeros(3, dtype=np.float32)
    base = np.zeros(2, dtype=np.float32)
    fn = getattr(lib.lib, lib.prefix + "optimization_move")
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(MoveOptions), C.c_void_p, C.c_void_p,
                   C.c_void_p]
    rc = fn(target_session.h, background_session.h if background_session is not None else None, a.ctypes.data, int(move),
            int(side), C.byref(opts), w.ctypes.data, sc.ctypes.data, base.ctypes.data)
    if rc != 0:
        raise RuntimeError(target_session.f("session_error")(target_session.h))
    return (int(w[0]), int(w[1])), tuple(float(x) for x in sc), tuple(float(x) for x in base)


def optimize(lib, target_session, background_session, pair, moves=(0, 1, 2, 4, 3, 5), **mo):
    """optimize() (optimize.cpp:14-207), non-multiplex; moves in main.cpp:82-95 order by default
    -> (best pair, (tc, bc, overlap))."""
    o = dict(DEFAULT_MOVE_OPTIONS)
    o.update(mo)
    opts = MoveOptions(**o)
    a = pairs_array([pair]).copy()
    mv = np.array(list(moves), dtype=np.int32)
    sc = np.zeros(3, dtype=np.float32)
    it = C.c_int(0)
    fn = getattr(lib.lib, lib.prefix + "optimize")
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(MoveOptions), C.c_void_p, C.POINTER(C.c_int)]
    rc = fn(target_session.h, background_session.h if background_session is not None else None, a.ctypes.data, mv.ctypes.data,
            len(mv), C.byref(opts), sc.ctypes.data, C.byref(it))
    if rc != 0:
        raise RuntimeError(target_session.f("session_error")(target_session.h))
    flat = a.reshape(-1)
    return ((int(flat[0]), int(flat[1])), (int(flat[2]), int(flat[3]))), tuple(float(x) for x in sc)


def make_degenerate(lib, target_session, pair, max_dimer=40.0, **mo):
    """make_degenerate (optimize.cpp:356-398 -> PCR::maximize_degeneracy) -> (assay, valid)."""
    o = dict(DEFAULT_MOVE_OPTIONS)
    o.update(mo)
    opts = MoveOptions(**o)
    a = pairs_array([pair]).copy()
    ok = C.c_int(0)
    fn = getattr(lib.lib, lib.prefix + "make_degenerate")
    fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(MoveOptions), C.c_float, C.POINTER(C.c_int)]
    rc = fn(target_session.h, a.ctypes.data, C.byref(opts), float(max_dimer), C.byref(ok))
    if rc != 0:
        raise RuntimeError(target_session.f("session_error")(target_session.h))
    flat = a.reshape(-1)
    return ((int(flat[0]), int(flat[1])), (int(flat[2]), int(flat[3]))), bool(ok.value)


def optimization_move_multiplex(lib, target_session, background_session, amplicon_session, pool, pair, move, side, **mo):
    """One local-search move with opt.use_multiplex -> ((w0, w1), (tc, bc, overlap), base (tc, bc, overlap))."""
    o = dict(DEFAULT_MOVE_OPTIONS)
    o.update(mo)
    opts = MoveOptions(**o)
    a = pairs_array([pair])
    pw = pairs_array(pool) if len(pool) else np.zeros((1, 4), dtype=np.uint64)
    w = np.zeros(2, dtype=np.uint64)
    sc = np.zeros(3, dtype=np.float32)
    base = np.zeros(3, dtype=np.float32)
    fn = getattr(lib.lib, lib.prefix + "optimization_move_multiplex")
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int, C.c_int, C.POINTER(MoveOptions),
                   C.c_void_p, C.c_void_p, C.c_void_p]
    rc = fn(target_session.h, background_session.h if background_session is not None else None, amplicon_session.h,
            pw.ctypes.data, len(pool), a.ctypes.data, int(move), int(side), C.byref(opts), w.ctypes.data, sc.ctypes.data,
            base.ctypes.data)
    if rc != 0:
        raise RuntimeError(target_session.f("session_error")(target_session.h))
    return (int(w[0]), int(w[1])), tuple(float(x) for x in sc), tuple(float(x) for x in base)


def optimize_multiplex(lib, target_session, background_session, amplicon_session, pool, pair, moves=(0, 1, 2, 4, 3, 5), **mo):
    """optimize() with opt.use_multiplex: multiplex background = pack of amplicon_session's sequences, pool = assays
    designed so far -> (best pair, (tc, bc incl. the multiplex term, oligo_overlap))."""
    o = dict(DEFAULT_MOVE_OPTIONS)
    o.update(mo)
    opts = MoveOptions(**o)
    a = pairs_array([pair]).copy()
    pw = pairs_array(pool) if len(pool) else np.zeros((1, 4), dtype=np.uint64)
    mv = np.array(list(moves), dtype=np.int32)
    sc = np.zeros(3, dtype=np.float32)
    fn = getattr(lib.lib, lib.prefix + "optimize_multiplex")
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_int,
                   C.POINTER(MoveOptions), C.c_void_p]
    rc = fn(target_session.h, background_session.h if background_session is not None else None, amplicon_session.h,
            pw.ctypes.data, len(pool), a.ctypes.data, mv.ctypes.data, len(mv), C.byref(opts), sc.ctypes.data)
    if rc != 0:
        raise RuntimeError(target_session.f("session_error")(target_session.h))
    r = a[0]
    return ((int(r[0]), int(r[1])), (int(r[2]), int(r[3]))), tuple(float(x) for x in sc)


class SamplerOptions(C.Structure):
    _fields_ = [("primer_min", C.c_int), ("primer_max", C.c_int), ("amp_min", C.c_int), ("amp_max", C.c_int),
                ("max_degen", C.c_double), ("salt", C.c_float), ("primer_strand", C.c_float), ("tm_min", C.c_float),
                ("tm_max", C.c_float), ("max_hairpin", C.c_float), ("max_dimer", C.c_float)]


DEFAULT_SAMPLER_OPTIONS = dict(primer_min=18, primer_max=25, amp_min=80, amp_max=200, max_degen=1.0, salt=0.05,
                               primer_strand=9.0e-7, tm_min=50.0, tm_max=70.0, max_hairpin=40.0, max_dimer=40.0)


def rand_r(lib, seed):
    """glibc rand_r -> (value, next seed)."""
    s = C.c_uint(seed)
    fn = getattr(lib.lib, lib.prefix + "rand_r")
    fn.restype = C.c_uint
    fn.argtypes = [C.POINTER(C.c_uint)]
    return fn(C.byref(s)), s.value


def random_assays(lib, session, seed, n_trials, **so):
    """PCR::random_assay x n_trials on one running seed (main.cpp:544-550 at one thread)
    -> ([(F, R)], seed afterwards)."""
    o = dict(DEFAULT_SAMPLER_OPTIONS)
    o.update(so)
    opts = SamplerOptions(**o)
    out = np.zeros((n_trials, 4), dtype=np.uint64)
    s = C.c_uint(seed)
    fn = getattr(lib.lib, lib.prefix + "random_assays")
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.c_uint, C.POINTER(SamplerOptions), C.c_void_p]
    rc = fn(session.h, C.byref(s), n_trials, C.byref(opts), out.ctypes.data)
    if rc != 0:
        raise RuntimeError(session.f("session_error")(session.h))
    return [((int(r[0]), int(r[1])), (int(r[2]), int(r[3]))) for r in out], s.value


def pairs_array(pairs):
    """pairs: list of (F, R) with F, R = (u64, u64) -> contiguous uint64 [n, 4]."""
    a = np.zeros((len(pairs), 4), dtype=np.uint64)
    for i, (f, r) in enumerate(pairs):
        a[i, 0], a[i, 1], a[i, 2], a[i, 3] = f[0], f[1], r[0], r[1]
    return a


class Session:
    def __init__(self, lib, **opts):
        self.L = lib
        self.f = lib._f
        self.h = self.f("session_create")()
        self.n = 0
        self.opts = dict(DEFAULT_OPTIONS)
        self.set_options(**opts)

    def __del__(self):
        try:
            self.f("session_destroy")(self.h)
        except Exception:
            pass

    def set_options(self, **opts):
        self.opts.update(opts)
        o = self.opts
        if isinstance(self.L, Oracle):
            s = OrcOptions(**o)
            self.f("session_set_options")(C.c_void_p(self.h), C.byref(s))
        else:
            self.f("session_set_options")(self.h, o["target_threshold"], o["search_multiplier"],
                                          o["amp_min"], o["amp_max"], o["use_taq_mama"],
                                          o["pack_max_degen"], o["pack_min_gc"], o["pack_max_gc"],
                                          o["min_primer"], o["optimize_5"], o["optimize_3"])

    def add_target(self, seq, weight=1.0, active=True):
        if self.f("session_add_target")(self.h, seq.encode(), weight, int(active)) != 0:
            raise RuntimeError(self.f("session_error")(self.h))
        self.n += 1

    def add_target_packed(self, packed, length, weight=1.0, active=True):
        assert isinstance(self.L, Oracle)
        buf = np.ascontiguousarray(packed, dtype=np.uint8)
        self.L.lib.orc_session_add_target_packed(self.h, buf.ctypes.data, int(length), weight, int(active))
        self.n += 1

    def set_active(self, idx, active):
        assert self.f("session_set_active")(self.h, idx, int(active)) == 0

    def split(self, idx, pos):
        assert self.f("session_split")(self.h, idx, pos) == 0

    def select(self, pairs, threshold=-1.0, min_len_override=-1):
        a = pairs_array(pairs)
        n = self.f("session_select")(self.h, a.ctypes.data, len(pairs), threshold, min_len_override)
        if n < 0:
            raise RuntimeError(self.f("session_error")(self.h))
        return n

    def db_entries(self):
        cap = 1 << 16
        while True:
            buf = (Entry * cap)()
            n = self.f("session_db_entries")(self.h, buf, cap)
            if n <= cap:
                break
            cap = n
        return sorted((e.w[0], e.w[1], e.loc, e.index, e.strand) for e in buf[:n])

    def target_match(self, pair, orient=False):
        a = pairs_array([pair])
        bits = np.zeros(self.n, dtype=np.uint8)
        if isinstance(self.L, Oracle):
            ori = np.zeros(self.n, dtype=np.uint8)
            rc = self.L.lib.orc_session_target_match(self.h, a.ctypes.data, bits.ctypes.data, ori.ctypes.data)
            if rc != 0:
                raise RuntimeError(self.f("session_error")(self.h))
            return (bits, ori) if orient else bits
        rc = self.L.lib.ref_session_target_match(self.h, a.ctypes.data, bits.ctypes.data)
        if rc != 0:
            raise RuntimeError(self.f("session_error")(self.h))
        return bits

    def target_coverage(self, pair):
        a = pairs_array([pair])
        return self.f("session_target_coverage")(self.h, a.ctypes.data)

    def move_coverage(self, base_pair, side, variants, orient=False):
        """optimize_pcr.cpp move evaluation: coverage of every variant of one oligo (side 0 = F, 1 = R)
        over the base pair's candidate amplicons -> float32[n_variants] (oracle: optionally the
        per-sequence orientation bits uint8[n_variants, n])."""
        a = pairs_array([base_pair])
        v = np.array([[int(w[0]), int(w[1])] for w in variants], dtype=np.uint64).reshape(-1, 2)
        cov = np.zeros(max(len(variants), 1), dtype=np.float32)
        if isinstance(self.L, Oracle):
            ori = np.zeros((max(len(variants), 1), max(self.n, 1)), dtype=np.uint8)
            fn = self.L.lib.orc_session_move_coverage
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
            rc = fn(self.h, a.ctypes.data, int(side), v.ctypes.data, len(variants), cov.ctypes.data, ori.ctypes.data)
            if rc != 0:
                raise RuntimeError(self.f("session_error")(self.h))
            return (cov[:len(variants)], ori[:len(variants), :self.n]) if orient else cov[:len(variants)]
        fn = self.L.lib.ref_session_move_coverage
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint, C.c_void_p]
        rc = fn(self.h, a.ctypes.data, int(side), v.ctypes.data, len(variants), cov.ctypes.data)
        if rc != 0:
            raise RuntimeError(self.f("session_error")(self.h))
        return cov[:len(variants)]

    def collect_amplicons(self, pair, threshold=1.0, amp_min=80, amp_max=200):
        """PCR::collect_unique_amplicons -> (bounds [(seq, begin, end)] in discovery order,
        unique amplicons as tuples of nibbles in the reference's sorted order)."""
        a = pairs_array([pair])
        cap_b, cap_c, cap_a = 1 << 16, 1 << 22, 1 << 16
        bounds = np.zeros(3 * cap_b, dtype=np.uint32)
        codes = np.zeros(cap_c, dtype=np.uint8)
        lens = np.zeros(cap_a, dtype=np.uint32)
        na = C.c_long(0)
        fn = self.f("session_collect_amplicons")
        fn.restype = C.c_long
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                       C.c_void_p, C.c_long, C.POINTER(C.c_long)]
        nb = fn(self.h, a.ctypes.data, threshold, amp_min, amp_max, bounds.ctypes.data, cap_b, codes.ctypes.data, cap_c,
                lens.ctypes.data, cap_a, C.byref(na))
        if nb < 0:
            raise RuntimeError("collect_amplicons failed: %d" % nb)
        amps, off = [], 0
        for k in range(na.value):
            amps.append(tuple(int(x) for x in codes[off:off + lens[k]]))
            off += int(lens[k])
        return [tuple(int(x) for x in bounds[3 * i:3 * i + 3]) for i in range(nb)], amps

    def multiplex_coverage(self, base_pair, side, variants, background_threshold=0.8, use_taq_mama=0):
        """The session's sequences as accepted amplicons -> (float32[n_variants], number of keys)."""
        a = pairs_array([base_pair])
        v = np.array([[int(w[0]), int(w[1])] for w in variants], dtype=np.uint64).reshape(-1, 2)
        cov = np.zeros(max(len(variants), 1), dtype=np.float32)
        nk = C.c_uint(0)
        fn = getattr(self.L.lib, self.L.prefix + "multiplex_coverage")
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint, C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_uint)]
        rc = fn(self.h, a.ctypes.data, int(side), v.ctypes.data, len(variants), background_threshold, int(use_taq_mama),
                cov.ctypes.data, C.byref(nk))
        if rc != 0:
            raise RuntimeError(self.f("session_error")(self.h))
        return cov[:len(variants)], nk.value

    def background_match(self, pair, bg_threshold=0.8, bg_multiplier=0.9, amp_min=0, amp_max=2000, use_taq_mama=0,
                         emulate_index_bug=0):
        """-> (bits uint8[n], n_amplicons or None).  Reference: returns None bits when the reference's
        odd-count out-of-bounds path would be hit."""
        a = pairs_array([pair])
        bits = np.zeros(self.n, dtype=np.uint8)
        if isinstance(self.L, Oracle):
            rc = self.L.lib.orc_session_background_match(self.h, a.ctypes.data, bg_threshold, bg_multiplier, amp_min,
                                                         amp_max, use_taq_mama, emulate_index_bug, bits.ctypes.data)
            if rc != 0:
                raise RuntimeError(self.f("session_error")(self.h))
            return bits, None
        n = self.L.lib.ref_session_background_match(self.h, a.ctypes.data, bg_threshold, bg_multiplier, amp_min,
                                                    amp_max, use_taq_mama, bits.ctypes.data)
        if n == -3:
            return None, None
        if n < 0:
            raise RuntimeError(self.f("session_error")(self.h))
        return bits, n

    def multiplex_match(self, pair, bg_threshold=0.8, use_taq_mama=0):
        a = pairs_array([pair])
        bits = np.zeros(self.n, dtype=np.uint8)
        fn = self.L.lib.orc_session_multiplex_match if isinstance(self.L, Oracle) else self.L.lib.ref_session_multiplex_match
        assert fn(self.h, a.ctypes.data, bg_threshold, use_taq_mama, bits.ctypes.data) == 0
        return bits

    def weighted_coverage(self, bits):
        assert isinstance(self.L, Oracle)
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        return self.L.lib.orc_weighted_coverage(self.h, b.ctypes.data)
