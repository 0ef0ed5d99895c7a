"""The reference PROGRAM's design loop behind one ABI call (scope row f-7): pcr_design runs main.cpp:471-1130 on the
loaded sets and leaves the bytes of the reference's output file in the handle.  This module only packs arguments --
`Options` defaults as options.cpp:20-98 sets them, command-line switches as options.cpp:161-214 names them -- so that a
test can replay a reference command line."""
import ctypes as C

import numpy as np

from . import api, words as W


class DesignArgs(C.Structure):
    _fields_ = [("num_assay", C.c_uint32), ("num_trial", C.c_uint32), ("seed", C.c_uint32), ("top_down_search", C.c_int32),
                ("optimize_5", C.c_int32), ("optimize_3", C.c_int32),
                ("target_threshold", C.c_float), ("target_search_multiplier", C.c_float),
                ("background_threshold", C.c_float), ("background_search_multiplier", C.c_float),
                ("min_target_cover", C.c_float), ("max_background_cover", C.c_float),
                ("target_amp_min", C.c_int32), ("target_amp_max", C.c_int32), ("background_amp_min", C.c_int32), ("background_amp_max", C.c_int32),
                ("primer_min", C.c_int32), ("primer_max", C.c_int32), ("max_degen", C.c_double),
                ("thermo", api.ThermoArgs), ("use_taq_mama", C.c_int32), ("use_multiplex", C.c_int32)]


# Options::Options(), options.cpp:20-98 / pcramp.h:14-52
DEFAULTS = dict(num_assay=100, num_trial=1000, seed=0, top_down_search=0, optimize_5=0, optimize_3=0,
                target_threshold=1.0, target_search_multiplier=0.9, background_threshold=0.8, background_search_multiplier=0.9,
                min_target_cover=0.0, max_background_cover=0.0, target_amp_min=80, target_amp_max=200,
                background_amp_min=0, background_amp_max=2000, primer_min=18, primer_max=25, max_degen=1.0,
                salt=0.05, primer_strand=900.0e-9, tm_min=50.0, tm_max=75.0, max_hairpin=40.0, max_dimer=40.0,
                use_taq_mama=0, use_multiplex=1, json=0)

# the switches of options.cpp:161-214 the design loop reads: name -> (option, type); flags have type None
SWITCHES = {"--count": ("num_assay", int), "--trial": ("num_trial", int), "--seed": ("seed", int), "-d": ("max_degen", float),
            "--optimize.top-down": ("top_down_search", None), "--optimize.5": ("optimize_5", None), "--optimize.3": ("optimize_3", None),
            "--target.threshold": ("target_threshold", float), "--background.threshold": ("background_threshold", float),
            "--target.amplicon.min": ("target_amp_min", int), "--target.amplicon.max": ("target_amp_max", int),
            "--background.amplicon.min": ("background_amp_min", int), "--background.amplicon.max": ("background_amp_max", int),
            "--target.cover": ("min_target_cover", float), "--background.cover": ("max_background_cover", float),
            "--salt": ("salt", float), "--primer.hairpin": ("max_hairpin", float), "--primer.dimer": ("max_dimer", float),
            "--primer.size.min": ("primer_min", int), "--primer.size.max": ("primer_max", int), "--primer.tm.min": ("tm_min", float),
            "--primer.tm.max": ("tm_max", float), "--primer.strand": ("primer_strand", float),
            "--target.search": ("target_search_multiplier", float), "--background.search": ("background_search_multiplier", float),
            "--o.json": ("json", None), "--primer.taq-mama": ("use_taq_mama", None)}
IGNORED_WITH_VALUE = {"-t", "-b", "-o", "--thread"}


def options_from_argv(argv):
    """The options a reference command line sets (argv[0] is the program name); unknown switches raise."""
    o = dict(DEFAULTS)
    i = 1
    while i < len(argv):
        a = argv[i]
        if a in IGNORED_WITH_VALUE:
            i += 2
            continue
        if a not in SWITCHES:
            raise ValueError("design.options_from_argv: switch %r is not mapped" % a)
        name, typ = SWITCHES[a]
        if typ is None:
            o[name] = 1
            i += 1
        else:
            o[name] = typ(argv[i + 1])
            i += 2
    return o


def design(scr, target_deflines, target_lengths, background_deflines=(), background_lengths=(), argv=("pcramp",), **opts):
    """pcr_design on the Screener's loaded TARGET / BACKGROUND sets -> (output file bytes, accepted assays)."""
    o = dict(DEFAULTS)
    o.update(opts)
    w = api.AssayWriter(list(target_deflines), list(target_lengths), list(background_deflines), list(background_lengths),
                        json=bool(o["json"]), use_multiplex=bool(o["use_multiplex"]))
    a = DesignArgs()
    for k in ("num_assay", "num_trial", "seed", "top_down_search", "optimize_5", "optimize_3", "target_threshold", "target_search_multiplier",
              "background_threshold", "background_search_multiplier", "min_target_cover", "max_background_cover", "target_amp_min", "target_amp_max",
              "background_amp_min", "background_amp_max", "primer_min", "primer_max", "max_degen", "use_taq_mama", "use_multiplex"):
        setattr(a, k, o[k])
    a.thermo = api.ThermoArgs(o["salt"], o["primer_strand"], o["tm_min"], o["tm_max"], o["max_hairpin"], o["max_dimer"])
    L = scr.L
    L.pcr_design.restype = C.c_int
    L.pcr_design.argtypes = [C.c_void_p, C.POINTER(DesignArgs), C.POINTER(api.Output), C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_uint32,
                             C.POINTER(C.c_uint32)]
    L.pcr_design_output.restype = C.c_void_p
    L.pcr_design_output.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    av = (C.c_char_p * len(argv))(*[x.encode() for x in argv])
    cap = int(o["num_assay"]) + 1
    pool = np.zeros((cap, 4), dtype=np.uint64)
    n_pool = C.c_uint32(0)
    rc = L.pcr_design(scr.h, C.byref(a), C.byref(w.o), len(argv), av, pool.ctypes.data, cap, C.byref(n_pool))
    n = C.c_uint64(0)
    ptr = L.pcr_design_output(scr.h, C.byref(n))
    text = C.string_at(ptr, n.value) if ptr else b""
    if rc != 0:
        raise api.PcrError("pcr_design: %s (output so far: %d bytes)" % (api._err(L), len(text)))
    pairs = [((int(r[0]), int(r[1])), (int(r[2]), int(r[3]))) for r in pool[:n_pool.value]]
    return text, pairs
