"""pcramp_amd -- MI355X-native primer-pair x target evaluation path (in-silico PCR screening).

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C-ABI of
``include/pcramp_hip.h``), ``api.py`` (the host-side mirror of the reference call sites over
that ABI), ``words.py`` (the reference's Word / Sequence packing conventions) and ``synth.py``
(the seeded synthetic workloads of BASELINE.json).  There is no CPU fallback: the compute
entry points raise if ``libpcramp_hip.so`` or a gfx950 GPU is missing.
"""
from .api import PcrError, Screener, AmplifyArgs, load_library, library_path  # noqa: F401
from . import words, synth  # noqa: F401

__all__ = ["PcrError", "Screener", "AmplifyArgs", "load_library", "library_path", "words", "synth"]
