"""The reference-side adapter of INTEGRATION.md EXECUTED (oracle/adapter_run.cpp): in one process, deque<Sequence>, vector<PCR>
and Options built with the reference's own constructors go through DeviceScreen (oracle/adapter_check.cpp) to libpcramp_hip.so
AND through the reference's own functions (Sequence::pack + select_words, PCR::find_target_match, compute_target_coverage,
PCR::find_background_match, PCR::is_valid, optimize(); main.cpp:579-691,824,898, optimize.cpp:14); every returned BitSet
element, coverage float, is_valid flag, optimised Word and Score is compared.  Needs oracle/_ref/libadapter_check.so, which
`make -C oracle adapter` builds where /root/reference exists and which travels to the GPU box prebuilt; skipped cleanly where
it is absent."""
import ctypes
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libadapter_check.so")


@pytest.mark.parametrize("seed,fam,per,length,trials", [(7, 3, 8, 900, 12), (2024, 2, 10, 1500, 9)])
def test_adapter_runs_and_equals_the_reference(seed, fam, per, length, trials):
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libadapter_check.so not built (no /root/reference where this tree was built)")
    from pcramp_amd import api
    api.load_library()                              # torch's HIP runtime first, then the product library (one HIP runtime per process)
    h = ctypes.CDLL(SO)
    if not hasattr(h, "adapter_run"):
        pytest.skip("prebuilt libadapter_check.so predates adapter_run")
    h.adapter_run.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(ctypes.c_longlong)]
    stats = (ctypes.c_longlong * 8)()
    bad = h.adapter_run(seed, fam, per, length, trials, stats)
    st = list(stats)
    assert bad == 0, (bad, st)
    comparisons, amp_bits, bg_bits, changed, db_t, db_b, valid_n, valid_true = st
    assert comparisons > 2 * trials * fam * per        # every (trial, target) bit of both rounds, and more
    assert amp_bits >= trials                          # a trial assay amplifies the family it was cut from
    assert db_t > 50 and db_b > 50
    assert changed >= 1                                # the search repaired a damaged primer: optimised Words compared non-trivially
    assert valid_n == 2 * trials and 0 < valid_true
