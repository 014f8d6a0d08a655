"""The GDB_HD kernel bodies (same source the HIP kernels compile), run serially on the CPU, against the
reference goldens.  Cases the device path does not support yet must fail loudly (UnsupportedOnDevice)."""
import pytest

import helpers
from golden_cases import CASES

# golden cases whose configuration the device path rejects today, with the reason it reports
DEVICE_UNSUPPORTED = {}   # every C++-path golden of the reference that the oracle covers now runs on the device path


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hostsim_matches_reference_golden(case):
    name, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    if name in DEVICE_UNSUPPORTED:
        # either the plan builder rejects the configuration or a kernel raises an error bit (-> exception in the product)
        try:
            _, errbits = helpers.hostsim_run(q, cells)
        except RuntimeError as e:
            assert "UnsupportedOnDevice" in str(e)
            return
        assert errbits != 0
        return
    txt, errbits = helpers.hostsim_run(q, cells)
    assert errbits == 0
    assert txt == helpers.golden_text(golden)


def test_unqueried_rows_still_split_intervals():
    """The reference's scan iterates ALL array rows and closes the current interval at every cell begin before it checks whether
    the row is queried (query_variants.cc:478-507).  A query for samples 0 and 2 therefore gets the interval splits caused by
    sample 1's cells: staging keeps those cells as position-only boundary markers."""
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    q["query_row_ranges"] = [{"range_list": [{"low": 0, "high": 0}, {"low": 2, "high": 2}]}]
    want, nrec, _ = helpers.oracle_run(q, cells, partition_begin=pb)
    got, errbits = helpers.hostsim_run(q, cells)
    assert errbits == 0 and got == want
    assert b"END=12144" in want    # the split at 12145 comes from the unqueried sample's cell


def test_median_selection_is_the_c_library_selection():
    """The reference takes medians with std::nth_element and prints the selected float: with -0 and +0 tied at the middle the
    printed sign depends on the permutation libstdc++'s introselect leaves.  The restatement the kernels use must leave the
    same permutation as the library for random, tie-heavy, sorted and adversarial (depth-limit -> heap select) inputs."""
    import ctypes
    import numpy as np
    lib = helpers.hostsim_lib()
    lib.hostsim_nth_element_same.restype = ctypes.c_int
    lib.hostsim_nth_element_same.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    rng = np.random.default_rng(7)

    def check(a, nth=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        nth = len(a) // 2 if nth is None else nth
        assert lib.hostsim_nth_element_same(a.ctypes.data, len(a), nth, -1) == 1, (len(a), nth)
        for depth in (0, 1, 3):        # the library's loop with a tiny depth budget: its heap-select branch
            assert lib.hostsim_nth_element_same(a.ctypes.data, len(a), nth, depth) == 1, (len(a), nth, depth)

    for n in list(range(1, 40)) + [63, 64, 65, 100, 257, 1000, 4097, 20000]:
        for rep in range(6):
            check(rng.standard_normal(n))
            z = rng.choice(np.array([-0.0, 0.0, -0.5, 0.25, 1.0], dtype=np.float32), size=n)          # many tied zeros of both signs
            check(z)
            check(np.round(rng.standard_normal(n), 1) + np.float32(0.0) * rng.choice([-1.0, 1.0], size=n))
            check(rng.standard_normal(n), nth=int(rng.integers(0, n)))
        check(np.arange(n)); check(np.arange(n)[::-1]); check(np.zeros(n)); check(np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]))
