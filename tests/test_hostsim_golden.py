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
