"""The six GTProfileStats counters of the reference (query_variants.h:67-124) come back per interval in gdbamd_interval_stats and,
summed, from VariantQueryProcessor::get_profile_stats().  Known answers are taken by hand from the reference's test inputs:
t0 / t1 / t2 (three samples; t0: 12141-12295 + 17385, t1: 12145-12277 + 17385, t2: 17385 only)."""
import ctypes

import pytest

import helpers
from golden_cases import CASES


def test_struct_mirror_has_the_counters_and_their_names():
    from genomicsdb_amd import _lib
    st = _lib.IntervalStats()
    assert len(st.gt_profile_stats) == 6
    assert list(st.gt_profile()) == ["GT_NUM_CELLS", "GT_NUM_CELLS_IN_LEFT_SWEEP", "GT_NUM_VALID_CELLS_IN_QUERY", "GT_NUM_ATTR_CELLS_ACCESSED",
                                     "GT_NUM_PQ_FLUSHES_DUE_TO_OVERLAPPING_CELLS", "GT_NUM_OPERATOR_INVOCATIONS"]
    # the C header declares the same number of counters in the same place (end of the struct)
    hdr = open(helpers.os.path.join(helpers.ROOT, "include", "genomicsdb_amd.h")).read()
    assert "uint64_t gt_profile_stats[6];\n} gdbamd_interval_stats;" in hdr
    assert ctypes.sizeof(_lib.IntervalStats) % 8 == 0


def _case(name):
    c = [c for c in CASES if c[0] == name][0]
    return helpers.cells_for(c[1], c[2]), helpers.query_json(c[1], c[2], c[3], c[5])[0]


@pytest.mark.gpu
def test_counters_on_the_three_sample_input():
    import genomicsdb_amd
    cells, q = _case("t0_1_2_vcf_at_0")
    e = genomicsdb_amd.CombineEngine(q)
    e.stage_cells(cells)
    nf = len(e.fields())
    body, st = e.run_interval(0, 1_000_000_000, arena_bytes=1 << 20)
    g = st.gt_profile()
    assert st.num_records == 4                       # 12141-12144, 12145-12277, 12278-12295, 17385
    assert g["GT_NUM_CELLS"] == 5 and g["GT_NUM_VALID_CELLS_IN_QUERY"] == 5     # 2 + 2 + 1 begin-cells
    assert g["GT_NUM_CELLS_IN_LEFT_SWEEP"] == 0 and g["GT_NUM_PQ_FLUSHES_DUE_TO_OVERLAPPING_CELLS"] == 0
    assert g["GT_NUM_ATTR_CELLS_ACCESSED"] == 5 * nf
    assert g["GT_NUM_OPERATOR_INVOCATIONS"] == 4
    # from 12150: both reference blocks began before the interval (left sweep), the three variant cells begin inside it
    body, st = e.run_interval(12150, 1_000_000_000, arena_bytes=1 << 20)
    g = st.gt_profile()
    assert st.num_records == 3
    assert g["GT_NUM_CELLS_IN_LEFT_SWEEP"] == 2 and g["GT_NUM_VALID_CELLS_IN_QUERY"] == 5 and g["GT_NUM_OPERATOR_INVOCATIONS"] == 3
    # a window that ends before 17385: only the two blocks contribute
    body, st = e.run_interval(12150, 12300, arena_bytes=1 << 20)
    g = st.gt_profile()
    assert g["GT_NUM_VALID_CELLS_IN_QUERY"] == 2 and g["GT_NUM_CELLS_IN_LEFT_SWEEP"] == 2 and g["GT_NUM_OPERATOR_INVOCATIONS"] == 2
    e.close()


@pytest.mark.gpu
def test_overlapping_cells_of_one_sample_are_counted():
    """t0_overlapping: the block 12141-12277 is cut by the deletion at 12200, 12201-12202 by nothing, 12203-12280 by the deletion at
    12207 and its rest by the block at 12210 ... : every cell whose END reaches the next cell's begin counts once"""
    import genomicsdb_amd
    cells, q = _case("t0_overlapping_loading")
    e = genomicsdb_amd.CombineEngine(q)
    e.stage_cells(cells)
    body, st = e.run_interval(0, 1_000_000_000, arena_bytes=1 << 20)
    g = st.gt_profile()
    assert g["GT_NUM_PQ_FLUSHES_DUE_TO_OVERLAPPING_CELLS"] >= 3
    assert g["GT_NUM_VALID_CELLS_IN_QUERY"] <= g["GT_NUM_CELLS"] and g["GT_NUM_OPERATOR_INVOCATIONS"] == st.num_records
    e.close()
