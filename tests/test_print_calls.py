"""`gt_mpi_gather --print-calls` (SingleCellOperatorBase family, variant_operations.cc:803-843; iterator: genomicsdb_iterators.cc:181-510):
the oracle's restatement and the device path against the reference's own "calls" goldens (tests/golden/outputs/*calls*, copied data
fixtures of /root/reference/tests/golden_outputs; parameters from tests/run.py).  run.py accepts a JSON-equal document; the bar here is
the bytes."""
import ctypes, json, os
import pytest

import helpers
from golden_cases import CALLS_CASES


def calls_query(callsets, vid, ranges, attributes):
    return {
        "vid_mapping_file": os.path.join(helpers.GOLDEN, "inputs", vid),
        "callset_mapping_file": os.path.join(helpers.GOLDEN, "inputs", "callsets", callsets),
        "query_column_ranges": ranges,
        "query_row_ranges": [{"range_list": [{"low": 0, "high": 3}]}],
        "attributes": attributes,
        "segment_size": 40,
    }


def oracle_print_calls(q, cells):
    lib = helpers.oracle_lib()
    fn = lib.oracle_print_calls
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_uint64]
    out, n = ctypes.c_void_p(), ctypes.c_uint64()
    err = ctypes.create_string_buffer(1024)
    rc = fn(json.dumps(q).encode(), cells, len(cells), ctypes.byref(out), ctypes.byref(n), err, 1024)
    assert rc == 0, err.value.decode()
    text = ctypes.string_at(out.value, n.value)
    lib.oracle_free(out)
    return text


@pytest.mark.parametrize("case", CALLS_CASES, ids=[c[0] for c in CALLS_CASES])
def test_oracle_prints_the_reference_calls_goldens(case):
    name, callsets, vid, ranges, attributes = case
    cells = helpers.cells_for(callsets, vid)
    got = oracle_print_calls(calls_query(callsets, vid, ranges, attributes), cells)
    want = helpers.golden_text(name)
    assert json.loads(got) == json.loads(want)      # what the reference's own test accepts (run.py:993-1001)
    assert got == want                              # and the bytes
