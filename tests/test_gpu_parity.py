"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI, must reproduce the
reference's golden VCFs and the CPU oracle byte for byte."""
import pytest

import helpers
from golden_cases import CASES
from test_hostsim_golden import DEVICE_UNSUPPORTED

pytestmark = pytest.mark.gpu

SUPPORTED = [c for c in CASES if c[0] not in DEVICE_UNSUPPORTED]


@pytest.fixture(scope="module")
def gdb():
    import genomicsdb_amd
    from genomicsdb_amd import _lib
    assert _lib.lib().gdb_mi355_device_count() > 0, "no HIP device"
    return genomicsdb_amd


@pytest.mark.parametrize("case", SUPPORTED, ids=[c[0] for c in SUPPORTED])
def test_stream_matches_golden_and_oracle(gdb, case):
    name, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    want, _, _ = helpers.oracle_run(q, cells, partition_begin=pb)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    got = s.read()
    s.close()
    assert got == want
    assert got == helpers.golden_text(golden)


@pytest.mark.parametrize("case", [c for c in SUPPORTED if c[5] == "query"][:8], ids=[c[0] for c in SUPPORTED if c[5] == "query"][:8])
def test_stream_small_pages(gdb, case):
    """buffer_capacity 128: one record per page, like the reference's '-p 128' batched_vcf runs"""
    name, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=128)
    got = b""
    while True:
        chunk = s.read(100)
        if not chunk:
            break
        got += chunk
    s.close()
    assert got == helpers.golden_text(golden)


@pytest.mark.parametrize("name", sorted(DEVICE_UNSUPPORTED))
def test_unsupported_configurations_fail_loudly(gdb, name):
    case = [c for c in CASES if c[0] == name][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    with pytest.raises(gdb.GenomicsDBException):
        s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells)
        s.read()


def test_engine_stats_and_header(gdb):
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    e = gdb.CombineEngine(q)
    e.stage_cells(cells)
    body, st = e.run_interval(0, 1000000000, arena_bytes=1 << 20)
    assert e.header + body == helpers.golden_text(golden)
    assert st.num_records == 4 and st.num_cells == 5 and st.bytes_out == len(body)
    assert st.bytes_in_reference_cells == len(cells)
    e.close()
