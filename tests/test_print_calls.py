"""`gt_mpi_gather --print-calls` (SingleCellOperatorBase family, variant_operations.cc:803-843; iterator: genomicsdb_iterators.cc:181-510):
the oracle's restatement and the device path against the reference's own "calls" goldens (tests/golden/outputs/*calls*, copied data
fixtures of /root/reference/tests/golden_outputs; parameters from tests/run.py).  run.py accepts a JSON-equal document; the bar here is
the bytes."""
import ctypes, json, os
import pytest

import helpers
from golden_cases import CALLS_CASES


@pytest.fixture()
def gdb():
    import genomicsdb_amd
    return genomicsdb_amd


def calls_query(callsets, vid, ranges, attributes):
    return {
        "vid_mapping_file": os.path.join(helpers.GOLDEN, "inputs", vid),
        "callset_mapping_file": os.path.join(helpers.GOLDEN, "inputs", "callsets", callsets),
        "query_column_ranges": ranges,
        "query_row_ranges": [{"range_list": [{"low": 0, "high": 3}]}],
        "attributes": attributes,
        "segment_size": 40,
    }


def oracle_print_calls(q, cells, mode=0):
    lib = helpers.oracle_lib()
    fn = lib.oracle_print_cells
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_uint64]
    out, n = ctypes.c_void_p(), ctypes.c_uint64()
    err = ctypes.create_string_buffer(1024)
    rc = fn(json.dumps(q).encode(), cells, len(cells), mode, ctypes.byref(out), ctypes.byref(n), err, 1024)
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


def hostsim_print_calls(q, cells, mode=0):
    lib = helpers.hostsim_lib()
    fn = lib.hostsim_print_cells
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_uint64]
    out, n = ctypes.c_void_p(), ctypes.c_uint64()
    err = ctypes.create_string_buffer(4096)
    rc = fn(json.dumps(q).encode(), cells, len(cells), mode, ctypes.byref(out), ctypes.byref(n), err, 4096)
    assert rc == 0, err.value.decode()
    text = ctypes.string_at(out.value, n.value)
    lib.hostsim_free(out)
    return text


@pytest.mark.parametrize("case", CALLS_CASES, ids=[c[0] for c in CALLS_CASES])
def test_kernel_bodies_print_the_reference_calls_goldens(case):
    """core/gdb_calls.hpp (the functions the device kernel k_calls runs per cell) compiled by g++ (tests/hostsim)"""
    name, callsets, vid, ranges, attributes = case
    cells = helpers.cells_for(callsets, vid)
    assert hostsim_print_calls(calls_query(callsets, vid, ranges, attributes), cells) == helpers.golden_text(name)


def _synth_calls_query(tmp_path, N, ranges):
    q = dict(helpers.synth_query(tmp_path, N, 10_000_000, 10_000_000))
    q["query_column_ranges"] = ranges
    return q


def test_oracle_and_kernel_bodies_agree_on_synthetic_cells(tmp_path):
    """200 synthetic samples, intervals that begin inside reference blocks (every row has an interval intersecting the begin), a single
    position, an interval without cells"""
    from genomicsdb_amd import synth
    N, B, L = 200, 10_000_000, 3000
    cells, _ = synth.Generator(N, B, L).chunk_bytes(B + L)
    q = _synth_calls_query(tmp_path, N, [{"range_list": [{"low": B + 700, "high": B + 900}, {"low": B + 1500, "high": B + 1500}, {"low": B + 2000, "high": B + 2600},
                                                           {"low": B + 50_000, "high": B + 50_010}]}])
    a = oracle_print_calls(q, cells)
    b = hostsim_print_calls(q, cells)
    assert a == b and a.count(b'"row"') > 3 * N
    doc = json.loads(a)
    assert [iv["query_interval"] for iv in doc["variant_calls"]] == [[B + 700, B + 900], [B + 1500, B + 1500], [B + 2000, B + 2600]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CALLS_CASES, ids=[c[0] for c in CALLS_CASES])
def test_device_prints_the_reference_calls_goldens(gdb, case):
    name, callsets, vid, ranges, attributes = case
    cells = helpers.cells_for(callsets, vid)
    eng = gdb.CombineEngine(calls_query(callsets, vid, ranges, attributes))
    eng.stage_cells(cells)
    got = eng.print_calls()
    eng.close()
    assert got == helpers.golden_text(name)


@pytest.mark.gpu
def test_device_print_calls_at_1000_samples_and_through_column_windows(gdb, tmp_path, monkeypatch):
    """1 000 samples x 20 kb (190 000 cells, ~100 MB of JSON) against the oracle; the same array streamed through HBM in column windows
    (intervals cut into pieces: a piece behind the first prints only the cells that begin in it)"""
    from genomicsdb_amd import synth
    N, B, L = 1000, 10_000_000, 20_000
    cells, _ = synth.Generator(N, B, L).chunk_bytes(B + L)
    q = _synth_calls_query(tmp_path, N, [{"range_list": [{"low": B + 5000, "high": B + 15_000}, {"low": B + 17_000, "high": B + 17_000}]}])
    want = oracle_print_calls(q, cells)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    got = eng.print_calls()
    eng.close()
    assert got == want
    monkeypatch.setenv("GDBAMD_STAGE_BUDGET_BYTES", str(len(cells) // 7))
    eng = gdb.CombineEngine(q)
    eng.open_memory_cells(cells)
    got2 = eng.print_calls()
    eng.close()
    assert got2 == want


@pytest.mark.gpu
def test_gt_mpi_gather_print_calls(gdb, tmp_path):
    """the command line: gt_mpi_gather -j query.json --print-calls (tools/src/gt_mpi_gather.cc:369-383, 602-606)"""
    import subprocess
    name, callsets, vid, ranges, attributes = CALLS_CASES[8]       # t0_1_2_calls_at_12150
    ws = tmp_path / "ws"
    (ws / "arr").mkdir(parents=True)
    (ws / "arr" / "cells.bin").write_bytes(helpers.cells_for(callsets, vid))
    q = calls_query(callsets, vid, ranges, attributes)
    q["workspace"], q["array"] = str(ws), "arr"
    qf = tmp_path / "q.json"
    qf.write_text(json.dumps(q))
    exe = os.path.join(helpers.ROOT, "genomicsdb_amd", "gt_mpi_gather")
    r = subprocess.run([exe, "-j", str(qf), "--print-calls"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == helpers.golden_text(name)


# ---- --print-csv / --print-AC: the reference's tests hold no golden for them (parity unpinned: oracle restatement = kernel bodies = device) ----------
def test_allele_counts_known_answer_on_t0_1_2():
    """hand-derived from the reference's golden t0_1_2_calls_at_0 (rows 0 / 1 / 2 at 17384: G -> A 0/1, G -> T 0/1 ... ): every GT element that names
    an ALT allele counts once under (column, REF, ALT)"""
    doc = json.loads(helpers.golden_text("t0_1_2_calls_at_0"))
    want = {}
    for c in doc["variant_calls"][0]["variant_calls"]:
        f = c["fields"]
        if "GT" not in f or "REF" not in f or "ALT" not in f:
            continue
        for g in f["GT"]:
            if g > 0:
                ref, alt = f["REF"], f["ALT"][g - 1]
                alt = "&" if alt == "<NON_REF>" else alt
                assert len(ref) == 1                # (no deletions among the called alleles of this input: nothing to normalise)
                k = (c["interval"][0], ref, alt)
                want[k] = want.get(k, 0) + 1
    lines = "".join("%d %s %s %d\n" % (k[0], k[1], k[2], n) for k, n in sorted(want.items()))
    case = [c for c in CALLS_CASES if c[0] == "t0_1_2_calls_at_0"][0]
    name, callsets, vid, ranges, attributes = case
    cells = helpers.cells_for(callsets, vid)
    q = calls_query(callsets, vid, ranges, attributes)
    assert oracle_print_calls(q, cells, 2).decode() == lines and len(want) >= 2
    assert hostsim_print_calls(q, cells, 2).decode() == lines


@pytest.mark.parametrize("case", CALLS_CASES, ids=[c[0] for c in CALLS_CASES])
@pytest.mark.parametrize("mode", [1, 2])
def test_csv_and_allele_counts_kernel_bodies_against_the_oracle(case, mode):
    name, callsets, vid, ranges, attributes = case
    if mode == 2 and attributes == ["MLEAC"]:
        pytest.skip("AlleleCountOperator needs GT in the query")
    cells = helpers.cells_for(callsets, vid)
    q = calls_query(callsets, vid, ranges, attributes)
    a = oracle_print_calls(q, cells, mode)
    assert hostsim_print_calls(q, cells, mode) == a
    if mode == 1 and b'"row"' in helpers.golden_text(name):
        assert a.count(b"\n") == helpers.golden_text(name).count(b'"row"')      # one line per cell of the JSON golden


def test_csv_and_allele_counts_on_synthetic_cells_with_deletions(tmp_path):
    from genomicsdb_amd import synth
    N, B, L = 300, 10_000_000, 4000
    cells, _ = synth.Generator(N, B, L).chunk_bytes(B + L)
    q = _synth_calls_query(tmp_path, N, [{"range_list": [{"low": B + 500, "high": B + 2500}, {"low": B + 3000, "high": B + 3000}]}])
    for mode in (1, 2):
        a = oracle_print_calls(q, cells, mode)
        assert hostsim_print_calls(q, cells, mode) == a and len(a) > 1000
    ac = oracle_print_calls(q, cells, 2).decode().splitlines()
    assert any(len(l.split()[1]) > 1 for l in ac)        # deletions are there (REF longer than one base after normalisation)


@pytest.mark.gpu
def test_device_prints_array_rows_for_a_row_subset(gdb):
    name, callsets, vid, ranges, attributes = CALLS_CASES[8]
    cells = helpers.cells_for(callsets, vid)
    for row in (1, 2):
        q = calls_query(callsets, vid, ranges, attributes)
        q["query_row_ranges"] = [{"range_list": [{"low": row, "high": row}]}]
        eng = gdb.CombineEngine(q)
        eng.stage_cells(cells)
        for mode in (0, 1, 2):
            assert eng.print_calls(mode) == oracle_print_calls(q, cells, mode), (row, mode)
        eng.close()


@pytest.mark.gpu
def test_device_csv_and_allele_counts(gdb, tmp_path):
    from genomicsdb_amd import synth
    for case in (CALLS_CASES[0], CALLS_CASES[6], CALLS_CASES[11], CALLS_CASES[16]):
        name, callsets, vid, ranges, attributes = case
        cells = helpers.cells_for(callsets, vid)
        q = calls_query(callsets, vid, ranges, attributes)
        eng = gdb.CombineEngine(q)
        eng.stage_cells(cells)
        for mode in (1, 2):
            assert eng.print_calls(mode) == oracle_print_calls(q, cells, mode), (name, mode)
        eng.close()
    N, B, L = 1000, 10_000_000, 20_000
    cells, _ = synth.Generator(N, B, L).chunk_bytes(B + L)
    q = _synth_calls_query(tmp_path, N, [{"range_list": [{"low": B + 5000, "high": B + 15_000}, {"low": B + 17_000, "high": B + 17_000}]}])
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    for mode in (1, 2):
        assert eng.print_calls(mode) == oracle_print_calls(q, cells, mode)
    eng.close()


def test_row_subsets_and_empty_results():
    """a query for one row prints that row's cells only (the search for intersecting intervals ends when every QUERIED row has been seen); an
    interval in front of every cell and a query over no cells at all print the empty document"""
    name, callsets, vid, ranges, attributes = CALLS_CASES[8]          # t0_1_2_calls_at_12150
    cells = helpers.cells_for(callsets, vid)
    full = json.loads(helpers.golden_text(name))
    for row in (0, 1, 2):
        q = calls_query(callsets, vid, ranges, attributes)
        q["query_row_ranges"] = [{"range_list": [{"low": row, "high": row}]}]
        a = oracle_print_calls(q, cells)
        assert hostsim_print_calls(q, cells) == a
        want = [c for c in full["variant_calls"][0]["variant_calls"] if c["row"] == row]
        assert json.loads(a)["variant_calls"][0]["variant_calls"] == want and want
    q = calls_query(callsets, vid, [{"range_list": [{"low": 5, "high": 100}]}], attributes)
    empty = b'{\n    "variant_calls": [\n\n    ]\n}\n'
    assert oracle_print_calls(q, cells) == empty == hostsim_print_calls(q, cells) == helpers.golden_text("t0_1_2_calls_at_12100")
    for mode in (1, 2):
        assert oracle_print_calls(q, cells, mode) == b"" == hostsim_print_calls(q, cells, mode)
