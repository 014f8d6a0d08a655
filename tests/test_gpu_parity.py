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
def test_stream_small_pages(gdb, case, monkeypatch):
    """128-byte device pages: one record per page, like the reference's '-p 128' batched_vcf runs (the device page size is
    independent of buffer_capacity: GDBAMD_DEVICE_PAGE_BYTES sets it)"""
    monkeypatch.setenv("GDBAMD_DEVICE_PAGE_BYTES", "128")
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


def test_unsupported_configurations_fail_loudly(gdb, tmp_path):
    """an output format htslib does not know: an error, never a silent fallback ("z" / "b" are written: tests/test_bgzf.py)"""
    case = CASES[0]
    qj, _ = helpers.query_json(case[1], case[2], case[3], case[5])
    qj["vcf_output_format"] = "zz"
    import subprocess, os, json as _json
    qf = tmp_path / "q.json"
    (tmp_path / "ws" / "a").mkdir(parents=True)
    (tmp_path / "ws" / "a" / "cells.bin").write_bytes(helpers.cells_for(case[1], case[2]))
    qj["workspace"], qj["array"] = str(tmp_path / "ws"), "a"
    qf.write_text(_json.dumps(qj))
    tool = os.path.join(helpers.ROOT, "genomicsdb_amd", "gt_mpi_gather")
    r = subprocess.run([tool, "-j", str(qf), "--produce-Broad-GVCF"], capture_output=True, timeout=120)
    assert r.returncode != 0 and b"output format" in r.stderr


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


def test_synthetic_matches_oracle(gdb, tmp_path, monkeypatch):
    """generator of SURVEY 8(d): 300 samples x 6 kb (SNVs, insertions, deletions with per-bp stepping), staged in two parts"""
    from genomicsdb_amd import synth
    N, B, L = 300, 10_000_000, 6000
    g = synth.Generator(N, B, L)
    c1, _ = g.chunk_bytes(B + 2500)
    c2, _ = g.chunk_bytes(B + L)
    q = helpers.synth_query(tmp_path, N, B + 700, B + L - 900)
    want, nrec, _ = helpers.oracle_run_synth(q, c1 + c2, synth.SEED, with_header=False)
    e = gdb.CombineEngine(q)
    e.stage_cells_begin()
    import ctypes
    for part in (c1, c2):
        buf = ctypes.create_string_buffer(part, len(part))
        e.stage_cells_append(ctypes.addressof(buf), len(part))
    e.stage_cells_end()
    e.set_reference(B, synth.reference(B, L + 16))
    body, st = e.run_interval(B + 700, B + L - 900, arena_bytes=1 << 20)   # many pages
    assert st.num_records == nrec and st.pages > 5
    assert body == want
    body2, st2 = e.run_interval(B + 700, B + L - 900, arena_bytes=1 << 30)  # one page
    assert st2.pages == 1 and body2 == want
    monkeypatch.setenv("GDBAMD_RESOLVED_MB", "0")   # (record, sample) matrix over budget: resolved page by page
    body3, st3 = e.run_interval(B + 700, B + L - 900, arena_bytes=1 << 20)
    assert st3.pages > 5 and body3 == want
    e.close()


def test_synthetic_window_split_equals_whole(gdb, tmp_path):
    """size-independent property: the records of two adjacent windows = the records of their union, except that an
    interval crossing the cut is split there (END of the first part = cut, same rule as the reference's partitions)"""
    from genomicsdb_amd import synth
    N, B, L = 500, 10_000_000, 30_000
    g = synth.Generator(N, B, L)
    cells, _ = g.chunk_bytes(B + L)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    e = gdb.CombineEngine(q)
    e.stage_cells(cells)
    e.set_reference(B, synth.reference(B, L + 16))
    whole, sw = e.run_interval(B + 1000, B + 20_999, arena_bytes=1 << 28)
    a, sa = e.run_interval(B + 1000, B + 10_999, arena_bytes=1 << 28)
    b, sb = e.run_interval(B + 11_000, B + 20_999, arena_bytes=1 << 28)
    assert sa.num_records + sb.num_records in (sw.num_records, sw.num_records + 1)
    pos_whole = [l.split(b"\t", 2)[1] for l in whole.splitlines()]
    pos_split = [l.split(b"\t", 2)[1] for l in (a + b).splitlines()]
    assert set(pos_whole) <= set(pos_split) and len(pos_split) - len(pos_whole) <= 1
    assert pos_split == sorted(pos_split, key=int)
    e.close()


@pytest.mark.parametrize("max_types", ["0", "1", "2"])
def test_untabled_record_types_take_the_direct_path(gdb, tmp_path, monkeypatch, max_types):
    """Record types beyond the text-table capacity (64) are emitted per (record, sample) by the field emitters instead of
    copied from the pool; GDBAMD_MAX_TYPES shrinks the capacity so that both paths mix in one page."""
    monkeypatch.setenv("GDBAMD_MAX_TYPES", max_types)
    for name in ("t0_1_2_vcf_at_0", "t6_7_8_vcf_at_0", "t0_overlapping_at_12202"):
        case = [c for c in CASES if c[0] == name]
        if not case:
            continue
        _, callsets, vid, ov, golden, mode = case[0]
        cells = helpers.cells_for(callsets, vid)
        q, pb = helpers.query_json(callsets, vid, ov, mode)
        s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 16)
        got = s.read()
        s.close()
        assert got == helpers.golden_text(golden)
    from genomicsdb_amd import synth
    N, B, L = 100, 10_000_000, 3000
    g = synth.Generator(N, B, L + 2500)
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B + 500, B + 500 + L - 1)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 4096))
    got, st = eng.run_interval(B + 500, B + 500 + L - 1, arena_bytes=1 << 20)
    assert st.num_record_types == int(max_types)
    assert got == want
    eng.close()


def _c2_engine(gdb, tmp_path, N, B, L, slack=2500):
    from genomicsdb_amd import synth
    g = synth.Generator(N, B, L + slack)
    cells, nc = g.chunk_bytes(B + L + slack)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + slack + 4096))
    return eng, q, cells


def test_c2_width_1000_samples_matches_oracle(gdb, tmp_path):
    """BASELINE.json configs[1] at its full sample count (1 000) on a column window the oracle finishes in seconds"""
    from genomicsdb_amd import synth
    N, B, L = 1000, 10_000_000, 1500
    eng, q, cells = _c2_engine(gdb, tmp_path, N, B, L)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    got, st = eng.run_interval(B, B + L - 1, arena_bytes=8 << 20)
    assert st.num_records == nrec and st.pages > 3
    assert got == want
    eng.close()


def test_c2_scale_properties(gdb, tmp_path, monkeypatch):
    """1 000 samples x 60 kb (2.7 GB of VCF text): properties that do not need the oracle at this size -
    (1) the bytes do not depend on paging, on the resolved-matrix mode, on the record visiting order or on the type-table
        capacity;  (2) every record line has 9 + N tab-separated columns, positions ascend, END >= POS;
    (3) the first 1 200 columns equal the oracle's output for that window (records are independent of what follows them)."""
    import hashlib
    from genomicsdb_amd import synth
    N, B, L = 1000, 10_000_000, 60_000
    eng, q, cells = _c2_engine(gdb, tmp_path, N, B, L)
    ref, st = eng.run_interval(B, B + L - 1, arena_bytes=4 << 30)
    assert st.pages == 1 and st.num_records > 50_000
    h_ref = hashlib.sha256(ref).hexdigest()
    for env, arena in (({"GDBAMD_RESOLVED_MB": "0"}, 256 << 20), ({"GDBAMD_ORDER_BLOCK_LOG2": "30"}, 1 << 30),
                       ({"GDBAMD_ORDER_BLOCK_LOG2": "0", "GDBAMD_MAX_TYPES": "3"}, 700 << 20)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got, st2 = eng.run_interval(B, B + L - 1, arena_bytes=arena)
        for k in env:
            monkeypatch.delenv(k)
        assert st2.num_records == st.num_records and len(got) == len(ref)
        assert hashlib.sha256(got).hexdigest() == h_ref, env
    lines = ref.split(b"\n")
    assert lines[-1] == b"" and len(lines) - 1 == st.num_records
    prev = 0
    for ln in lines[:-1:97]:                      # every 97th record: column count, order, END
        cols = ln.split(b"\t")
        assert len(cols) == 9 + N
        pos = int(cols[1])
        assert pos > prev
        prev = pos
        info = cols[7]
        if info.startswith(b"END="):
            assert int(info[4:].split(b";")[0]) >= pos
    # window prefix against the oracle
    n_small = 1200
    g = synth.Generator(N, B, n_small + 2500)
    small_cells, _ = g.chunk_bytes(B + n_small + 2500)
    q_small = helpers.synth_query(tmp_path, N, B, B + n_small - 1)
    want, nrec, _ = helpers.oracle_run_synth(q_small, small_cells, synth.SEED, with_header=False)
    want_lines = want.split(b"\n")[:-1]
    # the last oracle record may be clipped by the smaller query window: compare all but the last
    assert lines[:len(want_lines) - 1] == want_lines[:-1]
    eng.close()


@pytest.mark.parametrize("max_alt", [50, 64])
def test_high_alt_dense_region_matches_oracle(gdb, tmp_path, max_alt):
    """BASELINE.json configs[4]-style stress at test size (see tests/test_synth_parity_cpu.py): ~50 merged alleles per hot
    site, entries of several KB each - the text-pool strips, the LDS images and the pages all take their oversize paths"""
    from genomicsdb_amd import synth
    N, B, L = 150, 10_000_000, 330
    g = synth.Generator(N, B, L + 500, dense=(B + 100, 200, 50, 64))
    cells, nc = g.chunk_bytes(B + L + 500)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    q["max_diploid_alt_alleles_that_can_be_genotyped"] = max_alt
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 4096))
    got, st = eng.run_interval(B, B + L - 1, arena_bytes=1 << 20)
    assert st.num_records == nrec
    assert got == want
    widest = max(len(l.split(b"\t")[4].split(b",")) for l in want.split(b"\n") if l)
    assert widest >= 45
    eng.close()


def test_gt_mpi_gather_cli_produces_the_golden(gdb, tmp_path):
    """the reference's command line (tools/src/gt_mpi_gather.cc:437-531, mode --produce-Broad-GVCF) on an array directory:
    stdout must be the reference's golden for the same query, for two paging sizes"""
    import json
    import os
    import subprocess
    tool = os.path.join(helpers.ROOT, "genomicsdb_amd", "gt_mpi_gather")
    assert os.path.exists(tool), "build() must produce the tool"
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, _ = helpers.query_json(callsets, vid, ov, mode)
    ws = tmp_path / "ws"
    (ws / "t0_1_2").mkdir(parents=True)
    (ws / "t0_1_2" / "cells.bin").write_bytes(cells)
    q["workspace"] = str(ws)
    q["array"] = "t0_1_2"
    qf = tmp_path / "query.json"
    qf.write_text(json.dumps(q))
    for page in ("0", "128"):
        r = subprocess.run([tool, "-j", str(qf), "-p", page, "--produce-Broad-GVCF"], capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()
        assert r.stdout == helpers.golden_text(golden)
        assert b"scan_and_produce_Broad_GVCF" in r.stderr
    r = subprocess.run([tool, "-j", str(qf), "--print-calls"], capture_output=True, timeout=60)      # (on the device since round 4: tests/test_print_calls.py pins its bytes)
    assert r.returncode == 0 and r.stdout.startswith(b'{\n    "variant_calls": [')
    r = subprocess.run([tool, "-j", str(qf), "--produce-interesting-positions"], capture_output=True, timeout=60)   # outside the path: refused, loudly
    assert r.returncode != 0 and r.stderr


def test_row_subset_query_drops_cells_at_staging(gdb):
    """query_row_ranges selecting samples 0 and 2 only: the device-side cell-stream parser drops the cells of sample 1 and
    renumbers the rest; the result must be what the oracle gives for the same query"""
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    q["query_row_ranges"] = [{"range_list": [{"low": 0, "high": 0}, {"low": 2, "high": 2}]}]
    want, nrec, _ = helpers.oracle_run(q, cells, partition_begin=pb)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    got = s.read()
    s.close()
    assert got == want
    assert got != helpers.golden_text(golden) and b"HG00141" in want and b"HG01958" not in want.split(b"#CHROM")[1].split(b"\n")[0]


def test_sorted_median_path_gives_the_same_bytes(gdb, tmp_path, monkeypatch):
    """intervals that average more than 16 variant calls per record order their median fields with a device-wide sort instead
    of the per-record quadratic scan; GDBAMD_SORTED_MEDIAN forces that path on small inputs"""
    monkeypatch.setenv("GDBAMD_SORTED_MEDIAN", "1")
    for name in ("info_ops0", "t0_1_2_vcf_at_0", "min_PL_spanning_deletion_vcf", "t6_7_8_vcf_at_0"):
        _, callsets, vid, ov, golden, mode = [c for c in CASES if c[0] == name][0]
        cells = helpers.cells_for(callsets, vid)
        q, pb = helpers.query_json(callsets, vid, ov, mode)
        s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 16)
        got = s.read()
        s.close()
        assert got == helpers.golden_text(golden), name
    from genomicsdb_amd import synth
    N, B, L = 400, 10_000_000, 2500
    eng, q, cells = _c2_engine(gdb, tmp_path, N, B, L)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    got, st = eng.run_interval(B, B + L - 1, arena_bytes=4 << 20)
    assert st.num_records == nrec and got == want
    eng.close()


@pytest.mark.parametrize("n_samples,seed", [(1, 11), (2, 12), (63, 13), (64, 14), (65, 15), (127, 16), (128, 17), (129, 18), (257, 19), (640, 20)])
def test_sample_counts_around_the_wavefront_width(gdb, tmp_path, n_samples, seed):
    """the sample axis is cut into chunks of 64 (one wavefront each): counts just below / at / above the multiples, other
    generator seeds, and a query window that starts and ends inside cells"""
    from genomicsdb_amd import synth
    B, L = 10_000_000, 1800 if n_samples < 300 else 900
    g = synth.Generator(n_samples, B, L + 2500, seed=seed)
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, n_samples, B + 333, B + 333 + L - 1)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, seed, with_header=False)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 4096, seed=seed))
    got, st = eng.run_interval(B + 333, B + 333 + L - 1, arena_bytes=256 << 10)
    assert st.num_records == nrec
    assert got == want
    eng.close()


@pytest.mark.parametrize("options", [
    {"sites_only_query": True},
    {"produce_GT_field": True},
    {"produce_GT_field": True, "produce_GT_with_min_PL_value_for_spanning_deletions": True},
    {"produce_FILTER_field": True, "max_diploid_alt_alleles_that_can_be_genotyped": 2},
    {"combined_vcf_records_buffer_size_limit": 4096, "produce_GT_field": True},
], ids=["sites_only", "GT", "GT_minPL", "FILTER_maxalt2", "small_buffer_GT"])
def test_query_options_on_synthetic_input(gdb, tmp_path, options):
    """the query-JSON switches of BroadCombinedGVCFOperator on an input with deletions, insertions and spanning calls"""
    from genomicsdb_amd import synth
    N, B, L = 120, 10_000_000, 3000
    g = synth.Generator(N, B, L + 2500, seed=77)
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B + 100, B + 100 + L - 1)
    q.update(options)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, 77, with_header=False)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 4096, seed=77))
    got, st = eng.run_interval(B + 100, B + 100 + L - 1, arena_bytes=1 << 20)
    assert st.num_records == nrec and got == want
    eng.close()


def test_columnar_fragment_file_round_trip(gdb, tmp_path):
    """stage from the cell stream, save the fragment as it lies in HBM, open it again (file -> HBM copies, no parsing) in a
    new engine and through the query stream / command line: identical bytes"""
    import json
    import os
    import subprocess
    case = [c for c in CASES if c[0] == "t6_7_8_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, _ = helpers.query_json(callsets, vid, ov, mode)
    ws = tmp_path / "ws"
    (ws / "arr").mkdir(parents=True)
    e1 = gdb.CombineEngine(q)
    e1.stage_cells(cells)
    body1, st1 = e1.run_interval(0, 10**9, arena_bytes=1 << 20)
    e1.save_fragment(ws / "arr" / "fragment.gdbamd")
    e1.close()
    e2 = gdb.CombineEngine(q)
    e2.load_fragment(ws / "arr" / "fragment.gdbamd")
    assert e2.staged_info()[0] == st1.num_cells
    body2, st2 = e2.run_interval(0, 10**9, arena_bytes=1 << 20)
    e2.close()
    assert body2 == body1 and st2.num_records == st1.num_records
    # the query stream / gt_mpi_gather open fragment.gdbamd when it is there (no cells.bin in this directory)
    q2 = dict(q)
    q2["workspace"] = str(ws)
    q2["array"] = "arr"
    qf = tmp_path / "query.json"
    qf.write_text(json.dumps(q2))
    tool = os.path.join(helpers.ROOT, "genomicsdb_amd", "gt_mpi_gather")
    r = subprocess.run([tool, "-j", str(qf), "--produce-Broad-GVCF"], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == helpers.golden_text(golden)


def test_differential_fuzz_against_the_oracle(gdb):
    """40 random synthetic configurations (sample counts, window offsets, dense high-ALT regions, overlapping intervals of one
    sample, FILTER ids, ID tokens, query switches, page sizes, staging in parts) through tests/tools/fuzz.py"""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "tests", "tools", "fuzz.py"), "40", "4242"], capture_output=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr).decode()[-2000:]
    assert b"40 cases, 0 mismatches" in r.stdout


@pytest.mark.parametrize("max_columns", ["1", "40", "5000"])
def test_wide_intervals_are_worked_off_in_pieces(gdb, monkeypatch, max_columns):
    """the query stream cuts a wide interval right before cell begins (where the sweep closes its interval anyway):
    GDBAMD_MAX_WINDOW_COLUMNS forces many pieces on the golden inputs - same bytes"""
    monkeypatch.setenv("GDBAMD_MAX_WINDOW_COLUMNS", max_columns)
    for name in ("t0_1_2_vcf_at_0", "t0_overlapping_loading", "t6_7_8_vcf_at_0", "min_PL_spanning_deletion_vcf", "t0_1_2_vcf_at_multiple_positions",
                 "t0_haploid_triploid_1_2_3_triploid_deletion_vcf"):
        _, callsets, vid, ov, golden, mode = [c for c in CASES if c[0] == name][0]
        cells = helpers.cells_for(callsets, vid)
        q, pb = helpers.query_json(callsets, vid, ov, mode)
        s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 16)
        got = s.read()
        s.close()
        assert got == helpers.golden_text(golden), name


def test_c3_width_10000_samples_matches_oracle(gdb, tmp_path, monkeypatch):
    """BASELINE.json configs[2] at its full sample count (10 000 rows: 157 wavefront-wide sample chunks per record, records of
    ~0.9 MB) on a window the oracle finishes in seconds; pages smaller than some records.  With this many calls per record
    the rank-sum medians land on zeros of both signs: which one is printed ("-0" / "0") follows the reference's
    std::nth_element, through the workgroup medians and through the sorted medians alike."""
    from genomicsdb_amd import synth
    N, B, L = 10_000, 10_000_000, 40
    eng, q, cells = _c2_engine(gdb, tmp_path, N, B, L)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    assert b"=-0;" in want and b"=0;" in want
    got, st = eng.run_interval(B, B + L - 1, arena_bytes=4 << 20)
    assert st.num_records == nrec and st.pages > 5
    assert got == want
    monkeypatch.setenv("GDBAMD_SORTED_MEDIAN", "1")
    got, st = eng.run_interval(B, B + L - 1, arena_bytes=64 << 20)
    assert got == want
    eng.close()


def test_c5_width_12000_samples_dense_site_matches_oracle(gdb, tmp_path, monkeypatch):
    """BASELINE.json configs[4]-style site at 12 000 rows: every sample starts an insertion from a pool of 64 alleles at one
    column - the allele merge of that record sees 12 000 variant calls, each PL vector is re-indexed over ~2 000 genotypes"""
    from genomicsdb_amd import synth
    N, B, L = 12_000, 10_000_000, 56
    g = synth.Generator(N, B, L + 2500, dense=(B + 40, 20, 50, 64))      # the hot column is B + 50
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B + 30, B + L - 1)
    q["max_diploid_alt_alleles_that_can_be_genotyped"] = 64
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 4096))
    got, st = eng.run_interval(B + 30, B + L - 1, arena_bytes=64 << 20)
    assert st.num_records == nrec
    assert got == want
    widest = max(len(l.split(b"\t")[4].split(b",")) for l in want.split(b"\n") if l)
    assert widest >= 60
    monkeypatch.setenv("GDBAMD_NO_HUGE_SITES", "1")     # the one-thread walk of the calls gives the same bytes
    got2, _ = eng.run_interval(B + 30, B + L - 1, arena_bytes=64 << 20)
    assert got2 == want
    eng.close()


def test_c5_50000_samples_hot_site_matches_oracle(gdb, tmp_path, monkeypatch):
    """BASELINE.json configs[4] at its stated sample count: 50 000 samples all start an insertion from a pool of 64 alleles at
    one column.  The record's call walk (allele merge, LUTs, reducers, medians by radix select) is done by one workgroup
    (k_site_huge); the bytes must be the oracle's, and the same with the serial walk forced (GDBAMD_NO_HUGE_SITES)"""
    from genomicsdb_amd import synth
    N, B = 50_000, 10_000_000
    g = synth.Generator(N, B, 2600, dense=(B + 40, 20, 50, 64))      # the hot column is B + 50
    cells, nc = g.chunk_bytes(B + 2600)
    g.close()
    q = helpers.synth_query(tmp_path, N, B + 50, B + 50)
    q["max_diploid_alt_alleles_that_can_be_genotyped"] = 70      # 64 pool alleles + '*' + <NON_REF>: PL stays (2 278 genotypes per call)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, 4096))
    got, st = eng.run_interval(B + 50, B + 50, arena_bytes=2 << 30)
    assert st.num_records == nrec == 1
    assert got == want
    assert len(want.split(b"\t")[4].split(b",")) >= 52 and st.num_remap_elements >= 50_000 * 52 * 53 // 2
    eng.close()


def test_c5_twenty_hot_sites_of_50000_calls_with_tied_zero_medians(gdb, tmp_path, monkeypatch):
    """BASELINE.json configs[4]'s sample count, 20 hot sites (every sample starts an insertion every 50 columns), rank sums rounded to
    halves: every median of every hot site is a zero with both signs present, i.e. what is printed ("0" / "-0") is whichever zero libstdc++'s
    introselect leaves at the middle of 50 000 values.  The workgroup of k_site_huge runs that selection itself (huge_tie_median: the Hoare
    sweeps as position lists built by scans); the bytes must be the oracle's (whose nth_element IS the library's), and the same with the
    one-thread walk forced"""
    from genomicsdb_amd import synth
    N, B, L = 50_000, 10_000_000, 1000
    g = synth.Generator(N, B, L + 2600, dense=(B, L, 50, 64), rank_sum_scale=2.0)
    cells, nc = g.chunk_bytes(B + L + 2600)
    g.close()
    sites = [B + 50 * i for i in range(1, 21)]
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    q["query_column_ranges"] = [[[s_, s_] for s_ in sites]]
    q["max_diploid_alt_alleles_that_can_be_genotyped"] = 10      # (PL / GT dropped at these sites: the test is about the INFO medians)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    infos = [l.split(b"\t")[7] for l in want.split(b"\n") if l]
    assert nrec == 20 and sum(i.count(b"=-0;") + i.count(b"=-0\t") + i.endswith(b"=-0") for i in infos) >= 5 and sum(b"RankSum=0;" in i for i in infos) >= 5
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 4096))
    got = b"".join(eng.run_interval(s_, s_, arena_bytes=1 << 30)[0] for s_ in sites)
    assert got == want
    monkeypatch.setenv("GDBAMD_NO_HUGE_SITES", "1")
    got2 = b"".join(eng.run_interval(s_, s_, arena_bytes=1 << 30)[0] for s_ in sites[:3])
    assert got2 == b"".join(want.splitlines(True)[:3])
    eng.close()


@pytest.mark.parametrize("n_samples,scale", [(40, 1.0), (333, 2.0), (333, 1.0)])
def test_tied_medians_print_the_zero_the_reference_selects(gdb, tmp_path, monkeypatch, n_samples, scale):
    """rank sums rounded to 1 / scale: most medians are ties and many of them zeros of both signs.  The per-thread medians
    (<= 48 calls), the workgroup medians (hot sites: 333 calls) and the sorted medians must all print the zero that
    std::nth_element leaves in the middle."""
    from genomicsdb_amd import synth
    N, B, L = n_samples, 10_000_000, 500
    g = synth.Generator(N, B, L + 2500, dense=(B + 100, 200, 50, 20), rank_sum_scale=scale)
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    assert want.count(b"=-0;") > 3 and want.count(b"=0;") > 3
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 4096))
    got, st = eng.run_interval(B, B + L - 1, arena_bytes=1 << 20)
    assert got == want
    monkeypatch.setenv("GDBAMD_SORTED_MEDIAN", "1")
    got, st = eng.run_interval(B, B + L - 1, arena_bytes=1 << 20)
    assert got == want
    eng.close()


@pytest.mark.parametrize("name", ["t0_1_2_loading", "t6_7_8_loading", "t0_overlapping_at_12202_partition_loading", "info_ops1"])
def test_vcf2tiledb_cli_imports_and_combines_in_line(gdb, tmp_path, name):
    """the reference's import command line (tools/src/vcf2tiledb.cc) on the reference's own test gVCFs, run from the fixture
    tree like run.py runs it from tests/: with produce_combined_vcf the loader's stdout is the '*_loading' golden; with
    produce_tiledb_array the array it leaves behind answers gt_mpi_gather queries with the query goldens."""
    import json
    import os
    import subprocess
    tool = os.path.join(helpers.ROOT, "genomicsdb_amd", "vcf2tiledb")
    assert os.path.exists(tool), "build() must produce the tool"
    _, callsets, vid, ov, golden, mode = [c for c in CASES if c[0] == name][0]
    assert mode == "load"
    ov = dict(ov)
    pb = ov.pop("partition_begin", 0)
    loader = {
        "row_based_partitioning": False, "produce_combined_vcf": True, "produce_tiledb_array": True,
        "column_partitions": [{"begin": pb, "workspace": str(tmp_path / "ws"), "array": "arr"}],
        "callset_mapping_file": os.path.join("inputs", "callsets", callsets), "vid_mapping_file": os.path.join("inputs", vid),
        "treat_deletions_as_intervals": True, "vcf_header_filename": os.path.join("inputs", "template_vcf_header.vcf"),
        "reference_genome": os.path.join("inputs", "chr1_10MB.fasta.gz"), "num_parallel_vcf_files": 1, "do_ping_pong_buffering": False,
        "size_per_column_partition": 3000, "offload_vcf_output_processing": False, "discard_vcf_index": True, "segment_size": 40,
    }
    loader.update(ov)
    lf = tmp_path / "loader.json"
    lf.write_text(json.dumps(loader))
    r = subprocess.run([tool, str(lf)], capture_output=True, timeout=120, cwd=helpers.GOLDEN)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == helpers.golden_text(golden)
    assert b"vcf2binary" in r.stderr and b"produce_combined_vcf" in r.stderr
    cells = (tmp_path / "ws" / "arr" / "cells.bin").read_bytes()
    if pb == 0:
        assert cells == helpers.cells_for(callsets, vid)
    if name == "t0_1_2_loading":      # query the imported array with the reference's query tool
        q, _ = helpers.query_json(callsets, vid, {"query_column_ranges": [{"range_list": [{"low": 0, "high": 1000000000}]}]}, "query")
        q["workspace"] = str(tmp_path / "ws")
        q["array"] = "arr"
        qf = tmp_path / "query.json"
        qf.write_text(json.dumps(q))
        r = subprocess.run([os.path.join(helpers.ROOT, "genomicsdb_amd", "gt_mpi_gather"), "-j", str(qf), "--produce-Broad-GVCF"], capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()
        assert r.stdout == helpers.golden_text("t0_1_2_vcf_at_0")
        # the loader's "compress_tiledb_array": the array also gets the columnar file with DEFLATE tiles; with cells.bin moved away the
        # query tool reads that file (tiles inflated on the device) and prints the same golden
        loader["compress_tiledb_array"] = True
        loader["produce_combined_vcf"] = False
        lf.write_text(json.dumps(loader))
        r = subprocess.run([tool, str(lf)], capture_output=True, timeout=120, cwd=helpers.GOLDEN)
        assert r.returncode == 0 and b"compress_tiledb_array" in r.stderr, r.stderr.decode()
        frag = tmp_path / "ws" / "arr" / "fragment.gdbamd"
        assert frag.exists() and frag.read_bytes()[8:12] == b"\x03\x00\x00\x00"
        os.rename(tmp_path / "ws" / "arr" / "cells.bin", tmp_path / "ws" / "arr" / "cells.bin.away")
        r = subprocess.run([os.path.join(helpers.ROOT, "genomicsdb_amd", "gt_mpi_gather"), "-j", str(qf), "--produce-Broad-GVCF"], capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()
        assert r.stdout == helpers.golden_text("t0_1_2_vcf_at_0")


def test_pages_stay_in_hbm_and_concat_over_rccl(gdb, tmp_path):
    """the pull interface of the C ABI hands out the pages where they are (HBM); torch aliases them without a copy and
    genomicsdb_amd.dist.gather_interval moves them with torch.distributed's "nccl" backend (= RCCL).  One GPU here, so the
    group has one rank: the collective and the device buffers are exercised, the 2-rank ordering is covered by the gloo test."""
    import socket
    import torch
    import torch.distributed as dist
    from genomicsdb_amd import dist as gdist, synth
    N, B, L = 200, 10_000_000, 3000
    eng, q, cells = _c2_engine(gdb, tmp_path, N, B, L)
    want, st = eng.run_interval(B, B + L - 1, arena_bytes=1 << 30)
    got = b"".join(bytes(t.cpu().numpy().tobytes()) for t in eng.page_tensors(B, B + L - 1, arena_bytes=1 << 20))
    assert st.num_records > 1000 and got == want
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        whole = gdist.gather_interval(eng, B, B + L - 1, arena_bytes=1 << 20, dst=0)
        whole_one_page = gdist.gather_interval(eng, B, B + L - 1, dst=0)        # the body as ONE page, sent where it lies
        assert torch.equal(whole, whole_one_page)
        assert whole.is_cuda and whole.dtype == torch.uint8
        assert bytes(whole.cpu().numpy().tobytes()) == want
    finally:
        dist.destroy_process_group()
    eng.close()


def test_empty_inputs(gdb, tmp_path):
    """edges the reference handles by doing nothing: a query interval without any cell, an array without cells, a row range
    whose samples have no data in the interval - the stream is the header alone (or an empty body), never an error"""
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    header = b"".join(l for l in helpers.golden_text(golden).splitlines(True) if l.startswith(b"#"))
    # (1) interval beyond the last cell
    q, pb = helpers.query_json(callsets, vid, {"query_column_ranges": [{"range_list": [{"low": 500_000_000, "high": 500_001_000}]}]}, mode)
    want, nrec, _ = helpers.oracle_run(q, cells, partition_begin=pb)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 16)
    got = s.read(); s.close()
    assert nrec == 0 and got == want == header
    # (2) interval between two cells of the array (a gap no interval covers)
    q, pb = helpers.query_json(callsets, vid, {"query_column_ranges": [{"range_list": [{"low": 13000, "high": 13010}]}]}, mode)
    want, nrec, _ = helpers.oracle_run(q, cells, partition_begin=pb)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 16)
    got = s.read(); s.close()
    assert got == want
    # (3) no cells at all
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=b"", buffer_capacity=1 << 16)
    got = s.read(); s.close()
    assert got == header
    # (4) engine level: empty interval -> zero records, zero bytes, zero pages
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    body, st = eng.run_interval(600_000_000, 600_000_100, arena_bytes=1 << 20)
    assert body == b"" and st.num_records == 0 and st.pages == 0
    assert list(eng.pages(600_000_000, 600_000_100)) == []
    eng.close()


def test_two_handles_in_two_threads(gdb, tmp_path):
    """the reference's threading rule at the boundary: one handle per thread, several handles per process.  Two engines with
    different queries work at the same time from two host threads (ctypes releases the GIL); each must produce what it
    produces alone.  Every pipeline owns ONE element of the per-interval context array in constant memory for its lifetime
    (nothing is handed over, there is no lock to prove): with the two plans differing in their FORMAT fields, a kernel of one
    engine reading the other's element would print the other's columns, in every one of the 12 x 2 concurrent intervals."""
    import hashlib
    import threading
    from genomicsdb_amd import synth
    B = 10_000_000
    specs = [(300, 4000, {}), (130, 6000, {"produce_GT_field": True})]
    engines, wants = [], []
    for i, (N, L, opts) in enumerate(specs):
        g = synth.Generator(N, B, L + 2500, seed=synth.SEED + i)
        cells, _ = g.chunk_bytes(B + L + 2500)
        d = tmp_path / ("q%d" % i)
        d.mkdir()
        q = helpers.synth_query(d, N, B, B + L - 1)
        q.update(opts)
        e = gdb.CombineEngine(q)
        e.stage_cells(cells)
        e.set_reference(B, synth.reference(B, L + 4096, seed=synth.SEED + i))
        body, st = e.run_interval(B, B + L - 1, arena_bytes=1 << 20)
        engines.append((e, L))
        wants.append(hashlib.sha256(body).hexdigest())
    errors = []

    def work(i):
        e, L = engines[i]
        try:
            for rep in range(12):
                body, _ = e.run_interval(B, B + L - 1, arena_bytes=1 << 20)
                if hashlib.sha256(body).hexdigest() != wants[i]:
                    errors.append((i, rep))
        except Exception as ex:      # noqa: BLE001
            errors.append((i, repr(ex)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []
    for e, _ in engines:
        e.close()


def test_pipelines_per_process_limit_is_reported(gdb, tmp_path):
    """GDBAMD_MAX_PIPELINES_PER_PROCESS (include/genomicsdb_amd.h; 256 since round 4 - round 3: 16): one engine more than that alive at a
    time is refused with a message that names the limit, closing one makes room again; engines far beyond the old 16 give the same bytes"""
    import os, re
    hdr = open(os.path.join(helpers.ROOT, "include", "genomicsdb_amd.h")).read()
    cap = int(re.search(r"#define GDBAMD_MAX_PIPELINES_PER_PROCESS (\d+)", hdr).group(1))
    assert cap >= 256
    from genomicsdb_amd import synth
    N, B, L = 5, 10_000_000, 101
    cells, _ = synth.Generator(N, B, L).chunk_bytes(B + L)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    engines = []
    try:
        for i in range(cap):
            engines.append(gdb.CombineEngine(q))
        with pytest.raises(gdb.GenomicsDBException, match="%d device pipelines" % cap):
            gdb.CombineEngine(q)
        engines.pop().close()
        engines.append(gdb.CombineEngine(q))
        outs = []
        for i in (0, 15, 16, 17, cap // 2, cap - 1):
            e = engines[i]
            e.stage_cells(cells)
            e.set_reference(B, synth.reference(B, L + 4096))
            got, st = e.run_interval(B, B + L - 1, arena_bytes=1 << 20)
            outs.append(got)
        assert len(set(outs)) == 1 and len(outs[0]) > 0
        want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
        assert outs[0] == want
    finally:
        for e in engines:
            e.close()


# ---- arrays larger than the staging budget: column windows streamed through HBM with carry-over ----------------------------
@pytest.mark.parametrize("case", SUPPORTED, ids=[c[0] for c in SUPPORTED])
def test_one_column_per_window_reproduces_the_goldens(gdb, case, monkeypatch):
    """staging budget of one byte: every begin column of the array is a window of its own, so every interval that is live
    across a column boundary - reference blocks, deletions, the overlapping-interval override - is carried over on the device;
    the stream must still be the reference's golden"""
    monkeypatch.setenv("GDBAMD_STAGE_BUDGET_BYTES", "1")
    name, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    got = s.read()
    s.close()
    assert got == helpers.golden_text(golden)


def _synth_cells(n_samples, B, L, seed=None, **kw):
    from genomicsdb_amd import synth
    g = synth.Generator(n_samples, B, L, **({"seed": seed} if seed else {}), **kw)
    cells, nc = g.chunk_bytes(B + L)
    g.close()
    return cells, nc


def test_windowed_streaming_equals_resident_bytes(gdb, tmp_path, monkeypatch):
    """an array of more than 8 staging budgets, from memory, from cells.bin, from the columnar fragment file and from a cell
    callback: each stream equals the all-resident stream and the oracle; a row-subset query keeps its boundary markers"""
    import json
    import os
    from genomicsdb_amd import synth
    N, B, L = 150, 10_000_000, 24_000
    cells, nc = _synth_cells(N, B, L)
    q = helpers.synth_query(tmp_path, N, B + 500, B + L - 700)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    monkeypatch.delenv("GDBAMD_STAGE_BUDGET_BYTES", raising=False)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    resident = s.read()
    s.close()
    budget = len(cells) // 9
    monkeypatch.setenv("GDBAMD_STAGE_BUDGET_BYTES", str(budget))
    # (a) cells in memory
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    assert s.read() == resident
    s.close()
    # (b) cells.bin in a workspace, through the file-based init; (c) the fragment file written from it, read window by window
    ws = tmp_path / "ws"
    (ws / "arr").mkdir(parents=True)
    (ws / "arr" / "cells.bin").write_bytes(cells)
    q2 = dict(q)
    q2["workspace"] = str(ws)
    q2["array"] = "arr"
    qf = tmp_path / "query.json"
    qf.write_text(json.dumps(q2))
    s = gdb.GenomicsDBQueryStream(query_json_file=str(qf), buffer_capacity=1 << 20)
    assert s.read() == resident
    s.close()
    monkeypatch.delenv("GDBAMD_STAGE_BUDGET_BYTES")
    e = gdb.CombineEngine(q)
    e.stage_cells(cells)
    e.save_fragment(ws / "arr" / "fragment.gdbamd")
    e.close()
    os.rename(ws / "arr" / "cells.bin", ws / "arr" / "cells.bin.away")     # only the fragment file is left
    monkeypatch.setenv("GDBAMD_STAGE_BUDGET_BYTES", str(budget))
    s = gdb.GenomicsDBQueryStream(query_json_file=str(qf), buffer_capacity=1 << 20)
    assert s.read() == resident
    s.close()
    # (d) engine level: a callback hands out 2 kb chunks; windows and intervals are driven by cover()
    g = synth.Generator(N, B, L)
    state = {"col": B}

    def next_chunk():
        if state["col"] >= B + L:
            return None
        state["col"] = min(B + L, state["col"] + 2000)
        p, n, _ = g.next_chunk(state["col"])
        return p, n
    e = gdb.CombineEngine(q)
    e.open_cell_callback(next_chunk)
    e.set_reference(B, synth.reference(B, L + 4096))
    body = b""
    pos, qe = B + 500, B + L - 700
    nwin = 0
    while pos <= qe:
        lo, hi = e.cover(pos)
        assert lo <= pos <= hi
        b, st = e.run_interval(pos, min(qe, hi), arena_bytes=1 << 20)
        body += b
        pos = hi + 1
        nwin += 1
    e.close()
    assert nwin >= 8
    e2 = gdb.CombineEngine(q)
    e2.stage_cells(cells)
    e2.set_reference(B, synth.reference(B, L + 4096))
    whole, _ = e2.run_interval(B + 500, qe, arena_bytes=1 << 20)
    e2.close()
    assert body == whole
    assert whole == want
    # row subset: cells of the other samples stay behind as boundary markers in every window
    q3 = dict(q)
    q3["query_row_ranges"] = [{"range_list": [{"low": 3, "high": 40}, {"low": 77, "high": 77}]}]
    monkeypatch.delenv("GDBAMD_STAGE_BUDGET_BYTES")
    s = gdb.GenomicsDBQueryStream(query_json=q3, cells=cells, buffer_capacity=1 << 20)
    sub_resident = s.read()
    s.close()
    monkeypatch.setenv("GDBAMD_STAGE_BUDGET_BYTES", str(budget))
    s = gdb.GenomicsDBQueryStream(query_json=q3, cells=cells, buffer_capacity=1 << 20)
    assert s.read() == sub_resident
    s.close()


def test_pinned_memory_cells_streamed_in_windows(gdb, tmp_path, monkeypatch):
    """the caller's cells page-locked with gdbamd_pin_host_memory (DMA copies under the kernels of the window in use): the windowed
    stream from the pinned range equals the oracle"""
    import numpy as np
    from genomicsdb_amd import synth, api
    N, B, L = 120, 10_000_000, 20_000
    cells, nc = _synth_cells(N, B, L)
    q = helpers.synth_query(tmp_path, N, B + 300, B + L - 500)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    keep = np.frombuffer(cells, dtype=np.uint8).copy()
    api.pin_host_memory(keep.ctypes.data, keep.nbytes)
    try:
        monkeypatch.setenv("GDBAMD_STAGE_BUDGET_BYTES", str(len(cells) // 7))
        e = gdb.CombineEngine(q)
        e.open_memory_cells((keep.ctypes.data, keep.nbytes))
        e.set_reference(B, synth.reference(B, L + 4096))
        body = b""
        pos, qe = B + 300, B + L - 500
        nwin = 0
        while pos <= qe:
            lo, hi = e.cover(pos)
            b, st = e.run_interval(pos, min(qe, hi), arena_bytes=1 << 20)
            body += b
            pos = hi + 1
            nwin += 1
        e.close()
        assert nwin >= 3
        assert body == want
    finally:
        api.unpin_host_memory(keep.ctypes.data)


def test_stale_or_foreign_fragment_files_are_refused(gdb, tmp_path):
    """a fragment file is checked against the array schema, the callset mapping and its own size before a byte of it reaches a
    kernel: a truncated file, a file written under another callset mapping and a file with a doctored header are errors (or,
    when a cells.bin is at hand, quietly replaced by it)"""
    import json
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, _ = helpers.query_json(callsets, vid, ov, mode)
    ws = tmp_path / "ws"
    (ws / "arr").mkdir(parents=True)
    frag = ws / "arr" / "fragment.gdbamd"
    e = gdb.CombineEngine(q)
    e.stage_cells(cells)
    e.save_fragment(frag)
    good = frag.read_bytes()
    # truncated
    frag.write_bytes(good[: len(good) - 40])
    with pytest.raises(gdb.GenomicsDBException):
        e.load_fragment(frag)
    # the engine still serves its staged fragment after the failed load
    body, _ = e.run_interval(0, 10**9, arena_bytes=1 << 20)
    assert body and helpers.golden_text(golden).endswith(body)
    # implausible cell count in the header
    bad = bytearray(good)
    bad[16:24] = (2**40).to_bytes(8, "little")
    frag.write_bytes(bytes(bad))
    with pytest.raises(gdb.GenomicsDBException):
        e.load_fragment(frag)
    e.close()
    # another callset mapping (t6_7_8) must not be served this file; with a cells.bin next to it the stream falls back to that
    case2 = [c for c in CASES if c[0] == "t6_7_8_vcf_at_0"][0]
    _, callsets2, vid2, ov2, golden2, mode2 = case2
    q2, _ = helpers.query_json(callsets2, vid2, ov2, mode2)
    frag.write_bytes(good)
    e2 = gdb.CombineEngine(q2)
    with pytest.raises(gdb.GenomicsDBException):
        e2.load_fragment(frag)
    e2.close()
    (ws / "arr" / "cells.bin").write_bytes(helpers.cells_for(callsets2, vid2))
    q2["workspace"] = str(ws)
    q2["array"] = "arr"
    qf = tmp_path / "q2.json"
    qf.write_text(json.dumps(q2))
    s = gdb.GenomicsDBQueryStream(query_json_file=str(qf), buffer_capacity=1 << 20)
    assert s.read() == helpers.golden_text(golden2)
    s.close()


def test_jni_natives_driven_like_the_jvm_would(gdb, tmp_path):
    """the seven JNI natives (csrc/jni/jni_query_stream.cc) called in GATK4's order through a JNIEnv function table at the
    specification's indices (tests/jni_harness): OneTimeInitialize, Init(chr, start, end), Read into the middle of a small Java
    array until 0, Close - the stream is the golden; an unknown contig surfaces as a pending IOException, not a crash"""
    import ctypes
    import json
    import subprocess
    import os
    subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "jni_harness")])
    H = ctypes.CDLL(os.path.join(helpers.ROOT, "tests", "jni_harness", "libjniharness.so"))
    H.jni_harness_read_stream.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_uint64]
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    q, _ = helpers.query_json(callsets, vid, ov, mode)
    ws = tmp_path / "ws"
    (ws / "t0_1_2").mkdir(parents=True)
    (ws / "t0_1_2" / "cells.bin").write_bytes(helpers.cells_for(callsets, vid))
    q["workspace"] = str(ws)
    q["array"] = "t0_1_2"
    qf = tmp_path / "query.json"
    qf.write_text(json.dumps(q))

    def run(chrom, start, end, array_len, byte_first):
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        err = ctypes.create_string_buffer(2048)
        rc = H.jni_harness_read_stream(b"", str(qf).encode(), chrom, start, end, 0, array_len, byte_first, ctypes.byref(out), ctypes.byref(n), err, 2048)
        if rc != 0:
            return None, err.value.decode()
        data = ctypes.string_at(out.value, n.value)
        H.jni_harness_free(out)
        return data, ""
    want = helpers.golden_text(golden)
    got, e = run(b"", 0, 0, 4096, 0)
    assert got == want, e
    got, e = run(b"", 0, 0, 100, 1)          # first byte through ReadNextByte, then 100-byte reads
    assert got == want, e
    # GATK's query(chr, start, end): contig "1", 1-based inclusive positions -> the same records as the column interval query
    hdr_len = len(gdb.CombineEngine(q).header)
    got, e = run(b"1", 12141, 12295, 4096, 0)
    assert got is not None and got[:hdr_len] == want[:hdr_len] and got[hdr_len:] and got[hdr_len:] in want
    got, e = run(b"no_such_contig", 1, 10, 4096, 0)
    assert got is None and "contig" in e
    # is_bcf = true, what GATK4's GenomicsDBFeatureReader asks for with BCF2Codec
    out, n = ctypes.c_void_p(), ctypes.c_uint64()
    err = ctypes.create_string_buffer(2048)
    assert H.jni_harness_read_stream(b"", str(qf).encode(), b"", 0, 0, 1, 3000, 0, ctypes.byref(out), ctypes.byref(n), err, 2048) == 0, err.value
    bcf = ctypes.string_at(out.value, n.value)
    H.jni_harness_free(out)
    assert bcf[:5] == b"BCF\x02\x02" and helpers.bcf_stream_to_text(bcf) == want


# ---- BCF2 ("bu"), what GATK4's GenomicsDBFeatureReader decodes with BCF2Codec -----------------------------------------------
@pytest.mark.parametrize("case", SUPPORTED, ids=[c[0] for c in SUPPORTED])
def test_bcf_stream_decodes_to_the_golden_text(gdb, case, monkeypatch):
    """output format "bu": the stream is 'BCF\\2\\2' + header + typed records; decoded by the tests' own BCF2 reader and printed
    the way htslib prints a record (tests/tools/bcf2text.py) it must be the reference's TEXT golden, header included - with and
    without IDX keys in the header, in one page and in pages of a few records"""
    name, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    want = helpers.golden_text(golden)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
    one_page = s.read()
    s.close()
    assert one_page[:5] == b"BCF\x02\x02"
    assert helpers.bcf_stream_to_text(one_page) == want
    assert b",IDX=" in one_page
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True, keep_idx_fields_in_bcf_header=False)
    no_idx = s.read()
    s.close()
    assert b",IDX=" not in no_idx and helpers.bcf_stream_to_text(no_idx) == want
    monkeypatch.setenv("GDBAMD_DEVICE_PAGE_BYTES", "700")
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
    paged = s.read()
    s.close()
    assert paged == one_page


def test_bcf_htsjdk_flag_uses_missing_values_only(gdb):
    """use_missing_values_only_not_vector_end (the JNI flag for htsjdk, which has no vector-end values): no vector-end code
    anywhere in the FORMAT blocks, absent GT = no-call alleles; the records still decode"""
    import bcf2text
    case = [c for c in CASES if c[0] == "t0_haploid_triploid_1_2_3_triploid_deletion_vcf"][0]
    name, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True, use_missing_values_only_not_vector_end=True)
    data = s.read()
    s.close()
    hdr, recs = bcf2text.parse_stream(data)
    assert recs
    text = helpers.bcf_stream_to_text(data).decode()
    body = [l for l in text.split("\n") if l and not l.startswith("#")]
    want = [l for l in helpers.golden_text(golden).decode().split("\n") if l and not l.startswith("#")]
    assert len(body) == len(want)
    for got_line, want_line in zip(body, want):
        g, w = got_line.split("\t"), want_line.split("\t")
        assert g[:9] == w[:9]
        for gs, ws in zip(g[9:], w[9:]):       # same values; shorter vectors are padded with '.' instead of being cut
            for gf, wf in zip(gs.split(":"), ws.split(":")):
                gf = gf.replace("-65", ".")      # a GT padded with the int8 MISSING code, as htslib's bcf_format_gt prints it ((-128 >> 1) - 1)
                gv, wv = gf.replace("/", ",").replace("|", ",").split(","), wf.replace("/", ",").replace("|", ",").split(",")
                assert gv[:len(wv)] == wv or wv == ["."], (gf, wf)
                assert all(x == "." for x in gv[len(wv):]), (gf, wf, got_line[:120])


def test_bcf_synthetic_widths_and_text_agree(gdb, tmp_path):
    """300 synthetic samples x 5 kb: int8 / int16 / int32 FORMAT vectors chosen per record like htslib's bcf_enc_vint, PL
    vectors of every merged-allele count; the decoded BCF stream equals the text stream of the same engine (which equals the
    oracle, test_synthetic_matches_oracle)"""
    import bcf2text
    import struct
    from genomicsdb_amd import synth
    N, B, L = 300, 10_000_000, 5000
    cells, _ = _synth_cells(N, B, L)
    q = helpers.synth_query(tmp_path, N, B + 300, B + L - 400)
    ref = synth.reference(B, L + 16)
    et = gdb.CombineEngine(q)
    et.stage_cells(cells)
    et.set_reference(B, ref)
    text_body, st = et.run_interval(B + 300, B + L - 400, arena_bytes=1 << 30)
    hdr_text = et.header
    et.close()
    eb = gdb.CombineEngine(q, is_bcf=True)
    eb.stage_cells(cells)
    eb.set_reference(B, ref)
    bcf_body, sb = eb.run_interval(B + 300, B + L - 400, arena_bytes=1 << 30)
    bcf_paged, sp = eb.run_interval(B + 300, B + L - 400, arena_bytes=1 << 20)
    eb.close()
    assert sb.num_records == st.num_records and sp.pages > 3 and bcf_paged == bcf_body
    h = bcf2text.Header(hdr_text.decode())
    at, lines, types_seen = 0, [], set()
    while at < len(bcf_body):
        l_shared, l_indiv = struct.unpack_from("<II", bcf_body, at)
        rec = bcf_body[at:at + 8 + l_shared + l_indiv]
        lines.append(bcf2text.record_to_text(h, rec, helpers.format_float))
        at += len(rec)
    assert ("\n".join(lines) + "\n").encode() == text_body
    assert len(bcf_body) < len(text_body)


def test_overlaps_filters_and_ids_on_synthetic_input(gdb, tmp_path):
    """the generator's overlap / FILTER / ID modes at a size where every record unites several samples: a sample's next record
    beginning inside its reference block or deletion (overlap override, query_variants.cc:512-543), FILTER unions of one id,
    sorted ID-token unions; two DIFFERENT filter ids in one record come in the iteration order of the reference's
    std::unordered_set<int> (tests/test_filter_union_order.py)"""
    from genomicsdb_amd import synth
    N, B, L = 400, 10_000_000, 3000
    g = synth.Generator(N, B, L + 2500, overlap_permille=300, filter_permille=400, id_permille=500, with_id=True)
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B + 200, B + L - 300, with_id=True)
    q["produce_FILTER_field"] = True
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    lines = [l.split(b"\t") for l in want.split(b"\n") if l]
    assert sum(1 for l in lines if b";" in l[2]) > 20 and sum(1 for l in lines if l[6] == b"LowQual") > 100
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 4096))
    got, st = eng.run_interval(B + 200, B + L - 300, arena_bytes=1 << 22)
    assert st.num_records == nrec and got == want
    eng.close()
    g2 = synth.Generator(N, B, L + 2500, filter_permille=400, filter2_permille=400)
    cells2, _ = g2.chunk_bytes(B + L + 2500)
    q2 = helpers.synth_query(tmp_path, N, B + 200, B + L - 300)
    q2["produce_FILTER_field"] = True
    want2, nrec2, _ = helpers.oracle_run_synth(q2, cells2, synth.SEED, with_header=False)
    assert sum(1 for l in want2.split(b"\n") if l and b";" in l.split(b"\t")[6]) > 20
    eng = gdb.CombineEngine(q2)
    eng.stage_cells(cells2)
    eng.set_reference(B, synth.reference(B, L + 4096))
    got2, st2 = eng.run_interval(B + 200, B + L - 300, arena_bytes=1 << 22)
    assert st2.num_records == nrec2 and got2 == want2
    eng.close()


def _stream_checksum(eng, begin, end, arena_bytes, split_every=None):
    """position-weighted checksums of the VCF body of [begin, end], computed on the device where the pages lie (no host copy):
    independent of how the stream was paged or cut into pieces"""
    import torch
    MOD = 1_000_003
    s1 = s2 = s3 = 0          # exact (Python integers): a chunk's partial sums stay below 2^63, the totals need not
    offset = 0
    pieces = [(begin, end)]
    if split_every:      # cuts that keep the stream byte-identical sit right before a cell begin: the engine names them
        pieces, cur = [], begin
        while cur <= end:
            pe = eng.split_point(cur, end, split_every)
            pieces.append((cur, pe))
            cur = pe + 1
    npages = 0
    for pb, pe in pieces:
        for page in eng.page_tensors(pb, pe, arena_bytes=arena_bytes):
            npages += 1
            n = page.numel()
            for c0 in range(0, n, 1 << 28):
                x = page[c0:c0 + (1 << 28)].to(torch.int64)
                w = (torch.arange(offset + c0, offset + c0 + x.numel(), device="cuda", dtype=torch.int64) % MOD) + 1
                s1 += int(x.sum())
                s2 += int((x * w).sum())
                s3 += int(((x == 10).to(torch.int64) * w).sum())      # newline positions: the record boundaries
            offset += n
    return offset, s1, s2, s3, npages


def test_c2_full_size_two_pagings_agree(gdb, tmp_path):
    """BASELINE.json configs[1] at its full size - 1 000 samples x 10 Mb, the workload bench.py times: ten 1 Mb windows with one
    44 GB page each against 500 kb pieces in 3 GB pages; byte count, byte sum and two position-weighted checksums of the stream,
    taken on the device, must agree (the oracle needs hours for this size; at 1 500 bp it is compared byte for byte)"""
    import torch
    from genomicsdb_amd import synth
    N, B, L = 1000, 10_000_000, 10_000_000
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    eng = gdb.CombineEngine(q)
    g = synth.Generator(N, B, L)
    eng.stage_cells_begin()
    col = B
    while col < B + L:
        col = min(B + L, col + 1_000_000)
        ptr, nbytes, nc = g.next_chunk(col)
        eng.stage_cells_append(ptr, nbytes)
    eng.stage_cells_end()
    g.close()
    eng.set_reference(B, synth.reference(B, L + 4096))
    totals_a, totals_b = [], []
    for w in range(10):
        wb, we = B + w * 1_000_000, B + (w + 1) * 1_000_000 - 1
        totals_a.append(_stream_checksum(eng, wb, we, 48 << 30))
        totals_b.append(_stream_checksum(eng, wb, we, 3 << 30, split_every=500_000))
        assert totals_a[-1][4] == 1 and totals_b[-1][4] > 10
        assert totals_a[-1][:4] == totals_b[-1][:4], "window %d" % w
        assert totals_a[-1][0] > 40e9
    eng.close()


def test_c2_full_size_interior_windows_match_the_oracle(gdb, tmp_path):
    """The array bench.py times (BASELINE.json configs[1]: 1 000 samples x 10 Mb, staged in 1 Mb parts exactly like the bench)
    against the ORACLE at interior positions: 300-bp query windows at seeded random offsets and across the 1 Mb cuts between the
    bench's steps.  The oracle cannot scan 10 Mb, but a record depends only on the cells that overlap it, and no interval of the
    generator is longer than 2 000 bp: the cells that begin in [s - 2 100, s + 300) are everything the reference would look at for
    the query [s, s + 299] (left sweep included).  The generator hands them over as chunks of their own while the array is staged."""
    import ctypes
    import random
    from genomicsdb_amd import synth
    N, B, L, W, REACH = 1000, 10_000_000, 10_000_000, 300, 2100
    rnd = random.Random(20260930)
    spots = [B + k * 1_000_000 - W // 2 for k in (1, 5, 9)]                         # straddle a cut between two bench steps
    spots += [B + k * 1_000_000 - 1 for k in (3,)] + [B + 7 * 1_000_000]            # end exactly in front of / begin exactly at a cut
    spots += [B + rnd.randrange(REACH + 10, L - W - 10) for _ in range(6)]          # anywhere
    spots = sorted(spots)
    cuts = sorted(set([B + i * 1_000_000 for i in range(1, 11)] + [s - REACH for s in spots] + [s + W for s in spots]))
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    eng = gdb.CombineEngine(q)
    g = synth.Generator(N, B, L)
    eng.stage_cells_begin()
    kept = []                                   # (first column, end column, bytes) of the chunks some window needs
    prev = B
    for col in cuts:
        ptr, nbytes, nc = g.next_chunk(col)
        eng.stage_cells_append(ptr, nbytes)
        if any(s - REACH <= prev and col <= s + W for s in spots):
            kept.append((prev, col, ctypes.string_at(ptr, nbytes)))
        prev = col
    eng.stage_cells_end()
    g.close()
    eng.set_reference(B, synth.reference(B, L + 4096))
    total = 0
    for i, s in enumerate(spots):
        cells = b"".join(c for lo, hi, c in kept if s - REACH <= lo and hi <= s + W)
        assert len(cells) > 1000 * 150
        d = tmp_path / ("spot%d" % i)
        d.mkdir()
        qs = helpers.synth_query(d, N, s, s + W - 1)
        want, nrec, _ = helpers.oracle_run_synth(qs, cells, synth.SEED, with_header=False)
        got, st = eng.run_interval(s, s + W - 1, arena_bytes=64 << 20)
        assert st.num_records == nrec and nrec >= W - 5, (s, nrec, st.num_records)
        assert got == want, "window at column %d" % s
        total += nrec
    assert total > 3000
    eng.close()


def test_c3_width_interior_window_of_a_streamed_array_matches_the_oracle(gdb, tmp_path, monkeypatch):
    """BASELINE.json configs[2]'s width (10 000 samples) the way the bench's c3 leg runs it - cells streamed through HBM in column
    windows with the carry-over between them - checked against the oracle on an interior window: 60 kb of cells (880 MB) through a
    64 MiB staging budget, the query window 120 bp wide two thirds into the array"""
    import ctypes
    import numpy as np
    from genomicsdb_amd import synth, api
    N, B, L, W, REACH = 10_000, 10_000_000, 60_000, 120, 2100
    s = B + 41_234
    g = synth.Generator(N, B, L)
    parts, spot = [], []
    prev = B
    for col in (s - REACH, s + W, B + L):
        ptr, nbytes, nc = g.next_chunk(col)
        parts.append(ctypes.string_at(ptr, nbytes))
        if prev == s - REACH:
            spot.append(parts[-1])
        prev = col
    g.close()
    cells = b"".join(parts)
    qs = helpers.synth_query(tmp_path, N, s, s + W - 1)
    want, nrec, _ = helpers.oracle_run_synth(qs, spot[0], synth.SEED, with_header=False)
    assert nrec >= W - 5
    keep = np.frombuffer(cells, dtype=np.uint8)
    monkeypatch.setenv("GDBAMD_STAGE_BUDGET_BYTES", str(64 << 20))
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    e = gdb.CombineEngine(q)
    e.open_memory_cells((keep.ctypes.data, keep.nbytes))
    e.set_reference(B, synth.reference(B, L + 4096))
    pos, nwin, body = B, 0, b""
    while pos <= s + W - 1:                       # walk the windows like the bench's streamed leg; fetch only the spot
        lo, hi = e.cover(pos)
        a, b = max(pos, s), min(hi, s + W - 1)
        if a <= b:
            part, st = e.run_interval(a, b, arena_bytes=256 << 20)
            body += part
        pos = hi + 1
        nwin += 1
    e.close()
    assert body == want
    assert nwin >= 3                              # the spot lies behind at least two window cuts (carry-over twice)


@pytest.mark.parametrize("lanes", [2, 3])
def test_intervals_in_flight_on_lanes_give_the_same_bytes(gdb, tmp_path, lanes):
    """gdbamd_engine_run_intervals: several windows of one staged fragment in flight at a time, every lane a device pipeline of its own
    over the SAME fragment (the sizing kernels of one window overlap with the page kernel of another: bench.py --lanes).  The bodies
    are the bodies of one window after the other, the whole is the oracle's output; restaging is followed (the lanes adopt the new
    fragment) and a lane count above the interval count is fine."""
    from genomicsdb_amd import synth
    N, B, L, W = 300, 10_000_000, 4200, 600
    eng, q, cells = _c2_engine(gdb, tmp_path, N, B, L)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    wins = [(B + i * W, B + (i + 1) * W - 1) for i in range(L // W)]
    seq = [eng.run_interval(a, b, arena_bytes=1 << 20)[0] for a, b in wins]
    assert b"".join(seq) == want
    res = eng.run_intervals(wins, arena_bytes=1 << 20, lanes=lanes, fetch=True)
    assert [r[0] for r in res] == seq
    assert sum(r[1].num_records for r in res) == nrec and all(r[1].pages >= 1 for r in res)
    stats = eng.run_intervals(wins[:2], arena_bytes=8 << 20, lanes=4)                       # pages left in HBM, more lanes than intervals
    assert [s.num_records for s in stats] == [r[1].num_records for r in res[:2]]
    # another fragment in the engine: the lanes must follow it
    g = synth.Generator(N, B, L + 2500, seed=77)
    cells2, _ = g.chunk_bytes(B + L + 2500)
    eng.stage_cells(cells2)
    eng.set_reference(B, synth.reference(B, L + 2500 + 4096, seed=77))
    want2, nrec2, _ = helpers.oracle_run_synth(q, cells2, 77, with_header=False)
    assert want2 != want
    res2 = eng.run_intervals(wins, arena_bytes=1 << 20, lanes=lanes, fetch=True)
    assert b"".join(r[0] for r in res2) == want2
    # round 6: what a new lane costs (the estimate run_intervals clamps new lanes with against the device's free memory), lanes given back,
    # and a single interval on the engine's own pipeline afterwards (the lanes' page-priority stream is switched off again behind a run)
    fp = eng.lane_footprint(W, arena_bytes=1 << 20)
    assert fp >= (1 << 20) + 18 * N * W and fp < (8 << 30)
    assert eng.lane_footprint(1_000_000, arena_bytes=64 << 30) > 45 * N * 1_000_000
    eng.release_lanes()
    assert eng.run_interval(wins[0][0], wins[0][1], arena_bytes=1 << 20)[0] == res2[0][0]
    res3 = eng.run_intervals(wins, arena_bytes=1 << 20, lanes=lanes, fetch=True)             # the lanes are created again
    assert b"".join(r[0] for r in res3) == want2
    eng.close()


@pytest.mark.parametrize("fmt", ["bu", "z"])
def test_lanes_give_the_same_bytes_in_the_other_output_formats(gdb, tmp_path, fmt):
    """the same through the BCF2 page assembly (k_bcf_write on the lanes' high-priority stream) and the on-device BGZF compressor:
    three windows in flight against one window after the other; the BCF2 records decode to the oracle's text"""
    import struct
    import zlib
    from genomicsdb_amd import synth
    import bcf2text
    N, B, L, W = 300, 10_000_000, 3600, 600
    g = synth.Generator(N, B, L + 2500)
    cells, _ = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    eng = gdb.CombineEngine(q, output_format=fmt)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 2500 + 4096))
    wins = [(B + i * W, B + (i + 1) * W - 1) for i in range(L // W)]
    seq = [eng.run_interval(a, b, arena_bytes=1 << 20)[0] for a, b in wins]
    res = eng.run_intervals(wins, arena_bytes=1 << 20, lanes=3, fetch=True)
    assert [r[0] for r in res] == seq
    assert sum(r[1].num_records for r in res) == nrec
    body = b"".join(seq)
    if fmt == "bu":
        et = gdb.CombineEngine(q)
        h = bcf2text.Header(et.header.decode())
        et.close()
        at, lines = 0, []
        while at < len(body):
            l_shared, l_indiv = struct.unpack_from("<II", body, at)
            lines.append(bcf2text.record_to_text(h, body[at:at + 8 + l_shared + l_indiv], helpers.format_float))
            at += 8 + l_shared + l_indiv
        assert ("\n".join(lines) + "\n").encode() == want
    else:
        out, at = [], 0
        while at < len(body):                                   # BGZF: a series of gzip members
            d = zlib.decompressobj(31)
            out.append(d.decompress(body[at:]))
            at = len(body) - len(d.unused_data)
        assert b"".join(out) == want
    eng.close()


def test_c5_one_piece_of_2000_columns_at_50000_samples(gdb, tmp_path):
    """BASELINE.json configs[4] at its stated width, one of the 50 pieces tests/tools/c5_full.py works off: 50 000 samples, 2 000 columns
    of the dense region (every sample starts an insertion out of a pool of 64 alleles every 50 columns: 40 hot sites of 50 000 calls,
    ~66 merged alleles, PL vectors of 2 278 genotypes; 5 M cells, ~12 GB of text).  Checked through what does not need the oracle at
    this size - two pagings and a cut into pieces give the same stream (device-side checksums), one line per record, a hot record
    has 9 + N columns with G = (A + 1)(A + 2) / 2 PL values - and against the ORACLE on the columns in front of the first hot site."""
    import ctypes
    import torch
    from genomicsdb_amd import synth
    N, B, L = 50_000, 10_000_000, 2000
    g = synth.Generator(N, B, L + 3000, dense=(B, L, 50, 64))
    eng_q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    eng_q["max_diploid_alt_alleles_that_can_be_genotyped"] = 64
    eng = gdb.CombineEngine(eng_q)
    eng.stage_cells_begin()
    head_cells = b""
    for col in (B + 45, B + L + 3000):
        ptr, nbytes, nc = g.next_chunk(col)
        if col == B + 45:
            head_cells = ctypes.string_at(ptr, nbytes)
        eng.stage_cells_append(ptr, nbytes)
    eng.stage_cells_end()
    g.close()
    eng.set_reference(B, synth.reference(B, L + 8000))
    a = _stream_checksum(eng, B, B + L - 1, 16 << 30)
    b = _stream_checksum(eng, B, B + L - 1, 1 << 30, split_every=500)
    assert a[:4] == b[:4] and b[4] > a[4] and a[0] > 8e9
    # the columns in front of the first hot site: every live interval of them begins in the first chunk
    d = tmp_path / "head"
    d.mkdir()
    qh = helpers.synth_query(d, N, B, B + 39)
    qh["max_diploid_alt_alleles_that_can_be_genotyped"] = 64
    want, nrec, _ = helpers.oracle_run_synth(qh, head_cells, synth.SEED, with_header=False)
    got, st = eng.run_interval(B, B + 39, arena_bytes=1 << 30)
    assert st.num_records == nrec and got == want
    # one hot record where it lies in HBM: columns and PL length
    pages = eng.page_tensors(B + 50, B + 50, arena_bytes=2 << 30)
    rec = b"".join(bytes(t.cpu().numpy().tobytes()) for t in pages)
    lines = rec.split(b"\n")
    assert lines[-1] == b"" and len(lines) == 2
    cols = lines[0].split(b"\t")
    assert len(cols) == 9 + N
    A = len(cols[4].split(b","))                       # ALT alleles incl. <NON_REF>
    assert A >= 60 and cols[4].endswith(b"<NON_REF>")
    fmt = cols[8].split(b":")
    if A > 64:                                         # more ALT alleles than max_diploid_alt_alleles_that_can_be_genotyped: the G-length fields are dropped
        assert b"PL" not in fmt and b"AD" in fmt
        assert len(cols[9].split(b":")[fmt.index(b"AD")].split(b",")) == A + 1
    else:
        pl = cols[9].split(b":")[fmt.index(b"PL")]
        assert len(pl.split(b",")) == (A + 1) * (A + 2) // 2
    eng.close()


def test_c4_sample_count_100000_rows_matches_oracle(gdb, tmp_path):
    """BASELINE.json configs[3]'s sample count on a narrow window: 100 000 samples (1 563 chunks of 64 sample columns, a first
    record with 100 000 calls starting at the partition begin - allele merge, medians and sums by its workgroup), 24 columns"""
    from genomicsdb_amd import synth
    N, B, L = 100_000, 10_000_000, 24
    g = synth.Generator(N, B, 400)
    cells, nc = g.chunk_bytes(B + 400)
    g.close()
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, 4096))
    got, st = eng.run_interval(B, B + L - 1, arena_bytes=64 << 20)
    assert st.num_records == nrec and st.pages > 1
    assert got == want
    eng.close()


def test_partitions_imported_separately_concatenate_to_the_whole_scan(gdb):
    """two column partitions imported on their own (the second one begins at 12202, inside reference blocks of all three samples:
    the importer replays those intervals at the partition begin, load_operators.cc:33-79) and scanned on the device; their
    bodies back to back = one scan of the whole array over the query intervals [0, 12201], [12202, ...]"""
    import os
    v = os.path.join(helpers.GOLDEN, "inputs", "vid.json")
    c = os.path.join(helpers.GOLDEN, "inputs", "callsets", "t0_1_2.json")
    bodies = []
    for begin, end in ((0, 12201), (12202, 2**62)):
        cells, ncells = gdb.import_cells(v, c, file_root=helpers.GOLDEN, column_begin=begin, column_end=end)
        qj, _ = helpers.query_json("t0_1_2.json", "vid.json", {"query_column_ranges": [[[begin, min(end, 1_000_000_000)]]]}, "query")
        e = gdb.CombineEngine(qj)
        e.stage_cells(cells)
        body, st = e.run_interval(begin, min(end, 1_000_000_000), arena_bytes=1 << 20)
        bodies.append(body)
        e.close()
    full = helpers.cells_for("t0_1_2.json", "vid.json")
    qj, _ = helpers.query_json("t0_1_2.json", "vid.json", {"query_column_ranges": [[[0, 12201], [12202, 1_000_000_000]]]}, "query")
    want, nrec, _ = helpers.oracle_run(qj, full, with_header=False)
    assert b"".join(bodies) == want


def test_reference_shaped_cpp_caller_produces_the_golden(gdb, tmp_path):
    """tests/compat/gt_mpi_gather_shaped (a C++ caller in the shape of the reference's tool, on the source-compatible operator
    classes): stdout is the golden, in one go and in 128-byte batches through VCFSerializedBufferAdapter + RWBuffer; a per-record
    operator of the caller's own is refused, the batched hook gets the pages"""
    import json
    import os
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "compat")])
    tool = os.path.join(helpers.ROOT, "tests", "compat", "gt_mpi_gather_shaped")
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    q, _ = helpers.query_json(callsets, vid, ov, mode)
    ws = tmp_path / "ws"
    (ws / "t0_1_2").mkdir(parents=True)
    (ws / "t0_1_2" / "cells.bin").write_bytes(helpers.cells_for(callsets, vid))
    q["workspace"] = str(ws)
    q["array"] = "t0_1_2"
    qf = tmp_path / "query.json"
    qf.write_text(json.dumps(q))
    for page in ("0", "128"):
        r = subprocess.run([tool, str(qf), page], capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()
        assert r.stdout == helpers.golden_text(golden)
    r = subprocess.run([tool, str(qf), "0", "more"], capture_output=True, timeout=120)
    assert r.returncode == 0 and b"per-record operator refused: yes" in r.stderr and b"batched hook: 1 pages" in r.stderr
    # the reference's profiling counters, as GTProfileStats::print_stats prints them (three samples, 5 begin-cells, 4 records)
    r = subprocess.run([tool, str(qf), "0"], capture_output=True, timeout=120, env=dict(os.environ, GDBAMD_PRINT_PROFILE="1"))
    assert r.returncode == 0 and r.stdout == helpers.golden_text(golden)
    lines = r.stderr.decode().splitlines()
    assert "stat_name,sum,sum_sq,mean,std-dev" in lines
    stat = {l.split(",")[0]: l.split(",")[1:] for l in lines if l.startswith("GT_NUM_")}
    assert stat["GT_NUM_CELLS"][0] == "5" and stat["GT_NUM_OPERATOR_INVOCATIONS"][0] == "4" and stat["GT_NUM_VALID_CELLS_IN_QUERY"][:2] == ["5", "25"]
    assert stat["GT_NUM_CELLS_IN_LEFT_SWEEP"][0] == "0" and len(stat) == 6


@pytest.mark.gpu
@pytest.mark.parametrize("seed,is_bcf", [(5, False), (6, False), (7, True)])
def test_allele_specific_annotations_fuzz_on_device(gdb, seed, is_bcf):
    """random per-allele vectors / histograms (empty vectors, NaN, repeated bins, ALT subsets per sample) through the device path:
    2-D element_wise_sum and histogram_sum (gdb_asa.hpp) against the oracle, as VCF text and decoded from the BCF2 stream"""
    import numpy as np
    from test_hostsim_golden import _asa_cells
    rng = np.random.default_rng(seed)
    cells = _asa_cells(rng, 3, 60)
    q, pb = helpers.query_json("t0_1_2_all_asa.json", "vid_all_asa.json", {}, "load")
    want, nrec, _ = helpers.oracle_run(q, cells, partition_begin=pb)
    assert nrec >= 40 and want.count(b"AS_RAW_MQRankSum=") > 15
    if not is_bcf:
        s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
        got = s.read()
        s.close()
        assert got == want
    else:
        s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
        data = s.read()
        s.close()
        assert helpers.bcf_stream_to_text(data) == want


@pytest.mark.gpu
def test_cell_walk_ignores_header_lookalikes_in_payload(gdb, monkeypatch):
    """the cell boundaries are found on the device from candidate headers (a plausible row / column / size at a byte position) by
    pointer doubling from position 0: bytes inside a cell that look exactly like a header - here the counts of a histogram, laid
    out as (row 1, column 5000, size 64) - are candidates nothing true points at.  Same output as the oracle, whole stream and in
    windows of a few cells (the window cut runs on the device too)."""
    import numpy as np
    from test_hostsim_golden import _asa_cells
    rng = np.random.default_rng(21)
    cells = _asa_cells(rng, 3, 50, lookalike=True)
    assert cells.count(np.array([1, 0, 5000, 0, 64, 0], dtype="<i4").tobytes()) > 100
    q, pb = helpers.query_json("t0_1_2_all_asa.json", "vid_all_asa.json", {}, "load")
    want, nrec, _ = helpers.oracle_run(q, cells, partition_begin=pb)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    got = s.read()
    s.close()
    assert got == want
    monkeypatch.setenv("GDBAMD_STAGE_BUDGET_BYTES", "3000")
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    got = s.read()
    s.close()
    assert got == want


@pytest.mark.gpu
def test_malformed_cell_streams_are_refused(gdb):
    """a stream cut inside a cell, a cell size that does not lead to the next cell, and a stream that does not begin with a cell are
    errors of the device walk, as they are errors of a sequential walk"""
    import struct
    case = CASES[0]
    cells = helpers.cells_for(case[1], case[2])
    q, _ = helpers.query_json(case[1], case[2], case[3], case[5])
    for bad in (cells[:-10], cells[:16] + struct.pack("<Q", struct.unpack_from("<Q", cells, 16)[0] + 3) + cells[24:], b"\xff" * 8 + cells[8:]):
        with pytest.raises(gdb.GenomicsDBException, match="truncated|malformed"):
            s = gdb.GenomicsDBQueryStream(query_json=q, cells=bad, buffer_capacity=1 << 20)
            s.read()


@pytest.mark.gpu
@pytest.mark.parametrize("max_alt", [50, 64])
def test_bcf_high_alt_dense_region_decodes_to_the_oracle_text(gdb, tmp_path, max_alt):
    """the dense high-ALT region as BCF2: PL vectors of more than 64 elements per sample (the lanes-are-samples format), entries
    longer than an LDS slot, records whose 64-sample values do not fit the persistent image (the field-by-field path), next to
    plain records on the fast path - decoded with the tests' BCF2 reader, every record equals the oracle's text"""
    import bcf2text
    import struct
    from genomicsdb_amd import synth
    N, B, L = 150, 10_000_000, 330
    g = synth.Generator(N, B, L + 500, dense=(B + 100, 200, 50, 64))
    cells, nc = g.chunk_bytes(B + L + 500)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    q["max_diploid_alt_alleles_that_can_be_genotyped"] = max_alt
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    eb = gdb.CombineEngine(q, is_bcf=True)
    eb.stage_cells(cells)
    eb.set_reference(B, synth.reference(B, L + 4096))
    hdr_text = eb.header
    body, st = eb.run_interval(B, B + L - 1, arena_bytes=1 << 30)
    paged, sp = eb.run_interval(B, B + L - 1, arena_bytes=1 << 14)
    eb.close()
    assert st.num_records == nrec and paged == body and sp.pages > 1
    text = hdr_text[hdr_text.index(b"##"):] if hdr_text[:3] == b"BCF" else hdr_text
    h = bcf2text.Header(text.decode(errors="replace").rstrip("\x00"))
    at, lines = 0, []
    while at < len(body):
        l_shared, l_indiv = struct.unpack_from("<II", body, at)
        rec = body[at:at + 8 + l_shared + l_indiv]
        lines.append(bcf2text.record_to_text(h, rec, helpers.format_float))
        at += len(rec)
    assert ("\n".join(lines) + "\n").encode() == want


@pytest.mark.gpu
def test_compressed_fragment_file_is_inflated_on_the_device(gdb, tmp_path, monkeypatch):
    """fragment file version 3: every column section (coordinates, offsets, data) as 8 KiB DEFLATE tiles (stored / fixed-Huffman blocks, any inflate reads them -
    checked here with zlib), compressed bytes to the device, one thread inflates one tile.  Loaded whole and read window by window
    the stream equals the stream from the raw cells; the file is smaller than the uncompressed one; tiles recompressed by a default
    zlib writer (dynamic Huffman codes) inflate to the same bytes; a flipped byte inside a tile is an error, not wrong output."""
    import json
    import os
    import struct
    import zlib
    from genomicsdb_amd import synth
    N, B, L = 150, 10_000_000, 24_000
    cells, nc = _synth_cells(N, B, L)
    q = helpers.synth_query(tmp_path, N, B + 500, B + L - 700)
    monkeypatch.delenv("GDBAMD_STAGE_BUDGET_BYTES", raising=False)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    resident = s.read()
    s.close()
    ws = tmp_path / "ws"
    (ws / "arr").mkdir(parents=True)
    e = gdb.CombineEngine(q)
    e.stage_cells(cells)
    e.save_fragment(ws / "arr" / "raw.gdbamd")
    e.save_fragment(ws / "arr" / "fragment.gdbamd", compress=True)
    e.close()
    raw_size, z_size = os.path.getsize(ws / "arr" / "raw.gdbamd"), os.path.getsize(ws / "arr" / "fragment.gdbamd")
    assert z_size < 0.5 * raw_size
    # (a) loaded whole
    bodies = []
    for name in ("raw.gdbamd", "fragment.gdbamd"):
        e = gdb.CombineEngine(q)
        e.load_fragment(ws / "arr" / name)
        e.set_reference(B, synth.reference(B, L + 4096))
        body, st = e.run_interval(B + 500, B + L - 700, arena_bytes=1 << 30)
        e.close()
        bodies.append(body)
    whole = bodies[1]
    assert bodies[0] == bodies[1] and len(whole) > 0
    # (b) as the array of a workspace, in windows of a ninth of the cells
    q2 = dict(q)
    q2["workspace"] = str(ws)
    q2["array"] = "arr"
    qf = tmp_path / "query.json"
    qf.write_text(json.dumps(q2))
    os.remove(ws / "arr" / "raw.gdbamd")
    monkeypatch.setenv("GDBAMD_STAGE_BUDGET_BYTES", str(len(cells) // 9))
    s = gdb.GenomicsDBQueryStream(query_json_file=str(qf), buffer_capacity=1 << 20)
    assert s.read() == resident
    s.close()
    monkeypatch.delenv("GDBAMD_STAGE_BUDGET_BYTES")
    # (c) the tiles are plain DEFLATE: walk the header to the first data section that has tiles; its first tile inflates with zlib
    blob = bytearray((ws / "arr" / "fragment.gdbamd").read_bytes())
    assert blob[:8] == b"GDBAMDF2" and struct.unpack_from("<I", blob, 8)[0] == 3
    nfields, C, M = struct.unpack_from("<Iqq", blob, 12)
    at = 88
    fields = []
    for _ in range(nfields):
        var, es, name_len, fixed_num, data_bytes = struct.unpack_from("<BBHiQ", blob, at)
        at += 16 + name_len
        fields.append((var, data_bytes))
    nsec = 3 + 2 * nfields                                  # row, begin, end, then (offsets, data) per field
    stored = struct.unpack_from("<%dQ" % nsec, blob, at)
    at += 8 * nsec
    al = lambda x: (x + 63) & ~63
    # (section size, index into `stored`) in file order; None = the boundary markers, which stay raw
    layout = [(C * 4, 0), (C * 8, 1), (C * 8, 2), (M * 8, None)]
    for i, (var, data_bytes) in enumerate(fields):
        if var:
            layout.append(((C + 1) * 4, 3 + 2 * i))
        layout.append((data_bytes, 4 + 2 * i))
    payload_at = offs = None
    for nbytes, si in layout:
        at = al(at)
        if si is None:
            at += nbytes
            continue
        st_bytes = stored[si]
        ntiles = struct.unpack_from("<Q", blob, at + st_bytes - 8)[0]
        assert ntiles == (nbytes + 8191) // 8192
        if ntiles and si >= 3 and payload_at is None:       # the first field section: damaged in (e)
            index_at = at + st_bytes - 8 - 8 * (ntiles + 1)
            offs = struct.unpack_from("<%dQ" % (ntiles + 1), blob, index_at)
            payload_at = at
            first_len = min(8192, nbytes)
        at += st_bytes
    assert at == len(blob) and payload_at is not None
    first = zlib.decompressobj(-15).decompress(bytes(blob[payload_at + offs[0]:payload_at + offs[1]]))
    assert len(first) == first_len
    # (d) the same file with every tile recompressed by a default zlib writer (dynamic Huffman codes, as gzip'd tiles have them): the
    #     device inflate builds the code tables per tile; same bytes out
    def rebuild(blob, recompress):
        out = bytearray(blob[:88])
        at = 88
        for _ in range(nfields):
            name_len = struct.unpack_from("<H", blob, at + 2)[0]
            out += blob[at:at + 16 + name_len]
            at += 16 + name_len
        stored_pos = len(out)
        out += b"\x00" * (8 * nsec)
        at += 8 * nsec
        pad = lambda buf: buf.extend(b"\x00" * ((64 - len(buf) % 64) % 64))
        new_stored = list(stored)
        for nbytes, si in layout:
            at = al(at); pad(out)
            if si is None:
                out += blob[at:at + nbytes]; at += nbytes
                continue
            st_bytes = stored[si]
            nt = (nbytes + 8191) // 8192
            index_at = at + st_bytes - 8 - 8 * (nt + 1)
            toffs = struct.unpack_from("<%dQ" % (nt + 1), blob, index_at)
            tiles = [recompress(zlib.decompressobj(-15).decompress(bytes(blob[at + toffs[i]:at + toffs[i + 1]]))) for i in range(nt)]
            new_offs = [0]
            for z in tiles:
                out += z
                new_offs.append(new_offs[-1] + len(z))
            out += struct.pack("<%dQ" % (nt + 1), *new_offs) + struct.pack("<Q", nt)
            new_stored[si] = new_offs[-1] + 8 * (nt + 1) + 8
            at += st_bytes
        struct.pack_into("<%dQ" % nsec, out, stored_pos, *new_stored)
        return bytes(out)

    def dynamic(raw):
        co = zlib.compressobj(9, zlib.DEFLATED, -15)
        return co.compress(raw) + co.flush()
    dyn_blob = rebuild(blob, dynamic)
    assert len(dyn_blob) < len(blob)
    (ws / "arr" / "dynamic.gdbamd").write_bytes(dyn_blob)
    e = gdb.CombineEngine(q)
    e.load_fragment(ws / "arr" / "dynamic.gdbamd")
    e.set_reference(B, synth.reference(B, L + 4096))
    body, _ = e.run_interval(B + 500, B + L - 700, arena_bytes=1 << 30)
    e.close()
    assert body == whole
    # (e) a damaged tile is an error, not wrong output
    bad = bytearray(blob)
    bad[payload_at + offs[0] + (offs[1] - offs[0]) // 2] ^= 0x5A
    (ws / "arr" / "fragment.gdbamd").write_bytes(bytes(bad))
    e = gdb.CombineEngine(q)
    with pytest.raises(gdb.GenomicsDBException, match="inflate|corrupt|tile"):
        e.load_fragment(ws / "arr" / "fragment.gdbamd")
        e.set_reference(B, synth.reference(B, L + 4096))
        got, _ = e.run_interval(B + 500, B + L - 700, arena_bytes=1 << 30)
        assert got == whole          # (a flip that happens to decode to the same size must still give the same bytes)
        raise gdb.GenomicsDBException("tile: damage left the bytes unchanged")
    e.close()


@pytest.mark.gpu
def test_damaged_fragment_columns_are_refused(gdb, tmp_path):
    """The header of a fragment file is validated at open; its columns are checked on the device where they arrive: cells out of
    (begin, row) order, a row outside the query's rows, END before begin or offsets of a variable-length column that decrease are an
    error at load, not an out-of-bounds read in a later kernel."""
    import struct
    N, B, L = 40, 10_000_000, 6_000
    cells, nc = _synth_cells(N, B, L)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    e = gdb.CombineEngine(q)
    e.stage_cells(cells)
    good = tmp_path / "good.gdbamd"
    e.save_fragment(good)
    e.close()
    blob = good.read_bytes()
    assert blob[:8] == b"GDBAMDF2" and struct.unpack_from("<I", blob, 8)[0] == 2
    nfields, C, M = struct.unpack_from("<Iqq", blob, 12)
    at = 88
    fields = []
    for _ in range(nfields):
        var, es, name_len, fixed_num, data_bytes = struct.unpack_from("<BBHiQ", blob, at)
        at += 16 + name_len
        fields.append((var, data_bytes))
    al = lambda x: (x + 63) & ~63
    row_at = al(at); begin_at = al(row_at + C * 4); end_at = al(begin_at + C * 8); marker_at = al(end_at + C * 8)
    at = marker_at + M * 8
    off_at = None
    for var, data_bytes in fields:
        if var:
            at = al(at)
            if off_at is None:
                off_at = at
            at += (C + 1) * 4
        at = al(at) + data_bytes
    assert at == len(blob) and off_at is not None and C > 20

    def refused(mutate, what):
        bad = bytearray(blob)
        mutate(bad)
        path = tmp_path / "bad.gdbamd"
        path.write_bytes(bytes(bad))
        e = gdb.CombineEngine(q)
        with pytest.raises(gdb.GenomicsDBException, match="damaged|corrupt|do not match"):
            e.load_fragment(path)
        e.close()

    e = gdb.CombineEngine(q)
    e.load_fragment(good)                                   # (the undamaged file loads)
    e.close()
    refused(lambda b: struct.pack_into("<q", b, begin_at + 8 * 10, B - 5), "a cell out of order")
    refused(lambda b: struct.pack_into("<i", b, row_at + 4 * 7, N + 3), "a row outside the query")
    refused(lambda b: struct.pack_into("<i", b, row_at + 4 * 7, -2), "a negative row")
    refused(lambda b: struct.pack_into("<q", b, end_at + 8 * 12, 17), "END before begin")
    o11 = struct.unpack_from("<I", blob, off_at + 4 * 11)[0]
    refused(lambda b: struct.pack_into("<I", b, off_at + 4 * 10, o11 + 40), "offsets that decrease")


@pytest.mark.gpu
def test_host_walk_and_device_walk_stage_the_same_cells(gdb, tmp_path, monkeypatch):
    """GDBAMD_HOST_WALK=1 (the walk of the cell sizes by one host thread, kept for comparison) and the device walk cut the windows at
    the same cells: same stream from cells.bin in windows of a ninth of the file"""
    import json
    import os
    import subprocess
    import sys
    N, B, L = 120, 10_000_000, 20_000
    cells, nc = _synth_cells(N, B, L)
    q = helpers.synth_query(tmp_path, N, B + 300, B + L - 500)
    ws = tmp_path / "ws"
    (ws / "arr").mkdir(parents=True)
    (ws / "arr" / "cells.bin").write_bytes(cells)
    q2 = dict(q)
    q2["workspace"] = str(ws)
    q2["array"] = "arr"
    qf = tmp_path / "query.json"
    qf.write_text(json.dumps(q2))
    # the switch is read once per process: one child process per setting
    code = ("import sys, hashlib; sys.path.insert(0, %r); import genomicsdb_amd as g; "
            "s = g.GenomicsDBQueryStream(query_json_file=%r, buffer_capacity=1 << 20); d = s.read(); s.close(); print(len(d), hashlib.sha256(d).hexdigest())"
            % (helpers.ROOT, str(qf)))
    outs = []
    for host_walk in (False, True):
        env = dict(os.environ, GDBAMD_STAGE_BUDGET_BYTES=str(len(cells) // 9))
        env.pop("GDBAMD_HOST_WALK", None)
        if host_walk:
            env["GDBAMD_HOST_WALK"] = "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and int(outs[0].split()[0]) > 100000


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in SUPPORTED if c[0] in ("t0_1_2_loading", "t6_7_8_loading", "t0_overlapping_loading", "t0_1_2_all_asa_loading",
                                                                   "t0_haploid_triploid_1_2_3_triploid_deletion_loading", "info_ops1")],
                         ids=lambda c: c[0])
def test_bcf_stream_with_one_column_per_window(gdb, case, monkeypatch):
    """BCF2 output from an array that is streamed one begin column per window (staging budget of one byte: every live interval
    crosses a carry-over): the decoded stream is still the reference's text golden"""
    monkeypatch.setenv("GDBAMD_STAGE_BUDGET_BYTES", "1")
    name, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
    data = s.read()
    s.close()
    assert helpers.bcf_stream_to_text(data) == helpers.golden_text(golden)
