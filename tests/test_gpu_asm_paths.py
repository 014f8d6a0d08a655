"""The three ways an interval's sample columns are sized and assembled (GDBAMD_ASM_PATH, gdb_pipeline.hip):
0 = k_assemble_size + k_assemble_write (default), 1 = k_size2 + matrix-free k_write2, 2 = k_size2 + k_fill2 + k_assemble_write.
Every path has to give the reference's bytes: goldens, untabled record types, paging down to one record per page, the per-page
matrix, BCF2, overflow texts (10 000 samples), and the same bytes as path 0 on a 1 000-sample window."""
import hashlib

import pytest

import helpers
from golden_cases import CASES

pytestmark = pytest.mark.gpu

GOLDENS = ("t0_1_2_vcf_at_0", "t6_7_8_vcf_at_0", "t0_overlapping_at_12202", "t0_1_2_combined_at_12150", "t0_with_missing_PL_SB_fields",
           "min_PL_spanning_deletion", "t0_haploid_triploid_1_2_3_triploid_deletion_vcf_at_0")


@pytest.fixture()
def gdb():
    import genomicsdb_amd
    return genomicsdb_amd


def _golden(gdb, name, **kw):
    case = [c for c in CASES if c[0].startswith(name)]
    if not case:
        return None, None
    _, callsets, vid, ov, golden, mode = case[0]
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 16, **kw)
    got = s.read()
    s.close()
    return got, helpers.golden_text(golden)


@pytest.mark.parametrize("path", ["1", "2", "3"])
def test_goldens_on_every_path(gdb, monkeypatch, path):
    monkeypatch.setenv("GDBAMD_ASM_PATH", path)
    seen = 0
    for name in GOLDENS:
        got, want = _golden(gdb, name)
        if got is None:
            continue
        seen += 1
        assert got == want, name
    assert seen >= 4


@pytest.mark.parametrize("path", ["1", "2", "3"])
@pytest.mark.parametrize("max_types", ["0", "2"])
def test_untabled_types_and_tiny_pages(gdb, tmp_path, monkeypatch, path, max_types):
    """record types without table slots take one slot per (record, sample); a page per record restarts every walker at every record"""
    from genomicsdb_amd import synth
    monkeypatch.setenv("GDBAMD_ASM_PATH", path)
    monkeypatch.setenv("GDBAMD_MAX_TYPES", max_types)
    N, B, L = 100, 10_000_000, 3000
    g = synth.Generator(N, B, L + 2500)
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B + 500, B + 500 + L - 1)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 4096))
    got, st = eng.run_interval(B + 500, B + 500 + L - 1, arena_bytes=1 << 20)
    assert st.num_records == nrec and got == want
    got, st = eng.run_interval(B + 500, B + 500 + L - 1, arena_bytes=1)          # pages of the largest record's size: one or two records each
    assert st.pages > nrec // 2 and got == want
    monkeypatch.setenv("GDBAMD_RESOLVED_MB", "0")                                 # (path 2: the matrix page by page)
    got, st = eng.run_interval(B + 500, B + 500 + L - 1, arena_bytes=1 << 16)
    assert st.pages > 5 and got == want
    eng.close()


def test_paths_agree_at_1000_samples(gdb, tmp_path, monkeypatch):
    """c2's width: 1 000 samples x 30 kb (1.3 GB of text) - all three paths, two pagings each, one hash; the first 1 200 columns against the oracle"""
    from genomicsdb_amd import synth
    N, B, L = 1000, 10_000_000, 30_000
    g = synth.Generator(N, B, L + 2500)
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 2500 + 4096))
    hashes = {}
    for path in ("0", "1", "2", "3"):
        monkeypatch.setenv("GDBAMD_ASM_PATH", path)
        for arena in (4 << 30, 64 << 20):
            got, st = eng.run_interval(B, B + L - 1, arena_bytes=arena)
            hashes[(path, arena)] = (hashlib.sha256(got).hexdigest(), st.num_records, len(got))
        if path == "1":
            q2 = helpers.synth_query(tmp_path, N, B, B + 1199)
            want, nrec, _ = helpers.oracle_run_synth(q2, cells, synth.SEED, with_header=False)
            head, st = eng.run_interval(B, B + 1199, arena_bytes=1 << 30)
            assert st.num_records == nrec and head == want
    assert len(set(hashes.values())) == 1, hashes
    eng.close()


def test_compact_and_wide_resolved_matrix_give_the_same_bytes(gdb, tmp_path, monkeypatch):
    """The (record, sample) matrix of the default path in its two layouts: compact (u32 offsets + u8 lengths, 5 bytes per pair; chosen
    per interval when no entry text is longer than 255 bytes - c2's width) and wide (8 bytes; GDBAMD_RES_COMPACT=0, or forced by a
    long entry).  Same bytes, whole-interval and page-by-page matrix alike; the first 1 200 columns against the oracle."""
    from genomicsdb_amd import synth
    N, B, L = 1000, 10_000_000, 30_000
    g = synth.Generator(N, B, L + 2500)
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 2500 + 4096))
    seen = {}
    import os
    check_mode = os.environ.get("GDBAMD_SIZE3_CHECK", "0") not in ("", "0")     # (the word-for-word check of the sizing kernels keeps the wide layout)
    for compact, want_bytes in (("1", 8 if check_mode else 5), ("0", 8)):
        monkeypatch.setenv("GDBAMD_RES_COMPACT", compact)
        for budget in (None, "0"):
            if budget is None:
                monkeypatch.delenv("GDBAMD_RESOLVED_MB", raising=False)
            else:
                monkeypatch.setenv("GDBAMD_RESOLVED_MB", budget)
            got, st = eng.run_interval(B, B + L - 1, arena_bytes=64 << 20)
            assert st.resolved_entry_bytes == want_bytes and st.pages > 5
            seen[(compact, budget)] = (hashlib.sha256(got).hexdigest(), st.num_records, len(got))
    assert len(set(seen.values())) == 1, seen
    monkeypatch.setenv("GDBAMD_RES_COMPACT", "1")
    monkeypatch.delenv("GDBAMD_RESOLVED_MB", raising=False)
    q2 = helpers.synth_query(tmp_path, N, B, B + 1199)
    want, nrec, _ = helpers.oracle_run_synth(q2, cells, synth.SEED, with_header=False)
    head, st = eng.run_interval(B, B + 1199, arena_bytes=1 << 30)
    assert st.num_records == nrec and head == want and st.resolved_entry_bytes == (8 if check_mode else 5)
    eng.close()


def test_an_entry_longer_than_255_bytes_takes_the_wide_matrix(gdb, tmp_path, monkeypatch):
    """a dense high-ALT region (entries of several KB) in front of plain columns: the interval with the long entries falls back to the
    8-byte layout by itself, a later interval without them is compact again"""
    from genomicsdb_amd import synth
    N, B, L = 150, 10_000_000, 1500
    g = synth.Generator(N, B, L + 2500, dense=(B + 100, 200, 50, 64))
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    q["max_diploid_alt_alleles_that_can_be_genotyped"] = 64
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    assert max(len(c) for l in want.split(b"\n") if l for c in l.split(b"\t")[9:]) > 255
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 2500 + 4096))
    got, st = eng.run_interval(B, B + L - 1, arena_bytes=1 << 20)
    assert st.resolved_entry_bytes == 8
    assert st.num_records == nrec and got == want
    q2 = helpers.synth_query(tmp_path, N, B + 700, B + L - 1)
    q2["max_diploid_alt_alleles_that_can_be_genotyped"] = 64
    want2, nrec2, _ = helpers.oracle_run_synth(q2, cells, synth.SEED, with_header=False)
    got2, st2 = eng.run_interval(B + 700, B + L - 1, arena_bytes=1 << 20)
    assert max(len(c) for l in want2.split(b"\n") if l for c in l.split(b"\t")[9:]) <= 255
    import os
    assert st2.resolved_entry_bytes == (8 if os.environ.get("GDBAMD_SIZE3_CHECK", "0") not in ("", "0") else 5) and st2.num_records == nrec2 and got2 == want2
    eng.close()


@pytest.mark.parametrize("path", ["1", "2", "3"])
def test_overflow_texts_at_10000_samples(gdb, tmp_path, monkeypatch, path):
    """10 000 samples: most variant entries are longer than an inline slot (overflow pool, texts longer than the registers hold)"""
    from genomicsdb_amd import synth
    monkeypatch.setenv("GDBAMD_ASM_PATH", path)
    N, B, L = 10_000, 10_000_000, 260
    g = synth.Generator(N, B, L + 2500)
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    eng = gdb.CombineEngine(q)
    eng.stage_cells(cells)
    eng.set_reference(B, synth.reference(B, L + 2500 + 4096))
    got, st = eng.run_interval(B, B + L - 1, arena_bytes=64 << 20)
    assert st.num_records == nrec and got == want
    eng.close()


def test_bcf_through_the_piece_walker_matrix(gdb, monkeypatch):
    """BCF2 pages read the (record, sample) matrix: path 2 fills it with k_fill2 (path 1 has no matrix: BCF falls back to 2)"""
    for path in ("1", "2"):
        monkeypatch.setenv("GDBAMD_ASM_PATH", path)
        for name in ("t0_1_2_vcf_at_0", "t6_7_8_vcf_at_0", "t0_haploid_triploid_1_2_3_triploid_deletion_vcf_at_0"):
            case = [c for c in CASES if c[0].startswith(name)]
            if not case:
                continue
            _, callsets, vid, ov, golden, mode = case[0]
            cells = helpers.cells_for(callsets, vid)
            q, pb = helpers.query_json(callsets, vid, ov, mode)
            s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
            raw = s.read()
            s.close()
            assert raw[:5] == b"BCF\x02\x02" and helpers.bcf_stream_to_text(raw) == helpers.golden_text(golden)
