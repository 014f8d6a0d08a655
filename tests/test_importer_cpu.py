"""The product's (g)VCF importer (csrc/host/vcf_importer.cc, C ABI gdbamd_import_cells) against the test
tree's fixture importer (tests/tools/vcf2cells.py, an independent Python restatement of the same reference code) on every
callset mapping of the reference's test inputs: the cell streams must be byte-identical.  Host code only - no device."""
import os
import struct

import pytest

import helpers
from golden_cases import CASES

PAIRS = sorted({(c[1], c[2]) for c in CASES})


@pytest.fixture(scope="module")
def gdb():
    from genomicsdb_amd import build as b
    b.build_native()
    import genomicsdb_amd
    return genomicsdb_amd


def _paths(callsets, vid):
    return os.path.join(helpers.GOLDEN, "inputs", vid), os.path.join(helpers.GOLDEN, "inputs", "callsets", callsets)


@pytest.mark.parametrize("callsets,vid", PAIRS, ids=["%s-%s" % p for p in PAIRS])
def test_import_matches_fixture_importer(gdb, callsets, vid):
    v, c = _paths(callsets, vid)
    got, ncells = gdb.import_cells(v, c, file_root=helpers.GOLDEN, treat_deletions_as_intervals=True)
    want = helpers.cells_for(callsets, vid)
    assert ncells > 0 and len(got) == len(want)
    assert got == want


def test_import_column_partition_and_deletion_switch(gdb):
    v, c = _paths("t0_1_2.json", "vid.json")
    full, n_full = gdb.import_cells(v, c, file_root=helpers.GOLDEN)
    cut = 12200
    lo, n_lo = gdb.import_cells(v, c, file_root=helpers.GOLDEN, column_begin=0, column_end=cut - 1)
    hi, n_hi = gdb.import_cells(v, c, file_root=helpers.GOLDEN, column_begin=cut)
    # a cell belongs to the partition of its begin column - and an interval that begins before a partition and reaches into it is
    # handed to that partition as well, first, at its own coordinates (load_operators.cc:33-79)
    def cells_of(buf):
        out, off = [], 0
        while off < len(buf):
            row, col, size, end = struct.unpack_from("<qqQq", buf, off)
            out.append((row, col, end, buf[off:off + size]))
            off += size
        return out
    all_cells = cells_of(full)
    want_spanning = {}
    for row, col, end, raw in all_cells:                      # per row the latest cell at or before the cut decides
        if col <= cut:
            want_spanning[row] = (row, col, end, raw)
    want_spanning = sorted((c for c in want_spanning.values() if c[1] < cut and c[2] >= cut), key=lambda c: (c[1], c[0]))
    assert want_spanning, "the fixture has reference blocks across column %d" % cut
    assert lo == b"".join(c[3] for c in all_cells if c[1] < cut)
    assert hi == b"".join(c[3] for c in want_spanning) + b"".join(c[3] for c in all_cells if c[1] >= cut)
    assert n_lo + n_hi == n_full + len(want_spanning)
    # deletions as intervals: END of a deletion cell = begin + len(REF) - 1, only when the switch is on
    def ends(buf):
        out, off = [], 0
        while off < len(buf):
            row, col, size, end = struct.unpack_from("<qqQq", buf, off)
            out.append((col, end))
            off += size
        return out
    v2, c2 = _paths("min_PL_spanning_deletion.json", "vid_phased_GT.json")
    on, _ = gdb.import_cells(v2, c2, file_root=helpers.GOLDEN, treat_deletions_as_intervals=True)
    off_, _ = gdb.import_cells(v2, c2, file_root=helpers.GOLDEN, treat_deletions_as_intervals=False)
    assert any(e > b for b, e in ends(on))
    assert [b for b, _ in ends(on)] == [b for b, _ in ends(off_)]
    assert sum(e - b for b, e in ends(on)) > sum(e - b for b, e in ends(off_))


def test_import_errors_are_loud(gdb, tmp_path):
    v, c = _paths("t0_1_2.json", "vid.json")
    with pytest.raises(gdb.GenomicsDBException, match="cannot open"):
        gdb.import_cells(v, c, file_root=str(tmp_path))

