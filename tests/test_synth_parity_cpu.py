"""Synthetic workload (SURVEY 8(d) generator): CPU-side checks - the generator is deterministic, and the kernel bodies
(hostsim) agree with the oracle on it byte for byte."""
import pytest

import helpers


def test_generator_deterministic_and_chunk_invariant():
    from genomicsdb_amd import synth
    g1 = synth.Generator(20, 10_000_000, 6000)
    a, na = g1.chunk_bytes(10_006_000)
    g2 = synth.Generator(20, 10_000_000, 6000)
    b1, n1 = g2.chunk_bytes(10_002_000)
    b2, n2 = g2.chunk_bytes(10_006_000)
    assert na == n1 + n2 and na > 20
    assert sorted_cells(a) == sorted_cells(b1 + b2)
    assert len(a) >= 150 * na


def sorted_cells(buf):
    import struct
    out, off = [], 0
    while off < len(buf):
        sz = struct.unpack_from("<Q", buf, off + 16)[0]
        out.append(buf[off:off + sz])
        off += sz
    return sorted(out)


def test_hostsim_matches_oracle_on_synthetic(tmp_path):
    from genomicsdb_amd import synth
    N, B, L = 37, 10_000_000, 4000
    g = synth.Generator(N, B, L)
    cells, nc = g.chunk_bytes(B + L)
    q = helpers.synth_query(tmp_path, N, B + 100, B + L - 300)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED)
    assert nrec > 500
    # hostsim has no synthetic-reference hook: compare on a query whose records all start at a cell begin is not
    # possible in general, so give both the same FASTA-less setup and mask REF of 'N' records via the oracle without seed
    want_n, _, _ = helpers.oracle_run(q, cells)
    got, errbits = helpers.hostsim_run(q, cells, rows_per_chunk=16, records_per_run=7)
    assert errbits == 0
    assert got == want_n
    assert want != want_n  # the synthetic reference does change REF of mid-block records


@pytest.mark.parametrize("max_alt", [50, 64])
def test_high_alt_dense_region_kernel_bodies_match_oracle(tmp_path, max_alt):
    """BASELINE.json configs[4]-style stress at test size: 80 samples all starting an insertion from a pool of 64 alleles at
    the same positions -> ~45-55 merged alleles per hot site, PL re-indexing over ~1 500 genotypes per sample; with the default
    limit (50 ALT alleles) the widest sites drop their genotype-length fields, with 64 none does."""
    from genomicsdb_amd import synth
    N, B, L = 80, 10_000_000, 330
    g = synth.Generator(N, B, L + 500, dense=(B + 100, 200, 50, 64))
    cells, nc = g.chunk_bytes(B + L + 500)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    q["max_diploid_alt_alleles_that_can_be_genotyped"] = max_alt
    want, nrec, _ = helpers.oracle_run(q, cells, with_header=False)   # (no FASTA hook in hostsim: 'N' for mid-block REF on both sides)
    got, errbits = helpers.hostsim_run(q, cells, with_header=False, rows_per_chunk=16, records_per_run=7)
    assert errbits == 0
    assert got == want
    widest = max(len(l.split(b"\t")[4].split(b",")) for l in want.split(b"\n") if l)
    assert widest >= 40


def test_tied_zero_medians_follow_the_library_selection(tmp_path):
    """10 000 samples: records with hundreds of variant calls whose rank-sum medians are zeros of both signs - the printed sign
    ("-0" / "0") is the one std::nth_element leaves in the middle (gdb_core.hpp: gdb_nth_element_libstdcxx)"""
    from genomicsdb_amd import synth
    N, B, L = 10_000, 10_000_000, 6
    g = synth.Generator(N, B, L + 2500)
    cells, nc = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B, B + L - 1)
    want, nrec, _ = helpers.oracle_run(q, cells, with_header=False)
    assert b"=-0;" in want and b"=0;" in want
    got, errbits = helpers.hostsim_run(q, cells, with_header=False, rows_per_chunk=64, records_per_run=8)
    assert errbits == 0
    assert got == want


SMALL_GENOME = [("1", 0, 3000), ("2", 3000, 500), ("3", 3500, 4000), ("X", 7500, 1200), ("MT", 8700, 1300)]


def test_generator_genome_mode_never_crosses_a_contig_end():
    import struct
    from genomicsdb_amd import synth
    g = synth.Generator(12, 0, 10_000, contigs=SMALL_GENOME)
    cells, nc = g.chunk_bytes(10_000)
    off, begins_at = 0, {c[1]: 0 for c in SMALL_GENOME}
    while off < len(cells):
        row, col, sz, end = struct.unpack_from("<qqQq", cells, off)
        ctg = [c for c in SMALL_GENOME if c[1] <= col < c[1] + c[2]]
        assert len(ctg) == 1 and end < ctg[0][1] + ctg[0][2]
        if col in begins_at:
            begins_at[col] += 1
        off += sz
    assert all(v == 12 for v in begins_at.values())     # every sample starts anew at every contig's first column


def test_hostsim_matches_oracle_across_contig_boundaries(tmp_path):
    """BASELINE configs[3] shape at test size: the columns are a flattened genome of five contigs; the query interval
    crosses four contig boundaries.  CHROM / POS / END are contig-relative (broad_combined_gvcf.cc:772-791,903-909)."""
    from genomicsdb_amd import synth
    N = 23
    g = synth.Generator(N, 0, 10_000, contigs=SMALL_GENOME)
    cells, nc = g.chunk_bytes(10_000)
    q = helpers.synth_query(tmp_path, N, 1500, 9500, contigs=SMALL_GENOME)
    want, nrec, _ = helpers.oracle_run(q, cells, with_header=False)
    chroms = [l.split(b"\t", 1)[0] for l in want.split(b"\n") if l]
    assert [c for i, c in enumerate(chroms) if i == 0 or chroms[i - 1] != c] == [b"1", b"2", b"3", b"X", b"MT"]
    got, errbits = helpers.hostsim_run(q, cells, with_header=False, rows_per_chunk=16, records_per_run=7)
    assert errbits == 0
    assert got == want
    # the END tag of the last record of contig "2" (500 long) is contig-relative
    last2 = [l for l in want.split(b"\n") if l.startswith(b"2\t")][-1]
    assert b"END=500" in last2 or last2.split(b"\t")[1] == b"500"
    hdr, _, _ = helpers.oracle_run(q, cells)
    for name, _, ln in SMALL_GENOME:
        assert b"##contig=<ID=%s,length=%d>" % (name.encode(), ln) in hdr
