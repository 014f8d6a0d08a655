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
