"""Several different FILTER ids in one output record: the reference writes them in the iteration order of a
std::unordered_set<int> filled with one range insert per live call (broad_combined_gvcf.cc:846-874) - a property of libstdc++,
like the tied medians.  gdb_core.hpp restates it (gdb_uset_insert_range); here the restatement is compared with the library
itself, insertion sequence by insertion sequence, and the kernel bodies / the device with the oracle (whose set IS the
library's) on records that unite two and more ids."""
import ctypes
import itertools
import random

import pytest

import helpers


def _orders(ranges):
    lib = helpers.hostsim_lib()
    lib.hostsim_uset_order.restype = ctypes.c_int
    flat = [x for r in ranges for x in r]
    ids = (ctypes.c_int32 * max(1, len(flat)))(*flat)
    lens = (ctypes.c_int32 * max(1, len(ranges)))(*[len(r) for r in ranges])
    mine, ref = (ctypes.c_int32 * 64)(), (ctypes.c_int32 * 64)()
    n = lib.hostsim_uset_order(ids, lens, len(ranges), mine, ref, 64)
    return n, list(mine[:max(n, 0)]), list(ref[:max(n, 0)])


def test_every_insertion_order_of_small_id_sets_matches_the_library():
    """all permutations of up to 6 distinct ids drawn from the range real vid mappings use (field indices 0 .. ~60), one id per
    call (single-element ranges) - the common case: calls with one FILTER each"""
    checked = 0
    for ids in ([0, 1], [1, 13], [3, 16, 29], [2, 15, 28, 41], [0, 11, 13, 26, 5], [7, 20, 33, 46, 59, 1]):
        for perm in itertools.permutations(ids):
            n, mine, ref = _orders([[x] for x in perm])
            assert n == len(ids) and mine == ref, (perm, mine, ref)
            checked += 1
    assert checked > 800


def test_random_range_inserts_match_the_library():
    """ranges of several ids per call (the hint _M_insert_range passes to the rehash policy), duplicates inside and across
    ranges, up to the 16 distinct ids the device keeps"""
    rnd = random.Random(20260929)
    seen_sizes = set()
    for case in range(4000):
        universe = rnd.sample(range(0, 90), rnd.randint(1, 16))
        ranges = []
        for _ in range(rnd.randint(1, 12)):
            k = rnd.randint(0, 5)
            ranges.append([rnd.choice(universe) for _ in range(k)])
        n, mine, ref = _orders(ranges)
        assert n >= 0 and mine == ref, (ranges, mine, ref)
        seen_sizes.add(n)
    assert max(seen_sizes) >= 14


def test_more_than_sixteen_ids_is_reported():
    n, _, _ = _orders([[i] for i in range(17)])
    assert n == -1


def _two_filter_cells(tmp_path, N=300, L=1200, seed=5):
    from genomicsdb_amd import synth
    B = 10_000_000
    g = synth.Generator(N, B, L + 2500, seed=seed, filter_permille=350, filter2_permille=350, filter_id=1, filter_id2=0, rank_sum_scale=10.0)
    cells, _ = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B + 20, B + L)
    q["produce_FILTER_field"] = True
    return B, cells, q


def test_kernel_bodies_write_multi_id_unions_like_the_oracle(tmp_path):
    B, cells, q = _two_filter_cells(tmp_path)
    want, nrec, _ = helpers.oracle_run(q, cells, with_header=False)
    both = [l for l in want.split(b"\n") if l and b";" in l.split(b"\t")[6]]
    assert len(both) >= 3                     # records that unite the two different ids exist
    got, err = helpers.hostsim_run(q, cells, with_header=False, rows_per_chunk=16, records_per_run=5)
    assert err == 0 and got == want


@pytest.mark.gpu
@pytest.mark.parametrize("n_samples", [300, 1200])
def test_device_writes_multi_id_unions_like_the_oracle(tmp_path, n_samples):
    """1 200 samples: records with more than 256 variant calls go through the workgroup-per-record site kernel, which hands a
    record with several ids back to the serial walk"""
    import genomicsdb_amd
    from genomicsdb_amd import synth
    L = 1200 if n_samples < 1000 else 300
    B, cells, q = _two_filter_cells(tmp_path, N=n_samples, L=L, seed=9)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, 9, with_header=False)
    assert sum(1 for l in want.split(b"\n") if l and b";" in l.split(b"\t")[6]) >= 3
    e = genomicsdb_amd.CombineEngine(q)
    e.stage_cells(cells)
    e.set_reference(B, synth.reference(B, L + 4096, seed=9))
    body, st = e.run_interval(B + 20, B + L, arena_bytes=1 << 22)
    e.close()
    assert st.num_records == nrec and body == want
    eb = genomicsdb_amd.CombineEngine(q, is_bcf=True)
    eb.stage_cells(cells)
    eb.set_reference(B, synth.reference(B, L + 4096, seed=9))
    bcf, _ = eb.run_interval(B + 20, B + L, arena_bytes=1 << 22)
    import struct
    import bcf2text
    h = bcf2text.Header(eb.header.decode())
    eb.close()
    at, lines = 0, []
    while at < len(bcf):
        l_shared, l_indiv = struct.unpack_from("<II", bcf, at)
        lines.append(bcf2text.record_to_text(h, bcf[at:at + 8 + l_shared + l_indiv], helpers.format_float))
        at += 8 + l_shared + l_indiv
    assert ("\n".join(lines) + "\n").encode() == want
