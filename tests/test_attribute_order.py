"""Query attribute order: the reference puts END, REF, ALT first BY PAIR-WISE SWAPS (variant_query_config.cc:161-185), so a query
list that does not already start END, REF, ALT displaces other fields, and with them the INFO / FORMAT emission order (the field
lists are built in query order, broad_combined_gvcf.cc:163-236; htslib moves GT to the front of FORMAT, INFO DP comes last
because it is written after the FORMAT pass, :603-727).  The reference's own tests always pass vcf_attributes_order, which
needs no swap - these tests cover the permuting path: a known answer derived by hand from the reference's code, the kernel
bodies (hostsim) against the oracle on shuffled lists, and (GPU) the device against the oracle."""
import random

import pytest

import helpers
from golden_cases import CASES, VCF_ATTRIBUTES_ORDER


def _t012():
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    name, callsets, vid, ov, golden, mode = case
    return helpers.cells_for(callsets, vid), helpers.query_json(callsets, vid, ov, mode)[0]


def swap_order(attrs, has_G_length=True):
    """the reference's rule, restated independently of both implementations: add END, ALT, REF (in that order) and GT if missing,
    then bring END, REF, ALT to positions 0, 1, 2 by swapping each with whatever stands there"""
    out = list(attrs)
    for n in ("END", "ALT", "REF"):
        if n not in out:
            out.append(n)
    if has_G_length and "GT" not in out:
        out.append("GT")
    for dst, n in enumerate(("END", "REF", "ALT")):
        src = out.index(n)
        if src > dst:
            out[src], out[dst] = out[dst], out[src]
    return out


def test_known_answer_for_a_displacing_list():
    """attributes = GQ MQ REF DP ALT BaseQRankSum PL END AD.  By hand: GT is appended (PL is genotype-length); END <-> GQ, REF <->
    MQ, ALT <-> MQ give END REF ALT DP MQ BaseQRankSum PL GQ AD GT.  INFO: MQ, BaseQRankSum in that order, DP last (= 120 + 76:
    sample 0 has no INFO DP and neither MIN_DP nor DP_FORMAT is queried); FORMAT: GT (moved first), PL, GQ, AD; no FORMAT DP."""
    cells, q = _t012()
    q["attributes"] = ["GQ", "MQ", "REF", "DP", "ALT", "BaseQRankSum", "PL", "END", "AD"]
    assert swap_order(q["attributes"]) == ["END", "REF", "ALT", "DP", "MQ", "BaseQRankSum", "PL", "GQ", "AD", "GT"]
    body, nrec, _ = helpers.oracle_run(q, cells, with_header=False)
    lines = body.decode().splitlines()
    assert nrec == 4
    f = lines[3].split("\t")
    assert f[:5] == ["1", "17385", ".", "G", "A,T,<NON_REF>"]
    assert f[7] == "MQ=31.72;BaseQRankSum=-2.074;DP=196"
    assert f[8] == "GT:PL:GQ:AD"
    assert f[9] == "./.:504,0,9807,678,1870,2548,678,1870,2548,2548:99:58,22,17,17"
    assert lines[0].split("\t")[7:10] == ["END=12144", "GT:PL:GQ", "./.:0,0,0:0"]
    got, err = helpers.hostsim_run(q, cells, with_header=False)
    assert err == 0 and got == body


def _shuffled(rnd):
    pool = [a for a in VCF_ATTRIBUTES_ORDER]
    rnd.shuffle(pool)
    k = rnd.randint(3, len(pool))
    return pool[:k]


@pytest.mark.parametrize("seed", range(12))
def test_kernel_bodies_follow_the_swap_order(tmp_path, seed):
    from genomicsdb_amd import synth
    rnd = random.Random(4200 + seed)
    attrs = _shuffled(rnd)
    cells, q = _t012()
    q["attributes"] = attrs
    want, _, _ = helpers.oracle_run(q, cells)
    got, err = helpers.hostsim_run(q, cells)
    assert err == 0 and got == want
    # INFO keys / FORMAT keys of the variant record follow the independently restated order
    order = swap_order(attrs, has_G_length="PL" in attrs)
    rec = [l for l in want.decode().splitlines() if l.startswith("1\t17385\t")][0].split("\t")
    info_keys = [kv.split("=")[0] for kv in rec[7].split(";") if kv != "."]
    info_fields = ["BaseQRankSum", "ClippingRankSum", "MQRankSum", "ReadPosRankSum", "MQ", "RAW_MQ", "MQ0"]
    assert [k for k in info_keys if k in info_fields] == [a for a in order if a in info_fields and a in info_keys]
    if "DP" in info_keys:
        assert info_keys[-1] == "DP"
    if len(rec) > 8:
        fmt = rec[8].split(":")
        fmt_fields = ["GQ", "SB", "AD", "PL", "PGT", "PID", "MIN_DP"]
        assert [k for k in fmt if k in fmt_fields] == [a for a in order if a in fmt_fields and a in fmt]
        if "GT" in fmt:
            assert fmt[0] == "GT"
        if "DP" in fmt:
            assert fmt[-1] == "DP"
    # synthetic input with the same list
    N, B, L = 29, 10_000_000, 2500
    g = synth.Generator(N, B, L + 2500, seed=seed + 1)
    sc, _ = g.chunk_bytes(B + L + 2500)
    sq = helpers.synth_query(tmp_path, N, B + 50, B + L)
    sq["attributes"] = attrs
    want, nrec, _ = helpers.oracle_run(sq, sc, with_header=False)
    got, err = helpers.hostsim_run(sq, sc, with_header=False, rows_per_chunk=16, records_per_run=5)
    assert err == 0 and nrec > 100 and got == want


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_device_follows_the_swap_order(tmp_path, seed):
    import genomicsdb_amd
    from genomicsdb_amd import synth
    rnd = random.Random(9100 + seed)
    attrs = _shuffled(rnd)
    cells, q = _t012()
    q["attributes"] = attrs
    want, _, _ = helpers.oracle_run(q, cells)
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    got = s.read()
    s.close()
    assert got == want
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
    bcf = s.read()
    s.close()
    assert helpers.bcf_stream_to_text(bcf) == want
    N, B, L = 333, 10_000_000, 3000
    g = synth.Generator(N, B, L + 2500, seed=seed + 1)
    sc, _ = g.chunk_bytes(B + L + 2500)
    sq = helpers.synth_query(tmp_path, N, B + 50, B + L)
    sq["attributes"] = attrs
    want, nrec, _ = helpers.oracle_run_synth(sq, sc, seed + 1, with_header=False)
    e = genomicsdb_amd.CombineEngine(sq)
    e.stage_cells(sc)
    e.set_reference(B, synth.reference(B, L + 4096, seed=seed + 1))
    body, st = e.run_interval(B + 50, B + L, arena_bytes=1 << 22)
    e.close()
    assert st.num_records == nrec and body == want
