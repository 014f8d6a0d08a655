"""Cells with many alleles: the reference's allele LUTs grow as needed (lut.h:65-343); here one input cell may carry up to
GDBAMD_MAX_INPUT_ALLELES = 64 alleles (REF and <NON_REF> included) and a record up to 128 merged ones.  Three hand-made gVCFs:
a site where one sample lists 45 ALT alleles, a second sample 38 partly different ones and a third a single SNV, so that the merged
list (more than 50 ALT alleles) exceeds max_diploid_alt_alleles_that_can_be_genotyped and PL is dropped at that site
(variant_operations.cc:614-630) while AD keeps every allele; a second site stays below the limit and keeps its PL vector."""
import gzip
import json
import os
import random

import pytest

import helpers

SAMPLES = ["HG00141", "HG01958", "HG01530"]


def _header(sample):
    src = os.path.join(helpers.GOLDEN, "inputs", "vcfs", "t0.vcf.gz")
    with gzip.open(src, "rt") as f:
        lines = [l for l in f if l.startswith("##")]
    return "".join(lines) + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n" % sample


def _alts(rnd, n, pool):
    return rnd.sample(pool, n)


def _variant_line(rnd, pos, ref, alts):
    na = len(alts) + 2                       # REF + ALTs + <NON_REF>
    ng = na * (na + 1) // 2
    ad = [rnd.randint(0, 60) for _ in range(na)]
    pl = [rnd.randint(0, 3000) for _ in range(ng)]
    pl[rnd.randrange(ng)] = 0
    gt = "0/%d" % rnd.randint(1, len(alts))
    return "1\t%d\t.\t%s\t%s\t100\t.\tDP=%d;MQ=40.5\tGT:AD:DP:GQ:PL\t%s:%s:%d:%d:%s\n" % (
        pos, ref, ",".join(alts + ["<NON_REF>"]), sum(ad), gt, ",".join(map(str, ad)), sum(ad), rnd.randint(0, 99), ",".join(map(str, pl)))


def _block(pos, end):
    return "1\t%d\t.\tN\t<NON_REF>\t.\t.\tEND=%d\tGT:DP:GQ:MIN_DP:PL\t0/0:20:60:18:0,60,900\n" % (pos, end)


def make_inputs(tmp_path, seed=1):
    rnd = random.Random(seed)
    # insertions after the anchor base: every ALT distinct, none equal to REF
    pool = ["G" + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 6))) for _ in range(400)]
    pool = sorted(set(pool))
    rnd.shuffle(pool)
    a0, a1 = _alts(rnd, 45, pool[:70]), _alts(rnd, 38, pool[30:110])
    b0, b1 = _alts(rnd, 20, pool[:30]), _alts(rnd, 18, pool[10:40])
    recs = {
        SAMPLES[0]: [_block(20000, 20099), _variant_line(rnd, 20100, "G", a0), _block(20101, 20199), _variant_line(rnd, 20200, "G", b0), _block(20201, 20300)],
        SAMPLES[1]: [_block(20010, 20099), _variant_line(rnd, 20100, "G", a1), _block(20101, 20199), _variant_line(rnd, 20200, "G", b1), _block(20201, 20290)],
        SAMPLES[2]: [_block(20000, 20099), _variant_line(rnd, 20100, "G", ["GT"]), _block(20101, 20300)],
    }
    callsets = {"callsets": {}}
    for i, s in enumerate(SAMPLES):
        p = tmp_path / ("s%d.vcf.gz" % i)
        with gzip.open(p, "wt") as f:
            f.write(_header(s) + "".join(recs[s]))
        callsets["callsets"][s] = {"row_idx": i, "idx_in_file": 0, "filename": str(p)}
    cs = tmp_path / "callsets.json"
    cs.write_text(json.dumps(callsets))
    vid = os.path.join(helpers.GOLDEN, "inputs", "vid.json")
    import genomicsdb_amd
    cells, ncells = genomicsdb_amd.import_cells(vid, str(cs))
    assert ncells == 13
    q = {"vid_mapping_file": vid, "callset_mapping_file": str(cs),
         "vcf_header_filename": os.path.join(helpers.GOLDEN, "inputs", "template_vcf_header.vcf"),
         "reference_genome": os.path.join(helpers.GOLDEN, "inputs", "chr1_10MB.fasta.gz"),
         "query_column_ranges": [[[0, 1_000_000_000]]], "query_row_ranges": [{"range_list": [{"low": 0, "high": 2}]}],
         "attributes": ["REF", "ALT", "DP", "MQ", "GT", "AD", "GQ", "PL", "DP_FORMAT", "MIN_DP"], "produce_GT_field": True}
    return cells, q, (a0, a1)


def _check_oracle_text(want, a0, a1):
    lines = [l.split("\t") for l in want.decode().splitlines() if not l.startswith("#")]
    hot = [f for f in lines if f[1] == "20100"][0]
    merged = hot[4].split(",")
    n_union = len(set(a0) | set(a1) | {"GT"})
    assert merged[-1] == "<NON_REF>" and len(merged) == n_union + 1 and n_union > 50
    assert merged[:45] == a0                                  # first appearance order: the first sample's list leads
    fmt = hot[8].split(":")
    assert "PL" not in fmt and "AD" in fmt                    # more than 50 ALT alleles: genotype-length fields are dropped
    assert len(hot[9].split(":")[fmt.index("AD")].split(",")) == n_union + 2
    warm = [f for f in lines if f[1] == "20200"][0]
    fmt = warm[8].split(":")
    n = len(warm[4].split(",")) + 1
    assert 20 <= n - 2 <= 50 and len(warm[9].split(":")[fmt.index("PL")].split(",")) == n * (n + 1) // 2


def test_oracle_and_kernel_bodies_on_45_and_38_alt_alleles(tmp_path):
    cells, q, (a0, a1) = make_inputs(tmp_path)
    want, nrec, _ = helpers.oracle_run(q, cells)
    _check_oracle_text(want, a0, a1)
    got, err = helpers.hostsim_run(q, cells)
    assert err == 0 and got == want


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_on_45_and_38_alt_alleles(tmp_path, seed):
    import genomicsdb_amd
    cells, q, (a0, a1) = make_inputs(tmp_path, seed)
    want, nrec, _ = helpers.oracle_run(q, cells)
    _check_oracle_text(want, a0, a1)
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    got = s.read()
    s.close()
    assert got == want
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
    bcf = s.read()
    s.close()
    assert helpers.bcf_stream_to_text(bcf) == want
    q["max_diploid_alt_alleles_that_can_be_genotyped"] = 100   # now the hot site keeps its PL vector (> 1 500 genotypes)
    want, _, _ = helpers.oracle_run(q, cells)
    hot = [l.split("\t") for l in want.decode().splitlines() if l.startswith("1\t20100\t")][0]
    assert "PL" in hot[8].split(":")
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    got = s.read()
    s.close()
    assert got == want
