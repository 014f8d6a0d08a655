"""INFO vectors of genotype length (`"length": "G"`) combined with element_wise_sum.

The reference remaps every allele-dependent field of a call into the merged alleles' order before the INFO combiners run
(GA4GHOperator::operate, variant_operations.cc:572-695; remap_data_based_on_genotype, variant_field_handler.cc:134-297), so the
combiners add vectors in MERGED-genotype order.  The expected bytes of the first test are derived by hand:

record 1:17385 of the t{0,1,2}_asa fixtures, merged alleles G -> A, T, <NON_REF> (golden), all three calls diploid.
Merged genotypes in VCF order (k outer, j <= k inner): 00 01 11 02 12 22 03 13 23 33.
  t0 and t2 carry G, A, <NON_REF>: merged 0 -> input 0, 1 (A) -> 1, 2 (T: not in the call) -> its <NON_REF> = 2, 3 -> 2; with
  bcf_alleles2gt(j, k) = k(k+1)/2 + j the ten merged genotypes read the inputs 0 1 2 3 4 5 3 4 5 5;
  t1 carries G, T, <NON_REF>: merged 0 -> 0, 1 (A) -> 2, 2 (T) -> 1, 3 -> 2: inputs 0 3 5 1 4 2 3 5 4 5.
With GLS = 1..6 (t0), 10..60 (t1), 100..600 (t2):
  sum = 1+10+100, 2+40+200, 3+60+300, 4+20+400, 5+50+500, 6+30+600, 4+40+400, 5+60+500, 6+50+600, 6+60+600
(`concatenate` is defined for variable-length fields only - the vid mapper refuses it for a G-length field, vid_mapper.cc - so
element_wise_sum is the one vector combiner such a field can have.)
"""
import gzip
import json
import math
import os

import pytest

import helpers

GOLDEN_CASE = "t0_1_2_all_asa_loading"
VALUES = {"t0_asa": [1, 2, 3, 4, 5, 6], "t1_asa": [10, 20, 30, 40, 50, 60], "t2_asa": [100, 200, 300, 400, 500, 600]}
READS = {"t0_asa": [0, 1, 2, 3, 4, 5, 3, 4, 5, 5], "t1_asa": [0, 3, 5, 1, 4, 2, 3, 5, 4, 5], "t2_asa": [0, 1, 2, 3, 4, 5, 3, 4, 5, 5]}
WANT_SUM = "GLS=111,242,363,424,555,636,444,565,656,666"


def test_the_hand_derived_sum_follows_from_the_reads():
    sums = [sum(VALUES[n][READS[n][g]] for n in VALUES) for g in range(10)]
    assert "GLS=" + ",".join(str(v) for v in sums) == WANT_SUM


def _make_inputs(tmp_path):
    src_vid = json.load(open(os.path.join(helpers.GOLDEN, "inputs", "vid_all_asa.json")))
    fields = {}
    for k, v in src_vid["fields"].items():
        fields[k] = v
        if k == "MQ0":
            fields["GLS"] = {"vcf_field_class": ["INFO"], "type": "int", "length": "G", "VCF_field_combine_operation": "element_wise_sum"}
    vid = dict(src_vid, fields=fields)
    vp = tmp_path / "vid.json"
    vp.write_text(json.dumps(vid))
    callsets = {"callsets": {}}
    for row, (name, sample) in enumerate([("t0_asa", "HG00141"), ("t1_asa", "HG01958"), ("t2_asa", "HG01530")]):
        with gzip.open(os.path.join(helpers.GOLDEN, "inputs", "vcfs", name + ".vcf.gz"), "rt") as f:
            lines = f.read().splitlines()
        out = []
        for l in lines:
            if l.startswith("#CHROM"):
                out.append('##INFO=<ID=GLS,Number=G,Type=Integer,Description="x">')
            if not l.startswith("#") and l.split("\t")[1] == "17385":
                c = l.split("\t")
                assert len(c[4].split(",")) == 2 and c[4].endswith("<NON_REF>") and "/" in c[9].split(":")[0]
                v = ",".join(str(x) for x in VALUES[name])
                c[7] += ";GLS=%s" % v
                l = "\t".join(c)
            out.append(l)
        p = tmp_path / (name + ".vcf.gz")
        with gzip.open(p, "wt") as f:
            f.write("\n".join(out) + "\n")
        callsets["callsets"][sample] = {"row_idx": row, "idx_in_file": 0, "filename": str(p)}
    cp = tmp_path / "callsets.json"
    cp.write_text(json.dumps(callsets))
    import vcf2cells
    cells = b"".join(c[3] for c in vcf2cells.build_cells(str(cp), str(vp), lambda fn: fn))
    q = {"vid_mapping_file": str(vp), "callset_mapping_file": str(cp),
         "vcf_header_filename": os.path.join(helpers.GOLDEN, "inputs", "template_vcf_header.vcf"),
         "reference_genome": os.path.join(helpers.GOLDEN, "inputs", "chr1_10MB.fasta.gz"),
         "query_column_ranges": [[[0, helpers.INT64_MAX - 1]]]}
    return cells, q


def _expected_body():
    body = [l for l in helpers.golden_text(GOLDEN_CASE).decode().splitlines() if not l.startswith("#")]
    out = []
    for l in body:
        c = l.split("\t")
        if c[1] == "17385":
            assert c[4] == "A,T,<NON_REF>" and "MQ0=3;" in c[7]
            c[7] = c[7].replace("MQ0=3;", "MQ0=3;%s;" % WANT_SUM)
        out.append("\t".join(c))
    return ("\n".join(out) + "\n").encode()


def test_genotype_length_info_vectors_oracle_and_kernel_bodies(tmp_path):
    cells, q = _make_inputs(tmp_path)
    want = _expected_body()
    txt, nrec, _ = helpers.oracle_run(q, cells, with_header=False)
    assert nrec == 4 and txt == want
    got, err = helpers.hostsim_run(q, cells, with_header=False)
    assert err == 0 and got == want


def test_above_the_alt_allele_limit_the_vector_is_dropped_like_PL(tmp_path):
    """handle_VCF_field_combine_operation returns early for a genotype-length field when the record has more ALT alleles than
    max_diploid_alt_alleles_that_can_be_genotyped (broad_combined_gvcf.cc:380-385), the same rule that drops PL from FORMAT: with a
    limit of 2 the record at 1:17385 (3 merged ALT alleles) loses GLS and PL, the other INFO fields stay"""
    cells, q = _make_inputs(tmp_path)
    q["max_diploid_alt_alleles_that_can_be_genotyped"] = 2
    txt, nrec, _ = helpers.oracle_run(q, cells, with_header=False)
    rec = [l for l in txt.decode().splitlines() if l.split("\t")[1] == "17385"][0].split("\t")
    assert rec[4] == "A,T,<NON_REF>" and "GLS=" not in rec[7] and "MQ0=3" in rec[7] and "PL" not in rec[8].split(":")
    got, err = helpers.hostsim_run(q, cells, with_header=False)
    assert err == 0 and got == txt


@pytest.mark.gpu
def test_genotype_length_info_vectors_device(tmp_path):
    import genomicsdb_amd
    cells, q = _make_inputs(tmp_path)
    want = _expected_body()
    hdr_and_body, _, _ = helpers.oracle_run(q, cells)
    assert hdr_and_body.endswith(want)
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    got = s.read()
    s.close()
    assert got == hdr_and_body
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
    bcf = s.read()
    s.close()
    assert helpers.bcf_stream_to_text(bcf) == hdr_and_body


# ---- haploid and triploid calls: every record of the reference's ploidy fixtures gets a G-length INFO vector -----------------------
def _ploidy_inputs(tmp_path, ops):
    """the fixtures behind golden t0_haploid_triploid_1_2_3_triploid_deletion with an INFO vector GV of genotype length in every
    record: C(ploidy + alleles - 1, alleles - 1) values, different per sample and genotype"""
    from golden_cases import CASES, HT
    name, callsets, vid_name, ov, golden, mode = [c for c in CASES if c[0] == HT + "_loading"][0]
    src_vid = json.load(open(os.path.join(helpers.GOLDEN, "inputs", vid_name)))
    fields = {}
    for k, v in src_vid["fields"].items():
        fields[k] = v
        if k == "MQ0":
            for op_name, op in ops.items():
                fields[op_name] = {"vcf_field_class": ["INFO"], "type": "int", "length": "G", "VCF_field_combine_operation": op}
    vp = tmp_path / "vid.json"
    vp.write_text(json.dumps(dict(src_vid, fields=fields)))
    cs = json.load(open(os.path.join(helpers.GOLDEN, "inputs", "callsets", callsets)))
    out_cs = {"callsets": {}}
    for si, (sample, info) in enumerate(sorted(cs["callsets"].items(), key=lambda kv: kv[1]["row_idx"])):
        with gzip.open(os.path.join(helpers.GOLDEN, info["filename"]), "rt") as f:
            lines = f.read().splitlines()
        out = []
        for l in lines:
            if l.startswith("#CHROM"):
                for op_name in ops:
                    out.append('##INFO=<ID=%s,Number=G,Type=Integer,Description="x">' % op_name)
            if not l.startswith("#"):
                c = l.split("\t")
                nal = 1 + len(c[4].split(","))
                gt = c[9].split(":")[0].replace("|", "/").split("/")
                ng = math.comb(len(gt) + nal - 1, nal - 1)
                vals = ",".join(str((si + 1) * 1000 + g) for g in range(ng))
                extra = ";".join("%s=%s" % (op_name, vals) for op_name in ops)
                c[7] = extra if c[7] in (".", "") else c[7] + ";" + extra
                l = "\t".join(c)
            out.append(l)
        p = tmp_path / ("s%d.vcf.gz" % si)
        with gzip.open(p, "wt") as f:
            f.write("\n".join(out) + "\n")
        out_cs["callsets"][sample] = dict(info, filename=str(p))
    cp = tmp_path / "callsets.json"
    cp.write_text(json.dumps(out_cs))
    import vcf2cells
    cells = b"".join(c[3] for c in vcf2cells.build_cells(str(cp), str(vp), lambda fn: fn))
    q, _ = helpers.query_json(callsets, vid_name, ov, mode)
    q["vid_mapping_file"] = str(vp)
    q["callset_mapping_file"] = str(cp)
    return cells, q


OPS = {"GVS": "element_wise_sum"}


def test_haploid_and_triploid_genotype_vectors_oracle_and_kernel_bodies(tmp_path):
    cells, q = _ploidy_inputs(tmp_path, OPS)
    txt, nrec, _ = helpers.oracle_run(q, cells, with_header=False)
    assert nrec > 0 and txt.count(b"GVS=") >= 3
    got, err = helpers.hostsim_run(q, cells, with_header=False)
    assert err == 0 and got == txt


@pytest.mark.gpu
def test_haploid_and_triploid_genotype_vectors_device(tmp_path):
    import genomicsdb_amd
    cells, q = _ploidy_inputs(tmp_path, OPS)
    want, _, _ = helpers.oracle_run(q, cells)
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    got = s.read()
    s.close()
    assert got == want
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
    bcf = s.read()
    s.close()
    assert helpers.bcf_stream_to_text(bcf) == want
