"""sum / mean / median over an INFO field whose length depends on the alleles.

handle_VCF_field_combine_operation hands the scalar reducers the REMAPPED variant for allele-dependent fields
(broad_combined_gvcf.cc:386-390) and they read element 0 of every call's vector (get_valid_sum / get_valid_median,
variant_field_handler.cc:529-607).  For an A-length field (one value per ALT allele) element 0 of the remapped vector is the
call's value for the FIRST MERGED ALT allele: its own allele of that name, else its <NON_REF>, else missing - not the first value
the call stored.  Hand derivation on the reference's fixtures, record 1:17385, merged alleles G -> A, T, <NON_REF> (golden):

  AVS (A-length, two values per call: own ALT, <NON_REF>) = 1,2 (t0: ALT A)   10,20 (t1: ALT T)   100,200 (t2: ALT A)
  first merged ALT = A:   t0 -> 1,   t1 has no A -> its <NON_REF> value 20,   t2 -> 100
  sum = 121;  median = the element of rank 3 / 2 = 1 of {1, 20, 100} = 20;  mean = 121 / 3u = 40 (integer division)
  (the values as stored would give 111, 10 and 37)

Element 0 of an R-length vector is the REF allele's value in both orders: RVS = 5,x,y in every call -> sum 15, median 5, mean 5.
"""
import gzip
import json
import os

import pytest

import helpers

VALUES = {"t0_asa": [1, 2], "t1_asa": [10, 20], "t2_asa": [100, 200]}
WANT = {("A", "sum"): 121, ("A", "median"): 20, ("A", "mean"): 40, ("R", "sum"): 15, ("R", "median"): 5, ("R", "mean"): 5}
CONFIGS = sorted(WANT)


def _make_inputs(tmp_path, length, op):
    src_vid = json.load(open(os.path.join(helpers.GOLDEN, "inputs", "vid_all_asa.json")))
    fields = {}
    for k, v in src_vid["fields"].items():
        fields[k] = v
        if k == "MQ0":
            fields["AVS"] = {"vcf_field_class": ["INFO"], "type": "int", "length": length, "VCF_field_combine_operation": op}
    vp = tmp_path / "vid.json"
    vp.write_text(json.dumps(dict(src_vid, fields=fields)))
    callsets = {"callsets": {}}
    for row, (name, sample) in enumerate([("t0_asa", "HG00141"), ("t1_asa", "HG01958"), ("t2_asa", "HG01530")]):
        with gzip.open(os.path.join(helpers.GOLDEN, "inputs", "vcfs", name + ".vcf.gz"), "rt") as f:
            lines = f.read().splitlines()
        out = []
        for l in lines:
            if l.startswith("#CHROM"):
                out.append('##INFO=<ID=AVS,Number=%s,Type=Integer,Description="x">' % length)
            if not l.startswith("#") and l.split("\t")[1] == "17385":
                c = l.split("\t")
                assert len(c[4].split(",")) == 2 and c[4].endswith("<NON_REF>")
                vals = VALUES[name] if length == "A" else [5] + VALUES[name]
                c[7] += ";AVS=" + ",".join(str(x) for x in vals)
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


def _expected_body(length, op):
    body = [l for l in helpers.golden_text("t0_1_2_all_asa_loading").decode().splitlines() if not l.startswith("#")]
    out = []
    for l in body:
        c = l.split("\t")
        if c[1] == "17385":
            assert c[4] == "A,T,<NON_REF>" and "MQ0=3;" in c[7]
            c[7] = c[7].replace("MQ0=3;", "MQ0=3;AVS=%d;" % WANT[(length, op)])
        out.append("\t".join(c))
    return ("\n".join(out) + "\n").encode()


@pytest.mark.parametrize("length,op", CONFIGS)
def test_scalar_reducers_read_the_remapped_vector_oracle_and_kernel_bodies(tmp_path, length, op):
    cells, q = _make_inputs(tmp_path, length, op)
    want = _expected_body(length, op)
    txt, nrec, _ = helpers.oracle_run(q, cells, with_header=False)
    assert nrec == 4 and txt == want
    got, err = helpers.hostsim_run(q, cells, with_header=False)
    assert err == 0 and got == want


@pytest.mark.gpu
@pytest.mark.parametrize("length,op", CONFIGS)
def test_scalar_reducers_read_the_remapped_vector_device(tmp_path, length, op):
    import genomicsdb_amd
    cells, q = _make_inputs(tmp_path, length, op)
    want = _expected_body(length, op)
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
