"""Known answers for behaviour the reference inherits from C++ itself rather than from its own logic - the places where a
restatement can agree with itself and still differ from the reference (DESIGN 2, table "C++ semantics the reference relies on").
The expected bytes below are derived BY HAND from the reference's sources; the oracle is one of the parties checked, not the
source of the expectation.

1. `int` mean.  VariantFieldHandler<int>::get_valid_mean ends in `*result_ptr = (*result_ptr) / num_valid_elements` with an
   `int` sum and an `unsigned` count (variant_field_handler.cc:596-607).  The usual arithmetic conversions turn the sum into an
   unsigned: (-6) / 3u = 4294967290 / 3 = 1431655763 (integer division), stored back into the int.  The reference prints
   `NEG=1431655763`, not `NEG=-2`.
2. Several histogram_sum fields.  handle_INFO_fields iterates `std::unordered_map<unsigned, ...> m_INFO_histogram_field_map`
   (broad_combined_gvcf.h:123, .cc:559), keyed by the vid index of the composite field and filled in query order
   (.cc:213-219).  libstdc++ keeps all nodes in one list; a new key goes to the HEAD of the list when its bucket (key % 13 for
   up to 13 keys) is empty and directly in front of the bucket's first node otherwise.  Two fields A (queried first), B:
   iteration B, A.  Three fields A, B, C with C in A's bucket: list after A: [A]; after B: [B, A]; C goes in front of A:
   [B, C, A].
"""
import gzip
import json
import os

import pytest

import helpers

GOLDEN_CASE = "t0_1_2_all_asa_loading"
# (bins, counts) of the additional histogram fields per input sample at 1:17385, written like AS_RAW_MQRankSum in the fixtures:
# one entry per allele of the sample (REF | ALT | <NON_REF>), "bin,count" pairs
SECOND = {"t0_asa": "|0.5,1|NaN", "t1_asa": "|0.25,2|NaN", "t2_asa": "|0.5,3|NaN"}
THIRD = {"t0_asa": "|7,1|NaN", "t1_asa": "|7,2,9,1|NaN", "t2_asa": "|8,3|NaN"}
NEG = {"t0_asa": -4, "t1_asa": -1, "t2_asa": -1}


def _make_inputs(tmp_path, n_pad_fields):
    """the reference's t{0,1,2}_asa fixtures with three more INFO fields in the variant record: NEG (int, mean) and two more
    histogram_sum fields; `n_pad_fields` unused fields in the vid move the third histogram field's vid index"""
    src_vid = json.load(open(os.path.join(helpers.GOLDEN, "inputs", "vid_all_asa.json")))
    fields = {}
    for k, v in src_vid["fields"].items():
        fields[k] = v
        if k == "MQ0":
            fields["NEG"] = {"vcf_field_class": ["INFO"], "type": "int", "VCF_field_combine_operation": "mean"}
    hist = dict(src_vid["fields"]["AS_RAW_MQRankSum"])
    fields["AS_Second"] = dict(hist)
    for i in range(n_pad_fields):
        fields["PAD%d" % i] = {"vcf_field_class": ["INFO"], "type": "int"}
    fields["AS_Third"] = dict(hist)
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
                out.append('##INFO=<ID=NEG,Number=1,Type=Integer,Description="x">')
                out.append('##INFO=<ID=AS_Second,Number=1,Type=String,Description="x">')
                out.append('##INFO=<ID=AS_Third,Number=1,Type=String,Description="x">')
            if not l.startswith("#") and l.split("\t")[1] == "17385":
                c = l.split("\t")
                c[7] += ";NEG=%d;AS_Second=%s;AS_Third=%s" % (NEG[name], SECOND[name], THIRD[name])
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
    return cells, q, vid


def _vid_index_of(vid, name):
    """vid field index as the reference's FileBasedVidMapper assigns it (vid_mapper.cc:1290-1440 + flatten_field :727-791): file
    order; a field that is both INFO and FORMAT is followed by its <name>_FORMAT twin, a tuple-typed field by one flattened field
    per tuple element"""
    idx = 0
    for k, v in vid["fields"].items():
        if k == name:
            return idx
        idx += 1
        cls = v.get("vcf_field_class", [])
        if "INFO" in cls and "FORMAT" in cls:
            idx += 1
        if isinstance(v.get("type"), list) and len(v["type"]) > 1:
            idx += len(v["type"])
    raise KeyError(name)


def _expected_body(order):
    """the golden body of the unmodified fixtures with the hand-derived additions spliced into the variant record"""
    body = [l for l in helpers.golden_text(GOLDEN_CASE).decode().splitlines() if not l.startswith("#")]
    # merged alleles at 1:17385 are G -> A, T, <NON_REF> (golden).  Per merged allele: REF nothing; A = t0 + t2; T = t1;
    # <NON_REF> = NaN bins (not valid) -> nothing.  Bins print with 3 decimals when float, counts are ints.
    texts = {
        "AS_RAW_MQRankSum": "AS_RAW_MQRankSum=|0.600,6,0.800,2,0.900,15|0.100,2,0.600,7|",      # as in the golden
        "AS_Second": "AS_Second=|0.500,4|0.250,2|",                                              # A: 0.5 -> 1 + 3; T: 0.25 -> 2
        "AS_Third": "AS_Third=|7.000,1,8.000,3|7.000,2,9.000,1|",                                # bins in std::map order 7 < 8 < 9
    }
    out = []
    for l in body:
        c = l.split("\t")
        if c[1] == "17385":
            assert "MQ0=3;AS_RAW_MQ=" in c[7] and texts["AS_RAW_MQRankSum"] + ";DP=276" in c[7]
            c[7] = c[7].replace("MQ0=3;", "MQ0=3;NEG=1431655763;")
            c[7] = c[7].replace(texts["AS_RAW_MQRankSum"], ";".join(texts[n] for n in order))
        out.append("\t".join(c))
    return ("\n".join(out) + "\n").encode()


def _libstdcxx_order(keys_in_insertion_order, nbkt=13):
    """the rule of the docstring, written out: list of keys after single inserts into an empty std::unordered_map<unsigned, T>"""
    lst = []
    for k in keys_in_insertion_order:
        same = [i for i, x in enumerate(lst) if x % nbkt == k % nbkt]
        lst.insert(same[0] if same else 0, k)
    return lst


CONFIGS = [
    # (unused vid fields in front of AS_Third, expected order of the three histogram fields)
    (0, ["AS_Third", "AS_Second", "AS_RAW_MQRankSum"]),      # three different buckets: newest first
    (7, ["AS_Second", "AS_Third", "AS_RAW_MQRankSum"]),      # AS_Third's index = AS_RAW_MQRankSum's + 13: it goes in front of that one
]


@pytest.mark.parametrize("pad,order", CONFIGS)
def test_the_expected_orders_follow_from_the_vid_indices(tmp_path, pad, order):
    """the hand derivation itself: indices 22 / 25 / 28 (pad 0) and 22 / 25 / 35 (pad 7), buckets mod 13"""
    _, _, vid = _make_inputs(tmp_path, pad)
    idx = {n: _vid_index_of(vid, n) for n in ("AS_RAW_MQRankSum", "AS_Second", "AS_Third")}
    assert idx["AS_RAW_MQRankSum"] == 22 and idx["AS_Second"] == 25 and idx["AS_Third"] == (28 if pad == 0 else 35)
    lst = _libstdcxx_order([idx["AS_RAW_MQRankSum"], idx["AS_Second"], idx["AS_Third"]])
    assert [n for k in lst for n in idx if idx[n] == k] == order


@pytest.mark.parametrize("pad,order", CONFIGS)
def test_int_mean_and_histogram_order_oracle_and_kernel_bodies(tmp_path, pad, order):
    cells, q, _ = _make_inputs(tmp_path, pad)
    want = _expected_body(order)
    txt, nrec, _ = helpers.oracle_run(q, cells, with_header=False)
    assert nrec == 4 and txt == want
    got, err = helpers.hostsim_run(q, cells, with_header=False)
    assert err == 0 and got == want
    hdr, _, _ = helpers.oracle_run(q, cells)
    for name in ("NEG", "AS_Second", "AS_Third"):
        assert hdr.count(b"##INFO=<ID=" + name.encode() + b",") == 1


@pytest.mark.gpu
@pytest.mark.parametrize("pad,order", CONFIGS)
def test_int_mean_and_histogram_order_device(tmp_path, pad, order):
    import genomicsdb_amd
    cells, q, _ = _make_inputs(tmp_path, pad)
    want = _expected_body(order)
    hdr_and_body, _, _ = helpers.oracle_run(q, cells)
    assert hdr_and_body.endswith(want)
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    got = s.read()
    s.close()
    assert got.endswith(want) and got == hdr_and_body
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
    bcf = s.read()
    s.close()
    assert helpers.bcf_stream_to_text(bcf) == hdr_and_body


def test_a_soft_masked_reference_base_becomes_N(tmp_path):
    """`m_legal_bases` holds A, T, G, C in upper case only (broad_combined_gvcf.cc:51,823-830): a record whose merged REF is the
    placeholder `N` (nobody starts at its position) takes the FASTA base - unless that base is not a legal one, e.g. the lower-case
    letter of a soft-masked region.  Golden t0_1_2_vcf_at_0: the record at 1:12278 has REF C from the FASTA; with that base written
    as `c` the reference prints N."""
    import gzip as gz
    from golden_cases import CASES
    name, callsets, vid, ov, golden, mode = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    with gz.open(q["reference_genome"], "rt") as f:
        lines = f.read().split("\n")
    assert lines[0].startswith(">1")
    width = len(lines[1])
    row, col = divmod(12278 - 1, width)
    assert lines[1 + row][col] == "C"
    lines[1 + row] = lines[1 + row][:col] + "c" + lines[1 + row][col + 1:]
    fa = tmp_path / "masked.fasta.gz"
    with gz.open(fa, "wt") as f:
        f.write("\n".join(lines))
    q["reference_genome"] = str(fa)
    want = helpers.golden_text(golden).replace(b"1\t12278\t.\tC\t<NON_REF>", b"1\t12278\t.\tN\t<NON_REF>")
    assert want != helpers.golden_text(golden)
    txt, _, _ = helpers.oracle_run(q, cells)
    assert txt == want
    got, err = helpers.hostsim_run(q, cells)
    assert err == 0 and got == want
