#!/usr/bin/env python3
"""Fixture importer: single- or multi-sample (g)VCF text -> GenomicsDB *binary cells*.

TEST TOOLING, not product code.  It exists so that the golden input VCFs of the
reference's own test-suite (tests/inputs/vcfs/*.vcf.gz) can be turned into the
cell stream the scan/combine hot path consumes.  The cell layout restated here
is the one VCF2Binary::convert_VCF_to_binary_for_callset writes
(reference src/main/cpp/src/vcf/vcf2binary.cc:991-1196, field conversion
:715-989) for the schema VidMapper::build_tiledb_array_schema declares
(src/main/cpp/src/utils/vid_mapper.cc:354-442):

  [row i64][col i64][cell_size u64][END i64]
  [REF  : i32 len + chars][ALT : i32 len + 'A|C|&' ('&' = <NON_REF>)]
  [ID   : i32 len + chars]          (only if the vid declares an "ID" field)
  [QUAL f32][FILTER : i32 n + n x i32 (vid field idx)]
  [INFO attributes in vid order][FORMAT attributes in vid order]
     fixed-length attr  = num x element           (missing -> TileDB null)
     var-length   attr  = i32 num + num x element (missing -> num = 0)

TileDB null sentinels (Intel TileDB fork constants; see DESIGN.md "sentinels"):
  int32 INT32_MAX, float FLT_MAX bit pattern, char CHAR_MAX (127).
bcf missing/vector_end (htslib): int32 INT32_MIN / INT32_MIN+1,
  float bits 0x7F800001 / 0x7F800002.

Two-dimensional fields ("length": [ "R", "var" ], allele-specific annotations written as
delimited strings in the VCF) are stored as the reference stores them
(GenomicsDBMultiDVectorField::parse_and_store_numeric, genomicsdb_multid_vector_field.cc:238-464;
vcf2binary.cc:857-913): one variable-length byte attribute per element of the type tuple,
   [i32 nbytes][u64 size of data][inner vectors back to back][u64 #entries][u64 offsets x (#entries + 1)]
An empty inner vector holds one bcf-missing element ("" and "NaN" parse to missing).
"""
import gzip
import json
import struct
import sys
from collections import OrderedDict

INT32_MAX = 2**31 - 1
TILEDB_NULL_FLOAT_BITS = 0x7F7FFFFF  # FLT_MAX
TILEDB_NULL_CHAR = 127
BCF_INT32_MISSING = -(2**31)
BCF_INT32_VECTOR_END = -(2**31) + 1
BCF_FLOAT_MISSING_BITS = 0x7F800001
BCF_FLOAT_VECTOR_END_BITS = 0x7F800002

INT_TYPES = {"int", "Int", "integer", "Integer"}
FLOAT_TYPES = {"float", "Float"}
FLAG_TYPES = {"bool", "Bool", "boolean", "Boolean", "flag", "Flag"}
STR_TYPES = {"string", "String", "char", "Char"}

KNOWN_LENGTH = {  # known_field_info.cc:239-283 (fields without "length" in vid)
    "REF": "VAR", "ALT": "VAR", "FILTER": "VAR", "AF": "A", "AC": "A", "AD": "R",
    "PL": "G", "GT": "PP", "SB": 4, "PGT": "VAR", "PID": "VAR",
}
KNOWN_FIELDS = {"END", "REF", "ALT", "QUAL", "FILTER", "BaseQRankSum", "ClippingRankSum",
                "MQRankSum", "ReadPosRankSum", "DP", "MQ", "RAW_MQ", "MQ0", "DP_FORMAT",
                "MIN_DP", "GQ", "SB", "AD", "PL", "AF", "AN", "AC", "GT", "PS", "PGT",
                "PID", "ExcessHet", "ID"}
LENGTH_ALIASES = {"A": "A", "BCF_VL_A": "A", "R": "R", "BCF_VL_R": "R", "G": "G",
                  "BCF_VL_G": "G", "P": "P", "BCF_VL_P": "P", "VAR": "VAR",
                  "BCF_VL_VAR": "VAR", "PP": "PP", "PHASED_PLOIDY": "PP"}


class Field:
    def __init__(self, name, idx, d):
        self.name = name          # unique attribute name (DP_FORMAT ...)
        self.vcf_name = name
        self.idx = idx
        self.cls = set(d.get("vcf_field_class", []))
        t = d["type"]
        kind_of = lambda x: ("int" if x in INT_TYPES else "float" if x in FLOAT_TYPES
                             else "flag" if x in FLAG_TYPES else "str")
        self.tuple_kinds = [kind_of(x) for x in t] if isinstance(t, list) else [kind_of(t)]
        self.kind = self.tuple_kinds[0]
        self.tuple_idx = None     # index in the parent's type tuple (flattened children only)
        self.ndim = 1
        length = d.get("length", None)
        if length is None:
            length = KNOWN_LENGTH.get(name, 1) if name in KNOWN_FIELDS else 1
        if isinstance(length, list) and len(length) == 1:
            length = length[0]
        if isinstance(length, dict):
            length = length.get("variable_length_descriptor", length.get("fixed_length"))
        if isinstance(length, list):
            if len(length) != 2:
                raise NotImplementedError("fields of more than 2 dimensions are not handled by the fixture importer")
            self.ndim = 2
            delim = d.get("vcf_delimiter", ["|", ","])
            self.delims = [delim] if isinstance(delim, str) else list(delim)
            length = "VAR"        # stored as a variable-length byte attribute
        if isinstance(length, str):
            up = length.upper()
            if up in LENGTH_ALIASES:
                length = LENGTH_ALIASES[up]
            else:
                try:
                    length = int(length, 0)
                except ValueError:
                    length = "VAR"
        self.length = length      # int (fixed) or 'A','R','G','P','PP','VAR'
        self.combine = d.get("VCF_field_combine_operation")
        if self.combine is None:  # known_field_info.cc:285-308 defaults
            self.combine = {"RAW_MQ": "sum", "DP": "DP"}.get(name)

    @property
    def is_sum_like(self):
        """FieldInfo::is_VCF_field_combine_operation_sum (vid_mapper.cc:1187-1193)"""
        return self.combine in ("sum", "DP", "element_wise_sum", "elementwise_sum", "histogram_sum")

    @property
    def fixed(self):
        return isinstance(self.length, int)


def load_vid(path):
    """Field list in vid index order with the INFO+FORMAT split of VidMapper::flatten_field
    (vid_mapper.cc:727-748): a field that is both INFO and FORMAT gets a second entry
    <name>_FORMAT right after it (vcf name unchanged)."""
    d = json.load(open(path), object_pairs_hook=OrderedDict)
    fields = []
    fd = d["fields"]
    items = fd.items() if isinstance(fd, dict) else [(x.get("name", x.get("field_name")), x) for x in fd]
    for name, info in items:
        f = Field(name, len(fields), info)
        fields.append(f)
        if "INFO" in f.cls and "FORMAT" in f.cls:
            g = Field(name, len(fields), info)
            g.name = name + "_FORMAT"
            g.cls = {"FORMAT"}
            f.cls = {"INFO"}
            fields.append(g)
        if len(f.tuple_kinds) > 1:
            # every element of the type tuple becomes a field <name>_tuple_element_<i> (vcf name unchanged); the
            # composite parent keeps its index but is not an attribute of the array (vid_mapper.cc:751-787, :399-404)
            f.is_composite = True
            for i, k in enumerate(f.tuple_kinds):
                g = Field(name, len(fields), info)
                g.name = "%s_tuple_element_%d" % (name, i)
                g.cls = set(f.cls)
                g.kind = k
                g.tuple_idx = i
                fields.append(g)
    contigs = OrderedDict()
    cd = d["contigs"]
    citems = cd.items() if isinstance(cd, dict) else [
        (x.get("name", x.get("contig_name", x.get("chromosome_name"))), x) for x in cd]
    for name, info in citems:
        contigs[name] = (int(info["tiledb_column_offset"]), int(info["length"]))
    return fields, contigs


def schema_attributes(fields):
    """Attribute order of the array schema (vid_mapper.cc:364-429)."""
    names = {f.name for f in fields}
    attrs = ["END", "REF", "ALT"]
    if "ID" in names:
        attrs.append("ID")
    attrs += ["QUAL", "FILTER"]
    info = [f for f in fields if "INFO" in f.cls and f.name != "END" and not getattr(f, "is_composite", False)]
    fmt = [f for f in fields if "FORMAT" in f.cls and f.name != "END" and not getattr(f, "is_composite", False)]
    return attrs, info, fmt


def variant_is_deletion_indel(ref, alt):
    """bcf_get_variant_type(line, j) == VCF_INDEL && strlen(ref) > strlen(alt)
    (vcf2binary.cc:1046-1057; htslib bcf_set_variant_type): restated for plain
    base alleles - symbolic alleles and '*' are never INDELs."""
    if alt.startswith("<") or alt == "*" or alt == ".":
        return False
    if len(ref) == 1 and len(alt) == 1:
        return False
    r, a = 0, 0
    while r < len(ref) and a < len(alt) and ref[r].upper() == alt[a].upper():
        r += 1
        a += 1
    if a < len(alt) and r == len(ref):
        return False                       # insertion
    if r < len(ref) and a == len(alt):
        return True                        # pure deletion: VCF_INDEL, ref longer
    if r == len(ref) and a == len(alt):
        return False
    re_, ae = len(ref) - 1, len(alt) - 1
    while re_ > r and ae > a and ref[re_].upper() == alt[ae].upper():
        re_ -= 1
        ae -= 1
    if ae == a:
        if re_ == r:
            return False                   # SNP
        return ref[re_].upper() == alt[ae].upper() and len(ref) > len(alt)
    if re_ == r:
        return ref[re_].upper() == alt[ae].upper() and len(ref) > len(alt)
    return False


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def parse_2d_element(kind, tok):
    """str_to_element<int|float> (genomicsdb_multid_vector_field.cc:31-87): "" and NaN are bcf missing."""
    if tok == "" or tok.lower() == "nan":
        return None
    return int(tok, 0) if kind == "int" else float(tok)


def encode_2d(field, text, n_samples=1, sample_idx=0):
    """A delimited 2-D string -> the bytes of this tuple element's attribute.  Tuple elements alternate inside an
    inner vector ("bin,count,bin,count"); shorter tuple elements are padded with missing; with several samples in
    the VCF a sum-like INFO field is divided up among them (vcf2binary.cc:862-884: histogram_sum divides the
    second tuple element only)."""
    if text is None or text == ".":
        return struct.pack("<i", 0)
    kinds = field.tuple_kinds
    me = field.tuple_idx or 0
    divide = field.is_sum_like and "INFO" in field.cls and n_samples > 1
    if divide and field.combine == "histogram_sum":
        divide = me == 1
    data = b""
    offsets = [0]
    for inner in text.split(field.delims[0]):
        toks = inner.split(field.delims[1])
        per = [[] for _ in kinds]
        for i, tok in enumerate(toks):
            per[i % len(kinds)].append(tok)
        n = max(len(x) for x in per)
        mine = per[me] + [""] * (n - len(per[me]))
        for tok in mine:
            v = parse_2d_element(kinds[me], tok)
            if kinds[me] == "int":
                if v is None:
                    v = BCF_INT32_MISSING
                elif divide:
                    q, r = divmod(v, n_samples)
                    v = q + (1 if sample_idx < r else 0)
                data += struct.pack("<i", v)
            else:
                if v is None:
                    data += struct.pack("<I", BCF_FLOAT_MISSING_BITS)
                else:
                    if divide:
                        v = struct.unpack("<f", struct.pack("<f", v))[0] / n_samples
                    data += struct.pack("<f", v)
        offsets.append(len(data))
    blob = struct.pack("<Q", len(data)) + data + struct.pack("<Q", len(offsets) - 1) + b"".join(struct.pack("<Q", o) for o in offsets)
    return struct.pack("<i", len(blob)) + blob


def encode_values(field, text, n_alt, n_samples=1, sample_idx=0):
    """One INFO/FORMAT value string -> attribute bytes (vcf2binary.cc:771-969)."""
    if field.ndim == 2:
        return encode_2d(field, text, n_samples, sample_idx)
    missing = text is None or text == "."
    if field.kind == "flag":
        return bytes([1 if text is not None else TILEDB_NULL_CHAR])
    if field.kind == "str":
        if missing:
            return struct.pack("<i", 0)
        b = text.encode()
        return struct.pack("<i", len(b)) + b
    toks = [] if missing else text.split(",")
    if len(toks) == 1 and toks[0] == ".":
        toks = []
    if not toks:
        if field.fixed:
            if field.kind == "int":
                return struct.pack("<i", INT32_MAX) * field.length
            return struct.pack("<I", TILEDB_NULL_FLOAT_BITS) * field.length
        return struct.pack("<i", 0)
    out = b""
    for t in toks:
        if field.kind == "int":
            v = BCF_INT32_MISSING if t == "." else int(t)
            if (field.is_sum_like and "INFO" in field.cls and n_samples > 1 and t != "."):
                # divide_up_among_samples (vcf2binary.cc:34-53)
                q, r = divmod(v, n_samples)
                v = q + (1 if sample_idx < r else 0)
            out += struct.pack("<i", v)
        else:
            if t == ".":
                out += struct.pack("<I", BCF_FLOAT_MISSING_BITS)
            else:
                v = float(t)
                if field.is_sum_like and "INFO" in field.cls and n_samples > 1:
                    v = struct.unpack("<f", struct.pack("<f", v))[0] / n_samples
                out += struct.pack("<f", v)
    if field.fixed:
        if len(toks) != field.length:
            raise ValueError("field %s: %d values, expected %d" % (field.name, len(toks), field.length))
        return out
    return struct.pack("<i", len(toks)) + out


def encode_gt(field, text):
    """GT -> allele idx ints, interleaved phase ints for 'PP' (vcf2binary.cc:923-959)."""
    if text is None or text == ".":
        # htslib parses GT '.' as one missing allele: value bcf_gt_missing -> allele -1
        alleles, phases = [-1], []
    else:
        alleles, phases, cur = [], [], ""
        for ch in text:
            if ch in "/|":
                alleles.append(cur)
                phases.append(1 if ch == "|" else 0)
                cur = ""
            else:
                cur += ch
        alleles.append(cur)
        alleles = [-1 if a == "." else int(a) for a in alleles]
    if field.length == "PP":
        vals = [alleles[0]]
        for p, a in zip(phases, alleles[1:]):
            vals += [p, a]
    else:
        vals = alleles
    return struct.pack("<i", len(vals)) + b"".join(struct.pack("<i", v) for v in vals)


def convert_vcf(vcf_path, fields, contigs, sample_to_row, treat_deletions_as_intervals=True):
    """Yield (row, col, end, cell_bytes) for every (record, sample) of one VCF."""
    attrs, info_fields, fmt_fields = schema_attributes(fields)
    has_id = "ID" in attrs
    name_to_idx = {f.name: f.idx for f in fields}
    opener = gzip.open if vcf_path.endswith(".gz") else open
    samples = []
    with opener(vcf_path, "rt") as fp:
        for line in fp:
            line = line.rstrip("\n")
            if line.startswith("##"):
                continue
            if line.startswith("#CHROM"):
                samples = line.split("\t")[9:]
                continue
            if not line:
                continue
            c = line.split("\t")
            chrom, pos, vid_, ref, alt, qual, flt, info = c[:8]
            fmt_keys = c[8].split(":") if len(c) > 8 else []
            col = contigs[chrom][0] + int(pos) - 1
            alts = [] if alt == "." else alt.split(",")
            info_d = OrderedDict()
            if info != ".":
                for kv in info.split(";"):
                    if "=" in kv:
                        k, v = kv.split("=", 1)
                        info_d[k] = v
                    else:
                        info_d[kv] = ""
            if "END" in info_d:
                end = contigs[chrom][0] + int(info_d["END"]) - 1
            else:
                end = col
                if treat_deletions_as_intervals:
                    for a in alts:
                        if variant_is_deletion_indel(ref, a):
                            end = col + len(ref) - 1
                            break
            alt_ser = "|".join("&" if a == "<NON_REF>" else a for a in alts)
            for sidx, sname in enumerate(samples):
                if sname not in sample_to_row:
                    continue
                row = sample_to_row[sname]
                body = struct.pack("<q", end)
                body += struct.pack("<i", len(ref)) + ref.encode()
                body += struct.pack("<i", len(alt_ser)) + alt_ser.encode()
                if has_id:
                    if vid_ and vid_ != ".":
                        body += struct.pack("<i", len(vid_)) + vid_.encode()
                    else:
                        body += struct.pack("<i", 0)
                if qual == ".":
                    body += struct.pack("<I", TILEDB_NULL_FLOAT_BITS)
                else:
                    body += struct.pack("<f", float(qual))
                if flt == ".":
                    body += struct.pack("<i", 0)
                else:
                    ids = [name_to_idx[x] for x in flt.split(";")]
                    body += struct.pack("<i", len(ids)) + b"".join(struct.pack("<i", i) for i in ids)
                for f in info_fields:
                    txt = info_d.get(f.vcf_name)
                    body += encode_values(f, txt, len(alts), len(samples), sidx)
                svals = c[9 + sidx].split(":") if len(c) > 9 else []
                fmt_d = dict(zip(fmt_keys, svals))
                for f in fmt_fields:
                    txt = fmt_d.get(f.vcf_name)
                    if f.vcf_name == "GT":
                        body += encode_gt(f, txt)
                    else:
                        body += encode_values(f, txt, len(alts))
                cell_size = 16 + 8 + len(body)
                cell = struct.pack("<qqQ", row, col, cell_size) + body
                yield row, col, end, cell


def load_callsets(path):
    d = json.load(open(path), object_pairs_hook=OrderedDict)
    cs = d["callsets"]
    items = cs.items() if isinstance(cs, dict) else [
        (x.get("sample_name", x.get("name", x.get("callset_name"))), x) for x in cs]
    out = []
    for name, info in items:
        out.append((name, int(info["row_idx"]), int(info.get("idx_in_file", 0)), info.get("filename")))
    return out


def build_cells(callsets_path, vid_path, vcf_dir_map=None):
    """All begin-cells of a callset mapping, in column-major (col,row) order - the order in
    which VCF2TileDBLoader hands cells to its operators (tiledb_loader.cc:845-965)."""
    fields, contigs = load_vid(vid_path)
    callsets = load_callsets(callsets_path)
    by_file = OrderedDict()
    for name, row, idx_in_file, fn in callsets:
        by_file.setdefault(fn, []).append((name, row, idx_in_file))
    cells = []
    for fn, lst in by_file.items():
        path = vcf_dir_map(fn) if vcf_dir_map else fn
        opener = gzip.open if path.endswith(".gz") else open
        with opener(path, "rt") as fp:
            for line in fp:
                if line.startswith("#CHROM"):
                    file_samples = line.rstrip("\n").split("\t")[9:]
                    break
        # callset name -> row through idx_in_file (the name in the JSON may differ from the VCF's)
        s2r = {file_samples[idx]: row for (_, row, idx) in lst}
        cells += list(convert_vcf(path, fields, contigs, s2r))
    cells.sort(key=lambda t: (t[1], t[0]))
    return cells


def main():
    if len(sys.argv) < 4:
        sys.exit("usage: vcf2cells.py <callsets.json> <vid.json> <out.cells> [vcf_root]")
    root = sys.argv[4] if len(sys.argv) > 4 else "."
    import os
    cells = build_cells(sys.argv[1], sys.argv[2], lambda fn: os.path.join(root, fn))
    with open(sys.argv[3], "wb") as fp:
        for _, _, _, b in cells:
            fp.write(b)
    print("%d cells, %d bytes" % (len(cells), sum(len(b) for *_, b in cells)))


if __name__ == "__main__":
    main()
