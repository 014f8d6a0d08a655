"""BCF2 ("bu": uncompressed, unblocked) -> VCF text, for the parity tests of the device's BCF page assembly.

TEST infrastructure.  An independent restatement of the BCF2 layout (VCFv4.2/BCFv2.2 specification, section 6) and of how
htslib prints a BCF record as VCF text (vcf.c: vcf_format, bcf_fmt_array, bcf_format_gt), so that a stream produced with
output format "bu" can be compared byte for byte with the reference's TEXT goldens: same header (the ',IDX=n' keys a BCF header
carries are dropped), same records.  Floats are printed by the oracle's format_float (the htslib-fork kputd rule the goldens pin).
"""
import re
import struct

INT8_MISSING, INT8_VEND = -128, -127
INT16_MISSING, INT16_VEND = -32768, -32767
INT32_MISSING, INT32_VEND = -2147483648, -2147483647
FLOAT_MISSING, FLOAT_VEND = 0x7F800001, 0x7F800002
BT_NULL, BT_INT8, BT_INT16, BT_INT32, BT_FLOAT, BT_CHAR = 0, 1, 2, 3, 5, 7


class BCFError(Exception):
    pass


def _typed_descriptor(buf, at):
    b = buf[at]
    at += 1
    n, t = b >> 4, b & 15
    if n == 15:
        n, at = _typed_int(buf, at)
    return n, t, at


def _typed_int(buf, at):
    n, t, at = _typed_descriptor(buf, at)
    if n != 1 or t not in (BT_INT8, BT_INT16, BT_INT32):
        raise BCFError("typed integer expected at %d" % at)
    vals, at = _values(buf, at, t, 1)
    return vals[0], at


_FMT = {BT_INT8: ("b", 1), BT_INT16: ("<h", 2), BT_INT32: ("<i", 4), BT_FLOAT: ("<I", 4), BT_CHAR: ("B", 1)}


def _values(buf, at, t, n):
    if t == BT_NULL or n == 0:
        return [], at
    f, w = _FMT[t]
    vals = [struct.unpack_from(f, buf, at + i * w)[0] for i in range(n)]
    return vals, at + n * w


def _missing_vend(t):
    return {BT_INT8: (INT8_MISSING, INT8_VEND), BT_INT16: (INT16_MISSING, INT16_VEND), BT_INT32: (INT32_MISSING, INT32_VEND),
            BT_FLOAT: (FLOAT_MISSING, FLOAT_VEND)}[t]


class Header:
    def __init__(self, text):
        self.text = text
        self.ids = {}       # dictionary of FILTER / INFO / FORMAT ids: index -> name
        self.contigs = {}
        self.samples = []
        next_id = 0
        next_ctg = 0
        seen = {}
        for line in text.split("\n"):
            m = re.match(r"##(FILTER|INFO|FORMAT|contig)=<(.*)>$", line)
            if m:
                kind, body = m.group(1), m.group(2)
                name = re.match(r"ID=([^,>]+)", body).group(1)
                idx = re.search(r",IDX=(\d+)", body)
                if kind == "contig":
                    i = int(idx.group(1)) if idx else next_ctg
                    self.contigs[i] = name
                    next_ctg = max(next_ctg, i + 1)
                else:
                    if name in seen:
                        continue
                    i = int(idx.group(1)) if idx else next_id
                    seen[name] = i
                    self.ids[i] = name
                    next_id = max(next_id, i + 1)
            elif line.startswith("#CHROM"):
                cols = line.split("\t")
                self.samples = cols[9:] if len(cols) > 9 else []
        if "PASS" not in seen:
            raise BCFError("no PASS filter in the header dictionary")

    def vcf_text(self):
        """the header as a VCF writer prints it: no IDX keys"""
        return re.sub(r",IDX=\d+>", ">", self.text)


def parse_stream(data):
    """-> (Header, [record bytes ...]) of a 'bu' stream"""
    if data[:5] != b"BCF\x02\x02":
        raise BCFError("bad magic %r" % data[:5])
    (l_text,) = struct.unpack_from("<I", data, 5)
    text = data[9:9 + l_text]
    if not text.endswith(b"\x00"):
        raise BCFError("header text is not NUL-terminated")
    hdr = Header(text[:-1].decode())
    at = 9 + l_text
    recs = []
    while at < len(data):
        l_shared, l_indiv = struct.unpack_from("<II", data, at)
        recs.append(data[at:at + 8 + l_shared + l_indiv])
        if len(recs[-1]) != 8 + l_shared + l_indiv:
            raise BCFError("truncated record at %d" % at)
        at += 8 + l_shared + l_indiv
    return hdr, recs


def _fmt_number(t, v, format_float):
    if t == BT_FLOAT:
        return format_float(struct.unpack("<f", struct.pack("<I", v))[0])
    return str(v)


def _fmt_array(t, vals, format_float):
    """htslib bcf_fmt_array"""
    if t == BT_CHAR:
        out = []
        for c in vals:
            if c == 0:
                break
            out.append("." if c == 7 else chr(c))
        return "".join(out) if out else "."     # (an empty string prints as '.' in vcf_format's callers)
    if not vals:
        return "."
    missing, vend = _missing_vend(t)
    out = []
    for v in vals:
        if v == vend:
            break
        out.append("." if v == missing else _fmt_number(t, v, format_float))
    return ",".join(out) if out else "."


def _fmt_gt(t, vals):
    """htslib bcf_format_gt"""
    _, vend = _missing_vend(t)
    out = ""
    n = 0
    for v in vals:
        if v == vend:
            break
        if n:
            out += "|" if (v & 1) else "/"
        out += "." if (v >> 1) == 0 else str((v >> 1) - 1)
        n += 1
    return out if n else "."


def record_to_text(hdr, rec, format_float):
    l_shared, l_indiv = struct.unpack_from("<II", rec, 0)
    at = 8
    rid, pos, rlen, qual_bits, n_allele_info, n_fmt_sample = struct.unpack_from("<iiiIII", rec, at)
    at += 24
    n_info, n_allele = n_allele_info & 0xFFFF, n_allele_info >> 16
    n_sample, n_fmt = n_fmt_sample & 0xFFFFFF, n_fmt_sample >> 24
    n, t, at = _typed_descriptor(rec, at)
    vals, at = _values(rec, at, t, n)
    rid_text = _fmt_array(BT_CHAR, vals, format_float) if n else "."
    alleles = []
    for _ in range(n_allele):
        n, t, at = _typed_descriptor(rec, at)
        vals, at = _values(rec, at, t, n)
        alleles.append(bytes(vals).decode())
    n, t, at = _typed_descriptor(rec, at)
    flt, at = _values(rec, at, t, n)
    cols = [hdr.contigs[rid], str(pos + 1), rid_text, alleles[0], ",".join(alleles[1:]) if n_allele > 1 else "."]
    cols.append("." if qual_bits == FLOAT_MISSING else format_float(struct.unpack("<f", struct.pack("<I", qual_bits))[0]))
    cols.append(";".join(hdr.ids[i] for i in flt) if flt else ".")
    info = []
    end_value = None
    for _ in range(n_info):
        key, at = _typed_int(rec, at)
        n, t, at = _typed_descriptor(rec, at)
        vals, at = _values(rec, at, t, n)
        name = hdr.ids[key]
        if name == "END" and vals:
            end_value = vals[0]
        info.append(name if t == BT_NULL or n == 0 else name + "=" + _fmt_array(t, vals, format_float))
    cols.append(";".join(info) if info else ".")
    if at != 8 + l_shared:
        raise BCFError("shared block length: parsed %d, l_shared %d" % (at - 8, l_shared))
    # rlen as this build defines it: END - POS when the record has an END, else the length of REF
    want_rlen = (end_value - pos) if end_value is not None else len(alleles[0])
    if rlen != want_rlen:
        raise BCFError("rlen %d, expected %d" % (rlen, want_rlen))
    if n_fmt:
        if n_sample != len(hdr.samples):
            raise BCFError("n_sample %d, header has %d samples" % (n_sample, len(hdr.samples)))
        keys, fields = [], []
        for _ in range(n_fmt):
            key, at = _typed_int(rec, at)
            n, t, at = _typed_descriptor(rec, at)
            w = _FMT[t][1] if t != BT_NULL else 0
            per_sample = []
            for s in range(n_sample):
                vals, _ = _values(rec, at + s * n * w, t, n)
                per_sample.append(vals)
            at += n_sample * n * w
            keys.append(hdr.ids[key])
            fields.append((t, per_sample))
        cols.append(":".join(keys))
        for s in range(n_sample):
            parts = []
            for key, (t, per_sample) in zip(keys, fields):
                parts.append(_fmt_gt(t, per_sample[s]) if key == "GT" else _fmt_array(t, per_sample[s], format_float))
            cols.append(":".join(parts))
    if at != len(rec):
        raise BCFError("individual block length: parsed %d, record %d" % (at, len(rec)))
    return "\t".join(cols)


def stream_to_vcf_text(data, format_float):
    hdr, recs = parse_stream(data)
    out = [hdr.vcf_text()]
    if not out[0].endswith("\n"):
        out[0] += "\n"
    for r in recs:
        out.append(record_to_text(hdr, r, format_float) + "\n")
    return "".join(out).encode()
