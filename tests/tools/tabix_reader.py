"""A pure-Python reader of .tbi / .csi indexes and of the BGZF files they point into: test infrastructure for index_output_VCF
(tests/test_vcf_index.py).  Written from the published formats (The Tabix index file format; CSIv1; SAM specification 4.1 BGZF and 5.3 the
binning scheme), independently of genomicsdb_amd/csrc/host/vcf_index.cc."""
import gzip
import struct
import zlib


class Bgzf:
    def __init__(self, path):
        self.data = open(path, "rb").read()
        self.cache = {}

    def block(self, coff):
        """(inflated bytes, file offset of the next block) of the block at file offset coff"""
        if coff not in self.cache:
            d = self.data
            if coff >= len(d):
                return b"", coff
            assert d[coff:coff + 4] == b"\x1f\x8b\x08\x04" and d[coff + 12:coff + 14] == b"BC"
            bsize = struct.unpack_from("<H", d, coff + 16)[0]
            raw = zlib.decompress(d[coff + 18:coff + bsize + 1 - 8], -15)
            assert len(raw) == struct.unpack_from("<I", d, coff + bsize + 1 - 4)[0]
            self.cache[coff] = (raw, coff + bsize + 1)
        return self.cache[coff]

    def read_from(self, voff):
        """generator of (virtual offset, byte) from virtual offset voff to the end of the file"""
        coff, uoff = voff >> 16, voff & 0xFFFF
        while True:
            raw, nxt = self.block(coff)
            if nxt == coff:
                return
            for i in range(uoff, len(raw)):
                yield (coff << 16) | i, raw[i]
            coff, uoff = nxt, 0


def reg2bins(beg, end, min_shift=14, depth=5):
    """all bins that may hold records overlapping [beg, end)"""
    bins = []
    end -= 1
    s, t = min_shift + depth * 3, 0
    for l in range(depth + 1):
        b, e = t + (beg >> s), t + (end >> s)
        bins.extend(range(b, e + 1))
        s -= 3
        t += 1 << (l * 3)
    return bins


class Index:
    """.tbi (fmt 'tbi') or .csi"""

    def __init__(self, path):
        raw = gzip.decompress(open(path, "rb").read())
        at = 0
        magic = raw[:4]
        at = 4
        self.names = None
        if magic == b"TBI\x01":
            self.kind = "tbi"
            n_ref, fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<8i", raw, at)
            at += 32
            assert (fmt, col_seq, col_beg, col_end, meta, skip) == (2, 1, 2, 0, ord("#"), 0)
            self.names = [x.decode() for x in raw[at:at + l_nm].split(b"\x00")[:-1]]
            at += l_nm
            self.min_shift, self.depth = 14, 5
        else:
            assert magic == b"CSI\x01"
            self.kind = "csi"
            self.min_shift, self.depth, l_aux = struct.unpack_from("<3i", raw, at)
            at += 12 + l_aux
            n_ref = struct.unpack_from("<i", raw, at)[0]
            at += 4
        self.refs = []
        for _ in range(n_ref):
            n_bin = struct.unpack_from("<i", raw, at)[0]
            at += 4
            bins, loff = {}, {}
            for _ in range(n_bin):
                b = struct.unpack_from("<I", raw, at)[0]
                at += 4
                if self.kind == "csi":
                    loff[b] = struct.unpack_from("<Q", raw, at)[0]
                    at += 8
                n_chunk = struct.unpack_from("<i", raw, at)[0]
                at += 4
                bins[b] = [struct.unpack_from("<QQ", raw, at + 16 * i) for i in range(n_chunk)]
                at += 16 * n_chunk
            linear = []
            if self.kind == "tbi":
                n_intv = struct.unpack_from("<i", raw, at)[0]
                at += 4
                linear = list(struct.unpack_from("<%dQ" % n_intv, raw, at))
                at += 8 * n_intv
            self.refs.append({"bins": bins, "linear": linear, "loffset": loff})
        assert len(raw) - at in (0, 8)
        self.meta_bin = ((1 << ((self.depth + 1) * 3)) - 1) // 7 + 1

    def chunks(self, tid, beg, end):
        """virtual-offset intervals of the file that may hold records of contig tid overlapping [beg, end)"""
        if tid >= len(self.refs):
            return []
        r = self.refs[tid]
        lo = 0
        if self.kind == "tbi" and r["linear"]:
            w = beg >> self.min_shift
            lo = r["linear"][min(w, len(r["linear"]) - 1)] if w < len(r["linear"]) else r["linear"][-1]
        elif self.kind == "csi":
            # hts_itr_query's min_off (htslib hts.c): the loffset of the leaf bin of `beg` if the index has it, else of the nearest
            # bin in front of it on the same level, else of the parent - chunks that end at or below it are dropped
            b = ((1 << (self.depth * 3)) - 1) // 7 + (beg >> self.min_shift)
            while b:
                if b in r["bins"] and b != self.meta_bin:
                    break
                parent = (b - 1) >> 3
                first = (parent << 3) + 1
                b = b - 1 if b > first else parent
            lo = r["loffset"].get(b, 0) if b in r["bins"] else 0
        out = []
        for b in reg2bins(beg, end, self.min_shift, self.depth):
            if b == self.meta_bin:
                continue
            for cb, ce in r["bins"].get(b, []):
                if ce > lo:
                    out.append((max(cb, lo) if self.kind == "tbi" else cb, ce))
        return sorted(out)


def fetch_vcf(path, idx, chrom, beg, end):
    """record lines of a bgzip'ed VCF that overlap [beg, end) (0-based, half open) of contig chrom, found through the index"""
    if chrom not in idx.names:
        return []
    bz = Bgzf(path)
    got, seen = [], set()
    for cb, ce in idx.chunks(idx.names.index(chrom), beg, end):
        line, start = bytearray(), None
        for voff, byte in bz.read_from(cb):
            if start is None:
                if voff >= ce:
                    break
                start = voff
            if byte == 10:
                if start not in seen:
                    seen.add(start)
                    c = bytes(line).split(b"\t")
                    if c[0].decode() == chrom:
                        b0, e0 = vcf_interval(c)
                        if b0 < end and e0 > beg:
                            got.append((start, bytes(line)))
                line, start = bytearray(), None
            else:
                line.append(byte)
    return [l for _, l in sorted(got)]


def vcf_interval(cols):
    b0 = int(cols[1]) - 1
    e0 = b0 + len(cols[3])
    for kv in cols[7].split(b";"):
        if kv.startswith(b"END="):
            e0 = int(kv[4:])
    return b0, e0


def fetch_bcf(path, idx, tid, beg, end):
    """(virtual offset, record bytes) of BCF2 records of contig index tid overlapping [beg, end)"""
    bz = Bgzf(path)
    got = {}
    for cb, ce in idx.chunks(tid, beg, end):
        it = bz.read_from(cb)
        while True:
            try:
                v0, b = next(it)
            except StopIteration:
                break
            if v0 >= ce:
                break
            head = bytes([b] + [next(it)[1] for _ in range(7)])
            l_shared, l_indiv = struct.unpack("<II", head)
            body = bytes(next(it)[1] for _ in range(l_shared + l_indiv))
            chrom, pos, rlen = struct.unpack_from("<iii", body, 0)
            if chrom == tid and pos < end and pos + max(1, rlen) > beg:
                got[v0] = head + body
    return [got[k] for k in sorted(got)]
