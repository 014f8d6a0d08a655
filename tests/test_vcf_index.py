"""index_output_VCF: the .tbi / .csi this build writes next to a BGZF output file (reference: vcf_adapter.cc:275-295 asks htslib for them).
A pure-Python reader of the two index formats (tests/tools/tabix_reader.py, written from the specifications) fetches regions through the
index; it must find exactly the records a scan of the whole file finds.  The CPU tests index bgzip'ed VCF text made here from the oracle's
output (several contigs, thousands of records, small blocks); the GPU tests index what gt_mpi_gather -O z / b writes."""
import json
import os
import random
import struct
import subprocess
import zlib

import pytest

import helpers
import tabix_reader

GENOME = [("1", 0, 60000), ("2", 60000, 9000), ("3", 69000, 50000), ("X", 119000, 25000), ("Y", 144000, 17000), ("MT", 161000, 4000)]
END = 165000


def _bgzip(data, block=3000, seed=1):
    """BGZF with small blocks of varying size (records straddle blocks), ending with the EOF block"""
    rnd = random.Random(seed)
    out = bytearray()
    at = 0
    while at < len(data):
        n = rnd.randint(block // 2, block)
        chunk = data[at:at + n]
        at += n
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25)
        out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return bytes(out)


def _build(path, is_bcf):
    import ctypes
    import genomicsdb_amd._lib as L
    lib = L.lib()
    lib.gdbamd_build_output_index.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert lib.gdbamd_build_output_index(os.fsencode(path), int(is_bcf)) == 0, lib.gdb_mi355_last_error().decode()


def _check_vcf(path, text, regions):
    idx = tabix_reader.Index(path + ".tbi")
    recs = [l for l in text.split(b"\n") if l and not l.startswith(b"#")]
    assert idx.kind == "tbi" and len(idx.names) >= 1
    order = []
    for l in recs:
        c = l.split(b"\t", 1)[0].decode()
        if not order or order[-1] != c:
            order.append(c)
    assert idx.names == order                                # contigs in order of appearance, like tabix
    nonempty = 0
    for chrom, beg, end in regions:
        want = []
        for l in recs:
            c = l.split(b"\t")
            if c[0].decode() == chrom:
                b0, e0 = tabix_reader.vcf_interval(c)
                if b0 < end and e0 > beg:
                    want.append(l)
        got = tabix_reader.fetch_vcf(path, idx, chrom, beg, end)
        assert got == want, (chrom, beg, end, len(got), len(want))
        nonempty += bool(want)
    assert nonempty >= len(regions) // 2


def test_tbi_of_a_multi_contig_gvcf_finds_what_a_scan_finds(tmp_path):
    from genomicsdb_amd import synth
    N = 40
    g = synth.Generator(N, 0, END, contigs=GENOME)
    cells, _ = g.chunk_bytes(END)
    q = helpers.synth_query(tmp_path, N, 0, END - 1, contigs=GENOME)
    text, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=True)
    assert nrec > 3000
    p = str(tmp_path / "out.vcf.gz")
    open(p, "wb").write(_bgzip(text))
    _build(p, False)
    rnd = random.Random(5)
    regions = [("1", 0, 1), ("1", 0, 60000), ("1", 16383, 16385), ("1", 16384, 32768), ("3", 49999, 50000), ("MT", 0, 4000), ("2", 100, 101), ("nope", 0, 10)]
    for _ in range(40):
        name, off, ln = GENOME[rnd.randrange(len(GENOME))]
        b = rnd.randrange(ln)
        regions.append((name, b, min(ln, b + rnd.choice([1, 10, 300, 5000, 40000]))))
    _check_vcf(p, text, regions)


def test_tbi_of_the_reference_goldens(tmp_path):
    """the reference's own golden outputs (3 samples, chr1 positions 12141 .. 8 Mb: several 16 kb windows and bin levels)"""
    for name in ("t0_1_2_combined", "t6_7_8_vcf_at_0"):
        files = [f for f in os.listdir(os.path.join(helpers.GOLDEN, "outputs")) if f.startswith(name)]
        if not files:
            continue
        text = helpers.golden_text(files[0])
        p = str(tmp_path / (files[0] + ".gz"))
        open(p, "wb").write(_bgzip(text, block=700))
        _build(p, False)
        recs = [l.split(b"\t") for l in text.split(b"\n") if l and not l.startswith(b"#")]
        regions = [(recs[0][0].decode(), 0, 300_000_000)]
        for c in recs[::3]:
            b0, e0 = tabix_reader.vcf_interval(c)
            regions += [(c[0].decode(), b0, b0 + 1), (c[0].decode(), max(0, e0 - 1), e0 + 5)]
        _check_vcf(p, text, regions)


def test_index_of_a_file_without_records(tmp_path):
    p = str(tmp_path / "empty.vcf.gz")
    open(p, "wb").write(_bgzip(b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"))
    _build(p, False)
    idx = tabix_reader.Index(p + ".tbi")
    assert idx.names == [] and idx.refs == []


def _bcf_file(records, n_contigs=2):
    """a minimal BCF2 stream: header text with `n_contigs` contig lines, records of (tid, pos, rlen) with empty ID / REF 'N' / no ALT"""
    text = b"##fileformat=VCFv4.2\n" + b"".join(b"##contig=<ID=c%d,length=100000000>\n" % i for i in range(n_contigs))
    text += b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n\x00"
    out = bytearray(b"BCF\x02\x02" + struct.pack("<I", len(text)) + text)
    recs = []
    for tid, pos, rlen in records:
        shared = struct.pack("<iiifII", tid, pos, rlen, 0.0, (1 << 16) | 0, 0) + b"\x07" + b"\x17N" + b"\x00"   # ID, REF, FILTER: typed values
        rec = struct.pack("<II", len(shared), 0) + shared
        recs.append(rec)
        out += rec
    return bytes(out), recs


def _check_bcf(path, recs, meta, regions):
    idx = tabix_reader.Index(path + ".csi")
    assert idx.kind == "csi"
    for tid, beg, end in regions:
        want = [r for r, (t, p, l) in zip(recs, meta) if t == tid and p < end and p + max(1, l) > beg]
        assert tabix_reader.fetch_bcf(path, idx, tid, beg, end) == want, (tid, beg, end)


def test_csi_loffset_is_the_first_record_overlapping_the_bins_first_window(tmp_path):
    """A record that begins in the 16 kb window BEFORE a bin's first window and reaches into it comes first in the file; htslib sets a
    bin's loffset from the overlap-based linear index (update_loff) and hts_itr_query drops chunks that end at or below the loffset of
    the leaf bin of the region's begin.  With loffset = first record BEGINNING in the bin, the reaching record's chunk was dropped:
    a gVCF reference block (pos 16 000, 1 000 long) was missed by a query at 16 400 once a second record began at 16 500."""
    meta = [(0, 16000, 1000), (0, 16500, 10)]
    data, recs = _bcf_file(meta)
    p = str(tmp_path / "a.bcf")
    open(p, "wb").write(_bgzip(data, block=70))
    _build(p, True)
    idx = tabix_reader.Index(p + ".csi")
    leaf = ((1 << 15) - 1) // 7 + (16400 >> 14)
    vrec0 = [c for b in idx.refs[0]["bins"] if b != idx.meta_bin for c in idx.refs[0]["bins"][b]]
    assert idx.refs[0]["loffset"][leaf] == min(c[0] for c in vrec0)          # = the reaching record's offset, not the second record's
    _check_bcf(p, recs, meta, [(0, 16400, 16450), (0, 16384, 16385), (0, 16999, 17000), (0, 17000, 17001), (0, 0, 16000), (1, 0, 10**8)])


def test_csi_of_random_intervals_finds_what_a_scan_finds(tmp_path):
    rnd = random.Random(5)
    meta, pos = [], {0: 0, 1: 0, 2: 0}
    for tid in (0, 1, 2):
        p0 = 0
        for _ in range(1500):
            p0 += rnd.choice([1, 1, 3, 40, 700, 9000, 20000, 140000])
            meta.append((tid, p0, rnd.choice([1, 1, 2, 150, 2000, 17000, 70000, 300000])))
    data, recs = _bcf_file(meta, n_contigs=4)
    p = str(tmp_path / "r.bcf")
    open(p, "wb").write(_bgzip(data, block=900))
    _build(p, True)
    regions = []
    for _ in range(400):
        t, b, l = meta[rnd.randrange(len(meta))]
        at = rnd.choice([b, b + l - 1, b + l, max(0, b - 1), b + l // 2])
        regions.append((t, at, at + rnd.choice([1, 5, 20000])))
    regions += [(3, 0, 10**8), (0, 0, 10**8)]
    _check_bcf(p, recs, meta, regions)


@pytest.mark.gpu
def test_gt_mpi_gather_writes_tbi_and_csi(tmp_path):
    """query JSON with "index_output_VCF": true: gt_mpi_gather -O z leaves <file>.tbi, -O b leaves <file>.csi; regions fetched through them
    = what a scan of the inflated file finds (VCF: lines; BCF2: records by CHROM index / POS / rlen)"""
    import gzip
    from golden_cases import CASES
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, _ = helpers.query_json(callsets, vid, ov, mode)
    ws = tmp_path / "ws"
    (ws / "t0_1_2").mkdir(parents=True)
    (ws / "t0_1_2" / "cells.bin").write_bytes(cells)
    q.update(workspace=str(ws), array="t0_1_2", index_output_VCF=True)
    tool = os.path.join(helpers.ROOT, "genomicsdb_amd", "gt_mpi_gather")
    for fmt, ext in (("z", ".tbi"), ("b", ".csi")):
        out = str(tmp_path / ("out_" + fmt))
        q["vcf_output_filename"] = out
        (tmp_path / "q.json").write_text(json.dumps(q))
        r = subprocess.run([tool, "-j", str(tmp_path / "q.json"), "-O", fmt, "-p", "300", "--produce-Broad-GVCF"], capture_output=True, timeout=300)
        assert r.returncode == 0 and b"WARNING" not in r.stderr, r.stderr.decode()[-1500:]
        assert os.path.exists(out + ext)
        plain = gzip.decompress(open(out, "rb").read())
        idx = tabix_reader.Index(out + ext)
        if fmt == "z":
            assert plain == helpers.golden_text(golden)
            recs = [l.split(b"\t") for l in plain.split(b"\n") if l and not l.startswith(b"#")]
            regions = [("1", 0, 300_000_000)] + [("1", tabix_reader.vcf_interval(c)[0], tabix_reader.vcf_interval(c)[0] + 1) for c in recs]
            _check_vcf(out, plain, regions)
        else:
            assert idx.kind == "csi" and idx.min_shift == 14 and idx.depth == 5
            l_text = struct.unpack_from("<I", plain, 5)[0]
            at, recs = 9 + l_text, []
            while at < len(plain):
                l_shared, l_indiv = struct.unpack_from("<II", plain, at)
                recs.append(plain[at:at + 8 + l_shared + l_indiv])
                at += 8 + l_shared + l_indiv
            assert len(recs) == helpers.golden_text(golden).count(b"\n") - sum(1 for l in helpers.golden_text(golden).split(b"\n") if l.startswith(b"#"))
            for beg, end in [(0, 300_000_000)] + [(struct.unpack_from("<i", r, 12)[0], struct.unpack_from("<i", r, 12)[0] + 1) for r in recs]:
                want = [r for r in recs if struct.unpack_from("<i", r, 12)[0] < end and struct.unpack_from("<i", r, 12)[0] + max(1, struct.unpack_from("<i", r, 16)[0]) > beg]
                assert tabix_reader.fetch_bcf(out, idx, 0, beg, end) == want, (beg, end)
