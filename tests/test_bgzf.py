"""BGZF output (vcf_output_format "z" / "b"; reference vcf_adapter.cc:340-372 writes them through htslib).  The blocks are
deflated on the device (kernels/gdb_bgzf.hip); the compressed bytes are this build's own, so parity is defined on what a reader
sees: every block is a well-formed BGZF block (gzip member, 'BC' extra field with the block size, CRC-32 and input size right),
the file ends with the EOF block, and the inflated stream equals the uncompressed stream ("" / "bu") byte for byte."""
import gzip
import os
import random
import struct
import subprocess
import zlib

import pytest

import helpers
from golden_cases import CASES

EOF_BLOCK = bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def bgzf_blocks(data):
    """[(compressed offset, inflated bytes)] of a BGZF stream, every field of every block checked (SAM specification 4.1)"""
    out, at = [], 0
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04", "not a gzip member with an extra field at %d" % at
        xlen = struct.unpack_from("<H", data, at + 10)[0]
        assert xlen == 6 and data[at + 12:at + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", data, at + 16)[0] + 1
        assert at + bsize <= len(data)
        payload = data[at + 18:at + bsize - 8]
        crc, isize = struct.unpack_from("<II", data, at + bsize - 8)
        d = zlib.decompressobj(-15)
        raw = d.decompress(payload)
        assert d.eof and d.unused_data == b"", "the DEFLATE stream does not end with the block"
        assert len(raw) == isize and isize <= 65536 and zlib.crc32(raw) == crc
        out.append((at, raw))
        at += bsize
    return out


def test_host_side_blocks_and_eof_marker():
    """the header goes through zlib on the host; the EOF marker is the one every BGZF file ends with (an empty fixed-Huffman block)"""
    blocks = bgzf_blocks(EOF_BLOCK)
    assert blocks == [(0, b"")]
    with gzip.open(os.path.join(helpers.GOLDEN, "inputs", "vcfs", "t0.vcf.gz"), "rb") as f:
        assert len(f.read()) > 1000          # (python's gzip reads the multi-member files htslib's bgzip writes)


@pytest.fixture(scope="module")
def gdb():
    import genomicsdb_amd
    from genomicsdb_amd import _lib
    assert _lib.lib().gdb_mi355_device_count() > 0, "no HIP device"
    return genomicsdb_amd


def _check_roundtrip(gdb, data, vcf_text=False):
    comp, ms = gdb.bgzf_compress(data, vcf_text=vcf_text)
    blocks = bgzf_blocks(comp)
    assert b"".join(r for _, r in blocks) == data
    if len(blocks) > 1:
        blk = len(blocks[0][1])
        assert blk in (4096, 6144, 8192, 16384)                            # GDBAMD_BGZF_BLOCK
        assert len(blocks) == (len(data) + blk - 1) // blk and all(len(r) == blk for _, r in blocks[:-1])
    return comp, ms


@pytest.mark.gpu
def test_device_deflate_on_hostile_inputs(gdb):
    rnd = random.Random(11)
    cases = {
        "empty": b"",
        "one byte": b"x",
        "three bytes": b"abc",
        "zeros": bytes(100_000),                                            # matches of 258 at distance 1 .. (overlapping copies)
        "one block exactly": bytes(rnd.getrandbits(8) for _ in range(8192)),    # incompressible: stored block
        "two blocks exactly": bytes(rnd.getrandbits(8) for _ in range(16384)),
        "one block + 1": b"ab" * 4096 + b"c",
        "16 KiB + 1": b"ab" * 8192 + b"c",
        "random": bytes(rnd.getrandbits(8) for _ in range(70_000)),
        "text": b"".join(b"./.:%d:.:%d:0,%d,%d\t" % (rnd.choice([0, 20, 50, 99]), rnd.randint(10, 60), rnd.randint(1, 300), rnd.randint(1, 4000)) for _ in range(9000)),
        "period 255": bytes(range(255)) * 300,
        "long then literal tail": b"q" * 8189 + b"xyz" + b"r" * 8189 + b"uvw",
        "high bytes": bytes(rnd.choice([200, 250, 255, 144, 143]) for _ in range(40_000)),
        "all byte values": bytes(range(256)) * 64 + bytes(reversed(range(256))) * 64,
    }
    for name, data in cases.items():
        comp, _ = _check_roundtrip(gdb, data)
        if name in ("zeros", "period 255"):     # (every block starts with its own 255 literals: small blocks find less)
            assert len(comp) * (20 if int(os.environ.get("GDBAMD_BGZF_BLOCK", "8192")) >= 8192 else 10) < len(data), name
        if name == "text":                                   # (columns of random numbers: little to find besides the separators)
            assert len(comp) * 1.7 < len(data), name
        if name in ("random", "one block exactly", "two blocks exactly"):
            blk = int(os.environ.get("GDBAMD_BGZF_BLOCK", "8192"))
            assert len(comp) <= len(data) + 31 * ((len(data) + blk - 1) // blk), name         # stored blocks: framing only
    with gzip.open(os.path.join(helpers.GOLDEN, "inputs", "chr1_10MB.fasta.gz"), "rb") as f:
        fasta = f.read()[:3_000_000]
    comp, _ = _check_roundtrip(gdb, fasta)
    assert len(comp) < len(fasta) * (0.5 if int(os.environ.get("GDBAMD_BGZF_BLOCK", "8192")) >= 8192 else 0.52)     # (4 KiB blocks: 1.98)


@pytest.mark.gpu
def test_device_deflate_text_kernel_on_hostile_inputs(gdb):
    """the anchored kernel of the "z" stream (k_bgzf_deflate_text: matches begin at tabs / newlines, four wavefronts per block, each a DEFLATE
    block of its own) must give a valid stream for ANY bytes - anchors only decide how much is found: the hostile inputs of the byte-level
    kernel, texts with every anchor density (a tab at every byte ... none at all), high bytes in the literals, blocks cut anywhere"""
    rnd = random.Random(12)
    def entry():
        gq = rnd.choice([0, 20, 50, 99])
        return b"\t./.:%d:.:.:0,%d,%d,%d,%d,%d:%d:%d" % (gq, 3 * gq, 45 * gq, 3 * gq, 45 * gq, 45 * gq, rnd.randint(10, 60), rnd.randint(10, 60))
    def record(i, nsamples):
        return b"1\t%d\t.\tA\t<NON_REF>\t.\t.\tEND=%d\tGT:GQ:SB:AD:PL:MIN_DP:DP" % (10_000_000 + i, 10_000_000 + i) + b"".join(entry() for _ in range(nsamples)) + b"\n"
    cases = {
        "empty": b"",
        "one byte": b"x",
        "one tab": b"\t",
        "three bytes": b"a\tc",
        "zeros": bytes(100_000),
        "tabs only": b"\t" * 50_000,
        "newlines and tabs": b"\t\n" * 30_000,
        "tab every 2": b"\t." * 40_000,
        "tab every 7": b"\t./.:.:" [:7] * 12_000,
        "tab every 9": b"\t12345678" * 9_000,
        "no delimiter at all": bytes(rnd.choice(b"ACGTN") for _ in range(60_000)),
        "colons only": b":".join(b"%d" % rnd.randint(0, 99999) for _ in range(12_000)),
        "long columns": b"".join(b"\t" + b",".join(b"%d" % rnd.randint(0, 5000) for _ in range(rnd.randint(1, 120))) for _ in range(800)),
        "vcf records": b"".join(record(i, 300) for i in range(12)),
        "vcf records, blocks cut anywhere": b"xy" + b"".join(record(i, 57) for i in range(60)),
        "one block exactly": bytes(rnd.getrandbits(8) for _ in range(8192)),
        "random": bytes(rnd.getrandbits(8) for _ in range(70_000)),
        "random with tabs": bytes(rnd.choice([9, 10, 58, 44]) if rnd.random() < 0.1 else rnd.getrandbits(8) for _ in range(70_000)),
        "high bytes in columns": b"".join(b"\t" + bytes(rnd.choice([200, 250, 255, 144, 143, 65]) for _ in range(rnd.randint(0, 40))) for _ in range(4_000)),
        "repeated column": b"\t./.:99:.:.:0,297,4455,297,4455,4455:34:55" * 3_000,
        "columns of 95+ bytes": (b"\t" + b"q" * 300) * 200,
        # wide-cohort columns: secondary anchors inside a column continue the anchor in front, the matches merge into one token and
        # runs are cut where they would pass 258 bytes
        "long PL columns repeated": b"".join((b"\t./.:99:.:.:" + b",".join(b"%d" % (37 * k % 5000) for k in range(g)) + b":34:55") * 3 for g in (21, 28, 36, 45, 66, 120)) * 40,
        "long PL columns, one value changed": b"".join(b"\t./.:99:.:.:" + b",".join(b"%d" % ((37 * k + (i == k) * 7) % 5000) for k in range(45)) + b":34:55" for i in range(300)),
        "period 255": bytes(range(255)) * 300,
        "8189 + 3": b"q" * 8189 + b"x\tz" + b"r" * 8189 + b"u\tw",
    }
    for name, data in cases.items():
        comp, _ = _check_roundtrip(gdb, data, vcf_text=True)
        if name in ("vcf records", "vcf records, blocks cut anywhere", "repeated column", "long PL columns repeated", "long PL columns, one value changed"):
            assert len(comp) * 3 < len(data), (name, len(comp), len(data))
        if name in ("random", "one block exactly"):
            assert len(comp) <= len(data) + 31 * ((len(data) + 8191) // 8192), name            # stored blocks: framing only
    with gzip.open(os.path.join(helpers.GOLDEN, "inputs", "chr1_10MB.fasta.gz"), "rb") as f:
        fasta = f.read()[:1_000_000]
    _check_roundtrip(gdb, fasta, vcf_text=True)
    for g in sorted(os.listdir(os.path.join(helpers.GOLDEN, "outputs")))[:20]:                 # the reference's golden VCFs as they are
        with open(os.path.join(helpers.GOLDEN, "outputs", g), "rb") as f:
            _check_roundtrip(gdb, f.read(), vcf_text=True)


@pytest.mark.gpu
def test_device_deflate_text_kernel_random_mixtures(gdb):
    """differential fuzz of the anchored kernel: 150 buffers of random length put together from random pieces - VCF columns that repeat with
    small changes, runs of one byte, random bytes, delimiters at random densities, long comma-separated vectors repeated exactly (matches that
    continue each other and merge, cut at 258) - each must inflate to itself through both kernels, and the compressed bytes of the text
    kernel must be the same on a second run (four wavefronts OR into one LDS image: the order they arrive in must not matter)"""
    rnd = random.Random(20260930)
    def piece():
        k = rnd.randrange(8)
        if k == 0:
            gq = rnd.choice([0, 20, 50, 99])
            return b"".join(b"\t./.:%d:.:.:0,%d,%d:%d:%d" % (gq, 3 * gq, 45 * gq, rnd.randint(10, 60), rnd.randint(10, 60)) for _ in range(rnd.randint(1, 400)))
        if k == 1:
            return bytes([rnd.randrange(256)]) * rnd.randint(1, 3000)
        if k == 2:
            return bytes(rnd.getrandbits(8) for _ in range(rnd.randint(1, 2000)))
        if k == 3:
            dens = rnd.choice([0.5, 0.1, 0.02, 0.005])
            return bytes(rnd.choice(b"\t\n:,") if rnd.random() < dens else rnd.choice(b"0123456789ACGT./|") for _ in range(rnd.randint(1, 5000)))
        if k == 4:
            col = b"\t0/1:" + b",".join(b"%d" % rnd.randint(0, 9999) for _ in range(rnd.randint(3, 150)))
            return col * rnd.randint(2, 12)
        if k == 5:
            col = bytearray(b"\t1|1:" + b",".join(b"%d" % rnd.randint(0, 9999) for _ in range(rnd.randint(10, 80))))
            out = bytearray()
            for _ in range(rnd.randint(2, 10)):
                col[rnd.randrange(len(col))] = rnd.choice(b"0123456789")
                out += col
            return bytes(out)
        if k == 6:
            return b"1\t%d\t.\tA\tC,<NON_REF>\t.\t.\tDP=%d;END=%d\tGT:AD:DP:GQ:PL\n" % (rnd.randint(1, 10**8), rnd.randint(0, 10**5), rnd.randint(1, 10**8))
        return bytes(rnd.choice([9, 10, 58, 44, 200, 255, 143, 144, 0]) for _ in range(rnd.randint(1, 300)))
    for it in range(150):
        want = rnd.randint(1, 40000)
        data = bytearray()
        while len(data) < want:
            data += piece()
        data = bytes(data[:want])
        comp, _ = _check_roundtrip(gdb, data, vcf_text=True)
        if it % 5 == 0:
            comp2, _ = gdb.bgzf_compress(data, vcf_text=True)
            assert comp2 == comp, "the text kernel's output depends on timing"
            _check_roundtrip(gdb, data, vcf_text=False)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c[0] in ("t0_1_2_vcf_at_0", "t6_7_8_vcf_at_0", "t0_1_2_all_asa_loading", "info_ops1.vcf", "t0_overlapping")],
                         ids=lambda c: c[0])
def test_golden_streams_as_bgzf(gdb, case):
    """"z": the inflated stream is the reference's golden VCF; "b": it is the "bu" stream, which decodes to the golden"""
    from test_hostsim_golden import DEVICE_UNSUPPORTED
    if case[0] in DEVICE_UNSUPPORTED:
        pytest.skip("not a device configuration")
    name, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    want = helpers.golden_text(golden)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, output_format="z")
    z = s.read()
    s.close()
    assert z.endswith(EOF_BLOCK)
    blocks = bgzf_blocks(z)
    assert blocks[-1][1] == b"" and b"".join(r for _, r in blocks) == want
    assert gzip.decompress(z) == want                      # what `zcat` / htslib see
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, output_format="b")
    b = s.read()
    s.close()
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, output_format="bu")
    bu = s.read()
    s.close()
    assert b.endswith(EOF_BLOCK) and gzip.decompress(b) == bu
    assert helpers.bcf_stream_to_text(bu) == want
    # header only: header block + EOF block
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, output_format="z", produce_header_only=True)
    h = s.read()
    s.close()
    assert h.endswith(EOF_BLOCK) and gzip.decompress(h) == b"".join(l for l in want.splitlines(True) if l.startswith(b"#"))


@pytest.mark.gpu
def test_synthetic_stream_compresses_and_virtual_offsets_work(gdb, tmp_path, monkeypatch):
    """300 samples x 4 kb through small device pages (many pages, each ending in a short block): the inflated stream equals the
    text stream; a virtual offset (compressed block offset << 16 | offset in the block) taken at a record start seeks to that
    record - what a .tbi index stores"""
    from genomicsdb_amd import synth
    N, B, L = 300, 10_000_000, 4000
    g = synth.Generator(N, B, L + 2500)
    cells, _ = g.chunk_bytes(B + L + 2500)
    q = helpers.synth_query(tmp_path, N, B + 100, B + L)
    monkeypatch.setenv("GDBAMD_DEVICE_PAGE_BYTES", str(3 << 20))
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    text = s.read()
    s.close()
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, output_format="z")
    z = s.read()
    st = s.stream_stats()
    s.close()
    assert st.pages > 5
    blocks = bgzf_blocks(z)
    assert b"".join(r for _, r in blocks) == text and z.endswith(EOF_BLOCK)
    assert len(z) * 3.5 < len(text)
    # virtual offsets of the first record of every 50th block that begins one
    upos = 0
    checked = 0
    for coff, raw in blocks:
        nl = raw.find(b"\n")
        if nl >= 0 and nl + 1 < len(raw) and checked < 40:
            voff = (coff << 16) | (nl + 1)
            c, u = voff >> 16, voff & 0xFFFF
            bsize = struct.unpack_from("<H", z, c + 16)[0] + 1
            block = zlib.decompress(z[c + 18:c + bsize - 8], -15)
            line = block[u:].split(b"\n", 1)[0]
            assert text[upos + nl + 1:upos + nl + 1 + len(line)] == line and (line.startswith(b"1\t") or line.startswith(b"#"))
            checked += 1
        upos += len(raw)
    assert checked >= 20


@pytest.mark.gpu
def test_cli_writes_bgzf_files(gdb, tmp_path):
    """gt_mpi_gather -O z / "vcf_output_format": "b" in the query JSON: files a gzip reader opens, ending with the EOF block"""
    import json
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    qj, _ = helpers.query_json(callsets, vid, ov, mode)
    (tmp_path / "ws" / "a").mkdir(parents=True)
    (tmp_path / "ws" / "a" / "cells.bin").write_bytes(helpers.cells_for(callsets, vid))
    qj["workspace"], qj["array"] = str(tmp_path / "ws"), "a"
    qj["vcf_output_filename"] = str(tmp_path / "out.vcf.gz")
    (tmp_path / "q.json").write_text(json.dumps(qj))
    tool = os.path.join(helpers.ROOT, "genomicsdb_amd", "gt_mpi_gather")
    r = subprocess.run([tool, "-j", str(tmp_path / "q.json"), "-O", "z", "--produce-Broad-GVCF"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    z = (tmp_path / "out.vcf.gz").read_bytes()
    assert z.endswith(EOF_BLOCK) and gzip.decompress(z) == helpers.golden_text(golden)
    qj["vcf_output_format"] = "b"
    qj["vcf_output_filename"] = str(tmp_path / "out.bcf")
    (tmp_path / "q.json").write_text(json.dumps(qj))
    r = subprocess.run([tool, "-j", str(tmp_path / "q.json"), "--produce-Broad-GVCF"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    b = (tmp_path / "out.bcf").read_bytes()
    assert b.endswith(EOF_BLOCK) and helpers.bcf_stream_to_text(gzip.decompress(b)) == helpers.golden_text(golden)
