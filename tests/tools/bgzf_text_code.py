"""Generator of the STATIC Huffman code of the "z" stream's text kernel (kernels/gdb_bgzf_text_code.inc).

k_bgzf_deflate_text writes DEFLATE blocks of BTYPE = 10 ("dynamic" Huffman codes, RFC 1951 3.2.7) whose code is the SAME in every
block: tuned once for VCF text - digits, separators and the match lengths / distances of the anchored parse are short - instead of the
fixed code's 8 bits for every literal.  The block header that describes the code is a constant bit string.  This script
  1. runs the kernel's parse model (anchors at tabs / newlines, one match per anchor, literals behind it) over sample texts and counts
     literal / length / distance symbols (every symbol keeps a floor count, so every byte value has a code),
  2. builds length-limited (12 bits) Huffman code lengths by package-merge, the canonical codes, and the header bits
     (HLIT / HDIST / HCLEN, the code-length code, the run-length coded lengths),
  3. checks the result with zlib (a stream using every symbol must inflate), and prints the .inc file.
usage: python tests/tools/bgzf_text_code.py sample1.vcf [sample2.vcf ...] > genomicsdb_amd/csrc/kernels/gdb_bgzf_text_code.inc
(the committed .inc was generated from 18 MB of the c2 workload's text - tests/tools: oracle_run_synth, 1 000 samples x 400 positions -
 and the reference's golden VCFs under tests/golden/outputs; tests/test_bgzf.py re-derives codes and header from the lengths in the .inc)"""
import sys, zlib

MAXBITS = 12

def package_merge(freqs, limit):
    """optimal length-limited prefix code lengths (package-merge); symbols with freq 0 get length 0"""
    syms = [(f, [i]) for i, f in enumerate(freqs) if f > 0]
    n = len(syms)
    if n == 0: return [0] * len(freqs)
    if n == 1:
        out = [0] * len(freqs); out[syms[0][1][0]] = 1; return out
    assert n <= (1 << limit)
    leaves = sorted(syms, key=lambda x: x[0])
    packages = list(leaves)
    for _ in range(limit - 1):
        merged = []
        for i in range(0, len(packages) - 1, 2):
            merged.append((packages[i][0] + packages[i + 1][0], packages[i][1] + packages[i + 1][1]))
        packages = sorted(leaves + merged, key=lambda x: x[0])
    out = [0] * len(freqs)
    for _, items in packages[:2 * n - 2]:
        for s in items: out[s] += 1
    return out

def canonical(lengths):
    """canonical codes (RFC 1951 3.2.2), returned bit-reversed for the LSB-first stream"""
    maxl = max(lengths)
    bl = [0] * (maxl + 2)
    for l in lengths:
        if l: bl[l] += 1
    code = 0; nxt = [0] * (maxl + 2)
    for b in range(1, maxl + 1):
        code = (code + bl[b - 1]) << 1 if b > 1 else 0
        nxt[b] = code
    # (RFC: code = (code + bl_count[bits-1]) << 1 starting with bl_count[0] = 0)
    code = 0; nxt = [0] * (maxl + 2); bl[0] = 0
    for b in range(1, maxl + 1):
        code = (code + bl[b - 1]) << 1
        nxt[b] = code
    out = []
    for l in lengths:
        if l == 0: out.append(0); continue
        c = nxt[l]; nxt[l] += 1
        out.append(int(format(c, "0%db" % l)[::-1], 2))
    return out

def kraft(lengths): return sum(2.0 ** -l for l in lengths if l)

LEN_BASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEN_EXTRA = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DIST_BASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DIST_EXTRA = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
def len_sym(L):
    for i in range(28, -1, -1):
        if L >= LEN_BASE[i]: return i
def dist_sym(D):
    for i in range(29, -1, -1):
        if D >= DIST_BASE[i]: return i

def count_symbols(data, lit, ln, ds):
    """the kernel's parse model: anchors at tabs / newlines (and the block's first byte), 8-byte keys, most recent earlier anchor, one match per anchor"""
    for off in range(0, len(data), 8192):
        buf = data[off:off + 8192]; n = len(buf)
        anch = [0] + [i for i in range(1, n) if buf[i] in b"\t\n"]
        table = {}
        for j, a in enumerate(anch):
            e = anch[j + 1] if j + 1 < len(anch) else n
            key = bytes(buf[a:a + 8]); c = table.get(key); L = 0
            if c is not None:
                mx = min(e - a, 258)
                while L < mx and buf[c + L] == buf[a + L]: L += 1
                if L < 4: L = 0
            if L: ln[len_sym(L)] += 1; ds[dist_sym(a - c)] += 1
            for x in buf[a + L:e]: lit[x] += 1
            if a + 8 <= n: table[key] = a

class Bits:
    def __init__(self): self.v = 0; self.n = 0
    def put(self, val, nb): self.v |= val << self.n; self.n += nb
    def bytes(self): return self.v.to_bytes((self.n + 7) // 8, "little")

CL_ORDER = [16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]
def header_bits(ll, dl):
    """HLIT HDIST HCLEN + code-length code + run-length coded lengths; WITHOUT the three bits BFINAL / BTYPE in front"""
    seq = ll + dl
    rle = []; i = 0
    while i < len(seq):
        v = seq[i]; j = i
        while j < len(seq) and seq[j] == v: j += 1
        run = j - i
        if v == 0:
            while run >= 11: r = min(run, 138); rle.append((18, r - 11, 7)); run -= r
            if run >= 3: rle.append((17, run - 3, 3)); run = 0
            rle += [(0, 0, 0)] * run
        else:
            rle.append((v, 0, 0)); run -= 1
            while run >= 3: r = min(run, 6); rle.append((16, r - 3, 2)); run -= r
            rle += [(v, 0, 0)] * run
        i = j
    clf = [0] * 19
    for s, _, _ in rle: clf[s] += 1
    cll = package_merge(clf, 7)
    clc = canonical(cll)
    hclen = 19
    while hclen > 4 and cll[CL_ORDER[hclen - 1]] == 0: hclen -= 1
    b = Bits()
    b.put(len(ll) - 257, 5); b.put(len(dl) - 1, 5); b.put(hclen - 4, 4)
    for k in range(hclen): b.put(cll[CL_ORDER[k]], 3)
    for s, extra, nb in rle:
        b.put(clc[s], cll[s])
        if nb: b.put(extra, nb)
    return b

def main():
    lit = [0] * 256; ln = [0] * 29; ds = [0] * 30
    for path in sys.argv[1:]:
        count_symbols(open(path, "rb").read(), lit, ln, ds)
    # (digits share their mean: which digits a sample happens to like - the synthetic generator's GQ / PL values - is not a property of VCF text)
    dmean = sum(lit[48:58]) // 10
    for x in range(48, 58): lit[x] = dmean
    tot = sum(lit) + sum(ln)
    floor = max(1, tot // 40000)
    lf = [max(f, floor) for f in lit] + [max(1, tot // 8192)] + [max(f, floor) for f in ln]          # 256 = end of block: once per part
    df = [max(f, max(1, sum(ds) // 2000)) for f in ds[:26]] + [0, 0, 0, 0]                            # distances above 8 192 do not occur in an 8 KiB block
    ll = package_merge(lf, MAXBITS); dl = package_merge(df, MAXBITS)
    assert abs(kraft(ll) - 1.0) < 1e-12 and abs(kraft(dl) - 1.0) < 1e-12 and max(ll) <= MAXBITS and max(dl) <= MAXBITS and min(ll) >= 1
    lc = canonical(ll); dc = canonical(dl)
    hb = header_bits(ll, dl)
    # --- check with zlib: header + every literal + every length / distance symbol that can occur + end of block
    b = Bits(); b.put(1, 1); b.put(2, 2); b.put(hb.v, hb.n)
    want = bytearray()
    for x in range(256): b.put(lc[x], ll[x]); want.append(x)
    for x in range(256): b.put(lc[x], ll[x]); want.append(x)
    for i in range(29):
        L = LEN_BASE[i] + (1 if LEN_EXTRA[i] else 0)
        for dsy in range(0, 18):
            D = DIST_BASE[dsy] + (1 if DIST_EXTRA[dsy] else 0)
            if D > len(want): continue
            b.put(lc[257 + i], ll[257 + i]); b.put(L - LEN_BASE[i], LEN_EXTRA[i])
            b.put(dc[dsy], dl[dsy]); b.put(D - DIST_BASE[dsy], DIST_EXTRA[dsy])
            for _ in range(L): want.append(want[-D])
    b.put(lc[256], ll[256])
    got = zlib.decompressobj(-15).decompress(b.bytes())
    assert got == bytes(want), "zlib does not inflate the test stream"
    avg = sum(lit[x] * ll[x] for x in range(256)) / max(1, sum(lit))
    w = sys.stdout.write
    w("// GENERATED by tests/tools/bgzf_text_code.py - the static Huffman code of k_bgzf_deflate_text (BTYPE = 10 with a constant header).\n")
    w("// sample: %d literals (average code %.2f bits, fixed code: 8+), %d matches; header %d bits; checked with zlib by the generator\n" % (sum(lit), avg, sum(ln), hb.n))
    w("static const uint8_t kTextLitLenBits[286] = {%s};\n" % ",".join(map(str, ll)))
    w("static const uint8_t kTextDistBits[30] = {%s};\n" % ",".join(map(str, dl)))
    hbytes = hb.bytes()
    w("static const uint32_t kTextHeaderNBits = %d;\n" % hb.n)
    w("static const uint8_t kTextHeader[%d] = {%s};\n" % (len(hbytes), ",".join(map(str, hbytes))))
if __name__ == "__main__":
    main()
