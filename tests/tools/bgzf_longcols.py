"""BGZF kernels on text with LONG columns (the shape of BASELINE configs[4]: PL vectors of hundreds of values per sample, columns of several KB
that repeat among a few classes with an occasional changed value): ratio and kernel time of the anchored text kernel against the byte-level one.
usage (GPU box): python tests/tools/bgzf_longcols.py [values per column] [MB]"""
import os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import genomicsdb_amd
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
MB = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rnd = random.Random(5)
classes = [[rnd.randint(0, 9999) for _ in range(G)] for _ in range(6)]
out = bytearray()
while len(out) < MB << 20:
    rec = bytearray(b"1\t%d\t.\tA\tC,G,T,<NON_REF>\t.\t.\tDP=%d\tGT:GQ:PL:DP" % (rnd.randint(1, 10**8), rnd.randint(0, 10**5)))
    for s in range(200):
        v = list(rnd.choice(classes))
        if rnd.random() < 0.3:
            v[rnd.randrange(G)] = rnd.randint(0, 9999)
        rec += b"\t./.:%d:" % rnd.choice([0, 20, 50, 99]) + b",".join(b"%d" % x for x in v) + b":%d" % rnd.randint(10, 60)
    out += rec + b"\n"
data = bytes(out)
for text in (True, False):
    best = None
    for _ in range(3):
        comp, ms = genomicsdb_amd.bgzf_compress(data, vcf_text=text)
        best = ms if best is None else min(best, ms)
    print("%-10s kernel: %d values per column, %.1f MB: ratio %.2f, %.2f ms = %.0f GB/s of input" % ("text" if text else "byte-level", G, len(data) / 1e6, len(data) / len(comp), best, len(data) / 1e9 / (best * 1e-3)))
print("zlib -6 on the first 8 MB: ratio %.2f" % (8e6 / len(zlib.compress(data[:8_000_000], 6))))
