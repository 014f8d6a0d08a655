"""device BGZF compression of the c2 text: ratio and kernel throughput (GPU box).  usage: bgzf_bench.py [samples] [bp] [formats, e.g. z or -,z,bu,b]"""
import os, sys, tempfile, time, gzip
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import genomicsdb_amd, helpers
from genomicsdb_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
B = 10_000_000
tmp = tempfile.mkdtemp()
q = helpers.synth_query(tmp, N, B, B + L - 1)
g = synth.Generator(N, B, L + 3000)
ptr, nbytes, nc = g.next_chunk(B + L + 3000)
FORMATS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["", "z", "bu", "b"]   # ("-" = VCF text)
for fmt in ["" if f == "-" else f for f in FORMATS]:
    eng = genomicsdb_amd.CombineEngine(q, output_format=fmt)
    eng.stage_cells_begin(); eng.stage_cells_append(ptr, nbytes); eng.stage_cells_end()
    eng.set_reference(B, synth.reference(B, L + 4096))
    eng.run_interval(B, B + L - 1, arena_bytes=8 << 30, fetch=False)
    t = time.time()
    _, st = eng.run_interval(B, B + L - 1, arena_bytes=8 << 30, fetch=False)
    dt = time.time() - t
    line = "format %-3r records %d bytes_out %.3f GB  ms_total %.1f wall %.1f ms" % (fmt, st.num_records, st.bytes_out / 1e9, st.ms_total, dt * 1e3)
    if st.bytes_compressed:
        line += "  compressed %.3f GB ratio %.2f  compress kernels %.1f ms = %.0f GB/s of input" % (st.bytes_compressed / 1e9, st.bytes_out / st.bytes_compressed, st.ms_compress, st.bytes_out / 1e9 / (st.ms_compress * 1e-3))
    print(line, flush=True)
    eng.close()
