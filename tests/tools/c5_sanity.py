"""BASELINE.json configs[4]-style stress at a size that fits a quick run: every sample starts an insertion from a pool of 64
alleles every 50 columns (PL vectors of ~1 800 genotypes per call at the hot sites)."""
import sys, os, tempfile, time
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import genomicsdb_amd, helpers
from genomicsdb_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
B = 10_000_000
tmp = tempfile.mkdtemp()
q = helpers.synth_query(tmp, N, B, B + L - 1)
q["max_diploid_alt_alleles_that_can_be_genotyped"] = 64
eng = genomicsdb_amd.CombineEngine(q)
g = synth.Generator(N, B, L + 3000, dense=(B, L, 50, 64))
ptr, nbytes, nc = g.next_chunk(B + L + 3000)
eng.stage_cells_begin(); eng.stage_cells_append(ptr, nbytes); eng.stage_cells_end()
eng.set_reference(B, synth.reference(B, L + 8000))
for rep in range(2):
    t0 = time.time()
    _, st = eng.run_interval(B, B + L - 1, arena_bytes=32 << 30, fetch=False)
    dt = time.time() - t0
    print("rep %d: %.1f ms; sweep %.2f site %.2f size %.2f write %.2f; records %d, %.2f GB out (%.0f KB/record), %.2f GB/s, %.0f positions/s, %.3g remap elements/s" % (
        rep, dt * 1e3, st.ms_sweep, st.ms_site, st.ms_size, st.ms_write, st.num_records, st.bytes_out / 1e9, st.bytes_out / max(1, st.num_records) / 1e3, st.bytes_out / dt / 1e9, st.num_records / dt,
        st.num_remap_elements / dt))
