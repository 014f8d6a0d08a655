import ctypes, sys, tempfile, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import genomicsdb_amd, helpers
from genomicsdb_amd import synth, _lib
N, B, L = 1000, 10_000_000, 200_000
tmp = tempfile.mkdtemp()
q = helpers.synth_query(tmp, N, B, B + 3 * L - 1)
eng = genomicsdb_amd.CombineEngine(q)
g = synth.Generator(N, B, 3 * L + 3000)
ptr, nbytes, nc = g.next_chunk(B + 3 * L + 3000)
eng.stage_cells_begin(); eng.stage_cells_append(ptr, nbytes); eng.stage_cells_end()
eng.set_reference(B, synth.reference(B, 3 * L + 8000))
for rep in range(2):
    for w in (0, 1, 2):
        t0 = time.time()
        _, st = eng.run_interval(B + w * L, B + w * L + L - 1, arena_bytes=16 << 30, fetch=False)
        print("rep %d window %d: %.1f ms wall; sweep %.2f site %.2f size %.2f write %.2f; records %d types %d slots %d" % (rep, w, (time.time() - t0) * 1e3, st.ms_sweep, st.ms_site, st.ms_size, st.ms_write, st.num_records, st.num_record_types, st.num_text_slots))
