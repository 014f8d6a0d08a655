"""PCIe-inclusive rate of the boundary that hands over HOST buffers: the same window with the pages copied to a host buffer
(gdbamd_engine_run_interval with host_out != NULL) next to the HBM-resident run bench.py reports."""
import ctypes, sys, tempfile, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import genomicsdb_amd, helpers
from genomicsdb_amd import synth, _lib
N, B, L = 1000, 10_000_000, 200_000
tmp = tempfile.mkdtemp()
q = helpers.synth_query(tmp, N, B, B + 2 * L - 1)
eng = genomicsdb_amd.CombineEngine(q)
g = synth.Generator(N, B, 2 * L + 3000)
ptr, nbytes, nc = g.next_chunk(B + 2 * L + 3000)
eng.stage_cells_begin(); eng.stage_cells_append(ptr, nbytes); eng.stage_cells_end()
eng.set_reference(B, synth.reference(B, 2 * L + 8000))
Lb = _lib.lib()
cap = 10 << 30
buf = ctypes.create_string_buffer(cap)
st = _lib.IntervalStats(); n = ctypes.c_uint64()
for rep in range(3):
    for fetch in (False, True):
        t0 = time.time()
        rc = Lb.gdbamd_engine_run_interval(eng._e, B + (rep % 2) * L, B + (rep % 2) * L + L - 1, 16 << 30, buf if fetch else None, cap if fetch else 0, ctypes.byref(n), ctypes.byref(st))
        dt = time.time() - t0
        assert rc == 0
        print("rep %d fetch=%s: %.1f ms, %d records, %.2f GB -> %.2f M positions/s, %.1f GB/s" % (rep, fetch, dt * 1e3, st.num_records, st.bytes_out / 1e9, st.num_records / dt / 1e6, st.bytes_out / dt / 1e9))
