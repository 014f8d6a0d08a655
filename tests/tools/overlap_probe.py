"""Do two engines on one GPU (two host threads, two streams) overlap usefully?  c2-shaped windows, pages left in HBM.
usage: python tests/tools/overlap_probe.py [window_bp] [windows_per_engine]"""
import sys, os, time, tempfile, threading
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import genomicsdb_amd, helpers
from genomicsdb_amd import synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N, B = 1000, 10_000_000
engines = []
for i in range(2):
    tmp = tempfile.mkdtemp()
    b0 = B + i * W * K
    q = helpers.synth_query(tmp, N, b0, b0 + W * K - 1)
    e = genomicsdb_amd.CombineEngine(q)
    g = synth.Generator(N, b0, W * K + 3000)
    e.stage_cells_begin()
    for w in range(K):
        ptr, nbytes, nc = g.next_chunk(b0 + (w + 1) * W + (3000 if w == K - 1 else 0))
        e.stage_cells_append(ptr, nbytes)
    e.stage_cells_end()
    e.set_reference(b0, synth.reference(b0, W * K + 8000))
    engines.append((e, b0))
arena = 30 << 30

def run(i, out):
    e, b0 = engines[i]
    n = 0
    for w in range(K):
        _, st = e.run_interval(b0 + w * W, b0 + (w + 1) * W - 1, arena_bytes=arena, fetch=False)
        n += st.num_records
    out[i] = n

for mode in ("warm", "serial", "threads"):
    out = [0, 0]
    t0 = time.time()
    if mode == "threads":
        ts = [threading.Thread(target=run, args=(i, out)) for i in range(2)]
        [t.start() for t in ts]; [t.join() for t in ts]
    else:
        run(0, out); run(1, out)
    dt = time.time() - t0
    print("%s: %d records in %.1f ms -> %.2f M positions/s" % (mode, sum(out), dt * 1e3, sum(out) / dt / 1e6), flush=True)
