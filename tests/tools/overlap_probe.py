"""How much of a step overlaps when two engines (two device pipelines, a compute stream each) work on alternate windows of the
same partition: the sizing kernels wait for loads, the page kernel for the memory system - different resources.
usage: python tests/tools/overlap_probe.py [engines] [steps]"""
import os, sys, time, tempfile, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, genomicsdb_amd, helpers
from genomicsdb_amd import synth
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, B, W, L = 1000, 10_000_000, 1_000_000, 10_000_000
arena = int(os.environ.get("ARENA_MB", "49152")) << 20
tmp = tempfile.mkdtemp(prefix="gdbamd_ovl_")
q = helpers.synth_query(tmp, N, B, B + L - 1)
need = W * min(10, K + E)
engs = []
for e in range(E):
    eng = genomicsdb_amd.CombineEngine(q, device=0)
    gen = synth.Generator(N, B, L)
    eng.stage_cells_begin()
    col = B
    while col < B + need:
        col = min(B + need, col + 1_000_000)
        ptr, nbytes, nc = gen.next_chunk(col)
        eng.stage_cells_append(ptr, nbytes)
    eng.stage_cells_end()
    eng.set_reference(B, synth.reference(B, need + 4096))
    engs.append(eng)
wins = [(B + (i % (need // W)) * W, B + (i % (need // W)) * W + W - 1) for i in range(K + E)]
for e in range(E): engs[e].run_interval(*wins[e], arena_bytes=arena, fetch=False)
torch.cuda.synchronize()
recs = [0] * E
def work(e):
    for i in range(E + e, K + E, E):
        _, st = engs[e].run_interval(*wins[i], arena_bytes=arena, fetch=False)
        recs[e] += st.num_records
t0 = time.time()
th = [threading.Thread(target=work, args=(e,)) for e in range(E)]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
dt = time.time() - t0
print("engines %d steps %d: %.2f ms per step, %.3e positions/s; HBM in use %.1f GB" % (E, K, dt / K * 1e3, sum(recs) / dt, (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9), flush=True)
