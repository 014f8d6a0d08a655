"""BASELINE.json configs[3] shape at a size a test can check: genome-mode generator (several contigs), column partitions whose
begins balance the cell counts, one rank per partition, paged ordered concat to rank 0, compared there byte for byte with the
CPU oracle run over the same partitioning.  Launch: python -m torch.distributed.run --nproc-per-node P tests/tools/c4_sanity.py
(GDBAMD_DIST_BACKEND=gloo lets the ranks share one GPU; pages then travel through host tensors).
The partitions come from the reference's own recipe (gt_mpi_gather --produce-histogram): ColumnHistogramOperator counted on the device, cut by
equi_partition_and_print_bins (dist.balanced_partition).
usage: c4_sanity.py [n_samples] [page_bytes] [contig length divisor]"""
import json, os, sys, tempfile, time
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import torch
import torch.distributed as dist

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
PAGE = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
DIV = int(sys.argv[3]) if len(sys.argv) > 3 else 1
_LENS = [("1", 6000), ("2", 900), ("3", 5000), ("X", 2500), ("Y", 1700), ("MT", 400)]
GENOME, END = [], 0
for _n, _l in _LENS:
    GENOME.append((_n, END, max(2, _l // DIV)))
    END += max(2, _l // DIV)


def main():
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("GDBAMD_DIST_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    dev = int(os.environ.get("LOCAL_RANK", "0")) % ndev
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=backend)
    import genomicsdb_amd, helpers
    from genomicsdb_amd import synth, dist as gdist
    tmp = tempfile.mkdtemp(prefix="c4_%d_" % rank)
    g = synth.Generator(N, 0, END, contigs=GENOME)
    cells, nc = g.chunk_bytes(END)                      # every rank generates the same array and keeps its own partition
    q = helpers.synth_query(tmp, N, 0, END - 1, contigs=GENOME)
    eng = genomicsdb_amd.CombineEngine(q, device=dev)
    eng.stage_cells(cells)
    eng.set_reference(0, synth.reference(0, END + 16))
    # partitions of about equal cell count, from the histogram of the cells' begin columns (one column per bin) taken on the GPU
    counts = eng.column_histogram(0, END - 1, 1)
    begins = sorted(set(gdist.balanced_partition(counts, 0, 1, r, world)[0] for r in range(world)))
    # with few ranks partition 1 is made to begin exactly at a contig offset
    if 1 < len(begins) <= 3:
        begins[1] = min((o for _, o, _ in GENOME), key=lambda o: (abs(o - begins[1]), o)) or begins[1]
        begins = sorted(set(begins))
    parts = [(b, (begins[i + 1] - 1) if i + 1 < len(begins) else END - 1) for i, b in enumerate(begins)]
    mine = parts[rank] if rank < len(parts) else None
    cells_per_part = [int(counts[b:e + 1].sum()) for b, e in parts]
    got = []
    host = backend != "nccl"

    def my_pages():
        if mine is None:
            return
        for t in eng.page_tensors(mine[0], mine[1], PAGE):
            yield t.cpu() if host else t
    t0 = time.time()
    total = gdist.paged_concat(my_pages(), lambda t: got.append(bytes(t.cpu().numpy().tobytes())), PAGE, dst=0, ring_slots=3,
                               device=None if host else torch.device("cuda", dev))
    dt = time.time() - t0
    ok = True
    if rank == 0:
        q2 = dict(q)
        q2["query_column_ranges"] = [[[b, e] for b, e in parts]]
        want, nrec, _ = helpers.oracle_run_synth(q2, cells, synth.SEED, with_header=False)
        body = b"".join(got)
        ok = body == want
        chroms = []
        for l in body.split(b"\n"):
            c = l.split(b"\t", 1)[0]
            if l and (not chroms or chroms[-1] != c):
                chroms.append(c)
        print(json.dumps({"ok": ok, "ranks": world, "backend": backend, "partitions": parts, "records": nrec, "bytes": len(body), "pages": len(got),
                          "max_page": max(len(p) for p in got) if got else 0, "contigs_in_order": [c.decode() for c in chroms], "seconds": dt, "cells_per_partition": cells_per_part, "cells": int(counts.sum()),
                          "largest_bin": int(counts.max())}), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)


main()
