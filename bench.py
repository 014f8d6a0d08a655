#!/usr/bin/env python3
"""bench.py - combined-gVCF positions/sec of the scan+combine hot path on MI355X.

Workload (BASELINE.json configs[1], "c2"): 1 000 synthetic-gVCF samples x 10 Mb column interval (generator of
SURVEY.md 8(d), genomicsdb_amd/synth), staged once in HBM as a columnar fragment.  A STEP is one pass of the hot path
over one batch = one --window-bp wide column window (default 1 Mb: the batch is sized for the 288 GB of HBM) of that array (sweep -> site merge -> sizing -> bit-exact VCF text
written into HBM pages); successive steps take successive windows.  `value` = output records (positions) per second
with the input resident in HBM; nothing is copied to the host inside the timed region.

N > 1 (torch.distributed.run, one rank per GPU): every rank owns its own column partition of the same shape
(weak scaling, no data-path collective - the reference's ranks do not communicate either, gt_mpi_gather.cc:322-366);
value = records of all ranks / max-over-ranks time.

Also on the JSON line: "roofline" for the dominant kernel (k_assemble_write, the HBM-bound gather-copy of entry texts into pages) and "cpu_baseline"
(the CPU oracle = port of the reference algorithm, single thread, on a bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--interval-bp", type=int, default=10_000_000)
    ap.add_argument("--window-bp", type=int, default=1_000_000, help="columns per step (one batch); 1 Mb = 44.6 GB of VCF text at 1 000 samples")
    ap.add_argument("--arena-mb", type=int, default=49152, help="HBM page for the output text (one page per window at the defaults)")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("GDBAMD_BENCH_LANES", "3")),
                    help="windows in flight at a time (gdbamd_engine_run_intervals): lane l takes steps l, l + lanes, ... on a device pipeline of its own over the same "
                         "staged fragment; the sweep / site / sizing kernels of one window overlap with the page kernel of another.  Needs lanes x (arena + ~12 GB) of HBM")
    ap.add_argument("--bcf", action="store_true", help="pages of BCF2 records (output format \"bu\") instead of VCF text; not the headline configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream-input", action="store_true",
                    help="c3 shape: the array is NOT resident - its cells pass through HBM in column windows of the staging budget "
                         "(GDBAMD_STAGE_BUDGET_MB) with carry-over; one pass over --interval-bp, --steps / --warmup are ignored")
    ap.add_argument("--stream-source", default="callback", help="--stream-input: \"callback\" (the generator inside the timed region) or \"memory\" (generated first)")
    ap.add_argument("--no-stream", action="store_true", help="skip the end-to-end leg through the query stream (gdb_mi355_read)")
    ap.add_argument("--no-alone-pass", action="store_true", help="lanes > 1: skip the three extra steps that time the page kernel with one window in flight")
    ap.add_argument("--no-c3", action="store_true", help="skip the c3-shape leg (10 000 samples streamed through HBM in column windows)")
    ap.add_argument("--c3-bp", type=int, default=10_000_000, help="columns of the c3-shape leg (10 000 samples: 14.6 GB of cells per Mb, generated into host memory first)")
    ap.add_argument("--cpu-sample-bp", type=int, default=12000, help="columns of the bounded CPU-baseline sample (~15 s of oracle time)")
    ap.add_argument("--base", type=int, default=10_000_000, help="first column of rank 0's partition")
    ap.add_argument("--c3-full", action="store_true",
                    help="BASELINE.json configs[2] at its stated size: 10 000 samples x whole chr1 (249 250 621 bp, ~3.6 TB of cells) streamed through HBM in one pass "
                         "(= --samples 10000 --interval-bp 249250621 --window-bp 50000 --stream-input); ~15 minutes, most of it the synthetic generator")
    ap.add_argument("--concat", action="store_true",
                    help="N > 1: after the timed steps every rank's pages of one window go to rank 0 in column order (dist.gather_interval_paged: the single-stream "
                         "view of the partitions' outputs); the line carries \"concat\": bytes, GB/s, how long the root and the senders were blocked")
    ap.add_argument("--concat-format", default="z", help="output format of the --concat leg's engine: \"z\" (BGZF blocks, ~6 x fewer bytes over the links) or \"\"")
    ap.add_argument("--dry-run", action="store_true",
                    help="rank set-up, partition arithmetic and the cross-rank reduction only, no device work and no number: "
                         "lets the CPU suite check the N-rank launch (the line says \"dry_run\": true and carries value null)")
    args = ap.parse_args()
    if args.c3_full:
        args.samples, args.interval_bp, args.window_bp, args.stream_input, args.base = 10000, 249_250_621, 50_000, True, 0   # (contig 1 of the vid: columns 0 .. 249 250 620)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare (`python bench.py --gpus N`): become N ranks, one per GPU, like `mpirun -n N gt_mpi_gather`
        # (gt_mpi_gather.cc:415-425: rank -> column partition); the ranks' stdout is this process's stdout
        return relaunch_ranks(args.gpus)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = max(1, torch.cuda.device_count())
    device_index = local_rank % ndev   # one GPU per rank on a full node; wraps only in the single-GPU smoke run (gloo)
    backend = os.environ.get("GDBAMD_DIST_BACKEND", "nccl")   # "nccl" = RCCL; "gloo" lets two ranks share one GPU in a smoke test
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
    if args.dry_run:
        return dry_run(args, rank, world, backend)
    torch.cuda.set_device(device_index)

    import genomicsdb_amd
    from genomicsdb_amd import synth
    import helpers

    N, Lbp, W = args.samples, args.interval_bp, args.window_bp
    W = max(1, min(W, Lbp))
    from genomicsdb_amd import dist as gdist
    if args.stream_input:
        return run_streamed(args, rank, world, device_index, backend, source=args.stream_source)
    B, _ = gdist.synthetic_partition(rank, args.base, Lbp)  # every rank scans its own column partition of the same shape
    nwin = max(1, Lbp // W)
    total_steps = args.steps + args.warmup
    need_bp = min(Lbp, W * min(nwin, total_steps))  # stage only the windows the run will touch
    tmp = tempfile.mkdtemp(prefix="gdbamd_bench_")
    q = helpers.synth_query(tmp, N, B, B + Lbp - 1)
    eng = genomicsdb_amd.CombineEngine(q, device=device_index, is_bcf=args.bcf)

    # ---- generate + stage (not timed): cells -> columnar fragment in HBM, in 1 Mb parts ---------------------------------
    t0 = time.time()
    gen = synth.Generator(N, B, Lbp)
    eng.stage_cells_begin()
    ncells = 0
    col = B
    while col < B + need_bp:
        col = min(B + need_bp, col + 1_000_000)
        ptr, nbytes, nc = gen.next_chunk(col)
        eng.stage_cells_append(ptr, nbytes)
        ncells += nc
    eng.stage_cells_end()
    ref = synth.reference(B, need_bp + 4096)
    eng.set_reference(B, ref)
    t_stage = time.time() - t0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    arena = args.arena_mb << 20
    windows = [(B + (i % (need_bp // W)) * W, B + (i % (need_bp // W)) * W + W - 1) for i in range(total_steps)]
    lanes = max(1, min(4, args.lanes))
    if lanes > 1:
        arena = min(arena, int(os.environ.get("GDBAMD_BENCH_LANE_ARENA_MB", "46080")) << 20)   # (two 48 GiB pages + two entry tables do not fit beside the fragment)
        # a lane needs its page arena and its entry table / matrix / sweep buffers (~14 GB at c2's width; scaled by the window for others):
        # no more lanes than the free HBM holds
        free_b, _ = torch.cuda.mem_get_info(device_index)
        per_lane = min(arena, 45 * N * W + (1 << 30)) + 18 * N * W + (2 << 30)     # (~45 bytes of text and up to ~18 bytes of tables per sample and position)
        while lanes > 1 and lanes * per_lane + (4 << 30) > free_b:
            lanes -= 1
        # every lane has to have run one full window before the clock starts: its pipeline is created, adopts and classifies the staged
        # fragment and sizes its grow-only buffers then (one-off work of the kind staging is).  With W >= lanes the W warm-up steps do that;
        # else the missing ones are run here, untimed and reported as "untimed_lane_preparation_steps".
        prep = lanes if args.warmup < lanes else 0
        if prep:
            eng.run_intervals([windows[i % len(windows)] for i in range(lanes)], arena_bytes=arena, lanes=lanes)
        if args.warmup:
            eng.run_intervals(windows[:args.warmup], arena_bytes=arena, lanes=lanes)
    else:
        for i in range(args.warmup):
            eng.run_interval(windows[i][0], windows[i][1], arena_bytes=arena, fetch=False)
    barrier()
    t1 = time.time()
    recs = cells_in = bytes_out = bytes_in = 0
    ms = {"sweep": 0.0, "site": 0.0, "size": 0.0, "write": 0.0}
    wk_ms = wk_launches = 0.0
    lane_stats = eng.run_intervals(windows[args.warmup:total_steps], arena_bytes=arena, lanes=lanes) if lanes > 1 else None
    for i in range(args.warmup, total_steps):
        if lane_stats is not None:
            st = lane_stats[i - args.warmup]
        else:
            _, st = eng.run_interval(windows[i][0], windows[i][1], arena_bytes=arena, fetch=False)
        recs += st.num_records
        cells_in += st.num_cells_in_window
        bytes_out += st.bytes_out
        ms["sweep"] += st.ms_sweep; ms["site"] += st.ms_site; ms["size"] += st.ms_size; ms["write"] += st.ms_write
        wk_ms += st.ms_write_kernel_avg * st.write_launches
        wk_launches += st.write_launches
    barrier()
    dt = time.time() - t1
    # With several windows in flight a launch of the page kernel shares the device with the other lanes' kernels: its duration (what
    # `roofline` is computed from, as everywhere) says how long it was resident, not how fast it can go.  A short pass of the same
    # windows one at a time, OUTSIDE the timed region, measures the kernel on its own; it goes into roofline["alone"].
    alone = None
    if lanes > 1 and not args.no_alone_pass:
        a_ms = a_n = 0.0
        for i in range(args.warmup, min(total_steps, args.warmup + 3)):
            _, st1 = eng.run_interval(windows[i][0], windows[i][1], arena_bytes=arena, fetch=False)
            a_ms += st1.ms_write_kernel_avg * st1.write_launches
            a_n += st1.write_launches
        if a_n:
            alone = a_ms / a_n
    # bytes_in: reference binary-cell bytes of the begin-cells consumed (pro rata of the staged total)
    st_total_cells = max(1, ncells)
    bytes_in = int(eng_reference_bytes(eng) * (cells_in / st_total_cells))

    dt, (recs_all, cells_all, bo_all, bi_all) = gdist.aggregate(dt, [recs, cells_in, bytes_out, bytes_in], device="cuda" if backend == "nccl" else None)

    concat = None
    if args.concat and world > 1 and not args.bcf:
        # the single-stream view: one window per rank as BGZF pages (blocks concatenate) to rank 0 over RCCL / xGMI, in column order
        eng.close()
        ec = genomicsdb_amd.CombineEngine(q, device=device_index, output_format=args.concat_format)
        gen2 = synth.Generator(N, B, W)
        ec.stage_cells_begin()
        col = B
        while col < B + W:
            col = min(B + W, col + 1_000_000)
            ptr, nbytes, _ = gen2.next_chunk(col)
            ec.stage_cells_append(ptr, nbytes)
        ec.stage_cells_end()
        ec.set_reference(B, synth.reference(B, W + 4096))
        barrier()
        dev = torch.device("cuda", device_index)
        concat = concat_leg(ec.page_tensors(B, B + W - 1, 1 << 30), 1 << 30, rank, world, dev)
        if concat is not None:
            concat["output_format"] = args.concat_format
        ec.close()
    out = None
    if rank == 0:
        # roofline of the dominant kernel: algorithmic bytes of one k_assemble_write launch / its average duration
        launches = max(1.0, wk_launches)
        alg_bytes_per_launch = (bytes_out + bytes_in) / launches
        avg_ms = wk_ms / launches
        achieved = alg_bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "combined-gVCF positions/sec",
            "value": recs_all / dt,
            "unit": "positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": "c2: %d synthetic-gVCF samples x %d bp (BASELINE.json configs[1]); step = one %d bp column window"
                       % (N, Lbp, W), "samples": N, "interval_bp": Lbp, "window_bp": W, "output": "BCF2 records (bu)" if args.bcf else "VCF text, bit-exact",
                       "partition_per_gpu": True},
            "cells_per_sec": cells_all / dt,
            "bytes_out_per_position": bo_all / max(1.0, recs_all),
            "bytes_in_per_cell": bi_all / max(1.0, cells_all),
            "whole_path_GBps": (bo_all + bi_all) / dt / 1e9,
            "phase_ms": {k: v / args.steps for k, v in ms.items()},
            "lanes": lanes,
            "untimed_lane_preparation_steps": (lanes if args.warmup < lanes else 0) if lanes > 1 else 0,
            "lanes_note": ("%d windows in flight at a time (gdbamd_engine_run_intervals: lane pipelines over one staged fragment); phase_ms and "
                           "roofline.avg_launch_ms are device times on each lane's own stream and overlap, so they add up to more than ms_per_step; "
                           "roofline.alone is the same kernel measured with one window at a time, outside the timed region" % lanes) if lanes > 1 else None,
            "whole_path_frac_of_hbm_peak": (bo_all + bi_all) / dt / 1e9 / HBM_PEAK_GBS / max(1, world),
            "stage_seconds_untimed": t_stage,
            "roofline": {"bound": "hbm", "kernel": "k_bcf_write" if args.bcf else "k_assemble_write", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None if args.bcf else pmc_traffic(N, W, arena),
                         "alg_bytes_per_launch": alg_bytes_per_launch, "avg_launch_ms": avg_ms, "launches": int(launches)},
        }
        if alone:
            out["roofline"]["alone"] = {"avg_launch_ms": alone, "achieved": alg_bytes_per_launch / (alone * 1e-3) / 1e9, "frac": alg_bytes_per_launch / (alone * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "what": "%s with one window in flight (3 steps after the timed region)" % ("k_bcf_write" if args.bcf else "k_assemble_write")}
        if concat is not None:
            out["concat"] = concat
        if not args.no_stream and world == 1 and not args.bcf:         # the boundary GATK drives: header + body through gdb_mi355_read
            eng.close()                                                 # (the timed engine's HBM - fragment, the lanes' arenas and tables - is given back first)
            # The same stream twice, the second one reported: the first stream of a process pays one-off costs - its pinned ring and, above all,
            # the first ~100 GB of device allocations after the timed engine gave ~180 GB back (3.3 - 3.4 s in front of the first byte on two of
            # this round's fresh boxes, 0.05 s on the others) - that say nothing about the path.  What the first one did is kept beside it.
            first = stream_end_to_end(N, B, min(W, Lbp), tmp, expect_body_bytes=None)
            out["stream_end_to_end"] = stream_end_to_end(N, B, min(W, Lbp), tmp, expect_body_bytes=None)
            out["stream_end_to_end"]["first_stream_of_the_process"] = {k: first[k] for k in ("positions_per_sec", "GBps", "t_first_byte_s", "t_drain_s")}
            out["stream_end_to_end_bgzf"] = stream_end_to_end(N, B, min(W, Lbp), tmp, expect_body_bytes=None, output_format="z")
            # BCF2, what GATK4's GenomicsDBFeatureReader decodes: uncompressed ("bu", the JNI's is_bcf stream) and as BGZF blocks ("b")
            out["stream_end_to_end_bcf"] = {fmt: stream_end_to_end(N, B, min(W, Lbp), tmp, expect_body_bytes=None, output_format=fmt) for fmt in ("bu", "b")}
        if not args.no_c3 and world == 1 and not args.bcf:
            # BASELINE configs[2] shape in the default line: 10 000 samples, the array does NOT fit next to the working buffers and
            # passes through HBM in column windows with carry-over; the input path (host memory -> HBM) is inside the timed region
            import copy
            try:
                eng.close()
            except Exception:
                pass
            a3 = copy.copy(args)
            a3.samples, a3.interval_bp, a3.window_bp, a3.arena_mb = 10000, args.c3_bp, 50_000, 49152
            out["c3_streamed"] = run_streamed(a3, 0, 1, device_index, backend, source="memory", emit=False)
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(N, args.cpu_sample_bp, tmp)
    if out is not None:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def relaunch_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU of this node"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.exit(rc)


def dry_run(args, rank, world, backend):
    """what the ranks share besides the scan: who owns which columns, max-over-ranks time, sums of the counters"""
    import torch.distributed as dist
    from genomicsdb_amd import dist as gdist
    B, E = gdist.synthetic_partition(rank, 10_000_000, args.interval_bp)
    dt, (cols, ranks) = gdist.aggregate(0.001 * (rank + 1), [E - B + 1, 1])
    concat = None
    if args.concat:   # the accounting of the concat leg on synthetic pages: rank r has 8 pages of (r + 1) MiB, every byte = its rank
        import torch
        page = (rank + 1) << 20
        concat = concat_leg((torch.full((page,), rank, dtype=torch.uint8) for _ in range(8)), 3 << 20, rank, world, None,
                            check=lambda t, st: st.__setitem__("ok", st.get("ok", True) and int(t[0]) >= st.get("last", 0)) or st.__setitem__("last", int(t[0])))
    if rank == 0:
        out = {"metric": "combined-gVCF positions/sec", "value": None, "unit": "positions/s", "n_gpus": world, "dry_run": True,
               "steps": args.steps, "warmup": args.warmup, "scaling": "weak", "ranks_reporting": int(ranks),
               "columns_all_ranks": int(cols), "max_over_ranks_s": dt, "backend": backend}
        if concat is not None:
            out["concat"] = concat
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def concat_leg(pages, page_bytes, rank, world, device, check=None):
    """every rank's pages to rank 0 in column order through dist.paged_concat; rank 0 returns the accounting: bytes and pages the sink saw,
    seconds, GB/s, how long the root waited for the rank it was draining and the longest any sender waited for a free slot"""
    from genomicsdb_amd import dist as gdist
    st, seen = {}, {"pages": 0}

    def sink(t):
        seen["pages"] += 1
        if check is not None:
            check(t, seen)
    gdist.paged_concat(pages, sink, page_bytes, dst=0, ring_slots=3, device=device, stats=st)
    _, (sender_blocked,) = gdist.aggregate(0.0, [0.0], device=device) if world == 1 else (0.0, [0.0])
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor([st["blocked_s"] if rank != 0 else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sender_blocked = float(t.item())
    if rank != 0:
        return None
    return {"ranks": world, "bytes": int(st["bytes"]), "pages": seen["pages"], "seconds": st["seconds"], "GBps": st["bytes"] / max(st["seconds"], 1e-9) / 1e9,
            "root_blocked_s": st["blocked_s"], "max_sender_blocked_s": sender_blocked, "page_bytes": page_bytes, "ordered": bool(seen.get("ok", True))}


def run_streamed(args, rank, world, device_index, backend, source="callback", emit=True):
    """BASELINE configs[2] shape (10 000 samples x chr1): the cells do not fit HBM next to the working buffers, so they are
    handed to the engine chunk by chunk (here straight from the synthetic generator, as a cell callback), staged in column
    windows of the staging budget, and the intervals still live at a window's end are carried into the next window on the
    device.  One pass over --interval-bp; the timed region INCLUDES the input path (host -> HBM copies, taking the cell stream
    apart, carry-over); the generator's own time is measured and reported separately."""
    import torch
    import torch.distributed as dist
    import genomicsdb_amd
    from genomicsdb_amd import synth
    from genomicsdb_amd import dist as gdist
    import helpers
    N, Lbp, W = args.samples, args.interval_bp, max(1, min(args.window_bp, args.interval_bp))
    B, _ = gdist.synthetic_partition(rank, args.base, Lbp)
    tmp = tempfile.mkdtemp(prefix="gdbamd_bench_")
    q = helpers.synth_query(tmp, N, B, B + Lbp - 1)
    eng = genomicsdb_amd.CombineEngine(q, device=device_index)
    gen = synth.Generator(N, B, Lbp)
    chunk_bp = max(1000, int((256 << 20) / (N / 106.0 * 153.0)))      # ~256 MB of cells per chunk
    state = {"col": B, "gen_s": 0.0, "bytes": 0, "cells": 0}

    def next_chunk():
        if state["col"] >= B + Lbp:
            return None
        t = time.time()
        state["col"] = min(B + Lbp, state["col"] + chunk_bp)
        p, n, nc = gen.next_chunk(state["col"])
        state["gen_s"] += time.time() - t
        state["bytes"] += n
        state["cells"] += nc
        return p, n
    t_pregen = 0.0
    keep = None
    if source == "memory":
        # the array in host memory first (untimed): what is timed is host memory -> HBM staging (window w + 1 while window w
        # computes) + scan + combine; the generator is not part of the path being measured
        import numpy as np
        tg = time.time()
        cap = int(N * Lbp / 106.0 * 160.0 * 1.10) + (64 << 20)       # ~153 bytes per cell, one cell per ~106 bp and sample
        keep = np.empty(cap, dtype=np.uint8)
        at = 0
        while True:
            r = next_chunk()
            if r is None:
                break
            if at + r[1] > cap:
                raise RuntimeError("c3 leg: the generated cells exceed the estimated buffer")
            ctypes.memmove(keep.ctypes.data + at, r[0], r[1])
            at += r[1]
        keep = keep[:at]
        t_pregen = time.time() - tg
        state["gen_s"] = 0.0
        # the caller's memory page-locked (untimed, like the pinned ring on the output side): the window staged ahead is then a
        # DMA transfer under the kernels of the window in use, not a bounce-buffer copy that blocks the launching thread
        t_pin = None
        if os.environ.get("GDBAMD_BENCH_PIN", "1") != "0":
            tp = time.time()
            try:
                genomicsdb_amd.api.pin_host_memory(keep.ctypes.data, keep.nbytes)
                t_pin = time.time() - tp
            except Exception as e:
                sys.stderr.write("[bench] the cells stay pageable: %s\n" % e)
        eng.open_memory_cells((keep.ctypes.data, keep.nbytes))
    else:
        eng.open_cell_callback(next_chunk)
    eng.set_reference(B, synth.reference(B, Lbp + 4096))
    arena = args.arena_mb << 20
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    recs = cells_in = bytes_out = 0
    dev_ms = 0.0
    wk_ms = wk_launches = 0.0
    windows = 0
    t_cover = 0.0
    t_run = 0.0
    n_intervals = 0
    lanes = max(1, min(4, int(os.environ.get("GDBAMD_BENCH_C3_LANES", str(getattr(args, "c3_lanes", 1))))))
    if lanes > 1:
        arena = min(arena, int(os.environ.get("GDBAMD_BENCH_LANE_ARENA_MB", "46080")) << 20)
    pos, qe = B, B + Lbp - 1
    while pos <= qe:
        tc = time.time()
        lo, hi = eng.cover(pos)
        t_cover += time.time() - tc
        windows += 1
        end = min(qe, hi)
        if lanes > 1:
            # the pieces of this staged window, `lanes` of them in flight at a time (the lanes adopt the window's fragment)
            ivs = []
            while pos <= end:
                pe = min(end, pos + W - 1)
                ivs.append((pos, pe))
                pos = pe + 1
            tr = time.time()
            sts = eng.run_intervals(ivs, arena_bytes=arena, lanes=lanes)
            t_run += time.time() - tr
        else:
            sts = []
            while pos <= end:
                pe = min(end, pos + W - 1)
                tr = time.time()
                _, st = eng.run_interval(pos, pe, arena_bytes=arena, fetch=False)
                t_run += time.time() - tr
                sts.append(st)
                pos = pe + 1
        for st in sts:
            n_intervals += 1
            recs += st.num_records; cells_in += st.num_cells_in_window; bytes_out += st.bytes_out
            dev_ms += st.ms_total
            wk_ms += st.ms_write_kernel_avg * st.write_launches; wk_launches += st.write_launches
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.time() - t0
    bytes_in = eng.staged_info()[1]
    dt, (recs_all, cells_all, bo_all, bi_all) = gdist.aggregate(dt, [recs, cells_in, bytes_out, bytes_in], device="cuda" if backend == "nccl" else None)
    if rank == 0:
        launches = max(1.0, wk_launches)
        avg_ms = wk_ms / launches
        alg = (bytes_out + bytes_in) / launches
        achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        t_input = t_cover - state["gen_s"]
        out = {
            "metric": "combined-gVCF positions/sec", "value": recs_all / dt, "unit": "positions/s", "n_gpus": world, "steps": 1, "warmup": 0,
            "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": "c3 shape: %d synthetic-gVCF samples x %d bp slice of chr1 (BASELINE.json configs[2]), cells streamed through HBM in column "
                                   "windows with carry-over; input path inside the timed region" % (N, Lbp),
                       "samples": N, "interval_bp": Lbp, "window_bp": W, "input": "streamed", "staging_windows": windows,
                       "staging_budget_MB": int(os.environ.get("GDBAMD_STAGE_BUDGET_MB", "8192")), "output": "VCF text, bit-exact, pages left in HBM"},
            "cells_per_sec": cells_all / dt,
            "positions_per_sec_excluding_generator": recs / max(1e-9, dt - state["gen_s"]),
            "positions_per_sec_device_only": recs / max(1e-9, dev_ms * 1e-3) if lanes == 1 else None,   # (lanes > 1: the intervals' device times overlap)
            "bytes_out_per_position": bo_all / max(1.0, recs_all), "bytes_in_per_cell": bi_all / max(1.0, state["cells"]),
            "whole_path_GBps": (bo_all + bi_all) / dt / 1e9,
            "input_path": {"cell_bytes": state["bytes"], "cells": state["cells"], "t_generator_s": state["gen_s"], "t_stage_s": t_input,
                           "stage_GBps": state["bytes"] / max(1e-9, t_input) / 1e9, "t_device_s": dev_ms * 1e-3, "wall_s": dt},
            "roofline": {"bound": "hbm", "kernel": "k_assemble_write", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_launch": alg, "avg_launch_ms": avg_ms, "launches": int(launches)},
        }
        # where the wall clock of the timed region went, by name: device time of the intervals (HIP events), the rest of the
        # run_interval calls (kernel launches, the syncs that fetch sizes and counters, the page hand-over), cover() (waiting for the
        # window staged ahead, swapping the pipelines; for a callback source also the generator), and the loop around them
        out["lanes"] = lanes
        out["wall_accounting"] = {"wall_s": dt, "device_s": dev_ms * 1e-3, "run_interval_beyond_device_s": t_run - dev_ms * 1e-3, "cover_s": t_cover,
                                  "loop_and_barrier_s": dt - t_run - t_cover, "intervals": n_intervals,
                                  "run_interval_beyond_device_us_per_interval": (t_run - dev_ms * 1e-3) / max(1, n_intervals) * 1e6}
        out["config"]["source"] = "host memory (generated before the timed region, %.1f s)" % t_pregen if source == "memory" else "cell callback (generator inside the timed region)"
        if source == "memory":
            out["config"]["source_pinned"] = t_pin is not None
            out["input_path"]["t_pin_s_untimed"] = t_pin
        out["config"]["overlapped_staging"] = os.environ.get("GDBAMD_OVERLAP_STAGING", "1") != "0"
        eng.close()
        if source == "memory" and t_pin is not None:
            genomicsdb_amd.api.unpin_host_memory(keep.ctypes.data)
        if not emit:
            return out
        print(json.dumps(out), flush=True)
    if world > 1 and emit:
        dist.destroy_process_group()
    return None


def stream_end_to_end(N, B, W, tmp, expect_body_bytes=None, output_format=None):
    """SURVEY 8(d) timing protocol, the part behind the device: t_stage (cells in host memory -> columnar fragment in HBM),
    t_drain and the end-to-end rate of the C-ABI query stream (gdb_mi355_init_from_memory / gdb_mi355_read, the six JNI entry
    points' twin) over one W-bp window of the same workload.  The caller's buffer is pinned host memory; the stream assembles
    GDBAMD_DEVICE_PAGE_MB pages in two HBM arenas and drains them through its pinned ring while the next page is assembled."""
    import torch
    import genomicsdb_amd
    from genomicsdb_amd import synth
    import helpers
    d = os.path.join(tmp, "stream")
    os.makedirs(d, exist_ok=True)
    # reference bases of the window as a FASTA the query names (positions before B are never asked for)
    fasta = os.path.join(d, "synth_ref.fa")
    with open(fasta, "wb") as f:
        f.write(b">1\n")
        f.write(b"N" * B)
        f.write(synth.reference(B, W + 4096))
        f.write(b"\n")
    q = helpers.synth_query(d, N, B, B + W - 1)
    q["reference_genome"] = fasta
    gen = synth.Generator(N, B, W)
    ptr, nbytes, ncells = gen.next_chunk(B + W)
    t0 = time.time()
    s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=(ptr, nbytes), buffer_capacity=1 << 20, output_format=output_format)
    t_stage = time.time() - t0
    cap = 256 << 20
    dst = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
    addr = dst.data_ptr()
    total = 0
    newlines = 0
    t1 = time.time()
    t_first = None
    while True:
        got = s.read_into(addr, cap)
        if got <= 0:
            break
        if t_first is None:
            t_first = time.time() - t1
        total += got
    t_read = time.time() - t1
    st = s.stream_stats()
    s.close()
    gen.close()
    body = int(st.bytes)
    # records of the window: counted on the device copy of the same text would cost another pass; the stream's own page
    # accounting gives the body bytes, the record count comes from one engine pass over the same cells
    eng = genomicsdb_amd.CombineEngine(q, device=torch.cuda.current_device())
    gen2 = synth.Generator(N, B, W)
    p2, n2, _ = gen2.next_chunk(B + W)
    eng.stage_cells_begin(); eng.stage_cells_append(p2, n2); eng.stage_cells_end()
    if output_format in ("bu", "b"):      # the engine's own page accounting in the stream's format ("b": before compression)
        eng.close()
        eng = genomicsdb_amd.CombineEngine(q, device=torch.cuda.current_device(), is_bcf=True)
        gen2 = synth.Generator(N, B, W)
        p2, n2, _ = gen2.next_chunk(B + W)
        eng.stage_cells_begin(); eng.stage_cells_append(p2, n2); eng.stage_cells_end()
    _, est = eng.run_interval(B, B + W - 1, arena_bytes=48 << 30, fetch=False)
    ok = int(est.bytes_out) == body if output_format not in ("z", "b") else None    # (compressed: checked on a sample below)
    recs = int(est.num_records)
    pcie_bound = 63.0   # GB/s, PCIe Gen5 x16 spec (MI355X_MICROARCH.md)
    eng.close()
    extra = {}
    if output_format in ("z", "b"):
        extra = {"output_format": output_format, "uncompressed_body_bytes": int(est.bytes_out), "compression_ratio": int(est.bytes_out) / max(1, body),
                 "uncompressed_GBps": int(est.bytes_out) / t_read / 1e9, "inflated_sample_matches_plain_stream": bgzf_sample_check(N, B, min(W, 4000), q, output_format)}
    return {
        **extra,
        "what": ("header + body of one %d bp window of the same workload through gdb_mi355_read into a pinned 256 MiB buffer" % W) +
                (" as BGZF blocks deflated on the device (vcf_output_format \"%s\")" % output_format if output_format in ("z", "b") else ""),
        "positions_per_sec": recs / t_read, "GBps": total / t_read / 1e9, "frac_of_pcie_spec": total / t_read / 1e9 / pcie_bound,
        "t_stage_s": t_stage, "stage_GBps": nbytes / t_stage / 1e9, "cells": int(ncells), "cell_bytes": int(nbytes),
        "t_drain_s": t_read, "t_first_byte_s": t_first, "bytes": int(total), "records": recs,
        "t_waiting_for_copies_s": st.seconds_waiting_for_copies, "t_producing_s": st.seconds_producing,
        "device_pages": int(st.pages), "ring_chunks": int(st.chunks), "body_bytes_match_engine": ok,
    }


def bgzf_sample_check(N, B, w, q, output_format):
    """untimed: the first w columns of the same workload as a "z" / "b" stream, inflated on the host with zlib (every member, to the end),
    against the plain "" / "bu" stream of the same query, byte for byte"""
    import gzip
    import genomicsdb_amd
    from genomicsdb_amd import synth
    q2 = dict(q)
    q2["query_column_ranges"] = [[[B, B + w - 1]]]
    outs = []
    for fmt in (output_format, "" if output_format == "z" else "bu"):
        gen = synth.Generator(N, B, w + 3000)
        ptr, nbytes, _ = gen.next_chunk(B + w + 3000)
        s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q2, cells=(ptr, nbytes), buffer_capacity=1 << 20, output_format=fmt)
        outs.append(s.read())
        s.close()
        gen.close()
    return gzip.decompress(outs[0]) == outs[1] and len(outs[1]) > 0


def kernel_source_hash():
    """what the traffic file is tied to: the sources the kernels are compiled from"""
    import hashlib
    h = hashlib.sha256()
    for rel in ("genomicsdb_amd/csrc/kernels/gdb_pipeline.hip", "genomicsdb_amd/csrc/core/gdb_core.hpp", "genomicsdb_amd/csrc/core/gdb_stages.hpp",
                "genomicsdb_amd/csrc/core/gdb_bcf.hpp", "genomicsdb_amd/csrc/core/gdb_asa.hpp", "genomicsdb_amd/csrc/core/gdb_types.h"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(N, W, arena):
    """HBM bytes per k_assemble_write launch from the rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate passes,
    MI355X_MICROARCH.md 'HBM'), measured with tests/tools/prof_traffic.sh on this very configuration and committed as
    profiles/traffic.json together with the hash of the kernel sources it was measured on.  null when the file was made for
    another configuration or the kernel sources have changed since (a stale figure is worse than none)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("samples") == N and t.get("window_bp") == W and t.get("arena_bytes") == arena and t.get("kernel_source_hash") == kernel_source_hash():
            return t["k_assemble_write"]["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def eng_reference_bytes(eng):
    """sum of the reference binary-cell sizes of everything staged ("bytes_in")"""
    return eng.staged_info()[1]


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_partition(args):
    """one column partition on one core: what a rank of `mpirun -n P gt_mpi_gather` does"""
    N, B, sample_bp, tmp, idx = args
    import helpers
    from genomicsdb_amd import synth
    g = synth.Generator(N, B, sample_bp + 3000)
    cells, nc = g.chunk_bytes(B + sample_bp + 3000, nthreads=1)
    q = helpers.synth_query(os.path.join(tmp, "p%d" % idx), N, B + 1000, B + 1000 + sample_bp - 1)
    txt, nrec, secs = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    return nrec, secs, len(txt)


def cpu_baseline(N, sample_bp, tmp):
    """the CPU oracle (single-threaded restatement of the reference algorithm) on a bounded sample of the same workload:
    (i) one core / one partition (the figure in "value"), (ii) min(128, cores / 2) partitions in parallel, one process each, like
    the reference's one-rank-per-partition MPI runs"""
    import multiprocessing as mp
    B = 10_000_000
    os.makedirs(os.path.join(tmp, "p0"), exist_ok=True)
    nrec, secs, nbytes = _oracle_partition((N, B, sample_bp, tmp, 0))
    out = {"value": nrec / secs, "unit": "positions/s", "cores": 1, "kind": "port",
           "sample": "%d samples x %d bp window of the same generator; %d records, %.1f s, %d output bytes"
                     % (N, sample_bp, nrec, secs, nbytes),
           "host_cpus": os.cpu_count(), "cpu_model": _cpu_model()}
    try:
        P = max(1, min(128, (os.cpu_count() or 1) // 2))     # min(P, cores): one rank per partition, half the host's hardware threads
        par_bp = max(1000, sample_bp // 3)
        for i in range(P):
            os.makedirs(os.path.join(tmp, "p%d" % (i + 1)), exist_ok=True)
        t0 = time.time()
        with mp.get_context("spawn").Pool(P) as pool:
            res = pool.map(_oracle_partition, [(N, B + (i + 1) * 1_000_000, par_bp, tmp, i + 1) for i in range(P)])
        wall = time.time() - t0
        out["parallel_partitions"] = {"value": sum(r[0] for r in res) / max(r[1] for r in res), "unit": "positions/s", "cores": P,
                                      "sample": "%d partitions x %d bp, one process each; slowest partition %.1f s, wall incl. process start and input generation %.1f s"
                                                % (P, par_bp, max(r[1] for r in res), wall)}
    except Exception as e:  # the single-core figure stands on its own
        out["parallel_partitions"] = {"error": str(e)}
    return out


if __name__ == "__main__":
    main()
