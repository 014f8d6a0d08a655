"""Synthetic gVCF input for bench.py and the parity tests: builds libgdbsynth.so (g++) on demand."""
import ctypes
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgdbsynth.so")
SEED = 20260928
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "gvcf_synth.cc")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            tmp = "%s.%d.tmp" % (LIB, os.getpid())   # several ranks may get here at once: build aside, then rename atomically
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", tmp, src])
            os.replace(tmp, LIB)
        L = ctypes.CDLL(LIB)
        L.gdbsynth_create.restype = ctypes.c_void_p
        L.gdbsynth_create.argtypes = [ctypes.c_uint64, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64]
        L.gdbsynth_create_dense.restype = ctypes.c_void_p
        L.gdbsynth_create_dense.argtypes = [ctypes.c_uint64, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32]
        L.gdbsynth_destroy.argtypes = [ctypes.c_void_p]
        L.gdbsynth_set_rank_sum_scale.argtypes = [ctypes.c_void_p, ctypes.c_double]
        L.gdbsynth_set_modes.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 7
        L.gdbsynth_set_contigs.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.c_int32]
        L.gdbsynth_next_chunk.restype = ctypes.c_int64
        L.gdbsynth_next_chunk.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
        L.gdbsynth_reference.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_char_p]
        _lib = L
    return _lib


class Generator:
    """cells of N samples over [B, B+L), handed out in column chunks (column-major order inside and across chunks)"""

    def __init__(self, n_samples, B, L, seed=SEED, dense=None, rank_sum_scale=None, overlap_permille=0, filter_permille=0, filter2_permille=0,
                 id_permille=0, filter_id=1, filter_id2=0, with_id=False, contigs=None, float_stress_permille=0):
        """dense = (begin, length, hot_stride, K): BASELINE.json configs[4]-style region where every sample starts an
        insertion, drawn from a pool of K alleles, at every multiple of hot_stride
        contigs = [(name, tiledb_column_offset, length), ...] (genome mode, BASELINE.json configs[3]): columns are the flattened
        genome, no record crosses a contig's end and every sample starts anew at a contig's first column"""
        self.n_samples, self.B, self.L, self.seed = n_samples, B, L, seed
        if dense:
            self._h = lib().gdbsynth_create_dense(seed, n_samples, B, L, dense[0], dense[1], dense[2], dense[3])
        else:
            self._h = lib().gdbsynth_create(seed, n_samples, B, L)
        if contigs:
            n = len(contigs)
            offs = (ctypes.c_int64 * n)(*[c[1] for c in contigs])
            lens = (ctypes.c_int64 * n)(*[c[2] for c in contigs])
            lib().gdbsynth_set_contigs(self._h, offs, lens, n)
        if float_stress_permille:   # floats outside the everyday range on that share of the variant calls (tiny, huge, subnormal, +inf, NaN)
            lib().gdbsynth_set_float_stress.argtypes = [ctypes.c_void_p, ctypes.c_int]
            lib().gdbsynth_set_float_stress(self._h, int(float_stress_permille))
        if rank_sum_scale:      # rank sums rounded to 1 / scale instead of 1 / 1000: many tied medians, -0 and +0 included
            lib().gdbsynth_set_rank_sum_scale(self._h, float(rank_sum_scale))
        if overlap_permille or filter_permille or filter2_permille or id_permille or with_id:
            # overlapping intervals of one sample, FILTER ids (vid field indices) and ID tokens on the variant cells; with_id: the
            # cells carry an ID attribute (the vid mapping must list an "ID" field: write_metadata(..., with_id=True))
            lib().gdbsynth_set_modes(self._h, overlap_permille, filter_permille, filter2_permille, id_permille, filter_id, filter_id2, int(with_id))

    def next_chunk(self, col_end, nthreads=None):
        """returns (host address, nbytes, ncells) valid until the next call"""
        p = ctypes.c_void_p()
        n = ctypes.c_uint64()
        nc = lib().gdbsynth_next_chunk(self._h, col_end, nthreads or min(48, os.cpu_count() or 1), ctypes.byref(p), ctypes.byref(n))
        return p.value, n.value, nc

    def chunk_bytes(self, col_end, nthreads=None):
        p, n, nc = self.next_chunk(col_end, nthreads)
        return ctypes.string_at(p, n), nc

    def close(self):
        if self._h:
            lib().gdbsynth_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reference(begin, length, seed=SEED):
    buf = ctypes.create_string_buffer(length)
    lib().gdbsynth_reference(seed, begin, length, buf)
    return buf.raw[:length]


GENOME_CONTIGS = ["1", "2", "3", "4", "5", "6", "7", "8", "9", "10", "11", "12", "13", "14", "15", "16", "17", "18", "19", "20", "21", "22", "X", "Y", "MT"]


def genome_contigs(vid_template_path):
    """contigs 1-22, X, Y, MT with the offsets of the reference's tests/inputs/vid.json as (name, offset, length)"""
    vid = json.load(open(vid_template_path))
    return [(n, vid["contigs"][n]["tiledb_column_offset"], vid["contigs"][n]["length"]) for n in GENOME_CONTIGS]


def write_metadata(dirname, n_samples, vid_template_path, with_id=False, contigs=None):
    """vid mapping (schema of the reference's tests/inputs/vid.json; contig '1' only unless `contigs` = [(name, offset, length)]
    is given) + callsets S%06d; returns the paths"""
    vid = json.load(open(vid_template_path))
    if contigs:
        vid["contigs"] = {n: {"length": ln, "tiledb_column_offset": off} for n, off, ln in contigs}
    else:
        vid["contigs"] = {"1": {"length": 249250621, "tiledb_column_offset": 0}}
    if with_id:
        vid["fields"]["ID"] = {"type": "char", "length": "VAR"}
    cs = {"callsets": {"S%06d" % i: {"row_idx": i, "idx_in_file": 0, "filename": "synthetic"} for i in range(n_samples)}}
    vp = os.path.join(dirname, "vid_synth%s%s.json" % ("_id" if with_id else "", "_ctg%d" % len(contigs) if contigs else ""))
    cp = os.path.join(dirname, "callsets_synth_%d.json" % n_samples)
    json.dump(vid, open(vp, "w"))
    json.dump(cs, open(cp, "w"))
    return vp, cp
