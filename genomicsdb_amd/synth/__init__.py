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
        L.gdbsynth_next_chunk.restype = ctypes.c_int64
        L.gdbsynth_next_chunk.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
        L.gdbsynth_reference.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_char_p]
        _lib = L
    return _lib


class Generator:
    """cells of N samples over [B, B+L), handed out in column chunks (column-major order inside and across chunks)"""

    def __init__(self, n_samples, B, L, seed=SEED, dense=None, rank_sum_scale=None, overlap_permille=0, filter_permille=0, filter2_permille=0,
                 id_permille=0, filter_id=1, filter_id2=0, with_id=False):
        """dense = (begin, length, hot_stride, K): BASELINE.json configs[4]-style region where every sample starts an
        insertion, drawn from a pool of K alleles, at every multiple of hot_stride"""
        self.n_samples, self.B, self.L, self.seed = n_samples, B, L, seed
        if dense:
            self._h = lib().gdbsynth_create_dense(seed, n_samples, B, L, dense[0], dense[1], dense[2], dense[3])
        else:
            self._h = lib().gdbsynth_create(seed, n_samples, B, L)
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


def write_metadata(dirname, n_samples, vid_template_path, with_id=False):
    """vid mapping (schema of the reference's tests/inputs/vid.json, contig '1' only) + callsets S%06d; returns the paths"""
    vid = json.load(open(vid_template_path))
    vid["contigs"] = {"1": {"length": 249250621, "tiledb_column_offset": 0}}
    if with_id:
        vid["fields"]["ID"] = {"type": "char", "length": "VAR"}
    cs = {"callsets": {"S%06d" % i: {"row_idx": i, "idx_in_file": 0, "filename": "synthetic"} for i in range(n_samples)}}
    vp = os.path.join(dirname, "vid_synth_id.json" if with_id else "vid_synth.json")
    cp = os.path.join(dirname, "callsets_synth_%d.json" % n_samples)
    json.dump(vid, open(vp, "w"))
    json.dump(cs, open(cp, "w"))
    return vp, cp
