import ctypes
import json
import os

from . import _lib

INT64_MAX = 2**63 - 1


class GenomicsDBException(RuntimeError):
    pass


def _check(ok, what):
    if not ok:
        raise GenomicsDBException("%s: %s" % (what, _lib.last_error()))


class _DevicePage:
    """an HBM page as an object torch can alias (CUDA array interface, version 2; ROCm builds of torch honour it)"""

    def __init__(self, addr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(addr), False), "version": 2}


class GenomicsDBQueryStream:
    """Byte stream of the combined gVCF (header first), the Python face of the six JNI entry points."""

    def __init__(self, loader_json_file=None, query_json_file=None, chr="", start=0, end=0, rank=0, buffer_capacity=1048576,
                 segment_size=1048576, is_bcf=False, produce_header_only=False, query_json=None, cells=None,
                 use_missing_values_only_not_vector_end=False, keep_idx_fields_in_bcf_header=True, output_format=None):
        """output_format (the reference's vcf_output_format: "", "bu", "z", "b") overrides is_bcf when given"""
        L = _lib.lib()
        if output_format is not None:
            fmt = output_format.encode()
            if query_json is not None:
                txt = query_json if isinstance(query_json, str) else json.dumps(query_json)
                if isinstance(cells, tuple):
                    self._cells = None
                    self._h = L.gdb_mi355_init_from_memory_output_format(txt.encode(), ctypes.cast(cells[0], ctypes.c_char_p), cells[1], buffer_capacity, int(produce_header_only),
                                                                         fmt, int(use_missing_values_only_not_vector_end), int(keep_idx_fields_in_bcf_header))
                else:
                    self._cells = bytes(cells or b"")
                    self._h = L.gdb_mi355_init_from_memory_output_format(txt.encode(), self._cells, len(self._cells), buffer_capacity, int(produce_header_only), fmt,
                                                                         int(use_missing_values_only_not_vector_end), int(keep_idx_fields_in_bcf_header))
            else:
                self._h = L.gdb_mi355_init_output_format((loader_json_file or "").encode(), (query_json_file or "").encode(), chr.encode(), start, end, rank, buffer_capacity,
                                                         segment_size, fmt, int(produce_header_only), int(use_missing_values_only_not_vector_end), int(keep_idx_fields_in_bcf_header))
            _check(self._h, "GenomicsDBQueryStream init")
            return
        if query_json is not None:
            txt = query_json if isinstance(query_json, str) else json.dumps(query_json)
            if isinstance(cells, tuple):      # (host address, nbytes): cells that already lie in native memory (synthetic generator)
                self._cells = None
                self._h = L.gdb_mi355_init_from_memory_format(txt.encode(), ctypes.cast(cells[0], ctypes.c_char_p), cells[1], buffer_capacity, int(produce_header_only),
                                                              int(is_bcf), int(use_missing_values_only_not_vector_end), int(keep_idx_fields_in_bcf_header))
            else:
                self._cells = bytes(cells or b"")
                self._h = L.gdb_mi355_init_from_memory_format(txt.encode(), self._cells, len(self._cells), buffer_capacity, int(produce_header_only),
                                                              int(is_bcf), int(use_missing_values_only_not_vector_end), int(keep_idx_fields_in_bcf_header))
        else:
            self._h = L.gdb_mi355_init((loader_json_file or "").encode(), (query_json_file or "").encode(), chr.encode(), start, end, rank,
                                       buffer_capacity, segment_size, int(is_bcf), int(produce_header_only), int(use_missing_values_only_not_vector_end),
                                       int(keep_idx_fields_in_bcf_header))
        _check(self._h, "GenomicsDBQueryStream init")

    def read(self, n=-1):
        L = _lib.lib()
        chunks = []
        want = n if n >= 0 else None
        while want is None or want > 0:
            k = 1 << 20 if want is None else min(want, 1 << 20)
            buf = ctypes.create_string_buffer(k)
            got = L.gdb_mi355_read(self._h, buf, 0, k)
            if got < 0:
                raise GenomicsDBException("read: " + _lib.last_error())
            if got == 0:
                break
            chunks.append(buf.raw[:got])
            if want is not None:
                want -= got
        return b"".join(chunks)

    def read_into(self, addr, n):
        """up to n bytes of the stream into host memory at `addr` (e.g. a pinned torch tensor's data_ptr()); returns the count"""
        got = _lib.lib().gdb_mi355_read(self._h, addr, 0, n)
        if got < 0:
            raise GenomicsDBException("read: " + _lib.last_error())
        return got

    def stream_stats(self):
        st = _lib.StreamStats()
        _check(_lib.lib().gdb_mi355_get_stream_stats(self._h, ctypes.byref(st)) == 0, "stream_stats")
        return st

    def skip(self, n):
        return _lib.lib().gdb_mi355_skip(self._h, n)

    def available(self):
        return _lib.lib().gdb_mi355_get_num_bytes_available(self._h)

    def close(self):
        if self._h:
            _lib.lib().gdb_mi355_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CombineEngine:
    """One column partition on one GPU: stage cells (or adopt device columns), run query intervals."""

    def __init__(self, query_json, device=0, is_bcf=False, use_missing_values_only_not_vector_end=False, output_format=None):
        L = _lib.lib()
        txt = query_json if isinstance(query_json, str) else json.dumps(query_json)
        if output_format is not None:
            self._e = L.gdbamd_engine_create_output_format(txt.encode(), device, output_format.encode(), int(use_missing_values_only_not_vector_end))
        else:
            self._e = L.gdbamd_engine_create_format(txt.encode(), device, int(is_bcf), int(use_missing_values_only_not_vector_end))
        _check(self._e, "CombineEngine create")
        self._keep = []

    @property
    def header(self):
        L = _lib.lib()
        n = L.gdbamd_engine_header(self._e, None, 0)
        buf = ctypes.create_string_buffer(n)
        L.gdbamd_engine_header(self._e, buf, n)
        return buf.raw[:n]

    def fields(self):
        L = _lib.lib()
        out = []
        for f in range(L.gdbamd_engine_num_fields(self._e)):
            et, var, num = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            L.gdbamd_engine_field_info(self._e, f, ctypes.byref(et), ctypes.byref(var), ctypes.byref(num))
            out.append({"name": L.gdbamd_engine_field_name(self._e, f).decode(), "elem": et.value, "var": bool(var.value), "num": num.value})
        return out

    def stage_cells(self, cells):
        cells = bytes(cells)
        _check(_lib.lib().gdbamd_engine_stage_cells(self._e, cells, len(cells)) == 0, "stage_cells")

    def stage_cells_begin(self):
        _check(_lib.lib().gdbamd_engine_stage_cells_begin(self._e) == 0, "stage_cells_begin")

    def stage_cells_append(self, ptr, nbytes):
        """ptr: host address (int / ctypes pointer) of `nbytes` of cells continuing the column-major order"""
        _check(_lib.lib().gdbamd_engine_stage_cells_append(self._e, ptr, nbytes) == 0, "stage_cells_append")

    def stage_cells_end(self):
        _check(_lib.lib().gdbamd_engine_stage_cells_end(self._e) == 0, "stage_cells_end")

    # ---- arrays larger than the staging budget: column windows streamed through HBM (see include/genomicsdb_amd.h) ----
    def open_array(self, directory):
        _check(_lib.lib().gdbamd_engine_open_array(self._e, str(directory).encode()) == 0, "open_array")

    def open_memory_cells(self, cells):
        """cells: bytes, or (host address, nbytes); kept alive by this object"""
        if isinstance(cells, tuple):
            addr, n = cells
        else:
            self._keep.append(bytes(cells))
            addr, n = ctypes.cast(ctypes.c_char_p(self._keep[-1]), ctypes.c_void_p).value, len(self._keep[-1])
        _check(_lib.lib().gdbamd_engine_open_memory_cells(self._e, addr, n) == 0, "open_memory_cells")

    def open_cell_callback(self, next_chunk):
        """next_chunk() -> (host address, nbytes) of the next whole-column chunk (valid until the next call) or None at the end"""
        def _fn(_user, pp, pn):
            r = next_chunk()
            if r is None:
                return 0
            pp[0], pn[0] = r[0], r[1]
            return 1
        cb = _lib.CELL_CHUNK_FN(_fn)
        self._keep.append(cb)
        _check(_lib.lib().gdbamd_engine_open_cell_callback(self._e, cb, None) == 0, "open_cell_callback")

    def cover(self, column):
        """stage windows until `column` is covered; returns (lo, hi): the query positions the staged fragment serves"""
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        _check(_lib.lib().gdbamd_engine_cover(self._e, column, ctypes.byref(lo), ctypes.byref(hi)) == 0, "cover")
        return lo.value, hi.value

    def adopt_device_fragment(self, ncells, row_ptr, begin_ptr, end_ptr, cols, reference_cell_bytes, keepalive=None):
        """cols: list of (data_ptr, off_ptr_or_0) device addresses, one per plan field."""
        arr = (_lib.DeviceColumn * len(cols))()
        for i, (d, o) in enumerate(cols):
            arr[i].data = d
            arr[i].off = o or None
        self._keep = [arr, keepalive]
        _check(_lib.lib().gdbamd_engine_adopt_device_fragment(self._e, ncells, row_ptr, begin_ptr, end_ptr, arr, len(cols), reference_cell_bytes) == 0,
               "adopt_device_fragment")

    def staged_info(self):
        n, b = ctypes.c_int64(), ctypes.c_uint64()
        _check(_lib.lib().gdbamd_engine_staged_info(self._e, ctypes.byref(n), ctypes.byref(b)) == 0, "staged_info")
        return n.value, b.value

    def set_reference(self, begin, bases):
        _check(_lib.lib().gdbamd_engine_set_reference(self._e, begin, bases, len(bases)) == 0, "set_reference")

    def print_calls(self, mode=0):
        """`gt_mpi_gather --print-calls` (mode 0: the JSON document of the cells of the query's column intervals, VariantCallPrintOperator),
        `--print-csv` (1) or `--print-AC` (2); bytes"""
        L = _lib.lib()
        L.gdbamd_engine_print_cells.restype = ctypes.c_int64
        L.gdbamd_engine_print_cells.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64]
        n = L.gdbamd_engine_print_cells(self._e, mode, None, 0)
        _check(n >= 0, "print_calls")
        buf = ctypes.create_string_buffer(max(1, n))
        _check(L.gdbamd_engine_print_cells(self._e, mode, buf, n) == n, "print_calls")
        return buf.raw[:n]

    def print_csv(self):
        return self.print_calls(1)

    def print_allele_counts(self):
        return self.print_calls(2)

    def column_histogram(self, hist_begin, hist_end, bin_size, counts=None):
        """ColumnHistogramOperator on the device: numpy uint64 counts[(hist_end - hist_begin) // bin_size + 1] of the staged begin-cells by
        begin column (counts given: added to, for arrays streamed in windows)"""
        import numpy as np
        nbins = (hist_end - hist_begin) // bin_size + 1
        acc = counts is not None
        if counts is None:
            counts = np.zeros(nbins, dtype=np.uint64)
        assert counts.dtype == np.uint64 and counts.size == nbins and counts.flags["C_CONTIGUOUS"]
        L = _lib.lib()
        L.gdbamd_engine_column_histogram.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]
        _check(L.gdbamd_engine_column_histogram(self._e, hist_begin, hist_end, bin_size, counts.ctypes.data, nbins, int(acc)) == 0, "column_histogram")
        return counts

    def split_point(self, begin, end, max_columns):
        """last column of the first piece of [begin, end] that can be run on its own with byte-identical output"""
        pe = ctypes.c_int64()
        _check(_lib.lib().gdbamd_engine_split_point(self._e, begin, end, max_columns, ctypes.byref(pe)) == 0, "split_point")
        return pe.value

    def save_fragment(self, path, compress=False):
        """the staged fragment as a columnar file; compress: DEFLATE tiles that are inflated on the device when the file is read"""
        fn = _lib.lib().gdbamd_engine_save_fragment_compressed if compress else _lib.lib().gdbamd_engine_save_fragment
        _check(fn(self._e, str(path).encode()) == 0, "save_fragment")

    def load_fragment(self, path):
        _check(_lib.lib().gdbamd_engine_load_fragment(self._e, str(path).encode()) == 0, "load_fragment")

    def run_interval(self, begin=0, end=INT64_MAX - 1, arena_bytes=1 << 30, fetch=True, host_cap=None):
        L = _lib.lib()
        st = _lib.IntervalStats()
        n = ctypes.c_uint64()
        if fetch:
            cap = host_cap or (1 << 26)
            while True:
                buf = ctypes.create_string_buffer(cap)
                rc = L.gdbamd_engine_run_interval(self._e, begin, end, arena_bytes, buf, cap, ctypes.byref(n), ctypes.byref(st))
                _check(rc == 0, "run_interval")
                if n.value <= cap:
                    return buf.raw[:n.value], st
                cap = n.value
        rc = L.gdbamd_engine_run_interval(self._e, begin, end, arena_bytes, None, 0, ctypes.byref(n), ctypes.byref(st))
        _check(rc == 0, "run_interval")
        return None, st

    def run_intervals(self, intervals, arena_bytes=1 << 30, lanes=2, fetch=False, host_cap=1 << 26):
        """several (begin, end) column intervals of the staged fragment, up to `lanes` in flight at a time on device pipelines that share
        the fragment (gdbamd_engine_run_intervals).  fetch=False: pages stay in HBM, returns the per-interval statistics in interval
        order; fetch=True: returns [(body bytes, statistics)]"""
        L = _lib.lib()
        n = len(intervals)
        b = (ctypes.c_int64 * max(1, n))(*[iv[0] for iv in intervals])
        e = (ctypes.c_int64 * max(1, n))(*[iv[1] for iv in intervals])
        st = (_lib.IntervalStats * max(1, n))()
        if not fetch:
            _check(L.gdbamd_engine_run_intervals(self._e, n, b, e, arena_bytes, lanes, st, None, None, None) == 0, "run_intervals")
            return [st[i] for i in range(n)]
        caps = [host_cap] * n
        while True:
            bufs = [ctypes.create_string_buffer(c) for c in caps]
            ptrs = (ctypes.c_char_p * max(1, n))(*[ctypes.cast(x, ctypes.c_char_p) for x in bufs])
            cap = (ctypes.c_uint64 * max(1, n))(*caps)
            lens = (ctypes.c_uint64 * max(1, n))()
            _check(L.gdbamd_engine_run_intervals(self._e, n, b, e, arena_bytes, lanes, st, ptrs, cap, lens) == 0, "run_intervals")
            if all(lens[i] <= caps[i] for i in range(n)):
                return [(bufs[i].raw[:lens[i]], st[i]) for i in range(n)]
            caps = [max(caps[i], lens[i]) for i in range(n)]

    def lane_footprint(self, interval_columns, arena_bytes=1 << 30):
        """HBM bytes a new lane pipeline of run_intervals takes at this query's sample count (gdbamd_engine_lane_footprint)"""
        n = ctypes.c_uint64(0)
        _check(_lib.lib().gdbamd_engine_lane_footprint(self._e, interval_columns, arena_bytes, ctypes.byref(n)) == 0, "lane_footprint")
        return n.value

    def release_lanes(self):
        """frees the lane pipelines run_intervals created (their HBM comes back; the next call creates them again)"""
        _check(_lib.lib().gdbamd_engine_release_lanes(self._e) == 0, "release_lanes")

    def pages(self, begin=0, end=INT64_MAX - 1, arena_bytes=1 << 30):
        """the VCF body of one column interval page by page, left in HBM: yields (device address, nbytes); an address is valid
        until the next page is asked for"""
        L = _lib.lib()
        _check(L.gdbamd_engine_prepare_interval(self._e, begin, end) == 0, "prepare_interval")
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        while True:
            rc = L.gdbamd_engine_next_page(self._e, arena_bytes, ctypes.byref(p), ctypes.byref(n))
            _check(rc >= 0, "next_page")
            if rc == 0:
                return
            yield p.value, n.value

    def page_tensors(self, begin=0, end=INT64_MAX - 1, arena_bytes=1 << 30, device=None):
        """pages() as torch uint8 tensors that alias the HBM page (no copy); a tensor is valid until the next page is asked for"""
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        for addr, n in self.pages(begin, end, arena_bytes):
            yield torch.as_tensor(_DevicePage(addr, n), device=dev)

    def close(self):
        if self._e:
            _lib.lib().gdbamd_engine_destroy(self._e)
            self._e = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pin_host_memory(addr, nbytes):
    """page-lock the caller's host range so that staging copies out of it are DMA transfers (gdbamd_pin_host_memory)"""
    _check(_lib.lib().gdbamd_pin_host_memory(addr, nbytes) == 0, "pin_host_memory")


def unpin_host_memory(addr):
    _check(_lib.lib().gdbamd_unpin_host_memory(addr) == 0, "unpin_host_memory")


def import_cells(vid_mapping_file, callset_mapping_file, file_root="", treat_deletions_as_intervals=True, column_begin=0, column_end=2**63 - 2):
    """(g)VCFs of a callset mapping -> begin-cells (bytes, reference binary cell layout, column-major) of one column partition:
    the conversion step of the reference's vcf2tiledb (vcf2binary.cc:991-1196).  Host code, no device needed."""
    L = _lib.lib()
    p = ctypes.c_void_p()
    n = ctypes.c_uint64()
    nc = ctypes.c_int64()
    rc = L.gdbamd_import_cells(os.fsencode(vid_mapping_file), os.fsencode(callset_mapping_file), os.fsencode(file_root or ""), 1 if treat_deletions_as_intervals else 0,
                               column_begin, column_end, ctypes.byref(p), ctypes.byref(n), ctypes.byref(nc))
    _check(rc == 0, "import_cells")
    try:
        return ctypes.string_at(p.value, n.value), nc.value
    finally:
        L.gdbamd_free(p)


def bgzf_compress(data, vcf_text=False):
    """BGZF blocks (no EOF block) of `data`, deflated by the device kernels (kernels/gdb_bgzf.hip); returns (bytes, kernel ms).
    vcf_text: the anchored kernel of the "z" stream (matches begin at tabs / newlines) instead of the byte-level one"""
    L = _lib.lib()
    data = bytes(data)
    cap = L.gdbamd_bgzf_bound(len(data))
    dst = ctypes.create_string_buffer(cap)
    n = ctypes.c_uint64()
    ms = ctypes.c_float()
    _check(L.gdbamd_bgzf_compress_mode(data, len(data), dst, cap, ctypes.byref(n), ctypes.byref(ms), 1 if vcf_text else 0) == 0, "bgzf_compress")
    return dst.raw[:n.value], ms.value


BGZF_EOF = bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0])
