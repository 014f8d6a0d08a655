"""ctypes binding of libgenomicsdb_amd.so.  No fallback: a missing library is an ImportError-grade failure."""
import ctypes
import os
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GDBAMD_LIB_PATH") or os.path.join(PKG, "libgenomicsdb_amd.so")   # (the override: A/B builds of the same sources under build/variants/)


class IntervalStats(ctypes.Structure):
    _fields_ = [("num_cells", ctypes.c_int64), ("num_cells_in_window", ctypes.c_int64), ("num_records", ctypes.c_int64),
                ("num_heavy_incidences", ctypes.c_int64), ("bytes_out", ctypes.c_uint64),
                ("bytes_in_reference_cells", ctypes.c_uint64), ("pages", ctypes.c_int32), ("write_launches", ctypes.c_int32),
                ("err_bits", ctypes.c_uint32), ("ms_sweep", ctypes.c_float), ("ms_site", ctypes.c_float),
                ("ms_size", ctypes.c_float), ("ms_write", ctypes.c_float), ("ms_total", ctypes.c_float),
                ("ms_write_kernel_avg", ctypes.c_float),
                ("num_record_types", ctypes.c_int32), ("resolved_entry_bytes", ctypes.c_int32),
                ("num_text_slots", ctypes.c_int64), ("text_pool_bytes", ctypes.c_int64),
                ("num_remap_elements", ctypes.c_uint64), ("bytes_compressed", ctypes.c_uint64), ("ms_compress", ctypes.c_float),
                ("reserved1", ctypes.c_int32), ("gt_profile_stats", ctypes.c_uint64 * 6)]

    # names of the reference's GTProfileStats counters (query_variants.h:67-124), in enum order
    GT_STAT_NAMES = ("GT_NUM_CELLS", "GT_NUM_CELLS_IN_LEFT_SWEEP", "GT_NUM_VALID_CELLS_IN_QUERY", "GT_NUM_ATTR_CELLS_ACCESSED",
                     "GT_NUM_PQ_FLUSHES_DUE_TO_OVERLAPPING_CELLS", "GT_NUM_OPERATOR_INVOCATIONS")

    def gt_profile(self):
        return dict(zip(self.GT_STAT_NAMES, (int(v) for v in self.gt_profile_stats)))


class StreamStats(ctypes.Structure):
    _fields_ = [("pages", ctypes.c_uint64), ("chunks", ctypes.c_uint64), ("bytes", ctypes.c_uint64),
                ("seconds_waiting_for_copies", ctypes.c_double), ("seconds_producing", ctypes.c_double)]


class DeviceColumn(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("off", ctypes.c_void_p)]


CELL_CHUNK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64))

_lib = None

# every symbol include/genomicsdb_amd.h declares
SYMBOLS = ["gdb_mi355_last_error", "gdb_mi355_device_count", "gdb_mi355_init", "gdb_mi355_init_output_format", "gdb_mi355_init_from_memory_output_format",
           "gdbamd_engine_create_output_format", "gdbamd_bgzf_compress", "gdbamd_bgzf_compress_mode", "gdbamd_bgzf_bound", "gdb_mi355_init_from_memory", "gdb_mi355_init_from_memory_format", "gdb_mi355_close",
           "gdb_mi355_get_num_bytes_available", "gdb_mi355_read_next_byte", "gdb_mi355_read", "gdb_mi355_skip", "gdb_mi355_peek", "gdb_mi355_get_stream_stats",
           "gdbamd_engine_create", "gdbamd_engine_create_format", "gdbamd_engine_destroy", "gdbamd_engine_num_fields", "gdbamd_engine_field_name",
           "gdbamd_engine_field_info", "gdbamd_engine_header", "gdbamd_engine_stage_cells", "gdbamd_engine_stage_cells_begin", "gdbamd_engine_stage_cells_append", "gdbamd_engine_stage_cells_end",
           "gdbamd_engine_adopt_device_fragment", "gdbamd_engine_open_array", "gdbamd_engine_open_memory_cells", "gdbamd_engine_open_cell_callback", "gdbamd_engine_cover", "gdbamd_engine_staged_info", "gdbamd_engine_set_reference", "gdbamd_engine_run_interval", "gdbamd_engine_run_intervals", "gdbamd_engine_lane_footprint", "gdbamd_engine_release_lanes", "gdbamd_engine_prepare_interval", "gdbamd_engine_next_page", "gdbamd_engine_split_point", "gdbamd_engine_save_fragment", "gdbamd_engine_save_fragment_compressed", "gdbamd_engine_load_fragment", "gdbamd_column_partition", "gdbamd_import_cells", "gdbamd_free",
           "gdbamd_engine_column_histogram", "gdbamd_equi_partition_text", "gdbamd_build_output_index", "gdbamd_engine_print_calls", "gdbamd_engine_print_cells", "gdbamd_pin_host_memory", "gdbamd_unpin_host_memory"]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libgenomicsdb_amd.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "the variant-combine path has no Python/CPU fallback")
    # One HIP runtime per process: the torch wheel carries its own libamdhip64 / libhsa-runtime64 under a different soname, so
    # a process that loads this library first (ROCm's runtime) and torch afterwards ends up with two runtimes, and the one that
    # initialises second sees no GPU.  With torch loaded first this library binds to the runtime torch brought, and device
    # pointers, streams and torch.distributed (RCCL) buffers are interchangeable (api.page_tensors, dist.gather_interval).
    if "torch" not in sys.modules and not os.environ.get("GDBAMD_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = ctypes.CDLL(LIB_PATH)
    c = ctypes
    L.gdb_mi355_last_error.restype = c.c_char_p
    L.gdb_mi355_device_count.restype = c.c_int
    L.gdb_mi355_init.restype = c.c_void_p
    L.gdb_mi355_init.argtypes = [c.c_char_p, c.c_char_p, c.c_char_p, c.c_int, c.c_int, c.c_int, c.c_uint64, c.c_uint64, c.c_int, c.c_int, c.c_int, c.c_int]
    L.gdb_mi355_init_from_memory.restype = c.c_void_p
    L.gdb_mi355_init_from_memory.argtypes = [c.c_char_p, c.c_char_p, c.c_uint64, c.c_uint64, c.c_int]
    L.gdb_mi355_init_from_memory_format.restype = c.c_void_p
    L.gdb_mi355_init_from_memory_format.argtypes = [c.c_char_p, c.c_char_p, c.c_uint64, c.c_uint64, c.c_int, c.c_int, c.c_int, c.c_int]
    L.gdb_mi355_init_output_format.restype = c.c_void_p
    L.gdb_mi355_init_output_format.argtypes = [c.c_char_p, c.c_char_p, c.c_char_p, c.c_int, c.c_int, c.c_int, c.c_uint64, c.c_uint64, c.c_char_p, c.c_int, c.c_int, c.c_int]
    L.gdb_mi355_init_from_memory_output_format.restype = c.c_void_p
    L.gdb_mi355_init_from_memory_output_format.argtypes = [c.c_char_p, c.c_char_p, c.c_uint64, c.c_uint64, c.c_int, c.c_char_p, c.c_int, c.c_int]
    L.gdbamd_engine_create_output_format.restype = c.c_void_p
    L.gdbamd_engine_create_output_format.argtypes = [c.c_char_p, c.c_int, c.c_char_p, c.c_int]
    L.gdbamd_bgzf_compress.argtypes = [c.c_char_p, c.c_uint64, c.c_char_p, c.c_uint64, c.POINTER(c.c_uint64), c.POINTER(c.c_float)]
    L.gdbamd_bgzf_compress_mode.argtypes = [c.c_char_p, c.c_uint64, c.c_char_p, c.c_uint64, c.POINTER(c.c_uint64), c.POINTER(c.c_float), c.c_int]
    L.gdbamd_bgzf_bound.restype = c.c_uint64
    L.gdbamd_bgzf_bound.argtypes = [c.c_uint64]
    L.gdbamd_engine_create_format.restype = c.c_void_p
    L.gdbamd_engine_create_format.argtypes = [c.c_char_p, c.c_int, c.c_int, c.c_int]
    L.gdb_mi355_close.restype = c.c_uint64
    L.gdb_mi355_close.argtypes = [c.c_void_p]
    L.gdb_mi355_get_num_bytes_available.restype = c.c_uint64
    L.gdb_mi355_get_num_bytes_available.argtypes = [c.c_void_p]
    L.gdb_mi355_read_next_byte.restype = c.c_int
    L.gdb_mi355_read_next_byte.argtypes = [c.c_void_p]
    L.gdb_mi355_read.restype = c.c_int64
    L.gdb_mi355_read.argtypes = [c.c_void_p, c.c_void_p, c.c_uint64, c.c_uint64]
    L.gdb_mi355_peek.argtypes = [c.c_void_p, c.POINTER(c.c_void_p), c.POINTER(c.c_uint64)]
    L.gdb_mi355_get_stream_stats.argtypes = [c.c_void_p, c.POINTER(StreamStats)]
    L.gdb_mi355_skip.restype = c.c_int64
    L.gdb_mi355_skip.argtypes = [c.c_void_p, c.c_uint64]
    L.gdbamd_engine_create.restype = c.c_void_p
    L.gdbamd_engine_create.argtypes = [c.c_char_p, c.c_int]
    L.gdbamd_engine_destroy.argtypes = [c.c_void_p]
    L.gdbamd_engine_num_fields.argtypes = [c.c_void_p]
    L.gdbamd_engine_field_name.restype = c.c_char_p
    L.gdbamd_engine_field_name.argtypes = [c.c_void_p, c.c_int]
    L.gdbamd_engine_field_info.argtypes = [c.c_void_p, c.c_int, c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_int)]
    L.gdbamd_engine_header.restype = c.c_uint64
    L.gdbamd_engine_header.argtypes = [c.c_void_p, c.c_char_p, c.c_uint64]
    L.gdbamd_engine_stage_cells.argtypes = [c.c_void_p, c.c_char_p, c.c_uint64]
    L.gdbamd_engine_stage_cells_begin.argtypes = [c.c_void_p]
    L.gdbamd_engine_stage_cells_append.argtypes = [c.c_void_p, c.c_void_p, c.c_uint64]
    L.gdbamd_engine_stage_cells_end.argtypes = [c.c_void_p]
    L.gdbamd_engine_adopt_device_fragment.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p, c.POINTER(DeviceColumn), c.c_int, c.c_uint64]
    L.gdbamd_engine_open_array.argtypes = [c.c_void_p, c.c_char_p]
    L.gdbamd_engine_open_memory_cells.argtypes = [c.c_void_p, c.c_void_p, c.c_uint64]
    L.gdbamd_pin_host_memory.argtypes = [c.c_void_p, c.c_uint64]
    L.gdbamd_unpin_host_memory.argtypes = [c.c_void_p]
    L.gdbamd_engine_open_cell_callback.argtypes = [c.c_void_p, CELL_CHUNK_FN, c.c_void_p]
    L.gdbamd_engine_cover.argtypes = [c.c_void_p, c.c_int64, c.POINTER(c.c_int64), c.POINTER(c.c_int64)]
    L.gdbamd_engine_staged_info.argtypes = [c.c_void_p, c.POINTER(c.c_int64), c.POINTER(c.c_uint64)]
    L.gdbamd_engine_set_reference.argtypes = [c.c_void_p, c.c_int64, c.c_char_p, c.c_uint64]
    L.gdbamd_engine_run_interval.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_uint64, c.c_void_p, c.c_uint64, c.POINTER(c.c_uint64), c.POINTER(IntervalStats)]
    L.gdbamd_engine_run_intervals.argtypes = [c.c_void_p, c.c_int, c.POINTER(c.c_int64), c.POINTER(c.c_int64), c.c_uint64, c.c_int, c.POINTER(IntervalStats),
                                              c.POINTER(c.c_char_p), c.POINTER(c.c_uint64), c.POINTER(c.c_uint64)]
    L.gdbamd_engine_lane_footprint.argtypes = [c.c_void_p, c.c_int64, c.c_uint64, c.POINTER(c.c_uint64)]
    L.gdbamd_engine_release_lanes.argtypes = [c.c_void_p]
    L.gdbamd_engine_split_point.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_int64, c.POINTER(c.c_int64)]
    L.gdbamd_engine_save_fragment.argtypes = [c.c_void_p, c.c_char_p]
    L.gdbamd_engine_save_fragment_compressed.argtypes = [c.c_void_p, c.c_char_p]
    L.gdbamd_engine_load_fragment.argtypes = [c.c_void_p, c.c_char_p]
    L.gdbamd_column_partition.argtypes = [c.c_char_p, c.c_int, c.POINTER(c.c_int64), c.POINTER(c.c_int64)]
    L.gdbamd_import_cells.argtypes = [c.c_char_p, c.c_char_p, c.c_char_p, c.c_int, c.c_int64, c.c_int64, c.POINTER(c.c_void_p), c.POINTER(c.c_uint64), c.POINTER(c.c_int64)]
    L.gdbamd_free.argtypes = [c.c_void_p]
    L.gdbamd_engine_prepare_interval.argtypes = [c.c_void_p, c.c_int64, c.c_int64]
    L.gdbamd_engine_next_page.argtypes = [c.c_void_p, c.c_uint64, c.POINTER(c.c_void_p), c.POINTER(c.c_uint64)]
    _lib = L
    return L


def last_error():
    return lib().gdb_mi355_last_error().decode(errors="replace")
