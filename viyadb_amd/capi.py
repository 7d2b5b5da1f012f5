"""ctypes mirror of include/viya_hip.h — the C-ABI boundary of the aggregate path.

Nothing in here computes: it loads ``libviya_hip.so`` (built in-tree by
``viyadb_amd.build``) and exposes the structs/functions one to one.  There is no
CPU fallback: if the library is missing or a call fails, a ``VhError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VIYA_HIP_LIB") or os.path.join(_HERE, "libviya_hip.so")   # VIYA_HIP_LIB: a measurement build (tools/build_variant.py)

# enum vh_elem
U8, U16, U32, U64, I8, I16, I32, I64, F32, F64, BITSET32, BITSET64 = range(12)
ELEM_NAMES = ["u8", "u16", "u32", "u64", "i8", "i16", "i32", "i64", "f32", "f64", "bitset32", "bitset64"]
ELEM_NP = ["uint8", "uint16", "uint32", "uint64", "int8", "int16", "int32", "int64", "float32", "float64"]
ELEM_SIZE = [1, 2, 4, 8, 1, 2, 4, 8, 4, 8]
# enum vh_kind
DIM_STRING, DIM_NUMERIC, DIM_TIME, DIM_BOOLEAN = 0, 1, 2, 3
METRIC_MAX, METRIC_MIN, METRIC_SUM, METRIC_AVG, METRIC_COUNT, METRIC_BITSET, METRIC_HIDDEN_COUNT = 16, 17, 18, 19, 20, 21, 22
# enum vh_fkind / vh_relop
F_TRUE, F_REL, F_IN, F_AND, F_OR = range(5)
OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE = range(6)
# enum vh_time_unit
T_YEAR, T_MONTH, T_WEEK, T_DAY, T_HOUR, T_MINUTE, T_SECOND, T_NONE = range(8)
MAX_ROLLUP = 8
# plan flags
PLAN_FORCE_HASH, PLAN_FORCE_GLOBAL, PLAN_NO_XCD_PRIVATE, PLAN_NO_FAST, PLAN_NO_PART, PLAN_NO_CARRIER, PLAN_FORCE_PART = 1, 2, 4, 8, 16, 32, 64
PLAN_NO_LANES, PLAN_FORCE_LANES, PLAN_NO_LDS_HASH, PLAN_NO_HASH_RECORDS, PLAN_FORCE_HASH_RECORDS = 128, 256, 512, 1024, 2048
PLAN_NO_PACK, PLAN_FORCE_PACK, PLAN_NO_PART2, PLAN_NO_SHAPE, PLAN_NO_NARROW = 4096, 8192, 16384, 32768, 65536
PLAN_NO_JIT, PLAN_FORCE_JIT, PLAN_NO_HPART, PLAN_FORCE_HPART, PLAN_NO_HP_PACK, PLAN_CARD32, PLAN_NO_NARROW_TUPLES = 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23
PLAN_NO_PREDPACK, PLAN_NO_QPAY, PLAN_FORCE_QPAY, PLAN_NO_SLICED = 1 << 24, 1 << 25, 1 << 26, 1 << 27
# paths
PATH_SCALAR, PATH_DENSE_LDS, PATH_DENSE_GLOBAL, PATH_HASH, PATH_DENSE_PART = range(5)
PATH_NAMES = ["scalar", "dense_lds", "dense_global", "hash", "dense_part"]
# generator
GEN_UNIFORM, GEN_ROWID, GEN_CONST, GEN_ZIPF, GEN_SORTED, GEN_HOT = range(6)


class ColDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("elem", C.c_int32)]


class AnyNum(C.Union):
    _fields_ = [("u8", C.c_uint8), ("u16", C.c_uint16), ("u32", C.c_uint32), ("u64", C.c_uint64),
                ("i8", C.c_int8), ("i16", C.c_int16), ("i32", C.c_int32), ("i64", C.c_int64),
                ("f32", C.c_float), ("f64", C.c_double)]


class FilterNode(C.Structure):
    _fields_ = [("kind", C.c_int32), ("col", C.c_int32), ("op", C.c_int32), ("count", C.c_int32),
                ("lit", C.c_int32), ("reserved", C.c_int32)]


class GroupCol(C.Structure):
    _fields_ = [("col", C.c_int32), ("granularity", C.c_int32), ("nrollup", C.c_int32),
                ("rollup_unit", C.c_int32 * MAX_ROLLUP), ("rollup_before", C.c_uint64 * MAX_ROLLUP),
                ("micro", C.c_int32), ("reserved", C.c_int32), ("cardinality", C.c_uint64)]


class Plan(C.Structure):
    _fields_ = [("filter", C.POINTER(FilterNode)), ("nfilter", C.c_int32),
                ("lits", C.POINTER(AnyNum)), ("nlits", C.c_int32),
                ("groups", C.POINTER(GroupCol)), ("ngroups", C.c_int32),
                ("metrics", C.POINTER(C.c_int32)), ("nmetrics", C.c_int32),
                ("seg_rows", C.POINTER(C.c_uint64)), ("nseg", C.c_uint32),
                ("flags", C.c_uint32), ("groups_hint", C.c_uint64),
                ("having", C.POINTER(FilterNode)), ("nhaving", C.c_int32), ("reserved2", C.c_int32),
                ("top_col", C.c_int32), ("top_desc", C.c_int32), ("top_k", C.c_uint64)]


class ResultInfo(C.Structure):
    _fields_ = [("ngroups", C.c_uint64), ("scanned_recs", C.c_uint64), ("scanned_segments", C.c_uint64),
                ("passed_recs", C.c_uint64), ("path", C.c_int32), ("ngroup_cols", C.c_int32),
                ("nmetrics", C.c_int32), ("has_hidden_count", C.c_int32), ("scan_kernel_ms", C.c_float),
                ("total_ms", C.c_float), ("algorithmic_bytes", C.c_uint64), ("retries", C.c_uint32),
                ("reserved", C.c_uint32), ("returned_groups", C.c_uint64)]


class SelectPlan(C.Structure):
    _fields_ = [("filter", C.POINTER(FilterNode)), ("nfilter", C.c_int32),
                ("lits", C.POINTER(AnyNum)), ("nlits", C.c_int32),
                ("cols", C.POINTER(C.c_int32)), ("ncols", C.c_int32),
                ("seg_rows", C.POINTER(C.c_uint64)), ("nseg", C.c_uint32),
                ("flags", C.c_uint32), ("skip", C.c_uint64), ("limit", C.c_uint64)]


class RowsInfo(C.Structure):
    _fields_ = [("nrows", C.c_uint64), ("scanned_recs", C.c_uint64), ("scanned_segments", C.c_uint64),
                ("passed_recs", C.c_uint64), ("kernel_ms", C.c_float), ("total_ms", C.c_float)]


COL_ROWID = -2


class DeviceBuffer(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("count", C.c_uint64), ("elem", C.c_int32), ("reduce", C.c_int32)]


class CommInfo(C.Structure):
    """vh_comm_info_t: the communicator as the transport itself reports it."""
    _fields_ = [("transport", C.c_int32), ("nranks", C.c_int32), ("rank", C.c_int32), ("device", C.c_int32), ("pci_bus_id", C.c_char * 32)]


COMM_RCCL, COMM_CALLBACKS = 1, 2


class SyncItem(C.Structure):
    """vh_sync_item: one contiguous row range of one segment (vh_table_sync_batch)."""
    _fields_ = [("seg", C.c_uint32), ("flags", C.c_uint32), ("row_first", C.c_uint64), ("nrows", C.c_uint64),
                ("new_size", C.c_uint64), ("col_ptrs", C.POINTER(C.c_void_p))]


SYNC_METRICS_ONLY, SYNC_DEVICE_SRC = 1, 2


class GenSpec(C.Structure):
    _fields_ = [("mode", C.c_int32), ("param", C.c_int32), ("mod", C.c_uint64), ("add", C.c_int64),
                ("scale", C.c_double)]


class CommOps(C.Structure):
    """vh_comm_ops: a transport as a table of callbacks (tests: gloo; see viyadb_amd/distributed.py)."""
    ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
    REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p)
    ALLTOALLV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p)
    _fields_ = [("ctx", C.c_void_p), ("allgather_host", ALLGATHER), ("reduce_device", REDUCE), ("alltoallv_device", ALLTOALLV)]


COMM_ID_BYTES = 128


class VhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"viya_hip error {code}: {msg}")
        self.code = code


# every symbol include/viya_hip.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "vh_init": (C.c_int, [C.c_int]),
    "vh_set_stream": (C.c_int, [_VP]),
    "vh_last_error": (C.c_char_p, []),
    "vh_version": (C.c_char_p, []),
    "vh_table_create": (C.c_int, [C.POINTER(ColDesc), C.c_int32, C.c_uint64, C.c_uint32, C.POINTER(_VP)]),
    "vh_table_destroy": (None, [_VP]),
    "vh_segment_sync": (C.c_int, [_VP, C.c_uint32, C.c_uint64, C.POINTER(_VP)]),
    "vh_segment_sync_range": (C.c_int, [_VP, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(_VP)]),
    "vh_table_sync_batch": (C.c_int, [_VP, C.POINTER(SyncItem), C.c_uint32]),
    "vh_host_register": (C.c_int, [_VP, C.c_uint64]),
    "vh_host_unregister": (C.c_int, [_VP]),
    "vh_table_sync_stats": (C.c_int, [_VP] + [C.POINTER(C.c_uint64)] * 5),
    "vh_segment_sync_bitset": (C.c_int, [_VP, C.c_uint32, C.c_int32, C.c_uint64, C.POINTER(C.c_uint64), _VP]),
    "vh_segment_generate": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(GenSpec), C.c_uint64]),
    "vh_table_pack": (C.c_int, [_VP, C.POINTER(C.c_int32), C.c_int32]),
    "vh_table_pack_ex": (C.c_int, [_VP, C.POINTER(C.c_int32), C.c_int32, C.c_uint32]),
    "vh_table_unpack": (C.c_int, [_VP]),
    "vh_table_relocate": (C.c_int, [_VP, C.c_uint32]),
    "vh_table_narrow": (C.c_int, [_VP, C.POINTER(C.c_int32), C.c_int32]),
    "vh_table_predpack": (C.c_int, [_VP, C.POINTER(C.c_int32), C.c_int32]),
    "vh_table_predpack_ex": (C.c_int, [_VP, C.POINTER(C.c_int32), C.c_int32, C.c_uint32]),
    "vh_segment_read": (C.c_int, [_VP, C.c_uint32, C.c_int32, C.c_uint64, _VP]),
    "vh_device_read": (C.c_int, [_VP, _VP, C.c_uint64]),
    "vh_table_info": (C.c_int, [_VP, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vh_segment_stats": (C.c_int, [_VP, C.c_uint32, C.c_int32, C.POINTER(AnyNum), C.POINTER(AnyNum)]),
    "vh_query_agg": (C.c_int, [_VP, C.POINTER(Plan), C.POINTER(_VP)]),
    "vh_query_launch": (C.c_int, [_VP, C.POINTER(Plan), C.POINTER(_VP)]),
    "vh_result_device_buffers": (C.c_int, [_VP, C.POINTER(DeviceBuffer), C.c_int32, C.POINTER(C.c_int32)]),
    "vh_result_finalize": (C.c_int, [_VP]),
    "vh_result_partition": (C.c_int, [_VP, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(DeviceBuffer), C.c_int32, C.POINTER(C.c_int32)]),
    "vh_result_partition_pairs": (C.c_int, [_VP, C.c_int32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(DeviceBuffer), C.c_int32, C.POINTER(C.c_int32)]),
    "vh_segment_sync_ids_device": (C.c_int, [_VP, C.c_uint32, C.c_int32, C.c_uint64, _VP]),
    "vh_comm_unique_id": (C.c_int, [_VP]),
    "vh_comm_init": (C.c_int, [_VP, C.c_int32, C.c_int32, C.POINTER(_VP)]),
    "vh_comm_init_custom": (C.c_int, [C.POINTER(CommOps), C.c_int32, C.c_int32, C.POINTER(_VP)]),
    "vh_comm_destroy": (None, [_VP]),
    "vh_comm_info": (C.c_int, [_VP, C.POINTER(CommInfo)]),
    "vh_query_agg_sharded": (C.c_int, [_VP, C.POINTER(Plan), _VP, C.c_int32, C.POINTER(_VP)]),
    "vh_query_select": (C.c_int, [_VP, C.POINTER(SelectPlan), C.POINTER(_VP)]),
    "vh_rows_get_info": (C.c_int, [_VP, C.POINTER(RowsInfo)]),
    "vh_rows_view": (C.c_int, [_VP, C.POINTER(_VP)]),
    "vh_rows_free": (None, [_VP]),
    "vh_result_get_info": (C.c_int, [_VP, C.POINTER(ResultInfo)]),
    "vh_table_prepare": (C.c_int, [_VP, C.POINTER(Plan), C.POINTER(ResultInfo)]),
    "vh_result_kernel": (C.c_char_p, [_VP]),
    "vh_result_state_elem": (C.c_int, [_VP, C.c_int32]),
    "vh_result_copy": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(C.c_uint64)]),
    "vh_result_view": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(C.POINTER(C.c_uint64))]),
    "vh_result_free": (None, [_VP]),
    "vh_measure_read_bandwidth": (C.c_int, [C.c_uint64, C.c_int32, C.POINTER(C.c_double)]),
    "vh_jit_selftest": (C.c_int, [C.c_int32, C.c_char_p, C.c_char_p, C.c_uint64]),
}

_lib = None


def load():
    """dlopen the in-tree HIP library and type its entry points. Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VhError(-100, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback for the aggregate path)")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise VhError(rc, load().vh_last_error().decode("utf-8", "replace"))
