"""ctypes handle on the C++ host shim (include/viya_host.h): JSON in, rows of strings out —
the same surface the reference's tests use (db::Database + SimpleLoader + MemoryRowOutput)."""
from __future__ import annotations

import ctypes as C
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libviya_host.so")
FS, RS = "\x1f", "\x1e"


class Stats(C.Structure):
    _fields_ = [("scanned_segments", C.c_uint64), ("scanned_recs", C.c_uint64), ("aggregated_recs", C.c_uint64),
                ("output_recs", C.c_uint64), ("passed_recs", C.c_uint64), ("compile_time", C.c_double),
                ("whole_time", C.c_double), ("scan_kernel_ms", C.c_double), ("device_total_ms", C.c_double),
                ("path", C.c_int32), ("reserved", C.c_int32)]


class HostError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code
        self.reference_exception = "invalid_argument" if code == 1 else "runtime_error"


SYMBOLS = ["vdb_open", "vdb_close", "vdb_join_node", "vdb_create_table", "vdb_load", "vdb_query", "vdb_query_partial", "vdb_query_merge",
           "vdb_table_info", "vdb_free", "vdb_last_error", "vdb_shim_text"]
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HostError(2, LIB_PATH + " is missing: run __graft_entry__.build()")
        from . import capi
        capi.load()  # libviya_hip.so first (RTLD_GLOBAL)
        lib = C.CDLL(LIB_PATH)
        lib.vdb_last_error.restype = C.c_char_p
        lib.vdb_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.vdb_close.argtypes = [C.c_void_p]
        lib.vdb_close.restype = None
        lib.vdb_create_table.argtypes = [C.c_void_p, C.c_char_p]
        lib.vdb_join_node.argtypes = [C.c_void_p, C.c_void_p]
        lib.vdb_load.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_int64]
        lib.vdb_query.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Stats)]
        lib.vdb_query_partial.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Stats)]
        lib.vdb_query_merge.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int32,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Stats)]
        lib.vdb_table_info.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.vdb_shim_text.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        lib.vdb_free.argtypes = [C.c_void_p]
        lib.vdb_free.restype = None
        _lib = lib
    return _lib


def _check(rc):
    if rc:
        raise HostError(rc, load().vdb_last_error().decode("utf-8", "replace"))


class Database:
    def __init__(self, conf: dict, device: int = 0):
        self.lib = load()
        h = C.c_void_p()
        _check(self.lib.vdb_open(json.dumps(conf).encode(), device, C.byref(h)))
        self.h = h

    def join_node(self, comm):
        """comm: viyadb_amd.distributed.Comm (or None to leave): aggregate queries then cover every rank's rows, rows on rank 0."""
        _check(self.lib.vdb_join_node(self.h, comm.handle if comm is not None else None))

    def close(self):
        if self.h:
            self.lib.vdb_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def create_table(self, conf: dict):
        _check(self.lib.vdb_create_table(self.h, json.dumps(conf).encode()))

    def load(self, table: str, rows, now=None):
        buf = "".join(FS.join(r) + FS + RS for r in rows).encode()
        _check(self.lib.vdb_load(self.h, table.encode(), buf, len(buf), -1 if now is None else int(now)))

    def query(self, q: dict, now=None):
        out, n, st = C.c_void_p(), C.c_size_t(), Stats()
        _check(self.lib.vdb_query(self.h, json.dumps(q).encode(), -1 if now is None else int(now), C.byref(out),
                                  C.byref(n), C.byref(st)))
        try:
            raw = C.string_at(out, n.value).decode("utf-8", "replace")
        finally:
            self.lib.vdb_free(out)
        rows = [r.split(FS)[:-1] for r in raw.split(RS)[:-1]]
        stats = {k: getattr(st, k) for k, _ in Stats._fields_}
        return rows, stats

    # ---- cluster aggregate with binary partial states (host/partial_state.h)
    def query_partial(self, q: dict, now=None):
        """Worker side: -> (partial-state blob, stats)."""
        out, n, st = C.c_void_p(), C.c_size_t(), Stats()
        _check(self.lib.vdb_query_partial(self.h, json.dumps(q).encode(), -1 if now is None else int(now), C.byref(out),
                                          C.byref(n), C.byref(st)))
        try:
            blob = C.string_at(out, n.value)
        finally:
            self.lib.vdb_free(out)
        return blob, {k: getattr(st, k) for k, _ in Stats._fields_}

    def query_merge(self, q: dict, blobs):
        """Controller side: merges the workers' blobs on the GPU and finishes the query -> (rows, stats)."""
        blobs = [bytes(b) for b in blobs]
        arr = (C.c_char_p * max(1, len(blobs)))()
        lens = (C.c_size_t * max(1, len(blobs)))()
        keep = []
        for i, b in enumerate(blobs):
            buf = C.create_string_buffer(b, len(b))
            keep.append(buf)
            arr[i] = C.cast(buf, C.c_char_p)
            lens[i] = len(b)
        out, n, st = C.c_void_p(), C.c_size_t(), Stats()
        _check(self.lib.vdb_query_merge(self.h, json.dumps(q).encode(), arr, lens, len(blobs), C.byref(out), C.byref(n), C.byref(st)))
        try:
            raw = C.string_at(out, n.value).decode("utf-8", "replace")
        finally:
            self.lib.vdb_free(out)
        rows = [r.split(FS)[:-1] for r in raw.split(RS)[:-1]]
        return rows, {k: getattr(st, k) for k, _ in Stats._fields_}

    def table_info(self, table: str):
        a, b = C.c_uint64(), C.c_uint64()
        _check(self.lib.vdb_table_info(self.h, table.encode(), C.byref(a), C.byref(b)))
        return {"segments": a.value, "first_segment_size": b.value}


def shim_text(table_json: str, query_json=None) -> str:
    """viya::shim::codegen::AggQueryText(table, query) — or UpsertHookText() when query_json is None — through the C facade."""
    lib = load()
    out, n = C.c_void_p(), C.c_size_t()
    _check(lib.vdb_shim_text(table_json.encode() if table_json is not None else None, query_json.encode() if query_json is not None else None, C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value).decode()
    finally:
        lib.vdb_free(out)
