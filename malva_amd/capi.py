"""ctypes binding of include/malva_hip.h.  No fallback: a missing or unloadable
libmalva_hip.so is an error.

One HIP runtime per process.  libmalva_hip.so asks the loader for `libamdhip64.so.7` and gets whichever copy is
already mapped, else /opt/rocm's.  PyTorch-ROCm bundles its own copy and asks for it by FILE name
(`libamdhip64.so`), which never matches an already mapped /opt/rocm copy: a process that mapped /opt/rocm's first
and imports torch later ends up with two runtimes, and torch then reports "No HIP GPUs".  lib() therefore maps
torch's copy first whenever torch is installed (without importing torch), so the order of `import torch` and
Context() no longer matters.  MALVA_HIP_RUNTIME=system keeps /opt/rocm's (for processes that never import torch)."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

BF_ALT, BF_CTX = 0, 1
STREAM_DEFAULT = 1          # MG_STREAM_DEFAULT: HIP's legacy default stream (a NULL handle means the context's own stream)
COMM_NONE, COMM_RCCL, COMM_LOCAL = 0, 1, 2
COMM_ID_BYTES = 128
GT_NORMAL, GT_OVERCOV, GT_SINGLE, GT_NOCOV = 0, 1, 2, 3

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class MalvaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("malva_hip error %d: %s" % (code, msg))
        self.code = code


def library_path():
    return os.path.join(_HERE, "lib", "libmalva_hip.so")


def _map_process_hip_runtime():
    """see the module docstring; returns the path mapped, or None"""
    if os.environ.get("MALVA_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return None                      # torch already imported: its runtime is mapped and ours will bind to it
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    C.CDLL(path, mode=C.RTLD_GLOBAL)
    return path


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise MalvaError(-100, "%s not built: run `make lib` (hipcc --offload-arch=gfx950)" % path)
    _map_process_hip_runtime()
    L = C.CDLL(path)
    vp, cp, sz, u64, u32, i32, i64 = C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int32, C.c_int64
    fl, it = C.c_float, C.c_int
    sig = {
        "mg_create": [C.POINTER(vp), it, u32, u32, u64],
        "mg_destroy": [vp],
        "mg_set_stream": [vp, vp],
        "mg_synchronize": [vp],
        "mg_bf_insert": [vp, it, vp, sz, sz],
        "mg_bf_test": [vp, it, vp, sz, sz, vp],
        "mg_bf_finalize": [vp, it],
        "mg_bf_increment": [vp, it, vp, sz, sz, vp],
        "mg_bf_get_count": [vp, it, vp, sz, sz, vp],
        "mg_bf_info": [vp, it, vp, vp, vp],
        "mg_map_insert": [vp, vp, sz, sz],
        "mg_map_test": [vp, vp, sz, sz, vp],
        "mg_map_increment": [vp, vp, sz, sz, vp],
        "mg_map_get_count": [vp, vp, sz, sz, vp],
        "mg_map_size": [vp, vp],
        "mg_ref_scan": [vp, vp, sz],
        "mg_ref_scan_resident": [vp, u64, sz],
        "mg_reference_upload_device": [vp, vp, sz],
        "mg_kmc_scan": [vp, vp, vp, vp, sz],
        "mg_kmc_scan_device": [vp, vp, vp, vp, sz],
        "mg_kmc_pack_rows_device": [vp, vp, vp, vp, sz, vp],
        "mg_kmc_scan_rows_device": [vp, vp, sz],
        "mg_host_alloc": [C.POINTER(vp), sz],
        "mg_host_free": [vp],
        "mg_kmc_set_lut": [vp, vp, sz, u32, u32, u32, u32, u64, u64],
        "mg_kmc_scan_records": [vp, vp, sz, u64],
        "mg_kmc_decode_records": [vp, vp, sz, u64, vp, vp, vp],
        "mg_counters_size": [vp, vp, vp],
        "mg_counters_export_device": [vp, vp],
        "mg_counters_import_device": [vp, vp],
        "mg_counters_reset": [vp],
        "mg_counters_view": [vp, vp, vp, vp],
        "mg_comm_unique_id": [vp],
        "mg_comm_init": [vp, it, it, vp],
        "mg_comm_init_all": [C.POINTER(vp), it],
        "mg_comm_destroy": [vp],
        "mg_comm_info": [vp, vp, vp, vp],
        "mg_counters_allreduce": [vp],
        "mg_counters_allreduce_all": [C.POINTER(vp), it],
        "mg_counters_allreduce_begin": [vp],
        "mg_counters_allreduce_end": [vp],
        "mg_exchange_stats": [vp, vp, vp],
        "mg_lookup_cover": [vp, vp, sz, sz, vp, vp, sz, vp, sz, vp],
        "mg_genotype": [vp, vp, vp, vp, sz, fl, it, it, vp, vp, vp, vp, vp, vp],
        "mg_cover_blocks": [vp, sz, vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, u32, it, vp, vp],
        "mg_decode_gt_text": [vp, vp, sz, sz, vp, vp, vp, u32, vp, it, vp, vp, vp, vp, vp],
        "mg_decode_gt_entries": [vp, vp, vp],
        "mg_cover_blocks_sparse": [vp, sz, vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, C.c_uint16, u32, it, vp, vp],
        "mg_index_blocks_sparse": [vp, sz, vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, C.c_uint16, u32, it, vp],
        "mg_cut_blocks": [vp, sz, vp, vp, vp, vp, vp, vp],
        "mg_cut_blocks_device": [vp, vp, vp, vp, vp],
        "mg_cover_blocks_device": [vp, vp, vp, vp, vp, it, vp, vp],
        "mg_index_blocks_device": [vp, vp, vp, vp, vp, it, vp],
        "mg_genotype_device": [vp, vp, vp, vp, sz, fl, it, it, vp, vp, vp, vp, vp, vp],
        "mg_index_isolated": [vp, sz, vp, vp, vp, vp, sz, vp, vp, vp],
        "mg_index_blocks": [vp, sz, vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, u32, it, vp],
        "mg_reference_upload": [vp, vp, sz],
        "mg_call_isolated": [vp, sz, vp, vp, vp, vp, sz, vp, vp, vp, fl, it, it, vp, vp, vp, vp, vp, vp, vp],
        "mg_call_isolated_device": [vp, sz, vp, vp, vp, vp, vp, vp, vp, fl, it, it, vp, vp, vp, vp, vp, vp, vp],
        "mg_bf_export": [vp, it, vp, vp],
        "mg_bf_import": [vp, it, it, u64, vp, vp, u64],
        "mg_bf_export_sparse": [vp, it, vp, vp],
        "mg_bf_import_sparse": [vp, it, it, u64, vp, vp, u64],
        "mg_map_export": [vp, vp, sz, vp],
        "mg_map_import": [vp, vp, sz, sz, vp],
        "mg_debug_bf_index": [vp, it, vp, sz, sz, vp],
        "mg_debug_packed_index": [vp, it, vp, vp, sz, u32, vp],
        "mg_scan_stats": [vp, vp, vp],
        "mg_blocks_stats": [vp, vp, vp],
        "mg_set_option": [vp, cp, i64],
        "mg_get_option": [vp, cp, C.POINTER(C.c_int64)],
    }
    for name, args in sig.items():
        f = getattr(L, name)          # AttributeError if the library lacks a declared symbol
        f.argtypes = args
        f.restype = C.c_int
    L.mg_kmc_rows_bytes.argtypes = [sz]
    L.mg_kmc_rows_bytes.restype = C.c_size_t
    L.mg_last_error.argtypes = [vp]
    L.mg_last_error.restype = cp
    _LIB = L
    return L


EXPORTED = ["mg_create", "mg_destroy", "mg_last_error", "mg_set_stream", "mg_synchronize", "mg_bf_insert", "mg_bf_test",
            "mg_bf_finalize", "mg_bf_increment", "mg_bf_get_count", "mg_bf_info", "mg_map_insert", "mg_map_test",
            "mg_map_increment", "mg_map_get_count", "mg_map_size", "mg_ref_scan", "mg_ref_scan_resident", "mg_reference_upload_device", "mg_kmc_scan", "mg_kmc_scan_device", "mg_kmc_rows_bytes", "mg_kmc_pack_rows_device", "mg_kmc_scan_rows_device",
            "mg_host_alloc", "mg_host_free", "mg_kmc_set_lut", "mg_kmc_scan_records", "mg_kmc_decode_records",
            "mg_counters_size", "mg_counters_export_device", "mg_counters_import_device", "mg_counters_reset", "mg_counters_view",
            "mg_comm_unique_id", "mg_comm_init", "mg_comm_init_all", "mg_comm_destroy", "mg_comm_info", "mg_counters_allreduce",
            "mg_counters_allreduce_all", "mg_counters_allreduce_begin", "mg_counters_allreduce_end", "mg_exchange_stats", "mg_decode_gt_text", "mg_decode_gt_entries", "mg_cut_blocks", "mg_cut_blocks_device", "mg_cover_blocks_device", "mg_index_blocks_device", "mg_genotype_device",
            "mg_index_isolated",
            "mg_lookup_cover", "mg_cover_blocks", "mg_index_blocks", "mg_cover_blocks_sparse", "mg_index_blocks_sparse", "mg_genotype", "mg_reference_upload", "mg_call_isolated", "mg_call_isolated_device",
            "mg_bf_export", "mg_bf_import", "mg_bf_export_sparse", "mg_bf_import_sparse", "mg_map_export", "mg_map_import", "mg_debug_bf_index",
            "mg_debug_packed_index", "mg_scan_stats", "mg_blocks_stats", "mg_set_option", "mg_get_option"]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class PanelDev(C.Structure):
    """mg_panel_dev: a panel resident in HBM (every pointer a device pointer)"""
    _fields_ = [("n_vars", C.c_uint64), ("n_contigs", C.c_uint32), ("n_samples", C.c_uint32), ("contig_base", C.c_void_p), ("contig_len", C.c_void_p),
                ("contig_id", C.c_void_p), ("pos", C.c_void_p), ("ref_size", C.c_void_p), ("min_size", C.c_void_p), ("present", C.c_void_p),
                ("var_allele_off", C.c_void_p), ("allele_off", C.c_void_p), ("pool", C.c_void_p), ("canon", C.c_void_p), ("gt", C.c_void_p),
                ("sp_off", C.c_void_p), ("sp_sample", C.c_void_p), ("sp_gt", C.c_void_p), ("sp_default", C.c_uint32), ("pool_bytes", C.c_uint64)]


def sparse_genotypes(gt, n_samples, default=1 << 14):
    """dense [n_vars, n_samples] genotype words -> (sp_off, sp_sample, sp_gt): the entries other than `default` (0|0 phased)"""
    gt = np.ascontiguousarray(gt, dtype=np.uint16).reshape(-1, max(int(n_samples), 1)) if n_samples else np.zeros((len(gt), 0), np.uint16)
    keep = gt != np.uint16(default)
    off = np.zeros(gt.shape[0] + 1, dtype=np.uint32)
    off[1:] = np.cumsum(keep.sum(axis=1))
    rows, cols = np.nonzero(keep)
    return off, cols.astype(np.uint32), gt[rows, cols].astype(np.uint16)


def host_alloc(n_bytes):
    """pinned host memory (mg_host_alloc) as a uint8 array; the array must be passed to host_free() when done"""
    ptr = C.c_void_p()
    rc = lib().mg_host_alloc(C.byref(ptr), n_bytes)
    if rc != 0:
        raise MalvaError(rc, "mg_host_alloc(%d) failed" % n_bytes)
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n_bytes,))
    return arr, ptr


def host_free(ptr):
    lib().mg_host_free(ptr)


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library: rank 0 calls it and ships the bytes to the other ranks"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().mg_comm_unique_id(buf)
    if rc != 0:
        raise MalvaError(rc, "mg_comm_unique_id failed (librccl.so.1 not loadable?)")
    return buf.raw


def _ctx_array(ctxs):
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    return arr


def comm_init_all(ctxs):
    """one process, several contexts: one per device (RCCL group) or all on one device (kernel sum)"""
    rc = lib().mg_comm_init_all(_ctx_array(ctxs), len(ctxs))
    if rc != 0:
        raise MalvaError(rc, lib().mg_last_error(ctxs[0].h).decode())


def counters_allreduce_all(ctxs):
    rc = lib().mg_counters_allreduce_all(_ctx_array(ctxs), len(ctxs))
    if rc != 0:
        raise MalvaError(rc, lib().mg_last_error(ctxs[0].h).decode())


def rows_of(kmers, stride=None):
    """list of bytes -> contiguous uint8 [n, stride], NUL padded"""
    n = len(kmers)
    if stride is None:
        stride = (max((len(k) for k in kmers), default=1) + 1 + 7) // 8 * 8
    arr = np.zeros((n, stride), dtype=np.uint8)
    for i, k in enumerate(kmers):
        if len(k) >= stride:
            raise ValueError("k-mer longer than the row stride")
        arr[i, : len(k)] = np.frombuffer(k, dtype=np.uint8)
    return arr


def _rows(rows):
    if isinstance(rows, (list, tuple)):
        rows = rows_of(rows)
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    assert rows.ndim == 2
    return rows


class Context:
    """One GPU's filters, exact map and kernels (mg_ctx)."""

    def __init__(self, k=35, ref_k=43, bf_bits=1 << 33, device=0):
        self._L = lib()
        self.h = C.c_void_p()
        rc = self._L.mg_create(C.byref(self.h), device, k, ref_k, bf_bits)
        if rc != 0:
            self.h = None
            raise MalvaError(rc, "mg_create failed (no usable HIP device, bad sizes or out of memory)")
        self.k, self.ref_k, self.bf_bits = k, ref_k, bf_bits

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self._L.mg_destroy(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the library may already be gone
            pass

    def _ck(self, rc):
        if rc != 0:
            raise MalvaError(rc, self._L.mg_last_error(self.h).decode())

    # lifetime / options
    def set_stream(self, stream_handle):
        self._ck(self._L.mg_set_stream(self.h, C.c_void_p(stream_handle)))

    def synchronize(self):
        self._ck(self._L.mg_synchronize(self.h))

    def set_option(self, name, value):
        self._ck(self._L.mg_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int64(0)
        self._ck(self._L.mg_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    # BF
    def bf_insert(self, which, rows):
        rows = _rows(rows)
        self._ck(self._L.mg_bf_insert(self.h, which, _p(rows), rows.shape[1], rows.shape[0]))

    def bf_test(self, which, rows):
        rows = _rows(rows)
        out = np.zeros(rows.shape[0], dtype=np.uint8)
        self._ck(self._L.mg_bf_test(self.h, which, _p(rows), rows.shape[1], rows.shape[0], _p(out)))
        return out.astype(bool)

    def bf_finalize(self, which):
        self._ck(self._L.mg_bf_finalize(self.h, which))

    def bf_increment(self, which, rows, counters):
        rows = _rows(rows)
        counters = np.ascontiguousarray(counters, dtype=np.uint32)
        assert counters.shape[0] == rows.shape[0]
        self._ck(self._L.mg_bf_increment(self.h, which, _p(rows), rows.shape[1], rows.shape[0], _p(counters)))

    def bf_get_count(self, which, rows):
        rows = _rows(rows)
        out = np.zeros(rows.shape[0], dtype=np.uint16)
        self._ck(self._L.mg_bf_get_count(self.h, which, _p(rows), rows.shape[1], rows.shape[0], _p(out)))
        return out

    def bf_info(self, which):
        size, nset, mode = C.c_uint64(), C.c_uint64(), C.c_int()
        self._ck(self._L.mg_bf_info(self.h, which, C.byref(size), C.byref(nset), C.byref(mode)))
        return size.value, nset.value, mode.value

    def bf_index(self, which, rows):
        rows = _rows(rows)
        out = np.zeros(rows.shape[0], dtype=np.uint64)
        self._ck(self._L.mg_debug_bf_index(self.h, which, _p(rows), rows.shape[1], rows.shape[0], _p(out)))
        return out

    def packed_index(self, which, hi, lo, klen):
        hi = np.ascontiguousarray(hi, dtype=np.uint64)
        lo = np.ascontiguousarray(lo, dtype=np.uint64)
        out = np.zeros(hi.shape[0], dtype=np.uint64)
        self._ck(self._L.mg_debug_packed_index(self.h, which, _p(hi), _p(lo), hi.shape[0], klen, _p(out)))
        return out

    def bf_export(self, which):
        size, nset, mode = self.bf_info(which)
        words = np.zeros((size + 63) // 64, dtype=np.uint64)
        counts = np.zeros(nset if mode else 0, dtype=np.uint16)
        self._ck(self._L.mg_bf_export(self.h, which, _p(words), _p(counts) if counts.size else None))
        return mode, size, words, counts

    def bf_import(self, which, mode, size, words, counts):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint16)
        self._ck(self._L.mg_bf_import(self.h, which, int(mode), size, _p(words), _p(counts) if counts.size else None,
                                      counts.size))

    def bf_export_sparse(self, which):
        size, nset, mode = self.bf_info(which)
        pos = np.zeros(nset, dtype=np.uint64)
        counts = np.zeros(nset, dtype=np.uint16)
        self._ck(self._L.mg_bf_export_sparse(self.h, which, _p(pos) if nset else None, _p(counts) if nset else None))
        return mode, size, pos, counts

    def bf_import_sparse(self, which, mode, size, pos, counts):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint16)
        self._ck(self._L.mg_bf_import_sparse(self.h, which, int(mode), size, _p(pos) if pos.size else None,
                                             _p(counts) if counts.size else None, pos.size))

    # KMAP
    def map_insert(self, rows):
        rows = _rows(rows)
        self._ck(self._L.mg_map_insert(self.h, _p(rows), rows.shape[1], rows.shape[0]))

    def map_test(self, rows):
        rows = _rows(rows)
        out = np.zeros(rows.shape[0], dtype=np.uint8)
        self._ck(self._L.mg_map_test(self.h, _p(rows), rows.shape[1], rows.shape[0], _p(out)))
        return out.astype(bool)

    def map_increment(self, rows, counters):
        rows = _rows(rows)
        counters = np.ascontiguousarray(counters, dtype=np.int32)
        self._ck(self._L.mg_map_increment(self.h, _p(rows), rows.shape[1], rows.shape[0], _p(counters)))

    def map_get_count(self, rows):
        rows = _rows(rows)
        out = np.zeros(rows.shape[0], dtype=np.int32)
        self._ck(self._L.mg_map_get_count(self.h, _p(rows), rows.shape[1], rows.shape[0], _p(out)))
        return out

    def map_size(self):
        n = C.c_uint64()
        self._ck(self._L.mg_map_size(self.h, C.byref(n)))
        return n.value

    def map_export(self):
        n = self.map_size()
        stride = (self.k + 1 + 7) // 8 * 8
        rows = np.zeros((n, stride), dtype=np.uint8)
        vals = np.zeros(n, dtype=np.int32)
        if n:
            self._ck(self._L.mg_map_export(self.h, _p(rows), stride, _p(vals)))
        keys = [bytes(r).split(b"\0", 1)[0] for r in rows]
        return keys, vals

    def map_import(self, keys, vals):
        rows = _rows(keys)
        vals = np.ascontiguousarray(vals, dtype=np.int32)
        self._ck(self._L.mg_map_import(self.h, _p(rows), rows.shape[1], rows.shape[0], _p(vals)))

    # scans
    def ref_scan(self, contig):
        """contig: bytes, or a uint8 array (no copy: a whole-genome contig is gigabytes)"""
        buf = np.frombuffer(contig, dtype=np.uint8) if isinstance(contig, (bytes, bytearray, memoryview)) else np.ascontiguousarray(contig, dtype=np.uint8)
        self._ck(self._L.mg_ref_scan(self.h, _p(buf), buf.size))

    def ref_scan_resident(self, offset, length):
        """the contig at [offset, offset + length) of the uploaded reference (no PCIe)"""
        self._ck(self._L.mg_ref_scan_resident(self.h, int(offset), int(length)))

    def reference_upload_device(self, d_ptr, length):
        self._ck(self._L.mg_reference_upload_device(self.h, C.c_void_p(d_ptr), int(length)))

    def kmc_scan(self, hi, lo, cnt):
        hi = np.ascontiguousarray(hi, dtype=np.uint64)
        lo = np.ascontiguousarray(lo, dtype=np.uint64)
        cnt = np.ascontiguousarray(cnt, dtype=np.uint32)
        assert hi.shape == lo.shape == cnt.shape
        self._ck(self._L.mg_kmc_scan(self.h, _p(hi), _p(lo), _p(cnt), hi.shape[0]))

    def kmc_scan_device(self, d_hi, d_lo, d_cnt, n):
        self._ck(self._L.mg_kmc_scan_device(self.h, C.c_void_p(d_hi), C.c_void_p(d_lo), C.c_void_p(d_cnt), n))

    # KMC database feed
    def kmc_set_lut(self, lut, lut_prefix_len, suffix_bytes, counter_bytes, min_count, max_count, total_records):
        lut = np.ascontiguousarray(lut, dtype=np.uint64)
        self._ck(self._L.mg_kmc_set_lut(self.h, _p(lut), lut.size, lut_prefix_len, suffix_bytes, counter_bytes, min_count, max_count,
                                        total_records))
        self._kmc_rec = suffix_bytes + counter_bytes

    def kmc_scan_records(self, records, first_record=0):
        records = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1)
        assert records.size % self._kmc_rec == 0
        self._ck(self._L.mg_kmc_scan_records(self.h, _p(records), records.size // self._kmc_rec, first_record))

    def kmc_decode_records(self, records, first_record=0):
        records = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1)
        n = records.size // self._kmc_rec
        hi, lo, cnt = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint32)
        self._ck(self._L.mg_kmc_decode_records(self.h, _p(records), n, first_record, _p(hi), _p(lo), _p(cnt)))
        return hi, lo, cnt

    def kmc_rows_bytes(self, n):
        return self._L.mg_kmc_rows_bytes(n)

    def kmc_pack_rows_device(self, d_hi, d_lo, d_cnt, n, d_rows_out):
        self._ck(self._L.mg_kmc_pack_rows_device(self.h, C.c_void_p(d_hi), C.c_void_p(d_lo), C.c_void_p(d_cnt), n, C.c_void_p(d_rows_out)))

    def kmc_scan_rows_device(self, d_rows, n):
        self._ck(self._L.mg_kmc_scan_rows_device(self.h, C.c_void_p(d_rows), n))

    def scan_stats(self):
        """-> (filter ms, probe ms, hits ms, open rows, bf-hit rows)"""
        ms = (C.c_float * 3)()
        nr = (C.c_uint64 * 2)()
        self._ck(self._L.mg_scan_stats(self.h, ms, nr))
        return float(ms[0]), float(ms[1]), float(ms[2]), int(nr[0]), int(nr[1])

    def blocks_stats(self):
        """-> (tier 1 ms, tier 2 ms, tier 3 ms, records beyond tier 1, lone signature k-mers, general signature k-mers, records tier 3 took)"""
        ms = (C.c_float * 3)()
        nr = (C.c_uint64 * 4)()
        self._ck(self._L.mg_blocks_stats(self.h, ms, nr))
        return float(ms[0]), float(ms[1]), float(ms[2]), int(nr[0]), int(nr[1]), int(nr[2]), int(nr[3])

    # counters exchange
    def decode_gt_text(self, text, span_off, span_len, gt_index, n_columns, keep=None, haploid=False):
        """-> (sp_default, sp_off, sp_sample, sp_gt, raw_mask, max_allele): the panel genotypes of a batch of records from
        the text of their sample columns (mg_decode_gt_text + mg_decode_gt_entries)"""
        text = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
        so, sl = np.ascontiguousarray(span_off, dtype=np.uint64), np.ascontiguousarray(span_len, dtype=np.uint32)
        gi = np.ascontiguousarray(gt_index, dtype=np.int32)
        n = len(so)
        kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
        dflt, ne = C.c_uint16(), C.c_uint64()
        sp_off = np.zeros(n + 1, dtype=np.uint32)
        mask, mx = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint32)
        self._ck(self._L.mg_decode_gt_text(self.h, _p(text), text.size, n, _p(so), _p(sl), _p(gi), n_columns, _p(kp), int(haploid), C.byref(dflt), _p(sp_off),
                                           _p(mask), _p(mx), C.byref(ne)))
        ss, sg = np.zeros(ne.value, dtype=np.uint32), np.zeros(ne.value, dtype=np.uint16)
        if n:
            self._ck(self._L.mg_decode_gt_entries(self.h, _p(ss), _p(sg)))
        return dflt.value, sp_off, ss, sg, mask, mx

    def counters_size(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self._L.mg_counters_size(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def counters_export_device(self, d_ptr):
        self._ck(self._L.mg_counters_export_device(self.h, C.c_void_p(d_ptr)))

    def counters_import_device(self, d_ptr):
        self._ck(self._L.mg_counters_import_device(self.h, C.c_void_p(d_ptr)))

    def counters_view(self):
        """-> (device pointer, n_bf, n_map) of the contiguous [bf | map] u32 counter vector"""
        p, a, b = C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._ck(self._L.mg_counters_view(self.h, C.byref(p), C.byref(a), C.byref(b)))
        return p.value, a.value, b.value

    def counters_reset(self):
        self._ck(self._L.mg_counters_reset(self.h))

    # multi-GPU exchange inside the library (RCCL)
    def comm_init(self, rank, world, comm_id: bytes):
        assert len(comm_id) == COMM_ID_BYTES
        buf = C.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        self._ck(self._L.mg_comm_init(self.h, rank, world, buf))

    def comm_destroy(self):
        self._ck(self._L.mg_comm_destroy(self.h))

    def comm_info(self):
        r, w, b = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._L.mg_comm_info(self.h, C.byref(r), C.byref(w), C.byref(b)))
        return r.value, w.value, b.value

    def counters_allreduce(self):
        self._ck(self._L.mg_counters_allreduce(self.h))

    def counters_allreduce_begin(self):
        """the exchange on a stream of its own, behind what the context's stream holds; counter-free work may follow"""
        self._ck(self._L.mg_counters_allreduce_begin(self.h))

    def counters_allreduce_end(self):
        self._ck(self._L.mg_counters_allreduce_end(self.h))

    def exchange_stats(self):
        """-> (ms of the most recent exchange, 1 if it ran in the 16-bit packed form)"""
        ms, pk = C.c_float(0), C.c_int(0)
        self._ck(self._L.mg_exchange_stats(self.h, C.byref(ms), C.byref(pk)))
        return float(ms.value), int(pk.value)

    # per-variant path
    def lookup_cover(self, rows, is_ref, sig_kmer_off, allele_sig_off):
        so = np.ascontiguousarray(sig_kmer_off, dtype=np.uint64)
        ao = np.ascontiguousarray(allele_sig_off, dtype=np.uint64)
        n_alleles, n_sigs = len(ao) - 1, len(so) - 1
        cov = np.zeros(n_alleles, dtype=np.uint32)
        if isinstance(rows, (list, tuple)) and len(rows) == 0:
            rows = np.zeros((0, 8), dtype=np.uint8)
        rows = _rows(rows)
        is_ref = np.ascontiguousarray(is_ref, dtype=np.uint8)
        self._ck(self._L.mg_lookup_cover(self.h, _p(rows), rows.shape[1], rows.shape[0], _p(is_ref), _p(so), n_sigs,
                                         _p(ao), n_alleles, _p(cov)))
        return cov

    def cover_blocks(self, blk_ref_base, blk_ref_len, blk_var_off, pos, ref_size, min_size, present, var_allele_off,
                     allele_off, pool, canon, gt, n_samples, haploid, sparse=False, sp_default=1 << 14):
        """sparse: hand the genotypes over as the entries other than the word sp_default (mg_cover_blocks_sparse)"""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        bb, bl, bo = a(blk_ref_base, np.uint64), a(blk_ref_len, np.uint32), a(blk_var_off, np.uint32)
        pos, rs, ms, pr = a(pos, np.int32), a(ref_size, np.uint32), a(min_size, np.uint32), a(present, np.uint8)
        vo, ao, pool, canon, gt = a(var_allele_off, np.uint32), a(allele_off, np.uint32), a(pool, np.uint8), a(canon, np.uint8), a(gt, np.uint16)
        n = len(pos)
        cov = np.zeros(int(vo[-1]), dtype=np.uint32)
        ovf = np.zeros(n, dtype=np.uint8)
        if sparse:
            so, ss, sg = sparse_genotypes(gt.reshape(n, -1) if n else gt, n_samples, sp_default)
            self._ck(self._L.mg_cover_blocks_sparse(self.h, len(bb), _p(bb), _p(bl), _p(bo), n, _p(pos), _p(rs), _p(ms), _p(pr), _p(vo), _p(ao),
                                                    _p(pool), pool.size, _p(canon), _p(so), _p(ss), _p(sg), sp_default, n_samples, int(haploid), _p(cov), _p(ovf)))
            return cov, ovf
        self._ck(self._L.mg_cover_blocks(self.h, len(bb), _p(bb), _p(bl), _p(bo), n, _p(pos), _p(rs), _p(ms), _p(pr), _p(vo), _p(ao),
                                         _p(pool), pool.size, _p(canon), _p(gt), n_samples, int(haploid), _p(cov), _p(ovf)))
        return cov, ovf

    def index_isolated(self, pos, var_allele_off, allele_off, pool, present_mask, flags):
        """-> overflow flags; lone short variants indexed on the device (extract_kmers + add_kmers_to_bf)"""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        pos, vo, ao, pool, pm, fl = a(pos, np.uint64), a(var_allele_off, np.uint32), a(allele_off, np.uint32), a(pool, np.uint8), a(present_mask, np.uint64), a(flags, np.uint8)
        n = len(pos)
        ovf = np.zeros(n, dtype=np.uint8)
        self._ck(self._L.mg_index_isolated(self.h, n, _p(pos), _p(vo), _p(ao), _p(pool), pool.size, _p(pm), _p(fl), _p(ovf)))
        return ovf

    def cut_blocks(self, pos, ref_size, min_size, contig_id):
        """-> blk_var_off (n_blocks + 1 entries) of the kept records, cut as the reference's record loops cut them"""
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        pos, rs, ms, cid = a(pos, np.int32), a(ref_size, np.uint32), a(min_size, np.uint32), a(contig_id, np.uint32)
        n = len(pos)
        off = np.zeros(n + 1, dtype=np.uint32)
        nb = C.c_size_t(0)
        self._ck(self._L.mg_cut_blocks(self.h, n, _p(pos), _p(rs), _p(ms), _p(cid), _p(off), C.byref(nb)))
        return off[:nb.value + 1] if n else off[:0]

    # the record loop on a resident panel (device pointers; asynchronous)
    def cut_blocks_device(self, panel: PanelDev, d_blk_var_off, d_var_block, d_n_blocks):
        v = C.c_void_p
        self._ck(self._L.mg_cut_blocks_device(self.h, C.byref(panel), v(d_blk_var_off), v(d_var_block), v(d_n_blocks)))

    def cover_blocks_device(self, panel: PanelDev, d_blk_var_off, d_var_block, d_n_blocks, haploid, d_cov, d_overflow):
        v = C.c_void_p
        self._ck(self._L.mg_cover_blocks_device(self.h, C.byref(panel), v(d_blk_var_off), v(d_var_block), v(d_n_blocks), int(haploid), v(d_cov), v(d_overflow)))

    def index_blocks_device(self, panel: PanelDev, d_blk_var_off, d_var_block, d_n_blocks, haploid, d_overflow):
        v = C.c_void_p
        self._ck(self._L.mg_index_blocks_device(self.h, C.byref(panel), v(d_blk_var_off), v(d_var_block), v(d_n_blocks), int(haploid), v(d_overflow)))

    def genotype_device(self, d_cov, d_freq, d_var_allele_off, n_vars, error_rate, max_cov, haploid, d_g1, d_g2, d_gq, d_st, d_probs=None, d_gt_off=None):
        v = C.c_void_p
        self._ck(self._L.mg_genotype_device(self.h, v(d_cov), v(d_freq), v(d_var_allele_off), n_vars, C.c_float(error_rate), max_cov, int(haploid),
                                            v(d_g1), v(d_g2), v(d_gq), v(d_st), v(d_probs), v(d_gt_off)))

    def index_blocks(self, blk_ref_base, blk_ref_len, blk_var_off, pos, ref_size, min_size, present, var_allele_off,
                     allele_off, pool, canon, gt, n_samples, haploid, sparse=False, sp_default=1 << 14):
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        bb, bl, bo = a(blk_ref_base, np.uint64), a(blk_ref_len, np.uint32), a(blk_var_off, np.uint32)
        pos, rs, ms, pr = a(pos, np.int32), a(ref_size, np.uint32), a(min_size, np.uint32), a(present, np.uint8)
        vo, ao, pool, canon, gt = a(var_allele_off, np.uint32), a(allele_off, np.uint32), a(pool, np.uint8), a(canon, np.uint8), a(gt, np.uint16)
        n = len(pos)
        ovf = np.zeros(n, dtype=np.uint8)
        if sparse:
            so, ss, sg = sparse_genotypes(gt.reshape(n, -1) if n else gt, n_samples, sp_default)
            self._ck(self._L.mg_index_blocks_sparse(self.h, len(bb), _p(bb), _p(bl), _p(bo), n, _p(pos), _p(rs), _p(ms), _p(pr), _p(vo), _p(ao),
                                                    _p(pool), pool.size, _p(canon), _p(so), _p(ss), _p(sg), sp_default, n_samples, int(haploid), _p(ovf)))
            return ovf
        self._ck(self._L.mg_index_blocks(self.h, len(bb), _p(bb), _p(bl), _p(bo), n, _p(pos), _p(rs), _p(ms), _p(pr), _p(vo), _p(ao),
                                         _p(pool), pool.size, _p(canon), _p(gt), n_samples, int(haploid), _p(ovf)))
        return ovf

    def genotype(self, cov, freq, var_allele_off, error_rate, max_cov, haploid, want_probs=False):
        cov = np.ascontiguousarray(cov, dtype=np.uint32)
        freq = np.ascontiguousarray(freq, dtype=np.float32)
        vo = np.ascontiguousarray(var_allele_off, dtype=np.uint32)
        n = len(vo) - 1
        g1 = np.zeros(n, dtype=np.int32)
        g2 = np.zeros(n, dtype=np.int32)
        gq = np.zeros(n, dtype=np.int32)
        st = np.zeros(n, dtype=np.uint8)
        probs = goff = None
        if want_probs:
            A = np.diff(vo.astype(np.int64))
            ng = A if haploid else A * (A + 1) // 2
            goff = np.zeros(n + 1, dtype=np.uint64)
            goff[1:] = np.cumsum(ng)
            probs = np.zeros(int(goff[-1]), dtype=np.float64)
        self._ck(self._L.mg_genotype(self.h, _p(cov), _p(freq), _p(vo), n, C.c_float(error_rate), max_cov, int(haploid),
                                     _p(g1), _p(g2), _p(gq), _p(st), _p(probs), _p(goff)))
        return g1, g2, gq, st, probs, goff

    def reference_upload(self, ascii_bytes):
        buf = np.frombuffer(ascii_bytes, dtype=np.uint8) if isinstance(ascii_bytes, (bytes, bytearray)) else ascii_bytes
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self._ck(self._L.mg_reference_upload(self.h, _p(buf), buf.size))

    def call_isolated(self, pos, var_allele_off, allele_off, allele_pool, freq, present_mask, flags, error_rate,
                      max_cov, haploid, want_probs=False):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        vo = np.ascontiguousarray(var_allele_off, dtype=np.uint32)
        ao = np.ascontiguousarray(allele_off, dtype=np.uint32)
        pool = np.frombuffer(allele_pool, dtype=np.uint8) if isinstance(allele_pool, (bytes, bytearray)) else allele_pool
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        freq = np.ascontiguousarray(freq, dtype=np.float32)
        pm = np.ascontiguousarray(present_mask, dtype=np.uint64)
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        n, na = len(pos), int(vo[-1])
        cov = np.zeros(na, dtype=np.uint32)
        g1 = np.zeros(n, dtype=np.int32)
        g2 = np.zeros(n, dtype=np.int32)
        gq = np.zeros(n, dtype=np.int32)
        st = np.zeros(n, dtype=np.uint8)
        probs = goff = None
        if want_probs:
            A = np.diff(vo.astype(np.int64))
            goff = np.zeros(n + 1, dtype=np.uint64)
            goff[1:] = np.cumsum(A if haploid else A * (A + 1) // 2)
            probs = np.zeros(int(goff[-1]), dtype=np.float64)
        self._ck(self._L.mg_call_isolated(self.h, n, _p(pos), _p(vo), _p(ao), _p(pool), pool.size, _p(freq), _p(pm),
                                          _p(fl), C.c_float(error_rate), max_cov, int(haploid), _p(cov), _p(g1), _p(g2),
                                          _p(gq), _p(st), _p(probs), _p(goff)))
        if want_probs:
            return cov, g1, g2, gq, st, probs, goff
        return cov, g1, g2, gq, st

    def call_isolated_device(self, n, d_pos, d_vo, d_ao, d_pool, d_freq, d_pm, d_flags, error_rate, max_cov, haploid,
                             d_cov, d_g1, d_g2, d_gq, d_st, d_probs=None, d_gt_off=None):
        v = C.c_void_p
        self._ck(self._L.mg_call_isolated_device(self.h, n, v(d_pos), v(d_vo), v(d_ao), v(d_pool), v(d_freq), v(d_pm),
                                                 v(d_flags), C.c_float(error_rate), max_cov, int(haploid), v(d_cov),
                                                 v(d_g1), v(d_g2), v(d_gq), v(d_st), v(d_probs), v(d_gt_off)))
