"""ctypes binding of libgmhip.so (include/gmhip.h).  There is no CPU fallback: if the HIP
library is missing, or no MI355X is visible, every compute call raises."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMHIP_LIB", os.path.join(_HERE, "libgmhip.so"))   # override: A/B builds of the library

METRIC_HAMMING = 0
METRIC_LEVEN = 1
MAX_GUIDELEN = 27
MAX_K = 64

_lib = None
_torch_lib_dir = None      # set when the HIP runtime in use is the copy shipped with PyTorch


class GmHipError(RuntimeError):
    pass


def build(force=False):
    """Compile guidemaker_amd/libgmhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        args.append("-B")
    subprocess.check_call(args)


def _share_torch_hip_runtime():
    """PyTorch's ROCm wheels ship their own libamdhip64.so.7 and HSA runtime.  A process that loads the SYSTEM copy
    first (through libgmhip.so) and imports torch afterwards leaves torch's libraries bound to a runtime they were not
    built with, and torch then reports "No HIP GPUs are available" (seen on the GPU box: `from guidemaker_amd import
    _lib; _lib.device_count(); import torch; torch.zeros(1).cuda()`).  When torch is installed, load ITS copy of the
    runtime before libgmhip.so -- same SONAME, so libgmhip.so binds to it and a later `import torch` finds its own
    runtime already in place -- without importing torch.  GMHIP_SYSTEM_RUNTIME=1 keeps the system's runtime (for
    processes that will never import torch)."""
    global _torch_lib_dir
    import sys
    if os.environ.get("GMHIP_SYSTEM_RUNTIME", "0") not in ("", "0"):
        return
    if "torch" in sys.modules:      # torch is already in: its runtime is the one loaded
        _torch_lib_dir = os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib")
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    d = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    rt = os.path.join(d, "libamdhip64.so")
    if os.path.exists(rt):
        try:
            ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
            _torch_lib_dir = d
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "guidemaker_amd: %s not found. Build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C guidemaker_amd/csrc). "
            "There is no CPU fallback." % LIB_PATH)
    _share_torch_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, i64, i32 = c.c_void_p, c.c_int64, c.c_int
    L.gm_version.restype = c.c_char_p
    L.gm_last_error.restype = c.c_char_p
    L.gm_device_count.restype = i32
    L.gm_pack.argtypes = [c.c_char_p, i64, i32, vp]
    L.gm_unpack.argtypes = [vp, i64, i32, vp]
    L.gm_format_doubles.argtypes = [vp, i64, vp, i64]
    L.gm_format_doubles.restype = i64
    L.gm_format_neighbor_rows.argtypes = [vp, i64, i32, vp, vp, vp, i64, i32, vp, i64, vp, i64, c.POINTER(i64), c.POINTER(i64)]
    L.gm_index_create.argtypes = [vp, i64, i32, i32, i32, c.POINTER(vp)]
    L.gm_index_create_dev.argtypes = [vp, i64, i32, i32, i32, vp, c.POINTER(vp)]
    L.gm_index_free.argtypes = [vp]
    L.gm_index_free.restype = None
    L.gm_index_size.argtypes = [vp]
    L.gm_index_size.restype = i64
    L.gm_index_guidelen.argtypes = [vp]
    L.gm_index_metric.argtypes = [vp]
    L.gm_knn_query.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.gm_knn_query_dev.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp]
    L.gm_nearest_dist.argtypes = [vp, vp, i64, vp]
    L.gm_nearest_dist_dev.argtypes = [vp, vp, i64, vp, vp]
    L.gm_cfd_batch.argtypes = [vp, vp, i64, i32, vp, vp, i32]
    L.gm_cfd_batch_dev.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.gm_last_kernel_ms.argtypes = [vp]
    L.gm_last_kernel_ms.restype = c.c_double
    L.gm_last_query_ms.argtypes = [vp]
    L.gm_last_query_ms.restype = c.c_double
    L.gm_last_kernel_name.argtypes = [vp]
    L.gm_last_kernel_name.restype = c.c_char_p
    L.gm_last_kernel_pairs.argtypes = [vp]
    L.gm_last_kernel_pairs.restype = i64
    L.gm_last_kernel_evaluated.argtypes = [vp]
    L.gm_last_kernel_evaluated.restype = i64
    L.gm_last_incomplete.argtypes = [vp]
    L.gm_last_incomplete.restype = i64
    L.gm_pigeon_blocks.argtypes = [vp]
    L.gm_pigeon_blocks.restype = i32
    L.gm_pigeon_block_info.argtypes = [vp, i32, c.POINTER(i32), c.POINTER(i64)]
    L.gm_pigeon_block_info.restype = i32
    L.gm_pigeon_plan.argtypes = [i64, i32, i32, c.c_double]
    L.gm_pigeon_plan.restype = i32
    L.gm_pigeon_plan_dist.argtypes = [i64, i64, i32, i32, c.c_double, c.POINTER(c.c_double), i32, c.POINTER(i32),
                                      c.POINTER(i32), c.POINTER(c.c_double), c.POINTER(i32)]
    L.gm_pigeon_plan_dist.restype = i32
    L.gm_pigeon_plan_sample.argtypes = [i64, i64, i32, i32, c.c_double, c.POINTER(c.c_uint32), i32, i32, c.POINTER(i32),
                                        c.POINTER(i32), c.POINTER(c.c_double), c.POINTER(i32)]
    L.gm_pigeon_plan_sample.restype = i32
    L.gm_levpart_layout.argtypes = [i32, i32, c.c_double, c.POINTER(i32), c.POINTER(i32), c.POINTER(c.c_double)]
    L.gm_levpart_layout.restype = i32
    L.gm_levpart_info.argtypes = [vp, c.POINTER(i32), c.POINTER(i32), c.POINTER(i64), c.POINTER(i64), c.POINTER(i64)]
    L.gm_levpart_info.restype = i32
    L.gm_pigeon_cascade_info.argtypes = [vp, i32, c.POINTER(i64), c.POINTER(i32), c.POINTER(i64), c.POINTER(i64)]
    L.gm_pigeon_cascade_info.restype = i32
    L.gm_set_variant.argtypes = [vp, i32]
    L.gm_set_param.argtypes = [vp, c.c_char_p, i32]
    L.gm_set_param.restype = i32
    L.gm_last_speculation.argtypes = [vp]
    L.gm_last_speculation.restype = i32
    L.gm_multi_create.argtypes = [vp, i64, i32, i32, c.POINTER(i32), i32, c.POINTER(vp)]
    L.gm_multi_create.restype = i32
    L.gm_multi_knn_query.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.gm_multi_knn_query.restype = i32
    L.gm_multi_nearest_dist.argtypes = [vp, vp, i64, vp]
    L.gm_multi_nearest_dist.restype = i32
    L.gm_multi_devices.argtypes = [vp]
    L.gm_multi_devices.restype = i32
    L.gm_multi_has_comm.argtypes = [vp]
    L.gm_multi_has_comm.restype = i32
    L.gm_multi_index.argtypes = [vp, i32]
    L.gm_multi_index.restype = vp
    L.gm_multi_free.argtypes = [vp]
    L.gm_multi_free.restype = None
    L.gm_timing_reserve.argtypes = [vp, i32]
    L.gm_timing_reserve.restype = i32
    L.gm_timing_read.argtypes = [vp, vp, i32]
    L.gm_timing_read.restype = i32
    L.gm_pack_keys_dev.argtypes = [vp, vp, i64, i32, vp, vp]
    L.gm_pack_keys_dev.restype = i32
    L.gm_unpack_keys_dev.argtypes = [vp, i64, i32, vp, vp, vp, vp]
    L.gm_unpack_keys_dev.restype = i32
    L.gm_pam_count.argtypes = [c.c_char_p, i64, c.c_char_p, i32, i32, i32, i32, c.POINTER(vp), c.POINTER(i64),
                               c.POINTER(i64)]
    L.gm_pam_count.restype = i32
    L.gm_pam_emit.argtypes = [vp, vp, vp, vp]
    L.gm_pam_emit.restype = i32
    L.gm_pam_free.argtypes = [vp]
    L.gm_pam_free.restype = None
    L.gm_site_flags.argtypes = [vp, i64, i32, c.POINTER(c.c_char_p), i32, vp, i32]
    L.gm_site_flags.restype = i32
    L.gm_seed_duplicated.argtypes = [vp, i64, i32, i32, i32, vp, i32]
    L.gm_seed_duplicated.restype = i32
    for name in ("gm_pack", "gm_unpack", "gm_index_create", "gm_index_create_dev", "gm_index_guidelen",
                 "gm_index_metric", "gm_knn_query", "gm_knn_query_dev", "gm_nearest_dist",
                 "gm_nearest_dist_dev", "gm_cfd_batch", "gm_cfd_batch_dev", "gm_set_variant"):
        getattr(L, name).restype = i32
    _lib = L
    return L


# every symbol include/gmhip.h declares (tests check the built library exports them all)
EXPORTS = [
    "gm_version", "gm_last_error", "gm_device_count", "gm_pack", "gm_unpack", "gm_format_doubles", "gm_format_neighbor_rows", "gm_index_create",
    "gm_index_create_dev", "gm_index_free", "gm_index_size", "gm_index_guidelen", "gm_index_metric",
    "gm_knn_query", "gm_knn_query_dev", "gm_nearest_dist", "gm_nearest_dist_dev", "gm_cfd_batch",
    "gm_cfd_batch_dev", "gm_last_kernel_ms", "gm_last_query_ms", "gm_last_kernel_name", "gm_last_kernel_pairs",
    "gm_set_variant", "gm_timing_reserve", "gm_timing_read", "gm_pam_count", "gm_pam_emit", "gm_pam_free",
    "gm_pack_keys_dev", "gm_unpack_keys_dev", "gm_site_flags", "gm_seed_duplicated",
    "gm_last_kernel_evaluated", "gm_last_incomplete", "gm_pigeon_blocks", "gm_pigeon_plan", "gm_pigeon_plan_dist", "gm_pigeon_plan_sample", "gm_levpart_layout", "gm_levpart_info", "gm_pigeon_cascade_info", "gm_pigeon_block_info", "gm_set_param", "gm_last_speculation",
    "gm_multi_create", "gm_multi_knn_query", "gm_multi_nearest_dist", "gm_multi_devices", "gm_multi_has_comm", "gm_multi_index", "gm_multi_free",
]


def _check(rc):
    if rc != 0:
        msg = lib().gm_last_error().decode(errors="replace")
        if rc == -4:
            raise ValueError(msg)
        raise GmHipError("libgmhip error %d: %s" % (rc, msg))


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def pack(seqs, guidelen):
    """list[str] (all of length guidelen, ACGT only) -> uint64 array, 2 bits per base."""
    seqs = list(seqs)
    n = len(seqs)
    out = np.empty((n,), np.uint64)
    if n == 0:
        return out
    joined = "".join(seqs)
    # every row must have the guide length (a mixed-length list whose lengths merely SUM to n * guidelen would
    # otherwise be mis-framed silently)
    if len(joined) != n * guidelen or not (np.fromiter(map(len, seqs), dtype=np.int64, count=n) == guidelen).all():
        raise ValueError("all sequences must have length %d" % guidelen)
    _check(lib().gm_pack(joined.encode("ascii"), n, guidelen, _ptr(out)))
    return out


def pack_bytes(buf, n, guidelen):
    """(n*guidelen) ASCII bytes / uint8 array -> uint64 array."""
    arr = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf)
    assert arr.size == n * guidelen
    out = np.empty((n,), np.uint64)
    if n:
        _check(lib().gm_pack(arr.ctypes.data_as(ctypes.c_char_p), n, guidelen, _ptr(out)))
    return out


def unpack(packed, guidelen):
    packed = np.ascontiguousarray(packed, dtype=np.uint64)
    n = packed.shape[0]
    buf = np.empty((n, guidelen), np.uint8)
    if n:
        _check(lib().gm_unpack(_ptr(packed), n, guidelen, _ptr(buf)))
    if n == 0:
        return []
    # one vectorised bytes -> str conversion instead of a Python-level decode per row
    return np.ascontiguousarray(buf).view("S%d" % guidelen).ravel().astype(str).tolist()


def format_doubles(values):
    """[str(x) for x in values] for a float64 array -- Python's float repr, formatted natively (host code)."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    n = int(v.shape[0])
    if n == 0:
        return []
    buf = np.empty((25 * n + 8,), np.uint8)
    w = lib().gm_format_doubles(_ptr(v), n, _ptr(buf), buf.shape[0])
    if w < 0:
        raise GmHipError("gm_format_doubles failed")
    return buf[:w].tobytes().decode("ascii").split("\n")


def format_neighbor_rows(db_packed, guidelen, ids, dists, counts):
    """(similar guides, similar guide distances): per result row the ';'-joined neighbour sequences and distances."""
    db = np.ascontiguousarray(db_packed, dtype=np.uint64)
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    dists = np.ascontiguousarray(dists, dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n, k = ids.shape
    if n == 0:
        return [], []
    bs = np.empty((n * k * (guidelen + 1),), np.uint8)
    bd = np.empty((n * k * 12,), np.uint8)
    ls, ld = ctypes.c_int64(), ctypes.c_int64()
    _check(lib().gm_format_neighbor_rows(_ptr(db), db.shape[0], guidelen, _ptr(ids), _ptr(dists), _ptr(counts), n, k, _ptr(bs),
                                         bs.shape[0], _ptr(bd), bd.shape[0], ctypes.byref(ls), ctypes.byref(ld)))
    return (bs[:ls.value].tobytes().decode("ascii").split("\n"), bd[:ld.value].tobytes().decode("ascii").split("\n"))


class Index:
    """Exact neighbour-search index resident in one GPU's HBM (replaces the nmslib index object)."""

    def __init__(self, packed, guidelen, metric, device=0, _dev_ptr=None, _n=None, _stream=None):
        self._h = ctypes.c_void_p()
        self.guidelen = guidelen
        self.metric = metric
        self.device = device
        if _dev_ptr is not None:
            _check(lib().gm_index_create_dev(ctypes.c_void_p(_dev_ptr), _n, guidelen, metric, device,
                                             ctypes.c_void_p(_stream or 0), ctypes.byref(self._h)))
            self.n = _n
        else:
            packed = np.ascontiguousarray(packed, dtype=np.uint64)
            self.n = int(packed.shape[0])
            _check(lib().gm_index_create(_ptr(packed), self.n, guidelen, metric, device, ctypes.byref(self._h)))

    @classmethod
    def from_device(cls, dev_ptr, n, guidelen, metric, device=0, stream=None):
        return cls(None, guidelen, metric, device, _dev_ptr=dev_ptr, _n=n, _stream=stream)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().gm_index_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.n

    def knn(self, q_packed, k):
        """-> (ids [nq,k] int32, dists [nq,k] int32 true units, counts [nq] int32); -1 padded."""
        q = np.ascontiguousarray(q_packed, dtype=np.uint64)
        nq = int(q.shape[0])
        ids = np.full((nq, k), -1, np.int32)
        dists = np.full((nq, k), -1, np.int32)
        counts = np.zeros((nq,), np.int32)
        if nq:
            _check(lib().gm_knn_query(self._h, _ptr(q), nq, k, _ptr(ids), _ptr(dists), _ptr(counts)))
        return ids, dists, counts

    def knn_dev(self, q_ptr, nq, k, ids_ptr, dists_ptr, counts_ptr, stream=None):
        _check(lib().gm_knn_query_dev(self._h, ctypes.c_void_p(q_ptr), nq, k, ctypes.c_void_p(ids_ptr),
                                      ctypes.c_void_p(dists_ptr), ctypes.c_void_p(counts_ptr),
                                      ctypes.c_void_p(stream or 0)))

    def nearest(self, q_packed):
        q = np.ascontiguousarray(q_packed, dtype=np.uint64)
        nq = int(q.shape[0])
        d1 = np.full((nq,), -1, np.int32)
        if nq:
            _check(lib().gm_nearest_dist(self._h, _ptr(q), nq, _ptr(d1)))
        return d1

    def last_speculation(self):
        """0 = the last K1P query waited for the device's row counts, 1 = it ran on the previous query's (validated at the
        end), 2 = it did and had to run again (gm_last_speculation)."""
        return int(lib().gm_last_speculation(self._h))

    def nearest_dev(self, q_ptr, nq, d1_ptr, stream=None):
        _check(lib().gm_nearest_dist_dev(self._h, ctypes.c_void_p(q_ptr), nq, ctypes.c_void_p(d1_ptr),
                                         ctypes.c_void_p(stream or 0)))

    def last_query_ms(self):
        """Device time (ms) of every kernel the last knn / nearest call enqueued (HIP events)."""
        return lib().gm_last_query_ms(self._h)

    def set_variant(self, v):
        _check(lib().gm_set_variant(self._h, v))

    def set_param(self, name, value):
        """Named tuning knob (include/gmhip.h, gm_set_param); results do not depend on it."""
        _check(lib().gm_set_param(self._h, name.encode("ascii"), int(value)))

    def timing_reserve(self, n):
        _check(lib().gm_timing_reserve(self._h, n))

    def timing_read(self, max_n=4096):
        """Durations (ms) of the scan-kernel launches recorded since timing_reserve (HIP events)."""
        buf = np.zeros((max_n,), np.float64)
        n = lib().gm_timing_read(self._h, _ptr(buf), max_n)
        if n < 0:
            _check(n)
        return buf[:n].copy()

    def last_kernel(self):
        """(name, ms, pairs) of the dominant scan kernel of the last query (HIP events)."""
        L = lib()
        return (L.gm_last_kernel_name(self._h).decode(), L.gm_last_kernel_ms(self._h),
                L.gm_last_kernel_pairs(self._h))

    def last_evaluated(self):
        """(pairs the dominant launch really evaluated, candidates re-scanned afterwards): differs from
        last_kernel()[2] only for the pigeonhole-partitioned scan."""
        L = lib()
        return int(L.gm_last_kernel_evaluated(self._h)), int(L.gm_last_incomplete(self._h))

    def pigeon_block_info(self):
        """[(positions evaluated per pair, pairs evaluated in the last launch)] per block of the K1P layout in use."""
        out = []
        for b in range(8):
            pos, pairs = ctypes.c_int(), ctypes.c_int64()
            if not lib().gm_pigeon_block_info(self._h, b, ctypes.byref(pos), ctypes.byref(pairs)):
                break
            out.append((int(pos.value), int(pairs.value)))
        return out

    def cascade_info(self):
        """K1P levels of the last query: list of dict(n_q, radius, pairs, left) -- level 0 first; [] = plain scan."""
        out = []
        n = int(lib().gm_pigeon_cascade_info(self._h, -1, None, None, None, None))
        for lv in range(n):
            nq, rad, pairs, left = ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int64(0)
            lib().gm_pigeon_cascade_info(self._h, lv, ctypes.byref(nq), ctypes.byref(rad), ctypes.byref(pairs), ctypes.byref(left))
            out.append(dict(n_q=int(nq.value), radius=int(rad.value), pairs=int(pairs.value), left=int(left.value)))
        return out

    def levpart_info(self):
        """K2P levels of the last Levenshtein query: dict(r1, r2, rows2, pairs2, left2) -- r1 = 0: the plain K2B scan ran."""
        r1, r2 = ctypes.c_int32(0), ctypes.c_int32(0)
        rows2, pairs2, left2 = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        lib().gm_levpart_info(self._h, ctypes.byref(r1), ctypes.byref(r2), ctypes.byref(rows2), ctypes.byref(pairs2), ctypes.byref(left2))
        return dict(r1=int(r1.value), r2=int(r2.value), rows2=int(rows2.value), pairs2=int(pairs2.value), left2=int(left2.value))

    def pigeon_blocks(self):
        """Blocks of the pigeonhole layout built for this index (0: the plain scan is in use)."""
        return int(lib().gm_pigeon_blocks(self._h))


class MultiIndex:
    """The pool replicated on several GPUs of one node, candidates sharded across them, one RCCL all-gather per query
    (gm_multi_* of include/gmhip.h): same results as Index.knn on one device."""

    def __init__(self, packed, guidelen, metric, devices):
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        devs = (ctypes.c_int * len(devices))(*devices)
        self._h = ctypes.c_void_p()
        self.n = int(packed.shape[0])
        L = lib()
        # one RCCL per process, and the one built for the HIP runtime in use: with PyTorch's runtime shared
        # (_share_torch_hip_runtime) that is PyTorch's librccl.so -- load it before gm_multi_create looks for a copy
        if _torch_lib_dir and os.path.exists(os.path.join(_torch_lib_dir, "librccl.so")):
            try:
                ctypes.CDLL(os.path.join(_torch_lib_dir, "librccl.so"), mode=ctypes.RTLD_LOCAL)
            except OSError:
                pass
        _check(L.gm_multi_create(_ptr(packed), self.n, guidelen, metric, devs, len(devices), ctypes.byref(self._h)))

    def devices(self):
        return int(lib().gm_multi_devices(self._h))

    def has_comm(self):
        return bool(lib().gm_multi_has_comm(self._h))

    def knn(self, q_packed, k):
        q = np.ascontiguousarray(q_packed, dtype=np.uint64)
        nq = int(q.shape[0])
        ids = np.full((nq, k), -1, np.int32)
        dists = np.full((nq, k), -1, np.int32)
        counts = np.zeros((nq,), np.int32)
        if nq:
            _check(lib().gm_multi_knn_query(self._h, _ptr(q), nq, k, _ptr(ids), _ptr(dists), _ptr(counts)))
        return ids, dists, counts

    def set_param(self, name, value):
        """gm_set_param on every device's handle (experiments, tests: e.g. a forced K1P layout)."""
        for i in range(self.devices()):
            _check(lib().gm_set_param(ctypes.c_void_p(lib().gm_multi_index(self._h, i)), name.encode(), int(value)))

    def nearest(self, q_packed):
        """Nearest-target distance of every candidate (get_control_seqs' query), sharded over the devices."""
        q = np.ascontiguousarray(q_packed, dtype=np.uint64)
        nq = int(q.shape[0])
        d1 = np.full((nq,), -1, np.int32)
        if nq:
            _check(lib().gm_multi_nearest_dist(self._h, _ptr(q), nq, _ptr(d1)))
        return d1

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().gm_multi_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cfd_batch(guides_packed, offs_packed, guidelen, mm, device=0):
    g = np.ascontiguousarray(guides_packed, dtype=np.uint64)
    o = np.ascontiguousarray(offs_packed, dtype=np.uint64)
    assert g.shape == o.shape
    table = np.ascontiguousarray(np.asarray(mm, dtype=np.float64).reshape(-1))
    assert table.size == 320
    out = np.empty(g.shape, np.float64)
    if g.size:
        _check(lib().gm_cfd_batch(_ptr(g), _ptr(o), g.size, guidelen, _ptr(table), _ptr(out), device))
    return out


def pam_scan(seq, pam, five_prime, guidelen, device=0):
    """GPU PAM scan of one record (upper-case str or bytes).  Returns (packed uint64, tstart int64,
    mstart int64, n_fwd): forward hits first (ascending), then reverse hits (ascending)."""
    buf = seq.encode("ascii", "replace") if isinstance(seq, str) else bytes(seq)
    h = ctypes.c_void_p()
    nf, nr = ctypes.c_int64(), ctypes.c_int64()
    try:
        # (inside the try: gm_pam_count may fail after it has handed out the scan handle)
        _check(lib().gm_pam_count(buf, len(buf), pam.encode("ascii"), len(pam), 1 if five_prime else 0, guidelen,
                                  device, ctypes.byref(h), ctypes.byref(nf), ctypes.byref(nr)))
        n = nf.value + nr.value
        packed = np.empty((n,), np.uint64)
        tstart = np.empty((n,), np.int64)
        mstart = np.empty((n,), np.int64)
        if n:
            _check(lib().gm_pam_emit(h, _ptr(packed), _ptr(tstart), _ptr(mstart)))
    finally:
        if h.value:
            lib().gm_pam_free(h)
    return packed, tstart, mstart, int(nf.value)


def pigeon_plan_dist(n_db, n_q, guidelen, k, p_short, p_match=0.25, levels=3):
    """The planner with a measured distribution p_short[r] (share of the candidates whose k-th neighbour lies beyond
    r): -> dict(blocks=[(len, radius), ...], cost, cascade).  Host arithmetic, no GPU needed."""
    import numpy as _np
    ps = _np.ascontiguousarray(p_short, dtype=_np.float64)
    if ps.shape[0] < guidelen + 1:
        raise ValueError("p_short needs guidelen + 1 entries")
    blen = (ctypes.c_int32 * 8)()
    trad = (ctypes.c_int32 * 8)()
    cost, casc = ctypes.c_double(0.0), ctypes.c_int32(0)
    B = int(lib().gm_pigeon_plan_dist(int(n_db), int(n_q), int(guidelen), int(k), float(p_match),
                                      ps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(levels), blen, trad,
                                      ctypes.byref(cost), ctypes.byref(casc)))
    return dict(blocks=[(int(blen[b]), int(trad[b])) for b in range(B)], cost=float(cost.value), cascade=int(casc.value))


def pigeon_plan_sample(n_db, n_q, guidelen, k, cum, p_match=0.25, levels=3):
    """The planner on a sample of the library's own kind: cum[i, d] = pool targets within Hamming distance d = 0 .. 8 of
    sampled candidate i.  -> dict(blocks=[(len, radius), ...], cost, cascade).  Host arithmetic, no GPU needed."""
    import numpy as _np
    cm = _np.ascontiguousarray(cum, dtype=_np.uint32)
    if cm.ndim != 2 or cm.shape[1] != 9:
        raise ValueError("cum must be [rows, 9]")
    blen = (ctypes.c_int32 * 8)()
    trad = (ctypes.c_int32 * 8)()
    cost, casc = ctypes.c_double(0.0), ctypes.c_int32(0)
    B = int(lib().gm_pigeon_plan_sample(int(n_db), int(n_q), int(guidelen), int(k), float(p_match),
                                        cm.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), int(cm.shape[0]), int(levels),
                                        blen, trad, ctypes.byref(cost), ctypes.byref(casc)))
    return dict(blocks=[(int(blen[b]), int(trad[b])) for b in range(B)], cost=float(cost.value), cascade=int(casc.value))


def levpart_layout(guidelen, r, p_match=0.25):
    """K2P's layout for radius r: dict(blocks=[lengths], copies, fraction) -- or None.  Host arithmetic, no GPU needed."""
    blen = (ctypes.c_int32 * 8)()
    n_copies, frac = ctypes.c_int32(0), ctypes.c_double(0.0)
    B = int(lib().gm_levpart_layout(int(guidelen), int(r), float(p_match), blen, ctypes.byref(n_copies), ctypes.byref(frac)))
    if B == 0:
        return None
    return dict(blocks=[int(blen[j]) for j in range(B)], copies=int(n_copies.value), fraction=float(frac.value))


def pigeon_plan(n_db, guidelen, k, p_match=0.25):
    """Blocks the pigeonhole-partitioned scan would use (0: plain scan).  Host arithmetic, no GPU needed."""
    return int(lib().gm_pigeon_plan(int(n_db), int(guidelen), int(k), float(p_match)))


def site_flags(packed, guidelen, sites, device=0):
    """bool[n]: target i contains one of the IUPAC `sites` (list the reverse complements yourself)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint64)
    flags = np.zeros(packed.shape, np.uint8)
    arr = (ctypes.c_char_p * max(1, len(sites)))(*[s.encode("ascii") for s in sites])
    _check(lib().gm_site_flags(_ptr(packed), packed.shape[0], guidelen, arr, len(sites), _ptr(flags), device))
    return flags.astype(bool)


def seed_duplicated(packed, guidelen, start, count, device=0):
    """bool[n]: an earlier row has the same `count` bases from position `start` (duplicated(keep='first'))."""
    packed = np.ascontiguousarray(packed, dtype=np.uint64)
    flags = np.zeros(packed.shape, np.uint8)
    _check(lib().gm_seed_duplicated(_ptr(packed), packed.shape[0], guidelen, start, count, _ptr(flags), device))
    return flags.astype(bool)


def pack_keys_dev(ids_ptr, dists_ptr, n_q, k, keys_ptr, stream=None):
    _check(lib().gm_pack_keys_dev(ctypes.c_void_p(ids_ptr), ctypes.c_void_p(dists_ptr), n_q, k,
                                  ctypes.c_void_p(keys_ptr), ctypes.c_void_p(stream or 0)))


def unpack_keys_dev(keys_ptr, n_q, k, ids_ptr, dists_ptr, counts_ptr, stream=None):
    _check(lib().gm_unpack_keys_dev(ctypes.c_void_p(keys_ptr), n_q, k, ctypes.c_void_p(ids_ptr),
                                    ctypes.c_void_p(dists_ptr), ctypes.c_void_p(counts_ptr),
                                    ctypes.c_void_p(stream or 0)))


def device_count():
    """GPUs visible to libgmhip; 0 when the library has not been built (the host-side numpy scanner and pandas filters
    of core.py stay usable without it -- every SEARCH entry point still fails loudly)."""
    try:
        return lib().gm_device_count()
    except (ImportError, OSError):
        return 0
