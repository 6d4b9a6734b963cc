"""ctypes binding of libasr_hip.so (include/asr_hip.h) and a thin `Engine` object.

There is NO CPU fallback: if the library is missing or no HIP device is
available every entry point raises (AsrLibraryError / AsrError).  The oracle
under oracle/ is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libasr_hip.so")

ASR_OK = 0
ASR_ERR_INVALID, ASR_ERR_HIP, ASR_ERR_STATE, ASR_ERR_COMM, ASR_ERR_NOMEM = 1, 2, 3, 4, 5      # include/asr_hip.h
IN_F32_PREPARED, IN_F32_RAW, IN_U8_RAW = 0, 1, 2
OUT_LATENT, OUT_FEATURES = 0, 1

#: every symbol include/asr_hip.h declares (tests check the .so exports them all)
EXPORTS = [
    "asr_create", "asr_destroy", "asr_last_error", "asr_version", "asr_sync", "asr_set_input_size",
    "asr_param_count", "asr_param_size", "asr_set_params", "asr_get_params", "asr_set_cca",
    "asr_embed_view1", "asr_embed_view2", "asr_embed_view1_dev", "asr_embed_view2_dev", "asr_embed_both",
    "asr_rank", "asr_rank_dev", "asr_topk", "asr_topk_dev", "asr_cca_fit", "asr_cca_fit_dev",
    "asr_db_create", "asr_db_refresh", "asr_db_destroy", "asr_db_size", "asr_topk_db_dev", "asr_rank_db_dev",
    "asr_topk_rank_db_dev", "asr_rank_dstar_db_dev", "asr_topk_count_db_dev", "asr_topk_merge_dev", "asr_rank_finish_dev",
    "asr_dev_alloc", "asr_dev_free", "asr_dev_upload", "asr_dev_download",
    "asr_host_alloc", "asr_host_free", "asr_eval_batches",
    "asr_profile_enable", "asr_profile_filter", "asr_profile_reset", "asr_profile_count", "asr_profile_get", "asr_profile_symbol",
    "asr_debug_activation",
    "asr_train_begin", "asr_train_end", "asr_train_set_global_batch", "asr_train_step", "asr_train_step_dev", "asr_valid_loss", "asr_set_objective", "asr_burn_in",
    "asr_compute_gradients",
    "asr_comm_unique_id", "asr_comm_init", "asr_comm_init_custom", "asr_comm_destroy", "asr_comm_info", "asr_comm_stats", "asr_comm_timing", "asr_comm_library",
    "asr_comm_allreduce_dev", "asr_comm_allgather_dev",
    "asr_rank_sharded_dev", "asr_slice_windows_dev", "asr_piece_vote_dev", "asr_gather_windows_dev", "asr_dtw_dev", "asr_spectrogram_dev", "asr_debug_tune_report",
    "asr_opt_state_size", "asr_get_opt_state", "asr_set_opt_state", "asr_debug_train_tensor", "asr_cca_train_debug",
]


#: host-callback transport of asr_comm_init_custom (include/asr_hip.h)
ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_int64, c_int)
ALLGATHER_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_int64)
COMM_ID_BYTES = 128
DTYPE_F32, DTYPE_F64, DTYPE_I32 = 0, 1, 2


class AsrLibraryError(ImportError):
    """libasr_hip.so is missing or cannot be loaded."""


class AsrError(RuntimeError):
    """An asr_* call returned a non-zero status."""

    def __init__(self, code, message):
        super().__init__("asr error %d: %s" % (code, message))
        self.code = code


class AsrConfig(ctypes.Structure):
    _fields_ = [
        ("struct_size", c_int32), ("device", c_int32), ("num_filters", c_int32), ("resize_view1", c_int32),
        ("h1", c_int32), ("w1", c_int32), ("h2", c_int32), ("w2", c_int32),
        ("dim_latent", c_int32), ("max_chunk", c_int32),
        ("r1", c_float), ("r2", c_float), ("rT", c_float), ("alpha", c_float), ("gamma", c_float), ("l2", c_float),
        ("pool_ties", c_int32),
    ]


# asr_config.pool_ties: which elements of a 2x2 pooling window with several equal maxima receive the gradient
POOL_TIES = {"all": 0, "first": 1}


_lib = None


def load_library(path=None):
    """dlopen libasr_hip.so and declare the prototypes; raises AsrLibraryError."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # multi-process GPU work on this platform (RCCL at world > 1, IPC handles) needs dmabuf IPC: the variable has to be
    # in place before the HSA runtime initialises, i.e. before the first HIP call of the process - this is that point.
    # An explicit setting of the caller wins.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not os.path.exists(p):
        raise AsrLibraryError(
            "%s not found - build it with `python -m audio_sheet_retrieval_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback." % p)
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise AsrLibraryError("cannot load %s: %s" % (p, e))
    fpp = POINTER(POINTER(c_float))
    i64p = POINTER(c_int64)
    proto = {
        "asr_create": (c_int, [POINTER(AsrConfig), POINTER(c_void_p)]),
        "asr_destroy": (None, [c_void_p]),
        "asr_last_error": (c_char_p, [c_void_p]),
        "asr_version": (c_char_p, []),
        "asr_sync": (c_int, [c_void_p]),
        "asr_set_input_size": (c_int, [c_void_p, c_int, c_int, c_int]),
        "asr_param_count": (c_int, [c_void_p]),
        "asr_param_size": (c_int, [c_void_p, c_int, i64p]),
        "asr_set_params": (c_int, [c_void_p, fpp, i64p, c_int]),
        "asr_get_params": (c_int, [c_void_p, fpp, i64p, c_int]),
        "asr_set_cca": (c_int, [c_void_p] + [c_void_p] * 4),
        "asr_embed_view1": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
        "asr_embed_view2": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
        "asr_embed_view1_dev": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
        "asr_embed_view2_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
        "asr_rank": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int,
                             c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
        "asr_rank_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int,
                                 c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
        "asr_topk": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                             c_int64, c_void_p, c_void_p]),
        "asr_topk_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                 c_int64, c_void_p, c_void_p]),
        "asr_db_create": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, POINTER(c_void_p)]),
        "asr_db_refresh": (c_int, [c_void_p, c_void_p]),
        "asr_db_destroy": (c_int, [c_void_p, c_void_p]),
        "asr_db_size": (c_int, [c_void_p, c_void_p, i64p, POINTER(c_int)]),
        "asr_topk_db_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int64, c_void_p, c_void_p]),
        "asr_rank_db_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                    c_void_p]),
        "asr_topk_rank_db_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int64, c_void_p, c_void_p,
                                         c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
        "asr_rank_dstar_db_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                          c_void_p, c_void_p]),
        "asr_topk_count_db_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int64, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p]),
        "asr_topk_merge_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int, c_void_p,
                                       c_void_p]),
        "asr_rank_finish_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
        "asr_cca_fit": (c_int, [c_void_p, c_void_p, c_void_p, c_int64] + [c_void_p] * 5),
        "asr_cca_fit_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64] + [c_void_p] * 4),
        "asr_host_alloc": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
        "asr_host_free": (c_int, [c_void_p, c_void_p]),
        "asr_eval_batches": (c_int, [c_void_p, POINTER(c_void_p), c_int, POINTER(c_void_p), c_int, c_int64,
                                     POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                     POINTER(c_void_p)]),
        "asr_dev_alloc": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
        "asr_dev_free": (c_int, [c_void_p, c_void_p]),
        "asr_dev_upload": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
        "asr_dev_download": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
        "asr_profile_enable": (c_int, [c_void_p, c_int]),
        "asr_profile_filter": (c_int, [c_void_p, c_char_p]),
        "asr_profile_reset": (c_int, [c_void_p]),
        "asr_profile_count": (c_int, [c_void_p]),
        "asr_profile_get": (c_int, [c_void_p, c_int, c_char_p, c_int, i64p, POINTER(c_double),
                                    POINTER(c_double), POINTER(c_double)]),
        "asr_profile_symbol": (c_int, [c_void_p, c_int, c_char_p, c_int]),
        "asr_debug_activation": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p,
                                         POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
        "asr_train_begin": (c_int, [c_void_p, c_int]),
        "asr_train_end": (c_int, [c_void_p]),
        "asr_train_set_global_batch": (c_int, [c_void_p, c_int64]),
        "asr_train_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, POINTER(c_float), c_void_p]),
        "asr_train_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, POINTER(c_float), c_void_p]),
        "asr_valid_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, POINTER(c_float)]),
        "asr_set_objective": (c_int, [c_void_p, c_float, c_float, c_int]),
        "asr_burn_in": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
        "asr_compute_gradients": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, POINTER(c_float)]),
        "asr_slice_windows_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_int,
                                          c_void_p]),
        "asr_piece_vote_dev": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int, c_void_p, c_void_p,
                                       POINTER(c_int32)]),
        "asr_gather_windows_dev": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_void_p]),
        "asr_dtw_dev": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                POINTER(c_int32), POINTER(c_double)]),
        "asr_spectrogram_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_float, c_float, c_int64, c_int, c_void_p]),
        "asr_debug_tune_report": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_float)]),
        "asr_comm_unique_id": (c_int, [c_void_p]),
        "asr_comm_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
        "asr_comm_init_custom": (c_int, [c_void_p, c_int, c_int, ALLREDUCE_FN, ALLGATHER_FN, c_void_p]),
        "asr_comm_destroy": (c_int, [c_void_p]),
        "asr_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
        "asr_comm_stats": (c_int, [c_void_p, i64p, c_int]),
        "asr_comm_timing": (c_int, [c_void_p, c_int, POINTER(ctypes.c_double), i64p]),
        "asr_comm_library": (c_int, [c_void_p, c_char_p, c_int]),
        "asr_comm_allreduce_dev": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
        "asr_comm_allgather_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
        "asr_rank_sharded_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
        "asr_embed_both": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
        "asr_opt_state_size": (c_int, [c_void_p, i64p]),
        "asr_get_opt_state": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, POINTER(c_int32)]),
        "asr_set_opt_state": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32]),
        "asr_debug_train_tensor": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_int64, i64p]),
        "asr_cca_train_debug": (c_int, [c_void_p, c_void_p, c_void_p, c_int64] + [c_void_p] * 7),
    }
    for name, (res, args) in proto.items():
        fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None and os.environ.get("ASR_ALLOW_STALE_LIB") != "1":
        # a stale binary next to newer sources would silently run old kernels: asr_version() carries the hash
        from . import build as _build
        built = (lib.asr_version() or b"").decode()
        want = _build.source_hash()
        if "src:" + want not in built:
            raise AsrLibraryError("%s was built from other sources (%s, sources now %s): re-run "
                                  "`python -m audio_sheet_retrieval_amd.build`" % (p, built, want))
    if path is None:
        _lib = lib
    return lib


#: reference model modules -> library configuration (models/mutopia_ccal_cont*.py)
MODEL_CONFIGS = {
    "mutopia_ccal_cont": dict(num_filters=12, resize_view1=0),
    "mutopia_ccal_cont_rsz": dict(num_filters=24, resize_view1=1),
}


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class DeviceBuffer(object):
    """A library-owned device allocation (plain pointer + size)."""

    def __init__(self, engine, nbytes):
        self.engine, self.nbytes = engine, int(nbytes)
        p = c_void_p()
        engine._check(engine.lib.asr_dev_alloc(engine.ctx, self.nbytes, byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.engine._check(self.engine.lib.asr_dev_upload(self.engine.ctx, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.engine._check(self.engine.lib.asr_dev_download(self.engine.ctx, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def offset(self, nbytes):
        return self.ptr + int(nbytes)

    def free(self):
        if self.ptr:
            self.engine._check(self.engine.lib.asr_dev_free(self.engine.ctx, self.ptr))
            self.ptr = None


class CodeDB(object):
    """asr_db handle: a resident candidate pool.  The rows stay in the caller's device buffer (keep it alive);
    refresh() after changing them in place."""

    def __init__(self, engine, codes_ptr, n, dim=32, ld=32):
        self.engine, self.n, self.dim = engine, int(n), int(dim)
        h = c_void_p()
        engine._check(engine.lib.asr_db_create(engine.ctx, codes_ptr, self.n, int(ld), self.dim, byref(h)))
        self.handle = h
        # the context owns the handle's device buffers (norms, reciprocal norms, unit rows: ~136 B per row): the engine
        # destroys the data bases that are still open when IT closes, whatever order the caller drops things in.  The
        # registry is a WeakSet: a handle the caller simply drops is collected (and __del__ frees its buffers) instead
        # of living until Engine.close() - a server that builds a data base per request must not pile them up
        engine._open_dbs.add(self)

    def refresh(self):
        self.engine._check(self.engine.lib.asr_db_refresh(self.engine.ctx, self.handle))

    def topk_dev(self, q_ptr, n_q, k, idx_ptr, dist_ptr, ld=32, idx_offset=0):
        self.engine._check(self.engine.lib.asr_topk_db_dev(self.engine.ctx, self.handle, q_ptr, n_q, ld, k, idx_offset,
                                                           idx_ptr, dist_ptr))

    def rank_dev(self, lv1_ptr, n1, ranks_ptr, dstar_ptr, ties_ptr, ld=32, query_offset=0, n1_global=None):
        self.engine._check(self.engine.lib.asr_rank_db_dev(self.engine.ctx, self.handle, lv1_ptr, n1, ld, query_offset,
                                                           n1 if n1_global is None else n1_global, ranks_ptr, dstar_ptr,
                                                           ties_ptr))

    def topk_rank_dev(self, q_ptr, n_q, k, idx_ptr, dist_ptr, ranks_ptr, dstar_ptr, ties_ptr, ld=32, idx_offset=0,
                      query_offset=0, n1_global=None):
        self.engine._check(self.engine.lib.asr_topk_rank_db_dev(
            self.engine.ctx, self.handle, q_ptr, n_q, ld, k, idx_offset, idx_ptr, dist_ptr, query_offset,
            n_q if n1_global is None else n1_global, ranks_ptr, dstar_ptr, ties_ptr))

    # -- this data base as one shard of a larger pool (query-sharded retrieval) --
    def rank_dstar_dev(self, q_ptr, n_q, item_offset, n2_global, query_offset, n1_global, dstar_ptr, jstar_ptr, ld=32):
        self.engine._check(self.engine.lib.asr_rank_dstar_db_dev(self.engine.ctx, self.handle, q_ptr, n_q, ld, item_offset,
                                                                 n2_global, query_offset, n1_global, dstar_ptr, jstar_ptr))

    def topk_count_dev(self, q_ptr, n_q, k, item_offset, idx_ptr, dist_ptr, dstar_ptr, jstar_ptr, counts_ptr, ld=32):
        self.engine._check(self.engine.lib.asr_topk_count_db_dev(self.engine.ctx, self.handle, q_ptr, n_q, ld, k, item_offset,
                                                                 idx_ptr, dist_ptr, dstar_ptr, jstar_ptr, counts_ptr))

    def topk(self, queries, k, idx_offset=0):
        """host queries (Q, dim) -> (idx (Q,k) int32, dist (Q,k) float64), like Engine.topk against the pool"""
        q = _f32c(queries)
        eng = self.engine
        dq = eng.alloc(max(q.nbytes, 4)).upload(q)
        di, dd = eng.alloc(max(q.shape[0] * k * 4, 4)), eng.alloc(max(q.shape[0] * k * 8, 8))
        try:
            self.topk_dev(dq.ptr, q.shape[0], k, di.ptr, dd.ptr, ld=q.shape[1], idx_offset=idx_offset)
            return di.download((q.shape[0], k), np.int32), dd.download((q.shape[0], k), np.float64)
        finally:
            for b in (dq, di, dd):
                b.free()

    def close(self):
        if getattr(self, "handle", None) is not None and getattr(self.engine, "ctx", None):
            self.engine.lib.asr_db_destroy(self.engine.ctx, self.handle)
        self.handle = None
        try:
            self.engine._open_dbs.discard(self)
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine(object):
    """One asr_ctx: network state on one GPU.  Mirrors what build_model() +
    theano.function(...) give the reference (models/mutopia_ccal_cont.py:61-149,
    run_eval.py:92-95)."""

    def __init__(self, model_name="mutopia_ccal_cont", device=0, h1=160, w1=200, h2=92, w2=42,
                 max_chunk=0, r1=1e-3, r2=1e-3, rT=1e-3, alpha=1.0, gamma=0.7, l2=1e-5, lib=None, pool_ties=None):
        if model_name not in MODEL_CONFIGS:
            raise ValueError("unknown model %r (have %s)" % (model_name, sorted(MODEL_CONFIGS)))
        self.lib = lib or load_library()
        self._open_dbs = weakref.WeakSet()
        mc = MODEL_CONFIGS[model_name]
        self.model_name = model_name
        # pool_ties: "all" (default; Theano's CPU MaxPoolGrad - every element equal to the window maximum receives the
        # gradient) or "first"; ASR_POOL_TIES in the environment sets the default of a process
        if pool_ties is None:
            pool_ties = os.environ.get("ASR_POOL_TIES", "all")
        if pool_ties not in POOL_TIES:
            raise ValueError("pool_ties must be one of %s, got %r" % (sorted(POOL_TIES), pool_ties))
        self.pool_ties = pool_ties
        cfg = AsrConfig(ctypes.sizeof(AsrConfig), device, mc["num_filters"], mc["resize_view1"],
                        h1, w1, h2, w2, 32, max_chunk, r1, r2, rT, alpha, gamma, l2, POOL_TIES[pool_ties])
        self.cfg = cfg
        ctx = c_void_p()
        rc = self.lib.asr_create(byref(cfg), byref(ctx))
        if rc != ASR_OK:
            raise AsrError(rc, (self.lib.asr_last_error(None) or b"").decode())
        self.ctx = ctx
        self.net_h1 = h1 // 2 if mc["resize_view1"] else h1
        self.net_w1 = w1 // 2 if mc["resize_view1"] else w1

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc):
        if rc != ASR_OK:
            raise AsrError(rc, (self.lib.asr_last_error(self.ctx) or b"").decode())

    def close(self):
        if getattr(self, "ctx", None):
            for db in list(getattr(self, "_open_dbs", ())):     # before asr_destroy: afterwards their buffers could not be freed
                db.close()
            for p in list(getattr(self, "_pinned", {}).values()):
                self.lib.asr_host_free(self.ctx, p)
            self._pinned = {}
            self.lib.asr_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self.lib.asr_sync(self.ctx))

    # -- piece identification (audio_sheet_server.py:213-300) ----------------------
    def slice_windows_dev(self, src_ptr, rows, T, r0, win_h, win_w, starts, out_ptr):
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        self._check(self.lib.asr_slice_windows_dev(self.ctx, src_ptr, rows, T, r0, win_h, win_w, starts.ctypes.data,
                                                   starts.size, out_ptr))

    def piece_vote_dev(self, idx_ptr, n_idx, ids_ptr, n_db, n_pieces, top_k):
        pieces, counts = np.empty(top_k, np.int32), np.empty(top_k, np.int32)
        m = c_int32()
        self._check(self.lib.asr_piece_vote_dev(self.ctx, idx_ptr, n_idx, ids_ptr, n_db, n_pieces, top_k,
                                                pieces.ctypes.data, counts.ctypes.data, byref(m)))
        return pieces[:m.value], counts[:m.value]

    def dtw(self, a, b, want_dists=True):
        """cosine distance matrix + DTW of two code sequences, rows = a (utils/dtw_by_dist.py:5-34 on
        cdist(a, b, "cosine")) -> (min_dist, dists (n_a,n_b) float64 or None, path_a, path_b)."""
        a, b = _f32c(a), _f32c(b)
        if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1]:
            raise ValueError("dtw expects (n_a,d) and (n_b,d) arrays")
        n_a, n_b, dim = a.shape[0], b.shape[0], a.shape[1]
        da, db = self.alloc(a.nbytes).upload(a), self.alloc(b.nbytes).upload(b)
        dists = np.empty((n_a, n_b), np.float64) if want_dists else None
        pa, pb = np.empty(n_a + n_b, np.int32), np.empty(n_a + n_b, np.int32)
        ln, md = c_int32(), c_double()
        try:
            self._check(self.lib.asr_dtw_dev(self.ctx, da.ptr, n_a, db.ptr, n_b, dim,
                                             dists.ctypes.data if want_dists else None, pa.ctypes.data, pb.ctypes.data,
                                             byref(ln), byref(md)))
        finally:
            da.free()
            db.free()
        return float(md.value), dists, pa[:ln.value].copy(), pb[:ln.value].copy()

    def spectrogram_dev(self, samples_ptr, n_samples, frame_size, hop, window, fb_start, fb_len, fb_weights, n_frames,
                        out_ptr, mul=1.0, add=1.0, transposed=True):
        window = np.ascontiguousarray(window, np.float32)
        fb_start = np.ascontiguousarray(fb_start, np.int32)
        fb_len = np.ascontiguousarray(fb_len, np.int32)
        fb_weights = np.ascontiguousarray(fb_weights, np.float32)
        self._check(self.lib.asr_spectrogram_dev(self.ctx, samples_ptr, n_samples, frame_size, hop, window.ctypes.data,
                                                 fb_start.ctypes.data, fb_len.ctypes.data, fb_weights.ctypes.data,
                                                 fb_start.size, mul, add, n_frames, 1 if transposed else 0, out_ptr))

    def tune_report(self):
        """(comparisons, mismatches, max deviation) of the autotuner's self-check (ASR_TUNE_VERIFY=1)."""
        a, b, d = c_int32(), c_int32(), c_float()
        self._check(self.lib.asr_debug_tune_report(self.ctx, byref(a), byref(b), byref(d)))
        return int(a.value), int(b.value), float(d.value)

    def gather_windows_dev(self, src_ptr, src_floats, desc, out_h, out_w, out_ptr):
        desc = np.ascontiguousarray(desc, dtype=np.float64).reshape(-1, 9)
        self._check(self.lib.asr_gather_windows_dev(self.ctx, src_ptr, src_floats, desc.ctypes.data, desc.shape[0],
                                                    out_h, out_w, out_ptr))

    # -- multi-GPU (one context per GPU) ----------------------------
    def comm_unique_id(self):
        """rank 0: the RCCL unique id (128 bytes) every rank passes to comm_init."""
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        rc = self.lib.asr_comm_unique_id(buf)
        if rc != ASR_OK:
            raise AsrError(rc, (self.lib.asr_last_error(None) or b"").decode())
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        assert len(unique_id) == COMM_ID_BYTES
        self._check(self.lib.asr_comm_init(self.ctx, rank, world, ctypes.c_char_p(unique_id)))

    def comm_init_custom(self, rank, world, allreduce, allgather):
        """allreduce(buf_dev, count, dtype) / allgather(send_dev, recv_dev, bytes_per_rank): host-synchronous
        Python callables returning 0; the library drains its stream before calling them."""
        self._comm_cbs = (ALLREDUCE_FN(lambda user, buf, count, dtype: int(allreduce(buf, count, dtype))),
                          ALLGATHER_FN(lambda user, send, recv, nbytes: int(allgather(send, recv, nbytes))))
        self._check(self.lib.asr_comm_init_custom(self.ctx, rank, world, self._comm_cbs[0], self._comm_cbs[1], None))

    def comm_destroy(self):
        self._check(self.lib.asr_comm_destroy(self.ctx))

    def comm_info(self):
        r, w = c_int(), c_int()
        self._check(self.lib.asr_comm_info(self.ctx, byref(r), byref(w)))
        return int(r.value), int(w.value)

    def comm_stats(self, reset=False):
        """collectives issued since the last reset: dict(allreduce_calls, allreduce_bytes, allgather_calls,
        allgather_bytes_per_rank)"""
        c = (c_int64 * 4)()
        self._check(self.lib.asr_comm_stats(self.ctx, c, 1 if reset else 0))
        return dict(allreduce_calls=int(c[0]), allreduce_bytes=int(c[1]), allgather_calls=int(c[2]),
                    allgather_bytes_per_rank=int(c[3]))

    def comm_timing(self, enable=True):
        """(ms, calls) spent in collectives since the previous call (asr_comm_timing); switches the bracketing on / off"""
        ms, calls = ctypes.c_double(), c_int64()
        self._check(self.lib.asr_comm_timing(self.ctx, 1 if enable else 0, byref(ms), byref(calls)))
        return float(ms.value), int(calls.value)

    def comm_library(self):
        """path of the librccl the communicator's entry points were bound from ('' without an RCCL communicator)"""
        buf = ctypes.create_string_buffer(1024)
        self._check(self.lib.asr_comm_library(self.ctx, buf, 1024))
        return buf.value.decode()

    def comm_allreduce_dev(self, buf_ptr, count, dtype=DTYPE_F64):
        self._check(self.lib.asr_comm_allreduce_dev(self.ctx, buf_ptr, count, dtype))

    def comm_allgather_dev(self, send_ptr, recv_ptr, bytes_per_rank):
        self._check(self.lib.asr_comm_allgather_dev(self.ctx, send_ptr, recv_ptr, bytes_per_rank))

    def allreduce_host(self, values):
        """sum of a small float64 host vector over all ranks of the communicator (staged through the device)"""
        v = np.ascontiguousarray(values, dtype=np.float64)
        buf = self.alloc(max(v.nbytes, 8)).upload(v)
        try:
            self.comm_allreduce_dev(buf.ptr, v.size, DTYPE_F64)
            return buf.download(v.shape, np.float64)
        finally:
            buf.free()

    def allgather_host(self, arr):
        """every rank's (equal-sized) host array, stacked along a new leading axis in rank order"""
        a = np.ascontiguousarray(arr)
        _, world = self.comm_info()
        send = self.alloc(max(a.nbytes, 1)).upload(a)
        recv = self.alloc(max(a.nbytes * world, 1))
        try:
            self.comm_allgather_dev(send.ptr, recv.ptr, a.nbytes)
            return recv.download((world,) + a.shape, a.dtype)
        finally:
            send.free()
            recv.free()

    def topk_merge_dev(self, part_idx_ptr, part_dist_ptr, n_parts, n_q_total, q_lo, n_q, k, idx_ptr, dist_ptr):
        self._check(self.lib.asr_topk_merge_dev(self.ctx, part_idx_ptr, part_dist_ptr, n_parts, n_q_total, q_lo, n_q, k,
                                                idx_ptr, dist_ptr))

    def rank_finish_dev(self, counts_ptr, dstar_ptr, n, ranks_ptr, dstar_out_ptr, ties_ptr):
        self._check(self.lib.asr_rank_finish_dev(self.ctx, counts_ptr, dstar_ptr, n, ranks_ptr, dstar_out_ptr, ties_ptr))

    def rank_sharded_dev(self, lv1_ptr, lv2_ptr, n_local, lv2_all_ptr, ranks_ptr, dstar_ptr, ties_ptr):
        self._check(self.lib.asr_rank_sharded_dev(self.ctx, lv1_ptr, lv2_ptr, n_local, lv2_all_ptr, ranks_ptr,
                                                  dstar_ptr, ties_ptr))

    def raw_download(self, ptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._check(self.lib.asr_dev_download(self.ctx, out.ctypes.data, ptr, out.nbytes))
        return out

    def raw_upload(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._check(self.lib.asr_dev_upload(self.ctx, ptr, arr.ctypes.data, arr.nbytes))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # -- host-buffer pipeline (run_eval.py:102-108,174 over a stream of host batches) ---------------
    def host_array(self, shape, dtype):
        """A NumPy array in page-locked host memory (asr_host_alloc); released with the engine or host_free()."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = c_void_p()
        self._check(self.lib.asr_host_alloc(self.ctx, nbytes, byref(p)))
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None and self.ctx:
            self._check(self.lib.asr_host_free(self.ctx, p))

    def eval_batches(self, sheets, specs, prepared=False, want_embeddings=False, out=None):
        """sheets[k] (n,1,H,W) uint8 / float32, specs[k] (n,1,92,42) float32 in host memory -> per batch the integer
        ranks, d*, tie counts (and the embeddings) of its all-pairs ranking; inputs double-buffered against
        compute.  out: optional dict of pre-allocated lists (ranks, dstar, ties[, lv1, lv2])."""
        nb = len(sheets)
        if nb != len(specs):
            raise ValueError("eval_batches: %d sheet batches but %d spectrogram batches" % (nb, len(specs)))
        if nb == 0:
            return dict(ranks=[], dstar=[], ties=[], lv1=[], lv2=[])
        xs, mode = [], None
        for a in sheets:
            a, m = self._view1_mode(a, prepared)
            if mode is not None and m != mode:
                raise ValueError("eval_batches: sheet batches of mixed dtype")
            xs.append(a)
            mode = m
        zs = [_f32c(z) for z in specs]
        n = xs[0].shape[0]
        if any(a.shape != xs[0].shape for a in xs) or any(z.shape != zs[0].shape for z in zs) or zs[0].shape[0] != n:
            raise ValueError("eval_batches: all batches must have the same shape")
        if zs[0].shape[2:] != (self.cfg.h2, self.cfg.w2):
            self.set_input_size(2, zs[0].shape[2], zs[0].shape[3])
        out = out or {}

        def outputs(key, shape, dtype):
            # the C side writes n (or n x 32) values through every pointer: a short, strided or wrong-typed array
            # would be a host heap overflow, so caller-supplied buffers are checked here
            arrs = out.get(key)
            if not arrs:
                return [np.empty(shape, dtype) for _ in range(nb)]
            if len(arrs) != nb:
                raise ValueError("eval_batches: out[%r] holds %d arrays for %d batches" % (key, len(arrs), nb))
            for a in arrs:
                if not isinstance(a, np.ndarray) or a.dtype != np.dtype(dtype) or a.shape != shape or \
                        not a.flags.c_contiguous or not a.flags.writeable:
                    raise ValueError("eval_batches: out[%r] needs writable C-contiguous %s arrays of shape %r"
                                     % (key, np.dtype(dtype).name, shape))
            return list(arrs)
        res = dict(ranks=outputs("ranks", (n,), np.int32), dstar=outputs("dstar", (n,), np.float64),
                   ties=outputs("ties", (n,), np.int32))
        if want_embeddings:
            res["lv1"] = outputs("lv1", (n, 32), np.float32)
            res["lv2"] = outputs("lv2", (n, 32), np.float32)

        def ptrs(arrs):
            return (c_void_p * nb)(*[a.ctypes.data for a in arrs])
        self._check(self.lib.asr_eval_batches(
            self.ctx, ptrs(xs), mode, ptrs(zs), nb, n, ptrs(res["ranks"]), ptrs(res["dstar"]), ptrs(res["ties"]),
            ptrs(res["lv1"]) if want_embeddings else None, ptrs(res["lv2"]) if want_embeddings else None))
        return res

    # -- parameters -------------------------------------------------------
    def param_sizes(self):
        n = self.lib.asr_param_count(self.ctx)
        out = []
        for i in range(n):
            v = c_int64()
            self._check(self.lib.asr_param_size(self.ctx, i, byref(v)))
            out.append(v.value)
        return out

    def set_params(self, params):
        """lasagne.layers.set_all_param_values(layers, params)."""
        arrs = [_f32c(p) for p in params]
        n = len(arrs)
        ptrs = (POINTER(c_float) * n)(*[a.ctypes.data_as(POINTER(c_float)) for a in arrs])
        sizes = (c_int64 * n)(*[a.size for a in arrs])
        self._check(self.lib.asr_set_params(self.ctx, ptrs, sizes, n))
        self._shapes = [a.shape for a in arrs]

    def get_params(self):
        """lasagne.layers.get_all_param_values(layers)."""
        sizes = self.param_sizes()
        shapes = getattr(self, "_shapes", [(s,) for s in sizes])
        arrs = [np.empty(shp, np.float32) for shp in shapes]
        n = len(arrs)
        ptrs = (POINTER(c_float) * n)(*[a.ctypes.data_as(POINTER(c_float)) for a in arrs])
        csz = (c_int64 * n)(*sizes)
        self._check(self.lib.asr_get_params(self.ctx, ptrs, csz, n))
        return arrs

    def set_cca(self, U, V, mean1, mean2):
        U, V, m1, m2 = _f32c(U), _f32c(V), _f32c(mean1), _f32c(mean2)
        assert U.shape == (32, 32) and V.shape == (32, 32) and m1.shape == (32,) and m2.shape == (32,)
        self._check(self.lib.asr_set_cca(self.ctx, U.ctypes.data, V.ctypes.data, m1.ctypes.data, m2.ctypes.data))

    # -- embedding ----------------------------------------------------------
    def set_input_size(self, view, h, w):
        """raw input size of `view`; the network is shape-agnostic (global pooling)."""
        self._check(self.lib.asr_set_input_size(self.ctx, view, h, w))
        if view == 1:
            self.cfg.h1, self.cfg.w1 = h, w
            rsz = MODEL_CONFIGS[self.model_name]["resize_view1"]
            self.net_h1, self.net_w1 = (h // 2, w // 2) if rsz else (h, w)
        else:
            self.cfg.h2, self.cfg.w2 = h, w

    def _view1_mode(self, x, prepared):
        if x.ndim != 4 or x.shape[1] != 1:
            raise ValueError("view-1 input must be (n, 1, H, W), got %r" % (x.shape,))
        rsz = MODEL_CONFIGS[self.model_name]["resize_view1"]
        if prepared:
            x = _f32c(x)
            if x.shape[2:] != (self.net_h1, self.net_w1):
                self.set_input_size(1, x.shape[2] * (2 if rsz else 1), x.shape[3] * (2 if rsz else 1))
            return x, IN_F32_PREPARED
        if x.shape[2:] != (self.cfg.h1, self.cfg.w1):
            self.set_input_size(1, x.shape[2], x.shape[3])
        if x.dtype == np.uint8:
            return np.ascontiguousarray(x), IN_U8_RAW
        return _f32c(x), IN_F32_RAW

    def embed_view1(self, x, prepared=True, features=False):
        """compute_v1_latent (run_eval.py:92-93); prepared=False folds
        model.prepare into the first kernel; features=True returns the
        pre-CCA tower output (refine_cca.py:86-87)."""
        x, mode = self._view1_mode(x, prepared)
        out = np.empty((x.shape[0], 32), np.float32)
        self._check(self.lib.asr_embed_view1(self.ctx, x.ctypes.data, mode, x.shape[0],
                                             OUT_FEATURES if features else OUT_LATENT, out.ctypes.data))
        return out

    def embed_view2(self, z, features=False):
        """compute_v2_latent (run_eval.py:94-95)."""
        z = _f32c(z)
        if z.ndim != 4 or z.shape[1] != 1:
            raise ValueError("view-2 input must be (n, 1, H, W), got %r" % (z.shape,))
        if z.shape[2:] != (self.cfg.h2, self.cfg.w2):
            self.set_input_size(2, z.shape[2], z.shape[3])
        out = np.empty((z.shape[0], 32), np.float32)
        self._check(self.lib.asr_embed_view2(self.ctx, z.ctypes.data, z.shape[0],
                                             OUT_FEATURES if features else OUT_LATENT, out.ctypes.data))
        return out

    def embed_both(self, x, z, prepared=False, features=False):
        """compute_output(X1, X2) -> [v1 latent, v2 latent] (utils/train_dcca_pool.py:158)."""
        x, mode = self._view1_mode(x, prepared)
        z = _f32c(z)
        if x.shape[0] != z.shape[0]:
            raise ValueError("embed_both: %d sheets but %d spectrograms" % (x.shape[0], z.shape[0]))
        if z.shape[2:] != (self.cfg.h2, self.cfg.w2):
            self.set_input_size(2, z.shape[2], z.shape[3])
        n = x.shape[0]
        o1, o2 = np.empty((n, 32), np.float32), np.empty((n, 32), np.float32)
        self._check(self.lib.asr_embed_both(self.ctx, x.ctypes.data, mode, z.ctypes.data, n,
                                            OUT_FEATURES if features else OUT_LATENT, o1.ctypes.data, o2.ctypes.data))
        return o1, o2

    def embed_view1_dev(self, x_ptr, mode, n, out_ptr, features=False):
        self._check(self.lib.asr_embed_view1_dev(self.ctx, x_ptr, mode, n,
                                                 OUT_FEATURES if features else OUT_LATENT, out_ptr))

    def embed_view2_dev(self, z_ptr, n, out_ptr, features=False):
        self._check(self.lib.asr_embed_view2_dev(self.ctx, z_ptr, n,
                                                 OUT_FEATURES if features else OUT_LATENT, out_ptr))

    # -- ranking -----------------------------------------------------------
    def rank(self, lv1, lv2, query_offset=0, n1_global=None):
        """Integer ranks / d* / tie counts of eval_retrieval
        (utils/train_dcca_pool.py:28-82) by counting, float64 distances."""
        lv1, lv2 = _f32c(lv1), _f32c(lv2)
        n1, dim = lv1.shape
        n2 = lv2.shape[0]
        assert lv2.shape[1] == dim
        ranks = np.empty(n1, np.int32)
        dstar = np.empty(n1, np.float64)
        ties = np.empty(n1, np.int32)
        self._check(self.lib.asr_rank(self.ctx, lv1.ctypes.data, n1, dim, lv2.ctypes.data, n2, dim, dim,
                                      query_offset, n1 if n1_global is None else n1_global,
                                      ranks.ctypes.data, dstar.ctypes.data, ties.ctypes.data))
        return ranks, dstar, ties

    def rank_dev(self, lv1_ptr, n1, lv2_ptr, n2, ranks_ptr, dstar_ptr, ties_ptr, dim=32, ld=32,
                 query_offset=0, n1_global=None):
        self._check(self.lib.asr_rank_dev(self.ctx, lv1_ptr, n1, ld, lv2_ptr, n2, ld, dim, query_offset,
                                          n1 if n1_global is None else n1_global, ranks_ptr, dstar_ptr, ties_ptr))

    def topk(self, db_codes, query_codes, k, idx_offset=0):
        """k nearest database codes per query by float64 cosine distance
        (audio_sheet_server.py:530-563) -> (idx (Q,k) int32, dist (Q,k) float64)."""
        db, q = _f32c(db_codes), _f32c(query_codes)
        if db.ndim != 2 or q.ndim != 2 or db.shape[1] != q.shape[1]:
            raise ValueError("topk expects (N,d) and (Q,d) arrays, got %r and %r" % (db.shape, q.shape))
        idx = np.empty((q.shape[0], k), np.int32)
        dist = np.empty((q.shape[0], k), np.float64)
        self._check(self.lib.asr_topk(self.ctx, db.ctypes.data, db.shape[0], db.shape[1], q.ctypes.data, q.shape[0],
                                      q.shape[1], q.shape[1], k, idx_offset, idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def topk_dev(self, db_ptr, n_db, q_ptr, n_q, k, idx_ptr, dist_ptr, dim=32, ld=32, idx_offset=0):
        self._check(self.lib.asr_topk_dev(self.ctx, db_ptr, n_db, ld, q_ptr, n_q, ld, dim, k, idx_offset,
                                          idx_ptr, dist_ptr))

    # -- resident code data base (audio_sheet_server.py:496-522, 530-563) ---------------------------------------
    def db_create(self, codes_ptr, n, dim=32, ld=None):
        """CodeDB over caller-owned device rows: norms / reciprocal norms / unit-length copy computed once."""
        return CodeDB(self, codes_ptr, n, dim, dim if ld is None else ld)

    # -- CCA re-estimation ------------------------------------------------------
    def cca_fit(self, H1, H2):
        """CCA('svd').fit (utils/cca.py:25-53,199-211) -> (U, V, m1, m2, coeffs);
        U, V, m1, m2 float32 as written back by refine_cca.py:104-107."""
        H1, H2 = _f32c(H1), _f32c(H2)
        if H1.ndim != 2 or H1.shape[1] != 32 or H2.shape != H1.shape:
            raise ValueError("cca_fit expects two (n, 32) arrays, got %r and %r" % (H1.shape, H2.shape))
        U, V = np.empty((32, 32), np.float32), np.empty((32, 32), np.float32)
        m1, m2 = np.empty(32, np.float32), np.empty(32, np.float32)
        coeffs = np.empty(32, np.float64)
        self._check(self.lib.asr_cca_fit(self.ctx, H1.ctypes.data, H2.ctypes.data, H1.shape[0], U.ctypes.data,
                                         V.ctypes.data, m1.ctypes.data, m2.ctypes.data, coeffs.ctypes.data))
        return U, V, m1, m2, coeffs

    def cca_fit_dev(self, h1_ptr, h2_ptr, n, u_ptr, v_ptr, means_ptr, coeffs_ptr):
        self._check(self.lib.asr_cca_fit_dev(self.ctx, h1_ptr, h2_ptr, n, u_ptr, v_ptr, means_ptr, coeffs_ptr))

    # -- training -------------------------------------------------------------
    def train_begin(self, batch_size):
        """allocate the device training state (Adam moments start at zero)."""
        self._check(self.lib.asr_train_begin(self.ctx, int(batch_size)))

    def train_end(self):
        self._check(self.lib.asr_train_end(self.ctx))

    def train_set_global_batch(self, n_global):
        """data parallel: the next steps carry this rank's distributed.shard_range(n_global, rank, world) rows of one
        batch of n_global rows (0: equal shards of batch * world rows)."""
        self._check(self.lib.asr_train_set_global_batch(self.ctx, int(n_global)))

    def train_step(self, x1_prepared, x2, lr):
        """iter_funcs['train'](X1, X2) -> (loss, corr) (utils/train_dcca_pool.py:154)."""
        x1, x2 = _f32c(x1_prepared), _f32c(x2)
        if x1.shape[0] != x2.shape[0]:
            raise ValueError("train_step: batch sizes differ")
        if x1.shape[2:] != (self.net_h1, self.net_w1) or x2.shape[2:] != (self.cfg.h2, self.cfg.w2):
            raise ValueError("train_step: input sizes %r / %r do not match the context (%d,%d)/(%d,%d); call "
                             "set_input_size before train_begin" % (x1.shape, x2.shape, self.net_h1, self.net_w1,
                                                                   self.cfg.h2, self.cfg.w2))
        loss = c_float()
        corr = np.empty(32, np.float32)
        self._check(self.lib.asr_train_step(self.ctx, x1.ctypes.data, x2.ctypes.data, x1.shape[0], lr,
                                            byref(loss), corr.ctypes.data))
        return float(loss.value), corr

    def burn_in(self, x1_prepared, x2):
        """iter_funcs['init_cca'](X1, X2) -> [v1 latent, v2 latent] of the train-mode graph
        (utils/train_dcca_pool.py:160-162): updates the BN / CCALayer running values only."""
        x1, x2 = _f32c(x1_prepared), _f32c(x2)
        n = x1.shape[0]
        lv1, lv2 = np.empty((n, 32), np.float32), np.empty((n, 32), np.float32)
        self._check(self.lib.asr_burn_in(self.ctx, x1.ctypes.data, x2.ctypes.data, n, lv1.ctypes.data, lv2.ctypes.data))
        return lv1, lv2

    def compute_gradients(self, x1_prepared, x2):
        """iter_funcs['compute_gradients'](X1, X2) (utils/train_dcca_pool.py:164): the gradients of the train loss
        wrt the trainable parameters as one flat array (order of get_params, other slots zero) and the loss."""
        x1, x2 = _f32c(x1_prepared), _f32c(x2)
        n = c_int64()
        self._check(self.lib.asr_opt_state_size(self.ctx, byref(n)))
        g = np.empty(n.value, np.float32)
        loss = c_float()
        self._check(self.lib.asr_compute_gradients(self.ctx, x1.ctypes.data, x2.ctypes.data, x1.shape[0],
                                                   g.ctypes.data, n.value, byref(loss)))
        return g, float(loss.value)

    def set_objective(self, weight=1.0, gamma=0.7, symmetric=False):
        """objectives() = get_contrastive_cos_loss(weight, gamma, symmetric) (models/objectives.py:30-69)"""
        self._check(self.lib.asr_set_objective(self.ctx, float(weight), float(gamma), 1 if symmetric else 0))

    def valid_loss(self, x1_prepared, x2):
        """iter_funcs['valid'](X1, X2) -> loss (utils/train_dcca_pool.py:155)."""
        x1, x2 = _f32c(x1_prepared), _f32c(x2)
        loss = c_float()
        self._check(self.lib.asr_valid_loss(self.ctx, x1.ctypes.data, x2.ctypes.data, x1.shape[0], byref(loss)))
        return float(loss.value)

    def get_opt_state(self):
        n = c_int64()
        self._check(self.lib.asr_opt_state_size(self.ctx, byref(n)))
        m, v = np.empty(n.value, np.float32), np.empty(n.value, np.float32)
        t = c_int32()
        self._check(self.lib.asr_get_opt_state(self.ctx, m.ctypes.data, v.ctypes.data, n.value, byref(t)))
        return dict(m=m, v=v, t=int(t.value))

    def set_opt_state(self, state):
        m, v = _f32c(state["m"]), _f32c(state["v"])
        self._check(self.lib.asr_set_opt_state(self.ctx, m.ctypes.data, v.ctypes.data, m.size, int(state["t"])))

    def debug_train_tensor(self, kind, view=0, index=0, batch=0):
        kinds = dict(z=0, x=1, stats=2, H=3, dH=4, lv=5, grad=6, master=7, loss=8, zsel=9, pool_mask=10)
        n = c_int64()
        self._check(self.lib.asr_debug_train_tensor(self.ctx, kinds[kind], view, index, batch, None, 0, byref(n)))
        out = np.empty(n.value, np.float32)
        self._check(self.lib.asr_debug_train_tensor(self.ctx, kinds[kind], view, index, batch, out.ctypes.data,
                                                    n.value, byref(n)))
        return out

    def cca_train_debug(self, H1, H2, cca_in, backward=True):
        H1, H2 = _f32c(H1), _f32c(H2)
        cin = np.concatenate([_f32c(a).ravel() for a in cca_in])
        assert cin.size == 5184
        B = H1.shape[0]
        cout = np.empty(5184, np.float32)
        lc = np.empty(33, np.float32)
        lv1, lv2 = np.empty((B, 32), np.float32), np.empty((B, 32), np.float32)
        dH1, dH2 = np.empty((B, 32), np.float32), np.empty((B, 32), np.float32)
        self._check(self.lib.asr_cca_train_debug(self.ctx, H1.ctypes.data, H2.ctypes.data, B, cin.ctypes.data,
                                                 cout.ctypes.data, lc.ctypes.data, lv1.ctypes.data, lv2.ctypes.data,
                                                 dH1.ctypes.data if backward else None,
                                                 dH2.ctypes.data if backward else None))
        new = [cout[0:1024].reshape(32, 32), cout[1024:2048].reshape(32, 32), cout[2048:2080], cout[2080:2112],
               cout[2112:3136].reshape(32, 32), cout[3136:4160].reshape(32, 32), cout[4160:5184].reshape(32, 32)]
        return dict(loss=float(lc[0]), corr=lc[1:], cca=new, lv1=lv1, lv2=lv2, dH1=dH1, dH2=dH2)

    # -- profiling -----------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.lib.asr_profile_enable(self.ctx, 1 if on else 0))

    def profile_filter(self, symbol=None):
        """only launches of this kernel symbol are timed (None: every kernel)"""
        self._check(self.lib.asr_profile_filter(self.ctx, (symbol or "").encode()))

    def profile_reset(self):
        self._check(self.lib.asr_profile_reset(self.ctx))

    def profile(self):
        """[{name, launches, total_ms, flops, bytes}] per kernel label."""
        out = []
        for i in range(self.lib.asr_profile_count(self.ctx)):
            name = ctypes.create_string_buffer(64)
            launches, ms, fl, by = c_int64(), c_double(), c_double(), c_double()
            self._check(self.lib.asr_profile_get(self.ctx, i, name, 64, byref(launches), byref(ms),
                                                 byref(fl), byref(by)))
            sym = ctypes.create_string_buffer(256)
            self._check(self.lib.asr_profile_symbol(self.ctx, i, sym, 256))
            out.append(dict(name=name.value.decode(), symbol=sym.value.decode(), launches=launches.value,
                            total_ms=ms.value, flops=fl.value, bytes=by.value))
        return out

    # -- debugging ------------------------------------------------------------
    def debug_activation(self, view, block, n):
        h, w, c = c_int(), c_int(), c_int()
        self._check(self.lib.asr_debug_activation(self.ctx, view, block, 0, None, byref(h), byref(w), byref(c)))
        out = np.empty((n, h.value, w.value, c.value), np.float32)
        self._check(self.lib.asr_debug_activation(self.ctx, view, block, n, out.ctypes.data,
                                                  byref(h), byref(w), byref(c)))
        return out
