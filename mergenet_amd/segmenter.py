"""Host-side mirror of the reference merger interfaces, bound to libmergenet_hip.so over ctypes.

Same names, argument meaning and error behaviour as the reference for this path:

* :func:`run_segmentation` -- the Cython binding ``utils/csegment/c_segment.pyx:30-86``
  (typed-buffer checks, clip to ``[eps32, 1-eps32]``, int32 offset array, output allocation,
  ``-1``-terminated class table read over ``range(H*W-1)``), calling the ABI-compatible
  ``c_run_segmentation`` (``utils/csegment/segment.cc:742-754``) of the HIP library.
* :class:`SegmenterOptions`, :class:`ObjectSegmenter` -- ``utils/segmenter.py:21-24,225-483``
  (Python-variant semantics: ``n1*n2`` denominator, bias inside, merge on ``>=``, ``prune``).
* :class:`Merger` -- device-resident API for PyTorch-ROCm callers (tensors in, tensors out,
  current HIP stream); PyTorch is only used for memory and streams.

There is NO CPU fallback: if the HIP library is missing or no GPU is visible the calls raise.
"""

from __future__ import annotations

import ctypes
import os
import threading
from collections import namedtuple
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MN_LIB") or os.path.join(_HERE, "libmergenet_hip.so")   # MN_LIB: a variant build (tuning only)

MN_VARIANT_CSEGMENT = 0
MN_VARIANT_PYSEGMENTER = 1
MN_MODE_AUTO, MN_MODE_EXACT, MN_MODE_ROUNDS, MN_MODE_COMPONENTS = 0, 1, 2, 3
MN_ERR_NO_BACKGROUND = -10
MN_ERR_UNPROVEN = -30
MN_DEBUG_GENERIC_EDGE_PASS, MN_DEBUG_NO_EVENTS, MN_DEBUG_NO_CORES = 1, 2, 4
MN_DEBUG_LEAN_EVENTS, MN_DEBUG_REPLAY = 16, 32
MN_PROVE_ALWAYS, MN_PROVE_BY_MODE, MN_PROVE_NEVER = 1, 0, -1   # mn_options.require_proof
MN_TIES_DEFAULT, MN_TIES_REFERENCE, MN_TIES_LOWEST_ID = 0, 1, 2   # mn_options.tie_order
MN_PROOF_NONE, MN_PROOF_CERTIFICATE, MN_PROOF_SEQUENTIAL, MN_PROOF_SEQUENTIAL_TIES = 0, 1, 2, 3   # mn_stats.proof

SegmenterOptions = namedtuple("SegmenterOptions",
                              ["same_different_bias", "object_merge_factor", "merge_logprob_bias"])


class MnOptions(ctypes.Structure):
    _fields_ = [("same_different_bias", ctypes.c_float), ("object_merge_factor", ctypes.c_float),
                ("merge_logprob_bias", ctypes.c_float), ("variant", ctypes.c_int),
                ("mode", ctypes.c_int), ("clip_inputs", ctypes.c_int),
                ("exact_limit", ctypes.c_int), ("finish_limit", ctypes.c_int),
                ("subrounds", ctypes.c_int), ("prune_threshold", ctypes.c_float),
                ("compute_logprob", ctypes.c_int), ("no_handover_refresh", ctypes.c_int),
                ("band_permille", ctypes.c_int), ("debug_flags", ctypes.c_int),
                ("require_proof", ctypes.c_int), ("core_radius", ctypes.c_int),
                ("tie_order", ctypes.c_int)]


class MnStats(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int), ("mode_used", ctypes.c_int), ("certified", ctypes.c_int),
                ("num_instances", ctypes.c_int), ("num_objects", ctypes.c_int),
                ("rounds", ctypes.c_int), ("finisher_steps", ctypes.c_int),
                ("cert_edge_violations", ctypes.c_int),
                ("cert_class_violations", ctypes.c_int), ("cert_record_violations", ctypes.c_int),
                ("initial_records", ctypes.c_longlong),
                ("merges", ctypes.c_longlong), ("total_logprob", ctypes.c_double),
                ("ms_score", ctypes.c_float), ("ms_class_pass", ctypes.c_float),
                ("ms_edge_pass", ctypes.c_float), ("ms_merge", ctypes.c_float),
                ("ms_output", ctypes.c_float), ("ms_total", ctypes.c_float),
                ("ms_cc_label", ctypes.c_float), ("ms_cc_sums", ctypes.c_float),
                ("ms_cc_edges", ctypes.c_float), ("ms_cc_cross", ctypes.c_float),
                ("proof", ctypes.c_int), ("cores_condemned", ctypes.c_int), ("tied_steps", ctypes.c_int), ("tied_merges", ctypes.c_int),
                ("tie_order_used", ctypes.c_int), ("tied_conflicts", ctypes.c_int)]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "reserved_i"}


_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_lib_handle = None

EXPORTS = ["mn_default_options", "mn_create", "mn_destroy", "mn_workspace_bytes",
           "mn_segment_device", "mn_segment_launch", "mn_segment_finish", "mn_segment_exact_batch", "mn_score_device", "mn_exact_phase_a_device", "mn_sweep_device", "mn_sweep_time_device", "mn_segment_host", "c_run_segmentation",
           "mn_prepare_device", "mn_upsample_mask_device", "mn_rle_points_device", "mn_rle_encode_host", "mn_sameness_targets_device", "mn_instance_scores_device",
           "mn_pack_wire_device", "mn_runs_wire_words", "mn_pack_runs_device", "mn_unpack_runs_device",
           "mn_unpack_runs_batch_device",
           "mn_last_status", "mn_status_string", "mn_version"]


def load_library() -> ctypes.CDLL:
    """Load libmergenet_hip.so (built in-tree by ``__graft_entry__.build``); fail loudly."""
    global _lib_handle
    if _lib_handle is not None:
        return _lib_handle
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP extension missing: %s (run __graft_entry__.build() / make -C "
                           "mergenet_amd/csrc); there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7.  If our library
    # were loaded first it would pull /opt/rocm's copy and torch would later see no device (or
    # vice versa), so when torch is installed its runtime is loaded first and we bind to it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.mn_default_options.argtypes = [ctypes.POINTER(MnOptions)]
    lib.mn_default_options.restype = None
    lib.mn_create.argtypes = [ctypes.c_int] * 5
    lib.mn_create.restype = ctypes.c_void_p
    lib.mn_destroy.argtypes = [ctypes.c_void_p]
    lib.mn_destroy.restype = None
    lib.mn_workspace_bytes.argtypes = [ctypes.c_void_p]
    lib.mn_workspace_bytes.restype = ctypes.c_size_t
    lib.mn_segment_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, _i32p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.POINTER(MnOptions), ctypes.c_void_p,
                                      ctypes.POINTER(MnStats)]
    lib.mn_segment_device.restype = ctypes.c_int
    lib.mn_segment_launch.argtypes = lib.mn_segment_device.argtypes[:-1]
    lib.mn_segment_launch.restype = ctypes.c_int
    lib.mn_segment_finish.argtypes = [ctypes.c_void_p, ctypes.POINTER(MnStats)]
    lib.mn_segment_finish.restype = ctypes.c_int
    _vpp = ctypes.POINTER(ctypes.c_void_p)
    lib.mn_segment_exact_batch.argtypes = [_vpp, ctypes.c_int, _vpp, ctypes.c_int, _vpp, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, _i32p, _vpp, _vpp, _vpp,
                                           ctypes.POINTER(MnOptions), ctypes.c_void_p, ctypes.POINTER(MnStats)]
    lib.mn_segment_exact_batch.restype = ctypes.c_int
    if hasattr(lib, "mn_sweep_time_device"):             # (absent from older variant builds: MN_LIB)
        lib.mn_sweep_time_device.argtypes = [ctypes.c_void_p, _vpp, _vpp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p,
                                             ctypes.POINTER(MnOptions), ctypes.c_void_p, ctypes.c_int, _f32p]
        lib.mn_sweep_time_device.restype = ctypes.c_int
    lib.mn_score_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p,
                                    ctypes.POINTER(MnOptions), ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, _f32p, _f32p]
    lib.mn_score_device.restype = ctypes.c_int
    lib.mn_exact_phase_a_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p,
                                            ctypes.POINTER(MnOptions), ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p]
    lib.mn_exact_phase_a_device.restype = ctypes.c_int
    lib.mn_sweep_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p, ctypes.POINTER(MnOptions),
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), _i32p]
    lib.mn_sweep_device.restype = ctypes.c_int
    lib.mn_segment_host.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int, _f32p, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p, _i32p, _i32p,
                                    _i32p, ctypes.POINTER(MnOptions), ctypes.POINTER(MnStats)]
    lib.mn_segment_host.restype = ctypes.c_int
    lib.c_run_segmentation.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, _i32p, _i32p, _i32p,
                                       ctypes.c_float, ctypes.c_float, ctypes.c_float]
    lib.c_run_segmentation.restype = None
    lib.mn_prepare_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.mn_prepare_device.restype = ctypes.c_int
    lib.mn_upsample_mask_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p]
    lib.mn_upsample_mask_device.restype = ctypes.c_int
    lib.mn_rle_points_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_int, _i32p, ctypes.c_void_p]
    lib.mn_rle_points_device.restype = ctypes.c_int
    lib.mn_rle_encode_host.argtypes = [_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong,
                                       ctypes.POINTER(ctypes.c_longlong), _i32p]
    lib.mn_rle_encode_host.restype = ctypes.c_longlong
    lib.mn_sameness_targets_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_int, _i32p, ctypes.c_int, ctypes.c_void_p,
                                               ctypes.c_void_p]
    lib.mn_sameness_targets_device.restype = ctypes.c_int
    lib.mn_instance_scores_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.mn_instance_scores_device.restype = ctypes.c_int
    lib.mn_pack_wire_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.mn_pack_wire_device.restype = ctypes.c_int
    lib.mn_runs_wire_words.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.mn_runs_wire_words.restype = ctypes.c_size_t
    lib.mn_pack_runs_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p]
    lib.mn_pack_runs_device.restype = ctypes.c_int
    lib.mn_unpack_runs_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.mn_unpack_runs_device.restype = ctypes.c_int
    lib.mn_last_status.restype = ctypes.c_int
    lib.mn_status_string.argtypes = [ctypes.c_int]
    lib.mn_status_string.restype = ctypes.c_char_p
    lib.mn_version.restype = ctypes.c_char_p
    _lib_handle = lib
    return lib


class MergeNetError(RuntimeError):
    def __init__(self, status: int):
        lib = load_library()
        super().__init__("mergenet_hip status %d: %s" % (status, lib.mn_status_string(status).decode()))
        self.status = status


def default_options(**overrides) -> MnOptions:
    o = MnOptions()
    load_library().mn_default_options(ctypes.byref(o))
    for k, v in overrides.items():
        setattr(o, k, v)
    return o


def _class_list(table: np.ndarray) -> List[int]:
    out = []
    for i in range(table.shape[0] - 1):       # c_segment.pyx:80-84
        if table[i] == -1:
            break
        out.append(int(table[i]))
    return out


def _check_buffer(name: str, a) -> np.ndarray:
    """The typed-buffer contract of c_segment.pyx:30-31 (float32, ndim 3, C-contiguous)."""
    if a is None:
        raise TypeError("Argument '%s' must not be None" % name)
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(a).__name__))
    if a.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float' but got '%s'" % a.dtype)
    if a.ndim != 3:
        raise ValueError("Buffer has wrong number of dimensions (expected 3, got %d)" % a.ndim)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("ndarray is not C-contiguous")
    return a


def run_segmentation(class_pred, adj_pred, num_classes: int, offset_list,
                     same_different_bias: float, object_merge_factor: float,
                     merge_logprob_bias: float):
    """Drop-in for ``csegment.c_segment.run_segmentation`` (c_segment.pyx:30-86).

    Returns ``(mask int32[H, W], object_class list)``; label 0 = every class-0 object.
    """
    class_pred = _check_buffer("class_pred", class_pred)
    adj_pred = _check_buffer("adj_pred", adj_pred)
    if offset_list is None or not isinstance(offset_list, list):
        raise TypeError("Argument 'offset_list' has incorrect type (expected list)")
    epsilon = np.finfo(np.float32).eps
    class_pred = np.ascontiguousarray(class_pred.clip(epsilon, 1.0 - epsilon), dtype=np.float32)
    adj_pred = np.ascontiguousarray(adj_pred.clip(epsilon, 1.0 - epsilon), dtype=np.float32)
    offset_array = np.ascontiguousarray(np.array(offset_list).astype(np.int32))
    class_dim = class_pred.shape[0]
    offset_dim = adj_pred.shape[0]
    img_height, img_width = adj_pred.shape[1], adj_pred.shape[2]
    mask_pred = np.zeros((img_height, img_width)).astype(np.int32)
    object_class_pred = np.zeros((1, img_height * img_width)).astype(np.int32)
    lib = load_library()
    lib.c_run_segmentation(class_pred.ctypes.data_as(_f32p), class_dim,
                           adj_pred.ctypes.data_as(_f32p), offset_dim, img_width, img_height,
                           int(num_classes), offset_array.ctypes.data_as(_i32p),
                           mask_pred.ctypes.data_as(_i32p), object_class_pred.ctypes.data_as(_i32p),
                           float(same_different_bias), float(object_merge_factor),
                           float(merge_logprob_bias))
    status = lib.mn_last_status()
    if status != 0:
        raise MergeNetError(status)      # the reference would exit(1) or crash
    return mask_pred, _class_list(object_class_pred[0])


class HostContext:
    """Owns an mn_context for host-pointer calls (numpy in, numpy out)."""

    def __init__(self, H: int, W: int, C: int, O: int, device: int = 0):
        self.lib = load_library()
        self.handle = self.lib.mn_create(device, H, W, C, O)
        if not self.handle:
            raise MergeNetError(self.lib.mn_last_status())
        self.shape = (H, W, C, O)

    def close(self):
        if self.handle:
            self.lib.mn_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workspace_bytes(self) -> int:
        return int(self.lib.mn_workspace_bytes(self.handle))

    def segment(self, class_probs: np.ndarray, same_probs: np.ndarray, offsets,
                opts: Optional[MnOptions] = None, want_partition: bool = True):
        cp = np.ascontiguousarray(class_probs, dtype=np.float32)
        sp = np.ascontiguousarray(same_probs, dtype=np.float32)
        C, H, W = cp.shape
        O = sp.shape[0]
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32).reshape(-1, 2))
        if off.shape[0] != O or sp.shape[1:] != (H, W):
            raise AssertionError("shape mismatch between class, sameness maps and offsets")
        opts = opts if opts is not None else default_options()
        mask = np.zeros((H, W), np.int32)
        table = np.zeros(H * W, np.int32)
        part = np.zeros((H, W), np.int32) if want_partition else None
        stats = MnStats()
        rc = self.lib.mn_segment_host(self.handle, cp.ctypes.data_as(_f32p), C,
                                      sp.ctypes.data_as(_f32p), O, W, H, C,
                                      off.ctypes.data_as(_i32p), mask.ctypes.data_as(_i32p),
                                      table.ctypes.data_as(_i32p),
                                      part.ctypes.data_as(_i32p) if part is not None else None,
                                      ctypes.byref(opts), ctypes.byref(stats))
        if rc != 0:
            raise MergeNetError(rc)
        return mask, _class_list(table), part, stats.as_dict()


_host_contexts = {}
_host_contexts_lock = threading.Lock()


def _cached_host_context(H: int, W: int, C: int, O: int) -> "HostContext":
    """The HostContext of this shape AND this thread, created on first use (at most two shapes per thread are
    kept).  A context serves one call at a time (mn_context: one thread at a time), so threads do not share
    one; the cache itself is guarded by a lock, and a context is only evicted by the thread that owns it."""
    tid = threading.get_ident()
    key = (tid, H, W, C, O)
    with _host_contexts_lock:
        ctx = _host_contexts.get(key)
        if ctx is None or not ctx.handle:
            mine = [k for k in _host_contexts if k[0] == tid]
            while len(mine) >= 2:
                _host_contexts.pop(mine.pop(0)).close()
            ctx = _host_contexts[key] = HostContext(H, W, C, O)
        return ctx


def close_cached_contexts() -> None:
    """Free the contexts ObjectSegmenter keeps between calls (call it when no run_segmentation is in flight)."""
    with _host_contexts_lock:
        while _host_contexts:
            _host_contexts.popitem()[1].close()


class ObjectSegmenter:
    """``utils/segmenter.py:225-483`` look-alike running on the GPU (Python-variant semantics).

    ``ObjectSegmenter(class_probs, sameness_probs, num_classes, offsets, opts).run_segmentation()``
    returns ``(mask int64[H, W], object_class list)`` after ``prune(200)``.  Shape mismatches
    raise ``AssertionError`` as the reference's asserts do (segmenter.py:245-250); a prune with
    no class-0 object raises ``NameError`` as the reference does (segmenter.py:356,365).
    """

    def __init__(self, nnet_class_probs, nnet_sameness_probs, num_classes, offsets, opts=None):
        self.opts = opts if opts is not None else self.default_options()
        epsilon = np.finfo(np.float32).eps
        self.class_probs = np.asarray(nnet_class_probs).clip(epsilon, 1.0 - epsilon)
        self.sameness_probs = np.asarray(nnet_sameness_probs).clip(epsilon, 1.0 - epsilon)
        self.num_classes = num_classes
        self.offsets = offsets
        class_dim, self.img_height, self.img_width = self.class_probs.shape
        offset_dim, img_height, img_width = self.sameness_probs.shape
        assert class_dim == self.num_classes
        assert offset_dim == len(self.offsets)
        assert self.img_height == img_height
        assert self.img_width == img_width
        self.stats = None

    def default_options(self):
        return SegmenterOptions(same_different_bias=0.0, object_merge_factor=1.0,
                                merge_logprob_bias=0.0)

    def run_segmentation(self, prune_threshold: float = 200.0, mode: int = MN_MODE_AUTO):
        # (one context per shape is kept between calls -- creating one allocates the whole workspace,
        #  hundreds of MB at full size, for a merge that takes a fraction of a millisecond)
        ctx = _cached_host_context(self.img_height, self.img_width, self.num_classes, len(self.offsets))
        o = default_options(same_different_bias=float(self.opts.same_different_bias),
                            object_merge_factor=float(self.opts.object_merge_factor),
                            merge_logprob_bias=float(self.opts.merge_logprob_bias),
                            variant=MN_VARIANT_PYSEGMENTER, mode=mode,
                            prune_threshold=float(prune_threshold))
        try:
            mask, classes, _, self.stats = ctx.segment(self.class_probs, self.sameness_probs,
                                                       self.offsets, o, want_partition=False)
        except MergeNetError as e:
            if e.status == MN_ERR_NO_BACKGROUND:
                raise NameError("name 'background_obj' is not defined") from None
            raise
        return mask.astype(np.int64), classes


class Merger:
    """Device-resident merger for PyTorch-ROCm callers: tensors in, tensors out, no host copies.

    ``class_probs`` [C,H,W] and ``same_probs`` [O,H,W] are float32 CUDA(HIP) tensors, e.g. the
    sigmoid outputs of the network (``utils/inference_utils.py:44,96``); ``clip_inputs=1`` fuses
    the binding's clip (c_segment.pyx:53-55) into the loads.
    """

    def __init__(self, H: int, W: int, C: int, O: int, device: Optional[int] = None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("mergenet_amd.Merger needs a HIP device; there is no CPU fallback")
        self.torch = torch
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.lib = load_library()
        self.handle = self.lib.mn_create(self.device, H, W, C, O)
        if not self.handle:
            raise MergeNetError(self.lib.mn_last_status())
        self.H, self.W, self.C, self.O = H, W, C, O

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mn_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workspace_bytes(self) -> int:
        return int(self.lib.mn_workspace_bytes(self.handle))

    def _check(self, class_probs, same_probs, offsets):
        torch = self.torch
        for t in (class_probs, same_probs):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 3):
                raise ValueError("expected contiguous float32 [K,H,W] tensors on the GPU")
            if t.device.index != self.device:
                raise ValueError("tensor lives on another device than the Merger")
        C, H, W = class_probs.shape
        O = same_probs.shape[0]
        if same_probs.shape[1:] != (H, W) or len(offsets) != O:
            raise AssertionError("shape mismatch between class, sameness maps and offsets")
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32).reshape(-1, 2))
        return C, H, W, O, off

    def segment(self, class_probs, same_probs, offsets, opts: Optional[MnOptions] = None,
                want_partition: bool = False):
        """Returns (mask int32[H,W] tensor, object_class int32[H*W] tensor, partition|None, stats)."""
        torch = self.torch
        C, H, W, O, off = self._check(class_probs, same_probs, offsets)
        opts = opts if opts is not None else default_options()
        dev = class_probs.device
        mask = torch.empty((H, W), dtype=torch.int32, device=dev)
        table = torch.empty((H * W,), dtype=torch.int32, device=dev)
        part = torch.empty((H, W), dtype=torch.int32, device=dev) if want_partition else None
        stats = MnStats()
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.mn_segment_device(self.handle, class_probs.data_ptr(), C,
                                        same_probs.data_ptr(), O, W, H, C,
                                        off.ctypes.data_as(_i32p), mask.data_ptr(),
                                        table.data_ptr(),
                                        part.data_ptr() if part is not None else None,
                                        ctypes.byref(opts), ctypes.c_void_p(stream),
                                        ctypes.byref(stats))
        if rc != 0:
            raise MergeNetError(rc)
        return mask, table, part, stats.as_dict()

    def segment_async(self, class_probs, same_probs, offsets, opts: Optional[MnOptions] = None,
                      want_partition: bool = False, out=None) -> "PendingSegment":
        """Queue one image (``mn_segment_launch``) and return at once; ``.result()`` of the returned
        object waits and gives what :meth:`segment` gives.  The Merger is busy until then -- use
        two of them alternately on one stream to keep the GPU busy across images: the launch of
        image i+1 then precedes the read-back of image i, and kernels of different images still
        do not overlap (their timings stay clean).

        ``out=(mask, table)`` (int32 [H,W] and [H*W] tensors of the caller) receives the result instead
        of fresh tensors.  A serving loop that feeds the same input and output buffers every time can
        add ``MN_DEBUG_REPLAY | MN_DEBUG_LEAN_EVENTS`` to ``opts.debug_flags``: from the third such call
        on, the library replays two recorded hipGraphs instead of issuing ~17 launches."""
        torch = self.torch
        C, H, W, O, off = self._check(class_probs, same_probs, offsets)
        opts = opts if opts is not None else default_options()
        dev = class_probs.device
        if out is not None:
            mask, table = out[0], out[1]
            if not (mask.is_cuda and mask.dtype == torch.int32 and mask.is_contiguous() and mask.numel() == H * W
                    and table.is_cuda and table.dtype == torch.int32 and table.numel() >= H * W):
                raise ValueError("out=(mask int32 [H,W], table int32 [H*W]) on the GPU")
        else:
            mask = torch.empty((H, W), dtype=torch.int32, device=dev)
            table = torch.empty((H * W,), dtype=torch.int32, device=dev)
        part = torch.empty((H, W), dtype=torch.int32, device=dev) if want_partition else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.mn_segment_launch(self.handle, class_probs.data_ptr(), C,
                                        same_probs.data_ptr(), O, W, H, C,
                                        off.ctypes.data_as(_i32p), mask.data_ptr(),
                                        table.data_ptr(),
                                        part.data_ptr() if part is not None else None,
                                        ctypes.byref(opts), ctypes.c_void_p(stream))
        if rc != 0:
            raise MergeNetError(rc)
        return PendingSegment(self, mask, table, part, (class_probs, same_probs, opts))

    def score(self, class_probs, same_probs, offsets, opts: Optional[MnOptions] = None,
              want_arrays: bool = False):
        """Phase A only.  Returns (ms_class_pass, ms_edge_pass[, cls uint8[H,W], best int64[H,W]])."""
        torch = self.torch
        C, H, W, O, off = self._check(class_probs, same_probs, offsets)
        opts = opts if opts is not None else default_options()
        dev = class_probs.device
        cls = torch.empty((H, W), dtype=torch.uint8, device=dev) if want_arrays else None
        best = torch.empty((H, W), dtype=torch.int64, device=dev) if want_arrays else None
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.mn_score_device(self.handle, class_probs.data_ptr(), C, same_probs.data_ptr(),
                                      O, W, H, C, off.ctypes.data_as(_i32p), ctypes.byref(opts),
                                      ctypes.c_void_p(stream),
                                      cls.data_ptr() if cls is not None else None,
                                      best.data_ptr() if best is not None else None,
                                      ctypes.byref(a), ctypes.byref(b))
        if rc != 0:
            raise MergeNetError(rc)
        if want_arrays:
            return a.value, b.value, cls, best
        return a.value, b.value


    def sweep(self, class_probs, same_probs, offsets, opts: Optional[MnOptions] = None):
        """The affinity-scoring sweep of the default path alone (mn_cc_sign).  Returns a dict: bits uint32[H,W]
        (as int32 tensor), neg float32[O,H,W] (NaN = not listed), cls uint8[H,W] | None, gsum int32[C,H*W/4] |
        None, logsum float, pixels_per_lane, fused_class, margin_edges."""
        torch = self.torch
        C, H, W, O, off = self._check(class_probs, same_probs, offsets)
        opts = opts if opts is not None else default_options()
        dev = class_probs.device
        bits = torch.empty((H, W), dtype=torch.int32, device=dev)
        neg = torch.empty((O, H, W), dtype=torch.float32, device=dev)
        cls = torch.zeros((H, W), dtype=torch.uint8, device=dev)
        gsum = torch.zeros((C, (H * W + 3) // 4), dtype=torch.int32, device=dev)
        logsum = ctypes.c_double(0.0)
        info = (ctypes.c_int * 3)()
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.mn_sweep_device(self.handle, class_probs.data_ptr(), C, same_probs.data_ptr(), O, W, H, C,
                                      off.ctypes.data_as(_i32p), ctypes.byref(opts), ctypes.c_void_p(stream),
                                      bits.data_ptr(), neg.data_ptr(), cls.data_ptr(), gsum.data_ptr(),
                                      ctypes.byref(logsum), info)
        if rc != 0:
            raise MergeNetError(rc)
        fused = bool(info[1])
        return dict(bits=bits, neg=neg, cls=cls if fused else None, gsum=gsum if fused else None,
                    logsum=logsum.value, pixels_per_lane=int(info[0]), fused_class=fused, margin_edges=int(info[2]))

    def sweep_time(self, inputs: Sequence, offsets, opts: Optional[MnOptions] = None, reps: int = 400) -> float:
        """Tuning aid (``mn_sweep_time_device``): microseconds per launch of the sweep alone, back to back over
        the (class_probs, same_probs) pairs of ``inputs`` in rotation."""
        C, H, W, O, off = self._check(inputs[0][0], inputs[0][1], offsets)
        opts = opts if opts is not None else default_options()
        n = len(inputs)
        vp = ctypes.c_void_p * n
        out = ctypes.c_float(0)
        stream = self.torch.cuda.current_stream(inputs[0][0].device).cuda_stream
        rc = self.lib.mn_sweep_time_device(self.handle, vp(*[a.data_ptr() for a, _ in inputs]),
                                           vp(*[b.data_ptr() for _, b in inputs]), n, C, O, W, H, C,
                                           off.ctypes.data_as(_i32p), ctypes.byref(opts), ctypes.c_void_p(stream),
                                           int(reps), ctypes.byref(out))
        if rc != 0:
            raise MergeNetError(rc)
        return out.value

    def exact_phase_a(self, class_probs, same_probs, offsets, opts: Optional[MnOptions] = None):
        """Phase A of the exact engine: (cls uint8[H,W], oml float32[O,H,W], prio float32[O,H,W]) in the
        layout of the oracle's phase-A export (NaN where an edge leaves the image)."""
        torch = self.torch
        C, H, W, O, off = self._check(class_probs, same_probs, offsets)
        opts = opts if opts is not None else default_options()
        dev = class_probs.device
        cls = torch.empty((H, W), dtype=torch.uint8, device=dev)
        oml = torch.empty((O, H, W), dtype=torch.float32, device=dev)
        prio = torch.empty((O, H, W), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.mn_exact_phase_a_device(self.handle, class_probs.data_ptr(), C, same_probs.data_ptr(),
                                              O, W, H, C, off.ctypes.data_as(_i32p), ctypes.byref(opts),
                                              ctypes.c_void_p(stream), cls.data_ptr(), oml.data_ptr(),
                                              prio.data_ptr())
        if rc != 0:
            raise MergeNetError(rc)
        return cls, oml, prio

    def prepare(self, maps, out_height: int, out_width: int, apply_sigmoid: bool = False,
                clip: bool = True):
        """Network output -> merger input on the device: optional sigmoid, bilinear resize with
        cv2.resize coordinates (egs/cityscape/local/segment.py:115-123) and clip, one pass.
        maps: float32 [K, Hin, Win] tensor on this GPU.  Returns float32 [K, out_height, out_width]."""
        torch = self.torch
        if not (maps.is_cuda and maps.dtype == torch.float32 and maps.is_contiguous() and maps.dim() == 3):
            raise ValueError("expected a contiguous float32 [K,H,W] tensor on the GPU")
        K, Hin, Win = maps.shape
        out = torch.empty((K, out_height, out_width), dtype=torch.float32, device=maps.device)
        stream = torch.cuda.current_stream(maps.device).cuda_stream
        rc = self.lib.mn_prepare_device(self.handle, maps.data_ptr(), K, Hin, Win, out.data_ptr(),
                                        out_height, out_width, int(apply_sigmoid), int(clip),
                                        ctypes.c_void_p(stream))
        if rc != 0:
            raise MergeNetError(rc)
        return out

    def upsample_mask(self, mask, out_height: int, out_width: int):
        """Instance mask back to the image size, cv2 INTER_NEAREST coordinates (segment.py:146-149)."""
        torch = self.torch
        if not (mask.is_cuda and mask.dtype == torch.int32 and mask.is_contiguous() and mask.dim() == 2):
            raise ValueError("expected a contiguous int32 [H,W] tensor on the GPU")
        out = torch.empty((out_height, out_width), dtype=torch.int32, device=mask.device)
        stream = torch.cuda.current_stream(mask.device).cuda_stream
        rc = self.lib.mn_upsample_mask_device(self.handle, mask.data_ptr(), mask.shape[0],
                                              mask.shape[1], out.data_ptr(), out_height, out_width,
                                              ctypes.c_void_p(stream))
        if rc != 0:
            raise MergeNetError(rc)
        return out


    def encode_rle(self, mask, num_instances: int, drop_zero_area: bool = False):
        """COCO run-length encoding of every instance of an int32 [H,W] mask on this GPU.

        Equivalent of ``[maskUtils.encode(np.asfortranarray(mask == i)) for i in 1..K]``
        (egs/cityscape/local/segment.py:165-186) from ONE device pass over the mask (the run
        boundaries of all instances at once) and one native host pass that groups them and writes
        pycocotools' compressed counts strings.  Returns a list of ``{"size": [H, W], "counts": bytes,
        "area": pixels, "label": k}``, one per instance in label order; ``drop_zero_area`` leaves out
        instances without pixels (e.g. lost in the nearest-neighbour resize), as
        ``egs/cityscape/local/evaluate.py:52-54`` does before COCOeval.
        """
        torch = self.torch
        if not (mask.is_cuda and mask.dtype == torch.int32 and mask.is_contiguous() and mask.dim() == 2):
            raise ValueError("expected a contiguous int32 [H,W] tensor on the GPU")
        H, W = mask.shape
        cap = getattr(self, "_rle_cap", max(1024, H * W // 16))
        while True:
            if getattr(self, "_rle_pts", None) is None or self._rle_pts.shape[1] != cap:
                self._rle_pts = torch.empty((3, cap), dtype=torch.int32, device=mask.device)
                self._rle_host = torch.empty((3, cap), dtype=torch.int32).pin_memory()
                self._rle_cap = cap
            n = ctypes.c_int(0)
            stream = torch.cuda.current_stream(mask.device).cuda_stream
            rc = self.lib.mn_rle_points_device(self.handle, mask.data_ptr(), H, W, self._rle_pts.data_ptr(),
                                               cap, ctypes.byref(n), ctypes.c_void_p(stream))
            if rc == -4 and n.value > cap:
                cap = n.value
                continue
            if rc != 0:
                raise MergeNetError(rc)
            break
        nn = n.value
        for r in range(3):                                    # only the used part of each row travels
            self._rle_host[r, :nn].copy_(self._rle_pts[r, :nn], non_blocking=True)
        torch.cuda.current_stream(mask.device).synchronize()
        K = int(num_instances)
        offsets = (ctypes.c_longlong * (K + 1))()
        areas = (ctypes.c_int * max(1, K))()
        out_cap = 8 * nn + 16 * K + 64                        # a count takes at most 7 bytes
        out = ctypes.create_string_buffer(out_cap)
        need = self.lib.mn_rle_encode_host(ctypes.cast(self._rle_host.data_ptr(), _i32p), cap, nn, H, W, K,
                                           ctypes.cast(out, ctypes.c_void_p), out_cap, offsets, areas)
        if need < 0 or need > out_cap:
            raise MergeNetError(int(need) if need < 0 else -20)
        raw = out.raw
        res = []
        for k in range(1, K + 1):
            if drop_zero_area and areas[k - 1] == 0:
                continue
            res.append({"size": [H, W], "counts": raw[offsets[k - 1]:offsets[k]], "area": int(areas[k - 1]),
                        "label": k})
        return res


    def sameness_targets(self, mask, offsets):
        """Instance mask int32 [H,W] -> float32 [O,H,W] sameness targets (utils/dataset.py:259-277)."""
        torch = self.torch
        if not (mask.is_cuda and mask.dtype == torch.int32 and mask.is_contiguous() and mask.dim() == 2):
            raise ValueError("expected a contiguous int32 [H,W] tensor on the GPU")
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32).reshape(-1, 2))
        H, W = mask.shape
        out = torch.empty((off.shape[0], H, W), dtype=torch.float32, device=mask.device)
        stream = torch.cuda.current_stream(mask.device).cuda_stream
        rc = self.lib.mn_sameness_targets_device(self.handle, mask.data_ptr(), H, W,
                                                 off.ctypes.data_as(_i32p), off.shape[0],
                                                 out.data_ptr(), ctypes.c_void_p(stream))
        if rc != 0:
            raise MergeNetError(rc)
        return out

    def instance_scores(self, num_instances: int, device=None):
        """lp[cls] - lp[0] of each instance of the last segment() call (float32 [K])."""
        torch = self.torch
        dev = torch.device("cuda", self.device) if device is None else device
        out = torch.zeros((max(1, num_instances),), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.mn_instance_scores_device(self.handle, out.data_ptr(), ctypes.c_void_p(stream))
        if rc != 0:
            raise MergeNetError(rc)
        return out[:num_instances]


class PendingSegment:
    """An image queued by :meth:`Merger.segment_async`; ``result()`` finishes it (once)."""

    def __init__(self, merger, mask, table, part, keepalive):
        self._merger, self._out, self._keepalive = merger, (mask, table, part), keepalive
        self._done = None

    def result(self):
        if self._done is None:
            stats = MnStats()
            rc = self._merger.lib.mn_segment_finish(self._merger.handle, ctypes.byref(stats))
            self._keepalive = None
            if rc != 0:
                raise MergeNetError(rc)
            self._done = self._out + (stats.as_dict(),)
        return self._done


class ExactBatch:
    """``count`` images of one shape through the exact engine in ONE launch of its loop
    (``mn_segment_exact_batch``): the engine is one wavefront per image, so images in flight are its
    throughput.  One context (workspace) per image; ``segment`` takes lists of [C,H,W] / [O,H,W] tensors
    (at most ``count``) and returns a list of what :meth:`Merger.segment` returns."""

    def __init__(self, H: int, W: int, C: int, O: int, count: int, device: Optional[int] = None):
        if count < 1:
            raise ValueError("count >= 1")
        self.mergers = [Merger(H, W, C, O, device=device) for _ in range(count)]
        self.lib = self.mergers[0].lib
        self.torch = self.mergers[0].torch

    def segment(self, class_probs: Sequence, same_probs: Sequence, offsets, opts: Optional[MnOptions] = None,
                want_partition: bool = False):
        torch = self.torch
        n = len(class_probs)
        if n < 1 or n > len(self.mergers) or len(same_probs) != n:
            raise ValueError("between 1 and %d images" % len(self.mergers))
        shape = None
        for cp, sp in zip(class_probs, same_probs):
            C, H, W, O, off = self.mergers[0]._check(cp, sp, offsets)
            if shape is not None and shape != (C, H, W, O):
                raise ValueError("the images of a batch have one shape")
            shape = (C, H, W, O)
        opts = opts if opts is not None else default_options()
        dev = class_probs[0].device
        masks = [torch.empty((H, W), dtype=torch.int32, device=dev) for _ in range(n)]
        tables = [torch.empty((H * W,), dtype=torch.int32, device=dev) for _ in range(n)]
        parts = [torch.empty((H, W), dtype=torch.int32, device=dev) for _ in range(n)] if want_partition else None
        vp = ctypes.c_void_p * n
        stats = (MnStats * n)()
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.mn_segment_exact_batch(
            vp(*[m.handle for m in self.mergers[:n]]), n, vp(*[t.data_ptr() for t in class_probs]), C,
            vp(*[t.data_ptr() for t in same_probs]), O, W, H, C, off.ctypes.data_as(_i32p),
            vp(*[t.data_ptr() for t in masks]), vp(*[t.data_ptr() for t in tables]),
            vp(*[t.data_ptr() for t in parts]) if parts is not None else None,
            ctypes.byref(opts), ctypes.c_void_p(stream), stats)
        if rc != 0:
            raise MergeNetError(rc)
        return [(masks[i], tables[i], parts[i] if parts is not None else None, stats[i].as_dict())
                for i in range(n)]

    def close(self):
        for m in self.mergers:
            m.close()
        self.mergers = []


class MergerPool:
    """Several images in flight on one GPU: ``depth`` contexts, each with its own HIP stream and
    its own host thread.

    One merge has a host round trip (record count and statistics) and mostly latency-bound
    kernels, so a single context leaves the GPU idle part of the time; four images in flight
    nearly double the images per second on an MI355X.  ``submit`` returns a
    ``concurrent.futures.Future`` whose result is what ``Merger.segment`` returns; inputs may
    still be in flight on the submitting thread's current stream (the worker waits for them), and
    the outputs are safe to use on that stream.

        pool = MergerPool(H, W, C, O, depth=4)
        futures = [pool.submit(cp, sp, offsets, opts) for cp, sp in images]
        for f in futures:
            mask, class_table, _, stats = f.result()
        pool.close()
    """

    def __init__(self, H: int, W: int, C: int, O: int, depth: int = 4, device: Optional[int] = None):
        import queue
        import threading
        import torch
        if depth < 1:
            raise ValueError("depth >= 1")
        self.torch = torch
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.mergers = [Merger(H, W, C, O, device=self.device) for _ in range(depth)]
        self.jobs = queue.Queue()
        self.threads = [threading.Thread(target=self._work, args=(m,), daemon=True) for m in self.mergers]
        for t in self.threads:
            t.start()

    def _work(self, merger):
        torch = self.torch
        dev = torch.device("cuda", self.device)
        torch.cuda.set_device(dev)
        stream = torch.cuda.Stream(dev)
        while True:
            job = self.jobs.get()
            if job is None:
                return
            fut, ready, home, args, kwargs = job
            if not fut.set_running_or_notify_cancel():
                continue
            try:
                with torch.cuda.stream(stream):
                    stream.wait_event(ready)             # the producer of the inputs
                    out = merger.segment(*args, **kwargs)   # returns after its stream has drained
                    for t in out[:3]:
                        if t is not None:
                            t.record_stream(home)        # allocator: also in use on the caller's stream
                fut.set_result(out)
            except BaseException as e:                   # noqa: BLE001 -- handed to the caller
                fut.set_exception(e)

    def submit(self, class_probs, same_probs, offsets, opts: Optional[MnOptions] = None,
               want_partition: bool = False):
        from concurrent.futures import Future
        torch = self.torch
        if not self.threads:
            raise RuntimeError("MergerPool is closed")
        home = torch.cuda.current_stream(torch.device("cuda", self.device))
        ready = torch.cuda.Event()
        ready.record(home)
        fut = Future()
        self.jobs.put((fut, ready, home, (class_probs, same_probs, offsets),
                       {"opts": opts, "want_partition": want_partition}))
        return fut

    def map(self, images, offsets, opts: Optional[MnOptions] = None):
        """Results for an iterable of (class_probs, same_probs), in order, all images in flight."""
        futures = [self.submit(cp, sp, offsets, opts) for cp, sp in images]
        return [f.result() for f in futures]

    def close(self):
        for _ in self.threads:
            self.jobs.put(None)
        for t in self.threads:
            t.join()
        self.threads = []
        for m in self.mergers:
            m.close()
        self.mergers = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_wire(mask, class_table, num_instances: int, wire, max_instances: int,
              total_logprob: float = float("nan")) -> None:
    """Device tensors: int32 mask [H,W] + class table -> int16 wire buffer of the mask exchange
    (``mn_pack_wire_device``; layout in ``mergenet_amd/distributed.py``).  Runs on the current
    torch stream of the mask's device."""
    import torch
    lib = load_library()
    n = mask.numel()
    if (mask.dtype != torch.int32 or class_table.dtype != torch.int32 or wire.dtype != torch.int16
            or not mask.is_contiguous() or wire.numel() < n + 1 + max_instances + 4
            or class_table.numel() < num_instances):
        raise AssertionError("pack_wire: int32 mask/table, int16 wire of n + 1 + max_instances + 4")
    stream = torch.cuda.current_stream(mask.device).cuda_stream
    rc = lib.mn_pack_wire_device(mask.data_ptr(), class_table.data_ptr(), int(num_instances),
                                 float(total_logprob), n, int(max_instances), wire.data_ptr(),
                                 ctypes.c_void_p(stream))
    if rc != 0:
        raise MergeNetError(rc)



def runs_wire_words(capacity: int, max_instances: int) -> int:
    """int32 words of the run-length wire buffer (``mn_runs_wire_words``)."""
    return 4 + capacity + (capacity + 1) // 2 + (max_instances + 3) // 4


def pack_runs(merger, mask, class_table, num_instances: int, wire, capacity: int, max_instances: int,
              total_logprob: float = float("nan")) -> None:
    """Device tensors: int32 mask [H,W] + class table -> int32 run-length wire buffer
    (``mn_pack_runs_device``: row-major label change points; layout in include/mergenet_hip.h).
    Runs on the current torch stream; no host synchronisation."""
    import torch
    n = mask.numel()
    if (mask.dtype != torch.int32 or class_table.dtype != torch.int32 or wire.dtype != torch.int32
            or not mask.is_contiguous() or wire.numel() < runs_wire_words(capacity, max_instances)
            or class_table.numel() < num_instances):
        raise AssertionError("pack_runs: int32 mask/table, int32 wire of runs_wire_words(capacity, max_instances)")
    stream = torch.cuda.current_stream(mask.device).cuda_stream
    rc = merger.lib.mn_pack_runs_device(merger.handle, mask.data_ptr(), class_table.data_ptr(),
                                        int(num_instances), float(total_logprob), n, int(capacity),
                                        int(max_instances), wire.data_ptr(), ctypes.c_void_p(stream))
    if rc != 0:
        raise MergeNetError(rc)


def unpack_runs_batch(wires, height: int, width: int, capacity: int, max_instances: int):
    """`count` run-length wires (device, int32 [count, words], rows may be strided) -> (masks int32 [count,H,W],
    class tables int32 [count, max_instances]) in ONE launch."""
    import torch
    lib = load_library()
    count = int(wires.shape[0])
    masks = torch.empty((count, height, width), dtype=torch.int32, device=wires.device)
    tables = torch.empty((count, max_instances), dtype=torch.int32, device=wires.device)
    stream = torch.cuda.current_stream(wires.device).cuda_stream
    lib.mn_unpack_runs_batch_device.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p]
    lib.mn_unpack_runs_batch_device.restype = ctypes.c_int
    rc = lib.mn_unpack_runs_batch_device(wires.data_ptr(), int(wires.stride(0)), count, height * width,
                                         int(capacity), int(max_instances), masks.data_ptr(), tables.data_ptr(),
                                         ctypes.c_void_p(stream))
    if rc != 0:
        raise MergeNetError(rc)
    return masks, tables


def unpack_runs(wire, height: int, width: int, capacity: int, max_instances: int):
    """Run-length wire buffer (device, int32) -> (mask int32 [H,W], class table int32 [max_instances])."""
    import torch
    lib = load_library()
    mask = torch.empty((height, width), dtype=torch.int32, device=wire.device)
    table = torch.empty((max_instances,), dtype=torch.int32, device=wire.device)
    stream = torch.cuda.current_stream(wire.device).cuda_stream
    rc = lib.mn_unpack_runs_device(wire.data_ptr(), height * width, int(capacity), int(max_instances),
                                   mask.data_ptr(), table.data_ptr(), ctypes.c_void_p(stream))
    if rc != 0:
        raise MergeNetError(rc)
    return mask, table
