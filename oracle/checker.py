"""ctypes front-ends of the CPU checkers + partition comparison helpers.  TEST INFRASTRUCTURE.

``run_csegment``    our restatement of the reference C++ merger (oracle/csegment_oracle.cpp)
``run_reference``   the reference's own segment.cc, compiled unmodified into oracle/_ref/
                    (present only if oracle/Makefile ran where /root/reference exists)
``run_pysegmenter`` our restatement of the reference Python merger (oracle/pysegmenter_oracle.cpp)

The host-side preprocessing mirrors the reference binding ``utils/csegment/c_segment.pyx:53-84``
(clip to [eps32, 1-eps32], int32 offset array, -1-terminated class table read over
``range(H*W-1)``) and, for the Python variant, ``utils/segmenter.py:232-234``.
"""

from __future__ import annotations

import contextlib
import ctypes
import os
import subprocess
import sys
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
EPS32 = np.finfo(np.float32).eps

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(verbose: bool = False) -> None:
    """Compile the checkers (and oracle/_ref when the reference tree is present)."""
    res = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building the oracle failed")


def _load(name: str) -> ctypes.CDLL:
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    return ctypes.CDLL(path)


def have_reference() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libcsegment_ref.so"))


@contextlib.contextmanager
def _quiet_stdout():
    """The reference prints progress to C++ cout; keep test logs readable."""
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    try:
        os.dup2(devnull, 1)
        yield
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)


@dataclass
class OracleResult:
    mask: np.ndarray                 # int32 [H, W]; 0 = every class-0 object, 1..K instances
    object_class: List[int]          # class of label k at index k-1
    partition: Optional[np.ndarray]  # int32 [H, W] surviving object id, before the class-0 collapse
    total_logprob: Optional[float]
    stats: dict


def _prep(class_probs, same_probs, offsets, dtype=np.float32):
    cp = np.ascontiguousarray(np.asarray(class_probs).clip(EPS32, 1.0 - EPS32), dtype=dtype)
    sp = np.ascontiguousarray(np.asarray(same_probs).clip(EPS32, 1.0 - EPS32), dtype=dtype)
    if cp.ndim != 3 or sp.ndim != 3 or cp.shape[1:] != sp.shape[1:]:
        raise ValueError("class_probs [C,H,W] and same_probs [O,H,W] must share H, W")
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32).reshape(-1, 2))
    if off.shape[0] != sp.shape[0]:
        raise ValueError("len(offsets) must equal same_probs.shape[0]")
    return cp, sp, off


def _class_list(table: np.ndarray) -> List[int]:
    out = []
    for i in range(table.shape[0] - 1):       # c_segment.pyx:80-84 reads range(H*W - 1)
        if table[i] == -1:
            break
        out.append(int(table[i]))
    return out


def run_csegment(class_probs, same_probs, num_classes: int, offsets: Sequence[Tuple[int, int]],
                 same_different_bias: float = 0.0, object_merge_factor: float = 1.0,
                 merge_logprob_bias: float = 0.0) -> OracleResult:
    lib = _load("libcsegment_oracle.so")
    fn = lib.oracle_csegment_run
    fn.restype = ctypes.c_int
    fn.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                   ctypes.c_int, _i32p, _i32p, _i32p, ctypes.c_float, ctypes.c_float,
                   ctypes.c_float, _i32p, _f64p]
    cp, sp, off = _prep(class_probs, same_probs, offsets)
    C, H, W = cp.shape
    mask = np.zeros((H, W), np.int32)
    table = np.zeros(H * W, np.int32)
    part = np.zeros((H, W), np.int32)
    stats = np.zeros(10, np.float64)
    rc = fn(cp.ctypes.data_as(_f32p), C, sp.ctypes.data_as(_f32p), sp.shape[0], W, H,
            int(num_classes), off.ctypes.data_as(_i32p), mask.ctypes.data_as(_i32p),
            table.ctypes.data_as(_i32p), float(same_different_bias), float(object_merge_factor),
            float(merge_logprob_bias), part.ctypes.data_as(_i32p), stats.ctypes.data_as(_f64p))
    if rc != 0:
        raise ValueError("oracle_csegment_run rejected its arguments (code %d)" % rc)
    names = ["total_logprob", "n_objects", "n_pops", "n_merges", "t_build_s", "t_loop_s",
             "n_rescored", "n_initial_records", "n_live_pops", "_"]
    return OracleResult(mask, _class_list(table), part, float(stats[0]),
                        dict(zip(names, stats.tolist())))


def phase_a(class_probs, same_probs, num_classes: int, offsets: Sequence[Tuple[int, int]],
            same_different_bias: float = 0.0, object_merge_factor: float = 1.0,
            merge_logprob_bias: float = 0.0):
    """Phase A of the reference constructor (segment.cc:5-46,107-150,198-231): returns
    (cls int32[H,W], oml float32[O,H,W], prio float32[O,H,W]); NaN where the edge leaves the image."""
    lib = _load("libcsegment_oracle.so")
    fn = lib.oracle_csegment_phase_a
    fn.restype = ctypes.c_int
    fn.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                   ctypes.c_int, _i32p, ctypes.c_float, ctypes.c_float, ctypes.c_float, _i32p,
                   _f32p, _f32p]
    cp, sp, off = _prep(class_probs, same_probs, offsets)
    C, H, W = cp.shape
    O = sp.shape[0]
    cls = np.zeros((H, W), np.int32)
    oml = np.zeros((O, H, W), np.float32)
    prio = np.zeros((O, H, W), np.float32)
    rc = fn(cp.ctypes.data_as(_f32p), C, sp.ctypes.data_as(_f32p), O, W, H, int(num_classes),
            off.ctypes.data_as(_i32p), float(same_different_bias), float(object_merge_factor),
            float(merge_logprob_bias), cls.ctypes.data_as(_i32p), oml.ctypes.data_as(_f32p),
            prio.ctypes.data_as(_f32p))
    if rc != 0:
        raise ValueError("oracle_csegment_phase_a rejected its arguments (code %d)" % rc)
    return cls, oml, prio


def best_initial_record(prio: np.ndarray, offsets) -> Tuple[np.ndarray, np.ndarray]:
    """Per pixel the best incident initial record with priority >= 0: (priority float32[H,W],
    partner pixel id int64[H,W]; -1 / NaN where the pixel has none).  Outgoing edge k of p is
    prio[k, p]; the incoming one is prio[k, p - o_k].  Ties: lowest partner id."""
    O, H, W = prio.shape
    best = np.full((H, W), -np.inf, np.float64)
    partner = np.full((H, W), -1, np.int64)
    ids = np.arange(H * W, dtype=np.int64).reshape(H, W)

    def consider(pr, q, r0, r1, c0, c1):
        cur_b, cur_p = best[r0:r1, c0:c1], partner[r0:r1, c0:c1]
        ok = ~np.isnan(pr) & (pr >= 0)
        better = ok & ((pr > cur_b) | ((pr == cur_b) & (q < cur_p)))
        cur_b[better] = pr[better]
        cur_p[better] = q[better]

    for k, (di, dj) in enumerate(offsets):
        r0, r1 = max(0, -di), min(H, H - di)
        c0, c1 = max(0, -dj), min(W, W - dj)
        if r0 >= r1 or c0 >= c1:
            continue
        src = prio[k, r0:r1, c0:c1].astype(np.float64)
        consider(src, ids[r0 + di:r1 + di, c0 + dj:c1 + dj], r0, r1, c0, c1)          # outgoing
        consider(src, ids[r0:r1, c0:c1], r0 + di, r1 + di, c0 + dj, c1 + dj)          # incoming
    out = np.where(partner >= 0, best, np.nan).astype(np.float32)
    return out, partner


def run_reference(class_probs, same_probs, num_classes: int, offsets: Sequence[Tuple[int, int]],
                  same_different_bias: float = 0.0, object_merge_factor: float = 1.0,
                  merge_logprob_bias: float = 0.0) -> OracleResult:
    """The reference's own compiled segment.cc (ABI: utils/csegment/segment.cc:742-754)."""
    lib = ctypes.CDLL(os.path.join(_HERE, "_ref", "libcsegment_ref.so"))
    fn = lib.c_run_segmentation
    fn.restype = None
    fn.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                   ctypes.c_int, _i32p, _i32p, _i32p, ctypes.c_float, ctypes.c_float,
                   ctypes.c_float]
    cp, sp, off = _prep(class_probs, same_probs, offsets)
    C, H, W = cp.shape
    mask = np.zeros((H, W), np.int32)
    table = np.zeros(H * W, np.int32)
    with _quiet_stdout():
        fn(cp.ctypes.data_as(_f32p), C, sp.ctypes.data_as(_f32p), sp.shape[0], W, H,
           int(num_classes), off.ctypes.data_as(_i32p), mask.ctypes.data_as(_i32p),
           table.ctypes.data_as(_i32p), float(same_different_bias), float(object_merge_factor),
           float(merge_logprob_bias))
    return OracleResult(mask, _class_list(table), None, None, {})


# ---------------------------------------------------------------------------------------------
# comparison helpers (plain numpy, shared with bench.py's id-match report)

from mergenet_amd.labels import (canonical, masks_equivalent, partition_mismatch,  # noqa: E402,F401
                                 same_partition)


# ---------------------------------------------------------------------------------------------
# Python-variant restatement (utils/segmenter.py semantics, float64)


class PySegmenterError(Exception):
    """Raised where the reference Python raises (NameError / KeyError / AssertionError)."""


def run_pysegmenter(class_probs, same_probs, num_classes: int, offsets: Sequence[Tuple[int, int]],
                    same_different_bias: float = 0.0, object_merge_factor: float = 1.0,
                    merge_logprob_bias: float = 0.0, prune_threshold: float = 200.0) -> OracleResult:
    lib = _load("libpysegmenter_oracle.so")
    fn = lib.oracle_pysegmenter_run
    fn.restype = ctypes.c_int
    i64p = ctypes.POINTER(ctypes.c_longlong)
    fn.argtypes = [_f64p, _f64p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p,
                   ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                   i64p, _i32p, _i32p, _f64p]
    cp, sp, off = _prep(class_probs, same_probs, offsets, dtype=np.float64)
    C, H, W = cp.shape
    if C != num_classes:
        raise AssertionError("class_dim == num_classes (utils/segmenter.py:247)")
    mask = np.zeros((H, W), np.int64)
    table = np.zeros(H * W, np.int32)
    part = np.zeros((H, W), np.int32)
    stats = np.zeros(6, np.float64)
    rc = fn(cp.ctypes.data_as(_f64p), sp.ctypes.data_as(_f64p), int(num_classes), sp.shape[0],
            H, W, off.ctypes.data_as(_i32p), float(same_different_bias),
            float(object_merge_factor), float(merge_logprob_bias), float(prune_threshold),
            mask.ctypes.data_as(i64p), table.ctypes.data_as(_i32p), part.ctypes.data_as(_i32p),
            stats.ctypes.data_as(_f64p))
    if rc == -10:
        raise PySegmenterError("NameError: no class-0 object to prune into (segmenter.py:356,365)")
    if rc != 0:
        raise PySegmenterError("oracle_pysegmenter_run failed with code %d" % rc)
    classes = []
    for v in table:
        if v == -1:
            break
        classes.append(int(v))
    names = ["total_logprob", "n_objects", "n_pops", "n_merges", "n_initial_records", "_"]
    return OracleResult(mask, classes, part, float(stats[0]), dict(zip(names, stats.tolist())))
