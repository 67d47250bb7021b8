"""ctypes front end of tests/tools/exact_model.cpp (CPU model of the exact engine's semantics: the reference's
lazy greedy with ties going to the lowest record id).  TEST / DESIGN TOOL."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libexact_model.so")


def build():
    src = os.path.join(_HERE, "exact_model.cpp")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", _SO])
    return ctypes.CDLL(_SO)


def run(class_probs, sameness_probs, offsets, omf, bias, clip=True, track=True):
    """-> (partition int32[H,W] of surviving object ids, class per pixel, stats dict)"""
    lib = build()
    cp = np.ascontiguousarray(class_probs, np.float32)
    sp = np.ascontiguousarray(sameness_probs, np.float32)
    if clip:
        eps = np.finfo(np.float32).eps
        cp = np.clip(cp, eps, 1 - eps)
        sp = np.clip(sp, eps, 1 - eps)
    C, H, W = cp.shape
    O = sp.shape[0]
    offs = np.ascontiguousarray(np.asarray(offsets, np.int32).reshape(-1))
    part = np.zeros((H, W), np.int32)
    ocls = np.zeros((H, W), np.int32)
    stats = np.zeros(12, np.float64)
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int)
    lib.exact_model_run.restype = ctypes.c_int
    lib.exact_model_run(cp.ctypes.data_as(fp), sp.ctypes.data_as(fp), C, O, W, H, offs.ctypes.data_as(ip),
                        ctypes.c_float(omf), ctypes.c_float(bias), int(track), part.ctypes.data_as(ip),
                        ocls.ctypes.data_as(ip), stats.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    names = ("steps", "merges", "tied_steps", "tied_merges", "tied_conflicts", "max_depth", "arena_entries", "reallocs", "repop_merges",
             "front_errors", "front_rebuilds", "executor_rounds")
    return part, ocls, {k: int(v) for k, v in zip(names, stats)}
