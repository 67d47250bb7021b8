"""mn_reforder.h -- the reference's order among bit-equal priorities (std::priority_queue + std::unordered_map of
libstdc++, restated on flat arrays) -- HOST build of the text the GPU runs (tests/tools/reforder_check.cpp):

* the containers against the standard library itself, operation by operation (iteration order of the map after
  every insert / erase, bucket counts through its growth, top of the heap after every push / pop with ties
  everywhere);
* the whole merge against the reference's own outputs (tests/golden), the tie-DECIDED vectors included.

CPU only.  The GPU tests (tests/test_gpu_exact.py) then pin the device build of the same header.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu
from mergenet_amd import labels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("reforder") / "libreforder_host.so")
    src = os.path.join(ROOT, "tests", "tools", "reforder_check.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", src, "-o", so], check=True)
    lib = ctypes.CDLL(so)
    f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    lib.reforder_containers_check.argtypes = [ctypes.c_ulonglong, ctypes.c_int, ctypes.c_int]
    lib.reforder_host_run.argtypes = [f32p, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, i32p, i32p,
                                      ctypes.POINTER(ctypes.c_longlong)]
    return lib


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_flat_array_containers_behave_as_libstdcxx(host_lib, seed):
    assert host_lib.reforder_containers_check(seed, 40000, 1) == 0


def test_flat_array_map_through_its_growth(host_lib):
    """400 000 operations: the map passes 13, 29, 59 ... 85 229 buckets, rehashing each time (bucket count checked
    at every step, the whole iteration order at every 499th)."""
    assert host_lib.reforder_containers_check(99, 400000, 499) == 0


def _host_run(lib, g):
    f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
    eps = np.float32(2.0 ** -23)
    cp = np.ascontiguousarray(np.clip(np.asarray(g["class_probs"], dtype=np.float32), eps, np.float32(1) - eps))
    sp = np.ascontiguousarray(np.clip(np.asarray(g["sameness_probs"], dtype=np.float32), eps, np.float32(1) - eps))
    offs = np.ascontiguousarray(np.asarray(g["offsets"], dtype=np.int32).reshape(-1, 2))
    part = np.zeros(H * W, np.int32)
    cls = np.zeros(H * W, np.int32)
    st = (ctypes.c_longlong * 8)()
    sdb, omf, bias = g["spec"]["opts"]
    rc = lib.reforder_host_run(cp.ctypes.data_as(f32p), C, sp.ctypes.data_as(f32p), len(offs), W, H,
                               offs.ctypes.data_as(i32p), sdb, omf, bias, part.ctypes.data_as(i32p),
                               cls.ctypes.data_as(i32p), st)
    assert rc == 0, rc
    mask = np.where(cls.reshape(H, W) == 0, 0, part.reshape(H, W) + 1)       # class-0 objects are background
    return mask, cls.reshape(H, W), int(st[0]), int(st[1])


BIG = ("1024x2048", "800x1333", "512x1024", "256x512", "400x667")     # (seconds each on the host: left to the GPU tests)
SMALL = [n for n in gu.names("cseg_") if not any(b in n for b in BIG)]


@pytest.mark.parametrize("name", SMALL)
def test_host_build_reproduces_the_reference(host_lib, name):
    """Partition, background set and per-instance class of the reference's own output -- also on
    cseg_blur4_*, where the order among bit-equal priorities decides instance borders."""
    g = gu.load(name)
    mask, cls, pops, merges = _host_run(host_lib, g)
    ref = np.asarray(g["mask"])
    assert np.array_equal(mask == 0, ref == 0)
    assert labels.same_partition(mask, ref)
    # the class of every instance
    ref_classes = list(g["object_class"])
    for lab in np.unique(ref):
        if lab == 0:
            continue
        ys, xs = np.nonzero(ref == lab)
        assert cls[ys[0], xs[0]] == ref_classes[int(lab) - 1]
    assert pops >= merges > 0
