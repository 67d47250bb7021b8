"""mn_ref_logf (mergenet_amd/csrc/mn_ref_logf.h) is glibc's logf bit for bit: the host build of the same
header against the C library on a sample of the clipped input range (the GPU tests then pin the device
build through the exact engine's phase-A arrays).  CPU only; ~2 s."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ref_logf_equals_the_c_librarys_logf(tmp_path):
    exe = str(tmp_path / "ref_logf_check")
    src = os.path.join(ROOT, "tests", "tools", "ref_logf_check.c")
    # -ffp-contract=off: every fused multiply-add of the routine is written out (__builtin_fma)
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", src, "-o", exe, "-lm"], check=True)
    res = subprocess.run([exe, "61"], capture_output=True, text=True)
    sys.stdout.write(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "mismatches 0" in res.stdout


def test_ref_expf_equals_the_c_librarys_expf(tmp_path):
    """mn_ref_expf (same header): glibc's expf bit for bit, on every 13th float of 2^-30 <= |x| <= 64 and every
    float of 8 <= |x| <= 18 (the reference's same_different_bias path, segment.cc:183-195)."""
    exe = str(tmp_path / "ref_expf_check")
    src = os.path.join(ROOT, "tests", "tools", "ref_expf_check.c")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", src, "-o", exe, "-lm"], check=True)
    res = subprocess.run([exe, "13"], capture_output=True, text=True)
    sys.stdout.write(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "mismatches 0" in res.stdout
