"""The fp64 logarithm of the MI evaluation (ldweaver_amd/csrc/ldw_log.h) checked on the HOST: the header is one source for the HIP kernels and for
tests/host/log_check.cpp, which runs its fold + polynomial + Newton step with a modelled reciprocal estimate (1 / x off by up to 4.5e-8, v_rcp_f64's
accuracy) against the long-double logarithm.  Asserts what the header states: |s| <= 0.2006 after the integer fold and a relative error below 5e-15
(the squared reciprocal error: 2e-15; the polynomial's truncation is 3e-17).  No GPU, no oracle: g++ only."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_log_header_on_the_host(tmp_path):
    exe = str(tmp_path / "log_check")
    subprocess.run(["g++", "-O2", "-mfma", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "log_check.cpp")], check=True, timeout=300)
    out = subprocess.run([exe, "4000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"largest \|s\| ([0-9.]+)\s+largest \|error\| ([0-9.e+-]+)\s+largest relative error \(\|log\| > 1e-3\) ([0-9.e+-]+)", out.stdout)
    assert m, out.stdout
    smax, eabs, erel = float(m.group(1)), float(m.group(2)), float(m.group(3))
    assert 0.19 < smax <= 0.2006      # the fold is exercised up to its bound and not beyond
    assert erel < 5e-15 and eabs < 1e-13
