"""The one-wave FFT core of csrc/spectral.hip compiled for the HOST (its functions are __host__ __device__) and run lane by
lane against a double-precision DFT: tests/native/fft_core_host.cpp.  No GPU involved; hipcc only as the compiler."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fft_core_on_the_host(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    exe = tmp_path / "fft_core_host"
    subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(REPO, "include"),
                    os.path.join(REPO, "tests/native/fft_core_host.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.count("rel_err") == 8, out
