"""GPU: the C ABI without PyTorch — tests/cabi/raster_cabi_smoke.cpp is compiled against include/*.h, linked to libsurfel_hip.so and
run as a plain process (hipMalloc memory, malloc-style allocator callbacks, NULL stream)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_roundtrip_without_torch(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = shutil.which("hipcc")
    assert hipcc, "hipcc not found"
    libdir = os.path.join(REPO, "2d-gaussian-splatting_amd", "lib")
    exe = str(tmp_path / "raster_cabi_smoke")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(REPO, "tests", "cabi", "raster_cabi_smoke.cpp"),
                           "-L" + libdir, "-lsurfel_hip", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cabi smoke ok" in out.stdout, out.stdout
