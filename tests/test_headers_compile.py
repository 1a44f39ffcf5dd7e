"""The drop-in boundary: the reference's own hot-path drivers compile UNCHANGED
against blas-on-flash_amd/include and link against our libraries (only checked
where the reference tree exists); our drivers with the identical argv build
everywhere."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "blas-on-flash_amd")
REF = "/root/reference"


def test_own_drivers_build():
    subprocess.run(["make", "-C", os.path.join(PKG, "drivers"), "-s"], check=True)
    for d in ("gemm_driver", "csrmm_driver", "csrgemv_driver"):
        assert os.access(os.path.join(PKG, "bin", d), os.X_OK)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
@pytest.mark.parametrize("drv", ["gemm", "csrmm", "csrgemv"])
def test_reference_driver_compiles_unchanged(tmp_path, drv):
    subprocess.run(["make", "-C", os.path.join(PKG, "drivers"), "-s"], check=True)
    out = str(tmp_path / f"ref_{drv}")
    cmd = ["g++", "-std=c++14", "-O1", "-w", "-I", os.path.join(PKG, "include"),
           os.path.join(REF, "drivers", f"{drv}.cpp"), "-o", out,
           "-L", os.path.join(PKG, "lib"), "-lflashblas", "-lbof_hip",
           f"-Wl,-rpath,{os.path.join(PKG, 'lib')}", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.access(out, os.X_OK)
