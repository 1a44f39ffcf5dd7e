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
    for d in ("gemm_driver", "csrmm_driver", "csrgemv_driver", "csrcsc_driver", "kmeans_driver"):
        assert os.access(os.path.join(PKG, "bin", d), os.X_OK)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
@pytest.mark.parametrize("drv", ["gemm", "csrmm", "csrgemv", "csrcsc", "csrmm_pmem", "kmeans"])
def test_reference_driver_compiles_unchanged(tmp_path, drv):
    subprocess.run(["make", "-C", os.path.join(PKG, "drivers"), "-s"], check=True)
    out = str(tmp_path / f"ref_{drv}")
    cmd = ["g++", "-std=c++14", "-O1", "-w", "-fopenmp", "-I", os.path.join(PKG, "include"),
           os.path.join(REF, "drivers", f"{drv}.cpp"), "-o", out,
           "-L", os.path.join(PKG, "lib"), "-lflashblas", "-lbof_hip",
           f"-Wl,-rpath,{os.path.join(PKG, 'lib')}", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.access(out, os.X_OK)


def test_generator_tools_reproduce_reference_kats(tmp_path):
    """blas-on-flash_amd/bin/{sparse,dense}_create write byte-identical files to the
    reference's misc tools (known-answer hashes)."""
    import hashlib
    subprocess.run(["make", "-C", os.path.join(PKG, "drivers"), "-s"], check=True)
    b = os.path.join(PKG, "bin")
    subprocess.run([os.path.join(b, "sparse_create"), str(tmp_path / "k."), "1000", "100000", "0.0001"], check=True)
    subprocess.run([os.path.join(b, "dense_create"), str(tmp_path / "d.bin"), "37", "53", "s"], check=True)
    h = lambda f: hashlib.sha256(open(tmp_path / f, "rb").read()).hexdigest()[:16]
    assert (h("k.csr"), h("k.col"), h("k.off")) == ("4ab90aabd38c5029", "df0b4409a7318e45", "f0510c987daf1cfe")
    assert h("d.bin") == "d08eb5a3728513a6"
    assert open(tmp_path / "k.info").read().split()[:2] == ["1000", "100000"]
