"""Property test of the file primitives behind FlashFileHandle::read/write/sread/swrite
(bof_file_sread / bof_file_swrite, src/file_handles/flash_file_handle.cpp:247-716 in the reference):
random strided regions -- aligned and unaligned offsets, strides, lengths and buffers, O_DIRECT and
buffered descriptors, kernel AIO and io_uring engines, small request sizes so transfers are cut
into many requests -- against a numpy image of the file.  CPU only."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import bofhip

FILE_BYTES = 1 << 20


def _open(path, direct):
    if direct:
        try:
            return os.open(path, os.O_RDWR | os.O_DIRECT)
        except OSError:
            pass
    return os.open(path, os.O_RDWR)


@st.composite
def regions(draw):
    unit = draw(st.sampled_from([1, 4, 512, 4096]))            # granularity of every quantity
    n_strides = draw(st.integers(1, 24))
    length = draw(st.integers(1, max(1, 16384 // unit))) * unit
    stride = length + draw(st.integers(0, max(1, 8192 // unit))) * unit
    span = (n_strides - 1) * stride + length
    if span > FILE_BYTES // 2:
        n_strides = 1
        span = length
    offset = draw(st.integers(0, (FILE_BYTES - span) // unit)) * unit
    buf_shift = draw(st.sampled_from([0, 0, 4, 64, 512]))      # misalign the memory side sometimes
    return offset, stride, n_strides, length, buf_shift


def _run_property(tmp_path, direct):
    L = bofhip.lib()
    path = str(tmp_path / f"prop_{int(direct)}.bin")
    rng = np.random.default_rng(5)
    image = rng.integers(0, 256, FILE_BYTES, dtype=np.uint8)
    image.tofile(path)
    fd = _open(path, direct)
    L.bof_file_set_request_bytes(8192)          # many requests per transfer
    scratch = np.zeros(FILE_BYTES + 8192, np.uint8)
    base = scratch.ctypes.data
    base += (-base) % 4096                       # page-aligned origin inside the array
    origin = base - scratch.ctypes.data

    @settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(regions(), st.booleans())
    def prop(region, write):
        offset, stride, n, length, shift = region
        packed = slice(origin + shift, origin + shift + n * length)
        idx = (offset + np.arange(n)[:, None] * stride + np.arange(length)[None, :]).ravel()
        if write:
            data = rng.integers(0, 256, n * length, dtype=np.uint8)
            scratch[packed] = data
            rc = L.bof_file_swrite(fd, offset, stride, n, length, base + shift, 1)
            assert rc == 0, L.bof_last_error()
            image[idx] = data
            os.fsync(fd)
            assert np.array_equal(np.fromfile(path, np.uint8), image)      # nothing else touched
        else:
            scratch[packed] = 0
            rc = L.bof_file_sread(fd, offset, stride, n, length, base + shift, 1)
            assert rc == 0, L.bof_last_error()
            assert np.array_equal(scratch[packed], image[idx])

    try:
        prop()
    finally:
        L.bof_file_set_request_bytes(4 << 20)
        L.bof_file_forget(fd)
        os.close(fd)


@pytest.mark.parametrize("direct", [True, False])
def test_strided_io_random_regions_aio(tmp_path, direct):
    _run_property(tmp_path, direct)


def test_strided_io_random_regions_io_uring(tmp_path):
    """the same property in a child process with BOF_IO_ENGINE=uring (read once per process)"""
    if os.environ.get("BOF_IO_ENGINE") == "uring":
        _run_property(tmp_path, True)
        return
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k", "io_uring"],
                       env=dict(os.environ, BOF_IO_ENGINE="uring"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
