"""Helpers for the -m gpu parity tests: torch is only plumbing (device buffers)."""
import numpy as np
import torch


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ptr(t):
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def rel_err(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    d = np.abs(ref).max()
    return float(np.abs(got - ref).max() / (d if d > 0 else 1.0))


from gpu_util_cpu import bits_equal_nan_aware, rel_err_elementwise, special_mismatch  # noqa: E402,F401
