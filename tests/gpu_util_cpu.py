"""Comparators of the parity tests that need neither torch nor a GPU."""
import numpy as np


def rel_err_elementwise(got, ref):
    """max |got - ref| / |ref| over the elements -- the reference's own measure (misc/gemm_run.sh:23:
    `np.max(np.divide(np.abs(a-b), b))`).  Meaningful where the operands are sign-definite (no
    cancellation: every |ref| is of the size of its sum of products); elements with ref == 0 must match
    exactly."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    z = ref == 0
    if z.any() and not np.array_equal(got[z], ref[z]):
        return float("inf")
    nz = ~z
    return float((np.abs(got[nz] - ref[nz]) / np.abs(ref[nz])).max()) if nz.any() else 0.0


def special_mismatch(got, ref, tol=1e-4):
    """Compares results that may hold NaN / Inf / signed zeros / denormals: NaN at the same places, infinities
    equal, the finite rest within tol of the largest finite |ref| (random-sign data: norm-wise).
    Returns None when they agree, else a description."""
    got = np.asarray(got)
    ref = np.asarray(ref)
    if got.shape != ref.shape:
        return f"shape {got.shape} != {ref.shape}"
    gn, rn = np.isnan(got), np.isnan(ref)
    if not np.array_equal(gn, rn):
        return f"NaN pattern differs at {int((gn != rn).sum())} places, first {np.argwhere(gn != rn)[0].tolist()}"
    gi, ri = np.isinf(got), np.isinf(ref)
    if not np.array_equal(gi, ri) or not np.array_equal(got[ri], ref[ri]):
        return "infinities differ"
    fin = ~(rn | ri)
    if fin.any():
        g64, r64 = got[fin].astype(np.float64), ref[fin].astype(np.float64)
        scale = np.abs(r64).max()
        err = np.abs(g64 - r64).max()
        if err > tol * scale and err > 0:
            return f"finite part: max |diff| {err:.3e} against scale {scale:.3e}"
    return None


def bits_equal_nan_aware(a, b):
    """bit-for-bit equality where every NaN counts as equal to every NaN (payload / sign of a NaN is not
    part of the contract; the sign of a zero is)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.shape != b.shape:
        return False
    an, bn = np.isnan(a), np.isnan(b)
    if not np.array_equal(an, bn):
        return False
    return np.array_equal(a.view(np.uint32)[~an], b.view(np.uint32)[~bn])
