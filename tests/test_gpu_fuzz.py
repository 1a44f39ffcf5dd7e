"""-m gpu randomized differential test of level 3 (files -> pipelines -> files) against the oracle.

The parametrized suites pin named cases; this draws the cases: random shapes (tails merged and not,
empty-ish edges), layouts, leading dimensions (aligned, unaligned, padded), file offsets, alpha / beta,
and -- the point -- random COMBINATIONS of the knobs that pick the code path: tile cache / row panels,
k-major panel copies, ramp group, row-group width of the tile cache, chunk size, HBM budget, thread and
stream counts, O_DIRECT / buffered, AIO / io_uring, widened O_DIRECT / buffered twin for unaligned files,
one device or a device list with repeats.  Every result must equal the oracle's bit for bit, and every
byte outside the result's extents (padding columns, header, trailer) must be untouched.

pytest runs a short fixed-seed batch; `python tests/test_gpu_fuzz.py --seconds 600 --seed 7` runs a long one
and prints the failing case's parameters (reproduce with --seed S --only I)."""
import os
import sys
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "blas-on-flash_amd"))
import bofhip  # noqa: E402
import orc  # noqa: E402

pytestmark = pytest.mark.gpu

ENV_KNOBS = ("BOF_TILE_GROUP", "BOF_UNALIGNED_DIRECT", "BOF_MMAP_WRITES", "BOF_PANEL_SLICES", "BOF_PANEL_SLICE_ROWS", "BOF_PANEL_SLICES_ALL", "BOF_PANEL_RAMP_K")
LAST = {}      # the parameters of the case being run (printed when it fails)
VERIFY_SUMS = [0]   # BOF_VERIFY: hand-over sums compared so far
FORCE_KIND = ""   # --kind: only gemm / kmeans / csr cases
OVERRIDE = {}  # --set: options forced on top of the drawn ones (bisecting a failing case)


def stored_shapes(ord_, ta, tb, m, n, k):
    a = (m, k) if (ta == "T") == (ord_ == "C") else (k, m)
    b = (k, n) if (tb == "T") == (ord_ == "C") else (n, k)
    c = (m, n) if ord_ == "R" else (n, m)
    return a, b, c


class BinFile:
    """header | payload | trailer in one file; payload = a (rows, ld) float32 / int64 image"""

    def __init__(self, path, payload, rng, head=0, tail=0, direct=True):
        self.path, self.head = path, head
        self.hbytes = rng.integers(0, 255, head, dtype=np.uint8).tobytes()
        self.tbytes = rng.integers(0, 255, tail, dtype=np.uint8).tobytes()
        self.payload = np.ascontiguousarray(payload)
        with open(path, "wb") as f:
            f.write(self.hbytes)
            f.write(self.payload.tobytes())
            f.write(self.tbytes)
        self.fd = -1
        if direct:
            try:
                self.fd = os.open(path, os.O_RDWR | os.O_DIRECT)
            except OSError:
                pass
        if self.fd < 0:
            self.fd = os.open(path, os.O_RDWR)

    def fptr(self):
        return bofhip.FPtr(self.fd, self.head)

    def check_against(self, expect):
        """whole file: header, the payload image `expect`, trailer"""
        raw = open(self.path, "rb").read()
        n = self.payload.nbytes
        assert raw[:self.head] == self.hbytes, "header bytes changed"
        assert raw[self.head + n:] == self.tbytes, "trailer bytes changed (or file size changed)"
        got = np.frombuffer(raw[self.head:self.head + n], self.payload.dtype).reshape(self.payload.shape)
        if not np.array_equal(got, expect):
            bad = np.argwhere(got != expect)
            raise AssertionError(f"payload differs at {len(bad)} elements, first {bad[0].tolist()}: "
                                 f"got {got[tuple(bad[0])]} want {expect[tuple(bad[0])]}")

    def close(self):
        bofhip.lib().bof_file_forget(self.fd)      # mandatory before close (include/bof_hip.h)
        os.close(self.fd)


def pick(rng, xs):
    return xs[int(rng.integers(0, len(xs)))]


def common_opts(rng, kw):
    kw["n_io_threads"] = int(rng.integers(1, 7))
    kw["n_streams"] = int(rng.integers(1, 5))
    kw["pinned_slots"] = int(rng.integers(2, 7))
    kw["use_odirect"] = int(rng.integers(0, 2))
    kw["io_engine"] = pick(rng, [0, 0, 1, 2])
    dev = pick(rng, [None, None, [0, 0], [0, 0, 0], [0]])
    if dev is not None:
        kw["devices"] = dev
        if len(dev) > 1:       # the shared operand device to device (row panels; ignored elsewhere)
            kw["peer_bcast"] = pick(rng, [0, 1, 1, 2])
    for k, v in OVERRIDE.items():
        if v is None:
            kw.pop(k, None)
        else:
            kw[k] = v
    env = {"BOF_TILE_GROUP": pick(rng, ["", "1", "3", "16"]),
           "BOF_UNALIGNED_DIRECT": pick(rng, ["", "0", "1"]),
           "BOF_MMAP_WRITES": pick(rng, ["", "0", "1"]),
           # row slices of the whole-K panel launches (round 6): off, the default, 2-3 slices of small panels
           "BOF_PANEL_SLICES": pick(rng, ["", "1", "2", "3"]),
           "BOF_PANEL_SLICE_ROWS": pick(rng, ["", "32", "64"]),
           "BOF_PANEL_SLICES_ALL": pick(rng, ["", "1"]),
           "BOF_PANEL_RAMP_K": pick(rng, ["", "", "2", "3"])}
    return env


def apply_env(env):
    for k in ENV_KNOBS:
        if env.get(k):
            os.environ[k] = env[k]
        else:
            os.environ.pop(k, None)


def dim(rng, blk):
    """a dimension around a few tiles: exact multiples, separate tails (>= 128), merged tails (< 128), tiny"""
    kind = int(rng.integers(0, 5))
    nb = int(rng.integers(1, 4))
    if kind == 0:
        return nb * blk
    if kind == 1:
        return nb * blk + int(rng.integers(128, blk)) if blk > 128 else nb * blk + 128
    if kind == 2:
        return nb * blk + int(rng.integers(1, 128))
    if kind == 3:
        return int(rng.integers(1, 200))
    return int(rng.integers(1, 3 * blk + 100))


def ld_of(rng, cols):
    kind = int(rng.integers(0, 4))
    if kind == 0:
        return cols
    if kind == 1:
        return cols + int(rng.integers(1, 40))
    if kind == 2:
        return (cols + 127) // 128 * 128          # rows sector aligned
    return (cols + 1023) // 1024 * 1024           # rows page aligned


def gemm_case(rng, tmp, kmeans=False):
    blk = pick(rng, [128, 256, 256])
    ord_, ta, tb = pick(rng, "RC"), pick(rng, "NT"), pick(rng, "NT")
    m, n, k = dim(rng, blk), dim(rng, blk), dim(rng, blk)
    alpha, beta = pick(rng, [(1.0, 0.0), (0.5, 2.0), (-1.0, 1.0), (2.0, 0.0)])
    if kmeans:
        alpha, beta = -2.0, 0.0
    shp = stored_shapes(ord_, ta, tb, m, n, k)
    lds = [ld_of(rng, s[1]) for s in shp]
    mats = [rng.uniform(-1, 1, (s[0], ld)).astype(np.float32) for s, ld in zip(shp, lds)]
    heads = [pick(rng, [0, 0, 512, 4096, 52, 1000]) for _ in range(3)]
    kw = {"gemm_blk": blk, "gemm_path": pick(rng, [0, 1, 2]), "io_chunk_mib": pick(rng, [1, 1, 2, 32]),
          "panel_kmajor": pick(rng, [0, 1, 2, 3]), "panel_group": pick(rng, [0, 0, 1, 2, 3]),
          "panel_streams": pick(rng, [0, 0, 1, 3]), "panel_writers": pick(rng, [0, 0, 1, 3])}
    if int(rng.integers(0, 4)) == 0:     # a budget of a few tiles: the tile cache under pressure / small panel rings
        kw["hbm_budget"] = int(rng.integers(6, 40)) * blk * blk * 4
    if not kmeans:       # the default arithmetic (one chain over the whole K) or the reference's (one rounding per k-block)
        kw["gemm_chain"] = pick(rng, [0, 0, 1])
    env = common_opts(rng, kw)
    desc = dict(kind="kmeans" if kmeans else "gemm", ord=ord_, ta=ta, tb=tb, m=m, n=n, k=k, alpha=alpha, beta=beta,
                lds=lds, heads=heads, opts=kw, env=env)
    LAST.clear(); LAST.update(desc)
    ref = mats[2].copy()
    if kmeans:
        cl = rng.uniform(0, 8, m).astype(np.float32)
        pl = rng.uniform(0, 8, n).astype(np.float32)
        ones = np.ones(max(m, n), np.float32)
        orc.flash_kmeans(ord_, ta, tb, m, n, k, alpha, beta, mats[0], mats[1], ref, lds[0], lds[1], lds[2], blk, cl, pl, ones)
    else:
        orc.flash_gemm(ord_, ta, tb, m, n, k, alpha, beta, mats[0], mats[1], ref, lds[0], lds[1], lds[2], blk,
                       chain=kw.get("gemm_chain", 0))
    direct = bool(rng.integers(0, 4))
    files = [BinFile(os.path.join(tmp, f"{nm}.bin"), x, rng, head=h, tail=int(pick(rng, [0, 0, 777, 4096])), direct=direct)
             for nm, x, h in zip("abc", mats, heads)]
    try:
        apply_env(env)
        opts = bofhip.default_options(**kw)
        args = (ord_, ta, tb, m, n, k, alpha, beta, files[0].fptr(), files[1].fptr(), files[2].fptr(), lds[0], lds[1], lds[2])
        try:
            if kmeans:
                bofhip.flash_kmeans(*args, cl.ctypes.data, pl.ctypes.data, ones.ctypes.data, opts)
            else:
                bofhip.flash_gemm(*args, opts)
        except bofhip.BofError as e:
            # a drawn budget may really be below one task's working set (merged tail tiles are up to 4x a
            # plain tile): the documented BOF_ENOMEM, and C must be untouched
            # ... and gemm_path = 2 is refused (BOF_EINVAL) when the panel pipeline cannot take the call
            if ("hbm_budget" in kw and "budget" in str(e)) or (kw["gemm_path"] == 2 and "not eligible" in str(e)):
                files[2].check_against(mats[2])
                desc["outcome"] = str(e)
                return desc
            raise
        try:
            files[2].check_against(ref)
        except AssertionError:
            if os.environ.get("BOF_FUZZ_DUMP"):      # everything needed to look at the failure offline
                got = np.fromfile(files[2].path, np.uint8)
                # what the library did during the failing call, hand-over by hand-over (the always-on event ring)
                bofhip.lib().bof_event_dump(os.path.join(os.environ["BOF_FUZZ_DUMP"],
                                                         f"fuzz_fail_{os.getpid()}_{int(time.time())}.events.txt").encode())
                # the same call once more on a restored C: does the mismatch come back?
                with open(files[2].path, "r+b") as f:
                    f.seek(files[2].head)
                    f.write(mats[2].tobytes())
                again = "not run"
                try:
                    if kmeans:
                        bofhip.flash_kmeans(*args, cl.ctypes.data, pl.ctypes.data, ones.ctypes.data, opts)
                    else:
                        bofhip.flash_gemm(*args, opts)
                    files[2].check_against(ref)
                    again = "second attempt matches"
                except Exception as e2:
                    again = f"second attempt: {e2}"
                print("     ", again)
                np.savez_compressed(os.path.join(os.environ["BOF_FUZZ_DUMP"], f"fuzz_fail_{os.getpid()}_{int(time.time())}.npz"),
                                    a=mats[0], b=mats[1], c0=mats[2], ref=ref, got_file=got, desc=repr(desc), again=again,
                                    **({"cl": cl, "pl": pl} if kmeans else {}))
            raise
        files[0].check_against(mats[0])
        files[1].check_against(mats[1])
        st = bofhip.flash_last_stats()
        assert st["bytes_written"] <= mats[2].nbytes, st        # C leaves once, whatever the path
        VERIFY_SUMS[0] += st["verify_checks"]
    finally:
        for f in files:
            f.close()
    return desc


def random_csr(rng, m, n):
    """columns sorted and unique within a row; rows of 0 .. 24 entries, a few heavy rows; small integer values"""
    nnz_row = rng.integers(0, min(n, 24) + 1, m)
    if m > 4 and n > 200:
        nnz_row[rng.integers(0, m, 2)] = min(n, 200)
    ia = np.zeros(m + 1, np.int64)
    np.cumsum(nnz_row, out=ia[1:])
    ja = np.empty(int(ia[-1]), np.int64)
    for r in range(m):
        ja[ia[r]:ia[r + 1]] = np.sort(rng.choice(n, int(nnz_row[r]), replace=False))
    val = rng.integers(1, 10, int(ia[-1])).astype(np.float32)
    return val, ja, ia


def csr_case(rng, tmp):
    m, n = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
    val, ja, ia = random_csr(rng, m, n)
    if val.size == 0:
        val, ja = np.zeros(1, np.float32), np.zeros(1, np.int64)       # files cannot be empty; nnz stays 0
    kind = pick(rng, ["csrmm", "csrmm", "csrgemv_N", "csrgemv_T", "csrmm_T", "csrcsc"])
    kw = {"max_nnzs": int(pick(rng, [500, 5000, 10_000_000])), "csrmm_rblk": int(pick(rng, [128, 300, 1000, 131072])),
          "csrmm_cblk": int(pick(rng, [64, 1024]))}
    env = common_opts(rng, kw)
    desc = dict(kind=kind, m=m, n=n, nnz=int(ia[-1]), opts=kw, env=env)
    LAST.clear(); LAST.update(desc)
    direct = bool(rng.integers(0, 4))
    fv = BinFile(os.path.join(tmp, "val.bin"), val, rng, direct=direct)
    fj = BinFile(os.path.join(tmp, "ja.bin"), ja, rng, direct=direct)
    fi = BinFile(os.path.join(tmp, "ia.bin"), ia, rng, direct=direct)
    files = [fv, fj, fi]
    try:
        apply_env(env)
        opts = bofhip.default_options(**kw)
        if kind in ("csrmm", "csrmm_T"):
            k = int(pick(rng, [1, 7, 64, 128, 130, 200]))
            ord_b = pick(rng, "RC") if kind == "csrmm" else "R"
            alpha, beta = pick(rng, [(1.0, 0.0), (0.5, 2.0), (2.0, 1.0)])
            rows_b, rows_c = (n, m) if kind == "csrmm" else (m, n)
            b = rng.integers(0, 7, (rows_b, k)).astype(np.float32)
            c0 = rng.integers(0, 5, (rows_c, k)).astype(np.float32)
            desc.update(k=k, ord_b=ord_b, alpha=alpha, beta=beta)
            if kind == "csrmm":
                if ord_b == "C":
                    b, c0 = np.ascontiguousarray(b.T), np.ascontiguousarray(c0.T)
                ref = orc.flash_csrmm(ord_b, m, n, k, alpha, beta, val, ia, ja, b, c0.copy(), kw["csrmm_rblk"], kw["max_nnzs"],
                                      kw["csrmm_cblk"])
            else:
                ref = orc.scsrmm_t(m, n, k, alpha, val, ia, ja, b, k, beta, c0.copy(), k)
            fb = BinFile(os.path.join(tmp, "b.bin"), b, rng, direct=direct)
            fc = BinFile(os.path.join(tmp, "c.bin"), c0, rng, tail=int(pick(rng, [0, 777])), direct=direct)
            files += [fb, fc]
            bofhip.flash_csrmm("T" if kind == "csrmm_T" else "N", m, n, k, alpha, beta, fv.fptr(), fi.fptr(), fj.fptr(), ord_b,
                               fb.fptr(), fc.fptr(), opts)
            try:
                fc.check_against(ref)
            except AssertionError:
                if os.environ.get("BOF_FUZZ_DUMP"):      # everything needed to look at the failure offline
                    raw = open(fc.path, "rb").read()
                    got = np.frombuffer(raw[fc.head:fc.head + fc.payload.nbytes], np.float32).reshape(fc.payload.shape)
                    cfile = bofhip.flash_last_c_file()
                    np.savez_compressed(os.path.join(os.environ["BOF_FUZZ_DUMP"], f"fuzz_csr_fail_{os.getpid()}_{int(time.time())}.npz"),
                                        got=got, ref=ref, c0=c0, ia=ia, ja=ja, val=val, b=b,
                                        desc=np.array(repr(dict(desc, c_file_mode=cfile[0], c_twin_bytes=cfile[1],
                                                                stats=bofhip.flash_last_stats()))))
                raise
            fb.check_against(b)
        elif kind == "csrcsc":
            vt, it, jt = orc.csrcsc(m, n, val, ia, ja)
            nnz = int(ia[-1])
            fvt = BinFile(os.path.join(tmp, "vt.bin"), np.zeros(max(nnz, 1), np.float32), rng, direct=direct)
            fjt = BinFile(os.path.join(tmp, "jt.bin"), np.zeros(max(nnz, 1), np.int64), rng, direct=direct)
            fit = BinFile(os.path.join(tmp, "it.bin"), np.zeros(n + 1, np.int64), rng, direct=direct)
            files += [fvt, fjt, fit]
            bofhip.flash_csrcsc(m, n, fi.fptr(), fj.fptr(), fv.fptr(), fit.fptr(), fjt.fptr(), fvt.fptr(), opts)
            fit.check_against(it)
            if nnz:
                fjt.check_against(jt)
                fvt.check_against(vt)
        else:
            trans = kind[-1]
            x = rng.integers(0, 10, m if trans == "T" else n).astype(np.float32)
            y = np.full(n if trans == "T" else m, -7.0, np.float32)
            ref = orc.flash_csrgemv(trans, m, n, val, ia, ja, x, np.zeros_like(y), kw["csrmm_rblk"], kw["max_nnzs"])
            bofhip.flash_csrgemv(trans, m, n, fv.fptr(), fi.fptr(), fj.fptr(), x.ctypes.data, y.ctypes.data, opts)
            assert np.array_equal(y, ref), f"y differs at {int((y != ref).sum())} elements"
        fv.check_against(val)
        fj.check_against(ja)
        fi.check_against(ia)
    finally:
        for f in files:
            f.close()
    return desc


def kernel_case(rng):
    """Level 1: one bof_sgemm / bof_skmeans_task on HBM-resident operands, shapes that mix the 256 x 256
    kernels' interior with ragged strips, K on and off the 64-slab grid, tight / padded leading dimensions."""
    import torch
    ord_, ta, tb = pick(rng, "RC"), pick(rng, "NT"), pick(rng, "NT")

    def edge():
        kind = int(rng.integers(0, 4))
        if kind == 0:
            return 256 * int(rng.integers(1, 5))
        if kind == 1:
            return 256 * int(rng.integers(1, 4)) + int(rng.integers(1, 256))
        if kind == 2:
            return 128 * int(rng.integers(1, 6))
        return int(rng.integers(1, 1100))
    m, n = edge(), edge()
    k = 64 * int(rng.integers(1, 9)) if int(rng.integers(0, 2)) else int(rng.integers(1, 600))
    alpha, beta = pick(rng, [(1.0, 0.0), (0.5, 2.0), (-1.5, 0.25), (1.0, 1.0), (0.0, 1.0)])
    kmeans = int(rng.integers(0, 5)) == 0
    shp = stored_shapes(ord_, ta, tb, m, n, k)

    def ld(c):
        kind = int(rng.integers(0, 3))
        return c if kind == 0 else ((c + 3) // 4 * 4 + 4 if kind == 1 else c + int(rng.integers(1, 9)))
    lds = [ld(sh[1]) for sh in shp]
    mats = [rng.uniform(-1, 1, (sh[0], l)).astype(np.float32) for sh, l in zip(shp, lds)]
    LAST.clear()
    LAST.update(kind="kernel kmeans" if kmeans else "kernel sgemm", ord=ord_, ta=ta, tb=tb, m=m, n=n, k=k, alpha=alpha,
                beta=beta, lds=lds)
    dev = [torch.from_numpy(x).cuda() for x in mats]
    st = torch.cuda.current_stream().cuda_stream
    ref = mats[2].copy()
    if kmeans:
        cl = rng.uniform(0, 8, m).astype(np.float32)
        pl = rng.uniform(0, 8, n).astype(np.float32)
        ones = rng.uniform(0.5, 1.5, max(m, n)).astype(np.float32)     # not constant: a swapped vector would show
        orc.skmeans_task(ord_, ta, tb, m, n, k, alpha, mats[0], lds[0], mats[1], lds[1], beta, ref, lds[2], cl, pl, ones)
        dv = [torch.from_numpy(x).cuda() for x in (cl, pl, ones)]
        bofhip.skmeans_task(ord_, ta, tb, m, n, k, alpha, dev[0].data_ptr(), lds[0], dev[1].data_ptr(), lds[1], beta,
                            dev[2].data_ptr(), lds[2], dv[0].data_ptr(), dv[1].data_ptr(), dv[2].data_ptr(), st)
    else:
        orc.sgemm(ord_, ta, tb, m, n, k, alpha, mats[0], lds[0], mats[1], lds[1], beta, ref, lds[2])
        bofhip.sgemm(ord_, ta, tb, m, n, k, alpha, dev[0].data_ptr(), lds[0], dev[1].data_ptr(), lds[1], beta,
                     dev[2].data_ptr(), lds[2], st)
    torch.cuda.synchronize()
    got = dev[2].cpu().numpy()
    if not np.array_equal(got, ref):
        bad = np.argwhere(got != ref)
        raise AssertionError(f"C differs at {len(bad)} elements, first {bad[0].tolist()}: got {got[tuple(bad[0])]} "
                             f"want {ref[tuple(bad[0])]}")
    return dict(LAST)


def one_case(seed, index, tmp):
    rng = np.random.default_rng([seed, index])
    kind = int(rng.integers(0, 10))
    for k in ENV_KNOBS:
        os.environ.pop(k, None)
    for f in os.listdir(tmp):
        if f.endswith(".bin"):          # the previous case's files, nothing else (tmp may be a directory the caller shares)
            os.unlink(os.path.join(tmp, f))
    if FORCE_KIND:
        kind = {"gemm": 0, "kmeans": 5, "csr": 9, "kernel": -1}[FORCE_KIND]
    try:
        if kind < 0:
            return kernel_case(rng)
        if kind < 5:
            return gemm_case(rng, tmp)
        if kind < 6:
            return gemm_case(rng, tmp, kmeans=True)
        return csr_case(rng, tmp)
    finally:
        for k in ENV_KNOBS:
            os.environ.pop(k, None)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_level3(dev, tmp_path, seed):
    """40 drawn cases per seed (a few seconds each batch)."""
    for i in range(40):
        try:
            one_case(seed, i, str(tmp_path))
        except Exception as e:
            raise AssertionError(f"fuzz case seed={seed} index={i} failed: {e}; case {LAST} "
                                 f"(python tests/test_gpu_fuzz.py --seed {seed} --only {i})") from e


@pytest.mark.parametrize("seed", [4])
def test_fuzz_kernels(dev, seed):
    """60 drawn level-1 cases (bof_sgemm / bof_skmeans_task), bit-exact against the oracle."""
    for i in range(60):
        try:
            rng = np.random.default_rng([seed, i])
            rng.integers(0, 10)            # the draw one_case spends on the kind: --only reproduces the case
            kernel_case(rng)
        except Exception as e:
            raise AssertionError(f"kernel fuzz case seed={seed} index={i} failed: {e}; case {LAST} "
                                 f"(python tests/test_gpu_fuzz.py --kind kernel --seed {seed} --only {i})") from e


if __name__ == "__main__":
    import argparse
    import tempfile
    import traceback
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, nargs="*", default=[])
    ap.add_argument("--dir", default=None)
    ap.add_argument("--range", type=int, nargs=2, default=None, help="run the cases A .. B-1 (instead of --only)")
    ap.add_argument("--kind", default="", choices=["", "gemm", "kmeans", "csr", "kernel"])
    ap.add_argument("--repeat", type=int, default=1, help="with --only: run every listed case this many times")
    ap.add_argument("--set", default="", help="with --only: override options of the drawn case, e.g. 'devices=None;n_streams=1'")
    ap.add_argument("--verify", action="store_true",
                    help="BOF_VERIFY=1: hand-over checksums inside every level-3 gemm / kmeans call; a mismatch fails the call, names "
                         "the two hand-over points and dumps the event ring (include/bof_hip.h, Instrumentation)")
    a = ap.parse_args()
    import faulthandler
    faulthandler.enable()          # a crash inside the library leaves the Python stack and the case's index behind
    if a.verify:
        os.environ["BOF_VERIFY"] = "1"
    if a.range:
        a.only = list(range(a.range[0], a.range[1]))
    quiet = bool(a.range)
    a.only = [i for i in a.only for _ in range(max(1, a.repeat))]
    for kv in [x for x in a.set.split(";") if x]:
        OVERRIDE[kv.split("=")[0]] = eval(kv.split("=", 1)[1])
    globals()["FORCE_KIND"] = a.kind
    bofhip.require_device()
    fails, i, t0 = 0, 0, time.time()
    with tempfile.TemporaryDirectory(dir=a.dir) as tmp:
        while (a.only and i < len(a.only)) or (not a.only and time.time() - t0 < a.seconds):
            idx = a.only[i] if a.only else i
            if i % 2000 == 0:
                print(f"... case {idx} ({i} done, {fails} failures, {time.time() - t0:.0f} s)", flush=True)
            try:
                d = one_case(a.seed, idx, tmp)
                if a.only and not quiet:
                    print("ok", idx, d)
            except Exception as e:
                fails += 1
                print(f"FAIL seed={a.seed} index={idx}: {e}\n     case: {LAST}")
                traceback.print_exc(limit=3)
            i += 1
    print(f"fuzz: {i} cases, {fails} failures, seed {a.seed}, {time.time() - t0:.0f} s"
          + (f", BOF_VERIFY on: {VERIFY_SUMS[0]} hand-over sums compared" if a.verify else ""))
    sys.exit(1 if fails else 0)
