#!/usr/bin/env python3
"""CPU baseline of bench.py (`cpu_baseline` leg), run as a CHILD process that never touches the GPU.

The reference computes with three Intel MKL routines: cblas_sgemm
(include/tasks/gemm_task.h:87-90, drivers/in_mem_gemm.cpp:63-70), mkl_scsrmm
(include/tasks/csrmm_task.h:226-228, drivers/in_mem_csrmm.cpp:116-121) and mkl_cspblas_scsrgemv
(include/tasks/csrgemv_task.h:74).  If an MKL runtime can be dlopen'ed on this box ($MKLROOT, conda,
the Python prefix, torch's lib directory, the loader path) those very routines are called with the
reference's call-site conventions (ILP64, "GXXC", zero-padding of the csrgemv offsets to a square
matrix) -- kind "port": the routine the reference calls on this box's cores, not the reference binary.
Otherwise the same operations go through PyTorch's CPU ops (torch.mm / torch.sparse.mm / torch.mv on
CSR; the wheel links MKL statically).  Threads = one per physical core the process may run on, pinned
(OMP_PROC_BIND / OMP_PLACES); one warm-up, best of 3, spread reported.  Bounded samples of the
BASELINE workloads (sizes in `sample`).  Inputs come from the oracle's generator restatement
(tests/orc.py: allowed here, this leg is the checker's side of the bench).  Prints one JSON object."""
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

# figures of the reference itself, measured on the survey container's 8 cores
SURVEY_8CORE = {"sgemm_gflops": 1070.0, "sgemm_4096_gflops": 755.0, "csrmm_gflops": 26.0}


def physical_cores():
    allowed = sorted(os.sched_getaffinity(0))
    seen, pick = set(), []
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            pick.append(c)
    return pick or allowed


def find_mkl():
    cands = []
    for root in filter(None, [os.environ.get("MKLROOT"), os.environ.get("CONDA_PREFIX"), "/opt/conda", sys.prefix,
                              sys.base_prefix, "/opt/intel/oneapi/mkl/latest", "/usr"]):
        for sub in ("lib", "lib/intel64", "lib64", "lib/x86_64-linux-gnu"):
            cands += sorted(glob.glob(os.path.join(root, sub, "libmkl_rt.so*")))
    try:
        import torch
        cands += sorted(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libmkl_rt.so*")))
    except Exception:
        pass
    cands += ["libmkl_rt.so", "libmkl_rt.so.2", "libmkl_rt.so.1"]
    for p in cands:
        try:
            return C.CDLL(p, mode=C.RTLD_GLOBAL), p
        except OSError:
            continue
    return None, None


def best_of(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), max(ts)


def main():
    cores = physical_cores()
    os.sched_setaffinity(0, cores)
    n_thr = len(cores)
    for k, v in (("OMP_NUM_THREADS", str(n_thr)), ("MKL_NUM_THREADS", str(n_thr)), ("MKL_DYNAMIC", "FALSE"),
                 ("OMP_PROC_BIND", "close"), ("OMP_PLACES", "cores"), ("MKL_INTERFACE_LAYER", "ILP64"),
                 ("MKL_THREADING_LAYER", "GNU")):
        os.environ[k] = v
    import numpy as np
    with_csr = "--no-csr" not in sys.argv
    out = {"unit": "GFLOP/s", "cores": n_thr, "kind": "port", "threads_pinned": True,
           "logical_cpus_allowed": len(os.sched_getaffinity(0))}
    mkl, where = find_mkl()
    i64, f32, P = C.c_int64, C.c_float, C.c_void_p

    def p(a):
        return a.ctypes.data_as(P)
    rng = np.random.default_rng(0)
    n_big = 16384
    a = rng.uniform(-1, 1, (n_big, n_big)).astype(np.float32)
    b = rng.uniform(-1, 1, (n_big, n_big)).astype(np.float32)
    c = np.empty((n_big, n_big), np.float32)
    if mkl is not None:
        buf = C.create_string_buffer(256)
        try:
            mkl.MKL_Get_Version_String(buf, 256)
        except Exception:
            pass
        out["library"] = f"{where}: {buf.value.decode(errors='replace')[:90]}"
        out["what"] = "Intel MKL cblas_sgemm / mkl_scsrmm / mkl_cspblas_scsrgemv (ILP64) called as the reference's tasks call them"
        mkl.cblas_sgemm.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, i64, f32, P, i64, P, i64, f32, P, i64]

        def sgemm(n):
            mkl.cblas_sgemm(101, 111, 111, n, n, n, 1.0, p(a), n_big, p(b), n_big, 0.0, p(c), n_big)
    else:
        import torch
        torch.set_num_threads(n_thr)
        out["library"] = "torch " + torch.__version__ + " CPU ops (MKL linked statically: " + str(torch.backends.mkl.is_available()) + ")"
        out["what"] = "no MKL runtime could be dlopen'ed on this box: MKL through torch.mm / torch.sparse.mm / torch.mv"
        ta, tb = torch.from_numpy(a), torch.from_numpy(b)

        def sgemm(n):
            torch.mm(ta[:n, :n], tb[:n, :n])
    def set_threads(nt):
        if mkl is not None:
            mkl.MKL_Set_Num_Threads(C.c_int(nt))
        else:
            import torch
            torch.set_num_threads(nt)
    sgemm(4096)                      # warm-up (MKL's first call is several times slower)
    t4, t4hi = best_of(lambda: sgemm(4096))
    sgemm(8192)
    t16, t16hi = best_of(lambda: sgemm(n_big), 3)
    out["value"] = round(2.0 * n_big ** 3 / t16 / 1e9, 1)
    out["value_worst_of_3"] = round(2.0 * n_big ** 3 / t16hi / 1e9, 1)
    out["sgemm_4096_gflops"] = round(2.0 * 4096 ** 3 / t4 / 1e9, 1)
    out["sgemm_4096_gflops_worst_of_3"] = round(2.0 * 4096 ** 3 / t4hi / 1e9, 1)
    out["sample"] = (f"sgemm 16384^3 fp32 (1/8 of a step), best of 3 after warm-up: {t16:.2f} s (worst {t16hi:.2f}); "
                     f"4096^3 (BASELINE configs[0], one tile task): {t4 * 1e3:.0f} ms (worst {t4hi * 1e3:.0f})")
    if "--full-step" in sys.argv:
        # the WORKLOAD itself, not a sample of it: one in-memory 32768^3 cblas_sgemm, what
        # drivers/in_mem_gemm.cpp:63-70 times for BASELINE configs[1]'s matrices -- one call, no warm-up beyond the
        # sample above; only when the sample predicts <= 150 s and the box has the 12 GiB + slack
        try:
            avail = 0
            for ln in open("/proc/meminfo"):
                if ln.startswith("MemAvailable"):
                    avail = int(ln.split()[1]) * 1024
            n2 = 2 * n_big
            if 8 * t16 > 150:
                out["full_step_skipped"] = f"predicted {8 * t16:.0f} s"
            elif avail < (40 << 30):
                out["full_step_skipped"] = f"{avail >> 30} GiB of memory available"
            else:
                a2 = np.empty((n2, n2), np.float32)
                b2 = np.empty((n2, n2), np.float32)
                for i in (0, 1):
                    for j in (0, 1):
                        a2[i * n_big:(i + 1) * n_big, j * n_big:(j + 1) * n_big] = a
                        b2[i * n_big:(i + 1) * n_big, j * n_big:(j + 1) * n_big] = b
                c2 = np.zeros((n2, n2), np.float32)        # (pages touched before the clock starts)
                t0 = time.perf_counter()
                if mkl is not None:
                    mkl.cblas_sgemm(101, 111, 111, n2, n2, n2, 1.0, p(a2), n2, p(b2), n2, 0.0, p(c2), n2)
                else:
                    import torch
                    torch.mm(torch.from_numpy(a2), torch.from_numpy(b2), out=torch.from_numpy(c2))
                tf = time.perf_counter() - t0
                out["full_step_s"] = round(tf, 2)
                out["full_step_gflops"] = round(2.0 * n2 ** 3 / tf / 1e9, 1)
                out["full_step_what"] = "ONE in-memory 32768^3 sgemm = BASELINE configs[1]'s matrices (drivers/in_mem_gemm.cpp:63-70), one call"
                del a2, b2, c2
        except Exception as e:
            out["full_step_skipped"] = f"{type(e).__name__}: {str(e)[:120]}"
    del a, b, c
    flags = []
    if out["value"] < SURVEY_8CORE["sgemm_gflops"]:
        flags.append(f"sgemm below the reference's own {SURVEY_8CORE['sgemm_gflops']:.0f} GFLOP/s on 8 survey cores")
    if with_csr:
        try:
            import orc
            # ---- CSRMM: rows [0, 500k) of the cfg3 matrix (5e7 nnz) x 1M x 128: 1/20 of BASELINE configs[2] ----
            m, n, k, npr = 500_000, 1_000_000, 128, 100
            val, col, off = orc.sparse_create(m, n, npr / n)
            bm = orc.dense_fill(n, k, "s")
            cm = np.zeros((m, k), np.float32)
            if mkl is not None:
                desc = C.create_string_buffer(b"GXXC", 6)
                tr = C.c_char(b"N")
                M, N, K, LDB, LDC = i64(m), i64(n), i64(k), i64(k), i64(k)
                al, be = f32(1.0), f32(0.0)
                ptre = off[1:]

                def csrmm():
                    mkl.mkl_scsrmm(C.byref(tr), C.byref(M), C.byref(K), C.byref(N), C.byref(al), desc, p(val), p(col),
                                   p(off), p(ptre), p(bm), C.byref(LDB), C.byref(be), p(cm), C.byref(LDC))
            else:
                import torch
                A = torch.sparse_csr_tensor(torch.from_numpy(off), torch.from_numpy(col), torch.from_numpy(val), size=(m, n))
                B = torch.from_numpy(bm)

                def csrmm():
                    cm[:] = torch.sparse.mm(A, B).numpy()
            # MKL's sparse routines do not always scale to every core of a large box: the thread count that
            # gives the best rate is the one reported (sweep: 8 as in the survey container, 32, all)
            sweep = {}
            t, thi, used = None, None, n_thr
            for nt in sorted({min(8, n_thr), min(32, n_thr), n_thr}):
                set_threads(nt)
                csrmm()
                lo, hi = best_of(csrmm)
                sweep[str(nt)] = round(2.0 * m * npr * k / lo / 1e9, 2)
                if t is None or lo < t:
                    t, thi, used = lo, hi, nt
            gf = 2.0 * m * npr * k / t / 1e9
            out["csrmm"] = {"value": round(gf, 2), "value_worst_of_3": round(2.0 * m * npr * k / thi / 1e9, 2),
                            "unit": "GFLOP/s", "cores": used, "gflops_by_threads": sweep,
                            "sample": f"rows [0, 500k) of the cfg3 matrix (5e7 nnz) x 1M x 128, best of 3: {t:.3f} s",
                            "checksum_first_row": [float(v) for v in cm[0, :4]]}     # [1950, 2446, 1692, 2188]: App. A-3
            if gf < SURVEY_8CORE["csrmm_gflops"]:
                flags.append(f"csrmm below the reference's own {SURVEY_8CORE['csrmm_gflops']:.0f} GFLOP/s on 8 survey cores")
            del val, col, off, bm, cm
            # ---- CSRGEMV 'N': rows [0, 2M) of the 50M x 50M matrix (10 nnz/row) ----------------------------
            m, n, npr = 2_000_000, 50_000_000, 10
            val, col, off = orc.sparse_create(m, n, npr / n)
            x = (np.arange(n) % 10).astype(np.float32)
            if mkl is not None:
                # csrgemv_task.h:36-44: the block is padded to a square matrix of dim = n rows (empty rows)
                iap = np.empty(n + 1, np.int64)
                iap[: m + 1] = off - off[0]
                iap[m + 1:] = iap[m]
                y = np.zeros(n, np.float32)
                tr = C.c_char(b"N")
                DIM = i64(n)

                def gemv():
                    mkl.mkl_cspblas_scsrgemv(C.byref(tr), C.byref(DIM), p(val), p(iap), p(col), p(x), p(y))
            else:
                import torch
                A = torch.sparse_csr_tensor(torch.from_numpy(off), torch.from_numpy(col), torch.from_numpy(val), size=(m, n))
                xt = torch.from_numpy(x)
                y = np.zeros(m, np.float32)

                def gemv():
                    y[:] = torch.mv(A, xt).numpy()
            sweep = {}
            t, thi, used = None, None, n_thr
            for nt in sorted({min(8, n_thr), min(32, n_thr), n_thr}):
                set_threads(nt)
                gemv()
                lo, hi = best_of(gemv)
                sweep[str(nt)] = round(2.0 * m * npr / lo / 1e9, 3)
                if t is None or lo < t:
                    t, thi, used = lo, hi, nt
            out["csrgemv_N"] = {"value": round(2.0 * m * npr / t / 1e9, 3), "value_worst_of_3": round(2.0 * m * npr / thi / 1e9, 3),
                                "unit": "GFLOP/s", "cores": used, "gflops_by_threads": sweep,
                                "sample": f"rows [0, 2M) of the 50M x 50M matrix (2e7 nnz) x vector"
                                          + (", padded to 50M rows as the reference's task does" if mkl is not None else "")
                                          + f", best of 3: {t * 1e3:.1f} ms",
                                "checksum_y0_6": [float(v) for v in y[:6]]}        # [230, 274, 243, 172, 222, 348]
        except Exception as e:
            out["csr_error"] = f"{type(e).__name__}: {str(e)[:160]}"
    out["below_survey_reference"] = flags
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
