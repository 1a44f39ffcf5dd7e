#!/usr/bin/env python3
"""What the GPU box's storage / page cache / PCIe paths deliver, measured with the library's own
file primitives (bof_file_sread / bof_file_swrite into pinned buffers) -- the numbers the level-3
pipeline of DESIGN.md section 4 is sized against.  Prints one JSON object."""
import argparse
import ctypes as C
import json
import mmap
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bofhip  # noqa: E402


def sysinfo(d):
    out = {"cpus": os.cpu_count(), "kernel": os.uname().release}
    try:
        nodes = sorted(x for x in os.listdir("/sys/devices/system/node") if x.startswith("node"))
        out["numa_nodes"] = {x: open(f"/sys/devices/system/node/{x}/cpulist").read().strip() for x in nodes}
    except OSError:
        pass
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith(("MemTotal", "MemAvailable")):
                out[ln.split(":")[0]] = ln.split(":")[1].strip()
    except OSError:
        pass
    best = ("", "?", "?")
    rp = os.path.realpath(d)
    for ln in open("/proc/mounts"):
        f = ln.split()
        if rp.startswith(f[1]) and len(f[1]) >= len(best[0]):
            best = (f[1], f[2], f[0])
    out["scratch_mount"] = {"point": best[0], "fs": best[1], "dev": best[2]}
    sv = os.statvfs(d)
    out["scratch_free_GB"] = round(sv.f_bavail * sv.f_frsize / 1e9, 1)
    # GPU -> NUMA node
    try:
        import glob
        for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
            out.setdefault("gpu_numa", {})[p.split("/")[4]] = open(p).read().strip()
    except OSError:
        pass
    # io_uring available?
    libc = C.CDLL(None, use_errno=True)
    params = (C.c_char * 120)()
    fd = libc.syscall(425, 4, params)
    out["io_uring_setup"] = "ok" if fd >= 0 else f"errno {C.get_errno()}"
    if fd >= 0:
        os.close(fd)
    return out


def run_threads(n, fn):
    th = [threading.Thread(target=fn, args=(i,)) for i in range(n)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default=os.environ.get("TMPDIR", "/tmp"))
    ap.add_argument("--gib", type=float, default=4.0)
    args = ap.parse_args()
    L = bofhip.lib()
    bofhip.require_device()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    res = {"sys": sysinfo(args.dir)}
    size = int(args.gib * 2**30)
    path = os.path.join(args.dir, "bof_iobench.bin")
    t = torch.empty(size // 4, dtype=torch.float32, device=dev)
    bofhip.gen_dense(t.data_ptr(), 0, t.numel(), "u", 1, st)
    torch.cuda.synchronize()

    # ---- file creation through the library (O_DIRECT, 8 writers) ---------------------------
    with open(path, "wb") as f:
        f.truncate(size)
    try:
        fd = os.open(path, os.O_RDWR | os.O_DIRECT)
        direct = True
    except OSError:
        fd = os.open(path, os.O_RDWR)
        direct = False
    res["o_direct_open"] = direct
    t0 = time.perf_counter()
    bofhip.device_to_file(bofhip.FPtr(fd, 0), size, t.data_ptr(), bofhip.default_options(use_odirect=int(direct)), st)
    res["create_first_write_GBps"] = round(size / (time.perf_counter() - t0) / 1e9, 2)
    t0 = time.perf_counter()
    bofhip.device_to_file(bofhip.FPtr(fd, 0), size, t.data_ptr(), bofhip.default_options(use_odirect=int(direct)), st)
    res["overwrite_odirect_8thr_GBps"] = round(size / (time.perf_counter() - t0) / 1e9, 2)
    os.fsync(fd)

    # pinned host buffers, one 64 MiB slot per thread
    slot = 64 << 20
    nthr_max = 16
    hbuf = []
    for _ in range(nthr_max):
        p = C.c_void_p()
        bofhip.check(L.bof_host_alloc(C.byref(p), slot), "host_alloc")
        hbuf.append(p.value)

    def reader(fd_, nthr, chunk, pattern, use_aio):
        nchunks = size // slot

        def work(i):
            for cidx in range(i, nchunks, nthr):
                off = cidx * slot
                if pattern == "contig":
                    for o in range(0, slot, chunk):
                        L.bof_file_sread(fd_, off + o, 0, 1, chunk, hbuf[i] + o, use_aio)
                else:  # tile rows: 4096 strides of 16 KB at 128 KB pitch, like a 4096^2 tile of a 32768-wide matrix
                    base = (cidx // 8) * (8 * slot) + (cidx % 8) * 16384
                    L.bof_file_sread(fd_, base, 131072, 4096, 16384, hbuf[i], use_aio)
        return size / run_threads(nthr, work) / 1e9

    def writer(fd_, nthr, chunk, use_aio):
        nchunks = size // slot

        def work(i):
            for cidx in range(i, nchunks, nthr):
                off = cidx * slot
                for o in range(0, slot, chunk):
                    L.bof_file_swrite(fd_, off + o, 0, 1, chunk, hbuf[i] + o, use_aio)
        return size / run_threads(nthr, work) / 1e9

    if direct:
        od = {}
        # one 32 MiB transfer per call, cut into `req`-sized requests that are submitted together
        for req_mib in (1, 2, 4, 8, 16, 32):
            L.bof_file_set_request_bytes(req_mib << 20)
            for nthr in (2, 4, 8, 16):
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                od[f"read_32M_as_{req_mib}M_requests_{nthr}thr"] = round(reader(fd, nthr, 32 << 20, "contig", 1), 2)
        L.bof_file_set_request_bytes(4 << 20)
        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        od["read_tile16K_rows_8thr"] = round(reader(fd, 8, 0, "tile", 1), 2)
        for req_mib in (1, 4, 8, 32):
            L.bof_file_set_request_bytes(req_mib << 20)
            for nthr in (2, 4, 8, 16):
                od[f"write_32M_as_{req_mib}M_requests_{nthr}thr"] = round(writer(fd, nthr, 32 << 20, 1), 2)
        L.bof_file_set_request_bytes(4 << 20)
        # reads and writes at the same time (the pipeline's steady state): 8 readers + 4 writers
        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        both = {}

        def rd_side():
            both["read"] = reader(fd, 8, 32 << 20, "contig", 1)

        def wr_side():
            nchunks = size // slot

            def work(i):
                for cidx in range(i, nchunks, 4):
                    L.bof_file_swrite(fd2, cidx * slot, 0, 1, 32 << 20, hbuf[8 + i], 1)
                    L.bof_file_swrite(fd2, cidx * slot + (32 << 20), 0, 1, 32 << 20, hbuf[8 + i] + (32 << 20), 1)
            both["write"] = size / run_threads(4, work) / 1e9
        path2 = path + ".2"
        with open(path2, "wb") as f:
            f.truncate(size)
        fd2 = os.open(path2, os.O_RDWR | os.O_DIRECT)
        bofhip.device_to_file(bofhip.FPtr(fd2, 0), size, t.data_ptr(), bofhip.default_options(use_odirect=1), st)
        ta = threading.Thread(target=rd_side)
        tb = threading.Thread(target=wr_side)
        ta.start(); tb.start(); ta.join(); tb.join()
        od["concurrent_read_8thr"] = round(both["read"], 2)
        od["concurrent_write_4thr"] = round(both["write"], 2)
        os.close(fd2)
        os.remove(path2)
        res["o_direct_GBps"] = od
    os.close(fd)

    # ---- page cache (buffered descriptor, file resident in DRAM) ---------------------------
    fd = os.open(path, os.O_RDWR)
    reader(fd, 8, 32 << 20, "contig", 0)  # warm
    pc = {}
    for nthr in (1, 2, 4, 8, 16):
        pc[f"pread_contig32M_{nthr}thr"] = round(reader(fd, nthr, 32 << 20, "contig", 0), 2)
    pc["pread_tile16K_rows_8thr"] = round(reader(fd, 8, 0, "tile", 0), 2)
    for nthr in (1, 2, 4, 8):
        pc[f"pwrite_contig32M_{nthr}thr"] = round(writer(fd, nthr, 32 << 20, 0), 2)
    # mmap store: no inode lock between writers
    mm = mmap.mmap(fd, size, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
    dst = np.frombuffer(mm, dtype=np.uint8)
    src = [np.ctypeslib.as_array((C.c_uint8 * slot).from_address(h)) for h in hbuf]
    for nthr in (1, 4, 8, 16):
        def work(i, nthr=nthr):
            for cidx in range(i, size // slot, nthr):
                np.copyto(dst[cidx * slot:(cidx + 1) * slot], src[i])
        pc[f"mmap_store_{nthr}thr"] = round(size / run_threads(nthr, work) / 1e9, 2)
    for nthr in (1, 4, 8, 16):
        def work(i, nthr=nthr):
            for cidx in range(i, size // slot, nthr):
                np.copyto(src[i], dst[cidx * slot:(cidx + 1) * slot])
        pc[f"mmap_load_{nthr}thr"] = round(size / run_threads(nthr, work) / 1e9, 2)
    del dst
    mm.close()
    res["page_cache_GBps"] = pc
    os.close(fd)
    os.remove(path)

    # ---- PCIe: pinned <-> HBM, linear and 2-D (tile rows) ----------------------------------
    hip = None
    for ln in open("/proc/self/maps"):
        if "libamdhip64" in ln:
            hip = C.CDLL(ln.split()[-1])
            break
    pcie = {}
    s1, s2 = C.c_void_p(), C.c_void_p()
    L.bof_stream_create(C.byref(s1))
    L.bof_stream_create(C.byref(s2))
    reps = 16

    def timed(fn):
        fn()
        L.bof_stream_sync(s1)
        L.bof_stream_sync(s2)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        L.bof_stream_sync(s1)
        L.bof_stream_sync(s2)
        return time.perf_counter() - t0

    d0 = t.data_ptr()
    pcie["h2d_linear_64M"] = round(reps * slot / timed(lambda: L.bof_memcpy_h2d(d0, hbuf[0], slot, s1)) / 1e9, 2)
    pcie["d2h_linear_64M"] = round(reps * slot / timed(lambda: L.bof_memcpy_d2h(hbuf[1], d0 + slot, slot, s2)) / 1e9, 2)

    def both():
        L.bof_memcpy_h2d(d0, hbuf[0], slot, s1)
        L.bof_memcpy_d2h(hbuf[1], d0 + slot, slot, s2)
    pcie["h2d_and_d2h_concurrent_each"] = round(reps * slot / timed(both) / 1e9, 2)

    def two_h2d():
        L.bof_memcpy_h2d(d0, hbuf[0], slot, s1)
        L.bof_memcpy_h2d(d0 + slot, hbuf[1], slot, s2)
    pcie["two_h2d_streams_total"] = round(2 * reps * slot / timed(two_h2d) / 1e9, 2)
    if hip is not None:
        hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                         C.c_int, C.c_void_p]
        # host panel chunk (512 rows x 128 KB) -> one packed tile's rows (16 KB wide), and back
        pcie["h2d_2d_16K_rows_of_128K_pitch"] = round(
            reps * (8 << 20) / timed(lambda: hip.hipMemcpy2DAsync(d0, 16384, hbuf[0], 131072, 16384, 512, 1, s1)) / 1e9, 2)
        pcie["d2h_2d_16K_rows_of_128K_pitch"] = round(
            reps * (8 << 20) / timed(lambda: hip.hipMemcpy2DAsync(hbuf[1], 131072, d0, 16384, 16384, 512, 2, s1)) / 1e9, 2)
    res["pcie_GBps"] = pcie
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
