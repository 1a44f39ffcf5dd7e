/*
 * bof_hip.h -- C ABI of the MI355X-native BLAS-on-flash hot path
 * (libbof_hip.so).  Plain pointers and sizes only; no C++/torch types.
 *
 * The library replaces, for the reference's _gemm / _csrmm / _csrgemv path:
 *
 *   level 1  the per-tile MKL calls inside the task objects
 *              cblas_sgemm            include/tasks/gemm_task.h:87-90
 *              mkl_scsrmm             include/tasks/csrmm_task.h:226-228, 310-312
 *              mkl_cspblas_scsrgemv   include/tasks/csrgemv_task.h:74, 165
 *            -> bof_sgemm / bof_scsrmm / bof_scsrgemv on DEVICE pointers.
 *   level 2  the tile DAG of flash::gemm / csrmm / csrgemv run over matrices
 *            that are already resident in HBM (the reference's "program cache"
 *            is host DRAM; ours is HBM)
 *              src/blas/gemm.cpp:27-202, src/blas/csrmm.cpp:64-126,203-266,
 *              src/blas/csrgemv.cpp:14-97
 *            -> bof_gemm_resident / bof_csrmm_resident / bof_csrgemv_resident.
 *   level 3  the same calls on file-resident matrices (O_DIRECT AIO reader ->
 *            pinned ring -> hipMemcpyAsync -> HBM tile cache -> kernels ->
 *            write-back): what the C++ flash::gemm/csrmm/csrgemv in
 *            blas-on-flash_amd/include/flash_blas.h bind to
 *              include/flash_blas.h:14-18, 37-40, 55-57
 *            -> bof_flash_gemm / bof_flash_csrmm / bof_flash_csrgemv.
 *   next rows: bof_*_csrcsc, csrmm 'T', column-major B/C, and flash::kmeans as
 *            a fused GEMM epilogue (include/tasks/kmeans_task.h:53-82) at all three levels
 *            -> bof_skmeans_task / bof_kmeans_resident / bof_flash_kmeans.
 *
 * Conventions: all integers 64-bit (reference builds with -DMKL_ILP64,
 * include/bof_types.h:11-28); CSR index and offset arrays are int64 exactly as
 * stored on disk; chars are 'N'/'T', 'R'/'C' as in the reference API.
 * Threading: level-2 and level-3 calls share per-device state (compute streams, scratch,
 * pinned staging rings, the HBM tile slab); the library serialises host threads that enter
 * them on one device for the duration of the call.  Level-2 calls are asynchronous: two of
 * them queued on DIFFERENT streams of one device may still overlap on the GPU and share
 * scratch, so issue level-2 work for one device on one stream (or synchronise in between).
 * Every function returns BOF_OK (0) or a negative error code;
 * bof_last_error() gives the message.  Nothing here falls back to a CPU
 * implementation: without a usable HIP device the compute entry points fail
 * with BOF_ENODEV.
 */
#ifndef BOF_HIP_H
#define BOF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BOF_OK 0
#define BOF_EINVAL (-1) /* bad argument (the reference returns -1 too, csrmm.cpp:433-448) */
#define BOF_EHIP (-2)   /* a HIP runtime call failed */
#define BOF_EIO (-3)    /* file I/O failed */
#define BOF_ENODEV (-4) /* no HIP device */
#define BOF_ENOMEM (-5)
#define BOF_EVERIFY (-6) /* bof_options.verify: a hand-over checksum did not match (bof_last_error names it) */

#define BOF_ABI_VERSION 5

/* ---- library ---------------------------------------------------------------- */
int bof_abi_version(void);
const char *bof_last_error(void);
int bof_device_count(void); /* hipGetDeviceCount; 0 when there is no GPU */
int bof_set_device(int dev);

/* Reference compile-time tunables (CMakeLists.txt:38-63) as run-time options. */
typedef struct {
  int64_t gemm_blk;      /* GEMM_BLK_SIZE            default 4096 (BASELINE cfg2) */
  int64_t max_nnzs;      /* MAX_NNZS                 default 10,000,000           */
  int64_t csrmm_rblk;    /* CSRMM_RM_RBLK_SIZE       default 131072               */
  int64_t csrmm_cblk;    /* CSRMM_RM_CBLK_SIZE       default 1024                 */
  int64_t hbm_budget;    /* PROGRAM_BUDGET analogue: bytes of HBM for the tile
                            cache; 0 = 80% of free HBM                          */
  int32_t n_io_threads;  /* N_IO_THR (reference 4)   default 8                    */
  int32_t n_streams;     /* compute streams (N_COMPUTE_THR analogue) default 4    */
  int32_t use_odirect;   /* 1 = O_DIRECT + kernel AIO (default), 0 = buffered     */
  int32_t pinned_slots;  /* pinned staging ring slots  default 8                  */
  int32_t gemm_path;     /* bof_flash_gemm: 0 = choose (default), 1 = tile cache
                            (packed tiles, Belady replacement: any budget, any
                            layout), 2 = row panels kept in HBM in file layout,
                            read/written as large contiguous requests (needs B,
                            two A panels and three C panels inside hbm_budget)    */
  int32_t io_chunk_mib;  /* size of one panel read/write request  default 32      */
  /* ---- ABI v3 ---------------------------------------------------------------- */
  /* Level-3 calls shard over these HIP devices INSIDE the calling process, by output row
   * block: C row panels (gemm / kmeans) or nnz-balanced row blocks (csrmm /
   * csrgemv) are dealt to the devices in contiguous ranges; an operand every device needs
   * (B, x) is read from its file ONCE into a pinned slot and copied from there to every
   * device over that device's own PCIe link; csrgemv 'T' partials are summed device to
   * device (peer access) before they leave for the host.  The reference runs its
   * N_COMPUTE_THR workers inside one process behind flash::gemm in the same way
   * (src/scheduler/scheduler.cpp:9-16, src/lib_funcs.cpp:9).
   * n_devices = 0: $BOF_DEVICES ("0,1,2" or "all") if set, else only the calling thread's
   * current device.  An ordinal may repeat: one GPU then plays several devices (how the
   * sharded path is exercised on a 1-GPU box). */
  int32_t n_devices;
  int32_t devices[16];
  /* per-call forms of what used to be process-wide environment knobs; 0 = the default,
   * which is the environment variable named, read at every call, else the built-in value.
   * io_engine and io_request_kib are applied by setting the file layer's PROCESS-WIDE state at
   * the start of the call: level-3 calls that run concurrently in one process (allowed on
   * disjoint device lists) must agree on them, or the later call's values also govern the rest
   * of the earlier one -- results are unaffected, request sizes / the engine are not. */
  int32_t io_engine;      /* 1 kernel AIO, 2 io_uring      ($BOF_IO_ENGINE=uring)           */
  int32_t io_request_kib; /* O_DIRECT request size         ($BOF_IO_REQUEST_KIB, 4096)      */
  int32_t panel_group;    /* C panels of the ramp group    ($BOF_PANEL_GROUP, computed)     */
  int32_t panel_streams;  /* compute streams, panel path   ($BOF_PANEL_STREAMS; 1, flash::kmeans min(n_streams, 2)) */
  int32_t panel_writers;  /* writer threads, panel path    ($BOF_PANEL_WRITERS, n_io_threads / 2)  */
  int32_t panel_kmajor;   /* k-major panel copies: 1 off, 2 on, 3 on even for tiles reused
                             fewer than 4 times            ($BOF_PANEL_KMAJOR + 1, 2)       */
  /* One PROCESS per GPU on one node (torchrun-style), every rank calling bof_flash_gemm on its
   * row slab of the same A / B / C files: with share_world > 1 the operand every rank needs (B for
   * row-major) is read from storage ONCE PER NODE instead of once per rank.  Its row panels are
   * dealt round-robin to the ranks (panel l belongs to rank l % share_world); the owner reads a
   * panel from the file (O_DIRECT, as always) and, besides copying it to its own GPU, publishes it
   * chunk by chunk in a node-shared staging ring (POSIX shared memory `share_name`.*: 64 chunk slots,
   * reused once every peer has taken a slot's chunk; futex wake-ups); the other ranks take the chunk
   * from there instead of from the file.  Same
   * panels, same order, same bits.  Every participating rank must make the same call (same
   * problem, same options) with its own share_rank in [0, share_world); share_name must be new
   * for every collective call and is removed with bof_share_cleanup once all ranks have returned.
   * Row-panel path only (the tile cache reads its tiles itself). */
  int32_t share_world;    /* 0 / 1: off */
  int32_t share_rank;
  char share_name[48];    /* e.g. "/bof_29500_7"; at most 40 characters */
  /* ---- ABI v4 ---------------------------------------------------------------- */
  int32_t kernel_timing;  /* 1: a pair of HIP timing events around every compute launch of a level-3
                             call, on the stream it is launched on; summed into
                             bof_flash_stats.kernel_seconds / kernel_launches.  0 = off           */
  int32_t verify;         /* hand-over checksums of the level-3 pipelines ($BOF_VERIFY): 0 = the
                             environment variable, else off; 1 = on; 2 = off whatever the
                             environment says.  See "Instrumentation" below.  flash::csrmm /
                             flash::csrgemv: a receipt per launch (every workgroup counts itself, a
                             checker behind the launch compares with 1; BOF_EVERIFY on a miss)      */
  int32_t peer_bcast;     /* in-process device list: an operand every device needs is copied over
                             PCIe to ONE device only and passed on device to device
                             (hipMemcpyPeerAsync over xGMI); 0 = $BOF_PEER_BCAST, else off; 1 on;
                             2 off                                                                */
  /* ---- ABI v5 ---------------------------------------------------------------- */
  int32_t gemm_chain;     /* how the k-blocks of a C tile are combined by flash::gemm (levels 2 and 3):
                             0 / 2 (default): ONE k-ordered fmaf chain per output element over the whole K, scaled once
                                at the end -- c = beta == 0 ? alpha*acc : fmaf(alpha, acc, beta*c) -- however the tiler,
                                the HBM budget or the schedule cut K: k-ranges that run as separate launches hand their
                                raw fp32 accumulators on, and a C panel whose operands are complete runs as a single
                                launch over the whole K.  The result equals bof_sgemm on the whole matrices bit for
                                bit, i.e. what drivers/in_mem_gemm.cpp:63-70 computes with its one cblas_sgemm call.
                             1: the reference's task arithmetic, one rounding per k-block: C = alpha*A_l*B_l + C for
                                l > 0 (src/blas/gemm.cpp:122-127, include/tasks/gemm_task.h:87-90) -- bit-identical to
                                the tile-by-tile oracle (oracle/bof_oracle.c: orc_flash_gemm).
                             flash::kmeans always runs its tasks the reference's way (every k-block's task adds the
                             rank-1 terms).  The two differ by rounding only (~1e-7 relative; the bar is 1e-4).      */
  int32_t reserved_[4];   /* must be zero */
} bof_options;
#define BOF_MAX_DEVICES 16
void bof_default_options(bof_options *o);

/* ---- device memory / streams (thin wrappers for non-torch hosts) ------------ */
int bof_malloc(void **dptr, size_t bytes);
int bof_free(void *dptr);
int bof_host_alloc(void **hptr, size_t bytes); /* pinned */
int bof_host_free(void *hptr);
int bof_memcpy_h2d(void *d, const void *h, size_t bytes, void *stream);
int bof_memcpy_d2h(void *h, const void *d, size_t bytes, void *stream);
int bof_memset(void *d, int value, size_t bytes, void *stream);
int bof_stream_create(void **stream);
int bof_stream_destroy(void *stream);
int bof_stream_sync(void *stream); /* NULL = default stream */
int bof_mem_info(size_t *free_bytes, size_t *total_bytes);

/* ---- level 1: per-tile compute on device pointers --------------------------- */
/* C = alpha*op(A)*op(B) + beta*C, fp32, cblas_sgemm argument meaning
 * (include/tasks/gemm_task.h:87-90).  Exact-f32 MFMA: every output element is a
 * k-ordered fmaf chain started at 0, then
 *   c = (beta == 0) ? alpha*acc : fmaf(alpha, acc, beta*c).
 * `stream` is a hipStream_t (NULL = default stream); the call is asynchronous. */
int bof_sgemm(char ord, char trans_a, char trans_b, int64_t m, int64_t n,
              int64_t k, float alpha, const float *a, int64_t lda, const float *b,
              int64_t ldb, float beta, float *c, int64_t ldc, void *stream);
/* KMeansTask::execute (include/tasks/kmeans_task.h:53-82) on one tile, as ONE kernel: the
 * sgemm above, then the task's two K = 1 products added in the store,
 *   C[r][c] += c_l2sq[r] * ones[c];   C[r][c] += ones[r] * p_l2sq[c]
 * (r along m, c along n, for both storage orders), each rounded as the task's
 * cblas_sgemm(.., K = 1, alpha = 1, beta = 1) rounds it: c = c + round(u*v).  The fused tile
 * equals the three-call sequence bit for bit; C crosses HBM once instead of five times.
 * c_l2sq: >= m, p_l2sq: >= n, ones: >= max(m, n) floats, all DEVICE pointers.
 * Row-major note: the reference hands its K = 1 products lda = a_nrows for an a_nrows x 1
 * row-major operand (kmeans_task.h:74-75), i.e. it reads c_l2sq[r * a_nrows] -- out of bounds;
 * its driver only ever uses 'C' (drivers/kmeans.cpp:37-39).  Here 'R' has the meaning above. */
int bof_skmeans_task(char ord, char trans_a, char trans_b, int64_t m, int64_t n,
                     int64_t k, float alpha, const float *a, int64_t lda, const float *b,
                     int64_t ldb, float beta, float *c, int64_t ldc, const float *c_l2sq,
                     const float *p_l2sq, const float *ones, void *stream);

/* C[m x n] = alpha * A[m x k] * B[k x n] + beta * C with A in 0-based CSR:
 * mkl_scsrmm('N', m, n, k, alpha, "GXXC"|"GXXF", val, col, ptr, ptr+1, b, ldb,
 * beta, c, ldc) (include/tasks/csrmm_task.h:226-228, 310-312).  `ptr` has m+1
 * entries and may carry a base: row i spans [ptr[i]-ptr[0], ptr[i+1]-ptr[0]) of
 * val/col.  col stays int64 as on disk and is NEVER modified (the reference's
 * in-place 1-based conversion, SURVEY App. B-15, is not reproduced).
 * ord_b 'R': B,C row-major; 'C': column-major. */
int bof_scsrmm(char ord_b, int64_t m, int64_t n, int64_t k, float alpha,
               const float *val, const int64_t *col, const int64_t *ptr,
               const float *b, int64_t ldb, float beta, float *c, int64_t ldc,
               void *stream);

/* mkl_cspblas_scsrgemv (include/tasks/csrgemv_task.h:74, 165) on a row block:
 * 'N': y[0..m) = A x (overwrites).  'T': y[0..n) += A^T x[0..m) (accumulates
 * with fp32 atomics; the caller zeroes y once, src/blas/csrgemv.cpp:64). */
int bof_scsrgemv(char trans, int64_t m, int64_t n, const float *val,
                 const int64_t *ptr, const int64_t *col, const float *x, float *y,
                 void *stream);

/* mkl_csrcsc(job = {0,0,0,-1,-1,1}) (include/tasks/csrcsc_task.h:66-75) without the
 * padding to a square: A = CSR(val, ptr[m+1] (any base), col) of shape m x n ->
 * A^T = CSR(val_tr, ptr_tr[n+1] (0-based), col_tr) of shape n x m, source rows
 * ascending inside every output row (stable), m, n <= INT32_MAX.  nnz =
 * ptr[m] - ptr[0] is passed by the caller (it planned the buffers with it).
 * Temporary HBM comes from the library's per-device scratch (bof_flash_release
 * returns it); bof_csrcsc_workspace_bytes tells how much that is. */
int bof_scsrcsc(int64_t m, int64_t n, int64_t nnz, const float *val, const int64_t *ptr,
                const int64_t *col, float *val_tr, int64_t *ptr_tr, int64_t *col_tr,
                void *stream);
uint64_t bof_csrcsc_workspace_bytes(int64_t n, int64_t nnz);

/* ---- planning (pure host code; usable without a GPU) ------------------------ */
/* One tile task as src/blas/gemm.cpp:83-129 builds it (offsets/LDs in elements). */
typedef struct {
  int64_t l, i, j;
  int64_t M, K, N;
  int64_t off[3];     /* A, B, C tile origin in the file / resident matrix */
  int64_t nrows[3];   /* StrideInfo.n_strides                              */
  int64_t ncols[3];   /* StrideInfo.len_per_stride / sizeof(float)         */
  int64_t ld_file[3]; /* StrideInfo.stride / sizeof(float)                 */
  float beta;         /* caller's beta for l == 0, 1 for l > 0            */
  int64_t parent;     /* index of (l-1,i,j) in the plan or -1             */
} bof_gemm_task;
/* Returns the task count (tasks listed in the reference's l-major injection
 * order); fills out[0..min(count,cap)).  nblk = {N_m, N_k, N_n}. */
int64_t bof_gemm_plan(char ord, char trans_a, char trans_b, int64_t m, int64_t n,
                      int64_t k, float beta, int64_t lda, int64_t ldb, int64_t ldc,
                      int64_t blk, bof_gemm_task *out, int64_t cap, int64_t nblk[3]);
/* CSR row blocks (include/blas_utils.h:72-97) with the rows-remaining clamp. */
int64_t bof_csr_blocks(const int64_t *ia, int64_t m, int64_t min_rows,
                       int64_t max_rows, int64_t max_nnz, int64_t *starts,
                       int64_t *sizes, int64_t cap);

/* ---- level 2: the tile DAG over HBM-resident matrices ----------------------- */
/* flash::gemm semantics (src/blas/gemm.cpp:27-202) with a,b,c device pointers to
 * the whole matrices (file layout, leading dims as the caller passes them; 0 =
 * default).  Default arithmetic (opts->gemm_chain 0 / 2): one k-ordered chain per output
 * element over the whole K -- with everything resident that is ONE launch over the whole
 * matrices, queued on `stream` (what drivers/in_mem_gemm.cpp:63-70 does with one call).
 * gemm_chain = 1 (and alpha == 0): the reference's tasks in the reference's order, every
 * (i,j) accumulate chain serialised on one of opts->n_streams compute streams forked from /
 * joined into `stream`.  Asynchronous with respect to the host.
 * Scratch: when an operand is stored k-contiguous (A 'N', B 'T' in row-major terms), every
 * tile is reused by >= 4 tasks and a copy of that operand fits in a QUARTER of the free HBM,
 * the call first writes a k-major copy of it into library scratch (kept until
 * bof_flash_release) so that all tile launches take the LDS-DMA kernel; otherwise the
 * register-staging kernel runs on the operand as it is.  Results are bit-identical either way. */
int bof_gemm_resident(char ord, char trans_a, char trans_b, int64_t m, int64_t n,
                      int64_t k, float alpha, float beta, const float *a,
                      const float *b, float *c, int64_t lda, int64_t ldb,
                      int64_t ldc, const bof_options *opts, void *stream);
/* flash::kmeans (src/blas/kmeans.cpp:27-198) over resident matrices: the gemm
 * tiler with one bof_skmeans_task per tile.  Tile (l, i, j) gets c_l2sq + i*blk_m and
 * p_l2sq + j*blk_n and the un-offset `ones` (kmeans.cpp:115-118, 128-131); like the
 * reference, EVERY k-block's task adds the two products, so they are added N_k times when k
 * spans several blocks (the reference's use has k = the point dimension <= one block).
 * A zero dimension: nothing to do, C untouched (the reference's tiler divides by zero there,
 * kmeans.cpp:52, 76-77).  c_l2sq: m, p_l2sq: n, ones: the largest tile
 * edge (<= min(max(m, n), gemm_blk + 127)) floats, DEVICE pointers. */
int bof_kmeans_resident(char ord, char trans_a, char trans_b, int64_t m, int64_t n,
                        int64_t k, float alpha, float beta, const float *a,
                        const float *b, float *c, int64_t lda, int64_t ldb,
                        int64_t ldc, const float *c_l2sq, const float *p_l2sq,
                        const float *ones, const bof_options *opts, void *stream);
/* flash::csrmm 'N' (src/blas/csrmm.cpp:64-126, 203-266): row blocks by nnz
 * budget x column panels; ia_host is the host copy of the offsets (the
 * reference also reads `ia` to the host first, csrmm.cpp:69-71), ia_dev the
 * same array in HBM. */
int bof_csrmm_resident(char trans_a, int64_t m, int64_t n, int64_t k, float alpha,
                       float beta, const float *val, const int64_t *ia_host,
                       const int64_t *ia_dev, const int64_t *ja, char ord_b,
                       const float *b, float *c, const bof_options *opts,
                       void *stream);
/* flash::csrgemv (src/blas/csrgemv.cpp:82-97) with x, y in HBM. */
int bof_csrgemv_resident(char trans_a, int64_t m, int64_t n, const float *val,
                         const int64_t *ia_host, const int64_t *ia_dev,
                         const int64_t *ja, const float *x, float *y,
                         const bof_options *opts, void *stream);

/* ---- level 3: file-resident matrices (the flash_ptr boundary) --------------- */
/* A flash_ptr<T> is {file, byte offset}: include/pointers/pointer.h:15-18. */
typedef struct {
  int fd;           /* open file descriptor (O_DIRECT or buffered)            */
  uint64_t foffset; /* byte offset of element 0                              */
} bof_fptr;
/* Blocking; returns after C has been written back to its file. */
int bof_flash_gemm(char ord, char trans_a, char trans_b, uint64_t m, uint64_t n,
                   uint64_t k, float alpha, float beta, bof_fptr a, bof_fptr b,
                   bof_fptr c, uint64_t lda, uint64_t ldb, uint64_t ldc,
                   const bof_options *opts);
/* flash::kmeans (include/flash_blas.h:20-25): bof_flash_gemm's pipeline with the tasks of
 * bof_kmeans_resident.  c_l2sq (m), p_l2sq (n) and ones (largest tile edge) are HOST arrays,
 * as in the reference; blocking. */
int bof_flash_kmeans(char ord, char trans_a, char trans_b, uint64_t m, uint64_t n,
                     uint64_t k, float alpha, float beta, bof_fptr a, bof_fptr b,
                     bof_fptr c, uint64_t lda, uint64_t ldb, uint64_t ldc,
                     const float *c_l2sq, const float *p_l2sq, const float *ones,
                     const bof_options *opts);
/* include/flash_blas.h:37-40.  A is an m x n CSR.  trans_a 'N': B n x k, C m x k.
 * trans_a 'T': C[n x k] = alpha * A^T * B[m x k] + beta * C; A^T is built in HBM by
 * bof_scsrcsc (the reference goes through csrcsc into temporary files,
 * src/blas/csrmm.cpp:355-422, and is wrong there -- SURVEY App. B-3). */
int bof_flash_csrmm(char trans_a, uint64_t m, uint64_t n, uint64_t k, float alpha,
                    float beta, bof_fptr a, bof_fptr ia, bof_fptr ja, char ord_b,
                    bof_fptr b, bof_fptr c, const bof_options *opts);
/* flash::csrcsc (include/flash_blas.h:49-52, src/blas/csrcsc.cpp:32-159):
 * CSR(ia, ja, a) of shape m x n -> CSR(ia_tr, ja_tr, a_tr) of shape n x m in the three
 * output files (ia_tr: n+1 int64, ja_tr: nnz int64, a_tr: nnz fp32), source rows
 * ascending inside every output row.  The matrix is transposed whole in HBM when
 * it fits opts->hbm_budget (0 = most of the free HBM); otherwise in row blocks
 * through temporary files + a column-block merge, as the reference does. */
int bof_flash_csrcsc(uint64_t m, uint64_t n, bof_fptr ia, bof_fptr ja, bof_fptr a,
                     bof_fptr ia_tr, bof_fptr ja_tr, bof_fptr a_tr,
                     const bof_options *opts);
/* csrmm with B (n x k) and C (m x k) in HOST memory: the reference's second overload
 * (include/flash_blas.h:43-46, src/blas/csrmm.cpp:453-472). */
int bof_flash_csrmm_inmem(char trans_a, uint64_t m, uint64_t n, uint64_t k, float alpha,
                          float beta, bof_fptr a, bof_fptr ia, bof_fptr ja, char ord_b,
                          const float *b, float *c, const bof_options *opts);
/* b (input vector) and c (output vector) are HOST pointers as in the reference
 * (include/flash_blas.h:55-57). */
int bof_flash_csrgemv(char trans_a, uint64_t m, uint64_t n, bof_fptr a,
                      bof_fptr ia, bof_fptr ja, const float *b, float *c,
                      const bof_options *opts);
/* A contiguous file region <-> HBM through the pinned rings of the level-3 reader
 * (n_io_threads workers, 32 MiB chunks; O_DIRECT AIO where the region is sector
 * aligned).  Building blocks of the multi-GPU file path (bof_dist.py): with 288 GB
 * per GPU a rank's A / C row slabs and the whole of B are simply made resident.
 * Blocking.  The copies run on a private stream that is first ordered behind the work already
 * queued on `stream` (hipStream_t; NULL = the default stream), so a buffer that was just
 * allocated-and-filled or produced by kernels on `stream` can be passed without a host-side
 * synchronisation. */
int bof_file_to_device(bof_fptr f, uint64_t bytes, void *dptr, const bof_options *opts,
                       void *stream);
int bof_device_to_file(bof_fptr f, uint64_t bytes, const void *dptr, const bof_options *opts,
                       void *stream);
/* Counters of the last level-3 call (bytes moved per stage, seconds). */
typedef struct {
  uint64_t bytes_read, bytes_written; /* file I/O                      */
  uint64_t bytes_h2d, bytes_d2h;      /* PCIe                          */
  uint64_t tasks;                     /* tile tasks executed           */
  uint64_t tile_hits, tile_misses;    /* HBM tile cache                */
  double seconds;                     /* wall time of the call         */
  uint64_t read_ops, write_ops;       /* requests handed to the kernel
                                         (iocbs + pread/pwrite calls)  */
  uint64_t bytes_peer;                /* bytes of a shared operand taken from another rank's
                                         staging ring instead of the file (share_world > 1) */
  /* ---- ABI v4 ---- */
  uint64_t kernel_launches;           /* compute launches timed (bof_options.kernel_timing)  */
  double kernel_seconds;              /* their summed durations, HIP events on their streams */
  uint64_t bytes_p2p;                 /* bytes of a shared operand that reached a device from
                                         another device's HBM (bof_options.peer_bcast)       */
  uint64_t verify_checks;             /* hand-over checksums compared (bof_options.verify)   */
} bof_flash_stats;
int bof_flash_last_stats(bof_flash_stats *out);
/* The same counters per device of the last level-3 call, in the order of the device list
 * (bytes_read = what was read for this device alone; an operand read once for all devices is
 * only in the call's totals).  Returns the number of devices the call ran on (1 for a
 * single-device call) or a negative error; fills out[0 .. min(count, cap)). */
int bof_flash_last_device_stats(bof_flash_stats *out, int cap);
/* Dry run of bof_flash_gemm's schedule (task order, HBM tile-slot replacement, write-back)
 * for a budget of n_slots tile slots and a prefetch lookahead in tasks: fills the byte and
 * hit/miss counters the real call would report, without touching files or the GPU. */
int bof_flash_gemm_simulate(char ord, char trans_a, char trans_b, uint64_t m, uint64_t n,
                            uint64_t k, float beta, uint64_t lda, uint64_t ldb, uint64_t ldc,
                            int64_t blk, int64_t n_slots, int32_t lookahead,
                            bof_flash_stats *out);
/* Which path bof_flash_gemm takes for a problem and an HBM budget, and how the row-panel path
 * would lay the matrices out (pure host code: usable without a GPU).  Panels = `blk` stored
 * rows x the full stored width of a matrix, kept in HBM in file layout; "resident" matrices keep
 * all their panels, the others a ring of n_slots.  `group` = C panels of the first group (the
 * ramp that runs while the resident operand streams in; later panels go one at a time);
 * bof_flash_gemm picks it from the panel sizes (BOF_PANEL_GROUP overrides), here it is given. */
typedef struct {
  int32_t eligible;     /* 1: row panels; 0: the tile cache takes the call            */
  int32_t why;          /* 0 ok, 1 empty problem / k = 0, 2 ld < stored width, 3 an operand
                           is mostly gaps (width < ld / 2), 4 C's rows are not
                           contiguous in its file (ldc != stored width), 5 budget     */
  int32_t streamed;     /* the operand whose panels stream through a ring: 0 A, 1 B,
                           -1 none (both resident)                                    */
  int32_t resident[3];  /* A, B, C kept whole                                         */
  int64_t n_panels[3];  /* panels per matrix                                          */
  int64_t n_slots[3];   /* panel slots held in HBM                                    */
  uint64_t slot_bytes[3];
  uint64_t need_bytes;  /* HBM the plan needs                                         */
  int64_t groups;       /* outer iterations: the first group, then one C panel each   */
  int64_t first_group;  /* C panels of the first group (`group` clamped to the count) */
  /* ---- ABI v5 ---- */
  uint64_t acc_bytes;   /* what bof_flash_gemm needs ON TOP of need_bytes when beta != 0, K spans several blocks
                           and gemm_chain is 0 / 2: one raw accumulator panel (a C slot) per C panel of the first
                           group -- the chains of the ramp group carry their sums there while C still holds the
                           caller's values (bof_options.gemm_chain)                                          */
} bof_panel_plan;
int bof_flash_gemm_panel_plan(char ord, char trans_a, char trans_b, uint64_t m, uint64_t n,
                              uint64_t k, uint64_t lda, uint64_t ldb, uint64_t ldc, int64_t blk,
                              uint64_t hbm_budget, int64_t group, bof_panel_plan *out);
/* Level-3 calls keep their pinned staging rings and HBM tile slab between calls (the
 * reference keeps its program cache for the life of the process, src/lib_funcs.cpp:9);
 * this frees them (and unmaps the write mappings of buffered files). */
int bof_flash_release(void);
/* Removes the node-shared staging objects of a finished share_world > 1 call (every rank has
 * returned from it: the caller's barrier); harmless if they are gone already. */
int bof_share_cleanup(const char *share_name);
/* Diagnostic of the staging ring alone (pure host code, no GPU): `world` processes call this with the same
 * name and sizes and their own rank; chunk c is produced by rank c % world (a pattern of (c, position)) and
 * consumed and checked by all others, through a ring of n_slots slots.  Returns the number of chunks this
 * rank verified or a negative errno (-ETIMEDOUT when a peer never delivers).  Remove with
 * bof_share_cleanup. */
int64_t bof_share_selftest(const char *share_name, int rank, int world, int64_t n_chunks,
                           int64_t chunk_bytes, int n_slots, double timeout_s);

/* ---- instrumentation --------------------------------------------------------------------------
 * Event ring: every hand-over inside the level-3 pipelines (chunk read, H2D queued, panel ready,
 * group dispatched, D2H queued / complete, chunk written, staging-ring produce / consume, watchdog,
 * verify mismatch) is recorded -- always, it costs one atomic increment -- in a process-wide ring of
 * the last 4096 events.  bof_event_dump appends the ring to `path` (NULL = stderr), oldest first, times
 * in ms relative to the newest level-3 call's begin; the library does the same by itself when its
 * stall watchdog fires, when a BOF_VERIFY check fails, and after every level-3 call when
 * $BOF_EVENT_DUMP names a file.  Returns the number of events recorded so far. */
uint64_t bof_event_dump(const char *path);
/* How the last bof_flash_csrmm call treated its C FILE (additive to ABI v5).  Returns the mode: 1 = every row block's
 * region was sector-aligned, O_DIRECT as it is; 2 = unaligned row blocks of a row-major C (k = 100: 400-byte rows) kept
 * on O_DIRECT -- whole pages with O_DIRECT, the partial first / last page of a block through the page cache, where the
 * page two neighbouring blocks share is merged (the reference: sector read-modify-write with neighbour ordering,
 * src/file_handles/flash_file_handle.cpp:558-716, src/scheduler/io_executor.cpp:28-156); 0 = the buffered twin for every
 * request (column-major C with unaligned column pieces, $BOF_UNALIGNED_DIRECT=0); -1 = C was not an O_DIRECT file.
 * *twin_bytes (may be NULL): bytes of whole row blocks written through the buffered twin (0 unless the mode is 0). */
int bof_flash_last_c_file(uint64_t *twin_bytes);
/* The compute launches of the last bof_flash_gemm call that took the row-panel path, by kind (additive to ABI v5; all
 * devices of the call added up): out[0] = one k-range of a chain (the ramp group's k-block launches, the <ChainEpi>
 * instantiation of the tile kernel), out[1] = one launch over the whole K for a whole C panel, out[2] = one launch over
 * the whole K for a row slice of a C panel (the last panel of a slab leaves in slices).  bench.py names the kernels
 * behind roofline.achieved from this instead of assuming a schedule. */
int bof_flash_last_launch_mix(uint64_t out[3]);

/* File handle primitives (FlashFileHandle::read/write/sread/swrite,
 * src/file_handles/flash_file_handle.cpp:247-716) exposed for tests: strided
 * region {stride, n_strides, len_per_stride} in bytes <-> packed host buffer. */
int bof_file_sread(int fd, uint64_t offset, uint64_t stride, uint64_t n_strides,
                   uint64_t len_per_stride, void *buf, int use_aio);
int bof_file_swrite(int fd, uint64_t offset, uint64_t stride, uint64_t n_strides,
                    uint64_t len_per_stride, const void *buf, int use_aio);
/* Size of the requests a contiguous O_DIRECT transfer is cut into (all submitted together);
 * default 4 MiB or $BOF_IO_REQUEST_KIB.  Multiple of 512, at most 32 MiB (the reference's
 * MAX_CHUNK_SIZE, flash_file_handle.cpp:25). */
int bof_file_set_request_bytes(uint64_t bytes);
/* Unaligned requests on an O_DIRECT descriptor go through a cached buffered descriptor of
 * the same file, and large buffered writes into cached pages through a shared mapping of it
 * (BOF_MMAP_WRITES=0 turns the mapping off).  Both are keyed by the descriptor NUMBER: calling
 * this before close(fd) is MANDATORY for callers of the C ABI (FlashFileHandle::close does it;
 * bof_flash_release drops every mapping) -- otherwise the twin and the mapping stay alive (an
 * unlinked file keeps its blocks) until the number is reused or the library is released.  It
 * waits for stores that are in flight through the mapping.  A file must not be truncated by
 * anybody while a level-3 call writes it: a store into a mapped page beyond the new end raises
 * SIGBUS where pwrite would have re-extended the file. */
int bof_file_forget(int fd);

/* ---- synthetic inputs generated in HBM (bench / tests) ---------------------- */
/* misc/dense_create.cpp:28-37: mode 's' -> x[i] = (first+i) % 10, 'z' -> 0;
 * mode 'u' (ours) -> uniform [-1,1) from a counter-based hash of (seed, index). */
int bof_gen_dense(float *d, int64_t first, int64_t count, char mode, uint64_t seed,
                  void *stream);
/* misc/sparse_create.cpp:50-81 for rows [row0, row0+nrows): csr/col get
 * nrows*nnz_per_row entries, off gets nrows+1 (global offsets). */
int bof_gen_sparse_rows(int64_t row0, int64_t nrows, int64_t ncols,
                        int64_t nnz_per_row, float *csr, int64_t *col, int64_t *off,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BOF_HIP_H */
