// bof_types.h -- scalar types of the flash BLAS API (MI355X-native build).
// Same names as the reference's include/bof_types.h:11-28, but no dependency on
// mkl.h: MKL_INT is the 64-bit integer the reference gets from -DMKL_ILP64
// (CMakeLists.txt:104), which is also the on-disk type of CSR index/offset files.
#pragma once

#include <cassert>
#include <cfloat>
#include <cstdint>

#define FP_SINGLE_PRECISION

typedef int64_t FBLAS_INT;
typedef uint64_t FBLAS_UINT;
typedef char CHAR;
typedef float FPTYPE;
typedef double LONGFPTYPE;
#define FPTYPE_MAX FLT_MAX

#ifndef MKL_INT
#define MKL_INT long long
#endif
static_assert(sizeof(MKL_INT) == 8, "CSR index/offset files are int64");

// mkl_dot / mkl_axpy / mkl_imin of the reference's bof_types.h:23-28, for its application drivers
#include "bof_host_blas1.h"

// Build-time macro contract of the reference (CMakeLists.txt:38-91): defaults are
// supplied here so drivers compile without -D flags.  The tile-size macros only
// seed the run-time options (see scheduler/scheduler.h); they can be overridden
// per process with BOF_* environment variables.
#ifndef SECTOR_LEN
#define SECTOR_LEN 512
#endif
#ifndef IS_ALIGNED
#define IS_ALIGNED IS_512_ALIGNED
#endif
#ifndef GEMM_BLK_SIZE
#define GEMM_BLK_SIZE 4096
#endif
#ifndef MAX_NNZS
#define MAX_NNZS 10000000
#endif
#ifndef CSRMM_RM_RBLK_SIZE
#define CSRMM_RM_RBLK_SIZE 131072
#endif
#ifndef CSRMM_RM_CBLK_SIZE
#define CSRMM_RM_CBLK_SIZE 1024
#endif
#ifndef N_IO_THR
#define N_IO_THR 4
#endif
#ifndef N_COMPUTE_THR
#define N_COMPUTE_THR 4
#endif
#ifndef PROGRAM_BUDGET
#define PROGRAM_BUDGET 0  /* 0 = 80% of free HBM */
#endif
