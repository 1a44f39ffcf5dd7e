// bof_utils.h -- alignment helpers of the flash BLAS API
// (reference include/bof_utils.h:12-20, 25-38, 40-44; src/utils.cpp:13-53).
#pragma once
#include <unistd.h>
#include <cstdint>
#include <cstdlib>

#include "bof_logger.h"
#include "bof_types.h"
#include "file_handles/file_handle.h"

#define ROUND_UP(X, Y) ((((uint64_t)(X) + (uint64_t)(Y) - 1) / (uint64_t)(Y)) * (uint64_t)(Y))
#define ROUND_DOWN(X, Y) (((uint64_t)(X) / (uint64_t)(Y)) * (uint64_t)(Y))
#define IS_512_ALIGNED(X) (((uint64_t)(X) & 511u) == 0)
#define IS_4096_ALIGNED(X) (((uint64_t)(X) & 4095u) == 0)

namespace flash {
  // sector-aligned host allocation (size must itself be sector aligned)
  void alloc_aligned(void** ptr, size_t size, size_t align = SECTOR_LEN);
  uint32_t fnv32a(const char* str, const uint32_t n_bytes);
  uint64_t fnv64a(const char* str, const uint64_t n_bytes);
  // bytes a packed buffer for `sinfo` needs (contiguous regions get one spare sector)
  FBLAS_UINT buf_size(const StrideInfo sinfo);

  template<typename T>
  inline T* offset_buf(T* buf, FBLAS_UINT offset) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(buf) + offset);
  }
}  // namespace flash
