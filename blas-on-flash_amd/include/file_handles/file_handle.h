// file_handle.h -- storage-backend interface of the flash BLAS API: Mode,
// StrideInfo (all fields in bytes) and BaseFileHandle, with the member names and
// signatures of the reference's include/file_handles/file_handle.h:13-73.
#pragma once
#include <functional>
#include <string>

#include "bof_logger.h"
#include "bof_types.h"

namespace flash {
  enum class Mode { READ, WRITE, READWRITE };

  // n_strides pieces of len_per_stride bytes, consecutive pieces `stride` bytes apart
  struct StrideInfo {
    FBLAS_UINT stride;
    FBLAS_UINT n_strides;
    FBLAS_UINT len_per_stride;

    operator std::string() const {
      return std::to_string(stride) + ":" + std::to_string(n_strides) + ":" +
             std::to_string(len_per_stride);
    }
    bool operator==(const StrideInfo& o) const {
      return stride == o.stride && n_strides == o.n_strides && len_per_stride == o.len_per_stride;
    }
  };

  extern std::function<void(void)> dummy_std_func;

  class BaseFileHandle {
   public:
    virtual ~BaseFileHandle() {}
    virtual FBLAS_INT open(std::string& fname, Mode fmode, FBLAS_UINT size = 0) = 0;
    virtual FBLAS_INT close() = 0;

    // contiguous
    virtual FBLAS_INT read(FBLAS_UINT offset, FBLAS_UINT len, void* buf,
                           const std::function<void(void)>& callback = dummy_std_func) = 0;
    virtual FBLAS_INT write(FBLAS_UINT offset, FBLAS_UINT len, void* buf,
                            const std::function<void(void)>& callback = dummy_std_func) = 0;
    virtual FBLAS_INT copy(FBLAS_UINT self_offset, BaseFileHandle& dest, FBLAS_UINT dest_offset,
                           FBLAS_UINT len,
                           const std::function<void(void)>& callback = dummy_std_func) = 0;
    // strided (memory side packed)
    virtual FBLAS_INT sread(FBLAS_UINT offset, StrideInfo sinfo, void* buf,
                            const std::function<void(void)>& callback = dummy_std_func) = 0;
    virtual FBLAS_INT swrite(FBLAS_UINT offset, StrideInfo sinfo, void* buf,
                             const std::function<void(void)>& callback = dummy_std_func) = 0;
    virtual FBLAS_INT scopy(FBLAS_UINT self_offset, BaseFileHandle& dest, FBLAS_UINT dest_offset,
                            StrideInfo sinfo,
                            const std::function<void(void)>& callback = dummy_std_func) = 0;
  };
}  // namespace flash
