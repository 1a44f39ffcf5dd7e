// file_handle.h -- storage-backend interface of the flash BLAS API.
//
// Source-compatible with the reference's include/file_handles/file_handle.h:13-73: the same
// type names (Mode, StrideInfo, BaseFileHandle, dummy_std_func), the same virtual member
// signatures, so code written against the reference compiles against this header.  In this
// implementation the hot path never goes through these virtuals (tiles travel file -> pinned
// ring -> HBM inside libbof_hip.so); they serve user code and the small helpers of lib_funcs.h.
#pragma once
#include <functional>
#include <string>

#include "bof_logger.h"
#include "bof_types.h"

namespace flash {
  enum class Mode { READ, WRITE, READWRITE };

  // completion callback of every I/O member (called once the transfer has finished)
  using IoCallback = std::function<void(void)>;
  extern IoCallback dummy_std_func;

  // A strided region, all three fields in BYTES: n_strides pieces of len_per_stride bytes whose
  // starts lie `stride` bytes apart (stride >= len_per_stride).  The memory side is packed.
  struct StrideInfo {
    FBLAS_UINT stride, n_strides, len_per_stride;
    operator std::string() const;  // "stride:n_strides:len_per_stride", used in log lines
  };
  inline StrideInfo::operator std::string() const {
    std::string out = std::to_string(stride);
    out += ':';
    out += std::to_string(n_strides);
    out += ':';
    return out + std::to_string(len_per_stride);
  }
  inline bool operator==(const StrideInfo& a, const StrideInfo& b) {
    return a.len_per_stride == b.len_per_stride && a.n_strides == b.n_strides && a.stride == b.stride;
  }

  class BaseFileHandle {
   public:
    virtual ~BaseFileHandle() = default;

    // strided transfers between the file (at `offset`) and a packed buffer / another handle
    virtual FBLAS_INT sread(FBLAS_UINT offset, StrideInfo sinfo, void* buf, const IoCallback& callback = dummy_std_func) = 0;
    virtual FBLAS_INT swrite(FBLAS_UINT offset, StrideInfo sinfo, void* buf, const IoCallback& callback = dummy_std_func) = 0;
    virtual FBLAS_INT scopy(FBLAS_UINT self_offset, BaseFileHandle& dest, FBLAS_UINT dest_offset, StrideInfo sinfo,
                            const IoCallback& callback = dummy_std_func) = 0;

    // contiguous transfers
    virtual FBLAS_INT read(FBLAS_UINT offset, FBLAS_UINT len, void* buf, const IoCallback& callback = dummy_std_func) = 0;
    virtual FBLAS_INT write(FBLAS_UINT offset, FBLAS_UINT len, void* buf, const IoCallback& callback = dummy_std_func) = 0;
    virtual FBLAS_INT copy(FBLAS_UINT self_offset, BaseFileHandle& dest, FBLAS_UINT dest_offset, FBLAS_UINT len,
                           const IoCallback& callback = dummy_std_func) = 0;

    // blocking; `size` > 0 creates / extends the file
    virtual FBLAS_INT open(std::string& fname, Mode fmode, FBLAS_UINT size = 0) = 0;
    virtual FBLAS_INT close() = 0;
  };
}  // namespace flash
