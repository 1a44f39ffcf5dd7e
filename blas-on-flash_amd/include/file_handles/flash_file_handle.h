// flash_file_handle.h -- file-backed BaseFileHandle (reference
// include/file_handles/flash_file_handle.h:21-76).  I/O goes through
// libbof_hip.so's reader (bof_file_sread/bof_file_swrite): kernel AIO on an
// O_DIRECT descriptor for sector-aligned requests, buffered otherwise.
#pragma once
#include <string>

#include "bof_types.h"
#include "file_handles/file_handle.h"

namespace flash {
  class FlashFileHandle : public BaseFileHandle {
    std::string filename;

   public:
    FBLAS_UINT file_sz;
    int file_desc;

    FlashFileHandle();
    ~FlashFileHandle();

    // AIO contexts are created lazily per calling thread; kept for API compatibility
    static void register_thread();
    static void deregister_thread();

    std::string get_filename() { return this->filename; }

    FBLAS_INT open(std::string& fname, Mode fmode, FBLAS_UINT size = 0);
    FBLAS_INT close();
    FBLAS_INT read(FBLAS_UINT offset, FBLAS_UINT len, void* buf,
                   const std::function<void(void)>& callback = dummy_std_func);
    FBLAS_INT write(FBLAS_UINT offset, FBLAS_UINT len, void* buf,
                    const std::function<void(void)>& callback = dummy_std_func);
    FBLAS_INT copy(FBLAS_UINT self_offset, BaseFileHandle& dest, FBLAS_UINT dest_offset,
                   FBLAS_UINT len, const std::function<void(void)>& callback = dummy_std_func);
    FBLAS_INT sread(FBLAS_UINT offset, StrideInfo sinfo, void* buf,
                    const std::function<void(void)>& callback = dummy_std_func);
    FBLAS_INT swrite(FBLAS_UINT offset, StrideInfo sinfo, void* buf,
                     const std::function<void(void)>& callback = dummy_std_func);
    FBLAS_INT scopy(FBLAS_UINT self_offset, BaseFileHandle& dest, FBLAS_UINT dest_offset,
                    StrideInfo sinfo, const std::function<void(void)>& callback = dummy_std_func);
  };
}  // namespace flash
