// flash_file_handle.h -- the file-backed BaseFileHandle (reference
// include/file_handles/flash_file_handle.h:21-76: same class name, same public members
// file_sz / file_desc / get_filename / register_thread / deregister_thread).
// Transfers go through libbof_hip.so's reader (bof_file_sread / bof_file_swrite): kernel AIO on
// an O_DIRECT descriptor for sector-aligned requests, the page cache for everything else.
#pragma once
#include <string>

#include "bof_types.h"
#include "file_handles/file_handle.h"

namespace flash {
  class FlashFileHandle : public BaseFileHandle {
   public:
    FBLAS_UINT file_sz;  // bytes, as found (or created) by open()
    int file_desc;       // what libbof_hip.so's level-3 entry points take (bof_fptr.fd)

    FlashFileHandle();
    ~FlashFileHandle() override;

    std::string get_filename() { return filename; }

    // The reference needs every I/O thread to create an AIO context first; here contexts are
    // made lazily per calling thread, so these two are accepted and do nothing.
    static void register_thread();
    static void deregister_thread();

    FBLAS_INT sread(FBLAS_UINT offset, StrideInfo sinfo, void* buf, const IoCallback& callback = dummy_std_func) override;
    FBLAS_INT swrite(FBLAS_UINT offset, StrideInfo sinfo, void* buf, const IoCallback& callback = dummy_std_func) override;
    FBLAS_INT scopy(FBLAS_UINT self_offset, BaseFileHandle& dest, FBLAS_UINT dest_offset, StrideInfo sinfo,
                    const IoCallback& callback = dummy_std_func) override;
    FBLAS_INT read(FBLAS_UINT offset, FBLAS_UINT len, void* buf, const IoCallback& callback = dummy_std_func) override;
    FBLAS_INT write(FBLAS_UINT offset, FBLAS_UINT len, void* buf, const IoCallback& callback = dummy_std_func) override;
    FBLAS_INT copy(FBLAS_UINT self_offset, BaseFileHandle& dest, FBLAS_UINT dest_offset, FBLAS_UINT len,
                   const IoCallback& callback = dummy_std_func) override;
    FBLAS_INT open(std::string& fname, Mode fmode, FBLAS_UINT size = 0) override;
    FBLAS_INT close() override;

   private:
    std::string filename;
  };
}  // namespace flash
