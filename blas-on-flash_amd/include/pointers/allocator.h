// allocator.h -- map_file / unmap_file: the pair that turns a file name into a flash_ptr<T>
// and back (call shapes of the reference's include/pointers/allocator.h:19-59).
//
// Ownership: map_file heap-allocates a FlashFileHandle and hands it out through the
// (non-owning) flash_ptr; unmap_file deletes it.  The mmap only provides what the API
// promises about flash_ptr::ptr -- a distinct address per file byte; this implementation never
// reads file data through it (tiles travel file -> pinned ring -> HBM).
#pragma once

#include <sys/mman.h>

#include <cerrno>
#include <cstring>
#include <fstream>  // user code written against the reference gets <fstream> through this header
#include <string>

#include "bof_logger.h"
#include "bof_types.h"
#include "file_handles/flash_file_handle.h"
#include "pointers/pointer.h"

namespace flash {
  namespace detail {
    struct FileView {
      void* address;             // first byte of the view (distinct per file byte, see above)
      FlashFileHandle* owner;    // heap object; unmap_file deletes it
    };
    // address space a view of `fh` starting `skip` bytes into the file occupies (never 0)
    inline size_t view_bytes(const FlashFileHandle& fh, FBLAS_UINT skip) {
      return fh.file_sz > skip ? (size_t) (fh.file_sz - skip) : (size_t) 1;
    }
    inline FileView open_view(std::string path, Mode how, FBLAS_UINT skip, int extra_flags) {
      FileView v{nullptr, new FlashFileHandle()};
      v.owner->open(path, how);
      const size_t len = view_bytes(*v.owner, skip);
      v.address = ::mmap(nullptr, len, how == Mode::READ ? PROT_READ : (PROT_READ | PROT_WRITE),
                         MAP_SHARED | extra_flags, v.owner->file_desc, 0);
      if (v.address == MAP_FAILED)  // empty or unmappable file: a reservation of addresses is enough
        v.address = ::mmap(nullptr, len, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (v.address == MAP_FAILED) GLOG_FATAL("map_file(", path, "): no address space: ", std::strerror(errno));
      return v;
    }
    inline void close_view(BaseFileHandle* any, void* address, FBLAS_UINT skip) {
      FlashFileHandle* owner = dynamic_cast<FlashFileHandle*>(any);
      if (owner == nullptr) return;  // not produced by map_file
      if (::munmap(address, view_bytes(*owner, skip)) != 0)
        GLOG_ERROR("unmap_file: munmap: ", std::strerror(errno));
      delete owner;
    }
  }  // namespace detail

  template<typename T>
  flash_ptr<T> map_file(std::string fname, Mode mode, FBLAS_UINT foffset = 0, int flags = 0) {
    GLOG_INFO("Mapping ", fname, ":", foffset, " to flash_ptr");
    const detail::FileView v = detail::open_view(fname, mode, foffset, flags);
    return flash_ptr<T>(static_cast<T*>(v.address), foffset, v.owner);
  }

  template<typename T>
  void unmap_file(flash_ptr<T> fptr) {
    detail::close_view(fptr.fop, (void*) fptr.ptr, fptr.foffset);
  }
}  // namespace flash
