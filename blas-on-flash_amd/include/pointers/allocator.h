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
    // bytes of address space a mapping of `fh` from `foffset` on occupies (never 0)
    inline size_t map_span(const FlashFileHandle& fh, FBLAS_UINT foffset) {
      return fh.file_sz > foffset ? (size_t) (fh.file_sz - foffset) : (size_t) 1;
    }
  }  // namespace detail

  template<typename T>
  flash_ptr<T> map_file(std::string fname, Mode mode, FBLAS_UINT foffset = 0, int flags = 0) {
    GLOG_INFO("Mapping ", fname, ":", foffset, " to flash_ptr");
    FlashFileHandle* handle = new FlashFileHandle();
    handle->open(fname, mode);
    const size_t span = detail::map_span(*handle, foffset);
    const int prot = mode == Mode::READ ? PROT_READ : PROT_READ | PROT_WRITE;
    void* tag = ::mmap(nullptr, span, prot, flags | MAP_SHARED, handle->file_desc, 0);
    if (tag == MAP_FAILED)  // empty file or an unmappable one: address space is all we need
      tag = ::mmap(nullptr, span, PROT_NONE, MAP_ANONYMOUS | MAP_NORESERVE | MAP_PRIVATE, -1, 0);
    if (tag == MAP_FAILED) GLOG_FATAL("map_file(", fname, "): mmap failed: ", std::strerror(errno));
    return flash_ptr<T>(static_cast<T*>(tag), foffset, handle);
  }

  template<typename T>
  void unmap_file(flash_ptr<T> fptr) {
    FlashFileHandle* handle = dynamic_cast<FlashFileHandle*>(fptr.fop);
    if (handle == nullptr) return;  // not one of ours
    if (::munmap((void*) fptr.ptr, detail::map_span(*handle, fptr.foffset)) != 0)
      GLOG_ERROR("unmap_file: munmap failed: ", std::strerror(errno));
    delete handle;
  }
}  // namespace flash
