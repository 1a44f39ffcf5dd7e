// allocator.h -- map_file / unmap_file: bind a file to a flash_ptr<T>
// (reference include/pointers/allocator.h:19-59).  map_file heap-allocates the
// FlashFileHandle and unmap_file deletes it; the flash_ptr itself is non-owning.
#pragma once

#include <sys/mman.h>
#include <cerrno>
#include <fstream>
#include <cstring>
#include <string>

#include "bof_logger.h"
#include "bof_types.h"
#include "file_handles/flash_file_handle.h"
#include "pointers/pointer.h"

namespace flash {
  template<typename T>
  flash_ptr<T> map_file(std::string fname, Mode mode, FBLAS_UINT foffset = 0, int flags = 0) {
    GLOG_INFO("Mapping ", fname, ":", foffset, " to flash_ptr");
    FlashFileHandle* fh = new FlashFileHandle();
    fh->open(fname, mode);
    // The mapping gives every byte of the file a distinct address tag; the hot path
    // never dereferences it (tiles travel file -> pinned ring -> HBM).
    const size_t span = fh->file_sz > foffset ? fh->file_sz - foffset : 1;
    const int prot = (mode == Mode::READ) ? PROT_READ : (PROT_READ | PROT_WRITE);
    void* addr = ::mmap(nullptr, span, prot, MAP_SHARED | flags, fh->file_desc, 0);
    if (addr == MAP_FAILED)  // e.g. empty file: reserve address space only
      addr = ::mmap(nullptr, span, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (addr == MAP_FAILED) GLOG_FATAL("mmap failed with error ", std::strerror(errno));
    return flash_ptr<T>(static_cast<T*>(addr), foffset, fh);
  }

  template<typename T>
  void unmap_file(flash_ptr<T> fptr) {
    FlashFileHandle* fh = dynamic_cast<FlashFileHandle*>(fptr.fop);
    if (fh == nullptr) return;
    const size_t span = fh->file_sz > fptr.foffset ? fh->file_sz - fptr.foffset : 1;
    if (::munmap((void*) fptr.ptr, span) != 0)
      GLOG_ERROR("munmap failed with error ", std::strerror(errno));
    delete fh;
  }
}  // namespace flash
