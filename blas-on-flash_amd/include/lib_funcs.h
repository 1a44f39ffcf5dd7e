// lib_funcs.h -- library lifecycle plus the small flash-memory helpers user code expects
// (names and call shapes of the reference's include/lib_funcs.h:17-127 and
// src/lib_funcs.cpp:7-33; the bodies are this code base's own, built on two blocking
// byte-transfer primitives instead of one hand-written wrapper per helper).
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "bof_utils.h"
#include "file_handles/flash_file_handle.h"
#include "pointers/allocator.h"
#include "pointers/pointer.h"
#include "scheduler/scheduler.h"

namespace flash {
  // process-wide state of the library (defined in src/flash_api.cpp)
  extern Logger __global_logger;
  extern Scheduler sched;
  extern std::string mnt_dir;

  // flash_setup binds the calling thread to the library: it selects the HIP device
  // (BOF_DEVICE, else LOCAL_RANK, else 0; fatal without a GPU) and remembers the directory in
  // which flash_malloc creates its files.  flash_destroy releases the device-side caches.
  void flash_setup(std::string mntdir);
  void flash_destroy();

  namespace detail {
    // blocking byte transfers between host memory and the file position a flash_ptr names
    template<typename T>
    inline FBLAS_INT pull(void* host, const flash_ptr<T>& from, FBLAS_UINT n_bytes) {
      return from.fop->read(from.foffset, n_bytes, host, dummy_std_func);
    }
    template<typename T>
    inline FBLAS_INT push(const flash_ptr<T>& to, const void* host, FBLAS_UINT n_bytes) {
      return to.fop->write(to.foffset, n_bytes, const_cast<void*>(host), dummy_std_func);
    }
    inline FlashFileHandle* file_of(BaseFileHandle* h) { return dynamic_cast<FlashFileHandle*>(h); }
  }  // namespace detail

  // element-count transfers (blocking)
  template<typename T>
  FBLAS_INT read_sync(T* dest, flash_ptr<T> src, size_t len) {
    return detail::pull(dest, src, len * sizeof(T));
  }
  template<typename T>
  FBLAS_INT write_sync(flash_ptr<T> dest, T* src, size_t len) {
    return detail::push(dest, src, len * sizeof(T));
  }

  // byte-count fill / copy on flash memory
  template<typename T>
  void flash_memset(flash_ptr<T> fptr, int val, FBLAS_UINT n_bytes) {
    const std::vector<unsigned char> fill(n_bytes, static_cast<unsigned char>(val));
    detail::push(fptr, fill.data(), n_bytes);
  }
  template<typename T, typename W>
  void flash_memcpy(flash_ptr<T> dest, flash_ptr<W>& src, FBLAS_UINT n_bytes) {
    src.fop->copy(src.foffset, *dest.fop, dest.foffset, n_bytes, dummy_std_func);
  }

  // file size := offset of fptr + new_size
  template<typename T>
  void flash_truncate(flash_ptr<T> fptr, uint64_t new_size) {
    FlashFileHandle* fh = detail::file_of(fptr.fop);
    const bool ok = fh != nullptr && ::ftruncate(fh->file_desc, (off_t) (fptr.foffset + new_size)) == 0;
    if (!ok) GLOG_ERROR("flash_truncate: ftruncate failed, errno=", errno, " (", ::strerror(errno), ")");
  }

  // process-wide serial number of flash_malloc'ed files (one counter for every element type;
  // defined in src/flash_api.cpp)
  unsigned long next_flash_malloc_serial();

  // scratch arrays backed by files in mnt_dir; names carry the pid and a process-wide serial
  // number and the file is created exclusively, so two allocations never share a file (same-size
  // unnamed allocations collide in the reference)
  template<typename T>
  flash_ptr<T> flash_malloc(FBLAS_UINT n_bytes, std::string opt_name = "") {
    const FBLAS_UINT rounded = ROUND_UP(n_bytes == 0 ? 1 : n_bytes, 4096);
    std::string stem = mnt_dir + "tmp_";
    if (!opt_name.empty()) stem += opt_name + "_";
    stem += std::to_string(rounded) + "_" + std::to_string(::getpid()) + "_";
    std::string path;
    int fd = -1;
    for (int attempt = 0; attempt < 1000 && fd < 0; attempt++) {
      path = stem + std::to_string(next_flash_malloc_serial());
      fd = ::open(path.c_str(), O_CREAT | O_EXCL | O_RDWR, 0666);
      if (fd < 0 && errno != EEXIST) break;
    }
    const bool sized = fd >= 0 && ::ftruncate(fd, (off_t) rounded) == 0;
    if (fd >= 0) ::close(fd);
    if (!sized) GLOG_FATAL("flash_malloc(", path, ") failed, errno=", errno);
    return map_file<T>(path, Mode::READWRITE);
  }
  template<typename T>
  void flash_free(flash_ptr<T> fptr) {
    FlashFileHandle* fh = detail::file_of(fptr.fop);
    const std::string path = fh != nullptr ? fh->get_filename() : std::string();
    unmap_file<T>(fptr);
    if (!path.empty()) ::remove(path.c_str());
  }

  // declared by the reference, never implemented there either
  template<typename T>
  flash_ptr<T> make_flash_ptr(T*, FBLAS_UINT) {
    throw std::runtime_error("make_flash_ptr() not implemented");
  }
  template<typename T>
  T* make_ptr(flash_ptr<T>) {
    throw std::runtime_error("make_ptr() not implemented");
  }
}  // namespace flash
