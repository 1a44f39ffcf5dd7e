// lib_funcs.h -- library lifecycle and flash-memory helpers
// (reference include/lib_funcs.h:17-127, src/lib_funcs.cpp:7-33).
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
  extern std::string mnt_dir;
  extern Scheduler sched;
  extern Logger __global_logger;

  // Binds the calling thread to the library (selects the HIP device, honouring
  // BOF_DEVICE / LOCAL_RANK) and sets the directory flash_malloc creates files in.
  void flash_setup(std::string mntdir);
  void flash_destroy();

  template<typename T>
  flash_ptr<T> make_flash_ptr(T*, FBLAS_UINT) {
    throw std::runtime_error("make_flash_ptr() not implemented");
  }
  template<typename T>
  T* make_ptr(flash_ptr<T>) {
    throw std::runtime_error("make_ptr() not implemented");
  }

  template<typename T>
  void flash_memset(flash_ptr<T> fptr, int val, FBLAS_UINT n_bytes) {
    std::vector<char> buf(n_bytes, (char) val);
    fptr.fop->write(fptr.foffset, n_bytes, buf.data(), dummy_std_func);
  }

  template<typename T, typename W>
  void flash_memcpy(flash_ptr<T> dest, flash_ptr<W>& src, FBLAS_UINT n_bytes) {
    src.fop->copy(src.foffset, *dest.fop, dest.foffset, n_bytes, dummy_std_func);
  }

  // blocking element-count transfers between a flash_ptr and host memory
  template<typename T>
  FBLAS_INT read_sync(T* dest, flash_ptr<T> src, size_t len) {
    return src.fop->read(src.foffset, len * sizeof(T), dest, dummy_std_func);
  }
  template<typename T>
  FBLAS_INT write_sync(flash_ptr<T> dest, T* src, size_t len) {
    return dest.fop->write(dest.foffset, len * sizeof(T), src, dummy_std_func);
  }

  template<typename T>
  void flash_truncate(flash_ptr<T> fptr, uint64_t new_size) {
    FlashFileHandle* ffh = dynamic_cast<FlashFileHandle*>(fptr.fop);
    if (ffh == nullptr || ::ftruncate(ffh->file_desc, fptr.foffset + new_size) != 0)
      GLOG_ERROR("ftruncate failed with errno=", errno, ", error=", ::strerror(errno));
  }

  // scratch matrices backed by files under mnt_dir
  template<typename T>
  flash_ptr<T> flash_malloc(FBLAS_UINT n_bytes, std::string opt_name = "") {
    static unsigned long serial = 0;  // unlike the reference, two same-size allocations never collide
    n_bytes = ROUND_UP(n_bytes ? n_bytes : 1, 4096);
    std::string fname = mnt_dir + "tmp_" + (opt_name.empty() ? "" : opt_name + "_") +
                        std::to_string(n_bytes) + "_" + std::to_string(::getpid()) + "_" +
                        std::to_string(serial++);
    int fd = ::open(fname.c_str(), O_RDWR | O_CREAT, 00666);
    if (fd == -1 || ::ftruncate(fd, n_bytes) == -1) GLOG_FATAL("flash_malloc failed, errno=", errno);
    ::close(fd);
    return map_file<T>(fname, Mode::READWRITE);
  }
  template<typename T>
  void flash_free(flash_ptr<T> fptr) {
    std::string fname = static_cast<FlashFileHandle*>(fptr.fop)->get_filename();
    unmap_file<T>(fptr);
    ::remove(fname.c_str());
  }
}  // namespace flash
