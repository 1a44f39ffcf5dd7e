// in_mem_util.h -- shared by the in_mem_* drivers (the analogues of the reference's drivers/in_mem_gemm.cpp,
// in_mem_csrmm.cpp, in_mem_csrgemv.cpp): whole files into HBM, ONE whole-matrix call, C back into its file.
// "Memory" here is the GPU's: the counterpart of the reference's single MKL call on host arrays is a single
// level-1 / level-2 call on device arrays.  No flash_setup: these binaries never touch the level-3 runtime
// (no scheduler, no program cache), exactly as the reference's in-memory drivers bypass libfblas.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "bof_hip.h"
#include "bof_timer.h"
#include "bof_types.h"
#include "bof_utils.h"

namespace inmem {
  inline void must(int rc, const char* what) {
    if (rc != BOF_OK) GLOG_FATAL(what, " failed: ", bof_last_error());
  }
  // a device array filled from the first `bytes` bytes of a file (short files: the rest stays zero,
  // as the reference's ifstream::read into a fresh array leaves it unspecified)
  struct DeviceArray {
    void* d = nullptr;
    uint64_t bytes = 0;
    DeviceArray() = default;
    DeviceArray(const DeviceArray&) = delete;
    DeviceArray& operator=(const DeviceArray&) = delete;
    ~DeviceArray() { if (d) bof_free(d); }
    void alloc(uint64_t n) {
      bytes = n;
      must(bof_malloc(&d, std::max<uint64_t>(n, 4)), "bof_malloc");
      must(bof_memset(d, 0, std::max<uint64_t>(n, 4), nullptr), "bof_memset");
    }
    void load(const std::string& path, uint64_t n) {
      alloc(n);
      const int fd = ::open(path.c_str(), O_RDONLY);
      if (fd < 0) GLOG_FATAL("cannot open ", path);
      struct stat sb;
      ::fstat(fd, &sb);
      const uint64_t take = std::min<uint64_t>(n, (uint64_t) sb.st_size);
      bof_options o;
      bof_default_options(&o);
      o.use_odirect = 0;
      bof_fptr f{fd, 0};
      if (take) must(bof_file_to_device(f, take, d, &o, nullptr), "bof_file_to_device");
      bof_file_forget(fd);
      ::close(fd);
    }
    void store(const std::string& path, uint64_t n) const {
      const int fd = ::open(path.c_str(), O_RDWR | O_CREAT, 0644);
      if (fd < 0) GLOG_FATAL("cannot open ", path, " for writing");
      if (::ftruncate(fd, (off_t) n)) GLOG_FATAL("cannot size ", path);
      bof_options o;
      bof_default_options(&o);
      o.use_odirect = 0;
      bof_fptr f{fd, 0};
      if (n) must(bof_device_to_file(f, n, d, &o, nullptr), "bof_device_to_file");
      bof_file_forget(fd);
      ::close(fd);
    }
    template<typename T>
    T* as() const { return static_cast<T*>(d); }
  };
  inline std::vector<MKL_INT> load_offsets(const std::string& path, FBLAS_UINT rows) {
    std::vector<MKL_INT> v(rows + 1, 0);
    FILE* f = ::fopen(path.c_str(), "rb");
    if (!f) GLOG_FATAL("cannot open ", path);
    if (::fread(v.data(), sizeof(MKL_INT), rows + 1, f) != rows + 1) GLOG_FATAL("short offsets file ", path);
    ::fclose(f);
    return v;
  }
  inline void need_gpu() {
    if (bof_device_count() <= 0) GLOG_FATAL("no HIP device: the in-memory drivers compute on the GPU (there is no CPU fallback)");
  }
}  // namespace inmem
