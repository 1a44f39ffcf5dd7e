// driver_util.h -- what the command-line drivers share: argument access with a usage message
// and a scope that maps the files named on the command line and unmaps them on exit.
#pragma once
#include <string>
#include <vector>

#include "bof_timer.h"
#include "bof_utils.h"
#include "flash_blas.h"
#include "lib_funcs.h"

namespace drv {
  // argv wrapper: positional arguments by index, typed
  class Args {
   public:
    Args(int argc, char** argv, int expected, const char* usage) : argv_(argv) {
      if (argc != expected + 1) GLOG_FATAL("usage : <exec> ", usage, "  (expected ", expected, " arguments, got ", argc - 1, ")");
    }
    std::string str(int i) const { return std::string(argv_[i]); }
    FBLAS_UINT u(int i) const { return std::stoull(argv_[i]); }
    FPTYPE f(int i) const { return std::stof(argv_[i]); }
    CHAR c(int i) const { return argv_[i][0]; }

   private:
    char** argv_;
  };

  // flash_setup on entry; every file mapped through it is unmapped, and the library torn down,
  // when it goes out of scope
  class Session {
   public:
    explicit Session(const std::string& mnt = "") { flash::flash_setup(mnt); }
    ~Session() {
      for (auto& undo : undo_) undo();
      flash::flash_destroy();
    }
    template<typename T>
    flash::flash_ptr<T> map(const std::string& path) {
      flash::flash_ptr<T> p = flash::map_file<T>(path, flash::Mode::READWRITE);
      undo_.push_back([p] { flash::unmap_file(p); });
      return p;
    }

   private:
    std::vector<std::function<void()>> undo_;
  };
}  // namespace drv
