// scheduler.h -- the run-time knobs of flash::sched (reference
// include/scheduler/scheduler.h:70-83,124-141).  In this build the task state
// machine lives inside libbof_hip.so (event-driven: hipEvents + condition
// variables, HBM tile cache with farthest-next-use eviction); this class only
// carries the options a caller may change between kernel calls.
#pragma once
#include "bof_types.h"

namespace flash {
  struct SchedulerOptions {
    bool enable_prioritizer = true;     // accepted; task order is static + Belady here
    bool enable_overlap_check = true;   // accepted; unaligned writes go through the page cache
    bool single_use_discard = false;    // accepted; eviction already knows every future use
  };

  class Scheduler {
   public:
    // reference ctor shape: (n_io_threads, n_compute_threads, program budget in bytes)
    Scheduler(FBLAS_UINT n_io_thr, FBLAS_UINT n_compute_thr, FBLAS_UINT max_mem);

    void set_options(SchedulerOptions& sched_opts) { opts_ = sched_opts; }
    void set_num_compute_threads(FBLAS_UINT n_thr) { n_compute_ = n_thr ? n_thr : 1; }
    // every kernel call writes its dirty tiles back before returning, so there is
    // never anything left to flush
    void flush_cache() {}

    // tunables (compile-time macros in the reference, CMakeLists.txt:38-63)
    FBLAS_UINT n_io_threads() const { return n_io_; }
    FBLAS_UINT n_compute_threads() const { return n_compute_; }
    FBLAS_UINT program_budget() const { return budget_; }
    FBLAS_UINT gemm_blk_size = GEMM_BLK_SIZE;
    FBLAS_UINT max_nnzs = MAX_NNZS;
    FBLAS_UINT csrmm_rblk_size = CSRMM_RM_RBLK_SIZE;
    FBLAS_UINT csrmm_cblk_size = CSRMM_RM_CBLK_SIZE;
    bool use_odirect = true;

   private:
    FBLAS_UINT n_io_, n_compute_, budget_;
    SchedulerOptions opts_;
  };
}  // namespace flash
