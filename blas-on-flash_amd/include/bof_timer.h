// bof_timer.h -- wall-clock timer; elapsed() is in milliseconds like the
// reference's flash::Timer (include/bof_timer.h:8-27).
#pragma once
#include <chrono>

namespace flash {
  class Timer {
    std::chrono::steady_clock::time_point t0_ = std::chrono::steady_clock::now();

   public:
    void reset() { t0_ = std::chrono::steady_clock::now(); }
    // whole milliseconds since construction / reset()
    float elapsed() const {
      using namespace std::chrono;
      return (float) duration_cast<milliseconds>(steady_clock::now() - t0_).count();
    }
  };
}  // namespace flash
