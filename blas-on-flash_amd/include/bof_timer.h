// bof_timer.h -- wall-clock stopwatch with the interface of the reference's flash::Timer
// (include/bof_timer.h:8-27): elapsed() in milliseconds, reset().
#pragma once
#include <chrono>

namespace flash {
  class Timer {
    using clock = std::chrono::steady_clock;
    clock::time_point origin_{clock::now()};

   public:
    Timer() = default;
    // whole milliseconds since construction or the last reset()
    float elapsed() const {
      const auto span = clock::now() - origin_;
      return static_cast<float>(std::chrono::duration_cast<std::chrono::milliseconds>(span).count());
    }
    void reset() { origin_ = clock::now(); }
  };
}  // namespace flash
