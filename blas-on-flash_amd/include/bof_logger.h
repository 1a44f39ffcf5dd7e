// bof_logger.h -- flash::Logger and the LOG_*/GLOG_* macro family the drivers use
// (same macro names and call shapes as the reference's include/bof_logger.h:9-59;
// one mutex-serialised line per call, `fatal` terminates the process with -1).
#pragma once

#include <cstdlib>
#include <ctime>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>

namespace flash {
  class Logger {
    std::string name_;
    std::mutex mu_;

    static void put(std::ostringstream &) {}
    template<typename T, typename... Rest>
    static void put(std::ostringstream &os, const T &v, const Rest &... rest) {
      os << v;
      put(os, rest...);
    }

    // tag, ANSI colour of the header / of the message
    template<typename F, typename L, typename... Args>
    void emit(const char *tag, const char *hdr_col, const char *msg_col, F func, L line,
              const Args &... args) {
      std::ostringstream os;
      char stamp[64];
      const std::time_t now = std::time(nullptr);
      std::tm tmv;
      localtime_r(&now, &tmv);
      std::strftime(stamp, sizeof(stamp), "%d/%m/%Y|%H:%M:%S", &tmv);
      os << "\033[" << hdr_col << "m[" << tag << "][" << stamp << "][" << name_
         << "][thread:" << std::this_thread::get_id() << "]:" << func << ":" << line << ":"
         << "\033[" << msg_col << "m";
      put(os, args...);
      os << "\033[0m\n";
      std::lock_guard<std::mutex> lk(mu_);
      std::cout << os.str() << std::flush;
    }

   public:
    explicit Logger(std::string name) : name_(std::move(name)) {}

#define BOF_LOG_METHOD(method, tag, hdr, msg)                          \
  template<typename F, typename L, typename... Args>                   \
  void method(F func, L line, const Args &... args) {                  \
    emit(tag, hdr, msg, func, line, args...);                          \
  }
    BOF_LOG_METHOD(info, "info", "1;37;40", "0;37;40")
    BOF_LOG_METHOD(debug, "dbg", "1;36;40", "0;36;40")
    BOF_LOG_METHOD(error, "err", "1;31;40", "0;31;40")
    BOF_LOG_METHOD(fail, "fail", "1;31;40", "0;31;40")
    BOF_LOG_METHOD(pass, "pass", "1;32;40", "0;32;40")
    BOF_LOG_METHOD(warn, "warn", "1;33;40", "0;33;40")
#undef BOF_LOG_METHOD
    template<typename F, typename L, typename... Args>
    [[noreturn]] void fatal(F func, L line, const Args &... args) {
      emit("fatal", "1;37;41", "0;37;41", func, line, args...);
      std::exit(-1);
    }
  };

  extern Logger __global_logger;
}  // namespace flash

#define LOG_INFO(lg, ...) (lg).info(__func__, __LINE__, __VA_ARGS__)
#define LOG_ERROR(lg, ...) (lg).error(__func__, __LINE__, __VA_ARGS__)
#define LOG_WARN(lg, ...) (lg).warn(__func__, __LINE__, __VA_ARGS__)
#define LOG_FATAL(lg, ...) (lg).fatal(__func__, __LINE__, __VA_ARGS__)
#define LOG_PASS(lg, ...) (lg).pass(__func__, __LINE__, __VA_ARGS__)
#define LOG_FAIL(lg, ...) (lg).fail(__func__, __LINE__, __VA_ARGS__)

// Debug logging and assertions exist only in -DDEBUG builds, as in the reference.
#ifdef DEBUG
#define LOG_DEBUG(lg, ...) (lg).debug(__func__, __LINE__, __VA_ARGS__)
#define LOG_ASSERT(lg, cond, ...) \
  do { if (!(cond)) (lg).fatal(__func__, __LINE__, "assert:(", #cond, ") failed: ", __VA_ARGS__); } while (0)
#define LOG_ASSERT_LE(lg, a, b) LOG_ASSERT(lg, (a) <= (b), "expected ", #a, "<=", (b), ", got ", #a, "=", (a))
#define LOG_ASSERT_LT(lg, a, b) LOG_ASSERT(lg, (a) < (b), "expected ", #a, "<", (b), ", got ", #a, "=", (a))
#define LOG_ASSERT_EQ(lg, a, b) LOG_ASSERT(lg, (a) == (b), "expected ", #a, "=", (b), ", got ", #a, "=", (a))
#define LOG_ASSERT_NOT_NULL(lg, p) LOG_ASSERT(lg, (p) != nullptr, " expected non-nullptr, got nullptr")
#else
#define LOG_DEBUG(lg, ...)
#define LOG_ASSERT(lg, cond, ...)
#define LOG_ASSERT_LE(lg, a, b)
#define LOG_ASSERT_LT(lg, a, b)
#define LOG_ASSERT_EQ(lg, a, b)
#define LOG_ASSERT_NOT_NULL(lg, p)
#endif

#define GLOG_INFO(...) LOG_INFO(flash::__global_logger, __VA_ARGS__)
#define GLOG_DEBUG(...) LOG_DEBUG(flash::__global_logger, __VA_ARGS__)
#define GLOG_ERROR(...) LOG_ERROR(flash::__global_logger, __VA_ARGS__)
#define GLOG_WARN(...) LOG_WARN(flash::__global_logger, __VA_ARGS__)
#define GLOG_FATAL(...) LOG_FATAL(flash::__global_logger, __VA_ARGS__)
#define GLOG_PASS(...) LOG_PASS(flash::__global_logger, __VA_ARGS__)
#define GLOG_FAIL(...) LOG_FAIL(flash::__global_logger, __VA_ARGS__)
#define GLOG_ASSERT(...) LOG_ASSERT(flash::__global_logger, __VA_ARGS__)
#define GLOG_ASSERT_LE(...) LOG_ASSERT_LE(flash::__global_logger, __VA_ARGS__)
#define GLOG_ASSERT_LT(...) LOG_ASSERT_LT(flash::__global_logger, __VA_ARGS__)
#define GLOG_ASSERT_EQ(...) LOG_ASSERT_EQ(flash::__global_logger, __VA_ARGS__)
#define GLOG_ASSERT_NOT_NULL(...) LOG_ASSERT_NOT_NULL(flash::__global_logger, __VA_ARGS__)
