// bof_host_blas1.h -- the three host BLAS-1 names the reference's APPLICATION drivers use on
// mapped files around the flash kernels (drivers/kmeans.cpp:15-17, 98, 148-151: squared norms,
// nearest centre, centroid update).  The reference's include/bof_types.h:21-28 maps mkl_dot /
// mkl_axpy / mkl_imin to MKL's cblas_sdot / cblas_saxpy / cblas_isamin; this build has no MKL, so
// they are plain loops with the CBLAS argument meaning.  They are not part of the hot path: no
// flash:: kernel calls them.
#pragma once
#include <cmath>
#include <cstddef>

inline float bof_host_sdot(long long n, const float* x, long long incx, const float* y, long long incy) {
  double s = 0.0;
  for (long long i = 0; i < n; i++) s += (double) x[i * incx] * (double) y[i * incy];
  return (float) s;
}
inline void bof_host_saxpy(long long n, float a, const float* x, long long incx, float* y, long long incy) {
  for (long long i = 0; i < n; i++) y[i * incy] += a * x[i * incx];
}
// index of the element of smallest absolute value (first one on ties), as cblas_isamin
inline std::size_t bof_host_isamin(long long n, const float* x, long long incx) {
  std::size_t best = 0;
  float bv = n > 0 ? std::fabs(x[0]) : 0.f;
  for (long long i = 1; i < n; i++) {
    const float v = std::fabs(x[i * incx]);
    if (v < bv) { bv = v; best = (std::size_t) i; }
  }
  return best;
}
#define mkl_dot bof_host_sdot
#define mkl_axpy bof_host_saxpy
#define mkl_imin bof_host_isamin
