// How long does hipMalloc take per GiB, and do concurrent hipMallocs overlap?  (first-call cost of the
// level-3 panel slabs: 0.85 s for 37 GiB at 65536^3)
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  const size_t G = 1ull << 30;
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    void *p[3];
    for (int i = 0; i < 3; i++) hipMalloc(&p[i], 12 * G);
    double t1 = now();
    printf("sequential 3 x 12 GiB: %.3f s\n", t1 - t0);
    for (int i = 0; i < 3; i++) hipFree(p[i]);
    double t2 = now();
    printf("  free: %.3f s\n", t2 - t1);
    std::vector<std::thread> th;
    t0 = now();
    for (int i = 0; i < 3; i++) th.emplace_back([&, i] { hipSetDevice(0); hipMalloc(&p[i], 12 * G); });
    for (auto &t : th) t.join();
    t1 = now();
    printf("3 threads x 12 GiB: %.3f s\n", t1 - t0);
    for (int i = 0; i < 3; i++) hipFree(p[i]);
    t0 = now();
    std::vector<void *> q(36);
    for (int i = 0; i < 36; i++) hipMalloc(&q[i], G);
    t1 = now();
    printf("36 x 1 GiB sequential: %.3f s\n", t1 - t0);
    for (auto x : q) hipFree(x);
  }
  return 0;
}
