// dense_create -- writes a raw fp32 row-major matrix file, same command line and fill modes as
// the reference tool (misc/dense_create.cpp: <filename> <nrows> <ncols> <fill_mode>):
//   s : x[i] = i % 10        z : zeros
//   r : (i + rand) % 10 in the reference, where the generator state is raced between OpenMP
//       threads and therefore not reproducible; here 'r' is a deterministic counter hash
//       with the same value range.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static inline uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

int main(int argc, char** argv) {
  if (argc < 5) {
    std::printf("usage : %s <filename> <nrows> <ncols> <fill_mode>\n", argv[0]);
    return 0;
  }
  const int64_t total = std::stoll(argv[2]) * std::stoll(argv[3]);
  const char mode = argv[4][0];
  FILE* f = std::fopen(argv[1], "wb");
  if (!f) { std::perror(argv[1]); return 1; }
  const int64_t chunk = 1 << 22;
  std::vector<float> buf((size_t) std::min<int64_t>(chunk, total > 0 ? total : 1));
  for (int64_t base = 0; base < total; base += chunk) {
    const int64_t cnt = std::min(chunk, total - base);
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < cnt; t++) {
      const int64_t i = base + t;
      buf[(size_t) t] = mode == 's' ? (float) (i % 10)
                       : mode == 'r' ? (float) ((i + (int64_t) (mix((uint64_t) i) >> 33)) % 10) : 0.0f;
    }
    if (std::fwrite(buf.data(), sizeof(float), (size_t) cnt, f) != (size_t) cnt) { std::perror("write"); return 1; }
  }
  std::fclose(f);
  return 0;
}
