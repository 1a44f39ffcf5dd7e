// sparse_create -- writes a CSR matrix as three raw files <name>csr (fp32 values), <name>col
// (int64 column indices), <name>off (int64 offsets) plus <name>info, same command line and the
// same deterministic content as the reference tool (misc/sparse_create.cpp:
// <name> <nrows> <ncols> <sparsity>): nnz_per_row = ceil(ncols*sparsity); row r draws
// nnz_per_row+40 columns from glibc's rand_r stream seeded with r, keeps the smallest
// nnz_per_row distinct ones; value of the i-th stored entry is (i % 9) + 1.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

// glibc rand_r: three LCG steps yielding 11 + 10 + 10 bits
static inline int64_t next_rand(uint32_t& s) {
  uint32_t out = 0;
  const int bits[3] = {11, 10, 10};
  for (int round = 0; round < 3; round++) {
    s = s * 1103515245u + 12345u;
    out = (out << bits[round]) ^ ((s >> 16) & ((1u << bits[round]) - 1u));
  }
  return (int64_t) out;
}

int main(int argc, char** argv) {
  if (argc < 5) {
    std::printf("usage : %s <name> <nrows> <ncols> <sparsity>\n", argv[0]);
    return 0;
  }
  const std::string name(argv[1]);
  const int64_t nrows = std::stoll(argv[2]), ncols = std::stoll(argv[3]);
  const double sparsity = std::stod(argv[4]);
  const int64_t per_row = (int64_t) std::ceil((double) ncols * sparsity);
  FILE* finfo = std::fopen((name + "info").c_str(), "w");
  FILE* fcsr = std::fopen((name + "csr").c_str(), "wb");
  FILE* fcol = std::fopen((name + "col").c_str(), "wb");
  FILE* foff = std::fopen((name + "off").c_str(), "wb");
  if (!finfo || !fcsr || !fcol || !foff) { std::perror("open"); return 1; }
  std::fprintf(finfo, "%lld %lld %g\n", (long long) nrows, (long long) ncols, sparsity);
  std::fclose(finfo);

  const int64_t rows_per_chunk = std::max<int64_t>(1, (1 << 22) / std::max<int64_t>(per_row, 1));
  std::vector<float> vals;
  std::vector<int64_t> cols, offs;
  for (int64_t r0 = 0; r0 < nrows; r0 += rows_per_chunk) {
    const int64_t nr = std::min(rows_per_chunk, nrows - r0);
    vals.resize((size_t) (nr * per_row));
    cols.resize((size_t) (nr * per_row));
    offs.resize((size_t) nr);
    bool short_row = false;
#pragma omp parallel for schedule(static)
    for (int64_t rr = 0; rr < nr; rr++) {
      const int64_t r = r0 + rr;
      uint32_t state = (uint32_t) r;
      std::vector<int64_t> cand((size_t) per_row + 40);
      for (auto& c : cand) {
        const int64_t lo = next_rand(state);   // low term is drawn first
        const int64_t hi = next_rand(state);
        c = (lo + hi * 2147483647LL) % ncols;
      }
      std::sort(cand.begin(), cand.end());
      cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
      if ((int64_t) cand.size() < per_row) { short_row = true; cand.resize((size_t) per_row, ncols - 1); }
      offs[(size_t) rr] = r * per_row;
      for (int64_t t = 0; t < per_row; t++) {
        cols[(size_t) (rr * per_row + t)] = cand[(size_t) t];
        vals[(size_t) (rr * per_row + t)] = (float) ((r * per_row + t) % 9 + 1);
      }
    }
    if (short_row) { std::fprintf(stderr, "a row had fewer than %lld distinct columns\n", (long long) per_row); return 1; }
    std::fwrite(vals.data(), sizeof(float), vals.size(), fcsr);
    std::fwrite(cols.data(), sizeof(int64_t), cols.size(), fcol);
    std::fwrite(offs.data(), sizeof(int64_t), offs.size(), foff);
  }
  const int64_t nnz = nrows * per_row;
  std::fwrite(&nnz, sizeof(int64_t), 1, foff);
  std::fclose(fcsr); std::fclose(fcol); std::fclose(foff);
  return 0;
}
