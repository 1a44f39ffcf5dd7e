// plan.cpp -- host-side tilers of the hot path (pure C++, no GPU calls).
//   GEMM   : 3-D tiling + tail-merge rule + k accumulate chains
//            (reference src/blas/gemm.cpp:39-129)
//   CSR    : nnz-budget row blocks (reference include/blas_utils.h:72-97; App. D-2)
#include <algorithm>
#include <cstdint>

#include "bof_hip.h"
#include "bof_internal.h"

namespace bof {

static constexpr int64_t kSectorFloats = 512 / sizeof(float);  // SECTOR_LEN / sizeof(FPTYPE)

GemmGeometry gemm_geometry(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
                           int64_t lda, int64_t ldb, int64_t ldc, int64_t blk) {
  GemmGeometry g;
  const bool colMajor = (ord == 'C');
  const int64_t S[3] = {m, k, n};
  int64_t ld[3] = {lda, ldb, ldc};
  // logical (row-dim, col-dim) of A, B, C over the dim order {m,k,n}; a matrix
  // whose stored orientation is the transpose of its logical one swaps them
  const int logical_row[3] = {0, 1, 0}, logical_col[3] = {1, 2, 2};
  const bool swapped[3] = {(ta == 'T') != colMajor, (tb == 'T') != colMajor, colMajor};
  for (int x = 0; x < 3; x++) {
    g.size[x] = S[x];
    g.blk[x] = std::min(blk, S[x]);
    const int64_t q = g.blk[x] ? S[x] / g.blk[x] : 0;
    // a remainder shorter than one sector of floats is folded into the last block
    g.nblk[x] = (g.blk[x] && S[x] - q * g.blk[x] < kSectorFloats) ? q : q + 1;
    if (S[x] == 0) g.nblk[x] = 0;
  }
  for (int mat = 0; mat < 3; mat++) {
    g.rdim[mat] = swapped[mat] ? logical_col[mat] : logical_row[mat];
    g.cdim[mat] = swapped[mat] ? logical_row[mat] : logical_col[mat];
    g.ld[mat] = ld[mat] ? ld[mat] : S[g.cdim[mat]];
  }
  return g;
}

void gemm_task_at(const GemmGeometry &g, int64_t l, int64_t i, int64_t j, float beta,
                  bof_gemm_task *t) {
  const int64_t idx[3] = {i, l, j};
  int64_t ext[3];
  for (int x = 0; x < 3; x++)
    ext[x] = (idx[x] == g.nblk[x] - 1) ? g.size[x] - idx[x] * g.blk[x] : g.blk[x];
  for (int mat = 0; mat < 3; mat++) {
    const int r = g.rdim[mat], c = g.cdim[mat];
    t->nrows[mat] = ext[r];
    t->ncols[mat] = ext[c];
    t->ld_file[mat] = g.ld[mat];
    t->off[mat] = idx[r] * g.blk[r] * g.ld[mat] + idx[c] * g.blk[c];
  }
  t->l = l; t->i = i; t->j = j;
  t->M = ext[0]; t->K = ext[1]; t->N = ext[2];
  t->beta = l > 0 ? 1.0f : beta;
  t->parent = l > 0 ? ((l - 1) * g.nblk[0] + i) * g.nblk[2] + j : -1;
}

// Row-panel layout (flash_gemm_panels.cpp).  D = the dimension C is paneled along (m for
// row-major C).  The operand without D ("Y") is needed whole by every group of C panels and is
// resident; the other one ("X") streams through a ring of 2*group slots when it is paneled along
// D too, else it is resident as well; C gets a ring of 2*group+1 slots, deepened with spare
// budget.  Slots are 2 MiB-aligned.
// with_acc: the call also needs one raw accumulator panel (a C slot) per C panel of the first group (beta != 0 with the
// default arithmetic, bof_options.gemm_chain): they are part of what must fit, and are set aside BEFORE spare budget
// deepens the C ring.  need_bytes stays the panel slots alone; acc_bytes says what comes on top.
bof_panel_plan plan_panels(const GemmGeometry &g, uint64_t budget, int64_t group, int64_t full_dC, bool with_acc) {
  bof_panel_plan P{};
  P.streamed = -1;
  auto up = [](uint64_t v) { return (v + (2u << 20) - 1) / (2u << 20) * (2u << 20); };
  if (g.nblk[0] * g.nblk[1] * g.nblk[2] == 0) { P.why = 1; return P; }
  const int dC = g.rdim[2];
  for (int x = 0; x < 3; x++) {
    const int64_t rows = g.size[g.rdim[x]], ld = g.ld[x];
    const int64_t cols = (full_dC > 0 && x < 2 && g.cdim[x] == dC) ? full_dC : g.size[g.cdim[x]];
    if (ld < cols) { P.why = 2; return P; }
    if (x < 2 && cols * 2 < ld) { P.why = 3; return P; }
    P.n_panels[x] = g.nblk[g.rdim[x]];
    const int64_t last_rows = rows - (P.n_panels[x] - 1) * g.blk[g.rdim[x]];
    const int64_t max_rows = std::max(last_rows, std::min(rows, g.blk[g.rdim[x]]));
    P.slot_bytes[x] = up(((uint64_t) (max_rows - 1) * (uint64_t) ld + (uint64_t) cols) * 4);
  }
  if (g.ld[2] != g.size[g.cdim[2]]) { P.why = 4; return P; }
  const int xm = dC == 0 ? 0 : 1, ym = 1 - xm;
  const bool x_streams = g.rdim[xm] == dC;
  const int64_t NpC = g.nblk[dC];
  group = std::max<int64_t>(1, std::min(group, NpC));
  P.first_group = group;
  P.groups = 1 + (NpC - group);   // the ramp group, then one C panel at a time
  P.resident[ym] = 1;
  P.n_slots[ym] = P.n_panels[ym];
  P.resident[xm] = x_streams ? 0 : 1;
  P.n_slots[xm] = x_streams ? std::min<int64_t>(P.n_panels[xm], 2 * group) : P.n_panels[xm];
  P.streamed = x_streams && P.n_slots[xm] < P.n_panels[xm] ? xm : -1;
  P.n_slots[2] = std::min<int64_t>(NpC, 2 * group + 1);
  // every panel slot is an HBM allocation of its own (flash_gemm_panels.cpp allocates them in
  // first-use order while the first panels are read), resident matrices hold one slot per panel
  auto need_of = [&](int x) { return (uint64_t) P.n_slots[x] * P.slot_bytes[x]; };
  P.acc_bytes = g.nblk[1] > 1 ? (uint64_t) group * P.slot_bytes[2] : 0;
  uint64_t need = need_of(0) + need_of(1) + need_of(2);
  P.need_bytes = need;
  if (with_acc) need += P.acc_bytes;
  if (need > budget) { P.why = 5; return P; }
  // spare budget: a deeper C ring lets compute run ahead of a slow write-back
  while (P.n_slots[2] < NpC && need + P.slot_bytes[2] <= budget && P.n_slots[2] < 2 * group + 4) {
    P.n_slots[2]++;
    need += P.slot_bytes[2];
  }
  if (P.n_slots[2] == NpC) P.resident[2] = 1;
  P.need_bytes = need_of(0) + need_of(1) + need_of(2);
  P.eligible = 1;
  return P;
}

}  // namespace bof

extern "C" int bof_flash_gemm_panel_plan(char ord, char ta, char tb, uint64_t m, uint64_t n, uint64_t k,
                                         uint64_t lda, uint64_t ldb, uint64_t ldc, int64_t blk,
                                         uint64_t hbm_budget, int64_t group, bof_panel_plan *out) {
  if (!out || blk <= 0 || !(ord == 'R' || ord == 'C') || !(ta == 'N' || ta == 'T') || !(tb == 'N' || tb == 'T'))
    return BOF_EINVAL;
  const bof::GemmGeometry g = bof::gemm_geometry(ord, ta, tb, (int64_t) m, (int64_t) n, (int64_t) k, (int64_t) lda,
                                                 (int64_t) ldb, (int64_t) ldc, blk);
  *out = bof::plan_panels(g, hbm_budget, group);
  return BOF_OK;
}

extern "C" int64_t bof_gemm_plan(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
                                 float beta, int64_t lda, int64_t ldb, int64_t ldc, int64_t blk,
                                 bof_gemm_task *out, int64_t cap, int64_t nblk[3]) {
  if (blk <= 0 || m < 0 || n < 0 || k < 0) return BOF_EINVAL;
  const bof::GemmGeometry g = bof::gemm_geometry(ord, ta, tb, m, n, k, lda, ldb, ldc, blk);
  if (nblk) { nblk[0] = g.nblk[0]; nblk[1] = g.nblk[1]; nblk[2] = g.nblk[2]; }
  const int64_t total = g.nblk[0] * g.nblk[1] * g.nblk[2];
  if (!out) return total;
  int64_t t = 0;
  for (int64_t l = 0; l < g.nblk[1]; l++)      // reference injects l-major
    for (int64_t i = 0; i < g.nblk[0]; i++)
      for (int64_t j = 0; j < g.nblk[2]; j++, t++) {
        if (t >= cap) return total;
        bof::gemm_task_at(g, l, i, j, beta, &out[t]);
      }
  return total;
}

extern "C" int64_t bof_csr_blocks(const int64_t *ia, int64_t m, int64_t min_rows,
                                  int64_t max_rows, int64_t max_nnz, int64_t *starts,
                                  int64_t *sizes, int64_t cap) {
  if (!ia || m < 0 || min_rows <= 0 || max_rows <= 0) return BOF_EINVAL;
  int64_t cur = 0, nb = 0;
  while (cur < m) {
    const int64_t left = m - cur;
    // The reference grows the block one row at a time from min_rows while its nnz stays
    // within the budget (so it ends one row PAST the budget).  `ia` is non-decreasing, so
    // the same end is the first row offset b >= min_rows with ia[cur+b] > ia[cur]+max_nnz,
    // found by binary search (the linear scan costs 0.1 s of host time at 50M rows).
    int64_t b = min_rows;
    if (b < left) {
      const int64_t *lo = ia + cur + min_rows, *hi = ia + cur + left;
      b = std::upper_bound(lo, hi, ia[cur] + max_nnz) - (ia + cur);
    }
    b = std::min(std::min(b, max_rows), left);  // clamp: reference over-runs ia here
    if (nb < cap) {
      if (starts) starts[nb] = cur;
      if (sizes) sizes[nb] = b;
    }
    nb++;
    cur += b;
  }
  return nb;
}
