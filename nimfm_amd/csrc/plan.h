// nimfm_amd/csrc/plan.h -- the per-epoch "batch plan": batch boundaries plus, for every batch, the
// transposed (feature-major) view of its rows that the column phase consumes.
#pragma once
#include "common.h"

namespace nfm {

// For batch b (samples bat_pos[b]..bat_pos[b+1]) relative to `begin`):
//   unique features  ucol[u], u in [bat_uoff[b], bat_uoff[b+1])
//   touches of ucol[u]: t in [uptr[u], uptr[u+1]) -> (tpos[t] = sample position inside the batch,
//   tx[t] = value, tq[t] = the touch's index in SAMPLE order, toff[pos] + q, where the row phase left its data),
//   sorted by sample position (stable sort) so that every per-feature sum has a fixed order.
struct Plan {
  // identity (cache key)
  uint64_t serial = 0;  // unique per build
  uint64_t ds_uid = 0;  // nfm_dataset::uid of the dataset the plan was built from (0 = none), and its nnz
  int64_t ds_nnz = -1;
  int64_t begin = 0, end = 0, batch = 0;
  int n_aug = 0;
  bool first_singleton = false, has_perm = false, use_singles = false;
  // geometry
  int64_t n_batches = 0, U = 0, T = 0, TM = 0, max_batch = 0, max_unique = 0;  // T touches, TM of them in the column phase
  std::vector<int64_t> bat_pos;   // host, n_batches + 1, relative to begin
  std::vector<int64_t> bat_uoff;  // host, n_batches + 1
  DevBuf perm;                    // int64[end-begin] (absolute sample ids) or empty
  DevBuf bat_pos_dev;             // int64[n_batches + 1]
  DevBuf ucol;                    // int32[U]
  DevBuf uptr;                    // int64[U + 1]
  // the same unique features, inside every batch ordered by DESCENDING touch count: the lane groups of a
  // wavefront walk their features' touch lists in lock step, so a wavefront lasts as long as its longest list --
  // neighbours of equal length waste nothing (cfg2: the longest of 8 Poisson(10.5) lists is 15)
  DevBuf ucol_s;                  // int32[U] feature
  DevBuf ubeg_s;                  // int64[U] first touch
  DevBuf ucnt_s;                  // int32[U] touches
  DevBuf tpos;                    // int32[TM]
  DevBuf tx;                      // double[TM]
  DevBuf tq;                      // int64[TM] (only when want_tq)
  DevBuf toff;                    // int64[end-begin+1] touch offset of every sample (with use_singles / want_tq)
  DevBuf single;                  // uint8[T] in sample order: 1 = the feature is touched once in its batch
  // "heavy" features (more than kHeavyTouches touches in one batch, e.g. Zipf heads or dummy features):
  // their touch lists are cut into segments of kHeavySegment touches that separate lane groups sum,
  // then one group per feature adds the segments' partial sums in segment order (deterministic).
  int64_t H = 0, HS = 0;          // heavy features / their segments, all batches
  std::vector<int64_t> bat_hoff;  // host, n_batches + 1: first heavy feature of every batch
  std::vector<int64_t> bat_soff;  // host, n_batches + 1: first heavy segment of every batch
  int64_t max_heavy = 0, max_segs = 0;
  DevBuf hv_u;                    // int64[H]   index into ucol/uptr
  DevBuf hv_seg0;                 // int64[H+1] first segment of every heavy feature
  void release();
};

// The column-major twin of a dataset (plan.hip: plans for dense batches come from it), built once per dataset on the
// first epoch that can use it: column j holds its entries [cptr[j], cptr[j+1]) in sample order.
struct CscIndex {
  bool built = false, usable = false;
  bool seg_unfit = false;  // a (batch, feature range) cell of this dataset overflowed once: plans by bucketing are not tried again
  int64_t max_col = 0;
  DevBuf cptr;  // int64[d + 1]
  DevBuf crow;  // int32[nnz] sample
  DevBuf cval;  // double[nnz]
  DevBuf cnz;   // uint32[nnz] index of the entry in the row-major arrays
};

// A lane group walks a feature's touches serially (~0.25 us per pair of touches, latency-bound): beyond
// kHeavyTouches the list is cut into segments of kHeavySegment touches summed by separate lane groups.
// (Measured on cfg2 with Zipf(1.1) popularity, B = 32768: 256/128 -> 245 us per batch, 48/32 -> 402 us: too
// many small segments make the per-feature sum of partials the new tail.)
constexpr int kHeavyTouches = 128;
constexpr int kHeavySegment = 64;

// how many entries repeat a column id of their row, and the first such row (-1: none).  decisionFunction takes repeats as
// the reference does -- every entry is one more term of the ANOVA recursion (kernels.nim:46-64) -- but every TRAINING kernel
// assumes distinct ids per row (the reference's behaviour for repeats is an accident of its lazy scaling and scratch layout,
// optimizer/sgd.nim:134-143, 176-188: the row is rescaled once per entry, the second entry overwrites the first one's
// derivative and the row's parameters are stepped twice with it): nfm_opt_epoch refuses such a dataset.
int check_rows_distinct(nfm_ctx* ctx, const CsrView& X, int64_t* n_repeats, int64_t* first_row);

// The device-drawn order (gen_permutation) is a keyed bijection evaluated per position -- and so is its INVERSE: the column
// path of plan_build asks "at which position does sample i stand" once per touch, and computes the answer from the key
// instead of gathering it from a table of the inverse permutation (one 64-byte sector per touch: 640 M of them per epoch of
// the headline).
constexpr int kFeistelRounds = 4;  // (a round is a full-avalanche 32-bit mix; eight rounds cost the column path 2.6 ms per headline epoch more)
struct FeistelKey {
  uint32_t k[kFeistelRounds];
  uint32_t mask;
  int h;
  int64_t ns, begin;  // the order: positions 0 .. ns-1 -> samples begin .. begin+ns-1
};
FeistelKey feistel_key(int64_t seed, uint64_t epoch, int64_t begin, int64_t ns);

// stream: where the build is enqueued and synchronised (default: the context's); perm_dev: the end - begin sample ids of
// the epoch already on the device (instead of perm_host); perm_key: perm_dev is gen_permutation's order of this key
int plan_build(nfm_ctx* ctx, const CsrView& X, int n_aug, const int64_t* perm_host, int64_t begin, int64_t end,
               int64_t batch, bool first_singleton, bool want_tq, bool use_singles, bool sort_by_count, Plan* out,
               hipStream_t stream = nullptr, const int64_t* perm_dev = nullptr, CscIndex* csc = nullptr,
               const FeistelKey* perm_key = nullptr);

// a random order of the samples begin .. begin+ns-1 drawn on the device from (seed, epoch): int64[ns] in *out
int gen_permutation(nfm_ctx* ctx, hipStream_t st, int64_t seed, uint64_t epoch, int64_t begin, int64_t ns, DevBuf* out);

}  // namespace nfm
