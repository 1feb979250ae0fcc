// nimfm_amd/csrc/plan.hip -- builds the batch plan on the device.
//
// The reference has no counterpart: its per-sample loop (optimizer/sgd.nim:298-308) needs no
// transposed view.  The deterministic mini-batch rule does: the column
// phase sums, per feature, the contributions of the batch's samples in sample order.  This is
// preprocessing (one stable radix sort of (batch, feature) keys per plan, rocPRIM via hipCUB);
// with shuffle off the plan is reused by every epoch.
#include <hipcub/hipcub.hpp>
#include <type_traits>

#include <stdlib.h>

#include <atomic>

#include "plan.h"

namespace nfm {

void Plan::release() {
  perm.release();
  bat_pos_dev.release();
  ucol.release();
  uptr.release();
  ucol_s.release();
  ubeg_s.release();
  ucnt_s.release();
  tpos.release();
  tx.release();
  tq.release();
  toff.release();
  single.release();
  hv_u.release();
  hv_seg0.release();
  bat_hoff.clear();
  bat_soff.clear();
  H = HS = max_heavy = max_segs = 0;
  bat_pos.clear();
  bat_uoff.clear();
  n_batches = U = T = TM = 0;
}

__device__ __forceinline__ int64_t batch_of(int64_t rel, int64_t batch, int first_singleton) {
  if (first_singleton) return rel == 0 ? 0 : 1 + (rel - 1) / batch;
  return rel / batch;
}
__device__ __forceinline__ int64_t batch_start(int64_t b, int64_t batch, int first_singleton) {
  if (first_singleton) return b == 0 ? 0 : 1 + (b - 1) * batch;
  return b * batch;
}

// sample ids of an order handed over by the host: all inside [0, n)?
__global__ void k_perm_range(int64_t ns, const int64_t* __restrict__ perm, int64_t n, unsigned long long* __restrict__ bad) {
  unsigned long long c = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < ns; r += (int64_t)gridDim.x * blockDim.x)
    c += (perm[r] < 0 || perm[r] >= n) ? 1 : 0;
  if (c) atomicAdd(bad, c);
}

__global__ void k_row_len(CsrView X, const int64_t* __restrict__ perm, int64_t begin, int64_t ns, int n_aug,
                          int64_t* __restrict__ len) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= ns; r += (int64_t)gridDim.x * blockDim.x) {
    if (r == ns) {
      len[r] = 0;
    } else {
      const int64_t i = perm ? perm[r] : begin + r;
      len[r] = X.indptr[i + 1] - X.indptr[i] + n_aug;
    }
  }
}

struct alignas(16) TouchPay {  // what a touch carries to its sorted place
  double x;
  int32_t pib, pad_;
};
// one wavefront per sample: writes the (batch, feature) key and the touch payload of every nnz
template <class KeyT>
__global__ void k_expand(CsrView X, const int64_t* __restrict__ perm, int64_t begin, int64_t ns, int n_aug,
                         int64_t batch, int first_singleton, int fbits, const int64_t* __restrict__ toff,
                         KeyT* __restrict__ keys, uint32_t* __restrict__ vals, TouchPay* __restrict__ pay_un) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t r = wave0; r < ns; r += nwaves) {
    const int64_t i = perm ? perm[r] : begin + r;
    const int64_t q0 = X.indptr[i];
    const int m = (int)(X.indptr[i + 1] - q0);
    const int m_tot = m + n_aug;
    const int64_t b = batch_of(r, batch, first_singleton);
    const int32_t pib = (int32_t)(r - batch_start(b, batch, first_singleton));
    const int64_t t0 = toff[r];
    for (int q = lane; q < m_tot; q += kWave) {
      const int64_t j = q < m ? (int64_t)X.indices[q0 + q] : X.d + (q - m);
      const int64_t t = t0 + q;
      keys[t] = (KeyT)(((uint64_t)b << fbits) | (uint64_t)j);
      vals[t] = (uint32_t)t;
      pay_un[t] = TouchPay{q < m ? X.data[q0 + q] : 1.0, pib, 0};
    }
  }
}

// per sorted touch r: is its feature touched exactly once in the batch (single), and is r the
// first touch of a feature with >= 2 touches (head)?
template <class KeyT>
__global__ void k_classify(int64_t T, const KeyT* __restrict__ keys, const uint32_t* __restrict__ vals, int use_singles,
                           uint8_t* __restrict__ single_un, int32_t* __restrict__ multi, int32_t* __restrict__ head) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= T; r += (int64_t)gridDim.x * blockDim.x) {
    if (r == T) {
      multi[r] = 0;
      head[r] = 0;
      continue;
    }
    const KeyT key = keys[r];
    const bool first = r == 0 || keys[r - 1] != key;
    const bool last = r + 1 == T || keys[r + 1] != key;
    const bool is_single = use_singles && first && last;
    if (single_un && is_single) single_un[vals[r]] = 1;  // the table is zeroed beforehand: a scattered byte per SINGLE touch only
                                                         // (this pass is bound by the number of scattered writes)
    multi[r] = is_single ? 0 : 1;
    head[r] = (!is_single && first) ? 1 : 0;
  }
}

template <class KeyT>
__global__ void k_compact(int64_t T, const KeyT* __restrict__ keys, const uint32_t* __restrict__ vals,
                          const int32_t* __restrict__ mpos, const int32_t* __restrict__ uidx, int fbits,
                          const TouchPay* __restrict__ pay_un, int32_t* __restrict__ tpos, double* __restrict__ tx,
                          int64_t* __restrict__ tq, int32_t* __restrict__ ucol, int64_t* __restrict__ uptr,
                          int64_t* __restrict__ ubatch) {
  const uint64_t fmask = (fbits >= 64) ? ~0ull : ((1ull << fbits) - 1);
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < T; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t mp = mpos[r];
    if (mpos[r + 1] == mp) continue;  // single
    const uint32_t t = vals[r];
    const TouchPay pay = pay_un[t];  // one 16-byte gather per touch (value and position used to be two arrays: two lines)
    tpos[mp] = pay.pib;
    tx[mp] = pay.x;
    if (tq) tq[mp] = (int64_t)t;  // the touch's index in SAMPLE order (row-phase contribution slot)
    const int64_t u = uidx[r];
    if (uidx[r + 1] != u) {  // head of a multi-touch feature
      const uint64_t key = (uint64_t)keys[r];
      ucol[u] = (int32_t)(key & fmask);
      uptr[u] = mp;
      ubatch[u] = (int64_t)(key >> fbits);
    }
  }
}

// The column phase walks a batch's unique features in the order of DESCENDING touch count (plan.h); counts of kCntMax and
// more share a class and keep their feature order among themselves.  16 classes: counts differ by a few around their
// mean (Poisson), and a 4-bit class keeps the (batch, class) key of the sort path at two radix passes.
constexpr int kCntClassBits = 4, kCntClasses = 1 << kCntClassBits, kCntMax = kCntClasses - 1;
__host__ __device__ __forceinline__ uint32_t count_class(int64_t c, uint32_t cmax) { return cmax - (c < (int64_t)cmax ? (uint32_t)c : cmax); }

// (batch, descending touch count) keys of the unique features, and the gather into the sorted tables
__global__ void k_ukeys(int64_t U, const int64_t* __restrict__ ubatch, const int64_t* __restrict__ uptr, int cbits,
                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t cmax = (1u << cbits) - 1;  // counts beyond it share a key: they stay in feature order among themselves
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < U; u += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = uptr[u + 1] - uptr[u];
    keys[u] = ((uint32_t)ubatch[u] << cbits) | count_class(c, cmax);
    vals[u] = (uint32_t)u;
  }
}
__global__ void k_usorted(int64_t U, const uint32_t* __restrict__ vals, const int32_t* __restrict__ ucol,
                          const int64_t* __restrict__ uptr, int32_t* __restrict__ ucol_s, int64_t* __restrict__ ubeg_s,
                          int32_t* __restrict__ ucnt_s, unsigned long long* __restrict__ n_heavy) {
  unsigned long long heavy = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t u = vals ? vals[i] : (uint32_t)i;  // no order given: the features' own order
    const int64_t c = uptr[u + 1] - uptr[u];
    ucol_s[i] = ucol[u];
    ubeg_s[i] = uptr[u];
    ucnt_s[i] = (int32_t)c;
    heavy += c > kHeavyTouches ? 1 : 0;
  }
  if (heavy) atomicAdd(n_heavy, heavy);  // most plans have none: the scans behind the heavy lists are skipped then
}

// first unique feature of every batch that has one
__global__ void k_batch_first(int64_t U, const int64_t* __restrict__ ubatch, int64_t* __restrict__ bat_first_u) {
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < U; u += (int64_t)gridDim.x * blockDim.x)
    if (u == 0 || ubatch[u - 1] != ubatch[u]) bat_first_u[ubatch[u]] = u;
}

// per unique feature: is it heavy, and how many segments does it need
__global__ void k_heavy_flags(int64_t U, const int64_t* __restrict__ uptr, int64_t* __restrict__ hflag, int64_t* __restrict__ nseg) {
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u <= U; u += (int64_t)gridDim.x * blockDim.x) {
    if (u == U) {
      hflag[u] = 0;
      nseg[u] = 0;
      continue;
    }
    const int64_t c = uptr[u + 1] - uptr[u];
    const bool heavy = c > kHeavyTouches;
    hflag[u] = heavy ? 1 : 0;
    nseg[u] = heavy ? (c + kHeavySegment - 1) / kHeavySegment : 0;
  }
}
__global__ void k_heavy_compact(int64_t U, const int64_t* __restrict__ hidx, const int64_t* __restrict__ segoff,
                                int64_t* __restrict__ hv_u, int64_t* __restrict__ hv_seg0) {
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < U; u += (int64_t)gridDim.x * blockDim.x)
    if (hidx[u + 1] != hidx[u]) {
      hv_u[hidx[u]] = u;
      hv_seg0[hidx[u]] = segoff[u];
    }
}
__global__ void k_gather_i64(int64_t n, const int64_t* __restrict__ src, const int64_t* __restrict__ at, int64_t* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[at[i]];
}

__global__ void k_set_i64(int64_t* p, int64_t n, int64_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---- distinct column ids inside every row (include/nimfm_hip.h, nfm_dataset_create_csr) ----
// one wavefront per row.  SORTED = false: counts the positions whose id is not larger than its predecessor's (0 = every
// row is strictly increasing, hence distinct -- the common case, one pass over the ids).  SORTED = true (ids is a copy
// sorted inside every row): counts equal neighbours = duplicates.
template <bool SORTED>
__global__ void k_row_order(int64_t n, const int64_t* __restrict__ indptr, const int32_t* __restrict__ ids,
                            unsigned long long* __restrict__ count, long long* __restrict__ first_row) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  unsigned long long bad = 0;
  for (int64_t r = wave0; r < n; r += nwaves) {
    const int64_t q0 = indptr[r], q1 = indptr[r + 1];
    bool row_bad = false;
    for (int64_t q = q0 + 1 + lane; q < q1; q += kWave) {
      const int32_t a = ids[q - 1], b = ids[q];
      if (SORTED ? a == b : b <= a) {
        ++bad;
        row_bad = true;
      }
    }
    if (row_bad) atomicMin(first_row, (long long)r);
  }
  if (bad) atomicAdd(count, bad);
}

int check_rows_distinct(nfm_ctx* ctx, const CsrView& X, int64_t* n_repeats, int64_t* first_row) {
  *n_repeats = 0;
  *first_row = -1;
  if (X.n == 0 || X.nnz < 2 || X.max_row < 2) return NFM_OK;
  hipStream_t st = ctx->stream;
  DevBuf stat;
  NFM_TRY(stat.alloc(16));
  long long h[2] = {0, INT64_MAX};
  auto run = [&](bool sorted, const int32_t* ids) -> int {
    h[0] = 0;
    h[1] = INT64_MAX;
    NFM_HIP_CHECK(hipMemcpyAsync(stat.p, h, sizeof(h), hipMemcpyHostToDevice, st));
    int64_t blocks = (X.n + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (sorted)
      hipLaunchKernelGGL(k_row_order<true>, dim3((unsigned)blocks), dim3(kBlock), 0, st, X.n, X.indptr, ids,
                         stat.as<unsigned long long>(), stat.as<long long>() + 1);
    else
      hipLaunchKernelGGL(k_row_order<false>, dim3((unsigned)blocks), dim3(kBlock), 0, st, X.n, X.indptr, ids,
                         stat.as<unsigned long long>(), stat.as<long long>() + 1);
    NFM_HIP_CHECK(hipGetLastError());
    NFM_HIP_CHECK(hipMemcpyAsync(h, stat.p, sizeof(h), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
    return NFM_OK;
  };
  NFM_TRY(run(false, X.indices));
  if (h[0] == 0) return NFM_OK;  // every row strictly increasing
  // some row is stored out of order (allowed, dataset.nim:597-612): sort a copy of the ids inside every row
  NFM_CHECK(X.nnz < (int64_t)2147483647, NFM_ERR_UNSUPPORTED, "more than 2^31-1 nnz with unsorted rows");
  DevBuf k0, k1, tmp;
  NFM_TRY(k0.alloc(sizeof(int32_t) * X.nnz));
  NFM_TRY(k1.alloc(sizeof(int32_t) * X.nnz));
  NFM_HIP_CHECK(hipMemcpyAsync(k0.p, X.indices, sizeof(int32_t) * X.nnz, hipMemcpyDeviceToDevice, st));
  hipcub::DoubleBuffer<int32_t> dk(k0.as<int32_t>(), k1.as<int32_t>());
  size_t bytes = 0;
  NFM_HIP_CHECK(hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, bytes, dk, (int)X.nnz, (int)X.n, X.indptr, X.indptr + 1, 0, 32, st));
  NFM_TRY(tmp.alloc(bytes));
  NFM_HIP_CHECK(hipcub::DeviceSegmentedRadixSort::SortKeys(tmp.p, bytes, dk, (int)X.nnz, (int)X.n, X.indptr, X.indptr + 1, 0, 32, st));
  NFM_TRY(run(true, dk.Current()));
  *n_repeats = h[0];
  *first_row = h[0] ? h[1] : -1;
  return NFM_OK;
}

static inline unsigned grid1d(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 256 * 32) b = 256 * 32;
  return (unsigned)b;
}

// ------------------------------------------------------------------------------------------------
// Plans from the column-major twin of the data (dense batches: every feature is touched in most batches).
// The (batch, feature) sort above moves every touch through several radix passes and a gather.  The COLUMN-major
// copy of the matrix (CscIndex, built once per dataset) already groups the touches by feature, in sample order; under
// a permutation only the order INSIDE a column changes.  Per epoch: ipos = inverse permutation; one wavefront per
// feature counts its touches per batch (k_csc_count), a scan over the (batch, feature) table gives every group its
// place, and the same wavefront ranks each touch inside its (batch, feature) group by position and writes it there
// (k_csc_fill).  No pass over all touches but these two; cfg2 (32 M touches): ~0.8 ms instead of ~2.9 ms.
// ------------------------------------------------------------------------------------------------
constexpr int kCscMaxCol = 1024;      // longest column the ranking kernel takes (16 entries per lane)
constexpr int kCscMaxBatches = 512;   // per-wavefront LDS: k_csc_fill 256 NE + 2 x 512 + 2 ints (NE = 16: 20.5 KB, two wavefronts per workgroup), k_csc_count 4 x 512 ints

__global__ void k_iota_u32(int64_t n, uint32_t* __restrict__ v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}
// column starts by binary search in the sorted column ids
__global__ void k_csc_ptr(int64_t d, int64_t nnz, const uint32_t* __restrict__ cols, int64_t* __restrict__ cptr) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= d; j += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)cols[mid] < j) lo = mid + 1; else hi = mid;
    }
    cptr[j] = lo;
  }
}
__global__ void k_csc_rows(CsrView X, int64_t nnz, const uint32_t* __restrict__ nz, int32_t* __restrict__ crow, double* __restrict__ cval) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = nz[e];
    int64_t lo = 0, hi = X.n;  // row of entry v: last i with indptr[i] <= v
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (X.indptr[mid] <= v) lo = mid; else hi = mid - 1;
    }
    crow[e] = (int32_t)lo;
    cval[e] = X.data[v];
  }
}
__global__ void k_col_maxlen(int64_t d, const int64_t* __restrict__ cptr, unsigned long long* __restrict__ out) {
  unsigned long long m = 0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < d; j += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long c = (unsigned long long)(cptr[j + 1] - cptr[j]);
    m = c > m ? c : m;
  }
  if (m) atomicMax(out, m);
}

int csc_build(nfm_ctx* ctx, hipStream_t st, const CsrView& X, CscIndex* C) {
  C->built = true;  // one attempt per dataset
  C->usable = false;
  if (X.nnz == 0 || X.nnz >= (int64_t)2147483647 || X.n >= (int64_t)2147483647) return NFM_OK;
  DevBuf k1, v0, tmp, mx;
  NFM_TRY(k1.alloc(sizeof(uint32_t) * X.nnz));
  NFM_TRY(v0.alloc(sizeof(uint32_t) * X.nnz));
  NFM_TRY(C->cnz.alloc(sizeof(uint32_t) * X.nnz));
  hipLaunchKernelGGL(k_iota_u32, dim3(grid1d(X.nnz)), dim3(kBlock), 0, st, X.nnz, v0.as<uint32_t>());
  int fbits = 1;
  while (((int64_t)1 << fbits) < X.d) ++fbits;
  size_t bytes = 0;
  const uint32_t* cols_in = reinterpret_cast<const uint32_t*>(X.indices);  // ids are non-negative int32
  NFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, cols_in, k1.as<uint32_t>(), v0.as<uint32_t>(), C->cnz.as<uint32_t>(),
                                                    (int)X.nnz, 0, fbits, st));
  NFM_TRY(tmp.alloc(bytes));
  NFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, cols_in, k1.as<uint32_t>(), v0.as<uint32_t>(), C->cnz.as<uint32_t>(),
                                                    (int)X.nnz, 0, fbits, st));  // stable: sample order inside a column
  NFM_TRY(C->cptr.alloc(sizeof(int64_t) * (X.d + 1)));
  NFM_TRY(C->crow.alloc(sizeof(int32_t) * X.nnz));
  NFM_TRY(C->cval.alloc(sizeof(double) * X.nnz));
  NFM_TRY(mx.alloc(sizeof(unsigned long long)));
  NFM_HIP_CHECK(hipMemsetAsync(mx.p, 0, sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_csc_ptr, dim3(grid1d(X.d + 1)), dim3(kBlock), 0, st, X.d, X.nnz, k1.as<uint32_t>(), C->cptr.as<int64_t>());
  hipLaunchKernelGGL(k_csc_rows, dim3(grid1d(X.nnz)), dim3(kBlock), 0, st, X, X.nnz, C->cnz.as<uint32_t>(), C->crow.as<int32_t>(),
                     C->cval.as<double>());
  hipLaunchKernelGGL(k_col_maxlen, dim3(grid1d(X.d)), dim3(kBlock), 0, st, X.d, C->cptr.as<int64_t>(), mx.as<unsigned long long>());
  NFM_HIP_CHECK(hipGetLastError());
  unsigned long long h = 0;
  NFM_HIP_CHECK(hipMemcpyAsync(&h, mx.p, sizeof(h), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  C->max_col = (int64_t)h;
  C->usable = C->max_col <= kCscMaxCol;
  if (!C->usable) { C->cptr.release(); C->crow.release(); C->cval.release(); C->cnz.release(); }
  return NFM_OK;
}

__global__ void k_ipos(int64_t ns, const int64_t* __restrict__ perm, int64_t begin, int32_t* __restrict__ ipos,
                       unsigned long long* __restrict__ clash) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < ns; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = perm ? perm[r] : begin + r;
    if (atomicExch(&ipos[i], (int32_t)r) != -1) atomicAdd(clash, 1ull);  // a sample twice in the order: not a permutation
  }
}

// ---- the device-drawn order as a function (plan.h: FeistelKey) ----
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  return x ^ (x >> 16);
}
__device__ __forceinline__ uint64_t feistel(uint64_t x, const FeistelKey& K) {
  uint32_t L = (uint32_t)(x >> K.h) & K.mask, R = (uint32_t)x & K.mask;
#pragma unroll
  for (int r = 0; r < kFeistelRounds; ++r) {
    const uint32_t t = L ^ (mix32(R ^ K.k[r]) & K.mask);
    L = R;
    R = t;
  }
  return ((uint64_t)L << K.h) | R;
}
// the inverse: the rounds run backwards (L' = R, R' = L ^ f(R): R = L', L = R' ^ f(L'))
__device__ __forceinline__ uint64_t feistel_inv(uint64_t x, const FeistelKey& K) {
  uint32_t L = (uint32_t)(x >> K.h) & K.mask, R = (uint32_t)x & K.mask;
#pragma unroll
  for (int r = kFeistelRounds - 1; r >= 0; --r) {
    const uint32_t t = R ^ (mix32(L ^ K.k[r]) & K.mask);
    R = L;
    L = t;
  }
  return ((uint64_t)L << K.h) | R;
}
// position of sample i in the order of key K (-1: the sample is not in the order's range): the cycle of the restricted
// bijection is walked backwards
__device__ __forceinline__ int32_t feistel_position(int64_t i, const FeistelKey& K) {
  if (i < K.begin || i >= K.begin + K.ns) return -1;
  uint64_t x = (uint64_t)(i - K.begin);
  do x = feistel_inv(x, K); while (x >= (uint64_t)K.ns);
  return (int32_t)x;
}

// 32-bit forms for the column path (positions and batch sizes below 2^31 there): a 64-bit divide is ~100 instructions
__device__ __forceinline__ int batch_of32(int32_t rel, uint32_t batch, int first_singleton) {
  if (first_singleton) return rel == 0 ? 0 : 1 + (int)((uint32_t)(rel - 1) / batch);
  return (int)((uint32_t)rel / batch);
}

// one wavefront per kCntGroup consecutive features: their touches per batch -> cnt[b * d + j].  A column's row ids are
// requested in one go (kCscMaxCol / 64 predicated loads per lane), then the positions of those rows; the counts of the
// group leave as runs of kCntGroup consecutive ints per batch.
constexpr int kCntGroup = 4;
template <bool FEISTEL>  // the positions computed from the order's key instead of gathered from the inverse permutation
__global__ __launch_bounds__(kBlock) void k_csc_count(int64_t d, const int64_t* __restrict__ cptr, const int32_t* __restrict__ crow,
                                                      const int32_t* __restrict__ ipos, int64_t batch, int first_singleton,
                                                      int n_batches, int32_t* __restrict__ cnt, int32_t* __restrict__ rpos, FeistelKey K) {
  extern __shared__ int s_hist[];  // [kWavesPerBlock][kCntGroup][n_batches]
  constexpr int NE = kCscMaxCol / kWave;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  int* hist = s_hist + wv * kCntGroup * n_batches;
  const uint32_t batch32 = (uint32_t)batch;
  const int64_t n_groups = (d + kCntGroup - 1) / kCntGroup;
  for (int i = lane; i < kCntGroup * n_batches; i += kWave) hist[i] = 0;
  __builtin_amdgcn_wave_barrier();
  for (int64_t g = (int64_t)blockIdx.x * kWavesPerBlock + wv; g < n_groups; g += (int64_t)gridDim.x * kWavesPerBlock) {
    const int64_t j0 = g * kCntGroup;
    for (int f = 0; f < kCntGroup && j0 + f < d; ++f) {
      const int64_t e0 = cptr[j0 + f];
      const int n_c = (int)(cptr[j0 + f + 1] - e0);
      int32_t row[NE];
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const int q = u * kWave + lane;
        row[u] = q < n_c ? crow[e0 + q] : -1;
      }
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const int q = u * kWave + lane;
        if (q < n_c) {
          // (table: a random 4-byte gather per touch); kept in column order for k_csc_fill
          const int32_t r = FEISTEL ? feistel_position(row[u], K) : ipos[row[u]];
          rpos[e0 + q] = r;
          if (r >= 0) atomicAdd(&hist[f * n_batches + batch_of32(r, batch32, first_singleton)], 1);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < kCntGroup * n_batches; i += kWave) {
      const int b = i / kCntGroup, f = i % kCntGroup;
      if (j0 + f < d) cnt[(size_t)b * d + j0 + f] = hist[f * n_batches + b];
      hist[f * n_batches + b] = 0;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// One scan over the (batch, feature) cells gives both prefix sums the plan needs: the touches before a cell (low word)
// and the non-empty cells -- the batches' unique features -- before it (high word).  Both stay below 2^31.
struct CellPack {
  __host__ __device__ __forceinline__ uint64_t operator()(int32_t c) const { return (uint64_t)(uint32_t)c | ((uint64_t)(c > 0 ? 1 : 0) << 32); }
};
// the cells that hold touches, in (batch, feature) order: the batch's unique features
__global__ void k_csc_units(int64_t cells, int64_t d, const int32_t* __restrict__ cnt, const uint64_t* __restrict__ ps,
                            int32_t* __restrict__ ucol, int64_t* __restrict__ uptr, int64_t* __restrict__ ubatch) {
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (int64_t)gridDim.x * blockDim.x)
    if (cnt[c] > 0) {
      const uint64_t v = ps[c];
      const int32_t u = (int32_t)(v >> 32);
      ucol[u] = (int32_t)(c % d);
      uptr[u] = (int64_t)(uint32_t)v;
      ubatch[u] = c / d;
    }
}

// one wavefront per feature: the column's touches are bucketed by batch in LDS (histogram -> scan -> slots handed out
// by LDS atomics, in any order), then every touch finds its rank inside its (batch, feature) group -- the touches of
// the same bucket at smaller positions, a dozen comparisons in the dense regime instead of the whole column.  Bucket
// start + rank is the touch's place in the column sorted by (batch, position); the touches are put there in LDS and
// leave from there, lane after lane in sorted order: neighbouring lanes then write neighbouring slots of the same
// (batch, feature) group (and read the same entry of the offset table) instead of 64 scattered 4- and 8-byte pieces --
// the kernel was bound by the NUMBER of memory requests, not by bytes.  A column has at most 64 NE entries.
template <int NE>
__global__ __launch_bounds__(kBlock) void k_csc_fill(CsrView X, const int64_t* __restrict__ cptr, const int32_t* __restrict__ crow,
                                                     const double* __restrict__ cval, const uint32_t* __restrict__ cnz,
                                                     const int32_t* __restrict__ rpos, int64_t batch, int first_singleton,
                                                     int n_batches, const uint64_t* __restrict__ ps,
                                                     const int64_t* __restrict__ toff, int32_t* __restrict__ tpos,
                                                     double* __restrict__ tx, int64_t* __restrict__ tq) {
  // per wavefront: sx[64 NE] doubles | hist[n_batches] | bstart[n_batches + 1] | bpos[64 NE] | sq[64 NE]
  extern __shared__ __align__(16) int s_dyn[];
  constexpr int NC = NE * kWave;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int per_wave = (2 * NC + 2 * n_batches + 1 + 2 * NC + 1) & ~1;
  int* base = s_dyn + wv * per_wave;
  double* sx = reinterpret_cast<double*>(base);
  int* hist = base + 2 * NC;
  int* bstart = hist + n_batches;
  int* bpos = bstart + n_batches + 1;  // the bucket's positions while ranking, then the sorted column's positions
  int* sq = bpos + NC;                 // sorted place -> entry of the column
  const int64_t d = X.d;
  const uint32_t batch32 = (uint32_t)batch;
  const int wpb = (int)(blockDim.x >> 6);  // wavefronts per workgroup (fewer for long columns: LDS)
  for (int64_t j = (int64_t)blockIdx.x * wpb + wv; j < d; j += (int64_t)gridDim.x * wpb) {
    const int64_t e0 = cptr[j];
    const int n_c = (int)(cptr[j + 1] - e0);
    const int nu = (n_c + kWave - 1) / kWave;  // live entries per lane (wave-uniform)
    for (int b = lane; b < n_batches; b += kWave) hist[b] = 0;
    __builtin_amdgcn_wave_barrier();
    int32_t r[NE];
    int bb[NE];
    double xv[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int q = u * kWave + lane;
      r[u] = q < n_c ? rpos[e0 + q] : -1;
      xv[u] = q < n_c ? cval[e0 + q] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      bb[u] = r[u] >= 0 ? batch_of32(r[u], batch32, first_singleton) : -1;
      if (bb[u] >= 0) atomicAdd(&hist[bb[u]], 1);
    }
    __builtin_amdgcn_wave_barrier();
    // exclusive scan of the histogram -> first slot of every bucket; the histogram is reused as the fill counter
    int carry = 0;
    for (int b0 = 0; b0 < n_batches; b0 += kWave) {
      const int v = b0 + lane < n_batches ? hist[b0 + lane] : 0;
      int incl = v;
#pragma unroll
      for (int sh = 1; sh < kWave; sh <<= 1) {
        const int o = __shfl_up(incl, sh, kWave);
        if (lane >= sh) incl += o;
      }
      if (b0 + lane < n_batches) {
        bstart[b0 + lane] = carry + incl - v;
        hist[b0 + lane] = 0;
      }
      carry += __shfl(incl, kWave - 1, kWave);
    }
    if (lane == 0) bstart[n_batches] = carry;
    const int n_v = carry;  // touches of this column inside the epoch's range
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < NE; ++u)
      if (bb[u] >= 0) bpos[bstart[bb[u]] + atomicAdd(&hist[bb[u]], 1)] = r[u];
    __builtin_amdgcn_wave_barrier();
    int place[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      place[u] = -1;
      if (u >= nu) break;  // wave-uniform
      if (bb[u] < 0) continue;
      const int s0 = bstart[bb[u]], s1 = bstart[bb[u] + 1];
      int rank = 0;
      for (int t = s0; t < s1; ++t) rank += bpos[t] < r[u] ? 1 : 0;
      place[u] = s0 + rank;
    }
    __builtin_amdgcn_wave_barrier();  // every lane is done reading the buckets: bpos is reused
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      if (u >= nu) break;
      if (place[u] < 0) continue;
      bpos[place[u]] = r[u];
      sx[place[u]] = xv[u];
      sq[place[u]] = u * kWave + lane;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int p = u * kWave + lane;
      if (u * kWave >= n_v) break;  // wave-uniform
      if (p >= n_v) continue;
      const int32_t rr = bpos[p];
      const int b = batch_of32(rr, batch32, first_singleton);
      int32_t lo;
      if (first_singleton) lo = b == 0 ? 0 : 1 + (b - 1) * (int32_t)batch32;
      else lo = b * (int32_t)batch32;
      const int32_t dst = (int32_t)(uint32_t)ps[(size_t)b * d + j] + (p - bstart[b]);
      tpos[dst] = rr - lo;
      tx[dst] = sx[p];
      if (tq) {
        const int q = sq[p];
        const int64_t i = crow[e0 + q];
        tq[dst] = toff[rr] + ((int64_t)cnz[e0 + q] - X.indptr[i]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- plans by bucketing: no sort over the epoch's touches ----
// A fresh order changes which samples share a batch, so the plan of a shuffled epoch is rebuilt from the rows.  The sort
// path orders all T (batch, feature) keys of the epoch with a device-wide radix sort (headline: 640 M pairs, four passes),
// although the touches ARRIVE grouped by batch and a feature's touches in a batch are few.  Here:
//   1. the feature range is cut into `nbk` buckets of 2^fl features; a "cell" is (batch, bucket), a few thousand touches.
//      k_seg_count: touches per cell (LDS histogram per chunk of samples, one global add per non-empty chunk bin); scan.
//   2. k_seg_scatter: every touch goes to its cell as ONE 64-bit item (feature's low bits, position in the batch, entry).
//      The order inside a cell is whatever the atomics gave: it is fixed in step 4 by comparing positions.
//   3. k_seg_classify: a workgroup per cell, histogram over the 2^fl features in LDS: features touched once get their
//      flag in sample order (sparse regime), the cell's number of column-phase features and touches is counted; scan.
//   4. k_seg_fill: the same histogram, then a touch's place = its feature's first slot + the number of touches of the
//      same feature at smaller positions (a feature has a handful: compared one by one) -- the stable order of the sort.
// The plan's arrays are bitwise those of the sort path (tests/test_gpu_plan_seg.py).  Falls back to the sort when a cell
// exceeds kSegCap touches (skewed popularity), when batches are tiny, or with dummy features.   NFM_PLAN_SEG=0: off.
constexpr int kSegCap = 4096;          // touches per cell a workgroup holds (16 per thread)
constexpr int kSegMaxBuckets = 4096;   // LDS histogram of the chunk kernels
constexpr int kSegPosBits = 26, kSegFlShift = 52;
// an item = feature's low bits << sh_fl | position in the batch << sh_pos | entry of the row; 32 bits when the three fit
// (headline: 11 + 13 + 6), else 64 (26 bits each for position and entry)
struct SegFmt {
  int sh_pos, sh_fl;
  uint64_t qmask, pmask;
};

struct SegGeo {
  const int64_t* bat_pos;   // device, n_batches + 1 (relative to begin)
  const int64_t* rowstart;  // first stored entry of the sample at every position
  const int64_t* toff;      // touch offset of every position (+ 1)
  int cpb, S, fl, nbk;      // chunks per batch, samples per chunk, features per bucket = 2^fl, buckets
  int64_t n_batches;
  int xcd;                  // seg_map
  SegFmt fmt;
};

// Workgroups are dealt to the 8 XCDs round-robin by their index.  All workgroups of ONE batch are put on ONE XCD: what they
// share -- the batch's rows (gathered values), its cells' item regions and its stretch of the single-touch flags, all
// written a few bytes at a time -- then meets in one L2, which merges the pieces into whole lines before they leave.
// xcd = 0: the plain order (workgroup = batch * per_batch + part).
__device__ __forceinline__ bool seg_map(int64_t blk, int per_batch, int64_t n_batches, int xcd, int64_t& b, int& part) {
  if (!xcd) {
    b = blk / per_batch;
    part = (int)(blk - b * per_batch);
    return true;
  }
  const int64_t g = blk / (8 * (int64_t)per_batch), r = blk - g * 8 * per_batch;
  b = g * 8 + (r & 7);
  part = (int)(r >> 3);
  return b < n_batches;
}
static inline int64_t seg_grid(int64_t n_batches, int per_batch, int xcd) {
  return (xcd ? (n_batches + 7) / 8 * 8 : n_batches) * per_batch;
}

__global__ void k_seg_rows(CsrView X, const int64_t* __restrict__ perm, int64_t begin, int64_t ns, int64_t* __restrict__ rowstart) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < ns; r += (int64_t)gridDim.x * blockDim.x)
    rowstart[r] = X.indptr[perm ? perm[r] : begin + r];
}

// every wavefront walks its rows four at a time (their first 64 ids requested together: a row at a time the walk was a
// chain of dependent round trips -- row start, ids, LDS -- per row); f(id, entry, position)
template <class F>
__device__ __forceinline__ void seg_walk_rows(const CsrView& X, const SegGeo& g, int64_t p0, int64_t p1, F&& f) {
  constexpr int U = 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  for (int64_t pb = p0 + (int64_t)wave * kWave; pb < p1; pb += (int64_t)kWavesPerBlock * kWave) {  // 64 rows per wavefront and trip
    const int64_t pl = pb + lane;
    const bool have = pl < p1;
    const int64_t q0_l = have ? g.rowstart[pl] : 0;
    const int m_l = have ? (int)(g.toff[pl + 1] - g.toff[pl]) : 0;
    const int rows = (int)(p1 - pb < kWave ? p1 - pb : kWave);
    for (int r = 0; r < rows; r += U) {
      int64_t q0[U];
      int m[U];
      int32_t j[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = r + u < rows ? r + u : r;
        q0[u] = __shfl((long long)q0_l, src, kWave);
        m[u] = r + u < rows ? __shfl(m_l, src, kWave) : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) j[u] = lane < m[u] ? X.indices[q0[u] + lane] : 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (lane < m[u]) f((uint32_t)j[u], lane, pb + r + u);
        for (int q = lane + kWave; q < m[u]; q += kWave) f((uint32_t)X.indices[q0[u] + q], q, pb + r + u);  // rows beyond 64 entries
      }
    }
  }
}

template <bool SCATTER, class IT>
__global__ __launch_bounds__(kBlock) void k_seg_chunks(CsrView X, SegGeo g, uint32_t* __restrict__ cellcnt,
                                                       const uint32_t* __restrict__ cellptr, IT* __restrict__ items) {
  extern __shared__ uint32_t seg_lds[];
  uint32_t* lh = seg_lds;          // [nbk] touches of this chunk per bucket, then the running rank
  uint32_t* lb = seg_lds + g.nbk;  // [nbk] (SCATTER) where the chunk's touches of a bucket start
  int64_t b;
  int c;
  if (!seg_map(blockIdx.x, g.cpb, g.n_batches, SCATTER ? g.xcd : 0, b, c)) return;
  const int64_t bp = g.bat_pos[b];
  const int64_t p0 = bp + (int64_t)c * g.S;
  int64_t p1 = p0 + g.S;
  if (p1 > g.bat_pos[b + 1]) p1 = g.bat_pos[b + 1];
  if (p0 >= p1) return;  // the whole workgroup
  for (int i = threadIdx.x; i < g.nbk; i += kBlock) lh[i] = 0;
  __syncthreads();
  const int fl = g.fl;
  seg_walk_rows(X, g, p0, p1, [&](uint32_t j, int, int64_t) { atomicAdd(&lh[j >> fl], 1u); });
  __syncthreads();
  const size_t cell0 = (size_t)b * g.nbk;
  if (!SCATTER) {
    for (int i = threadIdx.x; i < g.nbk; i += kBlock)
      if (lh[i]) atomicAdd(&cellcnt[cell0 + i], lh[i]);
    return;
  }
  for (int i = threadIdx.x; i < g.nbk; i += kBlock) {
    const uint32_t n = lh[i];
    if (n) lb[i] = cellptr[cell0 + i] + atomicAdd(&cellcnt[cell0 + i], n);  // cellcnt: zeroed again, the cells' cursors
    lh[i] = 0;
  }
  __syncthreads();
  const uint32_t lowmask = (1u << fl) - 1;
  seg_walk_rows(X, g, p0, p1, [&](uint32_t j, int q, int64_t p) {
    const uint32_t bk = j >> fl;
    const uint32_t slot = lb[bk] + atomicAdd(&lh[bk], 1u);
    items[slot] = (IT)(((uint64_t)(j & lowmask) << g.fmt.sh_fl) | ((uint64_t)(p - bp) << g.fmt.sh_pos) | (uint64_t)q);
  });
}

__global__ void k_seg_maxcell(int64_t cells, const uint32_t* __restrict__ cellcnt, unsigned int* __restrict__ out) {
  unsigned int mx = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x)
    mx = cellcnt[i] > mx ? cellcnt[i] : mx;
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned int o = __shfl_xor(mx, d);
    mx = o > mx ? o : mx;
  }
  if ((threadIdx.x & (kWave - 1)) == 0 && mx) atomicMax(out, mx);
}

// exclusive scan of one value per thread over the workgroup (kBlock threads); sh: kWavesPerBlock words of LDS
__device__ __forceinline__ uint64_t seg_block_scan(uint64_t v, uint64_t* sh, uint64_t* total) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  uint64_t inc = v;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const uint64_t t = (uint64_t)__shfl_up((unsigned long long)inc, d);
    if (lane >= d) inc += t;
  }
  if (lane == kWave - 1) sh[wave] = inc;
  __syncthreads();
  uint64_t woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWavesPerBlock; ++w) {
    const uint64_t x = sh[w];
    woff += w < wave ? x : 0;
    tot += x;
  }
  __syncthreads();
  if (total) *total = tot;
  return woff + inc - v;
}

// the same for NW words per thread (one pair of barriers for all of them)
template <int NW>
__device__ __forceinline__ void seg_block_scan_n(uint64_t (&v)[NW], uint64_t (*sh)[kWavesPerBlock]) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  uint64_t inc[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    inc[w] = v[w];
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const uint64_t t = (uint64_t)__shfl_up((unsigned long long)inc[w], d);
      if (lane >= d) inc[w] += t;
    }
    if (lane == kWave - 1) sh[w][wave] = inc[w];
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint64_t woff = 0;
#pragma unroll
    for (int q = 0; q < kWavesPerBlock; ++q) woff += q < wave ? sh[w][q] : 0;
    v[w] = woff + inc[w] - v[w];
  }
  __syncthreads();
}

// one workgroup per cell: flags of the single-touch features (thr = 2), the cell's column-phase features and touches
template <int IPT, class IT>  // items per thread: the workgroup holds up to kBlock * IPT touches of its cell
__global__ __launch_bounds__(kBlock) void k_seg_classify(SegFmt fmt, int nbk, int fl, int thr, const uint32_t* __restrict__ cellptr,
                                                          const IT* __restrict__ items, const int64_t* __restrict__ bat_pos,
                                                          const int64_t* __restrict__ toff, uint8_t* __restrict__ single,
                                                          uint64_t* __restrict__ cellUT, int by_count, uint16_t* __restrict__ cls,
                                                          unsigned long long* __restrict__ n_heavy, int64_t n_batches, int xcd) {
  extern __shared__ uint32_t seg_lds[];
  __shared__ uint64_t sh[kWavesPerBlock];
  __shared__ uint32_t cl[kCntClasses];  // column-phase features of the cell per touch-count class
  int64_t b_;
  int part_;
  if (!seg_map(blockIdx.x, nbk, n_batches, xcd, b_, part_)) return;
  const int64_t cell = b_ * nbk + part_;
  const uint32_t i0 = cellptr[cell];
  const int n = (int)(cellptr[cell + 1] - i0);
  if (n == 0) {
    if (threadIdx.x == 0) cellUT[cell] = 0;
    if (threadIdx.x < kCntClasses) cls[cell * kCntClasses + threadIdx.x] = 0;
    return;
  }
  if (threadIdx.x < kCntClasses) cl[threadIdx.x] = 0;
  const int NB = 1 << fl;
  for (int i = threadIdx.x; i < NB; i += kBlock) seg_lds[i] = 0;
  __syncthreads();
  uint64_t it[IPT];
#pragma unroll
  for (int e = 0; e < IPT; ++e) {
    const int idx = threadIdx.x + e * kBlock;
    if (idx < n) {
      it[e] = items[i0 + idx];
      atomicAdd(&seg_lds[it[e] >> fmt.sh_fl], 1u);
    }
  }
  __syncthreads();
  if (single) {
    const int64_t bp = bat_pos[cell / nbk];
#pragma unroll
    for (int e = 0; e < IPT; ++e) {
      const int idx = threadIdx.x + e * kBlock;
      if (idx < n && seg_lds[it[e] >> fmt.sh_fl] == 1u)
        single[toff[bp + (int64_t)((it[e] >> fmt.sh_pos) & fmt.pmask)] + (int64_t)(it[e] & fmt.qmask)] = 1;
    }
  }
  uint64_t ut = 0;
  unsigned int heavy = 0;
  for (int i = threadIdx.x; i < NB; i += kBlock) {
    const uint32_t c = seg_lds[i];
    if (c >= (uint32_t)thr) {
      ut += ((uint64_t)1 << 32) | c;
      atomicAdd(&cl[by_count ? count_class(c, kCntMax) : 0], 1u);
      heavy += c > (uint32_t)kHeavyTouches ? 1 : 0;
    }
  }
  uint64_t tot;
  seg_block_scan(ut, sh, &tot);  // (its barriers also cover cl)
  if (threadIdx.x == 0) cellUT[cell] = tot;
  if (threadIdx.x < kCntClasses) cls[cell * kCntClasses + threadIdx.x] = (uint16_t)cl[threadIdx.x];
  if (heavy) atomicAdd(n_heavy, (unsigned long long)heavy);
}

// one workgroup per batch: where the features of every (cell, class) start in the batch's list ordered by (class, feature):
// an exclusive scan over the batch's cells, class after class
__global__ __launch_bounds__(kBlock) void k_seg_unit_offsets(int nbk, const uint16_t* __restrict__ cls, uint32_t* __restrict__ uoffc) {
  __shared__ uint64_t sh[kWavesPerBlock];
  const size_t c0 = (size_t)blockIdx.x * nbk;
  const int len = kCntClasses * nbk;
  const int per = (len + kBlock - 1) / kBlock;
  const int s0 = threadIdx.x * per;
  uint64_t mine = 0;
  for (int s = s0; s < s0 + per && s < len; ++s) mine += cls[(c0 + (s % nbk)) * kCntClasses + s / nbk];
  uint64_t run = seg_block_scan(mine, sh, nullptr);
  for (int s = s0; s < s0 + per && s < len; ++s) {
    const size_t at = (c0 + (s % nbk)) * kCntClasses + s / nbk;
    uoffc[at] = (uint32_t)run;
    run += cls[at];
  }
}
__global__ void k_seg_batch_uoff(int64_t n_batches, int nbk, const uint64_t* __restrict__ cellOff, int64_t* __restrict__ out) {
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= n_batches; b += (int64_t)gridDim.x * blockDim.x)
    out[b] = (int64_t)(cellOff[b * nbk] >> 32);  // (cellOff has cells + 1 entries: the last one is the total)
}

struct SegOut {
  int32_t* tpos;
  double* tx;
  int64_t* tq;  // or null
  int32_t* ucol;
  int64_t* uptr;
  int32_t* ucol_s;  // the same features in the column phase's order (batch, class, feature)
  int64_t* ubeg_s;
  int32_t* ucnt_s;
};

// one workgroup per cell: every column-phase touch to its place in (feature, position) order
template <int IPT, class IT>
__global__ __launch_bounds__(kBlock) void k_seg_fill(SegFmt fmt, CsrView X, int nbk, int fl, int thr, const uint32_t* __restrict__ cellptr,
                                                      const IT* __restrict__ items, const int64_t* __restrict__ bat_pos,
                                                      const int64_t* __restrict__ rowstart, const int64_t* __restrict__ toff,
                                                      const uint64_t* __restrict__ cellOff, int by_count,
                                                      const uint32_t* __restrict__ uoffc, SegOut o, int64_t n_batches, int xcd) {
  extern __shared__ uint32_t seg_lds[];
  __shared__ uint64_t sh[kWavesPerBlock];
  __shared__ uint64_t sh4[kCntClasses / 4][kWavesPerBlock];
  __shared__ uint32_t uo[kCntClasses];
  int64_t b_;
  int part_;
  if (!seg_map(blockIdx.x, nbk, n_batches, xcd, b_, part_)) return;
  const int64_t cell = b_ * nbk + part_;
  const uint32_t i0 = cellptr[cell];
  const int n = (int)(cellptr[cell + 1] - i0);
  if (n == 0) return;
  const int NB = 1 << fl;
  uint32_t* A = seg_lds;        // [NB] count | first slot among all touches of the cell << 16
  uint32_t* Bv = seg_lds + NB;  // [NB] first column-phase touch | column-phase feature index << 16
  uint32_t* ps = seg_lds + 2 * NB;  // [kBlock * IPT] positions, grouped by feature
  uint16_t* Cv = reinterpret_cast<uint16_t*>(seg_lds + 2 * NB + kBlock * IPT);  // [NB] the feature's rank among the cell's features of its class
  if (threadIdx.x < kCntClasses) uo[threadIdx.x] = uoffc[cell * kCntClasses + threadIdx.x];
  for (int i = threadIdx.x; i < NB; i += kBlock) A[i] = 0;
  __syncthreads();
  uint64_t it[IPT];
  uint32_t rk[IPT];
#pragma unroll
  for (int e = 0; e < IPT; ++e) {
    const int idx = threadIdx.x + e * kBlock;
    if (idx < n) {
      it[e] = items[i0 + idx];
      rk[e] = atomicAdd(&A[it[e] >> fmt.sh_fl], 1u);
    }
  }
  __syncthreads();
  {  // bins [t * per, (t + 1) * per) belong to thread t
    const int per = NB / kBlock;  // fl >= 8
    uint64_t mine = 0;            // all touches | column-phase touches << 16 | column-phase features << 32
    for (int i = 0; i < per; ++i) {
      const uint32_t c = A[threadIdx.x * per + i];
      mine += (uint64_t)c + (c >= (uint32_t)thr ? ((uint64_t)c << 16) + ((uint64_t)1 << 32) : 0);
    }
    // ... and, per class, the column-phase features before this thread's bins: 16 counters of 16 bits in four words
    uint64_t cw[kCntClasses / 4] = {0, 0, 0, 0};
    for (int i = 0; i < per; ++i) {
      const uint32_t c = A[threadIdx.x * per + i];
      if (c >= (uint32_t)thr) {
        const uint32_t k = by_count ? count_class(c, kCntMax) : 0;
        const uint64_t inc = (uint64_t)1 << ((k & 3) * 16);
#pragma unroll
        for (int w = 0; w < kCntClasses / 4; ++w) cw[w] += (k >> 2) == (uint32_t)w ? inc : 0;
      }
    }
    uint64_t run = seg_block_scan(mine, sh, nullptr);
    seg_block_scan_n(cw, sh4);
    for (int i = 0; i < per; ++i) {
      const int bin = threadIdx.x * per + i;
      const uint32_t c = A[bin];
      A[bin] = c | ((uint32_t)(run & 0xFFFF) << 16);
      Bv[bin] = (uint32_t)((run >> 16) & 0xFFFF) | ((uint32_t)((run >> 32) & 0xFFFF) << 16);
      run += (uint64_t)c + (c >= (uint32_t)thr ? ((uint64_t)c << 16) + ((uint64_t)1 << 32) : 0);
      if (c >= (uint32_t)thr) {
        const uint32_t k = by_count ? count_class(c, kCntMax) : 0;
        const int sft = (int)(k & 3) * 16;
        uint64_t word = 0;
#pragma unroll
        for (int w = 0; w < kCntClasses / 4; ++w) word = (k >> 2) == (uint32_t)w ? cw[w] : word;
        Cv[bin] = (uint16_t)((word >> sft) & 0xFFFF);
        const uint64_t inc = (uint64_t)1 << sft;
#pragma unroll
        for (int w = 0; w < kCntClasses / 4; ++w) cw[w] += (k >> 2) == (uint32_t)w ? inc : 0;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < IPT; ++e) {
    const int idx = threadIdx.x + e * kBlock;
    if (idx < n) ps[(A[it[e] >> fmt.sh_fl] >> 16) + rk[e]] = (uint32_t)((it[e] >> fmt.sh_pos) & fmt.pmask);
  }
  __syncthreads();
  const int64_t b = cell / nbk;
  const int64_t bp = bat_pos[b];
  const uint64_t off = cellOff[cell];
  const int64_t u0 = (int64_t)(off >> 32), t0 = (int64_t)(off & 0xFFFFFFFFull);
  const int64_t ub0 = (int64_t)(cellOff[b * nbk] >> 32);  // the batch's first feature
  const int32_t jbase = (int32_t)((cell - b * nbk) << fl);
#pragma unroll
  for (int e = 0; e < IPT; ++e) {
    const int idx = threadIdx.x + e * kBlock;
    if (idx >= n) continue;
    const uint32_t bin = (uint32_t)(it[e] >> fmt.sh_fl);
    const uint32_t a = A[bin];
    const uint32_t c = a & 0xFFFF;
    if (c < (uint32_t)thr) continue;
    const uint32_t pos = (uint32_t)((it[e] >> fmt.sh_pos) & fmt.pmask);
    const uint32_t sb = a >> 16;
    int rank = 0;
    for (uint32_t i = 0; i < c; ++i) rank += ps[sb + i] < pos ? 1 : 0;
    const uint32_t bv = Bv[bin];
    const int64_t tb = t0 + (bv & 0xFFFF);
    const int64_t ts = tb + rank;
    const int64_t q = (int64_t)(it[e] & fmt.qmask);
    o.tpos[ts] = (int32_t)pos;
    o.tx[ts] = X.data[rowstart[bp + pos] + q];
    if (o.tq) o.tq[ts] = toff[bp + pos] + q;
    if (rank == 0) {
      const int64_t u = u0 + (bv >> 16);
      o.ucol[u] = jbase + (int32_t)bin;
      o.uptr[u] = tb;
      const int64_t us = ub0 + uo[by_count ? count_class(c, kCntMax) : 0] + Cv[bin];
      o.ucol_s[us] = jbase + (int32_t)bin;
      o.ubeg_s[us] = tb;
      o.ucnt_s[us] = (int32_t)c;
    }
  }
}

template <class KeyT>
static int plan_build_t(nfm_ctx* ctx, hipStream_t st, const CsrView& X, int n_aug, const int64_t* perm_host, const int64_t* perm_given_dev,
                        int64_t begin, int64_t end, int64_t batch, bool first_singleton, bool want_tq, bool use_singles,
                        bool sort_by_count, Plan* out, CscIndex* csc, const FeistelKey* perm_key) {
  static std::atomic<uint64_t> g_serial{0};  // ranks of one process build plans concurrently (dp.h)
  Plan& P = *out;
  P.release();
  P.serial = ++g_serial;
  const int64_t ns = end - begin;
  // with an explicit permutation the range counts positions of the index stream, which may be longer than the data (MBPSGD)
  NFM_CHECK(ns >= 0 && begin >= 0 && (end <= X.n || perm_host || perm_given_dev), NFM_ERR_INVALID, "epoch range [%lld,%lld) outside [0,%lld)",
            (long long)begin, (long long)end, (long long)X.n);
  NFM_CHECK(batch >= 1, NFM_ERR_INVALID, "batch must be >= 1");
  P.begin = begin; P.end = end; P.batch = batch; P.n_aug = n_aug;
  P.first_singleton = first_singleton; P.has_perm = perm_host != nullptr || perm_given_dev != nullptr; P.use_singles = use_singles;
  // batch boundaries
  P.bat_pos.clear();
  P.bat_pos.push_back(0);
  int64_t pos = 0;
  if (first_singleton && ns > 0) { pos = 1; P.bat_pos.push_back(1); }
  while (pos < ns) { pos = pos + batch < ns ? pos + batch : ns; P.bat_pos.push_back(pos); }
  P.n_batches = (int64_t)P.bat_pos.size() - 1;
  P.max_batch = 0;
  for (int64_t b = 0; b < P.n_batches; ++b) P.max_batch = std::max(P.max_batch, P.bat_pos[b + 1] - P.bat_pos[b]);
  P.bat_uoff.assign(P.n_batches + 1, 0);
  if (ns == 0) return NFM_OK;
  NFM_TRY(P.bat_pos_dev.alloc(sizeof(int64_t) * (P.n_batches + 1)));
  NFM_HIP_CHECK(hipMemcpyAsync(P.bat_pos_dev.p, P.bat_pos.data(), sizeof(int64_t) * (P.n_batches + 1), hipMemcpyHostToDevice, st));
  const int64_t* perm_dev = nullptr;
  if (perm_host || perm_given_dev) {  // perm_given_dev: ns sample ids already on the device (relative to begin)
    NFM_TRY(P.perm.alloc(sizeof(int64_t) * ns));
    if (perm_given_dev)
      NFM_HIP_CHECK(hipMemcpyAsync(P.perm.p, perm_given_dev, sizeof(int64_t) * ns, hipMemcpyDeviceToDevice, st));
    else
      NFM_HIP_CHECK(hipMemcpyAsync(P.perm.p, perm_host + begin, sizeof(int64_t) * ns, hipMemcpyHostToDevice, st));
    perm_dev = P.perm.as<int64_t>();
  }
  DevBuf bad;
  if (perm_host) {  // validated on the device (a host loop over 1e7 ids costs as much as a twentieth of the epoch)
    NFM_TRY(bad.alloc(sizeof(unsigned long long)));
    NFM_HIP_CHECK(hipMemsetAsync(bad.p, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_perm_range, dim3(256 * 4), dim3(kBlock), 0, st, ns, perm_dev, X.n, bad.as<unsigned long long>());
    unsigned long long h_bad = 0;
    NFM_HIP_CHECK(hipMemcpyAsync(&h_bad, bad.p, sizeof(h_bad), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));  // before any kernel indexes the data with these ids
    NFM_CHECK(h_bad == 0, NFM_ERR_INVALID, "%llu entries of the permutation are out of range [0,%lld)", h_bad, (long long)X.n);
  }
  // 1. row lengths -> touch offsets
  DevBuf len, toff, tmp;
  NFM_TRY(len.alloc(sizeof(int64_t) * (ns + 1)));
  NFM_TRY(toff.alloc(sizeof(int64_t) * (ns + 1)));
  hipLaunchKernelGGL(k_row_len, dim3(grid1d(ns + 1)), dim3(kBlock), 0, st, X, perm_dev, begin, ns, n_aug, len.as<int64_t>());
  size_t tmp_bytes = 0;
  NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, len.as<int64_t>(), toff.as<int64_t>(), (int)(ns + 1), st));
  NFM_TRY(tmp.alloc(tmp_bytes));
  NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, len.as<int64_t>(), toff.as<int64_t>(), (int)(ns + 1), st));
  int64_t T = 0;
  NFM_HIP_CHECK(hipMemcpyAsync(&T, toff.as<int64_t>() + ns, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  P.T = T;
  NFM_CHECK(T < (int64_t)2147483647, NFM_ERR_UNSUPPORTED, "more than 2^31-1 nnz in one epoch range (%lld)", (long long)T);
  len.release();
  if (T == 0) { NFM_TRY(P.uptr.alloc(sizeof(int64_t))); NFM_HIP_CHECK(hipMemsetAsync(P.uptr.p, 0, sizeof(int64_t), st)); return NFM_OK; }
  int fbits = 1;
  while (((int64_t)1 << fbits) < X.d + n_aug) ++fbits;
  int bbits = 1;
  while (((int64_t)1 << bbits) < P.n_batches) ++bbits;
  NFM_CHECK(fbits + bbits <= 64, NFM_ERR_UNSUPPORTED, "key overflow");
  int64_t U = 0, TM = 0;
  DevBuf bfu, ubatch;
  const int64_t none = INT64_MAX;
  NFM_TRY(bfu.alloc(sizeof(int64_t) * P.n_batches));
  hipLaunchKernelGGL(k_set_i64, dim3(grid1d(P.n_batches)), dim3(kBlock), 0, st, bfu.as<int64_t>(), P.n_batches, none);
  // Dense batches (no singles: every feature a batch touches goes to the column phase), no dummy features, an order
  // without repeats: the plan comes from the column-major copy of the data (above).  NFM_PLAN_CSC=0 switches it off.
  static const bool csc_on = !(getenv("NFM_PLAN_CSC") && atoi(getenv("NFM_PLAN_CSC")) == 0);
  bool use_csc = csc_on && csc && !use_singles && n_aug == 0 && end <= X.n && P.n_batches <= kCscMaxBatches &&
                 (double)P.n_batches * (double)X.d <= 268435456.0 && (double)T >= 0.5 * (double)P.n_batches * (double)X.d;
  if (use_csc && !csc->built) NFM_TRY(csc_build(ctx, st, X, csc));
  use_csc = use_csc && csc->usable;
  if (use_csc) {
    const int64_t cells = P.n_batches * X.d;
    DevBuf ipos, clash, cnt, ps, rpos;
    NFM_TRY(rpos.alloc(sizeof(int32_t) * X.nnz));
    NFM_TRY(clash.alloc(sizeof(unsigned long long)));
    NFM_HIP_CHECK(hipMemsetAsync(clash.p, 0, sizeof(unsigned long long), st));
    static const bool key_on = !(getenv("NFM_PLAN_KEY") && atoi(getenv("NFM_PLAN_KEY")) == 0);  // 0: the table, also for device-drawn orders
    const bool by_key = key_on && perm_key != nullptr && perm_key->ns == ns && perm_key->begin == begin;
    if (!by_key) {
      NFM_TRY(ipos.alloc(sizeof(int32_t) * X.n));
      NFM_HIP_CHECK(hipMemsetAsync(ipos.p, 0xFF, sizeof(int32_t) * X.n, st));  // -1: not in this epoch's range
      hipLaunchKernelGGL(k_ipos, dim3(grid1d(ns)), dim3(kBlock), 0, st, ns, perm_dev, begin, ipos.as<int32_t>(),
                         clash.as<unsigned long long>());
    }
    NFM_TRY(cnt.alloc(sizeof(int32_t) * (cells + 1)));
    NFM_TRY(ps.alloc(sizeof(uint64_t) * (cells + 1)));
    NFM_HIP_CHECK(hipMemsetAsync(cnt.as<int32_t>() + cells, 0, sizeof(int32_t), st));
    {
      const int64_t groups = (X.d + kCntGroup - 1) / kCntGroup;
      int64_t blocks = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
      if (blocks > 256 * 16) blocks = 256 * 16;
      const size_t lds_ = sizeof(int) * kWavesPerBlock * kCntGroup * (size_t)P.n_batches;
      if (by_key)
        hipLaunchKernelGGL(k_csc_count<true>, dim3((unsigned)blocks), dim3(kBlock), lds_, st, X.d, csc->cptr.as<int64_t>(), csc->crow.as<int32_t>(),
                           nullptr, batch, first_singleton ? 1 : 0, (int)P.n_batches, cnt.as<int32_t>(), rpos.as<int32_t>(), *perm_key);
      else
        hipLaunchKernelGGL(k_csc_count<false>, dim3((unsigned)blocks), dim3(kBlock), lds_, st, X.d, csc->cptr.as<int64_t>(), csc->crow.as<int32_t>(),
                           ipos.as<int32_t>(), batch, first_singleton ? 1 : 0, (int)P.n_batches, cnt.as<int32_t>(), rpos.as<int32_t>(), FeistelKey{});
    }
    NFM_HIP_CHECK(hipGetLastError());
    hipcub::TransformInputIterator<uint64_t, CellPack, const int32_t*> packed(cnt.as<int32_t>(), CellPack());
    tmp_bytes = 0;
    NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, packed, ps.as<uint64_t>(), (int)(cells + 1), st));
    NFM_TRY(tmp.alloc(tmp_bytes));
    NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, packed, ps.as<uint64_t>(), (int)(cells + 1), st));
    uint64_t h_tot = 0;
    unsigned long long h_clash = 0;
    NFM_HIP_CHECK(hipMemcpyAsync(&h_tot, ps.as<uint64_t>() + cells, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipMemcpyAsync(&h_clash, clash.p, sizeof(h_clash), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
    const int32_t U32 = (int32_t)(h_tot >> 32), T32 = (int32_t)(uint32_t)h_tot;
    if (h_clash != 0 || (int64_t)T32 != T) {
      use_csc = false;  // the order repeats a sample (an index stream, not a permutation): the general path below
    } else {
      U = U32;
      TM = T;
      P.U = U;
      P.TM = TM;
      NFM_TRY(P.tpos.alloc(sizeof(int32_t) * TM)); NFM_TRY(P.tx.alloc(sizeof(double) * TM));
      if (want_tq) NFM_TRY(P.tq.alloc(sizeof(int64_t) * TM));
      NFM_TRY(P.ucol.alloc(sizeof(int32_t) * U));
      NFM_TRY(P.uptr.alloc(sizeof(int64_t) * (U + 1)));
      NFM_TRY(ubatch.alloc(sizeof(int64_t) * (U + 1)));
      hipLaunchKernelGGL(k_csc_units, dim3(grid1d(cells)), dim3(kBlock), 0, st, cells, X.d, cnt.as<int32_t>(), ps.as<uint64_t>(),
                         P.ucol.as<int32_t>(), P.uptr.as<int64_t>(), ubatch.as<int64_t>());
      auto fill = [&](auto ne_tag) {
        constexpr int NE = decltype(ne_tag)::value;
        const size_t per_wave = (size_t)((4 * NE * kWave + 2 * P.n_batches + 2) & ~(int64_t)1);
        const int wpb = NE > 8 ? 2 : kWavesPerBlock;  // at most 48 KB of LDS per workgroup
        int64_t blocks = (X.d + wpb - 1) / wpb;
        if (blocks > 256 * 16 * (kWavesPerBlock / wpb)) blocks = 256 * 16 * (kWavesPerBlock / wpb);
        hipLaunchKernelGGL((k_csc_fill<NE>), dim3((unsigned)blocks), dim3(wpb * kWave), sizeof(int) * wpb * per_wave, st, X,
                           csc->cptr.as<int64_t>(), csc->crow.as<int32_t>(), csc->cval.as<double>(), csc->cnz.as<uint32_t>(),
                           rpos.as<int32_t>(), batch, first_singleton ? 1 : 0, (int)P.n_batches, ps.as<uint64_t>(),
                           toff.as<int64_t>(), P.tpos.as<int32_t>(), P.tx.as<double>(), want_tq ? P.tq.as<int64_t>() : nullptr);
      };
      if (csc->max_col <= 4 * kWave) fill(std::integral_constant<int, 4>{});
      else if (csc->max_col <= 8 * kWave) fill(std::integral_constant<int, 8>{});
      else fill(std::integral_constant<int, kCscMaxCol / kWave>{});
      if (U > 0)
        hipLaunchKernelGGL(k_batch_first, dim3(grid1d(U)), dim3(kBlock), 0, st, U, ubatch.as<int64_t>(), bfu.as<int64_t>());
      NFM_HIP_CHECK(hipGetLastError());
      NFM_HIP_CHECK(hipStreamSynchronize(st));  // the temporaries of this block go out of scope
    }
  }
  // Everything else whose cells fit: plans by bucketing (above).
  bool use_seg = false;
  unsigned long long seg_heavy = 0;  // (use_seg) features with more than kHeavyTouches touches
  {
    const char* e = getenv("NFM_PLAN_SEG");  // read per build: the tests switch it
    const bool seg_on = !(e && atoi(e) == 0);
    const double bt = (double)T / (double)P.n_batches;  // touches per batch
    if (!use_csc && seg_on && n_aug == 0 && bt >= 8192.0 && P.max_batch < ((int64_t)1 << kSegPosBits) &&
        X.max_row < (1 << kSegPosBits) && X.d < ((int64_t)1 << 31) && !(csc && csc->seg_unfit)) {
      static const int fl_env = getenv("NFM_SEG_FL") ? atoi(getenv("NFM_SEG_FL")) : 0;
      const double target = 2048.0;  // touches per cell aimed at (measured +-0 from 500 to 2000: no knob)
      int fl = 8;
      while (fl < 12 && (double)(((X.d - 1) >> (fl + 1)) + 1) * target >= bt) ++fl;  // the largest bucket count with >= target per cell
      if (fl_env >= 8 && fl_env <= 12) fl = fl_env;
      const int64_t nbk = ((X.d - 1) >> fl) + 1;
      const int64_t cells = P.n_batches * nbk;
      if (nbk <= kSegMaxBuckets && cells <= ((int64_t)1 << 23) && bt / (double)nbk <= 0.75 * kSegCap) {
        DevBuf rowstart, cellcnt, cellptr, items, mx, cellUT, cellOff;
        NFM_TRY(rowstart.alloc(sizeof(int64_t) * ns));
        NFM_TRY(cellcnt.alloc(sizeof(uint32_t) * (cells + 1)));
        NFM_TRY(cellptr.alloc(sizeof(uint32_t) * (cells + 1)));
        NFM_TRY(mx.alloc(sizeof(unsigned int)));
        NFM_HIP_CHECK(hipMemsetAsync(cellcnt.p, 0, sizeof(uint32_t) * (cells + 1), st));
        NFM_HIP_CHECK(hipMemsetAsync(mx.p, 0, sizeof(unsigned int), st));
        hipLaunchKernelGGL(k_seg_rows, dim3(grid1d(ns)), dim3(kBlock), 0, st, X, perm_dev, begin, ns, rowstart.as<int64_t>());
        SegGeo g;
        g.bat_pos = P.bat_pos_dev.as<int64_t>();
        g.rowstart = rowstart.as<int64_t>();
        g.toff = toff.as<int64_t>();
        static const int s_env = getenv("NFM_SEG_S") ? atoi(getenv("NFM_SEG_S")) : 0;
        g.S = s_env > 0 ? s_env : 256;  // samples per chunk workgroup: 64 rows per wavefront and trip (seg_walk_rows)
        g.cpb = (int)((P.max_batch + g.S - 1) / g.S);
        g.fl = fl;
        g.nbk = (int)nbk;
        g.n_batches = P.n_batches;
        int pbits = 1, qbits = 1;
        while (((int64_t)1 << pbits) < P.max_batch) ++pbits;
        while ((1 << qbits) < X.max_row) ++qbits;
        const bool narrow = fl + pbits + qbits <= 32;
        g.fmt = narrow ? SegFmt{qbits, qbits + pbits, ((uint64_t)1 << qbits) - 1, ((uint64_t)1 << pbits) - 1}
                       : SegFmt{kSegPosBits, kSegFlShift, ((uint64_t)1 << kSegPosBits) - 1, ((uint64_t)1 << kSegPosBits) - 1};
        static const int xcd = !(getenv("NFM_SEG_XCD") && atoi(getenv("NFM_SEG_XCD")) == 0) ? 1 : 0;
        g.xcd = xcd;
        const int64_t chunk_blocks = P.n_batches * g.cpb;
        NFM_CHECK(chunk_blocks < ((int64_t)1 << 31), NFM_ERR_UNSUPPORTED, "too many sample chunks");
        hipLaunchKernelGGL((k_seg_chunks<false, uint64_t>), dim3((unsigned)chunk_blocks), dim3(kBlock), sizeof(uint32_t) * 2 * nbk, st, X, g,
                           cellcnt.as<uint32_t>(), (const uint32_t*)nullptr, (uint64_t*)nullptr);
        hipLaunchKernelGGL(k_seg_maxcell, dim3(grid1d(cells)), dim3(kBlock), 0, st, cells, cellcnt.as<uint32_t>(), mx.as<unsigned int>());
        tmp_bytes = 0;
        NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cellcnt.as<uint32_t>(), cellptr.as<uint32_t>(), (int)(cells + 1), st));
        NFM_TRY(tmp.alloc(tmp_bytes));
        NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, cellcnt.as<uint32_t>(), cellptr.as<uint32_t>(), (int)(cells + 1), st));
        unsigned int h_mx = 0;
        NFM_HIP_CHECK(hipMemcpyAsync(&h_mx, mx.p, sizeof(h_mx), hipMemcpyDeviceToHost, st));
        NFM_HIP_CHECK(hipStreamSynchronize(st));
        if (h_mx > (unsigned)kSegCap) {
          if (csc) csc->seg_unfit = true;  // a popular feature: this dataset's plans come from the sort
        } else {
          use_seg = true;
          TimedLaunch tl(ctx, "plan_seg");  // (the tests count these to know which path built the plan)
          const int thr = use_singles ? 2 : 1;
          NFM_TRY(items.alloc((narrow ? sizeof(uint32_t) : sizeof(uint64_t)) * T));
          NFM_TRY(cellUT.alloc(sizeof(uint64_t) * (cells + 1)));
          NFM_TRY(cellOff.alloc(sizeof(uint64_t) * (cells + 1)));
          DevBuf cls, uoffc, nheavy, buoff;
          NFM_TRY(cls.alloc(sizeof(uint16_t) * kCntClasses * cells));
          NFM_TRY(uoffc.alloc(sizeof(uint32_t) * kCntClasses * cells));
          NFM_TRY(nheavy.alloc(sizeof(unsigned long long)));
          NFM_TRY(buoff.alloc(sizeof(int64_t) * (P.n_batches + 1)));
          NFM_HIP_CHECK(hipMemsetAsync(nheavy.p, 0, sizeof(unsigned long long), st));
          NFM_HIP_CHECK(hipMemsetAsync(cellcnt.p, 0, sizeof(uint32_t) * (cells + 1), st));
          NFM_HIP_CHECK(hipMemsetAsync(cellUT.as<uint64_t>() + cells, 0, sizeof(uint64_t), st));
          if (narrow)
            hipLaunchKernelGGL((k_seg_chunks<true, uint32_t>), dim3((unsigned)seg_grid(P.n_batches, g.cpb, xcd)), dim3(kBlock), sizeof(uint32_t) * 2 * nbk, st,
                               X, g, cellcnt.as<uint32_t>(), cellptr.as<uint32_t>(), items.as<uint32_t>());
          else
            hipLaunchKernelGGL((k_seg_chunks<true, uint64_t>), dim3((unsigned)seg_grid(P.n_batches, g.cpb, xcd)), dim3(kBlock), sizeof(uint32_t) * 2 * nbk, st,
                               X, g, cellcnt.as<uint32_t>(), cellptr.as<uint32_t>(), items.as<uint64_t>());
          if (use_singles) {
            NFM_TRY(P.single.alloc(sizeof(uint8_t) * T));
            NFM_HIP_CHECK(hipMemsetAsync(P.single.p, 0, sizeof(uint8_t) * T, st));
          }
          const size_t NB = (size_t)1 << fl;
          auto classify = [&](auto ipt) {
            constexpr int IPT = decltype(ipt)::value;
            auto go = [&](auto* it_ptr) {
              using IT = std::remove_pointer_t<decltype(it_ptr)>;
              hipLaunchKernelGGL((k_seg_classify<IPT, IT>), dim3((unsigned)seg_grid(P.n_batches, (int)nbk, xcd)), dim3(kBlock), sizeof(uint32_t) * NB, st,
                                 g.fmt, (int)nbk, fl, thr, cellptr.as<uint32_t>(), (const IT*)it_ptr, P.bat_pos_dev.as<int64_t>(), toff.as<int64_t>(),
                                 use_singles ? P.single.as<uint8_t>() : nullptr, cellUT.as<uint64_t>(), sort_by_count ? 1 : 0,
                                 cls.as<uint16_t>(), nheavy.as<unsigned long long>(), P.n_batches, xcd);
            };
            if (narrow) go(items.as<uint32_t>());
            else go(items.as<uint64_t>());
          };
          if (h_mx <= 4 * kBlock) classify(std::integral_constant<int, 4>{});
          else if (h_mx <= 8 * kBlock) classify(std::integral_constant<int, 8>{});
          else classify(std::integral_constant<int, kSegCap / kBlock>{});
          hipLaunchKernelGGL(k_seg_unit_offsets, dim3((unsigned)P.n_batches), dim3(kBlock), 0, st, (int)nbk, cls.as<uint16_t>(),
                             uoffc.as<uint32_t>());
          tmp_bytes = 0;
          NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cellUT.as<uint64_t>(), cellOff.as<uint64_t>(), (int)(cells + 1), st));
          NFM_TRY(tmp.alloc(tmp_bytes));
          NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, cellUT.as<uint64_t>(), cellOff.as<uint64_t>(), (int)(cells + 1), st));
          uint64_t h_tot = 0;
          NFM_HIP_CHECK(hipMemcpyAsync(&h_tot, cellOff.as<uint64_t>() + cells, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
          hipLaunchKernelGGL(k_seg_batch_uoff, dim3(grid1d(P.n_batches + 1)), dim3(kBlock), 0, st, P.n_batches, (int)nbk,
                             cellOff.as<uint64_t>(), buoff.as<int64_t>());
          NFM_HIP_CHECK(hipMemcpyAsync(P.bat_uoff.data(), buoff.p, sizeof(int64_t) * (P.n_batches + 1), hipMemcpyDeviceToHost, st));
          NFM_HIP_CHECK(hipMemcpyAsync(&seg_heavy, nheavy.p, sizeof(seg_heavy), hipMemcpyDeviceToHost, st));
          NFM_HIP_CHECK(hipStreamSynchronize(st));
          U = (int64_t)(h_tot >> 32);
          TM = (int64_t)(h_tot & 0xFFFFFFFFull);
          P.U = U;
          P.TM = TM;
          NFM_TRY(P.tpos.alloc(sizeof(int32_t) * TM)); NFM_TRY(P.tx.alloc(sizeof(double) * TM));
          if (want_tq) NFM_TRY(P.tq.alloc(sizeof(int64_t) * TM));
          NFM_TRY(P.ucol.alloc(sizeof(int32_t) * U));
          NFM_TRY(P.uptr.alloc(sizeof(int64_t) * (U + 1)));
          NFM_TRY(P.ucol_s.alloc(sizeof(int32_t) * std::max<int64_t>(U, 1)));
          NFM_TRY(P.ubeg_s.alloc(sizeof(int64_t) * std::max<int64_t>(U, 1)));
          NFM_TRY(P.ucnt_s.alloc(sizeof(int32_t) * std::max<int64_t>(U, 1)));
          SegOut so{P.tpos.as<int32_t>(), P.tx.as<double>(), want_tq ? P.tq.as<int64_t>() : nullptr, P.ucol.as<int32_t>(),
                    P.uptr.as<int64_t>(), P.ucol_s.as<int32_t>(), P.ubeg_s.as<int64_t>(), P.ucnt_s.as<int32_t>()};
          auto fill = [&](auto ipt) {
            constexpr int IPT = decltype(ipt)::value;
            auto go = [&](auto* it_ptr) {
              using IT = std::remove_pointer_t<decltype(it_ptr)>;
              hipLaunchKernelGGL((k_seg_fill<IPT, IT>), dim3((unsigned)seg_grid(P.n_batches, (int)nbk, xcd)), dim3(kBlock),
                                 sizeof(uint32_t) * (2 * NB + (size_t)kBlock * IPT) + sizeof(uint16_t) * NB, st, g.fmt, X, (int)nbk, fl, thr,
                                 cellptr.as<uint32_t>(), (const IT*)it_ptr, P.bat_pos_dev.as<int64_t>(), rowstart.as<int64_t>(),
                                 toff.as<int64_t>(), cellOff.as<uint64_t>(), sort_by_count ? 1 : 0, uoffc.as<uint32_t>(), so, P.n_batches, xcd);
            };
            if (narrow) go(items.as<uint32_t>());
            else go(items.as<uint64_t>());
          };
          if (TM > 0) {
            if (h_mx <= 4 * kBlock) fill(std::integral_constant<int, 4>{});
            else if (h_mx <= 8 * kBlock) fill(std::integral_constant<int, 8>{});
            else fill(std::integral_constant<int, kSegCap / kBlock>{});
          }
          NFM_HIP_CHECK(hipGetLastError());
          NFM_HIP_CHECK(hipStreamSynchronize(st));  // the temporaries of this block go out of scope
        }
      }
    }
  }
  if (!use_csc && !use_seg) {
  // 2. expand to (batch, feature) keys
  DevBuf k0, k1, v0, v1, pay_un;
  NFM_TRY(k0.alloc(sizeof(KeyT) * T)); NFM_TRY(k1.alloc(sizeof(KeyT) * T));
  NFM_TRY(v0.alloc(sizeof(uint32_t) * T)); NFM_TRY(v1.alloc(sizeof(uint32_t) * T));
  NFM_TRY(pay_un.alloc(sizeof(TouchPay) * T));
  {
    int64_t blocks = (ns + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_expand<KeyT>, dim3((unsigned)blocks), dim3(kBlock), 0, st, X, perm_dev, begin, ns, n_aug, batch,
                       first_singleton ? 1 : 0, fbits, toff.as<int64_t>(), k0.as<KeyT>(), v0.as<uint32_t>(),
                       pay_un.as<TouchPay>());
    NFM_HIP_CHECK(hipGetLastError());
  }
  // 3. stable sort by (batch, feature); ties keep sample order
  hipcub::DoubleBuffer<KeyT> dk(k0.as<KeyT>(), k1.as<KeyT>());
  hipcub::DoubleBuffer<uint32_t> dv(v0.as<uint32_t>(), v1.as<uint32_t>());
  tmp_bytes = 0;
  NFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, dk, dv, (int)T, 0, fbits + bbits, st));
  NFM_TRY(tmp.alloc(tmp_bytes));
  NFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, dk, dv, (int)T, 0, fbits + bbits, st));
  const KeyT* keys = dk.Current();
  const uint32_t* vals = dv.Current();
  // 4. classify: a feature touched exactly once in its batch is a "single" -- its update is applied
  //    by the row phase itself (flag per nnz in sample order); only features with >= 2 touches go
  //    to the column phase.  Their touches are compacted, payload in sorted order.
  DevBuf multi, mpos, head, uidx;
  if (use_singles) {
    NFM_TRY(P.single.alloc(sizeof(uint8_t) * T));
    NFM_HIP_CHECK(hipMemsetAsync(P.single.p, 0, sizeof(uint8_t) * T, st));
  }
  // flags and their scans in 32 bits (T < 2^31 is checked above): half the traffic of these passes
  NFM_TRY(multi.alloc(sizeof(int32_t) * (T + 1))); NFM_TRY(mpos.alloc(sizeof(int32_t) * (T + 1)));
  NFM_TRY(head.alloc(sizeof(int32_t) * (T + 1))); NFM_TRY(uidx.alloc(sizeof(int32_t) * (T + 1)));
  hipLaunchKernelGGL(k_classify<KeyT>, dim3(grid1d(T + 1)), dim3(kBlock), 0, st, T, keys, vals, use_singles ? 1 : 0,
                     use_singles ? P.single.as<uint8_t>() : nullptr, multi.as<int32_t>(), head.as<int32_t>());
  tmp_bytes = 0;
  NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, head.as<int32_t>(), uidx.as<int32_t>(), (int)(T + 1), st));
  NFM_TRY(tmp.alloc(tmp_bytes));
  NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, head.as<int32_t>(), uidx.as<int32_t>(), (int)(T + 1), st));
  NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, multi.as<int32_t>(), mpos.as<int32_t>(), (int)(T + 1), st));
  int32_t U32 = 0, TM32 = 0;
  NFM_HIP_CHECK(hipMemcpyAsync(&U32, uidx.as<int32_t>() + T, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipMemcpyAsync(&TM32, mpos.as<int32_t>() + T, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  U = U32;
  TM = TM32;
  P.U = U;
  P.TM = TM;
  NFM_TRY(P.tpos.alloc(sizeof(int32_t) * TM)); NFM_TRY(P.tx.alloc(sizeof(double) * TM));
  if (want_tq) NFM_TRY(P.tq.alloc(sizeof(int64_t) * TM));
  NFM_TRY(P.ucol.alloc(sizeof(int32_t) * U));
  NFM_TRY(P.uptr.alloc(sizeof(int64_t) * (U + 1)));
  NFM_TRY(ubatch.alloc(sizeof(int64_t) * (U + 1)));
  hipLaunchKernelGGL(k_compact<KeyT>, dim3(grid1d(T)), dim3(kBlock), 0, st, T, keys, vals, mpos.as<int32_t>(), uidx.as<int32_t>(),
                     fbits, pay_un.as<TouchPay>(),
                     P.tpos.as<int32_t>(), P.tx.as<double>(), want_tq ? P.tq.as<int64_t>() : nullptr, P.ucol.as<int32_t>(),
                     P.uptr.as<int64_t>(), ubatch.as<int64_t>());
  if (U > 0)
    hipLaunchKernelGGL(k_batch_first, dim3(grid1d(U)), dim3(kBlock), 0, st, U, ubatch.as<int64_t>(), bfu.as<int64_t>());
  NFM_HIP_CHECK(hipGetLastError());
  NFM_HIP_CHECK(hipStreamSynchronize(st));  // the sort's temporaries go out of scope with this block
  }
  NFM_HIP_CHECK(hipMemcpyAsync(P.uptr.as<int64_t>() + U, &TM, sizeof(int64_t), hipMemcpyHostToDevice, st));
  unsigned long long h_heavy = seg_heavy;
  if (!use_seg) {  // (the bucketing path has written the ordered tables and the batch offsets itself)
  // the per-batch order by descending touch count (the batch ranges [bat_uoff[b], bat_uoff[b+1]) are unchanged)
  NFM_TRY(P.ucol_s.alloc(sizeof(int32_t) * std::max<int64_t>(U, 1)));
  NFM_TRY(P.ubeg_s.alloc(sizeof(int64_t) * std::max<int64_t>(U, 1)));
  NFM_TRY(P.ucnt_s.alloc(sizeof(int32_t) * std::max<int64_t>(U, 1)));
  DevBuf n_heavy;
  NFM_TRY(n_heavy.alloc(sizeof(unsigned long long)));
  NFM_HIP_CHECK(hipMemsetAsync(n_heavy.p, 0, sizeof(unsigned long long), st));
  DevBuf uk0, uk1, uv0, uv1, utmp;  // alive until the synchronisation below
  if (U > 0) {
    NFM_CHECK(P.n_batches < ((int64_t)1 << 31), NFM_ERR_UNSUPPORTED, "too many batches");
    // (batch, descending touch count) in 32 bits, the count clamped to 4 bits (fewer when there are more than 2^28
    // batches): two radix passes instead of the five to six of a 64-bit (batch << 32 | count) key
    const int cbits = bbits <= 32 - kCntClassBits ? kCntClassBits : 32 - bbits;
    const uint32_t* order = nullptr;
    // Parameter rows shorter than a 128-byte line (k <= 8) keep the feature order: neighbours in the list are
    // neighbours in memory and share their lines, which is worth more than balanced wavefronts (cfg5, k = 8:
    // column phase 59 us in feature order, 64 us by count; cfg2, k = 16: 38 vs 35 us)
    if (sort_by_count && cbits >= 1) {
      NFM_TRY(uk0.alloc(sizeof(uint32_t) * U)); NFM_TRY(uk1.alloc(sizeof(uint32_t) * U));
      NFM_TRY(uv0.alloc(sizeof(uint32_t) * U)); NFM_TRY(uv1.alloc(sizeof(uint32_t) * U));
      hipcub::DoubleBuffer<uint32_t> udk(uk0.as<uint32_t>(), uk1.as<uint32_t>());
      hipcub::DoubleBuffer<uint32_t> udv(uv0.as<uint32_t>(), uv1.as<uint32_t>());
      hipLaunchKernelGGL(k_ukeys, dim3(grid1d(U)), dim3(kBlock), 0, st, U, ubatch.as<int64_t>(), P.uptr.as<int64_t>(), cbits,
                         uk0.as<uint32_t>(), uv0.as<uint32_t>());
      size_t ub = 0;
      NFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, ub, udk, udv, (int)U, 0, cbits + bbits, st));
      NFM_TRY(utmp.alloc(ub));
      NFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(utmp.p, ub, udk, udv, (int)U, 0, cbits + bbits, st));
      order = udv.Current();
    }
    hipLaunchKernelGGL(k_usorted, dim3(grid1d(U)), dim3(kBlock), 0, st, U, order, P.ucol.as<int32_t>(),
                       P.uptr.as<int64_t>(), P.ucol_s.as<int32_t>(), P.ubeg_s.as<int64_t>(), P.ucnt_s.as<int32_t>(),
                       n_heavy.as<unsigned long long>());
  }
  std::vector<int64_t> first(P.n_batches);
  NFM_HIP_CHECK(hipMemcpyAsync(first.data(), bfu.p, sizeof(int64_t) * P.n_batches, hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipMemcpyAsync(&h_heavy, n_heavy.p, sizeof(h_heavy), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));  // also: the sort's temporaries may go
  P.bat_uoff[P.n_batches] = U;
  for (int64_t b = P.n_batches - 1; b >= 0; --b) P.bat_uoff[b] = first[b] != none ? first[b] : P.bat_uoff[b + 1];
  }
  P.max_unique = 0;
  for (int64_t b = 0; b < P.n_batches; ++b) P.max_unique = std::max(P.max_unique, P.bat_uoff[b + 1] - P.bat_uoff[b]);
  // heavy features: flags -> scans -> compact lists + per-batch offsets
  P.bat_hoff.assign(P.n_batches + 1, 0);
  P.bat_soff.assign(P.n_batches + 1, 0);
  if (U > 0 && h_heavy == 0) {
    NFM_TRY(P.hv_u.alloc(sizeof(int64_t)));
    NFM_TRY(P.hv_seg0.alloc(sizeof(int64_t)));
  } else if (U > 0) {
    DevBuf hflag, nseg, hidx, segoff, uoff_dev, hoff_dev, soff_dev;
    NFM_TRY(hflag.alloc(sizeof(int64_t) * (U + 1))); NFM_TRY(nseg.alloc(sizeof(int64_t) * (U + 1)));
    NFM_TRY(hidx.alloc(sizeof(int64_t) * (U + 1))); NFM_TRY(segoff.alloc(sizeof(int64_t) * (U + 1)));
    hipLaunchKernelGGL(k_heavy_flags, dim3(grid1d(U + 1)), dim3(kBlock), 0, st, U, P.uptr.as<int64_t>(), hflag.as<int64_t>(),
                       nseg.as<int64_t>());
    tmp_bytes = 0;
    NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, hflag.as<int64_t>(), hidx.as<int64_t>(), (int)(U + 1), st));
    NFM_TRY(tmp.alloc(tmp_bytes));
    NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, hflag.as<int64_t>(), hidx.as<int64_t>(), (int)(U + 1), st));
    NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, nseg.as<int64_t>(), segoff.as<int64_t>(), (int)(U + 1), st));
    // offsets at the batch boundaries
    NFM_TRY(uoff_dev.alloc(sizeof(int64_t) * (P.n_batches + 1)));
    NFM_TRY(hoff_dev.alloc(sizeof(int64_t) * (P.n_batches + 1)));
    NFM_TRY(soff_dev.alloc(sizeof(int64_t) * (P.n_batches + 1)));
    NFM_HIP_CHECK(hipMemcpyAsync(uoff_dev.p, P.bat_uoff.data(), sizeof(int64_t) * (P.n_batches + 1), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gather_i64, dim3(grid1d(P.n_batches + 1)), dim3(kBlock), 0, st, P.n_batches + 1, hidx.as<int64_t>(),
                       uoff_dev.as<int64_t>(), hoff_dev.as<int64_t>());
    hipLaunchKernelGGL(k_gather_i64, dim3(grid1d(P.n_batches + 1)), dim3(kBlock), 0, st, P.n_batches + 1, segoff.as<int64_t>(),
                       uoff_dev.as<int64_t>(), soff_dev.as<int64_t>());
    NFM_HIP_CHECK(hipMemcpyAsync(P.bat_hoff.data(), hoff_dev.p, sizeof(int64_t) * (P.n_batches + 1), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipMemcpyAsync(P.bat_soff.data(), soff_dev.p, sizeof(int64_t) * (P.n_batches + 1), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
    P.H = P.bat_hoff[P.n_batches];
    P.HS = P.bat_soff[P.n_batches];
    NFM_TRY(P.hv_u.alloc(sizeof(int64_t) * std::max<int64_t>(P.H, 1)));
    NFM_TRY(P.hv_seg0.alloc(sizeof(int64_t) * (P.H + 1)));
    if (P.H > 0) {
      hipLaunchKernelGGL(k_heavy_compact, dim3(grid1d(U)), dim3(kBlock), 0, st, U, hidx.as<int64_t>(), segoff.as<int64_t>(),
                         P.hv_u.as<int64_t>(), P.hv_seg0.as<int64_t>());
      NFM_HIP_CHECK(hipMemcpyAsync(P.hv_seg0.as<int64_t>() + P.H, &P.HS, sizeof(int64_t), hipMemcpyHostToDevice, st));
      NFM_HIP_CHECK(hipStreamSynchronize(st));
    }
    for (int64_t b = 0; b < P.n_batches; ++b) {
      P.max_heavy = std::max(P.max_heavy, P.bat_hoff[b + 1] - P.bat_hoff[b]);
      P.max_segs = std::max(P.max_segs, P.bat_soff[b + 1] - P.bat_soff[b]);
    }
  }
  if (use_singles || want_tq) {  // the row phase finds a sample's per-nnz slots at toff[pos] + q
    P.toff.take(toff);
  }
  return NFM_OK;  // temporaries are released by their destructors
}

// ---- a fresh random order of the samples begin .. end-1, drawn on the device (nfm_opt_set_shuffle) ----
// The reference shuffles `indices` on the host with Nim's global generator once per epoch (optimizer/sgd.nim:297).  Here
// the order is a keyed pseudo-random BIJECTION evaluated per position: a balanced Feistel network of kFeistelRounds
// rounds over 2h bits (2^(2h) >= ns, < 4 ns), its round keys hashed from (seed, epoch), walked along its cycle until the
// value falls below ns ("cycle walking": a bijection of [0, 2^(2h)) restricted that way is a bijection of [0, ns)).
// No sort, no key collisions, reproducible from (seed, epoch), no host work and no upload; a radix sort of hashed
// 64-bit keys, which this replaces, cost a tenth of a cfg2 epoch.
__global__ void k_perm_feistel(int64_t ns, int64_t begin, FeistelKey K, int64_t* __restrict__ perm) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i;
    do x = feistel(x, K); while (x >= (uint64_t)ns);  // ends: x walks the cycle of i, which returns to i < ns
    perm[i] = begin + (int64_t)x;
  }
}
FeistelKey feistel_key(int64_t seed, uint64_t epoch, int64_t begin, int64_t ns) {
  FeistelKey K{};
  int bits = 1;
  while (((int64_t)1 << bits) < ns) ++bits;
  K.h = (bits + 1) / 2;
  K.mask = (uint32_t)(((uint64_t)1 << K.h) - 1);
  const uint64_t base = mix64((uint64_t)seed * 0x9E3779B97F4A7C15ull + epoch + 0x632BE59BD9B4E019ull);
  for (int r = 0; r < kFeistelRounds; ++r) K.k[r] = (uint32_t)(mix64(base ^ ((uint64_t)(r + 1) * 0xD1342543DE82EF95ull)) >> 32);
  K.ns = ns;
  K.begin = begin;
  return K;
}
int gen_permutation(nfm_ctx* ctx, hipStream_t st, int64_t seed, uint64_t epoch, int64_t begin, int64_t ns, DevBuf* out) {
  NFM_CHECK(ns < (int64_t)2147483647, NFM_ERR_UNSUPPORTED, "more than 2^31-1 samples in one shuffled epoch");
  NFM_TRY(out->ensure(sizeof(int64_t) * (size_t)std::max<int64_t>(ns, 1)));
  if (ns == 0) return NFM_OK;
  const FeistelKey K = feistel_key(seed, epoch, begin, ns);
  hipLaunchKernelGGL(k_perm_feistel, dim3(grid1d(ns)), dim3(kBlock), 0, st, ns, begin, K, out->as<int64_t>());
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

int plan_build(nfm_ctx* ctx, const CsrView& X, int n_aug, const int64_t* perm_host, int64_t begin, int64_t end, int64_t batch,
               bool first_singleton, bool want_tq, bool use_singles, bool sort_by_count, Plan* out, hipStream_t stream,
               const int64_t* perm_dev, CscIndex* csc, const FeistelKey* perm_key) {
  hipStream_t st = stream ? stream : ctx->stream;
  // (batch, feature) keys of at most 32 bits -- cfg2: 5 + 17, the headline shape: 11 + 20 -- sort as uint32: a third less
  // traffic in every pass of the radix sort and in the passes that read the sorted keys
  int fbits = 1, bbits = 1;
  while (((int64_t)1 << fbits) < X.d + n_aug) ++fbits;
  const int64_t nb = batch > 0 ? (end - begin + batch - 1) / batch + 1 : 1;
  while (((int64_t)1 << bbits) < nb) ++bbits;
  if (fbits + bbits <= 32)
    return plan_build_t<uint32_t>(ctx, st, X, n_aug, perm_host, perm_dev, begin, end, batch, first_singleton, want_tq, use_singles,
                                  sort_by_count, out, csc, perm_key);
  return plan_build_t<uint64_t>(ctx, st, X, n_aug, perm_host, perm_dev, begin, end, batch, first_singleton, want_tq, use_singles,
                                sort_by_count, out, csc, perm_key);
}

}  // namespace nfm
