// nimfm_amd/csrc/metrics.hip -- rmse, accuracy and rocauc of device-resident scores against the
// dataset's targets, so that score / evaluation callbacks do not move n-vectors to the host.
//
// Replaces (reference: /root/reference/src/nimfm/) metrics.nim:5-13 (rmse), :39-47 (accuracy), :76-103
// (rocauc) as used after decisionFunction by `score` (model/fm_base.nim:39-48).
//   rmse      sum (score - y)^2 by a fixed two-stage tree (bitwise reproducible; the reference adds left
//             to right, so the two agree to rounding), sqrt(sum / n)
//   accuracy  count of sgn(score) == sgn(y): an integer, exact
//   rocauc    scores sorted descending (hipcub radix sort), running true/false-positive counts by a
//             scan, one trapezoid per group of equal scores: sum (fp - fpPrev)(tp + tpPrev) is an
//             integer (< 2^53), so the result equals the reference's bit for bit
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace nfm {

__device__ __forceinline__ int sgn_d(double v) { return (v > 0) - (v < 0); }

__global__ __launch_bounds__(kBlock) void k_sqerr_partial(int64_t n, const double* __restrict__ s, const double* __restrict__ y,
                                                          double* __restrict__ part, unsigned long long* __restrict__ hits) {
  __shared__ double red[kBlock];
  double acc = 0.0;
  unsigned long long h = 0;
  // contiguous slice per block, strided inside the block: the summation order depends on n only
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += kBlock) {
    const double d = s[i] - y[i];
    acc += d * d;
    h += sgn_d(s[i]) == sgn_d(y[i]);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
  for (int sft = 1; sft < kWave; sft <<= 1) h += __shfl_xor(h, sft, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0 && h) atomicAdd(hits, h);
}

__global__ __launch_bounds__(kBlock) void k_sum_final(int n, const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double red[kBlock];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) acc += part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void k_pos_flags(int64_t n, const double* __restrict__ y, int32_t* __restrict__ pos) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    pos[i] = sgn_d(y[i]) == 1 ? 1 : 0;  // rocauc's pos = 1 (metrics.nim:76)
}

// after the sort and the scan: tp[i] = positives among the i+1 highest scores.  One trapezoid per
// group of equal scores, attributed to the group's last element e (first element f):
//   (fp_e - fp_{f-1}) * (tp_e + tp_{f-1})
__global__ void k_auc_groups(int64_t n, const double* __restrict__ key, const int64_t* __restrict__ tp,
                             const int64_t* __restrict__ gstart, unsigned long long* __restrict__ area2) {
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool last = i + 1 == n || key[i + 1] != key[i];
    if (!last) continue;
    const int64_t f = gstart[i];  // first index of this group
    const int64_t tp_e = tp[i], fp_e = (i + 1) - tp_e;
    const int64_t tp_p = f > 0 ? tp[f - 1] : 0, fp_p = f > 0 ? f - tp_p : 0;
    acc += (unsigned long long)((fp_e - fp_p) * (tp_e + tp_p));
  }
  for (int s = 1; s < kWave; s <<= 1) acc += __shfl_xor(acc, s, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0 && acc) atomicAdd(area2, acc);
}

// gstart[i] = index of the first element of i's group of equal keys: a max-scan over "i if a group starts at i"
__global__ void k_group_heads(int64_t n, const double* __restrict__ key, int64_t* __restrict__ head) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    head[i] = (i == 0 || key[i - 1] != key[i]) ? i : 0;
}

static inline unsigned grid_of(int64_t n, int cap) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

// scores, y: device arrays of n.  Any output pointer may be null.
int launch_metrics(nfm_ctx* ctx, int64_t n, const double* scores, const double* y, double* rmse, double* accuracy,
                   double* rocauc) {
  hipStream_t st = ctx->stream;
  NFM_CHECK(n > 0, NFM_ERR_INVALID, "metrics of an empty dataset");
  NFM_CHECK(n < (int64_t)2147483647, NFM_ERR_UNSUPPORTED, "more than 2^31-1 samples");
  if (rmse || accuracy) {
    const int nb = (int)grid_of(n, 1024);
    DevBuf part, out, hits;
    NFM_TRY(part.alloc(sizeof(double) * nb));
    NFM_TRY(out.alloc(sizeof(double)));
    NFM_TRY(hits.alloc(sizeof(unsigned long long)));
    NFM_HIP_CHECK(hipMemsetAsync(hits.p, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_sqerr_partial, dim3(nb), dim3(kBlock), 0, st, n, scores, y, part.as<double>(),
                       hits.as<unsigned long long>());
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kBlock), 0, st, nb, part.as<double>(), out.as<double>());
    double sum = 0.0;
    unsigned long long h = 0;
    NFM_HIP_CHECK(hipMemcpyAsync(&sum, out.p, sizeof(double), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipMemcpyAsync(&h, hits.p, sizeof(h), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
    if (rmse) *rmse = sqrt(sum / (double)n);
    if (accuracy) *accuracy = (double)h / (double)n;
  }
  if (rocauc) {
    DevBuf k0, k1, v0, v1, tp, head, gstart, tmp, area;
    NFM_TRY(k0.alloc(sizeof(double) * n)); NFM_TRY(k1.alloc(sizeof(double) * n));
    NFM_TRY(v0.alloc(sizeof(int32_t) * n)); NFM_TRY(v1.alloc(sizeof(int32_t) * n));
    NFM_TRY(tp.alloc(sizeof(int64_t) * n)); NFM_TRY(head.alloc(sizeof(int64_t) * n)); NFM_TRY(gstart.alloc(sizeof(int64_t) * n));
    NFM_TRY(area.alloc(sizeof(unsigned long long)));
    NFM_HIP_CHECK(hipMemcpyAsync(k0.p, scores, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_pos_flags, dim3(grid_of(n, 4096)), dim3(kBlock), 0, st, n, y, v0.as<int32_t>());
    hipcub::DoubleBuffer<double> dk(k0.as<double>(), k1.as<double>());
    hipcub::DoubleBuffer<int32_t> dv(v0.as<int32_t>(), v1.as<int32_t>());
    size_t bytes = 0;
    NFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, dk, dv, (int)n, 0, 64, st));
    NFM_TRY(tmp.alloc(bytes));
    NFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(tmp.p, bytes, dk, dv, (int)n, 0, 64, st));
    const double* key = dk.Current();
    const int32_t* pos = dv.Current();
    size_t b2 = 0;
    NFM_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, b2, pos, tp.as<int64_t>(), (int)n, st));
    NFM_TRY(tmp.ensure(b2));
    NFM_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(tmp.p, b2, pos, tp.as<int64_t>(), (int)n, st));
    hipLaunchKernelGGL(k_group_heads, dim3(grid_of(n, 4096)), dim3(kBlock), 0, st, n, key, head.as<int64_t>());
    size_t b3 = 0;
    NFM_HIP_CHECK(hipcub::DeviceScan::InclusiveScan(nullptr, b3, head.as<int64_t>(), gstart.as<int64_t>(), hipcub::Max(), (int)n, st));
    NFM_TRY(tmp.ensure(b3));
    NFM_HIP_CHECK(hipcub::DeviceScan::InclusiveScan(tmp.p, b3, head.as<int64_t>(), gstart.as<int64_t>(), hipcub::Max(), (int)n, st));
    NFM_HIP_CHECK(hipMemsetAsync(area.p, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_auc_groups, dim3(grid_of(n, 4096)), dim3(kBlock), 0, st, n, key, tp.as<int64_t>(), gstart.as<int64_t>(),
                       area.as<unsigned long long>());
    unsigned long long a2 = 0;
    int64_t np_ = 0;
    NFM_HIP_CHECK(hipMemcpyAsync(&a2, area.p, sizeof(a2), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipMemcpyAsync(&np_, tp.as<int64_t>() + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
    const int64_t nn_ = n - np_;
    *rocauc = ((double)a2 / 2.0) / (double)(nn_ * np_);  // 0/0 = NaN when one class is absent, as in the reference
  }
  return NFM_OK;
}

}  // namespace nfm
