// nimfm_amd/csrc/util.hip -- error/timing plumbing and the small streaming kernels around the hot
// path: layout conversion between the reference's parameter layouts and the device layout,
// the dense "finalize" rescale (optimizer/sgd.nim:99-113), squared norms for the verbose
// `regularization` line (optimizer/utils.nim:56-59).  All are plain HBM-bound grid-stride
// kernels, 16 B per lane where the layout allows.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <thread>
#include <mutex>

#include "common.h"

namespace nfm {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
const char* last_error() { return g_err; }

// ---- device memory: a small caching allocator ----
// hipMalloc / hipFree cost 0.1-1 ms each and hipFree synchronises the device; the batch plan of an epoch
// with a fresh permutation makes ~35 temporary allocations (8 of the 10 ms such an epoch took on cfg2).
// Released blocks are kept per device and handed out again to requests of similar size.  A block may still be in
// use by kernels ENQUEUED on the releasing thread's stream: handing it to the same host thread again is safe (its
// work is stream-ordered behind them), handing it to ANOTHER thread -- another context with its own stream: the ranks
// of a local data-parallel group on one GPU (nfm_dp_create_local) -- is not.  The free lists are therefore kept per
// (device, host thread); a request that finds nothing in its own list may take a block another thread released only
// after the whole device has drained (hipDeviceSynchronize: rare -- a rank's first allocations).  Found the hard way:
// 4 and 8 ranks on one GPU faulted now and then (a rank's plan tables overwritten by a neighbour's fresh buffer).
// NFM_POOL=0 disables the cache, NFM_POOL_MAX_GB (default 64) bounds what it keeps.
namespace {
struct BlockPool {
  std::mutex mu;
  std::map<std::thread::id, std::multimap<size_t, void*>> free_blocks[16];
  size_t kept = 0;
  bool enabled = !(getenv("NFM_POOL") && atoi(getenv("NFM_POOL")) == 0);
  size_t max_kept = (size_t)(getenv("NFM_POOL_MAX_GB") ? atof(getenv("NFM_POOL_MAX_GB")) : 64.0) << 30;
};
BlockPool& pool() {
  static BlockPool* p = new BlockPool();  // never destroyed: buffers may be released during process exit
  return *p;
}
size_t round_request(size_t n) {
  if (n <= (1u << 20)) return (n + 255) / 256 * 256;
  return (n + (1u << 20) - 1) >> 20 << 20;  // whole MiB: sizes that differ a little between epochs match
}
bool fits(size_t have, size_t want) { return have >= want && have <= want + want / 4 + (1u << 20); }
}  // namespace

int DevBuf::alloc(size_t nbytes) {
  release();
  if (nbytes == 0) nbytes = 16;
  const size_t want = round_request(nbytes);
  BlockPool& bp = pool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (bp.enabled && dev >= 0 && dev < 16) {
    const std::thread::id me = std::this_thread::get_id();
    void* taken = nullptr;  // a block ANOTHER thread released: removed from its list under the lock, used after a device sync
    size_t taken_cap = 0;
    {
      std::lock_guard<std::mutex> lk(bp.mu);
      auto mine = bp.free_blocks[dev].find(me);
      if (mine != bp.free_blocks[dev].end()) {
        auto it = mine->second.lower_bound(want);
        if (it != mine->second.end() && fits(it->first, want)) {
          p = it->second;
          cap = it->first;
          bytes = nbytes;
          device = dev;
          bp.kept -= it->first;
          mine->second.erase(it);
          return NFM_OK;
        }
      }
      for (auto& kv : bp.free_blocks[dev]) {
        if (kv.first == me) continue;
        auto it = kv.second.lower_bound(want);
        if (it != kv.second.end() && fits(it->first, want)) {
          taken = it->second;
          taken_cap = it->first;
          bp.kept -= it->first;
          kv.second.erase(it);
          break;
        }
      }
    }
    if (taken) {
      // The block left its owner's list while the lock was held, so whatever the owner had enqueued on it was enqueued
      // BEFORE this point: the device-wide drain below covers it.  (Looking the lists over again AFTER the drain -- the
      // earlier version -- could pick a block released in between, its kernels still in flight.)
      (void)hipDeviceSynchronize();
      p = taken;
      cap = taken_cap;
      bytes = nbytes;
      device = dev;
      return NFM_OK;
    }
  }
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess && bp.enabled) {  // out of memory: give the cached blocks back and retry
    {
      std::lock_guard<std::mutex> lk(bp.mu);
      for (auto& per_dev : bp.free_blocks) {
        for (auto& per_thread : per_dev)
          for (auto& kv : per_thread.second) (void)hipFree(kv.second);
        per_dev.clear();
      }
      bp.kept = 0;
    }
    (void)hipGetLastError();
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess) {
    p = nullptr;
    return set_error(NFM_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  }
  bytes = nbytes;
  cap = want;
  device = dev;
  return NFM_OK;
}
void DevBuf::release() {
  if (p) {
    BlockPool& bp = pool();
    bool kept = false;
    if (bp.enabled && device >= 0 && device < 16 && cap > 0) {
      std::lock_guard<std::mutex> lk(bp.mu);
      if (bp.kept + cap <= bp.max_kept) {
        bp.free_blocks[device][std::this_thread::get_id()].emplace(cap, p);
        bp.kept += cap;
        kept = true;
      }
    }
    if (!kept) (void)hipFree(p);
  }
  p = nullptr;
  bytes = 0;
  cap = 0;
}

// ---- timing ----
TimedLaunch::TimedLaunch(nfm_ctx* c, const char* f) : ctx(c), family(f) {
  if (!ctx->timing.enabled) return;
  auto get = [&]() {
    hipEvent_t e = nullptr;
    if (!ctx->timing.pool.empty()) {
      e = ctx->timing.pool.back();
      ctx->timing.pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
      e = nullptr;
    }
    return e;
  };
  start = get();
  stop = get();
  if (start) (void)hipEventRecord(start, ctx->stream);
}
TimedLaunch::~TimedLaunch() {
  if (!start || !stop) return;
  (void)hipEventRecord(stop, ctx->stream);
  ctx->timing.pending.push_back({family, start, stop});
}
int timing_flush(nfm_ctx* ctx) {
  for (auto& p : ctx->timing.pending) {
    float ms = 0.f;
    NFM_HIP_CHECK(hipEventSynchronize(p.stop));
    NFM_HIP_CHECK(hipEventElapsedTime(&ms, p.start, p.stop));
    auto& a = ctx->timing.acc[p.family];
    a.launches += 1;
    a.ms += ms;
    ctx->timing.pool.push_back(p.start);
    ctx->timing.pool.push_back(p.stop);
  }
  ctx->timing.pending.clear();
  return NFM_OK;
}

static inline unsigned grid_for(int64_t n, int per_thread = 1) {
  int64_t b = (n + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
  if (b < 1) b = 1;
  if (b > 256 * 16) b = 256 * 16;
  return (unsigned)b;
}

__global__ void k_fill(double* __restrict__ p, int64_t n, double v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
int launch_fill(nfm_ctx* ctx, double* p, int64_t n, double v) {
  if (n <= 0) return NFM_OK;
  hipLaunchKernelGGL(k_fill, dim3(grid_for(n)), dim3(kBlock), 0, ctx->stream, p, n, v);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

// reference FM model layout [nb][k][da]  ->  device [nb][da][Kp] (zero padded).
// Tiled through LDS so that both the read (along j) and the write (along s) are coalesced.
__global__ void k_fm_to_device(const double* __restrict__ src, double* __restrict__ dst, int k, int Kp, int64_t da, int64_t bs, int64_t rs,
                               int b0) {
  __shared__ double tile[32][33];
  const int o = blockIdx.z;
  const int64_t j0 = (int64_t)blockIdx.x * 32;
  const int s0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int s = s0 + r;
    const int64_t j = j0 + tx;
    tile[r][tx] = (s < k && j < da) ? src[((size_t)o * k + s) * da + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t j = j0 + r;
    const int s = s0 + tx;
    if (j < da && s < Kp) dst[((size_t)(b0 + o) * bs + (size_t)j * rs) * Kp + s] = tile[tx][r];  // ModelView::row
  }
}
int launch_fm_to_device(nfm_ctx* ctx, const double* src_ref, double* dst_dev, int nb, int k, int Kp, int64_t da, int64_t bs, int64_t rs,
                        int b0) {
  if (nb == 0 || da == 0) return NFM_OK;
  if (bs == 0) { bs = da; rs = 1; }  // order-major
  dim3 grid((unsigned)((da + 31) / 32), (unsigned)((Kp + 31) / 32), (unsigned)nb);
  hipLaunchKernelGGL(k_fm_to_device, grid, dim3(kBlock), 0, ctx->stream, src_ref, dst_dev, k, Kp, da, bs, rs, b0);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

__global__ void k_fm_from_device(const double* __restrict__ src, double* __restrict__ dst, int k, int Kp, int64_t da,
                                 const double* __restrict__ scale, int64_t bs, int64_t rs, int b0) {
  __shared__ double tile[32][33];
  const double sc = scale ? *scale : 1.0;
  const int o = blockIdx.z;
  const int64_t j0 = (int64_t)blockIdx.x * 32;
  const int s0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int64_t j = j0 + r;
    const int s = s0 + tx;
    tile[r][tx] = (j < da && s < Kp) ? src[((size_t)(b0 + o) * bs + (size_t)j * rs) * Kp + s] * sc : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int s = s0 + r;
    const int64_t j = j0 + tx;
    if (s < k && j < da) dst[((size_t)o * k + s) * da + j] = tile[tx][r];
  }
}
int launch_fm_from_device(nfm_ctx* ctx, const double* src_dev, double* dst_ref, int nb, int k, int Kp, int64_t da,
                          const double* scale_dev, int64_t bs, int64_t rs, int b0) {
  if (nb == 0 || da == 0) return NFM_OK;
  if (bs == 0) { bs = da; rs = 1; }  // order-major
  dim3 grid((unsigned)((da + 31) / 32), (unsigned)((Kp + 31) / 32), (unsigned)nb);
  hipLaunchKernelGGL(k_fm_from_device, grid, dim3(kBlock), 0, ctx->stream, src_dev, dst_ref, k, Kp, da, scale_dev, bs, rs, b0);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

// [rows][k] <-> [rows][Kp] (FFM parameters and every AdaGrad state tensor)
// nb > 0: device row j * nb + b holds reference row b * (rows / nb) + j (feature-major FFM layout, common.h)
__global__ void k_rows_to_device(const double* __restrict__ src, double* __restrict__ dst, int64_t rows, int k, int Kp,
                                 double pad, int nb) {
  const int64_t total = rows * Kp, da = nb > 0 ? rows / nb : 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / Kp;
    const int s = (int)(e % Kp);
    const int64_t rr = nb > 0 ? (r % nb) * da + r / nb : r;
    dst[e] = s < k ? src[rr * k + s] : pad;
  }
}
int launch_rows_to_device(nfm_ctx* ctx, const double* src_ref, double* dst_dev, int64_t rows, int k, int Kp, double pad, int nb_major) {
  if (rows == 0) return NFM_OK;
  hipLaunchKernelGGL(k_rows_to_device, dim3(grid_for(rows * Kp)), dim3(kBlock), 0, ctx->stream, src_ref, dst_dev, rows, k, Kp, pad, nb_major);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}
__global__ void k_rows_from_device(const double* __restrict__ src, double* __restrict__ dst, int64_t rows, int k, int Kp,
                                   const double* __restrict__ scale, int nb) {
  const double sc = scale ? *scale : 1.0;
  const int64_t total = rows * k, da = nb > 0 ? rows / nb : 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / k;  // reference row b * da + j
    const int s = (int)(e % k);
    const int64_t rr = nb > 0 ? (r % da) * nb + r / da : r;
    dst[e] = src[rr * Kp + s] * sc;
  }
}
int launch_rows_from_device(nfm_ctx* ctx, const double* src_dev, double* dst_ref, int64_t rows, int k, int Kp,
                            const double* scale_dev, int nb_major) {
  if (rows == 0) return NFM_OK;
  hipLaunchKernelGGL(k_rows_from_device, dim3(grid_for(rows * k)), dim3(kBlock), 0, ctx->stream, src_dev, dst_ref, rows, k, Kp, scale_dev, nb_major);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

// FMs with more than 128 factors (ModelView::kc): reference [no * da][k] <-> device [no * kc][da][Kp]; device block
// o * kc + c holds the factors c * kb ... c * kb + kb - 1 of order o
__global__ void k_rows_split_to_device(const double* __restrict__ src, double* __restrict__ dst, int64_t no, int64_t da, int k, int kc,
                                       int kb, int Kp, double pad, int64_t bs, int64_t rs) {
  const int64_t total = no * kc * da * Kp;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int sl = (int)(e % Kp);
    const int64_t r = e / Kp;  // (block b = o * kc + c, feature j), enumerated block-major
    const int64_t j = r % da, b = r / da;
    const int64_t o = b / kc;
    const int s = (int)(b % kc) * kb + sl;
    dst[(b * bs + j * rs) * Kp + sl] = (sl < kb && s < k) ? src[(o * da + j) * k + s] : pad;  // ModelView::row
  }
}
int launch_rows_split_to_device(nfm_ctx* ctx, const double* src_ref, double* dst_dev, int64_t no, int64_t da, int k, int kc, int kb, int Kp,
                                double pad, int64_t bs, int64_t rs) {
  if (no == 0 || da == 0) return NFM_OK;
  hipLaunchKernelGGL(k_rows_split_to_device, dim3(grid_for(no * kc * da * Kp)), dim3(kBlock), 0, ctx->stream, src_ref, dst_dev, no, da, k, kc,
                     kb, Kp, pad, bs, rs);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}
__global__ void k_rows_split_from_device(const double* __restrict__ src, double* __restrict__ dst, int64_t no, int64_t da, int k, int kc,
                                         int kb, int Kp, const double* __restrict__ scale, int64_t bs, int64_t rs) {
  const double sc = scale ? *scale : 1.0;
  const int64_t total = no * da * k;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(e % k);
    const int64_t r = e / k;  // reference row o * da + j
    const int64_t j = r % da, o = r / da;
    dst[e] = src[((o * kc + s / kb) * bs + j * rs) * Kp + s % kb] * sc;
  }
}
int launch_rows_split_from_device(nfm_ctx* ctx, const double* src_dev, double* dst_ref, int64_t no, int64_t da, int k, int kc, int kb, int Kp,
                                  const double* scale_dev, int64_t bs, int64_t rs) {
  if (no == 0 || da == 0) return NFM_OK;
  hipLaunchKernelGGL(k_rows_split_from_device, dim3(grid_for(no * da * k)), dim3(kBlock), 0, ctx->stream, src_dev, dst_ref, no, da, k, kc, kb,
                     Kp, scale_dev, bs, rs);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

// finalize (optimizer/sgd.nim:99-113): with one global scale per tensor it is a dense multiply.
__global__ void k_rescale(double* __restrict__ p, int64_t n2 /*double2 count*/, const double* __restrict__ scale) {
  const double sc = *scale;
  double2* p2 = reinterpret_cast<double2*>(p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    double2 v = p2[i];
    v.x *= sc;
    v.y *= sc;
    p2[i] = v;
  }
}
__global__ void k_rescale_scalar(double* __restrict__ p, int64_t n, const double* __restrict__ scale) {
  const double sc = *scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] *= sc;
}
__global__ void k_reset_scales(double* sc, int reset_w) {
  sc[SC_SCALE_P] = 1.0;
  if (reset_w) sc[SC_SCALE_W] = 1.0;
}
int launch_rescale(nfm_ctx* ctx, const ModelView& M) {
  TimedLaunch tl(ctx, "rescale");
  const int64_t nP = (int64_t)M.nb * M.da * M.Kp;
  if (nP > 0)
    hipLaunchKernelGGL(k_rescale, dim3(grid_for(nP / 2)), dim3(kBlock), 0, ctx->stream, M.P, nP / 2, M.sc + SC_SCALE_P);
  if (M.fit_linear && M.d > 0)
    hipLaunchKernelGGL(k_rescale_scalar, dim3(grid_for(M.d)), dim3(kBlock), 0, ctx->stream, M.w, M.d, M.sc + SC_SCALE_W);
  hipLaunchKernelGGL(k_reset_scales, dim3(1), dim3(1), 0, ctx->stream, M.sc, M.fit_linear);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

// sum of squares, deterministic: fixed grid, per-block tree in LDS, fixed-order final pass.
__global__ void k_sqnorm_partial(const double* __restrict__ p, int64_t n, double scale_idx_unused, double* __restrict__ part) {
  __shared__ double red[kBlock];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += p[i] * p[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void k_sqnorm_final(const double* __restrict__ part, int nparts, const double* __restrict__ scale, double* __restrict__ out) {
  __shared__ double red[kBlock];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kBlock) acc += part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0] * (*scale) * (*scale);
}
int launch_sqnorms(nfm_ctx* ctx, const ModelView& M, double* out2_dev) {
  const int nparts = 1024;
  DevBuf part;
  NFM_TRY(part.alloc(sizeof(double) * nparts));
  const int64_t nP = (int64_t)M.nb * M.da * M.Kp;
  hipLaunchKernelGGL(k_sqnorm_partial, dim3(nparts), dim3(kBlock), 0, ctx->stream, M.P, nP, 0.0, part.as<double>());
  hipLaunchKernelGGL(k_sqnorm_final, dim3(1), dim3(kBlock), 0, ctx->stream, part.as<double>(), nparts, M.sc + SC_SCALE_P, out2_dev);
  hipLaunchKernelGGL(k_sqnorm_partial, dim3(nparts), dim3(kBlock), 0, ctx->stream, M.w, M.d, 0.0, part.as<double>());
  hipLaunchKernelGGL(k_sqnorm_final, dim3(1), dim3(kBlock), 0, ctx->stream, part.as<double>(), nparts, M.sc + SC_SCALE_W, out2_dev + 1);
  NFM_HIP_CHECK(hipGetLastError());
  NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  part.release();
  return NFM_OK;
}

__global__ void k_narrow(const int64_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (int32_t)src[i];
}
int launch_narrow_i64_i32(nfm_ctx* ctx, const int64_t* src, int32_t* dst, int64_t n) {
  if (n <= 0) return NFM_OK;
  hipLaunchKernelGGL(k_narrow, dim3(grid_for(n)), dim3(kBlock), 0, ctx->stream, src, dst, n);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

}  // namespace nfm
