// tools/membench.hip -- what this MI355X's memory system delivers for the access shapes of the FM hot
// path, so DESIGN.md can quote measured ceilings next to the 8 TB/s spec:
//   stream read / copy, random gather of R-byte rows (16 B per lane, a row = R/16 consecutive lanes),
//   random read-modify-write of distinct rows, and gather + RMW of a fraction of the same rows.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/membench tools/membench.hip   (tools/bin/ is not tracked)
// usage: membench [table_MB] [row_bytes] [rows_per_launch]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int kBlock = 256;

__global__ void k_stream_read(const double2* __restrict__ a, size_t n, double* out) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const double2 v = a[i];
    s += v.x + v.y;
  }
  if (s == 12345.678) out[0] = s;
}

__global__ void k_stream_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) b[i] = a[i];
}

// LPR lanes per row; every lane group walks U rows at a time (all requested before the first use)
template <int LPR, int U, int MODE>  // MODE 0: gather, 1: RMW, 2: gather all then RMW the rows with flag
__global__ __launch_bounds__(kBlock) void k_rows(double2* __restrict__ tab, const int* __restrict__ idx,
                                                 const uint8_t* __restrict__ flag, size_t n_rows, double* out) {
  const size_t g = ((size_t)blockIdx.x * kBlock + threadIdx.x) / LPR;
  const int l = threadIdx.x % LPR;
  const size_t n_groups = (size_t)gridDim.x * kBlock / LPR;
  double s = 0.0;
  for (size_t r = g * U; r < n_rows; r += n_groups * U) {
    int j[U];
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) j[u] = r + u < n_rows ? idx[r + u] : idx[r];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = tab[(size_t)j[u] * LPR + l];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 0) {
        s += v[u].x + v[u].y;
      } else if (r + u < n_rows) {
        v[u].x = v[u].x * 0.999 + 1e-3;
        v[u].y = v[u].y * 0.999 - 1e-3;
        tab[(size_t)j[u] * LPR + l] = v[u];
      }
    }
  }
  if (s == 12345.678) out[0] = s;
}

// the row phase's shape: one lane group per "sample" of M rows: gather all M (sum), then RMW the flagged
template <int LPR, int U>
__global__ __launch_bounds__(kBlock) void k_sample(double2* __restrict__ tab, const int* __restrict__ idx,
                                                   const uint8_t* __restrict__ flag, size_t n_samples, int M, double* out) {
  const size_t g = ((size_t)blockIdx.x * kBlock + threadIdx.x) / LPR;
  const int l = threadIdx.x % LPR;
  if (g >= n_samples) return;
  const int* ix = idx + g * M;
  const uint8_t* fl = flag + g * M;
  double2 a = {0.0, 0.0};
  for (int q = 0; q < M; q += U) {
    int j[U];
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) j[u] = ix[q + u];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = tab[(size_t)j[u] * LPR + l];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a.x += v[u].x;
      a.y += v[u].y;
    }
  }
  for (int q = 0; q < M; q += U) {
    int j[U];
    bool f[U];
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      j[u] = ix[q + u];
      f[u] = fl[q + u] != 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (f[u]) v[u] = tab[(size_t)j[u] * LPR + l];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (f[u]) {
        v[u].x = v[u].x * 0.999 + 1e-9 * a.x;
        v[u].y = v[u].y * 0.999 + 1e-9 * a.y;
        tab[(size_t)j[u] * LPR + l] = v[u];
      }
  }
}


// same work, rows kept on chip between the gather and the RMW (no second visit of HBM):
// REG: every lane keeps its M/2 row pieces in registers (wave = sample, two 32-lane groups take
// alternate rows); LDS: the flagged rows are parked in LDS (CAP rows per group, the rest re-read).
template <int M>
__global__ __launch_bounds__(kBlock) void k_sample_reg(double2* __restrict__ tab, const int* __restrict__ idx,
                                                       const uint8_t* __restrict__ flag, size_t n_samples, double* out) {
  constexpr int LPR = 32, NQ = M / 2;
  const size_t smp = ((size_t)blockIdx.x * kBlock + threadIdx.x) / 64;
  const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
  if (smp >= n_samples) return;
  const int* ix = idx + smp * M;
  const uint8_t* fl = flag + smp * M;
  int j[NQ];
  double2 v[NQ];
#pragma unroll
  for (int u = 0; u < NQ; ++u) j[u] = ix[g + 2 * u];
#pragma unroll
  for (int u = 0; u < NQ; ++u) v[u] = tab[(size_t)j[u] * LPR + l];
  double2 a = {0.0, 0.0};
#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    a.x += v[u].x;
    a.y += v[u].y;
  }
  a.x += __shfl_xor(a.x, 32, 64);
  a.y += __shfl_xor(a.y, 32, 64);
#pragma unroll
  for (int u = 0; u < NQ; ++u)
    if (fl[g + 2 * u]) {
      v[u].x = v[u].x * 0.999 + 1e-9 * a.x;
      v[u].y = v[u].y * 0.999 + 1e-9 * a.y;
      tab[(size_t)j[u] * LPR + l] = v[u];
    }
}

template <int U, int CAP>
__global__ __launch_bounds__(kBlock) void k_sample_lds(double2* __restrict__ tab, const int* __restrict__ idx,
                                                       const uint8_t* __restrict__ flag, size_t n_samples, int M, double* out) {
  constexpr int LPR = 32;
  __shared__ double2 rows[kBlock / 32][CAP][LPR];  // [group of the block][slot][lane]
  __shared__ int jrow[kBlock / 32][CAP];
  const size_t smp = ((size_t)blockIdx.x * kBlock + threadIdx.x) / 64;
  const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR, gb = threadIdx.x / LPR;
  if (smp >= n_samples) return;
  const int* ix = idx + smp * M;
  const uint8_t* fl = flag + smp * M;
  double2 a = {0.0, 0.0};
  int cnt = 0;  // flagged rows seen by this group so far (uniform over the group's lanes)
  for (int q = g; q < M; q += 2 * U) {
    int j[U];
    bool f[U];
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      j[u] = ix[q + 2 * u];
      f[u] = fl[q + 2 * u] != 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = tab[(size_t)j[u] * LPR + l];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a.x += v[u].x;
      a.y += v[u].y;
      if (f[u]) {
        if (cnt < CAP) {
          rows[gb][cnt][l] = v[u];
          if (l == 0) jrow[gb][cnt] = j[u];
        }
        ++cnt;
      }
    }
  }
  a.x += __shfl_xor(a.x, 32, 64);
  a.y += __shfl_xor(a.y, 32, 64);
  const int staged = cnt < CAP ? cnt : CAP;
  for (int s = 0; s < staged; ++s) {
    double2 v = rows[gb][s][l];
    const int j = jrow[gb][s];
    v.x = v.x * 0.999 + 1e-9 * a.x;
    v.y = v.y * 0.999 + 1e-9 * a.y;
    tab[(size_t)j * LPR + l] = v;
  }
  if (cnt > CAP) {  // overflow: second visit of the remaining flagged rows
    int seen = 0;
    for (int q = g; q < M; q += 2) {
      if (!fl[q]) continue;
      if (seen++ < CAP) continue;
      const int j = ix[q];
      double2 v = tab[(size_t)j * LPR + l];
      v.x = v.x * 0.999 + 1e-9 * a.x;
      v.y = v.y * 0.999 + 1e-9 * a.y;
      tab[(size_t)j * LPR + l] = v;
    }
  }
}

template <class F>
static double time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int LPR>
static void run_rows(double2* tab, size_t tab_rows, size_t n_rows, double* out) {
  // distinct random rows (a random sample of a permutation), so RMW has no conflicts
  std::vector<int> perm(tab_rows);
  std::iota(perm.begin(), perm.end(), 0);
  std::mt19937_64 rng(7);
  std::shuffle(perm.begin(), perm.end(), rng);
  n_rows = std::min(n_rows, tab_rows);
  std::vector<uint8_t> fl(n_rows);
  for (size_t i = 0; i < n_rows; ++i) fl[i] = (rng() % 100) < 60;
  int* idx;
  uint8_t* flag;
  CK(hipMalloc(&idx, n_rows * sizeof(int)));
  CK(hipMalloc(&flag, n_rows));
  CK(hipMemcpy(idx, perm.data(), n_rows * sizeof(int), hipMemcpyHostToDevice));
  CK(hipMemcpy(flag, fl.data(), n_rows, hipMemcpyHostToDevice));
  const double row_b = LPR * 16.0;
  const size_t groups_per_block = kBlock / LPR;
  auto grid_for = [&](size_t rows_per_group) {
    size_t gneed = (n_rows + rows_per_group - 1) / rows_per_group;
    return (unsigned)std::min<size_t>((gneed + groups_per_block - 1) / groups_per_block, 1u << 20);
  };
  for (int U : {4, 8}) {
    const unsigned grid = grid_for(U);  // one round of U rows per lane group
    double ms;
    if (U == 4) ms = time_ms([&] { hipLaunchKernelGGL((k_rows<LPR, 4, 0>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, n_rows, out); }, 20);
    else ms = time_ms([&] { hipLaunchKernelGGL((k_rows<LPR, 8, 0>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, n_rows, out); }, 20);
    printf("  gather   rows=%zu U=%d (1 round/group)      %8.1f us  %7.1f GB/s\n", n_rows, U, ms * 1e3, n_rows * row_b / ms / 1e6);
    if (U == 4) ms = time_ms([&] { hipLaunchKernelGGL((k_rows<LPR, 4, 1>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, n_rows, out); }, 20);
    else ms = time_ms([&] { hipLaunchKernelGGL((k_rows<LPR, 8, 1>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, n_rows, out); }, 20);
    printf("  rmw      rows=%zu U=%d (1 round/group)      %8.1f us  %7.1f GB/s (r+w)\n", n_rows, U, ms * 1e3, 2 * n_rows * row_b / ms / 1e6);
  }
  {
    // persistent-ish: 256 CUs x 8 blocks, each group walks many rounds
    const unsigned grid = 256 * 8;
    double ms = time_ms([&] { hipLaunchKernelGGL((k_rows<LPR, 8, 0>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, n_rows, out); }, 20);
    printf("  gather   rows=%zu U=8 grid=2048 (many rounds) %8.1f us  %7.1f GB/s\n", n_rows, ms * 1e3, n_rows * row_b / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_rows<LPR, 8, 1>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, n_rows, out); }, 20);
    printf("  rmw      rows=%zu U=8 grid=2048 (many rounds) %8.1f us  %7.1f GB/s (r+w)\n", n_rows, ms * 1e3, 2 * n_rows * row_b / ms / 1e6);
  }
  for (int M : {32, 64}) {
    const size_t ns = n_rows / M;
    const unsigned grid = (unsigned)((ns + groups_per_block - 1) / groups_per_block);
    double ms = time_ms([&] { hipLaunchKernelGGL((k_sample<LPR, 8>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, ns, M, out); }, 20);
    const double bytes = ns * M * row_b * (1.0 + 2 * 0.6);
    printf("  sample   M=%d samples=%zu gather + 60%% rmw  %8.1f us  %7.1f GB/s (alg. r + 0.6(r+w)), %7.1f GB/s without the re-read\n",
           M, ns, ms * 1e3, bytes / ms / 1e6, ns * M * row_b * 1.6 / ms / 1e6);
  }

  if (LPR == 32) {
    const int M = 64;
    const size_t ns = n_rows / M;
    const unsigned grid = (unsigned)((ns + 3) / 4);
    const double alg = ns * M * row_b * 1.6;
    double ms = time_ms([&] { hipLaunchKernelGGL((k_sample_reg<64>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, ns, out); }, 20);
    printf("  sample_reg       M=64 samples=%zu  %8.1f us  %7.1f GB/s (r + 0.6 w)\n", ns, ms * 1e3, alg / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_sample_lds<8, 19>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, ns, M, out); }, 20);
    printf("  sample_lds U8 C19 M=64 samples=%zu  %8.1f us  %7.1f GB/s (r + 0.6 w)\n", ns, ms * 1e3, alg / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_sample_lds<16, 19>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, ns, M, out); }, 20);
    printf("  sample_lds U16 C19 M=64 samples=%zu %8.1f us  %7.1f GB/s (r + 0.6 w)\n", ns, ms * 1e3, alg / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_sample_lds<8, 12>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, ns, M, out); }, 20);
    printf("  sample_lds U8 C12 M=64 samples=%zu  %8.1f us  %7.1f GB/s (r + 0.6 w)\n", ns, ms * 1e3, alg / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_sample_lds<8, 32>), dim3(grid), dim3(kBlock), 0, 0, tab, idx, flag, ns, M, out); }, 20);
    printf("  sample_lds U8 C32 M=64 samples=%zu  %8.1f us  %7.1f GB/s (r + 0.6 w)\n", ns, ms * 1e3, alg / ms / 1e6);
  }
  CK(hipFree(idx));
  CK(hipFree(flag));
}

int main(int argc, char** argv) {
  const size_t tab_mb = argc > 1 ? atoll(argv[1]) : 512;
  const int row_bytes = argc > 2 ? atoi(argv[2]) : 512;
  const size_t n_rows = argc > 3 ? atoll(argv[3]) : 524288;
  const size_t tab_bytes = tab_mb << 20;
  double2 *tab, *dst;
  double* out;
  CK(hipMalloc(&tab, tab_bytes));
  CK(hipMalloc(&dst, tab_bytes));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(tab, 0, tab_bytes));
  CK(hipMemset(dst, 0, tab_bytes));
  const size_t n2 = tab_bytes / 16;
  printf("table %zu MB, rows of %d B, %zu rows per launch\n", tab_mb, row_bytes, n_rows);
  double ms = time_ms([&] { hipLaunchKernelGGL(k_stream_read, dim3(256 * 16), dim3(kBlock), 0, 0, tab, n2, out); }, 10);
  printf("  stream read  %8.1f us  %7.1f GB/s\n", ms * 1e3, tab_bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL(k_stream_copy, dim3(256 * 16), dim3(kBlock), 0, 0, tab, dst, n2); }, 10);
  printf("  stream copy  %8.1f us  %7.1f GB/s (r+w)\n", ms * 1e3, 2.0 * tab_bytes / ms / 1e6);
  ms = time_ms([&] { CK(hipMemcpyAsync(dst, tab, tab_bytes, hipMemcpyDeviceToDevice, 0)); }, 10);
  printf("  hipMemcpy D2D %7.1f us  %7.1f GB/s (r+w)\n", ms * 1e3, 2.0 * tab_bytes / ms / 1e6);
  const size_t tab_rows = tab_bytes / row_bytes;
  switch (row_bytes) {
    case 128: run_rows<8>(tab, tab_rows, n_rows, out); break;
    case 256: run_rows<16>(tab, tab_rows, n_rows, out); break;
    case 512: run_rows<32>(tab, tab_rows, n_rows, out); break;
    case 1024: run_rows<64>(tab, tab_rows, n_rows, out); break;
    default: fprintf(stderr, "row_bytes must be 128/256/512/1024\n"); return 1;
  }
  CK(hipDeviceSynchronize());
  return 0;
}
