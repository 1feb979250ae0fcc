// tools/divtest.hip -- is  q2 = fma(r1, y, q1), r1 = fma(-b, q1, a), q1 = fma(r0, y, q0), r0 = fma(-b, q0, a), q0 = a*y  with
// y = 1/b (IEEE division) equal to a/b (IEEE division) for every pair?  (Markstein's two-step correction of a
// reciprocal-multiply quotient; used in seqwin.hip where one divisor -- the lazy L2 scale -- divides a whole sample's rows.)
// Counts mismatches over random pairs: b in (1e-9, 1] (what the scale is), b in [1, 2), b anywhere; a over many decades.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/divtest.hip -o tools/bin/divtest ; run: tools/bin/divtest [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
__device__ inline uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
__device__ inline double mant(uint64_t r) { return __longlong_as_double((long long)((r >> 12) | 0x3FF0000000000000ULL)); }  // [1, 2)
__global__ void k(uint64_t seed, int mode, unsigned long long* bad, double* ex) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long nb = 0;
  for (int it = 0; it < 256; ++it) {
    const uint64_t r0 = mix(seed + t * 256 + it), r1 = mix(r0 + 0x9E3779B97F4A7C15ULL), r2 = mix(r1 + 12345);
    double b = mant(r0), a = mant(r1);
    const int ea = (int)(r2 % 121) - 60;
    a = ldexp(a, ea);
    if (r2 & (1ull << 40)) a = -a;
    if (mode == 0) b = ldexp(b, -(int)((r2 >> 8) % 30) - 1);  // (2^-31, 1)
    else if (mode == 2) b = ldexp(b, (int)((r2 >> 8) % 200) - 100);
    else if (mode == 3) { b = 1.0 - ldexp(mant(r0) - 1.0, -(int)((r2 >> 8) % 40)); }  // just below 1: 1 - eta beta products
    const double y = 1.0 / b;
    const double q0 = a * y;
    const double e0 = fma(-b, q0, a);
    const double q1 = fma(e0, y, q0);
    const double e1 = fma(-b, q1, a);
    const double q2 = fma(e1, y, q1);
    const double qt = a / b;
    if (q2 != qt) {
      if (nb == 0 && atomicAdd(bad + 1, 1ull) == 0) { ex[0] = a; ex[1] = b; ex[2] = q2; ex[3] = qt; }
      ++nb;
    }
  }
  if (nb) atomicAdd(bad, nb);
}
int main(int argc, char** argv) {
  int rounds = argc > 1 ? atoi(argv[1]) : 8;
  unsigned long long* bad; double* ex;
  hipMalloc(&bad, 16); hipMalloc(&ex, 32);
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(bad, 0, 16);
    unsigned long long total = 0;
    for (int r = 0; r < rounds; ++r) {
      hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, (uint64_t)r * 0x1234567ULL + mode * 77, mode, bad, ex);
      total += 65536ull * 256 * 256;
    }
    unsigned long long h[2]; double hx[4];
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(hx, ex, 32, hipMemcpyDeviceToHost);
    printf("mode %d: %llu pairs, %llu mismatches", mode, total, h[0]);
    if (h[0]) printf("  e.g. a=%a b=%a got %a want %a", hx[0], hx[1], hx[2], hx[3]);
    printf("\n");
  }
  return 0;
}
