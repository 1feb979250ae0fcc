// nimfm_amd/csrc/ingest.hip -- svmlight / libffm text -> CSR resident in HBM, parsed ON the GPU.
//
// Replaces the reference's two-pass, line-by-line host loaders loadSVMLightFile (dataset.nim:562-632)
// and loadFFMFile (dataset.nim:696-790): "target idx:val idx:val ..." / "target field:idx:val ...",
// one sample per line, index base detected from the file (0-based iff an index 0 occurs, :589/:732;
// negative index = ValueError), nFeatures = maxIndex + 1 - offset (nFields likewise, :733-734, with the
// reference's initial maxFieldIndex = 1).
//
// The file goes to HBM as bytes (four reader threads, pinned buffers); everything after that is data-parallel:
//   1. count per 16-KiB tile, scan, then list the positions of '\n' and ':' (16-byte loads, exact byte masks)
//   2. one thread per LINE: its span, the colons inside it (binary search) -> row length and, because
//      entries appear in file order, indptr[i] = (#colons before the line) / colons-per-entry; the
//      target with parse_float (parse_num.h: correctly rounded, as Nim's parseFloat)
//   3. one thread per ENTRY: the integer token that ends at the colon (backward scan), the value that
//      starts after it; wave-reduced min/max of the indices -> offset
//   4. indices narrowed to int32 with the offset applied
// Tokens the GPU cannot round for sure (> 19 significant digits where the cut matters) are listed and
// re-read by the host with strtod.  Where the reference's tokenizer is well defined (single separators)
// the result is identical; malformed text (a token glued to the wrong neighbour, colons that do not
// pair up) is an error here, while the reference would read garbage.
#include <hipcub/hipcub.hpp>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ingest.h"
#include "parse_num.h"

namespace nfm {

using num::Pow5;

static const Pow5 kPow5Host[] = NFM_POW5_TABLE_INIT;
constexpr int kPow5N = NFM_POW5_QMAX - NFM_POW5_QMIN + 1;

enum {  // status words the kernels raise (device int64 counters)
  ST_MALFORMED = 0, ST_NO_TARGET = 1, ST_STRTOD = 2, ST_MIN_IDX = 3, ST_MAX_IDX = 4, ST_MIN_FLD = 5, ST_MAX_FLD = 6,
  ST_MAX_ROW = 7, ST_FIRST_BAD = 8, ST_COUNT = 16
};
constexpr int kMaxFix = 1 << 20;  // tokens the host may have to re-read

struct FixRec {
  int64_t pos;    // byte offset of the token
  int64_t slot;   // entry index (value) or line index (target)
  int32_t kind;   // 0 = value, 1 = target
  int32_t pad;
};

// ---- positions of '\n' and ':' -----------------------------------------------------------------
// The text is cut into tiles of kTile bytes, one workgroup per tile, 64 bytes per thread read as four
// 16-byte loads.  Pass 1 counts both characters per tile; after an exclusive scan of the tile counts
// pass 2 reads the tile again and writes the positions (ascending) at tile offset + in-tile rank.
// (hipcub::DeviceSelect over a counting iterator did the same at 42 GB/s: one byte load per item.)
constexpr int kChunk = 64;                  // bytes per thread
constexpr int kTile = kBlock * kChunk;      // bytes per workgroup (16 KiB)

// 0x80 in every byte of w that equals c (exact per byte)
__device__ __forceinline__ uint64_t eq_bytes(uint64_t w, uint64_t c8) {
  const uint64_t x = w ^ c8;
  const uint64_t lo7 = 0x7F7F7F7F7F7F7F7Full;
  return ~(((x & lo7) + lo7) | x | lo7);
}

// the thread's 64 bytes as 8 words (zero beyond len; the buffer is padded, see upload)
__device__ __forceinline__ void load_chunk64(const char* __restrict__ t, int64_t len, int64_t off, uint64_t (&w)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ulonglong2 v = {0ull, 0ull};
    if (off + 16 * q < len) v = *reinterpret_cast<const ulonglong2*>(t + off + 16 * q);
    w[2 * q] = v.x;
    w[2 * q + 1] = v.y;
  }
  // bytes past the end of the text must not match
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int64_t wo = off + 8 * q;
    if (wo + 8 > len) {
      const int keep = wo >= len ? 0 : (int)(len - wo);
      w[q] = keep == 0 ? 0ull : (w[q] & ((~0ull) >> (8 * (8 - keep))));
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_tile_counts(const char* __restrict__ t, int64_t len, int64_t n_tiles,
                                                        int64_t* __restrict__ cnt_nl, int64_t* __restrict__ cnt_co) {
  __shared__ int red[2][kWavesPerBlock];
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t off = tile * kTile + (int64_t)threadIdx.x * kChunk;
    uint64_t w[8];
    load_chunk64(t, len, off, w);
    int nl = 0, co = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      nl += __popcll(eq_bytes(w[q], 0x0A0A0A0A0A0A0A0Aull));
      co += __popcll(eq_bytes(w[q], 0x3A3A3A3A3A3A3A3Aull));
    }
    // '\0' padding never equals '\n' or ':', so no masking of the counts is needed
    for (int s = 1; s < kWave; s <<= 1) {
      nl += __shfl_xor(nl, s, kWave);
      co += __shfl_xor(co, s, kWave);
    }
    __syncthreads();
    if ((threadIdx.x & (kWave - 1)) == 0) {
      red[0][threadIdx.x >> 6] = nl;
      red[1][threadIdx.x >> 6] = co;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int a = 0, b = 0;
      for (int v = 0; v < kWavesPerBlock; ++v) {
        a += red[0][v];
        b += red[1][v];
      }
      cnt_nl[tile] = a;
      cnt_co[tile] = b;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_tile_positions(const char* __restrict__ t, int64_t len, int64_t n_tiles,
                                                           const int64_t* __restrict__ off_nl, const int64_t* __restrict__ off_co,
                                                           int64_t* __restrict__ pos_nl, int64_t* __restrict__ pos_co) {
  __shared__ int scan[2][kBlock];
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t off = tile * kTile + (int64_t)threadIdx.x * kChunk;
    uint64_t w[8];
    load_chunk64(t, len, off, w);
    uint64_t mn[8], mc[8];
    int nl = 0, co = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      mn[q] = eq_bytes(w[q], 0x0A0A0A0A0A0A0A0Aull);
      mc[q] = eq_bytes(w[q], 0x3A3A3A3A3A3A3A3Aull);
      nl += __popcll(mn[q]);
      co += __popcll(mc[q]);
    }
    __syncthreads();
    scan[0][threadIdx.x] = nl;
    scan[1][threadIdx.x] = co;
    __syncthreads();
    for (int s = 1; s < kBlock; s <<= 1) {  // inclusive scan over the workgroup's threads
      int a = 0, b = 0;
      if ((int)threadIdx.x >= s) {
        a = scan[0][threadIdx.x - s];
        b = scan[1][threadIdx.x - s];
      }
      __syncthreads();
      scan[0][threadIdx.x] += a;
      scan[1][threadIdx.x] += b;
      __syncthreads();
    }
    int64_t on = off_nl[tile] + scan[0][threadIdx.x] - nl, oc = off_co[tile] + scan[1][threadIdx.x] - co;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      uint64_t m = mn[q];
      while (m) {
        const int b = (__ffsll((long long)m) - 1) >> 3;
        pos_nl[on++] = off + 8 * q + b;
        m &= m - 1;
      }
      m = mc[q];
      while (m) {
        const int b = (__ffsll((long long)m) - 1) >> 3;
        pos_co[oc++] = off + 8 * q + b;
        m &= m - 1;
      }
    }
  }
}

__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* __restrict__ a, int64_t n, int64_t v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ bool is_sep(char c) { return c == ' ' || c == '\t'; }

// what may follow a target or a value at position p: the end of the line (after at most one
// separator, as the reference's "skip one character" allows), or one separator and the next
// entry's integer token, which has to end in ':'
__device__ __forceinline__ bool good_tail(const char* __restrict__ t, int64_t len, int64_t p) {
  auto eol = [&](int64_t q) { return q >= len || t[q] == '\n' || t[q] == '\r'; };
  if (eol(p)) return true;
  if (!is_sep(t[p])) return false;
  int64_t q = p + 1;
  if (eol(q)) return true;
  const int64_t q0 = q;
  while (q < len && q - q0 < 24 && (num::is_digit(t[q]) || t[q] == '-' || t[q] == '+')) ++q;
  return q > q0 && q < len && t[q] == ':';
}

__device__ __forceinline__ void raise_bad(long long* st, int64_t pos) {
  atomicAdd((unsigned long long*)&st[ST_MALFORMED], 1ull);
  atomicMin(&st[ST_FIRST_BAD], (long long)pos);
}

__device__ __forceinline__ void push_fix(long long* st, FixRec* fix, int64_t pos, int64_t slot, int kind) {
  const unsigned long long k = atomicAdd((unsigned long long*)&st[ST_STRTOD], 1ull);
  if (k < (unsigned long long)kMaxFix) fix[k] = FixRec{pos, slot, kind, 0};
}

// one thread per line
__global__ void k_lines(const char* __restrict__ t, int64_t len, const int64_t* __restrict__ nl, int64_t n_nl, int64_t n_lines,
                        const int64_t* __restrict__ co, int64_t n_co, int cpe, const Pow5* __restrict__ table,
                        int64_t* __restrict__ indptr, double* __restrict__ y, uint8_t* __restrict__ y_missing,
                        long long* __restrict__ st, FixRec* __restrict__ fix) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_lines; i += (int64_t)gridDim.x * blockDim.x) {
    if (i == n_lines) {
      indptr[i] = n_co / cpe;
      continue;
    }
    const int64_t s = i ? nl[i - 1] + 1 : 0;
    int64_t e = i < n_nl ? nl[i] : len;
    if (e > s && t[e - 1] == '\r') --e;
    const int64_t c0 = lower_bound_i64(co, n_co, s), c1 = lower_bound_i64(co, n_co, e);
    if ((c0 % cpe) || ((c1 - c0) % cpe)) raise_bad(st, s);
    indptr[i] = c0 / cpe;
    atomicMax(&st[ST_MAX_ROW], (long long)((c1 - c0) / cpe));
    double v = 0.0;
    int fl = 0;
    const int used = num::parse_float(t + s, e - s, &v, &fl, table);
    y_missing[i] = used == 0;
    if (used == 0) {
      atomicAdd((unsigned long long*)&st[ST_NO_TARGET], 1ull);
      if (e > s) raise_bad(st, s);  // a non-empty line has to start with its target
    } else if (!good_tail(t, e, s + used)) {
      raise_bad(st, s + used);
    }
    if (fl & num::kNeedsStrtod) push_fix(st, fix, s, i, 1);
    y[i] = v;
  }
}

// one thread per entry
__global__ void k_entries(const char* __restrict__ t, int64_t len, const int64_t* __restrict__ co, int64_t n_ent, int cpe,
                          const Pow5* __restrict__ table, int64_t* __restrict__ idx_raw, int64_t* __restrict__ fld_raw,
                          double* __restrict__ data, long long* __restrict__ st, FixRec* __restrict__ fix) {
  long long mn = INT64_MAX, mx = INT64_MIN, fmn = INT64_MAX, fmx = INT64_MIN;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_ent; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p0 = co[e * cpe], p1 = co[e * cpe + cpe - 1];
    // the integer token that ends right before the first colon
    int64_t ts = p0;
    while (ts > 0 && p0 - ts < 24 && (num::is_digit(t[ts - 1]) || t[ts - 1] == '-' || t[ts - 1] == '+')) --ts;
    int64_t a = 0, b = 0;
    bool ok = ts < p0 && ts > 0 && is_sep(t[ts - 1]) && num::parse_int(t + ts, p0 - ts, &a) == p0 - ts;
    if (cpe == 2) {
      ok = ok && p1 > p0 + 1 && num::parse_int(t + p0 + 1, p1 - p0 - 1, &b) == p1 - p0 - 1;
      fld_raw[e] = a;
      idx_raw[e] = b;
      fmn = a < fmn ? a : fmn;
      fmx = a > fmx ? a : fmx;
    } else {
      idx_raw[e] = a;
      b = a;
    }
    mn = b < mn ? b : mn;
    mx = b > mx ? b : mx;
    double v = 0.0;
    int fl = 0;
    const int used = num::parse_float(t + p1 + 1, len - (p1 + 1), &v, &fl, table);
    const int64_t ve = p1 + 1 + used;
    ok = ok && used > 0 && good_tail(t, len, ve);
    if (!ok) raise_bad(st, p0);
    if (fl & num::kNeedsStrtod) push_fix(st, fix, p1 + 1, e, 0);
    data[e] = v;
  }
  for (int s = 1; s < kWave; s <<= 1) {
    const long long a = __shfl_xor(mn, s, kWave), b = __shfl_xor(mx, s, kWave);
    const long long c = __shfl_xor(fmn, s, kWave), d = __shfl_xor(fmx, s, kWave);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
    fmn = c < fmn ? c : fmn;
    fmx = d > fmx ? d : fmx;
  }
  if ((threadIdx.x & (kWave - 1)) == 0) {
    if (mn != INT64_MAX) {
      atomicMin(&st[ST_MIN_IDX], mn);
      atomicMax(&st[ST_MAX_IDX], mx);
    }
    if (fmn != INT64_MAX) {
      atomicMin(&st[ST_MIN_FLD], fmn);
      atomicMax(&st[ST_MAX_FLD], fmx);
    }
  }
}

__global__ void k_narrow(int64_t n, const int64_t* __restrict__ raw, int64_t off, int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (int32_t)(raw[i] - off);
}

static inline unsigned grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 256 * 64) b = 256 * 64;
  return (unsigned)b;
}

// file -> device bytes: kReaders host threads, each pread()s its share of 32-MiB chunks into its own two
// pinned buffers and queues the copy on its own stream, so reading the page cache (one core moves ~9 GB/s)
// and the PCIe copies of several chunks overlap.  The pinned buffers and streams are made once per process.
constexpr int kReaders = 4;
constexpr size_t kUpChunk = (size_t)32 << 20;
struct UploadSlots {
  void* pin[kReaders][2] = {};
  hipEvent_t ev[kReaders][2] = {};
  hipStream_t st[kReaders] = {};
  int device = -1;
  bool ok = false;
};
static UploadSlots& upload_slots(int device) {
  static UploadSlots* S = new UploadSlots();  // lives until process exit
  if (!S->ok || S->device != device) {
    if (S->ok) {  // another device: start over
      for (int r = 0; r < kReaders; ++r) {
        for (int b = 0; b < 2; ++b) {
          (void)hipHostFree(S->pin[r][b]);
          (void)hipEventDestroy(S->ev[r][b]);
        }
        (void)hipStreamDestroy(S->st[r]);
      }
      *S = UploadSlots();
    }
    bool good = true;
    for (int r = 0; r < kReaders && good; ++r) {
      good = hipStreamCreateWithFlags(&S->st[r], hipStreamNonBlocking) == hipSuccess;
      for (int b = 0; b < 2 && good; ++b)
        good = hipHostMalloc(&S->pin[r][b], kUpChunk, hipHostMallocDefault) == hipSuccess &&
               hipEventCreateWithFlags(&S->ev[r][b], hipEventDisableTiming) == hipSuccess;
    }
    S->ok = good;
    S->device = device;
  }
  return *S;
}

static int upload_file(nfm_ctx* ctx, const char* path, DevBuf* text, int64_t* len_out) {
  const int fd = open(path, O_RDONLY);
  NFM_CHECK(fd >= 0, NFM_ERR_INVALID, "%s cannot be read.", path);
  struct stat sb;
  if (fstat(fd, &sb) != 0) {
    close(fd);
    return set_error(NFM_ERR_INVALID, "%s cannot be read.", path);
  }
  const int64_t len = (int64_t)sb.st_size;
  int rc = text->alloc((size_t)len + 64);
  if (rc != NFM_OK) { close(fd); return rc; }
  UploadSlots& S = upload_slots(ctx->device);
  if (!S.ok) { close(fd); return set_error(NFM_ERR_HIP, "ingest upload: pinned staging buffers unavailable"); }
  NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the text buffer may be a recycled block
  const int64_t n_chunks = (len + (int64_t)kUpChunk - 1) / (int64_t)kUpChunk;
  std::atomic<int> failed{0};
  char* dst = text->as<char>();
  const int device = ctx->device;
  auto worker = [&](int r) {
    if (hipSetDevice(device) != hipSuccess) { failed = 1; return; }
    bool used[2] = {false, false};
    int which = 0;
    for (int64_t c = r; c < n_chunks && !failed; c += kReaders) {
      const int64_t off = c * (int64_t)kUpChunk;
      const size_t want = (size_t)std::min<int64_t>((int64_t)kUpChunk, len - off);
      if (used[which] && hipEventSynchronize(S.ev[r][which]) != hipSuccess) { failed = 1; return; }
      size_t got = 0;
      while (got < want) {
        const ssize_t k = pread(fd, (char*)S.pin[r][which] + got, want - got, (off_t)(off + (int64_t)got));
        if (k <= 0) { failed = 2; return; }
        got += (size_t)k;
      }
      if (hipMemcpyAsync(dst + off, S.pin[r][which], want, hipMemcpyHostToDevice, S.st[r]) != hipSuccess ||
          hipEventRecord(S.ev[r][which], S.st[r]) != hipSuccess) { failed = 1; return; }
      used[which] = true;
      which ^= 1;
    }
    if (hipStreamSynchronize(S.st[r]) != hipSuccess) failed = 1;
  };
  std::vector<std::thread> th;
  for (int r = 1; r < kReaders; ++r) th.emplace_back(worker, r);
  worker(0);
  for (auto& t_ : th) t_.join();
  close(fd);
  if (failed == 2) return set_error(NFM_ERR_INVALID, "%s: short read", path);
  if (failed) return set_error(NFM_ERR_HIP, "ingest upload failed: %s", hipGetErrorString(hipGetLastError()));
  *len_out = len;
  return NFM_OK;
}

int ingest_text(nfm_ctx* ctx, const char* path, const char* mem, int64_t mem_len, bool with_fields, IngestResult* out) {
  hipStream_t st = ctx->stream;
  hipEvent_t e0, e1, e2;
  NFM_HIP_CHECK(hipEventCreate(&e0));
  NFM_HIP_CHECK(hipEventCreate(&e1));
  NFM_HIP_CHECK(hipEventCreate(&e2));
  struct EvGuard {
    hipEvent_t a, b, c;
    ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(c); }
  } guard{e0, e1, e2};
  NFM_HIP_CHECK(hipEventRecord(e0, st));
  DevBuf text;
  int64_t len = 0;
  if (path) {
    NFM_TRY(upload_file(ctx, path, &text, &len));
  } else {
    NFM_CHECK(mem || mem_len == 0, NFM_ERR_INVALID, "null text");
    len = mem_len;
    NFM_TRY(text.alloc((size_t)len + 64));
    if (len) NFM_HIP_CHECK(hipMemcpyAsync(text.p, mem, (size_t)len, hipMemcpyHostToDevice, st));
  }
  NFM_HIP_CHECK(hipEventRecord(e1, st));
  const char* t = text.as<char>();
  const int cpe = with_fields ? 2 : 1;
  // 1. counts and positions
  DevBuf table, status, fix, cnt_nl, cnt_co, off_nl, off_co, scan_tmp;
  const int64_t n_tiles = (len + kTile - 1) / kTile;
  NFM_TRY(cnt_nl.alloc(sizeof(int64_t) * (n_tiles + 1)));
  NFM_TRY(cnt_co.alloc(sizeof(int64_t) * (n_tiles + 1)));
  NFM_TRY(off_nl.alloc(sizeof(int64_t) * (n_tiles + 1)));
  NFM_TRY(off_co.alloc(sizeof(int64_t) * (n_tiles + 1)));
  NFM_HIP_CHECK(hipMemsetAsync(cnt_nl.p, 0, sizeof(int64_t) * (n_tiles + 1), st));
  NFM_HIP_CHECK(hipMemsetAsync(cnt_co.p, 0, sizeof(int64_t) * (n_tiles + 1), st));
  const unsigned tile_grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_tiles, 256 * 32));
  if (n_tiles)
    hipLaunchKernelGGL(k_tile_counts, dim3(tile_grid), dim3(kBlock), 0, st, t, len, n_tiles, cnt_nl.as<int64_t>(),
                       cnt_co.as<int64_t>());
  {
    size_t bytes = 0;
    NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, cnt_nl.as<int64_t>(), off_nl.as<int64_t>(), (int)(n_tiles + 1), st));
    NFM_TRY(scan_tmp.alloc(bytes));
    NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, bytes, cnt_nl.as<int64_t>(), off_nl.as<int64_t>(), (int)(n_tiles + 1), st));
    NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, bytes, cnt_co.as<int64_t>(), off_co.as<int64_t>(), (int)(n_tiles + 1), st));
  }
  int64_t h_cnt[2] = {0, 0};
  NFM_HIP_CHECK(hipMemcpyAsync(&h_cnt[0], off_nl.as<int64_t>() + n_tiles, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipMemcpyAsync(&h_cnt[1], off_co.as<int64_t>() + n_tiles, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  char last = '\n';
  if (len) NFM_HIP_CHECK(hipMemcpyAsync(&last, t + len - 1, 1, hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  const int64_t n_nl = h_cnt[0], n_co = h_cnt[1];
  const int64_t n_lines = n_nl + ((len > 0 && last != '\n') ? 1 : 0);  // Nim's `lines`: no empty line after a final "\n"
  NFM_CHECK(n_co % cpe == 0, NFM_ERR_INVALID, "malformed file: %lld ':' do not form %s entries", (long long)n_co,
            with_fields ? "field:index:value" : "index:value");
  const int64_t n_ent = n_co / cpe;
  DevBuf nl, co;
  NFM_TRY(nl.alloc(sizeof(int64_t) * std::max<int64_t>(n_nl, 1)));
  NFM_TRY(co.alloc(sizeof(int64_t) * std::max<int64_t>(n_co, 1)));
  if (n_tiles)
    hipLaunchKernelGGL(k_tile_positions, dim3(tile_grid), dim3(kBlock), 0, st, t, len, n_tiles, off_nl.as<int64_t>(),
                       off_co.as<int64_t>(), nl.as<int64_t>(), co.as<int64_t>());
  NFM_HIP_CHECK(hipGetLastError());
  // 2. lines, 3. entries
  NFM_TRY(table.alloc(sizeof(Pow5) * kPow5N));
  NFM_HIP_CHECK(hipMemcpyAsync(table.p, kPow5Host, sizeof(Pow5) * kPow5N, hipMemcpyHostToDevice, st));
  long long h_st[ST_COUNT];
  for (int i = 0; i < ST_COUNT; ++i) h_st[i] = 0;
  h_st[ST_MIN_IDX] = h_st[ST_MIN_FLD] = h_st[ST_FIRST_BAD] = INT64_MAX;
  h_st[ST_MAX_IDX] = h_st[ST_MAX_FLD] = INT64_MIN;
  NFM_TRY(status.alloc(sizeof(h_st)));
  NFM_HIP_CHECK(hipMemcpyAsync(status.p, h_st, sizeof(h_st), hipMemcpyHostToDevice, st));
  NFM_TRY(fix.alloc(sizeof(FixRec) * kMaxFix));
  DevBuf y_missing, idx_raw, fld_raw;
  NFM_TRY(out->indptr.alloc(sizeof(int64_t) * (n_lines + 1)));
  NFM_TRY(out->y.alloc(sizeof(double) * std::max<int64_t>(n_lines, 1)));
  NFM_TRY(y_missing.alloc((size_t)std::max<int64_t>(n_lines, 1)));
  NFM_TRY(out->data.alloc(sizeof(double) * std::max<int64_t>(n_ent, 1)));
  NFM_TRY(idx_raw.alloc(sizeof(int64_t) * std::max<int64_t>(n_ent, 1)));
  if (with_fields) NFM_TRY(fld_raw.alloc(sizeof(int64_t) * std::max<int64_t>(n_ent, 1)));
  hipLaunchKernelGGL(k_lines, dim3(grid_for(n_lines + 1)), dim3(kBlock), 0, st, t, len, nl.as<int64_t>(), n_nl, n_lines,
                     co.as<int64_t>(), n_co, cpe, table.as<Pow5>(), out->indptr.as<int64_t>(), out->y.as<double>(),
                     y_missing.as<uint8_t>(), status.as<long long>(), fix.as<FixRec>());
  if (n_ent)
    hipLaunchKernelGGL(k_entries, dim3(grid_for(n_ent)), dim3(kBlock), 0, st, t, len, co.as<int64_t>(), n_ent, cpe,
                       table.as<Pow5>(), idx_raw.as<int64_t>(), fld_raw.as<int64_t>(), out->data.as<double>(),
                       status.as<long long>(), fix.as<FixRec>());
  NFM_HIP_CHECK(hipGetLastError());
  NFM_HIP_CHECK(hipMemcpyAsync(h_st, status.p, sizeof(h_st), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  NFM_CHECK(h_st[ST_MALFORMED] == 0, NFM_ERR_INVALID, "malformed %s text: %lld bad tokens, first near byte %lld",
            with_fields ? "libffm" : "svmlight", h_st[ST_MALFORMED], h_st[ST_FIRST_BAD]);
  // offsets and shapes (dataset.nim:586-590, 729-734)
  long long min_idx = 1, max_idx = 0, min_fld = 1, max_fld = 1;  // the reference's initial values
  if (n_ent) {
    min_idx = std::min<long long>(min_idx, h_st[ST_MIN_IDX]);
    max_idx = std::max<long long>(max_idx, h_st[ST_MAX_IDX]);
    if (with_fields) {
      min_fld = std::min<long long>(min_fld, h_st[ST_MIN_FLD]);
      max_fld = std::max<long long>(max_fld, h_st[ST_MAX_FLD]);
    }
  }
  NFM_CHECK(min_idx >= 0, NFM_ERR_INVALID, "Negative index is included.");
  NFM_CHECK(!with_fields || min_fld >= 0, NFM_ERR_INVALID, "Negative field index is included.");
  out->min_index = min_idx;
  out->max_index = max_idx;
  out->offset = min_idx == 0 ? 0 : 1;
  out->d = max_idx + 1 - out->offset;
  out->offset_field = min_fld == 0 ? 0 : 1;
  out->n_fields = with_fields ? max_fld + 1 - out->offset_field : 0;
  NFM_CHECK(out->d < (int64_t)2147483647 - 64, NFM_ERR_UNSUPPORTED, "feature index %lld does not fit int32", max_idx);
  out->n = n_lines;
  out->nnz = n_ent;
  out->max_row = (int)h_st[ST_MAX_ROW];
  // 4. narrow with the offsets applied
  NFM_TRY(out->indices.alloc(sizeof(int32_t) * std::max<int64_t>(n_ent, 1)));
  if (n_ent) hipLaunchKernelGGL(k_narrow, dim3(grid_for(n_ent)), dim3(kBlock), 0, st, n_ent, idx_raw.as<int64_t>(), out->offset,
                                out->indices.as<int32_t>());
  if (with_fields) {
    NFM_TRY(out->fields.alloc(sizeof(int32_t) * std::max<int64_t>(n_ent, 1)));
    if (n_ent) hipLaunchKernelGGL(k_narrow, dim3(grid_for(n_ent)), dim3(kBlock), 0, st, n_ent, fld_raw.as<int64_t>(),
                                  out->offset_field, out->fields.as<int32_t>());
  }
  // tokens for strtod (rare): re-read from the device text
  if (h_st[ST_STRTOD] > 0) {
    NFM_CHECK(h_st[ST_STRTOD] <= kMaxFix, NFM_ERR_UNSUPPORTED, "%lld tokens with more than 19 significant digits", h_st[ST_STRTOD]);
    std::vector<FixRec> fx((size_t)h_st[ST_STRTOD]);
    NFM_HIP_CHECK(hipMemcpyAsync(fx.data(), fix.p, sizeof(FixRec) * fx.size(), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
    char buf[512];
    for (const FixRec& r : fx) {
      const size_t take = (size_t)std::min<int64_t>((int64_t)sizeof(buf) - 1, len - r.pos);
      NFM_HIP_CHECK(hipMemcpy(buf, t + r.pos, take, hipMemcpyDeviceToHost));
      buf[take] = 0;
      for (size_t k = 0; k < take; ++k)
        if (buf[k] == ' ' || buf[k] == '\t' || buf[k] == '\n' || buf[k] == '\r' || buf[k] == ':') { buf[k] = 0; break; }
      const double v = strtod(buf, nullptr);
      double* dst = r.kind == 0 ? out->data.as<double>() + r.slot : out->y.as<double>() + r.slot;
      NFM_HIP_CHECK(hipMemcpy(dst, &v, sizeof(double), hipMemcpyHostToDevice));
    }
  }
  // empty lines: parseFloat reads nothing and `target` keeps the previous line's value (dataset.nim:574,602)
  if (h_st[ST_NO_TARGET] > 0) {
    std::vector<double> yh((size_t)n_lines);
    std::vector<uint8_t> miss((size_t)n_lines);
    NFM_HIP_CHECK(hipMemcpyAsync(yh.data(), out->y.p, sizeof(double) * n_lines, hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipMemcpyAsync(miss.data(), y_missing.p, (size_t)n_lines, hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
    double prev = 0.0;
    for (int64_t i = 0; i < n_lines; ++i) {
      if (miss[i]) yh[i] = prev;
      prev = yh[i];
    }
    NFM_HIP_CHECK(hipMemcpyAsync(out->y.p, yh.data(), sizeof(double) * n_lines, hipMemcpyHostToDevice, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
  }
  NFM_HIP_CHECK(hipEventRecord(e2, st));
  NFM_HIP_CHECK(hipEventSynchronize(e2));
  float ms_up = 0, ms_parse = 0;
  NFM_HIP_CHECK(hipEventElapsedTime(&ms_up, e0, e1));
  NFM_HIP_CHECK(hipEventElapsedTime(&ms_parse, e1, e2));
  out->bytes = len;
  out->upload_ms = ms_up;
  out->parse_ms = ms_parse;
  return NFM_OK;
}


// ------------------------------------------------------------------------------------------------
// STREAMCSR binary files (the reference's out-of-core format, tensor/sparse_stream.nim:3-33):
//   magic "STREAMCSR" (9 bytes) | header {nRows, nCols, nnz: int64; max, min: float64} |
//   per row: nnz: int64, then nnz x {val: float64, id: int64}
//   magic "STREAMCSRFIELD" (14 bytes) | header {nRows, nCols, nnz, nFields: int64; max, min} |
//   per row: nnz: int64, then nnz x {field: int64, val: float64, id: int64}
// The reference streams row blocks through a host cache (readCache, sgd_multi.nim:83-97); with 288 GB of
// HBM the whole matrix is made resident instead.  Row starts depend on all earlier row lengths, so the
// host walks the row headers once (8 bytes per row) for indptr; the (value, id) pairs are split on the GPU.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t load_i64_unaligned(const unsigned char* p) {
  uint64_t v = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) v |= (uint64_t)p[b] << (8 * b);
  return (int64_t)v;
}

__global__ void k_stream_split(const unsigned char* __restrict__ raw, int64_t base, int esize, int with_fields, int64_t n_rows,
                               int64_t n_cols, int64_t n_fields,
                               const int64_t* __restrict__ indptr, int64_t nnz, int32_t* __restrict__ indices,
                               double* __restrict__ data, int32_t* __restrict__ fields, long long* __restrict__ st) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n_rows;  // row of entry e: last r with indptr[r] <= e
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (indptr[mid] <= e) lo = mid; else hi = mid - 1;
    }
    const unsigned char* p = raw + base + 8 * (lo + 1) + (int64_t)esize * e;
    int64_t f = 0;
    if (with_fields) {
      f = load_i64_unaligned(p);
      p += 8;
    }
    const int64_t vb = load_i64_unaligned(p), id = load_i64_unaligned(p + 8);
    data[e] = __longlong_as_double(vb);
    indices[e] = (int32_t)id;
    if (with_fields) fields[e] = (int32_t)f;
    // an id / field beyond the header's nCols / nFields would index the parameter tables out of bounds later
    if (id < 0 || id >= n_cols || f < 0 || (with_fields && f >= n_fields)) atomicAdd((unsigned long long*)&st[ST_MALFORMED], 1ull);
  }
}

static int read_whole(const char* path, std::vector<unsigned char>* buf) {
  FILE* f = fopen(path, "rb");
  NFM_CHECK(f, NFM_ERR_INVALID, "%s cannot be opened.", path);
  fseeko(f, 0, SEEK_END);
  const int64_t len = (int64_t)ftello(f);
  fseeko(f, 0, SEEK_SET);
  buf->resize((size_t)len);
  const size_t got = len ? fread(buf->data(), 1, (size_t)len, f) : 0;
  fclose(f);
  NFM_CHECK((int64_t)got == len, NFM_ERR_INVALID, "%s: short read", path);
  return NFM_OK;
}

int ingest_stream(nfm_ctx* ctx, const char* x_path, const char* y_path, IngestResult* out) {
  hipStream_t st = ctx->stream;
  // the bytes go to the device through the reader threads of upload_file; the host walks the row headers
  // through a read-only mapping of the same file
  DevBuf dev_raw;
  int64_t len = 0;
  NFM_TRY(upload_file(ctx, x_path, &dev_raw, &len));
  struct Mapping {
    const unsigned char* p = nullptr;
    size_t n = 0;
    ~Mapping() { if (p) munmap(const_cast<unsigned char*>(p), n); }
    const unsigned char* data() const { return p; }
  } raw;
  if (len > 0) {
    const int fd = open(x_path, O_RDONLY);
    NFM_CHECK(fd >= 0, NFM_ERR_INVALID, "%s cannot be opened.", x_path);
    void* mp = mmap(nullptr, (size_t)len, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    NFM_CHECK(mp != MAP_FAILED, NFM_ERR_INVALID, "%s cannot be mapped.", x_path);
    raw.p = static_cast<const unsigned char*>(mp);
    raw.n = (size_t)len;
  }
  bool with_fields = false;
  int64_t base = 0;
  if (len >= 14 && !memcmp(raw.data(), "STREAMCSRFIELD", 14)) {
    with_fields = true;
    base = 14 + 48;
  } else if (len >= 9 && !memcmp(raw.data(), "STREAMCSR", 9)) {
    base = 9 + 40;
  } else if (len >= 9 && !memcmp(raw.data(), "STREAMCSC", 9)) {
    return set_error(NFM_ERR_UNSUPPORTED, "%s is a column-major (STREAMCSC) file; the row-wise optimizers need STREAMCSR", x_path);
  } else {
    return set_error(NFM_ERR_INVALID, "%s is not a StreamCSR file.", x_path);
  }
  NFM_CHECK(len >= base, NFM_ERR_INVALID, "%s: truncated header", x_path);
  int64_t hdr[4] = {0, 0, 0, 0};
  memcpy(hdr, raw.data() + (with_fields ? 14 : 9), with_fields ? 32 : 24);
  const int64_t n = hdr[0], d = hdr[1], nnz = hdr[2], nf = with_fields ? hdr[3] : 0;
  NFM_CHECK(n >= 0 && d >= 0 && nnz >= 0 && d < (int64_t)2147483647 - 64, NFM_ERR_INVALID, "%s: bad header", x_path);
  NFM_CHECK(!with_fields || (nf >= 0 && nf < (int64_t)2147483647), NFM_ERR_INVALID, "%s: bad nFields in the header", x_path);
  const int esize = with_fields ? 24 : 16;
  // every row costs at least its 8-byte length word and every entry esize bytes: a header promising more than the
  // file can hold is refused before anything is sized by it
  NFM_CHECK(n <= (len - base) / 8 && nnz <= (len - base) / esize, NFM_ERR_INVALID,
            "%s: the header promises %lld rows / %lld entries, the file holds %lld bytes", x_path, (long long)n, (long long)nnz,
            (long long)len);
  std::vector<int64_t> indptr((size_t)n + 1);
  int64_t pos = base, acc = 0, max_row = 0;
  indptr[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    NFM_CHECK(pos + 8 <= len, NFM_ERR_INVALID, "%s: truncated at row %lld", x_path, (long long)i);
    int64_t r;
    memcpy(&r, raw.data() + pos, 8);
    NFM_CHECK(r >= 0 && r <= (len - pos - 8) / esize, NFM_ERR_INVALID, "%s: row %lld overruns the file", x_path, (long long)i);
    pos += 8 + r * esize;
    acc += r;
    indptr[i + 1] = acc;
    max_row = std::max(max_row, r);
  }
  NFM_CHECK(acc == nnz, NFM_ERR_INVALID, "%s: rows hold %lld entries, header says %lld", x_path, (long long)acc, (long long)nnz);
  DevBuf status;
  NFM_TRY(out->indptr.alloc(sizeof(int64_t) * (n + 1)));
  NFM_HIP_CHECK(hipMemcpyAsync(out->indptr.p, indptr.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, st));
  NFM_TRY(out->indices.alloc(sizeof(int32_t) * std::max<int64_t>(nnz, 1)));
  NFM_TRY(out->data.alloc(sizeof(double) * std::max<int64_t>(nnz, 1)));
  if (with_fields) NFM_TRY(out->fields.alloc(sizeof(int32_t) * std::max<int64_t>(nnz, 1)));
  long long h_st[ST_COUNT];
  for (int i = 0; i < ST_COUNT; ++i) h_st[i] = 0;
  NFM_TRY(status.alloc(sizeof(h_st)));
  NFM_HIP_CHECK(hipMemcpyAsync(status.p, h_st, sizeof(h_st), hipMemcpyHostToDevice, st));
  if (nnz)
    hipLaunchKernelGGL(k_stream_split, dim3(grid_for(nnz)), dim3(kBlock), 0, st, dev_raw.as<unsigned char>(), base, esize,
                       with_fields ? 1 : 0, n, d, nf, out->indptr.as<int64_t>(), nnz, out->indices.as<int32_t>(),
                       out->data.as<double>(), with_fields ? out->fields.as<int32_t>() : nullptr, status.as<long long>());
  NFM_HIP_CHECK(hipGetLastError());
  NFM_HIP_CHECK(hipMemcpyAsync(h_st, status.p, sizeof(h_st), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  NFM_CHECK(h_st[ST_MALFORMED] == 0, NFM_ERR_INVALID, "%s: %lld entries with an id outside [0, nCols) or a field outside [0, nFields) of the header", x_path,
            h_st[ST_MALFORMED]);
  NFM_TRY(out->y.alloc(sizeof(double) * std::max<int64_t>(n, 1)));
  if (y_path) {  // loadStreamLabel (dataset.nim:1007-1014): raw float64, one per sample
    std::vector<unsigned char> yraw;
    NFM_TRY(read_whole(y_path, &yraw));
    NFM_CHECK((int64_t)yraw.size() == 8 * n, NFM_ERR_INVALID, "%s holds %lld labels, the matrix has %lld rows", y_path,
              (long long)(yraw.size() / 8), (long long)n);
    if (n) NFM_HIP_CHECK(hipMemcpy(out->y.p, yraw.data(), (size_t)(8 * n), hipMemcpyHostToDevice));
  } else {
    NFM_HIP_CHECK(hipMemsetAsync(out->y.p, 0, sizeof(double) * std::max<int64_t>(n, 1), st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
  }
  out->n = n;
  out->d = d;
  out->nnz = nnz;
  out->n_fields = nf;
  out->max_row = (int)max_row;
  out->bytes = len;
  return NFM_OK;
}

// ------------------------------------------------------------------------------------------------
// STREAMCSR files in row blocks: the reference's out-of-core epoch (readCache, tensor/sparse_stream.nim:232-270; loop
// optimizer/sgd_multi.nim:83-97: `while nRest > 0: X.readCache(nSamples - nRest) ... dec(nRest, X.nCached)`) keeps a
// cache of rows in host memory and walks the file block by block, in file order.  Here a block is made resident in HBM
// (StreamFile::load_rows) and the epoch runs over it; for files that fit (288 GB) nfm_dataset_load_stream makes
// everything resident at once.
// ------------------------------------------------------------------------------------------------
StreamFile::~StreamFile() {
  if (map) munmap(const_cast<unsigned char*>(map), (size_t)len);
}

int StreamFile::open_file(const char* x_path, const char* y_path_, StreamFile* S) {
  S->path = x_path;
  S->y_path = y_path_ ? y_path_ : "";
  const int fd = open(x_path, O_RDONLY);
  NFM_CHECK(fd >= 0, NFM_ERR_INVALID, "%s cannot be opened.", x_path);
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); return set_error(NFM_ERR_INVALID, "%s cannot be opened.", x_path); }
  S->len = (int64_t)sb.st_size;
  void* mp = S->len > 0 ? mmap(nullptr, (size_t)S->len, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
  close(fd);
  NFM_CHECK(mp != MAP_FAILED, NFM_ERR_INVALID, "%s cannot be mapped.", x_path);
  S->map = static_cast<const unsigned char*>(mp);
  if (S->len >= 14 && !memcmp(S->map, "STREAMCSRFIELD", 14)) {
    S->with_fields = true;
    S->base = 14 + 48;
  } else if (S->len >= 9 && !memcmp(S->map, "STREAMCSR", 9)) {
    S->base = 9 + 40;
  } else if (S->len >= 9 && !memcmp(S->map, "STREAMCSC", 9)) {
    return set_error(NFM_ERR_UNSUPPORTED, "%s is a column-major (STREAMCSC) file; the row-wise optimizers need STREAMCSR", x_path);
  } else {
    return set_error(NFM_ERR_INVALID, "%s is not a StreamCSR file.", x_path);
  }
  NFM_CHECK(S->len >= S->base, NFM_ERR_INVALID, "%s: truncated header", x_path);
  int64_t hdr[4] = {0, 0, 0, 0};
  memcpy(hdr, S->map + (S->with_fields ? 14 : 9), S->with_fields ? 32 : 24);
  S->n = hdr[0]; S->d = hdr[1]; S->nnz = hdr[2]; S->nf = S->with_fields ? hdr[3] : 0;
  S->esize = S->with_fields ? 24 : 16;
  NFM_CHECK(S->n >= 0 && S->d >= 0 && S->nnz >= 0 && S->d < (int64_t)2147483647 - 64, NFM_ERR_INVALID, "%s: bad header", x_path);
  NFM_CHECK(!S->with_fields || (S->nf >= 0 && S->nf < (int64_t)2147483647), NFM_ERR_INVALID, "%s: bad nFields in the header", x_path);
  NFM_CHECK(S->n <= (S->len - S->base) / 8 && S->nnz <= (S->len - S->base) / S->esize, NFM_ERR_INVALID,
            "%s: the header promises %lld rows / %lld entries, the file holds %lld bytes", x_path, (long long)S->n,
            (long long)S->nnz, (long long)S->len);
  S->mark_row.assign(1, 0);
  S->mark_off.assign(1, S->base);
  return NFM_OK;
}

// byte offset of row r: walked from the nearest mark at or before it (blocks are asked for in file order, so the walk
// normally continues where the last one stopped); a mark is left every kMarkEvery rows
int StreamFile::offset_of(int64_t r, int64_t* off_out) {
  constexpr int64_t kMarkEvery = 65536;
  size_t lo = 0, hi = mark_row.size() - 1;
  while (lo < hi) {
    const size_t mid = (lo + hi + 1) / 2;
    if (mark_row[mid] <= r) lo = mid; else hi = mid - 1;
  }
  int64_t row = mark_row[lo], pos = mark_off[lo];
  while (row < r) {
    NFM_CHECK(pos + 8 <= len, NFM_ERR_INVALID, "%s: truncated at row %lld", path.c_str(), (long long)row);
    int64_t c;
    memcpy(&c, map + pos, 8);
    NFM_CHECK(c >= 0 && c <= (len - pos - 8) / esize, NFM_ERR_INVALID, "%s: row %lld overruns the file", path.c_str(), (long long)row);
    pos += 8 + c * esize;
    ++row;
    if (row % kMarkEvery == 0 && row > mark_row.back()) {
      mark_row.push_back(row);
      mark_off.push_back(pos);
    }
  }
  *off_out = pos;
  return NFM_OK;
}

int StreamFile::load_rows(nfm_ctx* ctx, int64_t r0, int64_t r1, IngestResult* out, hipStream_t st_in) {
  NFM_CHECK(r0 >= 0 && r0 <= r1 && r1 <= n, NFM_ERR_INVALID, "rows [%lld,%lld) outside [0,%lld)", (long long)r0, (long long)r1, (long long)n);
  hipStream_t st = st_in ? st_in : ctx->stream;  // every copy below is ordered on st (a prefetch runs beside the epoch's stream)
  const int64_t nr = r1 - r0;
  int64_t off0 = 0;
  NFM_TRY(offset_of(r0, &off0));
  std::vector<int64_t> indptr((size_t)nr + 1);
  int64_t pos = off0, acc = 0, max_row = 0;
  indptr[0] = 0;
  for (int64_t i = 0; i < nr; ++i) {
    NFM_CHECK(pos + 8 <= len, NFM_ERR_INVALID, "%s: truncated at row %lld", path.c_str(), (long long)(r0 + i));
    int64_t c;
    memcpy(&c, map + pos, 8);
    NFM_CHECK(c >= 0 && c <= (len - pos - 8) / esize, NFM_ERR_INVALID, "%s: row %lld overruns the file", path.c_str(), (long long)(r0 + i));
    pos += 8 + c * esize;
    acc += c;
    indptr[i + 1] = acc;
    max_row = std::max(max_row, c);
  }
  const int64_t nbytes = pos - off0, nz = acc;
  DevBuf raw, status;
  NFM_TRY(raw.alloc((size_t)nbytes + 64));
  if (nbytes) NFM_HIP_CHECK(hipMemcpyAsync(raw.p, map + off0, (size_t)nbytes, hipMemcpyHostToDevice, st));
  NFM_TRY(out->indptr.alloc(sizeof(int64_t) * (nr + 1)));
  NFM_HIP_CHECK(hipMemcpyAsync(out->indptr.p, indptr.data(), sizeof(int64_t) * (nr + 1), hipMemcpyHostToDevice, st));
  NFM_TRY(out->indices.alloc(sizeof(int32_t) * std::max<int64_t>(nz, 1)));
  NFM_TRY(out->data.alloc(sizeof(double) * std::max<int64_t>(nz, 1)));
  if (with_fields) NFM_TRY(out->fields.alloc(sizeof(int32_t) * std::max<int64_t>(nz, 1)));
  long long h_st[ST_COUNT];
  for (int i = 0; i < ST_COUNT; ++i) h_st[i] = 0;
  NFM_TRY(status.alloc(sizeof(h_st)));
  NFM_HIP_CHECK(hipMemcpyAsync(status.p, h_st, sizeof(h_st), hipMemcpyHostToDevice, st));
  if (nz)
    hipLaunchKernelGGL(k_stream_split, dim3(grid_for(nz)), dim3(kBlock), 0, st, raw.as<unsigned char>(), (int64_t)0, esize,
                       with_fields ? 1 : 0, nr, d, nf, out->indptr.as<int64_t>(), nz, out->indices.as<int32_t>(),
                       out->data.as<double>(), with_fields ? out->fields.as<int32_t>() : nullptr, status.as<long long>());
  NFM_HIP_CHECK(hipGetLastError());
  NFM_HIP_CHECK(hipMemcpyAsync(h_st, status.p, sizeof(h_st), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  NFM_CHECK(h_st[ST_MALFORMED] == 0, NFM_ERR_INVALID, "%s: %lld entries with an id outside [0, nCols) or a field outside [0, nFields) of the header",
            path.c_str(), h_st[ST_MALFORMED]);
  NFM_TRY(out->y.alloc(sizeof(double) * std::max<int64_t>(nr, 1)));
  if (!y_path.empty()) {  // loadStreamLabel (dataset.nim:1007-1014): raw float64, one per sample
    FILE* f = fopen(y_path.c_str(), "rb");
    NFM_CHECK(f, NFM_ERR_INVALID, "%s cannot be opened.", y_path.c_str());
    std::vector<double> yb((size_t)nr);
    const bool ok = fseeko(f, (off_t)(8 * r0), SEEK_SET) == 0 && (nr == 0 || fread(yb.data(), 8, (size_t)nr, f) == (size_t)nr);
    fclose(f);
    NFM_CHECK(ok, NFM_ERR_INVALID, "%s holds fewer than %lld labels", y_path.c_str(), (long long)r1);
    if (nr) {
      NFM_HIP_CHECK(hipMemcpyAsync(out->y.p, yb.data(), (size_t)(8 * nr), hipMemcpyHostToDevice, st));
      NFM_HIP_CHECK(hipStreamSynchronize(st));  // yb goes out of scope
    }
  } else {
    NFM_HIP_CHECK(hipMemsetAsync(out->y.p, 0, sizeof(double) * std::max<int64_t>(nr, 1), st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
  }
  out->n = nr;
  out->d = d;
  out->nnz = nz;
  out->n_fields = nf;
  out->max_row = (int)max_row;
  out->bytes = nbytes;
  return NFM_OK;
}

int convert_svmlight(nfm_ctx* ctx, const char* f_in, const char* f_out_x, const char* f_out_y) {
  IngestResult r;
  NFM_TRY(ingest_text(ctx, f_in, nullptr, 0, false, &r));
  hipStream_t st = ctx->stream;
  std::vector<int64_t> indptr((size_t)r.n + 1);
  std::vector<int32_t> idx((size_t)r.nnz);
  std::vector<double> val((size_t)r.nnz), y((size_t)r.n);
  NFM_HIP_CHECK(hipMemcpyAsync(indptr.data(), r.indptr.p, sizeof(int64_t) * (r.n + 1), hipMemcpyDeviceToHost, st));
  if (r.nnz) {
    NFM_HIP_CHECK(hipMemcpyAsync(idx.data(), r.indices.p, sizeof(int32_t) * r.nnz, hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipMemcpyAsync(val.data(), r.data.p, sizeof(double) * r.nnz, hipMemcpyDeviceToHost, st));
  }
  if (r.n) NFM_HIP_CHECK(hipMemcpyAsync(y.data(), r.y.p, sizeof(double) * r.n, hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  // dataset.nim:1021-1058: ids are shifted by the smallest index seen (NOT by the 0/1 base of the
  // text loaders), nCols = maxIndex - minIndex + 1, the header carries the value range
  const int64_t shift = r.min_index - r.offset;  // what is still to subtract from the loader's 0-based ids
  double vmin = HUGE_VAL, vmax = -HUGE_VAL;  // Nim's high(float64) / low(float64) are +-Inf
  for (double v : val) {
    vmin = v < vmin ? v : vmin;  // min(minVal, val) / max(maxVal, val): a NaN never replaces the bound
    vmax = v > vmax ? v : vmax;
  }
  FILE* fx = fopen(f_out_x, "wb");
  NFM_CHECK(fx, NFM_ERR_INVALID, "%s cannot be read.", f_out_x);
  FILE* fy = fopen(f_out_y, "wb");
  if (!fy) {
    fclose(fx);
    return set_error(NFM_ERR_INVALID, "%s cannot be read.", f_out_y);
  }
  const int64_t hdr[3] = {r.n, r.max_index - r.min_index + 1, r.nnz};
  const double mm[2] = {vmax, vmin};
  bool ok = fwrite("STREAMCSR", 1, 9, fx) == 9 && fwrite(hdr, 8, 3, fx) == 3 && fwrite(mm, 8, 2, fx) == 2;
  std::vector<unsigned char> row;
  for (int64_t i = 0; ok && i < r.n; ++i) {
    const int64_t a = indptr[i], b = indptr[i + 1], m = b - a;
    row.resize((size_t)(8 + 16 * m));
    memcpy(row.data(), &m, 8);
    for (int64_t q = 0; q < m; ++q) {
      const int64_t id = (int64_t)idx[a + q] - shift;
      memcpy(row.data() + 8 + 16 * q, &val[a + q], 8);
      memcpy(row.data() + 16 + 16 * q, &id, 8);
    }
    ok = fwrite(row.data(), 1, row.size(), fx) == row.size();
  }
  ok = ok && (r.n == 0 || fwrite(y.data(), 8, (size_t)r.n, fy) == (size_t)r.n);
  fclose(fx);
  fclose(fy);
  NFM_CHECK(ok, NFM_ERR_INVALID, "write to %s / %s failed", f_out_x, f_out_y);
  return NFM_OK;
}

}  // namespace nfm
