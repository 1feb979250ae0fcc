// nimfm_amd/csrc/parse_num.h -- text -> number for the ingest kernels (ingest.hip), compiled for host
// and device from the same source.
//
// The reference reads svmlight/FFM text with Nim's parseutils.parseInt / parseFloat
// (dataset.nim:574-582, 708-724), whose parseFloat is correctly rounded (an exact fast path, C strtod
// otherwise).  parse_float below is correctly rounded too: decimal significand (up to 19 digits) and
// exponent, then the Eisel-Lemire conversion (D. Lemire, "Number Parsing at a Gigabyte per Second",
// SPE 2021) with the 128-bit power-of-five table of pow5_table.h; by N. Mushtak & D. Lemire, "Fast
// Number Parsing Without Fallback" (SPE 2023) the two-word product always decides the rounding for a
// significand below 2^64.  Longer significands are truncated to 19 digits and converted twice (w and
// w + 1); if the two disagree the token is flagged and the host re-reads it with strtod.
// Accepted syntax: [+-] digits [. digits] [(e|E) [+-] digits] | [+-] nan | [+-] inf[inity]
// (Nim also allows '_' between digits; data files do not use it).  A call returns the number of
// characters consumed, 0 when there is no number at the position (as Nim's procs do).
#pragma once
#include <stdint.h>

#include "pow5_table.h"

#if defined(__HIPCC__)
#define NFM_HD __host__ __device__ __forceinline__
#else
#define NFM_HD inline
#endif

namespace nfm {
namespace num {

struct Pow5 {
  uint64_t hi, lo;
};

enum { kNeedsStrtod = 1 };  // parse_float flag: > 19 significant digits and the truncation matters

NFM_HD void mul64(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  lo = a * b;
  hi = __umul64hi(a, b);
#else
  const unsigned __int128 p = (unsigned __int128)a * b;
  lo = (uint64_t)p;
  hi = (uint64_t)(p >> 64);
#endif
}

NFM_HD int clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __clzll((long long)x);
#else
  return __builtin_clzll(x);
#endif
}

NFM_HD double bits_to_double(uint64_t b) {
  union {
    uint64_t u;
    double d;
  } c;
  c.u = b;
  return c.d;
}

// w * 10^q -> nearest binary64 (ties to even); w != 0
NFM_HD double decimal_to_double(uint64_t w, int64_t q, const Pow5* table) {
  if (q < NFM_POW5_QMIN) return 0.0;
  if (q > NFM_POW5_QMAX) return bits_to_double(0x7FFull << 52);
  int lz = clz64(w);
  w <<= lz;
  const Pow5 t = table[q - NFM_POW5_QMIN];
  uint64_t first_hi, first_lo;
  mul64(w, t.hi, first_hi, first_lo);
  if ((first_hi & 0x1FF) == 0x1FF) {  // the 55 bits that matter could still change: add the low word
    uint64_t second_hi, second_lo;
    mul64(w, t.lo, second_hi, second_lo);
    first_lo += second_hi;
    if (second_hi > first_lo) ++first_hi;
  }
  const uint64_t lower = first_lo, upper = first_hi;
  const int upperbit = (int)(upper >> 63);
  uint64_t mantissa = upper >> (upperbit + 9);
  int64_t power2 = (((152170 + 65536) * q) >> 16) + 63 + upperbit - lz + 1023;
  if (power2 <= 0) {  // subnormal (or zero)
    if (-power2 + 1 >= 64) return 0.0;
    mantissa >>= -power2 + 1;
    mantissa += mantissa & 1;
    mantissa >>= 1;
    power2 = mantissa < (1ull << 52) ? 0 : 1;
    return bits_to_double((mantissa & ~(1ull << 52)) | ((uint64_t)power2 << 52));
  }
  // exactly halfway between two doubles: round to even
  if (lower <= 1 && q >= -4 && q <= 23 && (mantissa & 3) == 1) {
    if ((mantissa << (upperbit + 9)) == upper) mantissa &= ~1ull;
  }
  mantissa += mantissa & 1;
  mantissa >>= 1;
  if (mantissa >= (2ull << 52)) {
    mantissa = 1ull << 52;
    ++power2;
  }
  mantissa &= ~(1ull << 52);
  if (power2 >= 0x7FF) return bits_to_double(0x7FFull << 52);
  return bits_to_double(mantissa | ((uint64_t)power2 << 52));
}

NFM_HD bool is_digit(char c) { return c >= '0' && c <= '9'; }
NFM_HD char lower(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c + 32) : c; }

// Nim parseutils.parseInt: [+-] digits.  Returns characters consumed (0: no integer here).
NFM_HD int parse_int(const char* s, int64_t n, int64_t* out) {
  int64_t i = 0;
  bool neg = false;
  if (i < n && (s[i] == '+' || s[i] == '-')) {
    neg = s[i] == '-';
    ++i;
  }
  if (i >= n || !is_digit(s[i])) return 0;
  // (Nim raises ValueError "Parsed integer outside of valid range" past int64; here the value saturates -- every caller
  // range-checks it against nFeatures / nFields afterwards -- instead of overflowing a signed integer: undefined behaviour,
  // found by the UBSan build of tests/test_sanitizers.py)
  const uint64_t lim = 0x7FFFFFFFFFFFFFFFull;
  uint64_t v = 0;
  bool sat = false;
  while (i < n && is_digit(s[i])) {
    const uint64_t dgt = (uint64_t)(s[i] - '0');
    if (v > (lim - dgt) / 10) sat = true;
    else v = v * 10 + dgt;
    ++i;
  }
  if (sat) v = lim;
  *out = neg ? -(int64_t)v : (int64_t)v;
  return (int)i;
}

// Nim parseutils.parseFloat.  Returns characters consumed (0: no number here); *flags |= kNeedsStrtod
// when the caller must re-read the token with strtod.
NFM_HD int parse_float(const char* s, int64_t n, double* out, int* flags, const Pow5* table) {
  int64_t i = 0;
  bool neg = false;
  if (i < n && (s[i] == '+' || s[i] == '-')) {
    neg = s[i] == '-';
    ++i;
  }
  if (i + 3 <= n) {
    const char a = lower(s[i]), b = lower(s[i + 1]), c = lower(s[i + 2]);
    if (a == 'n' && b == 'a' && c == 'n') {
      *out = bits_to_double(0x7FF8ull << 48);
      return (int)(i + 3);
    }
    if (a == 'i' && b == 'n' && c == 'f') {
      int64_t e = i + 3;
      if (e + 5 <= n && lower(s[e]) == 'i' && lower(s[e + 1]) == 'n' && lower(s[e + 2]) == 'i' && lower(s[e + 3]) == 't' &&
          lower(s[e + 4]) == 'y')
        e += 5;
      *out = bits_to_double((neg ? 0xFFFull : 0x7FFull) << 52);
      return (int)e;
    }
  }
  uint64_t w = 0;
  int nd = 0;          // significant digits taken into w
  int64_t dropped = 0; // integer digits beyond the 19th
  bool truncated = false, any = false;
  while (i < n && is_digit(s[i])) {
    any = true;
    if (w == 0 && s[i] == '0') {
      // leading zero
    } else if (nd < 19) {
      w = w * 10 + (uint64_t)(s[i] - '0');
      ++nd;
    } else {
      ++dropped;
      if (s[i] != '0') truncated = true;
    }
    ++i;
  }
  int64_t frac = 0;  // fractional digits taken into w
  if (i < n && s[i] == '.') {
    int64_t k = i + 1;
    bool anyf = false;
    while (k < n && is_digit(s[k])) {
      anyf = true;
      if (w == 0 && s[k] == '0') {
        ++frac;  // leading zero after the point
      } else if (nd < 19) {
        w = w * 10 + (uint64_t)(s[k] - '0');
        ++nd;
        ++frac;
      } else if (s[k] != '0') {
        truncated = true;
      }
      ++k;
    }
    if (any || anyf) {
      any = true;
      i = k;
    }
  }
  if (!any) return 0;
  int64_t ex = 0;
  if (i < n && (s[i] == 'e' || s[i] == 'E')) {
    int64_t k = i + 1;
    bool eneg = false;
    if (k < n && (s[k] == '+' || s[k] == '-')) {
      eneg = s[k] == '-';
      ++k;
    }
    if (k < n && is_digit(s[k])) {
      while (k < n && is_digit(s[k])) {
        if (ex < 100000) ex = ex * 10 + (s[k] - '0');
        ++k;
      }
      if (eneg) ex = -ex;
      i = k;
    }
  }
  double v = 0.0;
  if (w != 0) {
    const int64_t q = ex - frac + dropped;
    v = decimal_to_double(w, q, table);
    if (truncated) {
      const double v1 = decimal_to_double(w + 1, q, table);
      if (v1 != v) *flags |= kNeedsStrtod;
    }
  }
  *out = neg ? -v : v;
  return (int)i;
}

}  // namespace num
}  // namespace nfm
