/* oracle/nimfm_ingest.c -- TEST / BASELINE INFRASTRUCTURE ONLY (never linked into the product path).
 * C restatement of the reference's svmlight loader, two passes, line by line, as the Nim code walks
 * the file (/root/reference/src/nimfm/dataset.nim:562-613): parseFloat / parseInt return the number
 * of characters consumed (0: the variable keeps its value), one character is skipped after every
 * token.  strtod is correctly rounded like Nim's parseFloat.  Used as the CPU baseline of
 * bench.py --workload ingest and checked against oracle/ingest.py in tests/test_oracle_ingest.py.
 * Parity unpinned against reference-run outputs (no Nim toolchain here). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int is_dig(char c) { return c >= '0' && c <= '9'; }

/* Nim parseutils.parseInt on line[k..n) */
static int64_t p_int(const char* s, int64_t k, int64_t n, int64_t* out) {
  int64_t i = k;
  int neg = 0;
  if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; ++i; }
  if (i >= n || !is_dig(s[i])) return 0;
  int64_t v = 0;
  while (i < n && is_dig(s[i])) { v = v * 10 + (s[i] - '0'); ++i; }
  *out = neg ? -v : v;
  return i - k;
}

/* Nim parseutils.parseFloat on line[k..n): the token is bounded first (strtod accepts more, e.g. hex) */
static int64_t p_float(const char* s, int64_t k, int64_t n, double* out) {
  int64_t i = k;
  if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
  if (i + 3 <= n && (s[i] == 'n' || s[i] == 'N' || s[i] == 'i' || s[i] == 'I')) {
    char buf[16];
    int64_t m = n - k < 15 ? n - k : 15;
    memcpy(buf, s + k, (size_t)m);
    buf[m] = 0;
    char* end;
    *out = strtod(buf, &end);
    return end - buf;
  }
  int any = 0;
  while (i < n && is_dig(s[i])) { ++i; any = 1; }
  if (i < n && s[i] == '.') {
    int64_t j = i + 1;
    int anyf = 0;
    while (j < n && is_dig(s[j])) { ++j; anyf = 1; }
    if (any || anyf) { any = 1; i = j; }
  }
  if (!any) return 0;
  if (i < n && (s[i] == 'e' || s[i] == 'E')) {
    int64_t j = i + 1;
    if (j < n && (s[j] == '+' || s[j] == '-')) ++j;
    if (j < n && is_dig(s[j])) { while (j < n && is_dig(s[j])) ++j; i = j; }
  }
  char buf[512];
  int64_t m = i - k < 511 ? i - k : 511;
  memcpy(buf, s + k, (size_t)m);
  buf[m] = 0;
  *out = strtod(buf, NULL);
  return i - k;
}

static int64_t line_end(const char* t, int64_t len, int64_t s, int64_t* next) {
  int64_t e = s;
  while (e < len && t[e] != '\n') ++e;
  *next = e + 1;
  if (e > s && t[e - 1] == '\r') --e;
  return e;
}

/* pass 1 (dataset.nim:571-590): counts and index range.  returns 0, or -1 for a negative index */
int orc_svmlight_scan(const char* t, int64_t len, int64_t* n_samples, int64_t* nnz, int64_t* n_features, int64_t* offset) {
  int64_t ns = 0, nz = 0, j = 0, mn = 1, mx = 0;
  double val = 0, target = 0;
  for (int64_t s = 0; s < len;) {
    int64_t next;
    const int64_t e = line_end(t, len, s, &next);
    ++ns;
    int64_t k = s;
    k += p_float(t, k, e, &target);
    ++k;
    while (k < e) {
      k += p_int(t, k, e, &j);
      if (j < mn) mn = j;
      if (j > mx) mx = j;
      ++k;
      k += p_float(t, k, e, &val);
      ++k;
      ++nz;
    }
    s = next;
  }
  *n_samples = ns;
  *nnz = nz;
  *offset = mn == 0 ? 0 : 1;
  *n_features = mx + 1 - *offset;
  return mn < 0 ? -1 : 0;
}

/* pass 2 (dataset.nim:598-612) */
void orc_svmlight_fill(const char* t, int64_t len, int64_t offset, int64_t* indptr, int64_t* indices, double* data, double* y) {
  int64_t i = 0, nz = 0, j = 0;
  double val = 0, target = 0;
  indptr[0] = 0;
  for (int64_t s = 0; s < len;) {
    int64_t next;
    const int64_t e = line_end(t, len, s, &next);
    int64_t k = s;
    k += p_float(t, k, e, &target);
    y[i] = target;
    ++k;
    while (k < e) {
      k += p_int(t, k, e, &j);
      indices[nz] = j - offset;
      ++k;
      k += p_float(t, k, e, &val);
      data[nz] = val;
      ++k;
      ++nz;
    }
    indptr[++i] = nz;
    s = next;
  }
}
