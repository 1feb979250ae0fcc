// nimfm_amd/csrc/ingest.h -- text datasets parsed on the GPU (ingest.hip)
#pragma once
#include "common.h"

#include <string>
#include <vector>

namespace nfm {

struct IngestResult {
  DevBuf indptr, indices, data, fields, y;  // int64[n+1], int32[nnz], double[nnz], int32[nnz] (FFM), double[n]
  int64_t n = 0, d = 0, nnz = 0, n_fields = 0;
  int64_t offset = 0, offset_field = 0;     // index base found in the file (dataset.nim:589, 733)
  int64_t min_index = 1, max_index = 0;      // as the reference tracks them (initial 1 / 0, dataset.nim:568-569)
  int max_row = 0;
  int64_t bytes = 0;
  double upload_ms = 0.0, parse_ms = 0.0;
};

// path != nullptr: read the file; else parse mem[0, mem_len).  with_fields: libffm "field:index:value".
int ingest_text(nfm_ctx* ctx, const char* path, const char* mem, int64_t mem_len, bool with_fields, IngestResult* out);

// STREAMCSR / STREAMCSRFIELD binary files (tensor/sparse_stream.nim:3-33) -> CSR in HBM; y_path may be null
int ingest_stream(nfm_ctx* ctx, const char* x_path, const char* y_path, IngestResult* out);
// a STREAMCSR / STREAMCSRFIELD file opened for block-wise use (nfm_stream_*): header, a read-only mapping, and marks
// (row -> byte offset) left behind by the walks over the per-row length words
struct StreamFile {
  std::string path, y_path;
  const unsigned char* map = nullptr;
  int64_t len = 0, base = 0, n = 0, d = 0, nnz = 0, nf = 0;
  int esize = 16;
  bool with_fields = false;
  std::vector<int64_t> mark_row, mark_off;
  ~StreamFile();
  static int open_file(const char* x_path, const char* y_path, StreamFile* S);
  int offset_of(int64_t r, int64_t* off_out);
  // rows [r0, r1) -> CSR in HBM; st: the stream the upload and the split run on (default: the context's)
  int load_rows(nfm_ctx* ctx, int64_t r0, int64_t r1, IngestResult* out, hipStream_t st = nullptr);
};

// convertSVMLightFile (dataset.nim:1017-1097): text -> STREAMCSR + raw float64 labels
int convert_svmlight(nfm_ctx* ctx, const char* f_in, const char* f_out_x, const char* f_out_y);

}  // namespace nfm
