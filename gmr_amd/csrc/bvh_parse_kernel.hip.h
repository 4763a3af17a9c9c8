// bvh_parse_kernel.hip.h -- the MOTION block of BVH files parsed on the GPU (text bytes in HBM -> float64 rows in HBM).
//
// Replaces the per-line `map(float, line.strip().split(' '))` loop of read_bvh (reference
// general_motion_retargeting/utils/lafan_vendor/extract.py:140-156), like the host parser gmr_bvh_parse_motion (api.hip) whose
// results it reproduces bit for bit: a decimal literal with at most 19 significant digits, a mantissa below 2^53 and a power of
// ten within 10^+-22 is ONE correctly rounded multiply or divide of two exact doubles (Clinger's fast path) -- what Python's
// float() returns.  Every other token (longer mantissas, huge exponents, inf / nan, anything malformed) is only reported: the host
// decides it with strtod and patches the value in, so the device never guesses.
//
// Layout of the work: the text of a batch of files is one byte array; a file's motion block is a segment of it, cut into chunks
// of 4096 bytes, one wavefront per chunk, 64 bytes per lane.  A lane owns the tokens that START in its 64 bytes and may follow one
// 32 bytes further.  Three launches: count the token starts of every chunk; scan the counts per file (a token's index in its file
// is its place in the output rows: row = index / n_cols); parse.  Line structure is checked where it matters: a token is the first
// of its line exactly when its index is a multiple of n_cols -- anything else (ragged rows) marks the file for the host parser, which
// reports it the way it always did.  Blank lines are skipped, tokens behind the first n_lines rows are ignored, as on the host.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmr {

constexpr int kTxtSpan = 64;                  // bytes whose token starts a lane owns
constexpr int kTxtHalo = 32;                  // bytes behind the span a token may run into (longer tokens go to the host)
constexpr int kTxtChunk = 64 * kTxtSpan;      // bytes per wavefront
constexpr int kTxtWords = (kTxtSpan + kTxtHalo) / 4;

struct TxtFile {      // one file's motion block (device array, one entry per file)
  int64_t seg_begin, seg_end;   // byte range in the text array
  int64_t chunk0;               // index of the file's first chunk among all chunks of the batch
  int64_t tok_limit;            // n_lines * n_cols: tokens to read
  int64_t out_base;             // row_begin * n_cols: where the file's first value goes in rows_out
  int64_t n_tokens;             // (out) token starts found in the whole block
  int32_t status;               // (out) 0 ok, bit 0: line structure differs from n_cols per row
  int32_t pad;
};

__device__ __forceinline__ bool txt_ws(unsigned c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }

// chunk -> file: the last file whose chunk0 <= chunk (wave-uniform binary search over the few-hundred-entry table)
__device__ __forceinline__ int txt_file_of(const TxtFile *files, int n_files, int64_t chunk) {
  int lo = 0, hi = n_files - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (files[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// This lane's 96 bytes (span + halo) as 24 words; bytes at or behind seg_end read as '\n'.
__device__ __forceinline__ void txt_load(const unsigned char *__restrict__ text, int64_t pos, int64_t seg_end, unsigned (&w)[kTxtWords]) {
#pragma unroll
  for (int k = 0; k < kTxtWords / 4; ++k) {
    const int64_t p = pos + 16 * k;
    uint4 v = make_uint4(0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au);
    if (p + 16 <= seg_end) {
      struct __attribute__((packed, aligned(1))) U4 { unsigned a, b, c, d; };
      const U4 u = *reinterpret_cast<const U4 *>(text + p);
      v = make_uint4(u.a, u.b, u.c, u.d);
    } else if (p < seg_end) {
      unsigned t[4] = {0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au};
      for (int i = 0; i < 16 && p + i < seg_end; ++i) t[i >> 2] = (t[i >> 2] & ~(0xffu << (8 * (i & 3)))) | ((unsigned)text[p + i] << (8 * (i & 3)));
      v = make_uint4(t[0], t[1], t[2], t[3]);
    }
    w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
  }
}

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int *total) {
  int s = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(s, d);
    if (lane >= d) s += o;
  }
  *total = __shfl(s, 63);
  return s - v;
}

// Launch 1: token starts per chunk.  A token starts at a non-blank byte whose predecessor is blank (or the segment's begin).
__global__ void __launch_bounds__(64) bvh_txt_count_kernel(const unsigned char *__restrict__ text, const TxtFile *__restrict__ files, int n_files,
                                                           int *__restrict__ counts) {
  const int lane = threadIdx.x;
  const int64_t chunk = blockIdx.x;
  const int f = txt_file_of(files, n_files, chunk);
  const int64_t seg_begin = files[f].seg_begin, seg_end = files[f].seg_end;
  const int64_t pos = seg_begin + (chunk - files[f].chunk0) * kTxtChunk + (int64_t)lane * kTxtSpan;
  int n = 0;
  if (pos < seg_end) {
    bool prev_ws = pos == seg_begin ? true : txt_ws(text[pos - 1]);
#pragma unroll
    for (int k = 0; k < kTxtSpan / 16; ++k) {
      unsigned w[4] = {0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au};
      const int64_t p = pos + 16 * k;
      if (p + 16 <= seg_end) {
        struct __attribute__((packed, aligned(1))) U4 { unsigned a, b, c, d; };
        const U4 u = *reinterpret_cast<const U4 *>(text + p);
        w[0] = u.a; w[1] = u.b; w[2] = u.c; w[3] = u.d;
      } else {
        for (int i = 0; i < 16 && p + i < seg_end; ++i) w[i >> 2] = (w[i >> 2] & ~(0xffu << (8 * (i & 3)))) | ((unsigned)text[p + i] << (8 * (i & 3)));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool ws = txt_ws((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
        n += (!ws && prev_ws) ? 1 : 0;
        prev_ws = ws;
      }
    }
  }
  int total;
  (void)wave_excl_scan(n, lane, &total);
  if (lane == 0) counts[chunk] = total;
}

// Launch 2: one wavefront per file turns its chunks' counts into exclusive prefix sums (in place) and leaves the file's total.
__global__ void __launch_bounds__(64) bvh_txt_scan_kernel(TxtFile *__restrict__ files, int *__restrict__ counts_lo, int64_t *__restrict__ bases) {
  const int lane = threadIdx.x;
  TxtFile &fl = files[blockIdx.x];
  const int64_t nchunk = (fl.seg_end - fl.seg_begin + kTxtChunk - 1) / kTxtChunk;
  int64_t run = 0;
  for (int64_t c0 = 0; c0 < nchunk; c0 += 64) {
    const int64_t c = c0 + lane;
    const int v = c < nchunk ? counts_lo[fl.chunk0 + c] : 0;
    int total;
    const int ex = wave_excl_scan(v, lane, &total);
    if (c < nchunk) bases[fl.chunk0 + c] = run + ex;
    run += total;
  }
  if (lane == 0) { fl.n_tokens = run; fl.status = 0; }
}

// Launch 3: parse.  rows_out[out_base + t] for every token t < tok_limit of every file; tokens outside the exact fast path are
// appended to slow_out as (file, token index, byte offset) triples (slow_count counts all of them, also those that did not fit).
__global__ void __launch_bounds__(64) bvh_txt_parse_kernel(const unsigned char *__restrict__ text, TxtFile *__restrict__ files, int n_files,
                                                           const int64_t *__restrict__ bases, int64_t n_cols, double *__restrict__ rows_out,
                                                           int64_t *__restrict__ slow_out, int64_t max_slow, unsigned long long *__restrict__ slow_count) {
  __shared__ unsigned lw[kTxtWords][64];  // word k of lane l at [k][l]: the byte loop reads conflict-free
  const int lane = threadIdx.x;
  const int64_t chunk = blockIdx.x;
  const int f = txt_file_of(files, n_files, chunk);
  const int64_t seg_begin = files[f].seg_begin, seg_end = files[f].seg_end;
  const int64_t limit = files[f].tok_limit, out_base = files[f].out_base;
  const int64_t pos = seg_begin + (chunk - files[f].chunk0) * kTxtChunk + (int64_t)lane * kTxtSpan;
  const bool live = pos < seg_end;
  {
    unsigned w[kTxtWords];
#pragma unroll
    for (int k = 0; k < kTxtWords; ++k) w[k] = 0x0a0a0a0au;
    if (live) txt_load(text, pos, seg_end, w);
#pragma unroll
    for (int k = 0; k < kTxtWords; ++k) lw[k][lane] = w[k];
  }
  __syncthreads();
  auto byte_at = [&](int i) -> unsigned { return (lw[i >> 2][lane] >> (8 * (i & 3))) & 0xffu; };

  // what precedes the span: is its first byte inside a token, and has a line ended since the last token?
  bool prev_ws = true, nl_pending = true;
  if (live && pos > seg_begin) {
    prev_ws = txt_ws(text[pos - 1]);
    nl_pending = false;
    if (prev_ws) {
      int64_t q = pos - 1;
      int steps = 0;
      for (;;) {
        const unsigned c = text[q];
        if (c == '\n') { nl_pending = true; break; }
        if (!txt_ws(c)) break;
        if (q == seg_begin) { nl_pending = true; break; }
        --q;
        if (++steps > 4096) { atomicOr(&files[f].status, 1); break; }  // (a kilobyte-long run of blanks: let the host look at it)
      }
    }
  }
  int n = 0;
  if (live) {
    bool pw = prev_ws;
    for (int i = 0; i < kTxtSpan; ++i) {
      const bool ws = txt_ws(byte_at(i));
      n += (!ws && pw) ? 1 : 0;
      pw = ws;
    }
  }
  int total;
  int64_t t = bases[chunk] + wave_excl_scan(n, lane, &total);  // index (in its file) of this lane's first token
  if (!live || n == 0 || t > limit) return;   // (t == limit: the first token behind the rows asked for still has to START a line, below)
  int64_t col = t % n_cols;

  const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  int i = 0;
  bool pw = prev_ws;
  while (i < kTxtSpan && n > 0 && t <= limit) {
    const unsigned c0 = byte_at(i);
    if (c0 == '\n') nl_pending = true;
    const bool ws0 = txt_ws(c0);
    if (ws0 || !pw) { pw = ws0; ++i; continue; }
    // a token starts at byte i
    if (t == limit) {  // nothing behind the last row is read -- but a token on the SAME line makes that row longer than n_cols (the host: -1)
      if (!nl_pending) atomicOr(&files[f].status, 1);
      break;
    }
    if ((col == 0) != nl_pending) atomicOr(&files[f].status, 1);
    nl_pending = false;
    const int start = i;
    bool neg = false, digits = false, fast = true, ok = true;
    unsigned long long mant = 0;
    int nd = 0, e10 = 0;
    unsigned c = c0;
    auto adv = [&]() { ++i; c = i < kTxtSpan + kTxtHalo ? byte_at(i) : 0xffu; };  // 0xff: ran out of the lane's bytes
    if (c == '-' || c == '+') { neg = c == '-'; adv(); }
    while (c >= '0' && c <= '9') {
      digits = true;
      if (nd < 19) { mant = mant * 10ull + (c - '0'); if (mant) ++nd; } else fast = false;
      adv();
    }
    if (c == '.') {
      adv();
      while (c >= '0' && c <= '9') {
        digits = true;
        if (nd < 19) { mant = mant * 10ull + (c - '0'); if (mant) ++nd; --e10; } else fast = false;
        adv();
      }
    }
    if (digits && (c == 'e' || c == 'E')) {
      // (the host parser takes the exponent only when a digit follows; a token it would reject goes to the host anyway)
      adv();
      bool eneg = false;
      if (c == '-' || c == '+') { eneg = c == '-'; adv(); }
      int ev = 0;
      bool ed = false;
      while (c >= '0' && c <= '9') { ed = true; if (ev < 10000) ev = ev * 10 + (int)(c - '0'); adv(); }
      if (!ed) ok = false;
      e10 += eneg ? -ev : ev;
    }
    const bool ends = c != 0xffu && txt_ws(c);
    if (ok && digits && ends && fast && mant < (1ull << 53) && e10 >= -22 && e10 <= 22) {
      double v = (double)mant;
      v = e10 < 0 ? v / p10[-e10] : v * p10[e10];
      rows_out[out_base + t] = neg ? -v : v;
    } else {
      const unsigned long long k = atomicAdd(slow_count, 1ull);
      if ((int64_t)k < max_slow) { slow_out[3 * k] = f; slow_out[3 * k + 1] = t; slow_out[3 * k + 2] = pos + start; }
      rows_out[out_base + t] = 0.0;
      while (c != 0xffu && !txt_ws(c)) adv();  // skip the rest of the token as far as this lane sees it
    }
    // the token's last byte decides what the next byte sees
    pw = false;
    --n; ++t;
    if (++col == n_cols) col = 0;
    if (i >= kTxtSpan) break;  // ran into the halo: the next lane owns what starts there
  }
}

}  // namespace gmr
