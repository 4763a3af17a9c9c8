// bvh_text.h -- host-side tokenizer for the HIERARCHY section and MOTION header of a BVH file (gmr_bvh_parse_header).
//
// Replaces the line-by-line hierarchy loop of the reference's read_bvh (general_motion_retargeting/utils/lafan_vendor/
// extract.py:60-139) with a recursive-descent parser over whitespace-separated tokens; line structure is irrelevant.
//
//   file     := "HIERARCHY" joint "MOTION" "Frames:" INT "Frame" "Time:" NUM  <motion rows>
//   joint    := ("ROOT" | "JOINT") NAME "{" "OFFSET" NUM NUM NUM "CHANNELS" INT CHANNEL{INT} (joint | endsite)* "}"
//   endsite  := "End" "Site" "{" "OFFSET" NUM NUM NUM "}"
//   CHANNEL  := ("X" | "Y" | "Z") ("position" | "rotation" | "scale")
//
// Semantics kept from the reference so that files load identically: joints are numbered in the order they appear; an
// end site contributes nothing; the Euler order is read off the first joint whose channels 0-2 (a 3-channel joint) or 3-5
// (any other count) are all rotations (:104-113: a positions-only root of the 9-channel layout is skipped and the next joint
// decides); the per-joint channel count that shapes the motion rows is the LAST joint's (:104-106), reported per
// joint here so the caller can reject mixed files it cannot lay out; NAME keeps the leading run of [A-Za-z0-9_] of its token
// (the reference captures `\w+`); numbers are decimal literals rounded like Python's float() (strtod).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace gmr_bvh {

struct Cursor {
  const char *p, *end;
  const char *tok = nullptr;
  size_t len = 0;
  bool next() {  // advance to the next whitespace-separated token
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
    if (p >= end) { tok = nullptr; len = 0; return false; }
    tok = p;
    while (p < end && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
    len = (size_t)(p - tok);
    return true;
  }
  bool is(const char *s) const { return tok && len == strlen(s) && memcmp(tok, s, len) == 0; }
};

struct Header {
  int max_joints; char *names; size_t names_cap, names_used = 0;
  int32_t *parents; double *offsets; int32_t *channels; int32_t order[3] = {-1, -1, -1};
  int n = 0;
};

// The tokens strtod and Python's float() read ALIKE: [+-] digits [. digits] [(e|E) [+-] digits] with at least one mantissa digit, or
// inf / infinity / nan in any case.  strtod alone would also take hexadecimal floats ("0x10" = 16) and "nan(...)", which float() --
// what the reference parses every number with (extract.py:140-156) -- rejects; float() alone also takes digit-group underscores
// ("1_0"), which are refused here: a refused token fails the file, it never yields a different number.
inline bool float_token(const char *s, size_t n) {
  size_t i = 0;
  if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
  auto word = [&](const char *w) {
    size_t k = 0;
    for (; w[k]; ++k)
      if (i + k >= n || (s[i + k] | 0x20) != w[k]) return false;
    return i + k == n;
  };
  if (word("inf") || word("infinity") || word("nan")) return true;
  bool digits = false;
  while (i < n && s[i] >= '0' && s[i] <= '9') { digits = true; ++i; }
  if (i < n && s[i] == '.') {
    ++i;
    while (i < n && s[i] >= '0' && s[i] <= '9') { digits = true; ++i; }
  }
  if (!digits) return false;
  if (i < n && (s[i] == 'e' || s[i] == 'E')) {
    ++i;
    if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
    bool ed = false;
    while (i < n && s[i] >= '0' && s[i] <= '9') { ed = true; ++i; }
    if (!ed) return false;
  }
  return i == n;
}

inline bool number(Cursor &c, double &v) {
  if (!c.next() || c.len == 0 || c.len > 63 || !float_token(c.tok, c.len)) return false;
  char buf[64];
  memcpy(buf, c.tok, c.len);
  buf[c.len] = 0;
  char *e = nullptr;
  v = strtod(buf, &e);
  return e == buf + c.len;
}
inline bool integer(Cursor &c, int64_t &v) {
  if (!c.next() || c.len == 0 || c.len > 18) return false;
  v = 0;
  for (size_t i = 0; i < c.len; ++i) {
    if (c.tok[i] < '0' || c.tok[i] > '9') return false;
    v = v * 10 + (c.tok[i] - '0');
  }
  return true;
}
inline bool offset3(Cursor &c, double *dst) {  // "OFFSET" already consumed
  double v[3];
  for (int i = 0; i < 3; ++i)
    if (!number(c, v[i])) return false;
  if (dst) { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; }
  return true;
}

// c.tok is "ROOT" or "JOINT".  Returns 0 ok, -1 malformed, -2 capacity.
inline int joint(Cursor &c, Header &h, int parent, int depth) {
  if (depth > 256) return -1;
  if (!c.next()) return -1;
  size_t nl = 0;
  while (nl < c.len && ((c.tok[nl] >= 'a' && c.tok[nl] <= 'z') || (c.tok[nl] >= 'A' && c.tok[nl] <= 'Z') || (c.tok[nl] >= '0' && c.tok[nl] <= '9') || c.tok[nl] == '_')) ++nl;
  if (nl == 0) return -1;
  if (h.n >= h.max_joints || h.names_used + nl + 1 > h.names_cap) return -2;
  const int me = h.n++;
  memcpy(h.names + h.names_used, c.tok, nl);
  h.names[h.names_used + nl] = 0;
  h.names_used += nl + 1;
  h.parents[me] = parent;
  h.offsets[3 * me] = h.offsets[3 * me + 1] = h.offsets[3 * me + 2] = 0.0;
  h.channels[me] = 0;
  if (!c.next() || !c.is("{")) return -1;
  bool have_offset = false, have_channels = false;
  while (c.next()) {
    if (c.is("}")) return have_channels ? 0 : -1;
    if (c.is("OFFSET")) {
      if (!offset3(c, h.offsets + 3 * me)) return -1;
      have_offset = true;
    } else if (c.is("CHANNELS")) {
      int64_t k;
      if (have_channels || !integer(c, k) || k < 0 || k > 9) return -1;
      int axes[9];
      for (int i = 0; i < (int)k; ++i) {
        if (!c.next() || c.len < 6 || c.tok[0] < 'X' || c.tok[0] > 'Z') return -1;
        const bool rot = c.len == 9 && memcmp(c.tok + 1, "rotation", 8) == 0, pos = c.len == 9 && memcmp(c.tok + 1, "position", 8) == 0;
        const bool scl = c.len == 6 && memcmp(c.tok + 1, "scale", 5) == 0;
        if (!rot && !pos && !scl) return -1;
        axes[i] = rot ? c.tok[0] - 'X' : -1;
      }
      h.channels[me] = (int32_t)k;
      have_channels = true;
      if (h.order[0] < 0) {  // Euler order: first joint whose rotation slice (0-2 of a 3-channel joint, else 3-5) is all rotations
        const int lo = k == 3 ? 0 : 3;
        if (lo + 3 <= (int)k && axes[lo] >= 0 && axes[lo + 1] >= 0 && axes[lo + 2] >= 0)
          for (int i = 0; i < 3; ++i) h.order[i] = axes[lo + i];
      }
    } else if (c.is("JOINT")) {
      int rc = joint(c, h, me, depth + 1);
      if (rc) return rc;
    } else if (c.is("End")) {
      if (!c.next() || !c.is("Site") || !c.next() || !c.is("{") || !c.next() || !c.is("OFFSET") || !offset3(c, nullptr) || !c.next() || !c.is("}"))
        return -1;
    } else {
      return -1;
    }
  }
  (void)have_offset;
  return -1;  // ran out of text inside a joint
}

}  // namespace gmr_bvh
