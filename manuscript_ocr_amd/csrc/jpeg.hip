// jpeg.hip — image ingest: baseline JPEG -> RGB u8 on the device.
//
// Replaces read_image's file decode (detectors/_east/utils.py:477-497: cv2.imread / PIL, both libjpeg-turbo with its default
// decompression settings: JDCT_ISLOW, fancy upsampling, JFIF YCbCr -> RGB) for the formats a scanner / camera page comes in:
// 8-bit baseline or extended-sequential Huffman JPEG, grayscale or YCbCr with 4:4:4, 4:2:2 (h2v1) or 4:2:0 (h2v2) sampling,
// one interleaved scan, restart markers.  Anything else (progressive, arithmetic, CMYK, 12-bit, multi-scan, h1v2) is reported
// as unsupported and the caller falls back to the host decoder.
//
// Split: the entropy-coded segment of a stream WITHOUT restart markers is one serial bit stream, so parsing + Huffman decoding run
// on the HOST (msocr_jpeg_parse_host / msocr_jpeg_entropy_decode_host -> quantised DCT coefficients, 2 bytes each).  A stream WITH
// a restart interval (DRI) is a sequence of independent, byte-aligned bit streams (DC predictors reset at every RSTn): the host only
// walks the markers (msocr_jpeg_scan_prepare_host: interval bounds + Huffman tables), the file bytes go to the device as they are
// and ONE THREAD PER INTERVAL decodes them there (msocr_jpeg_entropy_decode_device, round 4) — 0.3-0.6 MB of file bytes cross PCIe
// instead of 9.4 MB of coefficients per 2048 x 1536 page, and no host core decodes anything.  Everything per-pixel —
// dequantisation + inverse DCT, chroma upsampling, colour conversion — runs on the DEVICE (msocr_jpeg_reconstruct), so the
// page's pixels are produced in HBM and never cross PCIe.  The reconstruction arithmetic is libjpeg's, restated from its
// published algorithms (jidctint.c "islow" 13-bit fixed point, jdsample.c triangle-filter upsampling, jdcolor.c 16-bit YCC
// tables) in __host__ __device__ functions: msocr_jpeg_reconstruct_host runs the same code on the CPU, which is how the CPU
// test-suite pins it bit for bit against PIL's decode of the same files.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "msocr.h"

#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define HD __host__ __device__ __forceinline__

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---------------------------------------------------------------------------------------------------- host: parse + Huffman
struct HuffTable {
  bool present = false;
  uint8_t bits[17] = {0}, vals[256] = {0};
  // fast path: 9-bit lookahead -> (length << 8) | symbol, 0 = longer code
  uint16_t look[512];
  int32_t maxcode[18];  // largest code of each length (-1 none), [17] = sentinel
  int32_t valoff[17];
  // false: the code lengths do not form a prefix code (libjpeg jdhuff.c jpeg_make_d_derived_tbl -> JERR_BAD_HUFF_TABLE)
  bool build() {
    int code = 0, k = 0;
    int32_t huffcode[256];
    uint8_t huffsize[256];
    for (int l = 1; l <= 16; ++l)
      for (int i = 0; i < bits[l]; ++i) { huffsize[k] = (uint8_t)l; ++k; }
    const int n = k;
    k = 0;
    int si = n ? huffsize[0] : 0;
    while (k < n) {
      while (k < n && huffsize[k] == si) huffcode[k++] = code++;
      if (code > (1 << si)) return false;  // over-subscribed at this length: more codes than si bits can hold
      code <<= 1;
      ++si;
    }
    int p = 0;
    for (int l = 1; l <= 16; ++l) {
      if (bits[l]) {
        valoff[l] = p - huffcode[p];
        p += bits[l];
        maxcode[l] = huffcode[p - 1];
      } else {
        maxcode[l] = -1;
        valoff[l] = 0;
      }
    }
    maxcode[17] = 0x7fffffff;
    memset(look, 0, sizeof(look));
    p = 0;
    for (int l = 1; l <= 9; ++l)
      for (int i = 0; i < bits[l]; ++i, ++p) {
        const int base = huffcode[p] << (9 - l);
        if (base + (1 << (9 - l)) > 512) return false;  // unreachable after the check above; keeps the table write in bounds
        for (int c = 0; c < (1 << (9 - l)); ++c) look[base + c] = (uint16_t)((l << 8) | vals[p]);
      }
    return true;
  }
};

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t acc = 0;
  int nbits = 0;
  bool hit_marker = false;
  void fill() {
    while (nbits <= 56) {
      int b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) p += 2;     // stuffed byte
          else { hit_marker = true; b = 0; }           // a marker: feed zeros (libjpeg does the same past the end of a segment)
        } else {
          ++p;
        }
      }
      acc = (acc << 8) | (uint64_t)b;
      nbits += 8;
    }
  }
  inline int peek(int n) { if (nbits < n) fill(); return (int)((acc >> (nbits - n)) & ((1u << n) - 1)); }
  inline void skip(int n) { nbits -= n; }
  inline int get(int n) { if (n == 0) return 0; const int v = peek(n); skip(n); return v; }
  void restart() { acc = 0; nbits = 0; hit_marker = false; }
};

inline int huff_decode(BitReader& br, const HuffTable& t) {
  const int look = br.peek(9);
  const uint16_t e = t.look[look];
  if (e) { br.skip(e >> 8); return e & 0xff; }
  int code = look, l = 9;
  for (;;) {  // codes longer than 9 bits
    ++l;
    if (l > 16) return -1;
    code = br.peek(l);
    if (code <= t.maxcode[l] && t.maxcode[l] >= 0) break;
  }
  br.skip(l);
  return t.vals[(code + t.valoff[l]) & 0xff];
}
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }  // HUFF_EXTEND

struct Parsed {
  msocr_jpeg_info info;
  HuffTable dc[4], ac[4];
  int dc_sel[3], ac_sel[3];
  int restart_interval = 0;
  const uint8_t* scan = nullptr;  // first byte of the entropy-coded segment
  int mcus_x = 0, mcus_y = 0;
};

inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// Largest frame taken (pixels): PIL refuses anything above 2 x MAX_IMAGE_PIXELS (DecompressionBombError), so the host
// decoder the caller falls back to gives the same answer; below it the coefficient array is at most 0.54 GB.
constexpr int64_t kMaxPixels = 2 * (int64_t)89478485;

// Exif Orientation (tag 0x0112 of IFD0) of an APP1 payload, 1 when absent / unreadable.
int exif_orientation(const uint8_t* s, int n) {
  if (n < 14 || memcmp(s, "Exif\0\0", 6) != 0) return 1;
  const uint8_t* t = s + 6;
  const int tn = n - 6;
  bool le;
  if (t[0] == 'I' && t[1] == 'I') le = true;
  else if (t[0] == 'M' && t[1] == 'M') le = false;
  else return 1;
  auto u16 = [&](int o) { return le ? (t[o] | (t[o + 1] << 8)) : ((t[o] << 8) | t[o + 1]); };
  auto u32 = [&](int o) {
    return le ? ((uint32_t)t[o] | ((uint32_t)t[o + 1] << 8) | ((uint32_t)t[o + 2] << 16) | ((uint32_t)t[o + 3] << 24))
              : (((uint32_t)t[o] << 24) | ((uint32_t)t[o + 1] << 16) | ((uint32_t)t[o + 2] << 8) | (uint32_t)t[o + 3]);
  };
  if (u16(2) != 42) return 1;
  const uint32_t ifd = u32(4);
  if (ifd > (uint32_t)tn || (int64_t)ifd + 2 > tn) return 1;
  const int cnt = u16((int)ifd);
  for (int e = 0; e < cnt; ++e) {
    const int64_t o = (int64_t)ifd + 2 + 12 * (int64_t)e;
    if (o + 12 > tn) return 1;
    if (u16((int)o) == 0x0112) return (u16((int)o + 2) == 3 && u32((int)o + 4) == 1) ? u16((int)o + 8) : 1;
  }
  return 1;
}

// Walks the markers up to the first SOS.  Returns MSOCR_OK, or MSOCR_E_ARG for a corrupt / unsupported stream.
int parse(const uint8_t* d, int64_t len, Parsed* P) {
  memset(&P->info, 0, sizeof(P->info));
  if (!d || len < 4 || d[0] != 0xFF || d[1] != 0xD8) return MSOCR_E_ARG;
  uint16_t qt[4][64];
  bool qt_present[4] = {false, false, false, false};
  int qsel[3] = {0, 0, 0}, comp_id[3] = {0, 0, 0};
  bool have_sof = false, adobe = false;
  int adobe_transform = -1;
  int64_t i = 2;
  while (i + 4 <= len) {
    if (d[i] != 0xFF) return MSOCR_E_ARG;
    while (i < len && d[i] == 0xFF) ++i;  // fill bytes
    if (i >= len) return MSOCR_E_ARG;
    const int m = d[i++];
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (m == 0xD9) return MSOCR_E_ARG;
    if (i + 2 > len) return MSOCR_E_ARG;
    const int L = rd16(d + i);
    if (L < 2 || i + L > len) return MSOCR_E_ARG;
    const uint8_t* s = d + i + 2;
    const int n = L - 2;
    if (m == 0xDB) {  // DQT
      int o = 0;
      while (o < n) {
        const int pq = s[o] >> 4, tq = s[o] & 15;
        ++o;
        if (tq > 3 || pq > 1 || o + 64 * (pq + 1) > n) return MSOCR_E_ARG;
        for (int k = 0; k < 64; ++k) {
          qt[tq][kZigzag[k]] = pq ? (uint16_t)rd16(s + o + 2 * k) : s[o + k];
        }
        o += 64 * (pq + 1);
        qt_present[tq] = true;
      }
    } else if (m == 0xC4) {  // DHT
      int o = 0;
      while (o < n) {
        if (o + 17 > n) return MSOCR_E_ARG;
        const int tc = s[o] >> 4, th = s[o] & 15;
        if (tc > 1 || th > 3) return MSOCR_E_ARG;
        HuffTable& t = tc ? P->ac[th] : P->dc[th];
        int cnt = 0;
        t.bits[0] = 0;
        for (int l = 1; l <= 16; ++l) { t.bits[l] = s[o + l]; cnt += t.bits[l]; }
        o += 17;
        if (cnt > 256 || o + cnt > n) return MSOCR_E_ARG;
        memcpy(t.vals, s + o, cnt);
        o += cnt;
        t.present = true;
        if (!t.build()) return MSOCR_E_ARG;
      }
    } else if (m == 0xC0 || m == 0xC1) {  // SOF0 / SOF1: baseline / extended sequential, Huffman
      if (n < 6 || have_sof) return MSOCR_E_ARG;
      if (s[0] != 8) return MSOCR_E_ARG;
      P->info.height = rd16(s + 1);
      P->info.width = rd16(s + 3);
      P->info.ncomp = s[5];
      if (P->info.height <= 0 || P->info.width <= 0 || (P->info.ncomp != 1 && P->info.ncomp != 3) || n < 6 + 3 * P->info.ncomp)
        return MSOCR_E_ARG;
      if ((int64_t)P->info.height * P->info.width > kMaxPixels) return MSOCR_E_ARG;
      for (int c = 0; c < P->info.ncomp; ++c) {
        comp_id[c] = s[6 + 3 * c];
        P->info.hs[c] = s[7 + 3 * c] >> 4;
        P->info.vs[c] = s[7 + 3 * c] & 15;
        qsel[c] = s[8 + 3 * c];
        if (qsel[c] > 3) return MSOCR_E_ARG;
      }
      have_sof = true;
    } else if ((m >= 0xC2 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return MSOCR_E_ARG;  // progressive / lossless / arithmetic / hierarchical: host decoder
    } else if (m == 0xDD) {
      if (n < 2) return MSOCR_E_ARG;
      P->restart_interval = rd16(s);
    } else if (m == 0xE1) {
      // cv2.imread (the reference's read_image) applies the Exif orientation; the device path does not rotate: such files
      // go to the host decoder, which does (detectors/_east/utils.py read_image)
      const int orient = exif_orientation(s, n);
      if (orient >= 2 && orient <= 8) return MSOCR_E_ARG;
    } else if (m == 0xEE && n >= 12 && memcmp(s, "Adobe", 5) == 0) {
      adobe = true;
      adobe_transform = s[11];
    } else if (m == 0xDA) {  // SOS
      if (!have_sof || n < 1) return MSOCR_E_ARG;
      const int ns = s[0];
      if (ns != P->info.ncomp || n < 1 + 2 * ns + 3) return MSOCR_E_ARG;  // one interleaved scan with every component
      for (int k = 0; k < ns; ++k) {
        int c = -1;
        for (int q = 0; q < P->info.ncomp; ++q)
          if (comp_id[q] == s[1 + 2 * k]) c = q;
        if (c != k) return MSOCR_E_ARG;  // scan order = frame order (what every baseline encoder writes)
        P->dc_sel[c] = s[2 + 2 * k] >> 4;
        P->ac_sel[c] = s[2 + 2 * k] & 15;
        if (P->dc_sel[c] > 3 || P->ac_sel[c] > 3 || !P->dc[P->dc_sel[c]].present || !P->ac[P->ac_sel[c]].present) return MSOCR_E_ARG;
      }
      if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return MSOCR_E_ARG;
      P->scan = d + i + L;
      break;
    }
    i += L;
  }
  if (!P->scan) return MSOCR_E_ARG;
  msocr_jpeg_info& f = P->info;
  if (f.ncomp == 3) {
    // colour space as libjpeg decides it (jdapimin.c default_decompress_parms): Adobe transform 0 = RGB / ids 'R','G','B' = RGB
    if (adobe && adobe_transform == 0) return MSOCR_E_ARG;
    if (!adobe && comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B') return MSOCR_E_ARG;
    if (f.hs[1] != 1 || f.vs[1] != 1 || f.hs[2] != 1 || f.vs[2] != 1) return MSOCR_E_ARG;
    if (!((f.hs[0] == 1 && f.vs[0] == 1) || (f.hs[0] == 2 && f.vs[0] == 1) || (f.hs[0] == 2 && f.vs[0] == 2))) return MSOCR_E_ARG;
  } else {
    f.hs[0] = f.vs[0] = 1;  // a single-component scan is non-interleaved: one block per MCU whatever the sampling factors say
  }
  const int hmax = f.hs[0], vmax = f.vs[0];
  P->mcus_x = (f.width + 8 * hmax - 1) / (8 * hmax);
  P->mcus_y = (f.height + 8 * vmax - 1) / (8 * vmax);
  int64_t off = 0;
  for (int c = 0; c < f.ncomp; ++c) {
    if (!qt_present[qsel[c]]) return MSOCR_E_ARG;
    for (int k = 0; k < 64; ++k) f.quant[c][k] = qt[qsel[c]][k];
    f.blocks_w[c] = P->mcus_x * f.hs[c];
    f.blocks_h[c] = P->mcus_y * f.vs[c];
    f.coef_off[c] = off;
    off += (int64_t)f.blocks_w[c] * f.blocks_h[c] * 64;
  }
  f.coef_total = off;
  f.supported = 1;
  return MSOCR_OK;
}

int entropy_decode(const Parsed& P, const uint8_t* end, int16_t* coef) {
  const msocr_jpeg_info& f = P.info;
  memset(coef, 0, sizeof(int16_t) * (size_t)f.coef_total);
  BitReader br;
  br.p = P.scan;
  br.end = end;
  int pred[3] = {0, 0, 0};
  int until_restart = P.restart_interval;
  for (int my = 0; my < P.mcus_y; ++my)
    for (int mx = 0; mx < P.mcus_x; ++mx) {
      if (P.restart_interval && until_restart == 0) {
        // byte-align, expect RSTn
        br.restart();
        const uint8_t* q = br.p;
        while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
        if (q + 1 >= end) return MSOCR_E_ARG;
        br.p = q + 2;
        pred[0] = pred[1] = pred[2] = 0;
        until_restart = P.restart_interval;
      }
      for (int c = 0; c < f.ncomp; ++c) {
        const HuffTable& dct = P.dc[P.dc_sel[c]];
        const HuffTable& act = P.ac[P.ac_sel[c]];
        for (int by = 0; by < f.vs[c]; ++by)
          for (int bx = 0; bx < f.hs[c]; ++bx) {
            int16_t* blk = coef + f.coef_off[c] + ((int64_t)(my * f.vs[c] + by) * f.blocks_w[c] + (mx * f.hs[c] + bx)) * 64;
            int s = huff_decode(br, dct);
            if (s < 0 || s > 15) return MSOCR_E_ARG;
            if (s) pred[c] = (int)((uint32_t)pred[c] + (uint32_t)extend(br.get(s), s));  // modulo 2^32: a hostile stream wraps
            blk[0] = (int16_t)pred[c];
            for (int k = 1; k < 64;) {
              const int rs = huff_decode(br, act);
              if (rs < 0) return MSOCR_E_ARG;
              const int r = rs >> 4, sz = rs & 15;
              if (sz == 0) {
                if (r != 15) break;  // EOB
                k += 16;
                continue;
              }
              k += r;
              if (k > 63) return MSOCR_E_ARG;
              blk[kZigzag[k]] = (int16_t)extend(br.get(sz), sz);
              ++k;
            }
          }
      }
      if (P.restart_interval) --until_restart;
    }
  return MSOCR_OK;
}

// ---------------------------------------------------------------------------------------------------- entropy decode per interval
// The decoder of ONE restart interval as a __host__ __device__ function: jpeg_huffman_kernel runs it one interval per thread,
// msocr_jpeg_entropy_decode_intervals_host runs the same code interval after interval on the CPU (how the CPU suite pins it against
// entropy_decode above and against PIL).  Same arithmetic and the same treatment of bad streams as entropy_decode: zeros are fed
// past the end of the interval (= at its marker), an undecodable code / a coefficient index past 63 is an error for the whole page.
struct DevTable {            // HuffTable without the construction state; 1420 bytes
  uint16_t look[512];
  int32_t maxcode[18];
  int32_t valoff[17];
  uint8_t vals[256];
};
struct ScanDesc {            // one page; msocr_jpeg_scan_desc_bytes() bytes, opaque to callers
  msocr_jpeg_info info;
  int64_t bytes_base;        // where the file's first byte sits in the batch byte buffer (the interval bounds are relative to the FILE)
  int32_t restart_interval, mcus_x, mcus_y, n_intervals;
  DevTable dc[3], ac[3];     // per component (duplicates when components share a table)
  uint8_t zigzag[64];
};

struct IntervalBits {
  const uint8_t* base;
  uint32_t pos, end;
  uint64_t acc;
  int nbits;
  // at least 33 valid bits afterwards (a code of up to 16 bits + up to 15 value bits per symbol)
  HD void refill() {
    if (nbits > 32) return;
    if (pos + 5 <= end) {
      // four stream bytes + one of lookahead, un-stuffed in registers: a lane that meets a 0xFF takes no memory round trips, so a
      // wave whose lanes are at different places of different streams pays the same few ALU instructions either way
      const uint64_t w = (uint64_t)base[pos] | ((uint64_t)base[pos + 1] << 8) | ((uint64_t)base[pos + 2] << 16) |
                         ((uint64_t)base[pos + 3] << 24) | ((uint64_t)base[pos + 4] << 32);
      const uint32_t lo = (uint32_t)w;
      if ((((~lo) - 0x01010101u) & lo & 0x80808080u) == 0) {  // no 0xFF among the four: no stuffing, no marker
        const uint32_t be = (lo << 24) | ((lo & 0xff00u) << 8) | ((lo >> 8) & 0xff00u) | (lo >> 24);
        acc = (acc << 32) | (uint64_t)be;
        nbits += 32;
        pos += 4;
        return;
      }
      uint32_t i = 0;
      bool marker = false;
#pragma unroll
      for (int step = 0; step < 4; ++step) {
        if (!marker && i < 4) {
          const uint32_t b = (uint32_t)(w >> (8 * i)) & 0xff, n = (uint32_t)(w >> (8 * i + 8)) & 0xff;
          if (b != 0xFF) { acc = (acc << 8) | b; nbits += 8; i += 1; }
          else if (n == 0) { acc = (acc << 8) | 0xFF; nbits += 8; i += 2; }   // stuffed byte
          else marker = true;                                                 // zeros from here on
        }
      }
      pos += i;
      if (marker) end = pos;
      if (nbits > 32) return;
    }
    while (nbits <= 56) {
      uint32_t b = 0;
      if (pos < end) {
        b = base[pos];
        if (b == 0xFF) {
          if (pos + 1 < end && base[pos + 1] == 0x00) pos += 2;   // stuffed byte
          else { b = 0; end = pos; }                              // a marker: zeros from here on
        } else {
          ++pos;
        }
      }
      acc = (acc << 8) | (uint64_t)b;
      nbits += 8;
    }
  }
  HD uint32_t peek(int n) const { return (uint32_t)(acc >> (nbits - n)) & ((1u << n) - 1u); }
  HD void skip(int n) { nbits -= n; }
};

HD int interval_symbol(IntervalBits& br, const DevTable& t) {
  const uint32_t look = br.peek(9);
  const uint32_t e = t.look[look];
  if (e) { br.skip((int)(e >> 8)); return (int)(e & 0xff); }
  int l = 9;
  int32_t code;
  for (;;) {
    ++l;
    if (l > 16) return -1;
    code = (int32_t)br.peek(l);
    if (t.maxcode[l] >= 0 && code <= t.maxcode[l]) break;
  }
  br.skip(l);
  return t.vals[(code + t.valoff[l]) & 0xff];
}

// One Huffman symbol per loop iteration, the same instruction sequence for a DC difference and an AC run/size: the 64 lanes of a
// wave decode 64 different intervals and stay converged except in the rare slow paths (codes longer than 9 bits, 0xFF bytes).
// `coef` = the page's zero-filled coefficient array.  Returns 0, or 1 for a bad stream.
HD int decode_interval(const msocr_jpeg_info& f, int mcus_x, const DevTable* dc, const DevTable* ac, const uint8_t* zigzag,
                       const uint8_t* bytes, uint32_t begin, uint32_t end, int first_mcu, int n_mcu, int16_t* coef) {
  IntervalBits br;
  br.base = bytes; br.pos = begin; br.end = end; br.acc = 0; br.nbits = 0;
  const int nb0 = f.hs[0] * f.vs[0];
  const int per_mcu = f.ncomp == 3 ? nb0 + 2 : 1;
  int pred0 = 0, pred1 = 0, pred2 = 0;
  int mcu = first_mcu, b = 0, k = 0, c = 0;
  int my = mcu / mcus_x, mx = mcu - my * mcus_x;
  int64_t blk = f.coef_off[0] + ((int64_t)(my * f.vs[0]) * f.blocks_w[0] + mx * f.hs[0]) * 64;
  const int last = first_mcu + n_mcu;
  while (mcu < last) {
    br.refill();
    const int sym = interval_symbol(br, k == 0 ? dc[c] : ac[c]);
    if (sym < 0) return 1;
    const int sz = k == 0 ? sym : (sym & 15);
    if (k == 0 && sz > 15) return 1;
    int v = 0;
    if (sz) {
      v = (int)br.peek(sz);
      br.skip(sz);
      v = v < (1 << (sz - 1)) ? v - (1 << sz) + 1 : v;   // HUFF_EXTEND
    }
    if (k == 0) {
      int pr = c == 0 ? pred0 : (c == 1 ? pred1 : pred2);
      pr = (int)((uint32_t)pr + (uint32_t)v);
      if (c == 0) pred0 = pr; else if (c == 1) pred1 = pr; else pred2 = pr;
      coef[blk] = (int16_t)pr;
      k = 1;
    } else {
      const int r = sym >> 4;
      if (sz == 0) {
        k = r == 15 ? k + 16 : 64;          // ZRL / EOB
      } else {
        k += r;
        if (k > 63) return 1;
        coef[blk + zigzag[k]] = (int16_t)v;
        ++k;
      }
    }
    if (k >= 64) {                          // next block of the MCU / next MCU
      k = 0;
      if (++b == per_mcu) {
        b = 0;
        ++mcu;
        if (++mx == mcus_x) { mx = 0; ++my; }
      }
      c = b < nb0 ? 0 : b - nb0 + 1;
      if (f.ncomp == 1) c = 0;
      const int bi = c == 0 ? b : 0;
      const int by = c == 0 ? bi / f.hs[0] : 0, bx = c == 0 ? bi - by * f.hs[0] : 0;
      blk = f.coef_off[c] + ((int64_t)(my * f.vs[c] + by) * f.blocks_w[c] + (mx * f.hs[c] + bx)) * 64;
    }
  }
  return 0;
}

void to_dev_table(const HuffTable& t, DevTable* o) {
  memcpy(o->look, t.look, sizeof(o->look));
  memcpy(o->maxcode, t.maxcode, sizeof(o->maxcode));
  memcpy(o->valoff, t.valoff, sizeof(o->valoff));
  memcpy(o->vals, t.vals, sizeof(o->vals));
}

// One wave = `lanes` consecutive intervals of ONE page (blockIdx.y); the page's six tables sit in LDS (all 64 threads load them).
// lanes < 64 when the batch has fewer intervals than the chip has wave slots: the decoder is a chain of dependent instructions, a
// lane costs the same whether its 63 neighbours work or not, and a wave executes the UNION of its lanes' slow paths (a 0xFF byte,
// a code longer than 9 bits) — so few intervals are spread one per wave over the 1024 SIMDs instead of packed into 32 waves
// (16 pages x 128 intervals: 22.7 ms packed).
__global__ __launch_bounds__(64) void jpeg_huffman_kernel(const uint8_t* __restrict__ bytes, const ScanDesc* __restrict__ descs,
                                                          const uint32_t* __restrict__ bounds, const int64_t* __restrict__ page_base,
                                                          int16_t* __restrict__ coef, int32_t* __restrict__ status, int lanes) {
  __shared__ DevTable s_dc[3], s_ac[3];
  __shared__ uint8_t s_zz[64];
  const ScanDesc& d = descs[blockIdx.y];
  if ((int)blockIdx.x * lanes >= d.n_intervals) return;    // uniform
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(d.dc);
    uint32_t* d0 = reinterpret_cast<uint32_t*>(s_dc);
    uint32_t* d1 = reinterpret_cast<uint32_t*>(s_ac);
    constexpr int W = (int)(3 * sizeof(DevTable) / 4);
    for (int i = threadIdx.x; i < W; i += 64) { d0[i] = src[i]; d1[i] = src[W + i]; }
    s_zz[threadIdx.x] = d.zigzag[threadIdx.x];
  }
  __syncthreads();
  const int iv = blockIdx.x * lanes + threadIdx.x;
  if ((int)threadIdx.x >= lanes || iv >= d.n_intervals) return;
  const int64_t coef_base = page_base[2 * blockIdx.y], first_interval = page_base[2 * blockIdx.y + 1];
  const uint32_t begin = bounds[2 * (first_interval + iv)], end = bounds[2 * (first_interval + iv) + 1];
  const int total = d.mcus_x * d.mcus_y;
  const int first = iv * d.restart_interval;
  const int n = total - first < d.restart_interval ? total - first : d.restart_interval;
  if (decode_interval(d.info, d.mcus_x, s_dc, s_ac, s_zz, bytes + d.bytes_base, begin, end, first, n, coef + coef_base)) status[blockIdx.y] = 1;
}

// ---------------------------------------------------------------------------------------------------- reconstruction
// jidctint.c, jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2
#define C_0_298631336 2446
#define C_0_390180644 3196
#define C_0_541196100 4433
#define C_0_765366865 6270
#define C_0_899976223 7373
#define C_1_175875602 9633
#define C_1_501321110 12299
#define C_1_847759065 15137
#define C_1_961570560 16069
#define C_2_053119869 16819
#define C_2_562915447 20995
#define C_3_072711026 25172

// All IDCT arithmetic is done modulo 2^32 on unsigned words (identical to libjpeg's signed arithmetic wherever that does not
// overflow, i.e. for every stream an encoder can produce; a hostile stream wraps instead of being undefined behaviour).
typedef uint32_t U32;
HD int descale(U32 x, int n) { return (int32_t)(x + (1u << (n - 1))) >> n; }
HD uint8_t idct_range_limit(int v) {  // sample_range_limit + CENTERJSAMPLE, indexed with (v & RANGE_MASK)
  const int x = v & 1023;
  return (uint8_t)(x < 128 ? x + 128 : (x < 512 ? 255 : (x < 896 ? 0 : x - 896)));
}

HD void idct_1d(U32 d0, U32 d1, U32 d2, U32 d3, U32 d4, U32 d5, U32 d6, U32 d7, int shift0, int o[8]) {
  // even part; d0/d4 are shifted up by CONST_BITS by the caller's convention: tmp0 = (d0 + d4) << CONST_BITS
  U32 z1 = (d2 + d6) * (U32)C_0_541196100;
  const U32 t2 = z1 + d6 * (U32)(-C_1_847759065);
  const U32 t3 = z1 + d2 * (U32)C_0_765366865;
  const U32 t0 = (d0 + d4) << 13;
  const U32 t1 = (d0 - d4) << 13;
  const U32 t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  // odd part
  U32 a0 = d7, a1 = d5, a2 = d3, a3 = d1;
  z1 = a0 + a3;
  U32 z2 = a1 + a2, z3 = a0 + a2, z4 = a1 + a3;
  const U32 z5 = (z3 + z4) * (U32)C_1_175875602;
  a0 *= (U32)C_0_298631336; a1 *= (U32)C_2_053119869; a2 *= (U32)C_3_072711026; a3 *= (U32)C_1_501321110;
  z1 *= (U32)(-C_0_899976223); z2 *= (U32)(-C_2_562915447); z3 *= (U32)(-C_1_961570560); z4 *= (U32)(-C_0_390180644);
  z3 += z5; z4 += z5;
  a0 += z1 + z3; a1 += z2 + z4; a2 += z2 + z3; a3 += z1 + z4;
  o[0] = descale(t10 + a3, shift0); o[7] = descale(t10 - a3, shift0);
  o[1] = descale(t11 + a2, shift0); o[6] = descale(t11 - a2, shift0);
  o[2] = descale(t12 + a1, shift0); o[5] = descale(t12 - a1, shift0);
  o[3] = descale(t13 + a0, shift0); o[4] = descale(t13 - a0, shift0);
}

// one 8x8 block: coefficients (natural order) x quantisation table -> 64 samples, row-major with `ld` bytes between rows
HD void idct_block(const int16_t* coef, const uint16_t* q, uint8_t* out, long ld) {
  int ws[64];
  auto dq = [&](int k) { return (U32)(int32_t)coef[k] * (U32)q[k]; };
  for (int c = 0; c < 8; ++c) {  // pass 1: columns (the all-AC-zero shortcut of the reference gives the same values)
    int o[8];
    idct_1d(dq(c), dq(8 + c), dq(16 + c), dq(24 + c), dq(32 + c), dq(40 + c), dq(48 + c), dq(56 + c), 13 - 2, o);
    for (int r = 0; r < 8; ++r) ws[r * 8 + c] = o[r];
  }
  for (int r = 0; r < 8; ++r) {  // pass 2: rows
    int o[8];
    const int* w = ws + r * 8;
    idct_1d((U32)w[0], (U32)w[1], (U32)w[2], (U32)w[3], (U32)w[4], (U32)w[5], (U32)w[6], (U32)w[7], 13 + 2 + 3, o);
    for (int c = 0; c < 8; ++c) out[r * ld + c] = idct_range_limit(o[c]);
  }
}

struct Planes {
  uint8_t* p[3];
  int ld[3];       // bytes per plane row (blocks_w * 8)
  int dsw[3], dsh[3];  // real (unpadded) extent of the component: ceil(image * samp / max)
};

// chroma sample at full-resolution position (X, Y): jdsample.c fullsize / h2v1 / h2v2, fancy (triangle) filters when the
// downsampled width is > 2, plain replication otherwise
HD int chroma_at(const uint8_t* pl, int ld, int dsw, int dsh, int hfac, int vfac, int X, int Y) {
  if (hfac == 1 && vfac == 1) return pl[(long)Y * ld + X];
  const int c = X >> 1;
  if (vfac == 1) {  // h2v1
    const uint8_t* row = pl + (long)Y * ld;
    if (dsw <= 2) return row[c];
    if (!(X & 1)) return c == 0 ? row[0] : (row[c] * 3 + row[c - 1] + 1) >> 2;
    return c == dsw - 1 ? row[c] : (row[c] * 3 + row[c + 1] + 2) >> 2;
  }
  const int r = Y >> 1;  // h2v2
  if (dsw <= 2) return pl[(long)r * ld + c];
  int rn = (Y & 1) ? r + 1 : r - 1;  // nearer neighbour row; the first / last real row is its own neighbour at the image edge
  rn = rn < 0 ? 0 : (rn > dsh - 1 ? dsh - 1 : rn);
  const uint8_t* r0 = pl + (long)r * ld;
  const uint8_t* r1 = pl + (long)rn * ld;
  const int cur = r0[c] * 3 + r1[c];
  if (!(X & 1)) {
    if (c == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + (r0[c - 1] * 3 + r1[c - 1]) + 8) >> 4;
  }
  if (c == dsw - 1) return (cur * 4 + 7) >> 4;
  return (cur * 3 + (r0[c + 1] * 3 + r1[c + 1]) + 7) >> 4;
}

HD uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// jdcolor.c ycc_rgb_convert: SCALEBITS 16 tables, evaluated directly
HD void ycc_to_rgb(int y, int cb, int cr, uint8_t* o) {
  const int xb = cb - 128, xr = cr - 128;
  const int cr_r = (91881 * xr + 32768) >> 16;
  const int cb_b = (116130 * xb + 32768) >> 16;
  const int g = (-22554 * xb + 32768 + (-46802) * xr) >> 16;
  o[0] = clamp255(y + cr_r);
  o[1] = clamp255(y + g);
  o[2] = clamp255(y + cb_b);
}

HD void pixel_rgb(const msocr_jpeg_info& f, const Planes& pl, int X, int Y, uint8_t* o) {
  const int y = pl.p[0][(long)Y * pl.ld[0] + X];
  if (f.ncomp == 1) {
    o[0] = o[1] = o[2] = (uint8_t)y;
    return;
  }
  const int cb = chroma_at(pl.p[1], pl.ld[1], pl.dsw[1], pl.dsh[1], f.hs[0], f.vs[0], X, Y);
  const int cr = chroma_at(pl.p[2], pl.ld[2], pl.dsw[2], pl.dsh[2], f.hs[0], f.vs[0], X, Y);
  ycc_to_rgb(y, cb, cr, o);
}

Planes make_planes(const msocr_jpeg_info& f, uint8_t* base) {
  Planes pl;
  int64_t off = 0;
  for (int c = 0; c < 3; ++c) {
    pl.p[c] = nullptr; pl.ld[c] = 0; pl.dsw[c] = 0; pl.dsh[c] = 0;
    if (c < f.ncomp) {
      pl.p[c] = base + off;
      pl.ld[c] = f.blocks_w[c] * 8;
      pl.dsw[c] = (f.width * f.hs[c] + f.hs[0] - 1) / f.hs[0];
      pl.dsh[c] = (f.height * f.vs[c] + f.vs[0] - 1) / f.vs[0];
      off += ((int64_t)f.blocks_w[c] * f.blocks_h[c] * 64 + 255) / 256 * 256;
    }
  }
  return pl;
}

int64_t planes_bytes(const msocr_jpeg_info& f) {
  int64_t off = 0;
  for (int c = 0; c < f.ncomp; ++c) off += ((int64_t)f.blocks_w[c] * f.blocks_h[c] * 64 + 255) / 256 * 256;
  return off;
}

bool info_ok(const msocr_jpeg_info* f) {
  if (!f || f->supported != 1 || (f->ncomp != 1 && f->ncomp != 3) || f->width <= 0 || f->height <= 0) return false;
  int64_t off = 0;
  for (int c = 0; c < f->ncomp; ++c) {
    if (f->hs[c] < 1 || f->hs[c] > 2 || f->vs[c] < 1 || f->vs[c] > 2 || f->blocks_w[c] <= 0 || f->blocks_h[c] <= 0) return false;
    if (f->coef_off[c] != off) return false;
    if ((int64_t)f->blocks_w[c] * 8 * f->hs[0] / f->hs[c] < f->width || (int64_t)f->blocks_h[c] * 8 * f->vs[0] / f->vs[c] < f->height) return false;
    off += (int64_t)f->blocks_w[c] * f->blocks_h[c] * 64;
  }
  return off == f->coef_total;
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(msocr_jpeg_info f, const int16_t* __restrict__ coef, Planes pl) {
  const long nb0 = (long)f.blocks_w[0] * f.blocks_h[0];
  const long nb1 = f.ncomp == 3 ? (long)f.blocks_w[1] * f.blocks_h[1] : 0;
  const long total = nb0 + 2 * nb1;
  for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < total; b += (long)gridDim.x * 256) {
    int c = 0;
    long k = b;
    if (k >= nb0) { k -= nb0; c = 1; if (k >= nb1) { k -= nb1; c = 2; } }
    const int by = (int)(k / f.blocks_w[c]), bx = (int)(k - (long)by * f.blocks_w[c]);
    idct_block(coef + f.coef_off[c] + k * 64, f.quant[c], pl.p[c] + ((long)by * 8) * pl.ld[c] + bx * 8, pl.ld[c]);
  }
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(msocr_jpeg_info f, Planes pl, uint8_t* __restrict__ rgb) {
  const long total = (long)f.width * f.height;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int Y = (int)(i / f.width), X = (int)(i - (long)Y * f.width);
    uint8_t o[3];
    pixel_rgb(f, pl, X, Y, o);
    rgb[3 * i] = o[0]; rgb[3 * i + 1] = o[1]; rgb[3 * i + 2] = o[2];
  }
}

}  // namespace

extern "C" int msocr_jpeg_parse_host(const uint8_t* data_host, int64_t len, msocr_jpeg_info* info_out) {
  if (!info_out) return MSOCR_E_ARG;
  Parsed P;
  const int rc = parse(data_host, len, &P);
  *info_out = P.info;
  if (rc != MSOCR_OK) info_out->supported = 0;
  return rc;
}

extern "C" int msocr_jpeg_entropy_decode_host(const uint8_t* data_host, int64_t len, const msocr_jpeg_info* info, int16_t* coef_out_host) {
  if (!coef_out_host || !info) return MSOCR_E_ARG;
  Parsed P;
  if (parse(data_host, len, &P) != MSOCR_OK) return MSOCR_E_ARG;
  if (P.info.width != info->width || P.info.height != info->height || P.info.ncomp != info->ncomp ||
      P.info.coef_total != info->coef_total) return MSOCR_E_ARG;  // `info` must be what parse returned for this stream
  return entropy_decode(P, data_host + len, coef_out_host);
}

extern "C" int64_t msocr_jpeg_scan_desc_bytes(void) { return (int64_t)sizeof(ScanDesc); }

// Walks the entropy-coded segment exactly as entropy_decode's restart handling does: interval k ends at the first marker (0xFF not
// followed by 0x00) at or after its start, interval k + 1 starts behind the first RSTn at or after that marker.
extern "C" int64_t msocr_jpeg_scan_prepare_host(const uint8_t* data_host, int64_t len, const msocr_jpeg_info* info, int64_t bytes_base,
                                                void* desc_out, uint32_t* bounds_out, int64_t bounds_cap) {
  if (!data_host || !info || !desc_out || !bounds_out || bytes_base < 0) return MSOCR_E_ARG;
  Parsed P;
  if (parse(data_host, len, &P) != MSOCR_OK) return MSOCR_E_ARG;
  if (P.info.width != info->width || P.info.height != info->height || P.info.ncomp != info->ncomp ||
      P.info.coef_total != info->coef_total) return MSOCR_E_ARG;
  if (P.restart_interval <= 0) return MSOCR_E_ARG;                       // one serial bit stream: the host decodes it
  if (len > 0xfffffff0LL) return MSOCR_E_ARG;                            // interval bounds are 32-bit offsets into the file
  const int64_t total = (int64_t)P.mcus_x * P.mcus_y;
  const int64_t n_iv = (total + P.restart_interval - 1) / P.restart_interval;
  if (n_iv > bounds_cap || n_iv > 0x7fffffff) return MSOCR_E_ARG;
  ScanDesc* d = static_cast<ScanDesc*>(desc_out);
  memset(d, 0, sizeof(*d));
  d->info = P.info;
  d->bytes_base = bytes_base;
  d->restart_interval = P.restart_interval;
  d->mcus_x = P.mcus_x; d->mcus_y = P.mcus_y; d->n_intervals = (int32_t)n_iv;
  for (int c = 0; c < P.info.ncomp; ++c) {
    to_dev_table(P.dc[P.dc_sel[c]], &d->dc[c]);
    to_dev_table(P.ac[P.ac_sel[c]], &d->ac[c]);
  }
  memcpy(d->zigzag, kZigzag, 64);
  const uint8_t* const end = data_host + len;
  const uint8_t* p = P.scan;
  for (int64_t k = 0; k < n_iv; ++k) {
    const uint8_t* e = p;
    for (;;) {                                                           // first 0xFF that is not a stuffed byte
      e = e < end ? static_cast<const uint8_t*>(memchr(e, 0xFF, (size_t)(end - e))) : nullptr;
      if (!e) { e = end; break; }
      if (e + 1 < end && e[1] == 0x00) { e += 2; continue; }
      break;
    }
    bounds_out[2 * k] = (uint32_t)(p - data_host);
    bounds_out[2 * k + 1] = (uint32_t)(e - data_host);
    if (k + 1 < n_iv) {
      const uint8_t* q = e;
      while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
      if (q + 1 >= end) return MSOCR_E_ARG;
      p = q + 2;
    }
  }
  return n_iv;
}

extern "C" int msocr_jpeg_entropy_decode_device(const uint8_t* bytes_dev, const void* descs_dev, int32_t n_pages, int32_t max_intervals,
                                                const uint32_t* bounds_dev, const int64_t* page_base_dev, int16_t* coef_dev,
                                                int64_t coef_total, int32_t* status_dev, void* stream) {
  if (!bytes_dev || !descs_dev || !bounds_dev || !page_base_dev || !coef_dev || !status_dev || n_pages <= 0 || n_pages > 65535 ||
      max_intervals <= 0 || coef_total <= 0 || (((uintptr_t)descs_dev | (uintptr_t)page_base_dev) & 7))
    return MSOCR_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(coef_dev, 0, (size_t)coef_total * sizeof(int16_t), s) != hipSuccess) return MSOCR_E_LAUNCH;
  if (hipMemsetAsync(status_dev, 0, (size_t)n_pages * sizeof(int32_t), s) != hipSuccess) return MSOCR_E_LAUNCH;
  // wave slots wanted: two per SIMD of the device; lanes per wave = the power of two that fills them
  static int slots = 0;
  static int lanes_env = -1;
  if (!slots) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MSOCR_E_LAUNCH;
    slots = prop.multiProcessorCount * 8;
    const char* e = getenv("MSOCR_JPEG_LANES");
    lanes_env = e ? atoi(e) : 0;
  }
  int lanes = 1;
  while (lanes < 64 && (int64_t)n_pages * max_intervals > (int64_t)slots * lanes) lanes *= 2;
  if (lanes_env >= 1 && lanes_env <= 64 && !(lanes_env & (lanes_env - 1))) lanes = lanes_env;
  MSOCR_LAUNCH(jpeg_huffman_kernel, dim3((unsigned)((max_intervals + lanes - 1) / lanes), (unsigned)n_pages), dim3(64), 0, s, bytes_dev,
               static_cast<const ScanDesc*>(descs_dev), bounds_dev, page_base_dev, coef_dev, status_dev, lanes);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

// HOST twin of jpeg_huffman_kernel: the same decode_interval, interval after interval (all pointers host memory).
extern "C" int msocr_jpeg_entropy_decode_intervals_host(const uint8_t* bytes_host, const void* descs_host, int32_t n_pages,
                                                        const uint32_t* bounds_host, const int64_t* page_base_host, int16_t* coef_host,
                                                        int64_t coef_total, int32_t* status_host) {
  if (!bytes_host || !descs_host || !bounds_host || !page_base_host || !coef_host || !status_host || n_pages <= 0 || coef_total <= 0)
    return MSOCR_E_ARG;
  memset(coef_host, 0, (size_t)coef_total * sizeof(int16_t));
  const ScanDesc* descs = static_cast<const ScanDesc*>(descs_host);
  for (int pg = 0; pg < n_pages; ++pg) {
    const ScanDesc& d = descs[pg];
    status_host[pg] = 0;
    const int64_t coef_base = page_base_host[2 * pg], first_interval = page_base_host[2 * pg + 1];
    if (coef_base < 0 || coef_base + d.info.coef_total > coef_total || first_interval < 0) return MSOCR_E_ARG;
    const int total = d.mcus_x * d.mcus_y;
    for (int iv = 0; iv < d.n_intervals; ++iv) {
      const int first = iv * d.restart_interval;
      const int n = total - first < d.restart_interval ? total - first : d.restart_interval;
      if (decode_interval(d.info, d.mcus_x, d.dc, d.ac, d.zigzag, bytes_host + d.bytes_base, bounds_host[2 * (first_interval + iv)],
                          bounds_host[2 * (first_interval + iv) + 1], first, n, coef_host + coef_base))
        status_host[pg] = 1;
    }
  }
  return MSOCR_OK;
}

extern "C" int64_t msocr_jpeg_workspace_bytes(const msocr_jpeg_info* info) { return info_ok(info) ? planes_bytes(*info) : -1; }

extern "C" int msocr_jpeg_reconstruct(const msocr_jpeg_info* info, const int16_t* coef_dev, void* workspace_dev, uint8_t* rgb_out_dev,
                                      void* stream) {
  if (!info_ok(info) || !coef_dev || !workspace_dev || !rgb_out_dev || ((uintptr_t)workspace_dev & 15)) return MSOCR_E_ARG;
  const Planes pl = make_planes(*info, (uint8_t*)workspace_dev);
  const long nblk = (long)info->coef_total / 64;
  long g1 = (nblk + 255) / 256;
  if (g1 > 65535) g1 = 65535;
  MSOCR_LAUNCH(jpeg_idct_kernel, dim3((unsigned)g1), dim3(256), 0, (hipStream_t)stream, *info, coef_dev, pl);
  if (hipGetLastError() != hipSuccess) return MSOCR_E_LAUNCH;
  long g2 = ((long)info->width * info->height + 255) / 256;
  if (g2 > 65535) g2 = 65535;
  MSOCR_LAUNCH(jpeg_color_kernel, dim3((unsigned)g2), dim3(256), 0, (hipStream_t)stream, *info, pl, rgb_out_dev);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

extern "C" int msocr_jpeg_reconstruct_host(const msocr_jpeg_info* info, const int16_t* coef_host, uint8_t* rgb_out_host) {
  if (!info_ok(info) || !coef_host || !rgb_out_host) return MSOCR_E_ARG;
  std::vector<uint8_t> ws((size_t)planes_bytes(*info));
  const Planes pl = make_planes(*info, ws.data());
  for (int c = 0; c < info->ncomp; ++c)
    for (int by = 0; by < info->blocks_h[c]; ++by)
      for (int bx = 0; bx < info->blocks_w[c]; ++bx)
        idct_block(coef_host + info->coef_off[c] + ((int64_t)by * info->blocks_w[c] + bx) * 64, info->quant[c],
                   pl.p[c] + ((long)by * 8) * pl.ld[c] + bx * 8, pl.ld[c]);
  for (int Y = 0; Y < info->height; ++Y)
    for (int X = 0; X < info->width; ++X) pixel_rgb(*info, pl, X, Y, rgb_out_host + 3 * ((int64_t)Y * info->width + X));
  return MSOCR_OK;
}
