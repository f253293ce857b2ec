// CPU sanitizer harness for the host half of csrc/jpeg.hip (parser, Huffman tables, entropy decoder, host reconstruction).
// Built by tests/test_jpeg_cpu.py with `hipcc --offload-host-only -fsanitize=address,undefined` (no device code, no GPU):
// reads seed JPEGs, applies byte mutations / truncations / hostile DHT segments and runs the three host entry points on each.
// Any out-of-bounds access or UB aborts the process (non-zero exit); a clean run prints the number of streams tried.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "msocr.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 11);
}

static long intervals_runs = 0;
static int run_one(const std::vector<uint8_t>& d, long* accepted) {
  long* intervals = &intervals_runs;
  msocr_jpeg_info info;
  // copy into an exactly-sized heap block so that a read past the end is an ASan error
  uint8_t* buf = (uint8_t*)malloc(d.size() ? d.size() : 1);
  memcpy(buf, d.data(), d.size());
  const int rc = msocr_jpeg_parse_host(buf, (int64_t)d.size(), &info);
  if (rc == 0 && info.supported && info.coef_total > 0 && info.coef_total < (int64_t)1 << 24 && (int64_t)info.width * info.height < (1 << 22)) {
    std::vector<int16_t> coef((size_t)info.coef_total);
    const int serial_rc = msocr_jpeg_entropy_decode_host(buf, (int64_t)d.size(), &info, coef.data());
    {
      // the per-interval decoder (host twin of the device Huffman kernel): same verdict, same coefficients as the serial decoder
      std::vector<uint64_t> desc((size_t)(msocr_jpeg_scan_desc_bytes() + 7) / 8);
      const int64_t cap = info.coef_total / 64 + 1;
      std::vector<uint32_t> bounds((size_t)(2 * cap));
      const int64_t niv = msocr_jpeg_scan_prepare_host(buf, (int64_t)d.size(), &info, 0, desc.data(), bounds.data(), cap);
      if (niv > 0) {
        std::vector<int16_t> coef2((size_t)info.coef_total);
        int32_t status = 0;
        const int64_t page_base[2] = {0, 0};
        if (msocr_jpeg_entropy_decode_intervals_host(buf, desc.data(), 1, bounds.data(), page_base, coef2.data(), info.coef_total, &status) != 0) return 1;
        if ((status != 0) != (serial_rc != 0)) { fprintf(stderr, "verdicts differ: intervals %d serial %d\n", status, serial_rc); return 1; }
        if (serial_rc == 0 && memcmp(coef.data(), coef2.data(), coef.size() * 2) != 0) { fprintf(stderr, "coefficients differ\n"); return 1; }
        ++*intervals;
      }
    }
    if (serial_rc == 0) {
      std::vector<uint8_t> rgb((size_t)info.width * info.height * 3);
      if (msocr_jpeg_reconstruct_host(&info, coef.data(), rgb.data()) != 0) return 1;
      ++*accepted;
    }
  }
  free(buf);
  return 0;
}

int main(int argc, char** argv) {
  long tried = 0, accepted = 0;
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  for (int a = 2; a < argc; ++a) {
    FILE* f = fopen(argv[a], "rb");
    if (!f) return 2;
    std::vector<uint8_t> seed;
    uint8_t tmp[4096];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) seed.insert(seed.end(), tmp, tmp + n);
    fclose(f);
    if (run_one(seed, &accepted)) return 3;
    ++tried;
    // locate DHT segments: they are the interesting target (code-length counts and symbol lists)
    std::vector<size_t> dht;
    for (size_t i = 2; i + 4 < seed.size(); ++i)
      if (seed[i] == 0xFF && seed[i + 1] == 0xC4) dht.push_back(i);
    for (int r = 0; r < rounds; ++r) {
      std::vector<uint8_t> m = seed;
      const uint32_t kind = rnd() % 6;
      if (kind == 0 && !dht.empty()) {  // hostile code-length counts
        const size_t o = dht[rnd() % dht.size()] + 5;
        for (int l = 0; l < 16 && o + l < m.size(); ++l)
          if (rnd() % 3 == 0) m[o + l] = (uint8_t)(rnd() % 4 == 0 ? 255 : rnd() % 20);
      } else if (kind == 1) {  // truncation
        m.resize(rnd() % (m.size() + 1));
      } else if (kind == 2) {  // header byte flips (first 700 bytes: DQT / SOF / DHT / SOS)
        const int flips = 1 + rnd() % 4;
        for (int k = 0; k < flips; ++k) m[rnd() % (m.size() < 700 ? m.size() : 700)] = (uint8_t)rnd();
      } else if (kind == 3) {  // body byte flips
        const int flips = 1 + rnd() % 16;
        for (int k = 0; k < flips; ++k) m[rnd() % m.size()] = (uint8_t)rnd();
      } else if (kind == 4) {  // segment length fields
        for (size_t i = 2; i + 4 < m.size() && i < 700; ++i)
          if (m[i] == 0xFF && m[i + 1] >= 0xC0 && m[i + 1] != 0xFF && rnd() % 4 == 0) { m[i + 2] = (uint8_t)rnd(); m[i + 3] = (uint8_t)rnd(); }
      } else {  // inserted marker bytes in the entropy-coded data
        const int ins = 1 + rnd() % 4;
        for (int k = 0; k < ins; ++k) { const size_t p = rnd() % m.size(); m[p] = 0xFF; if (p + 1 < m.size()) m[p + 1] = (uint8_t)(0xD0 + rnd() % 16); }
      }
      if (run_one(m, &accepted)) return 3;
      ++tried;
    }
  }
  printf("jpeg_fuzz: %ld streams, %ld decoded to the end, %ld also through the per-interval decoder\n", tried, accepted, intervals_runs);
  return 0;
}
