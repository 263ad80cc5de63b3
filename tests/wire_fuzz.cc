// Test infrastructure: feeds truncated and mutated copies of valid messages to the two readers of the host layer that take bytes from
// outside the process — the Substrait ExtendedExpression reader (host/substrait.cc) and the Arrow IPC stream reader (host/ipc.cc) —
// through their device-less entry points.  Built by tests/test_wire_fuzz.py with -fsanitize=address,undefined from the host layer's
// sources, so that an out-of-bounds read that happens not to crash is still a failure.
//   wire_fuzz <corpus> <substrait|ipc> <mutations per message> <seed>
// corpus: repeated { uint32 little-endian length, bytes }.  Prints "messages truncations mutations accepted" and exits 0.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/arrowhip_compute.h"

static uint64_t rng_state;
static uint64_t rnd() {   // splitmix64
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: wire_fuzz corpus substrait|ipc mutations seed\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("corpus"); return 2; }
  const bool ipc = strcmp(argv[2], "ipc") == 0;
  const long nmut = atol(argv[3]);
  rng_state = strtoull(argv[4], nullptr, 10);
  std::vector<std::vector<uint8_t>> msgs;
  for (;;) {
    uint32_t len;
    if (fread(&len, 4, 1, f) != 1) break;
    std::vector<uint8_t> m(len);
    if (len && fread(m.data(), 1, len, f) != len) { fprintf(stderr, "short corpus\n"); return 2; }
    msgs.push_back(std::move(m));
  }
  fclose(f);
  static char out[1 << 16];
  long ntrunc = 0, nmutated = 0, accepted = 0;
  auto feed = [&](const uint8_t* p, size_t n) {
    // an exact-size heap copy: ASan's red zone starts at byte n
    uint8_t* copy = (uint8_t*)malloc(n ? n : 1);
    if (n) memcpy(copy, p, n);
    const int rc = ipc ? ahc_ipc_inspect(copy, (int64_t)n, out, (int64_t)sizeof out) : ahc_substrait_inspect(copy, (int64_t)n, out, (int64_t)sizeof out);
    free(copy);
    if (rc == 0) accepted++;
  };
  for (auto& m : msgs) {
    const size_t step = m.size() > 4096 ? m.size() / 4096 : 1;
    for (size_t n = 0; n <= m.size(); n += step) { feed(m.data(), n); ntrunc++; }
    for (long k = 0; k < nmut; k++) {
      std::vector<uint8_t> x = m;
      const int edits = 1 + (int)(rnd() % 4);
      for (int e = 0; e < edits && !x.empty(); e++) {
        const size_t pos = (size_t)(rnd() % x.size());
        switch (rnd() % 5) {
          case 0: x[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;           // a bit flip
          case 1: x[pos] = (uint8_t)rnd(); break;                          // a random byte
          case 2: x[pos] = 0xFF; break;                                    // a varint that goes on
          case 3: x.erase(x.begin() + (long)pos); break;                   // a byte missing
          default: x.insert(x.begin() + (long)pos, (uint8_t)rnd()); break; // a byte too many
        }
      }
      feed(x.data(), x.size());
      nmutated++;
    }
  }
  // plain noise, and noise behind a plausible first tag
  for (long k = 0; k < nmut; k++) {
    std::vector<uint8_t> x((size_t)(rnd() % 200));
    for (auto& b : x) b = (uint8_t)rnd();
    if (!x.empty() && (k & 1)) x[0] = ipc ? 0xFF : 0x1A;
    feed(x.data(), x.size());
    nmutated++;
  }
  printf("%zu %ld %ld %ld\n", msgs.size(), ntrunc, nmutated, accepted);
  return 0;
}
