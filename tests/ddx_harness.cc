// Host build of csrc/ah_ddsum.h for the CPU tests (tests/test_ddsum_host.py): g++ -O2 -ffp-contract=off -shared.
// ddx_sum: `lanes` interleaved accumulators merged in order; ddx_merge_result: an array of four-word accumulators merged in order.
#include "../arrow_go_amd/csrc/ah_ddsum.h"
#include <stddef.h>
#include <vector>
extern "C" double ddx_sum(const double* v, size_t n, int lanes, double* out4) {
  std::vector<ah_ddx> a((size_t)lanes);
  for (auto& x : a) ah_ddx_init(x);
  for (size_t i = 0; i < n; i++) ah_ddx_add(a[i % (size_t)lanes], v[i]);
  ah_ddx t;
  ah_ddx_init(t);
  for (auto& x : a) ah_ddx_merge(t, x);
  if (out4) { out4[0] = t.s; out4[1] = t.e; out4[2] = t.bs; out4[3] = t.be; }
  return ah_ddx_result(t);
}
extern "C" double ddx_merge_result(const double* parts4, size_t nparts) {
  ah_ddx t;
  ah_ddx_init(t);
  for (size_t i = 0; i < nparts; i++) {
    ah_ddx p = {parts4[4 * i], parts4[4 * i + 1], parts4[4 * i + 2], parts4[4 * i + 3]};
    ah_ddx_merge(t, p);
  }
  return ah_ddx_result(t);
}
