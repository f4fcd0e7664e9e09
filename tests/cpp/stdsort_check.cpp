// kimera_vio_amd/csrc/kvfe_stdsort.inl (what select_kernel runs for BrownANMS) against the host's std::sort with
// the reference's comparator (anms/anms.h sort_pred: left.first > right.first) on tie-heavy inputs: the two must
// produce the same PERMUTATION, ties included.  Host only, no GPU: part of the `-m "not gpu"` suite.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
#define KVFE_HD
static long heap_fallbacks = 0;
#define KVFE_STDSORT_COUNT_HEAP heap_fallbacks
#include "../../kimera_vio_amd/csrc/kvfe_stdsort.inl"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 16);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 400;
  long checked = 0;
  for (int round = 0; round < rounds; round++) {
    int n;
    switch (round % 8) {
      case 0: n = (int)(rnd() % 40); break;         // below / around the insertion-sort threshold
      case 1: n = 16 + (int)(rnd() % 3); break;
      case 2: n = 1 + (int)(rnd() % 8192); break;
      default: n = 100 + (int)(rnd() % 3000); break;
    }
    const int kind = (round / 8) % 6;
    std::vector<std::pair<float, int>> ref(n);
    std::vector<BrownRI> mine(n);
    for (int i = 0; i < n; i++) {
      float r;
      switch (kind) {
        case 0: r = (float)(rnd() % 4); break;                 // almost everything ties
        case 1: r = (float)(rnd() % 64) * 0.5f; break;
        case 2: r = (float)i; break;                           // ascending = worst order for `>`
        case 3: r = (float)(n - i); break;                     // already sorted
        case 4: r = (float)((i * 7919) % 13); break;           // periodic (drives the depth limit / heap sort)
        default: r = (float)rnd() / 65536.0f; break;
      }
      if (i == 0) r = 3.402823466e+38f;
      ref[i] = std::make_pair(r, i);
      mine[i].r = r;
      mine[i].i = i;
    }
    if (kind == 4 && n > 64) {  // median-of-3 killer-ish pattern: organ pipe
      for (int i = 0; i < n; i++) {
        const float r = (float)(i < n / 2 ? i : n - i);
        ref[i].first = r;
        mine[i].r = r;
      }
    }
    std::sort(ref.begin(), ref.end(),
              [](const std::pair<float, int>& l, const std::pair<float, int>& r) { return l.first > r.first; });
    std::vector<int> stack(3 * 64);
    brown_std_sort(mine.data(), n, stack.data());
    for (int i = 0; i < n; i++) {
      if (ref[i].second != mine[i].i || ref[i].first != mine[i].r) {
        printf("MISMATCH round %d kind %d n %d at %d: std::sort (%g,%d) restated (%g,%d)\n", round, kind, n, i,
               ref[i].first, ref[i].second, mine[i].r, mine[i].i);
        return 1;
      }
    }
    checked += n;
  }
  if (rounds >= 400 && heap_fallbacks == 0) {
    printf("the depth-limit / heap-sort branch was never taken\n");
    return 1;
  }
  printf("OK rounds=%d elements=%ld heap_fallbacks=%ld\n", rounds, checked, heap_fallbacks);
  return 0;
}
