// CPU check of the join's partition geometry (dwarf_bench_amd/csrc/join_common.hpp): every row count must give a
// geometry the kernels can run — at most 1024 level-0 buckets, a power-of-two level-1 fan-out of at most 1024,
// partitions that hold their expected rows, giant lists that fit the kernels' LDS list.  Built with hipcc (the header
// pulls in the HIP runtime header), runs without a GPU.
#include <cstdio>
#include <cstdlib>

#include "join_common.hpp"

using namespace dbhip;

static int bad = 0;
#define CHECK(c)                                                                                   \
  do {                                                                                             \
    if (!(c)) {                                                                                    \
      if (bad < 20) std::printf("FAILED %s at n = %zu rows_per_part = %zu\n", #c, n, rows);         \
      ++bad;                                                                                       \
    }                                                                                              \
  } while (0)

static void check(size_t n, size_t rows) {
  const JlLayout L = jl_layout(n, rows);
  CHECK(L.k1 >= 1 && L.k1 <= 1024);
  CHECK(L.k2 >= 1 && L.k2 <= 1024 && (L.k2 & (L.k2 - 1)) == 0 && L.k2 == (1u << L.log2_k2));
  CHECK(L.parts == L.k1 * L.k2 && L.parts <= (1u << 20) + 1024);
  CHECK(static_cast<size_t>(L.parts) * rows >= n || L.parts >= (1u << 20));  // a partition expects at most `rows` rows
  const size_t want = (n + rows - 1) / rows;
  CHECK(L.parts < 2 * (want ? want : 1) + 1024);                              // ... and not far fewer
  CHECK(L.k2 == 1 || L.parts > 1024);                                          // one level up to 1024 partitions
  CHECK(L.max_giants == jl_max_giants(n) && L.max_giants < kJlMaxGiantList);
  CHECK(L.total >= L.giant_off + jl_giant_bytes(L.max_giants) && L.giant_off >= L.meta_off + L.meta_bytes);
}

int main() {
  const size_t rows_options[2] = {kJlRowsPerPart, kJrRowsPerPart};
  for (size_t rows : rows_options) {
    for (size_t n = 0; n < 70000; n += 17) check(n, rows);
    for (unsigned lg = 10; lg <= 31; ++lg)
      for (long d = -3; d <= 3; ++d)
        for (size_t mul : {2u, 3u, 5u, 7u}) {
          const size_t base = (static_cast<size_t>(mul) << lg) / 2;
          const size_t n = base + d > kJlMaxRows ? kJlMaxRows : base + d;
          check(n, rows);
        }
    unsigned long long x = 88172645463325252ull;  // xorshift: arbitrary sizes up to 2^31
    for (int i = 0; i < 200000; ++i) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      check(static_cast<size_t>(x % (kJlMaxRows + 1)), rows);
    }
  }
  for (unsigned a = 10; a <= 31; ++a)
    for (unsigned b = 10; b <= 31; ++b) {
      const size_t n = static_cast<size_t>(1) << a, rows = static_cast<size_t>(1) << b;
      CHECK(jr_max_giants(n, rows) < kJlMaxGiantList);
    }
  std::printf(bad ? "join layout: %d violations\n" : "join layout ok\n", bad);
  return bad ? 1 : 0;
}
