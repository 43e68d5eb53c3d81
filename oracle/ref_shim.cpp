// ref_shim.cpp — extern "C" doors into the few REFERENCE sources that compile with plain g++.
// Built only where /root/reference exists (this container), output only into oracle/_ref/ (git-ignored,
// travels to the GPU box as a binary).  No reference source is copied: the headers are included
// from where they lie.  Used by tests/golden/make_golden.py to pin oracle/dbo.c and by the CSV parity test.
#include <climits>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "common/dpcpp/hashfunctions.hpp"      // Murmur3 / Polynomial / Simple hashers
#include "join/join_helpers/join_helpers.hpp"  // seq_join, row/col store, order-insensitive ==
#include "common/registry.hpp"
#include "common/result.hpp"

namespace {
// PolynomialHasher draws p from random_device in its ctor (hashfunctions.hpp:4-11) and keeps it
// private; the restatement needs a chosen p, so re-run the ctor until it lands on the wanted prime.
struct PolyProbe : PolynomialHasher {
  using PolynomialHasher::PolynomialHasher;
};
}  // namespace

extern "C" {

uint32_t ref_murmur3_x86_32(uint32_t key, uint32_t seed, uint64_t sz) {
  MurmurHash3_x86_32 h(sz, sizeof(uint32_t), seed);
  return static_cast<uint32_t>(h(key));
}

uint32_t ref_simple_hash(uint32_t key, uint64_t sz) {
  SimpleHasher<uint32_t> h(sz);
  return static_cast<uint32_t>(h(key));
}

// hash of `key` under PolynomialHasher(sz) for the instance whose hidden prime equals p.
// The prime is recovered from the public behaviour: hash(1) == p % sz for sz > 43.
uint32_t ref_polynomial_hash(uint32_t key, int p, uint64_t sz) {
  for (int tries = 0; tries < 100000; ++tries) {
    PolynomialHasher h(sz);
    bool match = true;
    // distinguish the 14 candidate primes through two probes that do not alias for sz > 43*43
    if (h(1) != static_cast<size_t>(p) % sz) match = false;
    if (match && h(10) != (static_cast<size_t>(p) * static_cast<size_t>(p)) % sz) match = false;
    if (match) return static_cast<uint32_t>(h(key));
  }
  return 0xFFFFFFFFu;
}

size_t ref_seq_join(const uint32_t *ak, const uint32_t *av, size_t na, const uint32_t *bk,
                    const uint32_t *bv, size_t nb, uint32_t *ok, uint32_t *o1, uint32_t *o2) {
  std::vector<uint32_t> a_k(ak, ak + na), a_v(av, av + na), b_k(bk, bk + nb), b_v(bv, bv + nb);
  auto r = join_helpers::seq_join(a_k, a_v, b_k, b_v);
  const size_t n = r.first.size();
  if (ok) {
    std::memcpy(ok, r.first.data(), n * sizeof(uint32_t));
    std::memcpy(o1, r.second.first.data(), n * sizeof(uint32_t));
    std::memcpy(o2, r.second.second.data(), n * sizeof(uint32_t));
  }
  return n;
}

// order-insensitive table equality exactly as join/join.cpp:133 uses it
int ref_joined_tables_equal(const uint32_t *k1, const uint32_t *a1, const uint32_t *b1, size_t n1,
                            const uint32_t *k2, const uint32_t *a2, const uint32_t *b2, size_t n2) {
  using namespace join_helpers;
  ColJoinedTableTy<uint32_t, uint32_t, uint32_t> t1 = {
      std::vector<uint32_t>(k1, k1 + n1),
      {std::vector<uint32_t>(a1, a1 + n1), std::vector<uint32_t>(b1, b1 + n1)}};
  ColJoinedTableTy<uint32_t, uint32_t, uint32_t> t2 = {
      std::vector<uint32_t>(k2, k2 + n2),
      {std::vector<uint32_t>(a2, a2 + n2), std::vector<uint32_t>(b2, b2 + n2)}};
  return t1 == t2 ? 1 : 0;
}

// Reference MeasureResults::write_csv (common/result.cpp:59-91) and Result printing (:9-40) on a
// caller-described list of runs: kind 0 = Result, 1 = HashJoinResult, 2 = GroupByAggResult.
// times_us: 4 doubles per run {host, kernel, t2, t3} (t2/t3 = build/probe or group_by/reduction).
int ref_write_csv(const char *path, const char *dwarf_name, const char *device_type,
                  const char *header, const int *kinds, const uint64_t *buf_sizes,
                  const double *times_us, size_t runs, char *printed, size_t printed_len) {
  MeasureResults res(dwarf_name);
  std::ostringstream os;
  for (size_t i = 0; i < runs; ++i) {
    std::unique_ptr<Result> r;
    const double *t = times_us + 4 * i;
    if (kinds[i] == 1) {
      auto h = std::make_unique<HashJoinResult>();
      h->build_time = Duration(t[2]);
      h->probe_time = Duration(t[3]);
      r = std::move(h);
    } else if (kinds[i] == 2) {
      auto g = std::make_unique<GroupByAggResult>();
      g->group_by_time = Duration(t[2]);
      g->reduction_time = Duration(t[3]);
      r = std::move(g);
    } else {
      r = std::make_unique<Result>();
    }
    r->host_time = Duration(t[0]);
    r->kernel_time = Duration(t[1]);
    os << *r;
    res.add_result({{"device_type", device_type}, {"buf_size", std::to_string(buf_sizes[i])}},
                   std::move(r));
  }
  if (header && *header) res.set_report_header(header);
  try {
    res.write_csv(path);
  } catch (const std::exception &) {
    return -1;
  }
  if (printed && printed_len) {
    std::strncpy(printed, os.str().c_str(), printed_len - 1);
    printed[printed_len - 1] = 0;
  }
  return 0;
}

// RunOptions::DeviceType parsing / printing (common/options.cpp:3-33): returns to_string(parse(s))
int ref_device_type_roundtrip(const char *s, char *out, size_t len) {
  std::istringstream in(s);
  RunOptions::DeviceType dt;
  in >> dt;
  std::string r = to_string(dt);
  std::strncpy(out, r.c_str(), len - 1);
  out[len - 1] = 0;
  return static_cast<int>(dt);
}

}  // extern "C"
