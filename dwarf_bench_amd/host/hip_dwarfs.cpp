// hip_dwarfs.cpp — host side of the `...Hip` dwarfs (see hip_dwarfs.hpp).
//
// Every _run follows the reference's shape: generate the inputs once per size, then per iteration
// time the device work between steady_clock stamps (host_time: launch + sync, the reference's
// figure) and between hipEvents (kernel_time), validate, meter.add_result({{"buf_size", n}}, result).
// Differences, all deliberate:
//   * inputs are generated ON the device by the counter-based generators of libdbhip (deterministic
//     seeds instead of std::random_device, common/common.hpp:34-35) and stay resident: host_time does
//     not contain the reference's H2D/D2H of whole columns (scan/scan.cpp:108-120);
//   * validation is ALWAYS on and covers every size (the reference validates every iteration at every size,
//     scan/scan.cpp:157-164, but DPLScan/Radix/GroupBy/JoinOmnisci only in Debug builds): up to
//     DWARF_BENCH_VALIDATE_MAX elements (default 2^24) with the same host algorithms the reference dwarfs use
//     (std::copy_if, std::sort, the expected_GroupBy loop, per-key match counts); above it, where a host check
//     would dominate the run or not fit, with the device-side validators of libdbhip (dbhip_check_*: ordered
//     fingerprint of the compaction, sortedness + multiset fingerprint, weighted group sums, per-row match
//     counts against the sorted build column + id permutation).  DWARF_BENCH_INJECT_FAULT=1 corrupts one word
//     of every result before it is checked: every Result must then come out valid = false (tests use it to
//     show that the checks can fail);
//   * DWARF_BENCH_TIME_TRANSFERS=1 (scan dwarfs): host_time then contains the blocking H2D of src and the D2H
//     of all n output ints, exactly the reference's timed region (scan/scan.cpp:107-128), so the figure is
//     comparable with existing dwarf_bench CSVs;
//   * HIP failures throw DwarfBenchException (setup errors are exceptions in the reference too).
#include "hip_dwarfs.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <unordered_map>

#include "../../include/dbhip.h"
#include "bench.hpp"

namespace {

using clk = std::chrono::steady_clock;

[[noreturn]] void fail(const std::string &what) { throw DwarfBench::DwarfBenchException(what); }
void hip_ok(hipError_t e, const char *what) {
  if (e != hipSuccess) fail(std::string(what) + ": " + hipGetErrorString(e));
}
void db_ok(int rc, const char *what) {
  if (rc != 0) fail(std::string(what) + " failed with status " + std::to_string(rc));
}

// device buffer with the 256-byte alignment the C ABI asks for (hipMalloc gives more)
template <class T>
class DevBuf {
 public:
  explicit DevBuf(size_t n) : n_(n) { hip_ok(hipMalloc(&p_, std::max<size_t>(n, 1) * sizeof(T)), "hipMalloc"); }
  ~DevBuf() { (void)hipFree(p_); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  T *get() const { return static_cast<T *>(p_); }
  size_t size() const { return n_; }
  // Validation copies go through one pinned staging buffer: a hipMemcpy into pageable memory pins and
  // lazily unpins the destination, and that unpin lands inside the NEXT iteration's timed launch
  // (measured: 40 us -> 7-13 ms of host_time after a 64 MiB pageable D2H).
  std::vector<T> to_host(size_t count) const {
    std::vector<T> h(count);
    const size_t bytes = count * sizeof(T);
    if (bytes == 0) return h;
    if (bytes < (static_cast<size_t>(1) << 20)) {
      hip_ok(hipMemcpy(h.data(), p_, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
      return h;
    }
    struct Pinned {
      void *p = nullptr;
      size_t cap = 0;
      ~Pinned() { (void)hipHostFree(p); }
    };
    static Pinned stage;
    constexpr size_t kChunk = static_cast<size_t>(64) << 20;
    if (!stage.p) {
      hip_ok(hipHostMalloc(&stage.p, kChunk, hipHostMallocDefault), "hipHostMalloc");
      stage.cap = kChunk;
    }
    for (size_t off = 0; off < bytes; off += stage.cap) {
      const size_t len = std::min(stage.cap, bytes - off);
      hip_ok(hipMemcpy(stage.p, static_cast<const char *>(p_) + off, len, hipMemcpyDeviceToHost), "hipMemcpy D2H");
      std::memcpy(reinterpret_cast<char *>(h.data()) + off, stage.p, len);
    }
    return h;
  }

 private:
  void *p_ = nullptr;
  size_t n_;
};

struct Events {
  hipEvent_t a, b;
  Events() {
    hip_ok(hipEventCreate(&a), "hipEventCreate");
    hip_ok(hipEventCreate(&b), "hipEventCreate");
  }
  ~Events() {
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
  }
  Duration elapsed() const {
    float ms = 0;
    hip_ok(hipEventElapsedTime(&ms, a, b), "hipEventElapsedTime");
    return Duration(ms * 1000.0);
  }
};

size_t validate_limit() {
  static const size_t v = [] {
    const char *e = std::getenv("DWARF_BENCH_VALIDATE_MAX");
    return e ? static_cast<size_t>(std::strtoull(e, nullptr, 10)) : (static_cast<size_t>(1) << 24);
  }();
  return v;
}

bool env_flag(const char *name) {
  const char *e = std::getenv(name);
  return e && std::atoi(e) != 0;
}
bool inject_fault() {
  static const bool v = env_flag("DWARF_BENCH_INJECT_FAULT");
  return v;
}
bool time_transfers() {
  static const bool v = env_flag("DWARF_BENCH_TIME_TRANSFERS");
  return v;
}
// fault injection: flip bits of one device word (after the timed region, before the check)
void poke_xor(void *dev_word, uint32_t mask) {
  uint32_t h = 0;
  hip_ok(hipMemcpy(&h, dev_word, sizeof(h), hipMemcpyDeviceToHost), "poke D2H");
  h ^= mask;
  hip_ok(hipMemcpy(dev_word, &h, sizeof(h), hipMemcpyHostToDevice), "poke H2D");
}

// result words of a dbhip_check_* call
class CheckWords {
 public:
  CheckWords() : dev_(4) {}
  uint64_t *dev() const { return dev_.get(); }
  std::array<uint64_t, 4> get() const {
    std::array<uint64_t, 4> h{};
    hip_ok(hipMemcpy(h.data(), dev_.get(), sizeof(h), hipMemcpyDeviceToHost), "check D2H");  // syncs the null stream
    return h;
  }

 private:
  DevBuf<uint64_t> dev_;
};

void check_status(const void *ws, const char *what) {
  uint32_t st = 0xFFFFFFFFu;
  db_ok(dbhip_workspace_status(ws, &st, nullptr), "dbhip_workspace_status");
  if (st != DBHIP_DEV_OK) fail(std::string(what) + ": device status " + std::to_string(st));
}

void banner(const char *dwarf) {
  char name[64] = {0};
  int cus = 0, wave = 0;
  int dev = 0;
  hip_ok(hipGetDevice(&dev), "hipGetDevice");
  db_ok(dbhip_device_info(dev, name, sizeof(name), &cus, &wave), "dbhip_device_info");
  std::cout << "Selected device: " << name << " (" << cus << " CUs, wave" << wave << ") for " << dwarf << "\n";
}

void common_init(Dwarf &d, const RunOptions &opts) {
  d.meter().set_opts(opts);
  d.meter().set_params({{"device_type", to_string(opts.device_ty)}});
}

DwarfParams size_param(size_t n) { return DwarfParams{{"buf_size", std::to_string(n)}}; }

// ---- scan: shared by TwoPassScanHip and DPLScanHip -------------------------------------------------
void run_scan(const char *who, size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  banner(who);
  const int filter_value = 5;  // scan/scan.cpp:73, scan/dplscan.cpp:43
  DevBuf<int32_t> src(n), out(n);
  DevBuf<uint64_t> out_size(1);
  const size_t ws_bytes = dbhip_copy_if_lt_i32_workspace_bytes(n);
  DevBuf<unsigned char> ws(ws_bytes);
  db_ok(dbhip_gen_uniform_u32(reinterpret_cast<uint32_t *>(src.get()), n, 42, 0, 1, 10000, nullptr), "gen");
  hip_ok(hipDeviceSynchronize(), "sync");

  std::vector<int32_t> expected;
  const bool host_check = n <= validate_limit();
  CheckWords want, got;
  const size_t fp_bytes = dbhip_check_fingerprint_workspace_bytes(n);
  DevBuf<unsigned char> fp_ws(fp_bytes);
  std::array<uint64_t, 4> want_fp{};
  if (host_check) {  // scan/scan.cpp:12-17 expected_out_lt
    const std::vector<int32_t> host = src.to_host(n);
    std::copy_if(host.begin(), host.end(), std::back_inserter(expected),
                 [](int v) { return v < filter_value; });
  } else {  // order-sensitive fingerprint + length of the matching subsequence, straight from src
    db_ok(dbhip_check_fingerprint_lt_i32(src.get(), n, filter_value, want.dev(), fp_ws.get(), fp_bytes, nullptr),
          "dbhip_check_fingerprint_lt_i32");
    want_fp = want.get();
  }
  // DWARF_BENCH_TIME_TRANSFERS=1: the reference's timed region (scan/scan.cpp:107-128) — blocking write of src,
  // kernel, blocking read of all n output ints and of out_size — from/to pageable host vectors like the reference's
  std::vector<int32_t> host_src, host_out;
  if (time_transfers()) {
    host_src = src.to_host(n);
    host_out.resize(n);
  }
  Events ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    auto result = std::make_unique<Result>();
    const auto host_start = clk::now();
    if (time_transfers() && n)
      hip_ok(hipMemcpy(src.get(), host_src.data(), n * sizeof(int32_t), hipMemcpyHostToDevice), "src H2D");
    hip_ok(hipEventRecord(ev.a, nullptr), "event");
    db_ok(dbhip_copy_if_lt_i32(src.get(), n, filter_value, out.get(), out_size.get(), ws.get(), ws_bytes, nullptr),
          "dbhip_copy_if_lt_i32");
    hip_ok(hipEventRecord(ev.b, nullptr), "event");
    if (time_transfers() && n)
      hip_ok(hipMemcpy(host_out.data(), out.get(), n * sizeof(int32_t), hipMemcpyDeviceToHost), "out D2H");
    uint64_t count = 0;
    hip_ok(hipMemcpy(&count, out_size.get(), sizeof(count), hipMemcpyDeviceToHost), "out_size D2H");  // syncs
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->kernel_time = ev.elapsed();
    result->bytes = n * sizeof(int32_t) + count * sizeof(int32_t);
    check_status(ws.get(), who);
    if (inject_fault() && count) poke_xor(out.get() + count / 2, 1u);
    bool ok;
    if (host_check) {
      ok = count == expected.size() && out.to_host(count) == expected;
    } else {  // every element of out passes the filter iff the lengths agree; order and values: the fingerprint
      db_ok(dbhip_check_fingerprint_lt_i32(out.get(), count <= n ? count : n, filter_value, got.dev(), fp_ws.get(),
                                           fp_bytes, nullptr),
            "dbhip_check_fingerprint_lt_i32");
      const auto g = got.get();
      ok = count == want_fp[1] && g[1] == want_fp[1] && g[0] == want_fp[0];
    }
    if (!ok) {
      std::cerr << "incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result(size_param(n), std::move(result));
  }
}

}  // namespace

// =====================================================================================================
TwoPassScanHip::TwoPassScanHip() : Dwarf("TwoPassScanHip") {}
void TwoPassScanHip::_run(const size_t n, Meter &meter) { run_scan("TwoPassScanHip", n, meter); }
void TwoPassScanHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void TwoPassScanHip::init(const RunOptions &opts) { common_init(*this, opts); }

DPLScanHip::DPLScanHip() : Dwarf("DPLScanHip") {}
void DPLScanHip::_run(const size_t n, Meter &meter) { run_scan("DPLScanHip", n, meter); }
void DPLScanHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void DPLScanHip::init(const RunOptions &opts) { common_init(*this, opts); }

// =====================================================================================================
RadixHip::RadixHip() : Dwarf("RadixHip") {}
void RadixHip::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  banner("RadixHip");
  const int bits = [] {
    const char *e = std::getenv("DWARF_BENCH_RADIX_BITS");
    return (e && std::atoi(e) == 4) ? 4 : 8;
  }();
  DevBuf<int32_t> src(n), keys(n), tmp(n);
  const size_t ws_bytes = dbhip_radix_sort_workspace_bytes(n, bits);
  DevBuf<unsigned char> ws(ws_bytes);
  db_ok(dbhip_gen_uniform_u32(reinterpret_cast<uint32_t *>(src.get()), n, 42, 0, 1, 10000, nullptr), "gen");
  hip_ok(hipDeviceSynchronize(), "sync");
  const bool host_check = n <= validate_limit();
  std::vector<int32_t> expected;
  CheckWords chk;
  std::array<uint64_t, 4> want{};
  if (host_check) {  // sort/radix.cpp:8-12
    expected = src.to_host(n);
    std::sort(expected.begin(), expected.end());
  } else {  // multiset fingerprint of the unsorted column
    db_ok(dbhip_check_sorted_u32(reinterpret_cast<uint32_t *>(src.get()), n, 1, chk.dev(), nullptr), "dbhip_check_sorted_u32");
    want = chk.get();
  }
  Events ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    // every iteration sorts the unsorted column again (the reference re-wraps the const host vector,
    // sort/radix.cpp:31); the refresh copy is not timed
    hip_ok(hipMemcpy(keys.get(), src.get(), n * sizeof(int32_t), hipMemcpyDeviceToDevice), "refresh");
    auto result = std::make_unique<Result>();
    const auto host_start = clk::now();
    hip_ok(hipEventRecord(ev.a, nullptr), "event");
    db_ok(dbhip_radix_sort_i32(keys.get(), tmp.get(), n, bits, ws.get(), ws_bytes, nullptr), "dbhip_radix_sort_i32");
    hip_ok(hipEventRecord(ev.b, nullptr), "event");
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->kernel_time = ev.elapsed();
    if (n) check_status(ws.get(), "RadixHip");
    if (inject_fault() && n) poke_xor(keys.get() + n / 2, 0x100u);
    bool ok;
    if (host_check) {
      ok = keys.to_host(n) == expected;
    } else {  // ascending as int32 and the same multiset as the input
      db_ok(dbhip_check_sorted_u32(reinterpret_cast<uint32_t *>(keys.get()), n, 1, chk.dev(), nullptr), "dbhip_check_sorted_u32");
      const auto g = chk.get();
      ok = g[0] == 0 && g[1] == want[1] && g[2] == want[2];
    }
    if (!ok) {
      std::cerr << "incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result(size_param(n), std::move(result));
  }
}
void RadixHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void RadixHip::init(const RunOptions &opts) { common_init(*this, opts); }

// =====================================================================================================
GroupByHip::GroupByHip() : Dwarf("GroupByHip") {}
void GroupByHip::_run(const size_t n, Meter &meter) {
  // callers hand a GroupByRunOptions to GroupBy-family dwarfs (main.cpp:87-92, bench.cpp:80)
  const auto &opts = static_cast<const GroupByRunOptions &>(meter.opts());
  banner("GroupByHip");
  const uint32_t groups = static_cast<uint32_t>(opts.groups_count ? opts.groups_count : 1);
  DevBuf<uint32_t> keys(n), vals(n), out(groups);
  const size_t ws_bytes = dbhip_groupby_sum_u32_workspace_bytes(n, groups);
  DevBuf<unsigned char> ws(ws_bytes);
  db_ok(dbhip_gen_uniform_u32(vals.get(), n, 43, 0, 1, 10000, nullptr), "gen vals");        // groupby.cpp:29-30
  db_ok(dbhip_gen_uniform_u32(keys.get(), n, 42, 0, 0, groups - 1, nullptr), "gen keys");  // groupby.cpp:31-32
  hip_ok(hipDeviceSynchronize(), "sync");
  const bool host_check = n <= validate_limit();
  std::vector<uint32_t> expected(groups, 0);
  CheckWords chk;
  std::array<uint64_t, 4> want{};
  if (host_check) {  // groupby/groupby.cpp:8-19 expected_GroupBy with f = +
    const auto hk = keys.to_host(n);
    const auto hv = vals.to_host(n);
    for (size_t i = 0; i < n; ++i) expected[hk[i]] = expected[hk[i]] + hv[i];
  } else {  // sum of val * w(key) mod 2^32 for two weight functions, over the rows
    db_ok(dbhip_check_weighted_sum_u32(keys.get(), vals.get(), n, chk.dev(), nullptr), "dbhip_check_weighted_sum_u32");
    want = chk.get();
  }
  Events ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    auto result = std::make_unique<Result>();
    const auto host_start = clk::now();
    hip_ok(hipEventRecord(ev.a, nullptr), "event");
    db_ok(dbhip_groupby_sum_u32(keys.get(), vals.get(), n, groups, out.get(), ws.get(), ws_bytes, nullptr),
          "dbhip_groupby_sum_u32");
    hip_ok(hipEventRecord(ev.b, nullptr), "event");
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->kernel_time = ev.elapsed();
    check_status(ws.get(), "GroupByHip");
    if (inject_fault()) poke_xor(out.get() + groups / 2, 1u);
    bool ok;
    if (host_check) {
      ok = out.to_host(groups) == expected;
    } else {  // ... and over (g, out[g]): equal iff every row's value reached its own group (mod 2^32, as the sums)
      db_ok(dbhip_check_weighted_sum_u32(nullptr, out.get(), groups, chk.dev(), nullptr), "dbhip_check_weighted_sum_u32");
      const auto g = chk.get();
      ok = g[0] == want[0] && g[1] == want[1];
    }
    if (!ok) {
      std::cerr << "Incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result(size_param(n), std::move(result));
  }
}
void GroupByHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void GroupByHip::init(const RunOptions &opts) { common_init(*this, opts); }

// =====================================================================================================
JoinOmnisciHip::JoinOmnisciHip() : Dwarf("JoinOmnisciHip") {}
void JoinOmnisciHip::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  banner("JoinOmnisciHip");
  DevBuf<uint32_t> a(n), b(n), ids(n), pos(n), cnt(n);
  const size_t ws_bytes = dbhip_join_workspace_bytes(n);
  DevBuf<unsigned char> ws(ws_bytes);
  db_ok(dbhip_gen_uniform_u32(a.get(), n, 42, 0, 1, 10000, nullptr), "gen a");  // join_omnisci.cpp:53-58
  db_ok(dbhip_gen_uniform_u32(b.get(), n, 43, 0, 1, 10000, nullptr), "gen b");
  hip_ok(hipDeviceSynchronize(), "sync");
  const bool host_check = n <= validate_limit();
  std::vector<uint32_t> ha, hb;
  std::unordered_map<uint32_t, uint32_t> key_count;
  // device-side check above the host limit: the build column sorted (an algorithm that shares nothing with the
  // hash join) gives every probe key's exact multiplicity by binary search
  DevBuf<uint32_t> sorted_a(host_check ? 0 : n), sort_tmp(host_check ? 0 : n);
  const size_t perm_bytes = dbhip_check_permutation_workspace_bytes(n);
  DevBuf<unsigned char> perm_ws(host_check ? 0 : perm_bytes);
  CheckWords chk;
  if (host_check) {
    ha = a.to_host(n);
    hb = b.to_host(n);
    for (uint32_t k : ha) ++key_count[k];
  } else {
    const size_t sort_bytes = dbhip_radix_sort_workspace_bytes(n, 8);
    DevBuf<unsigned char> sort_ws(sort_bytes);
    hip_ok(hipMemcpy(sorted_a.get(), a.get(), n * sizeof(uint32_t), hipMemcpyDeviceToDevice), "copy");
    db_ok(dbhip_radix_sort_u32(sorted_a.get(), sort_tmp.get(), n, 8, sort_ws.get(), sort_bytes, nullptr), "sort build keys");
    hip_ok(hipDeviceSynchronize(), "sync");
  }
  Events build_ev, probe_ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    auto result = std::make_unique<HashJoinResult>();
    const auto host_start = clk::now();
    hip_ok(hipEventRecord(build_ev.a, nullptr), "event");
    db_ok(dbhip_join_build_u32(a.get(), n, ids.get(), ws.get(), ws_bytes, nullptr), "dbhip_join_build_u32");
    hip_ok(hipEventRecord(build_ev.b, nullptr), "event");
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    const auto build_end = clk::now();
    hip_ok(hipEventRecord(probe_ev.a, nullptr), "event");
    db_ok(dbhip_join_probe_u32(b.get(), n, ws.get(), n, pos.get(), cnt.get(), nullptr), "dbhip_join_probe_u32");
    hip_ok(hipEventRecord(probe_ev.b, nullptr), "event");
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->build_time = build_end - host_start;
    result->probe_time = host_end - build_end;
    result->kernel_time = build_ev.elapsed() + probe_ev.elapsed();
    check_status(ws.get(), "JoinOmnisciHip");
    if (inject_fault() && n) poke_xor(cnt.get() + n / 2, 1u);
    bool ok = true;
    if (host_check) {
      // join/join_omnisci.cpp:31-45 are_equal: size per probe row + every returned id really matches; on top of
      // it the id buffer must be a permutation of the build rows (a key's ids are then distinct rows)
      const auto hpos = pos.to_host(n), hcnt = cnt.to_host(n), hids = ids.to_host(n);
      std::vector<char> seen(n, 0);
      for (size_t j = 0; j < n && ok; ++j) {
        ok = hids[j] < n && !seen[hids[j]];
        if (ok) seen[hids[j]] = 1;
      }
      for (size_t i = 0; i < n && ok; ++i) {
        const auto f = key_count.find(hb[i]);
        const uint32_t want = f == key_count.end() ? 0u : f->second;
        ok = hcnt[i] == want && static_cast<size_t>(hpos[i]) + hcnt[i] <= n;
        // every id of a short list, a spread of 64 (always with both ends) of a long one
        const uint32_t step = hcnt[i] > 64 ? hcnt[i] / 64 : 1;
        for (uint32_t j = 0; ok && j < hcnt[i]; j += step) ok = ha[hids[hpos[i] + j]] == hb[i];
        if (ok && hcnt[i]) ok = ha[hids[hpos[i] + hcnt[i] - 1]] == hb[i];
      }
    } else {
      db_ok(dbhip_check_join_u32(sorted_a.get(), n, b.get(), n, pos.get(), cnt.get(), ids.get(), a.get(), 0, 0, 0,
                                 chk.dev(), nullptr),
            "dbhip_check_join_u32");
      const auto g = chk.get();
      db_ok(dbhip_check_permutation_u32(ids.get(), n, chk.dev(), perm_ws.get(), perm_bytes, nullptr),
            "dbhip_check_permutation_u32");
      ok = g[0] == 0 && chk.get()[0] == 0;
    }
    if (!ok) {
      std::cerr << "Incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result(size_param(n), std::move(result));
  }
}
void JoinOmnisciHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void JoinOmnisciHip::init(const RunOptions &opts) { common_init(*this, opts); }

// =====================================================================================================
JoinHip::JoinHip() : Dwarf("JoinHip") {}
void JoinHip::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  banner("JoinHip");
  if (10ull * n > 0xFFFFFFFFull) fail("JoinHip: keys are drawn from [0, 10*n) and must fit 32 bits");
  DevBuf<uint32_t> ak(n), av(n), bk(n), bv(n), ok_(n), o1(n), o2(n);
  const size_t ws_bytes = dbhip_ujoin_workspace_bytes(n);
  DevBuf<unsigned char> ws(ws_bytes);
  // unique, ascending keys in [0, 10n) like helpers::make_unique_random (common/common.cpp:7-20)
  db_ok(dbhip_gen_unique_sorted_u32(ak.get(), n, 11, 0, nullptr), "gen");
  db_ok(dbhip_gen_unique_sorted_u32(av.get(), n, 12, 0, nullptr), "gen");
  db_ok(dbhip_gen_unique_sorted_u32(bk.get(), n, 13, 0, nullptr), "gen");
  db_ok(dbhip_gen_unique_sorted_u32(bv.get(), n, 14, 0, nullptr), "gen");
  hip_ok(hipDeviceSynchronize(), "sync");
  const bool host_check = n <= validate_limit();
  std::unordered_map<uint32_t, uint32_t> a_payload;
  std::vector<uint32_t> hbk, hbv;
  CheckWords chk;
  if (host_check) {
    const auto hak = ak.to_host(n), hav = av.to_host(n);
    hbk = bk.to_host(n);
    hbv = bv.to_host(n);
    for (size_t i = 0; i < n; ++i) a_payload.emplace(hak[i], hav[i]);
  }
  Events build_ev, probe_ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    auto result = std::make_unique<HashJoinResult>();
    const auto host_start = clk::now();
    hip_ok(hipEventRecord(build_ev.a, nullptr), "event");
    db_ok(dbhip_ujoin_build_u32(ak.get(), av.get(), n, ws.get(), ws_bytes, nullptr), "dbhip_ujoin_build_u32");
    hip_ok(hipEventRecord(build_ev.b, nullptr), "event");
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    const auto build_end = clk::now();
    hip_ok(hipEventRecord(probe_ev.a, nullptr), "event");
    db_ok(dbhip_ujoin_probe_u32(bk.get(), bv.get(), n, ws.get(), n, ok_.get(), o1.get(), o2.get(), nullptr),
          "dbhip_ujoin_probe_u32");
    hip_ok(hipEventRecord(probe_ev.b, nullptr), "event");
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->build_time = build_end - host_start;
    result->probe_time = host_end - build_end;
    result->kernel_time = build_ev.elapsed() + probe_ev.elapsed();
    check_status(ws.get(), "JoinHip");
    if (inject_fault() && n) poke_xor(o1.get() + n / 2, 1u);
    bool ok = true;
    if (host_check) {
      // same table as seq_join would produce (join.cpp:27-28, :133): unique keys -> per probe row
      const auto hk = ok_.to_host(n), h1 = o1.to_host(n), h2 = o2.to_host(n);
      for (size_t i = 0; i < n && ok; ++i) {
        const auto f = a_payload.find(hbk[i]);
        if (f == a_payload.end())
          ok = hk[i] == 0xFFFFFFFFu && h1[i] == 0xFFFFFFFFu && h2[i] == 0xFFFFFFFFu;
        else
          ok = hk[i] == hbk[i] && h1[i] == f->second && h2[i] == hbv[i];
      }
    } else {  // the build keys are generated ascending and unique: binary search finds every probe row's partner
      db_ok(dbhip_check_ujoin_u32(ak.get(), av.get(), n, bk.get(), bv.get(), n, ok_.get(), o1.get(), o2.get(), chk.dev(),
                                  nullptr),
            "dbhip_check_ujoin_u32");
      ok = chk.get()[0] == 0;
    }
    if (!ok) {
      std::cerr << "Incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result(size_param(n), std::move(result));
  }
}
void JoinHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void JoinHip::init(const RunOptions &opts) { common_init(*this, opts); }

// =====================================================================================================
// GroupByLocalHip — the reference's privatised group-by as its own dwarf (groupby/groupby_local.cpp:24-142):
// GroupByAggResult with the two phases timed separately and the CSV header
// "total_time,group_by_time,reduction_time"; --executors caps the number of private tables.
GroupByLocalHip::GroupByLocalHip() : Dwarf("GroupByLocalHip") {}
void GroupByLocalHip::_run(const size_t n, Meter &meter) {
  const auto &opts = static_cast<const GroupByRunOptions &>(meter.opts());
  banner("GroupByLocalHip");
  const uint32_t groups = static_cast<uint32_t>(opts.groups_count ? opts.groups_count : 1);
  const uint32_t executors = static_cast<uint32_t>(opts.executors);
  DevBuf<uint32_t> keys(n), vals(n), out(groups);
  const size_t ws_bytes = dbhip_groupby_sum_u32_workspace_bytes(n, groups);
  DevBuf<unsigned char> ws(ws_bytes);
  db_ok(dbhip_gen_uniform_u32(vals.get(), n, 43, 0, 1, 10000, nullptr), "gen vals");
  db_ok(dbhip_gen_uniform_u32(keys.get(), n, 42, 0, 0, groups - 1, nullptr), "gen keys");
  hip_ok(hipDeviceSynchronize(), "sync");
  const bool host_check = n <= validate_limit();
  std::vector<uint32_t> expected(groups, 0);
  CheckWords chk;
  std::array<uint64_t, 4> want{};
  if (host_check) {
    const auto hk = keys.to_host(n);
    const auto hv = vals.to_host(n);
    for (size_t i = 0; i < n; ++i) expected[hk[i]] = expected[hk[i]] + hv[i];
  } else {  // sum of val * w(key) mod 2^32 for two weight functions, over the rows
    db_ok(dbhip_check_weighted_sum_u32(keys.get(), vals.get(), n, chk.dev(), nullptr), "dbhip_check_weighted_sum_u32");
    want = chk.get();
  }
  Events ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    auto result = std::make_unique<GroupByAggResult>();
    const auto host_start = clk::now();
    hip_ok(hipEventRecord(ev.a, nullptr), "event");
    db_ok(dbhip_groupby_partial_u32(keys.get(), vals.get(), n, groups, executors, ws.get(), ws_bytes, nullptr),
          "dbhip_groupby_partial_u32");
    hip_ok(hipStreamSynchronize(nullptr), "sync");  // the reference waits between the two kernels (:83, :112)
    const auto group_by_end = clk::now();
    db_ok(dbhip_groupby_merge_u32(groups, executors, out.get(), ws.get(), nullptr), "dbhip_groupby_merge_u32");
    hip_ok(hipEventRecord(ev.b, nullptr), "event");
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->group_by_time = group_by_end - host_start;
    result->reduction_time = host_end - group_by_end;
    result->kernel_time = ev.elapsed();
    check_status(ws.get(), "GroupByLocalHip");
    if (inject_fault()) poke_xor(out.get() + groups / 2, 1u);
    bool ok;
    if (host_check) {
      ok = out.to_host(groups) == expected;
    } else {  // ... and over (g, out[g]): equal iff every row's value reached its own group (mod 2^32, as the sums)
      db_ok(dbhip_check_weighted_sum_u32(nullptr, out.get(), groups, chk.dev(), nullptr), "dbhip_check_weighted_sum_u32");
      const auto g = chk.get();
      ok = g[0] == want[0] && g[1] == want[1];
    }
    if (!ok) {
      std::cerr << "Incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result(size_param(n), std::move(result));
  }
}
void GroupByLocalHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void GroupByLocalHip::init(const RunOptions &opts) {
  reporting_header_ = "total_time,group_by_time,reduction_time";  // groupby_local.cpp:138
  common_init(*this, opts);
}

// =====================================================================================================
// HashBuildHip — build-only timing of the bitmask-claimed table (hash/hash_build.cpp:8-98): every row
// inserts (key, key) into a table of 2n slots, Murmur3 hash; afterwards every key must be found.
HashBuildHip::HashBuildHip() : Dwarf("HashBuildHip") {}
void HashBuildHip::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  banner("HashBuildHip");
  const size_t ht_size = n ? n * 2 : 1;  // hash_build.cpp:19
  const uint32_t seed = 421;             // the reference draws it at random (helpers::make_random)
  DevBuf<uint32_t> src(n), found(n);
  const size_t ws_bytes = dbhip_bitmask_table_workspace_bytes(ht_size);
  DevBuf<unsigned char> ws(ws_bytes);
  db_ok(dbhip_gen_uniform_u32(src.get(), n, 42, 0, 1, 10000, nullptr), "gen");
  Events ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    db_ok(dbhip_bitmask_table_reset(ws.get(), ws_bytes, ht_size, nullptr), "reset");  // fresh table, untimed (:23-26)
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    auto result = std::make_unique<Result>();
    const auto host_start = clk::now();
    hip_ok(hipEventRecord(ev.a, nullptr), "event");
    db_ok(dbhip_bitmask_table_insert_u32(src.get(), src.get(), n, ws.get(), ws_bytes, ht_size, 1, seed, 0, nullptr),
          "dbhip_bitmask_table_insert_u32");
    hip_ok(hipEventRecord(ev.b, nullptr), "event");
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->kernel_time = ev.elapsed();
    check_status(ws.get(), "HashBuildHip");
    // hash_build.cpp:60-83: has(key) must be 1 for every inserted key
    db_ok(dbhip_bitmask_table_lookup_u32(src.get(), n, ws.get(), ht_size, 1, seed, nullptr, found.get(), nullptr),
          "dbhip_bitmask_table_lookup_u32");
    const auto h = found.to_host(n);
    if (!std::all_of(h.begin(), h.end(), [](uint32_t f) { return f == 1u; })) {
      std::cerr << "Incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result(size_param(n), std::move(result));
  }
}
void HashBuildHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void HashBuildHip::init(const RunOptions &opts) { common_init(*this, opts); }

// =====================================================================================================
// HashBuildNonBitmaskHip — build-only timing of the CAS-claimed table (hash/hash_build_non_bitmask.cpp:7-91):
// distinct keys claim slots with atomicCAS, duplicates land on the same slot; every key must be found.
HashBuildNonBitmaskHip::HashBuildNonBitmaskHip() : Dwarf("HashBuildNonBitmaskHip") {}
void HashBuildNonBitmaskHip::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  banner("HashBuildNonBitmaskHip");
  DevBuf<uint32_t> src(n), ok_(n), o1(n), o2(n);
  const size_t ws_bytes = dbhip_ujoin_workspace_bytes(n);
  DevBuf<unsigned char> ws(ws_bytes);
  db_ok(dbhip_gen_uniform_u32(src.get(), n, 42, 0, 1, 10000, nullptr), "gen");
  hip_ok(hipDeviceSynchronize(), "sync");
  Events ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    auto result = std::make_unique<Result>();
    const auto host_start = clk::now();
    hip_ok(hipEventRecord(ev.a, nullptr), "event");
    db_ok(dbhip_ujoin_build_u32(src.get(), src.get(), n, ws.get(), ws_bytes, nullptr), "dbhip_ujoin_build_u32");
    hip_ok(hipEventRecord(ev.b, nullptr), "event");
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->kernel_time = ev.elapsed();
    check_status(ws.get(), "HashBuildNonBitmaskHip");
    db_ok(dbhip_ujoin_probe_u32(src.get(), src.get(), n, ws.get(), n, ok_.get(), o1.get(), o2.get(), nullptr), "probe");
    const auto hk = ok_.to_host(n), hs = src.to_host(n);
    if (hk != hs) {  // every key found (a miss would leave the 0xFFFFFFFF sentinel)
      std::cerr << "Incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result(size_param(n), std::move(result));
  }
}
void HashBuildNonBitmaskHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void HashBuildNonBitmaskHip::init(const RunOptions &opts) { common_init(*this, opts); }

// =====================================================================================================
// ProbeHip — probe-only timing (probe/slab_probe.cpp:9-107: the table is built untimed, :38-62, the timed region is the
// lookup kernel alone, :64-95, over the SAME unique keys that were inserted, so every lookup must hit, :100-103).
// Table = the LDS-partitioned one-to-many table of dwarf 4a (DWARF_BENCH_PROBE_TABLE=bitmask: the bitmask-claimed
// SimpleNonOwningHashTable instead); keys from the make_unique_random twin.  Isolates the probe's share of the join:
// algorithmic bytes 4n (keys) + 8n (position, count).
ProbeHip::ProbeHip() : Dwarf("ProbeHip") {}
void ProbeHip::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  banner("ProbeHip");
  if (10ull * n > 0xFFFFFFFFull) fail("ProbeHip: keys are drawn from [0, 10*n) and must fit 32 bits");
  const char *table_env = std::getenv("DWARF_BENCH_PROBE_TABLE");
  const bool bitmask = table_env && std::string(table_env) == "bitmask";
  DevBuf<uint32_t> keys(n), ids(n), pos(n), cnt(n);
  db_ok(dbhip_gen_unique_sorted_u32(keys.get(), n, 11, 0, nullptr), "gen");  // slab_probe.cpp:17
  const size_t ht_size = n ? 2 * n : 1;
  const uint32_t seed = 421;
  const size_t ws_bytes = bitmask ? dbhip_bitmask_table_workspace_bytes(ht_size) : dbhip_join_workspace_bytes(n);
  DevBuf<unsigned char> ws(ws_bytes);
  CheckWords chk;
  Events ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    // build: untimed, a fresh table every iteration like the reference's AllocAdapter (:26-33)
    if (bitmask) {
      db_ok(dbhip_bitmask_table_reset(ws.get(), ws_bytes, ht_size, nullptr), "reset");
      db_ok(dbhip_bitmask_table_insert_u32(keys.get(), keys.get(), n, ws.get(), ws_bytes, ht_size, 1, seed, 0, nullptr),
            "dbhip_bitmask_table_insert_u32");
    } else {
      db_ok(dbhip_join_build_u32(keys.get(), n, ids.get(), ws.get(), ws_bytes, nullptr), "dbhip_join_build_u32");
    }
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    auto result = std::make_unique<Result>();
    const auto host_start = clk::now();
    hip_ok(hipEventRecord(ev.a, nullptr), "event");
    if (bitmask)
      db_ok(dbhip_bitmask_table_lookup_u32(keys.get(), n, ws.get(), ht_size, 1, seed, pos.get(), cnt.get(), nullptr),
            "dbhip_bitmask_table_lookup_u32");
    else
      db_ok(dbhip_join_probe_u32(keys.get(), n, ws.get(), n, pos.get(), cnt.get(), nullptr), "dbhip_join_probe_u32");
    hip_ok(hipEventRecord(ev.b, nullptr), "event");
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->kernel_time = ev.elapsed();
    result->bytes = 12 * n;
    if (n) check_status(ws.get(), "ProbeHip");
    if (inject_fault() && n) poke_xor(cnt.get() + n / 2, 1u);
    // slab_probe.cpp:36-37, :100-103: output == vector(n, 1) — every key found (once); the payload / the id the
    // probe leads to must be the key's own
    bool ok = true;
    if (n <= validate_limit()) {
      const auto hc = cnt.to_host(n), hp = pos.to_host(n);
      ok = std::all_of(hc.begin(), hc.end(), [](uint32_t c) { return c == 1u; });
      if (ok && bitmask) {
        ok = hp == keys.to_host(n);  // payload = key
      } else if (ok) {
        const auto hi = ids.to_host(n);
        for (size_t i = 0; i < n && ok; ++i) ok = hp[i] < n && hi[hp[i]] == i;
      }
    } else {  // sum of the counts == n with every count <= 1 is "all ones"; the join check covers both on the device
      if (bitmask) {
        db_ok(dbhip_check_sorted_u32(cnt.get(), n, 0, chk.dev(), nullptr), "dbhip_check_sorted_u32");
        const auto g = chk.get();
        ok = g[0] == 0 && g[2] == n;  // non-decreasing and summing to n over n entries that are 0 or 1
      } else {
        db_ok(dbhip_check_join_u32(keys.get(), n, keys.get(), n, pos.get(), cnt.get(), ids.get(), keys.get(), 0, 0, 0,
                                   chk.dev(), nullptr),
              "dbhip_check_join_u32");
        const auto g = chk.get();
        ok = g[0] == 0 && g[1] == n;
      }
    }
    if (!ok) {
      std::cerr << "Incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result(size_param(n), std::move(result));
  }
}
void ProbeHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void ProbeHip::init(const RunOptions &opts) { common_init(*this, opts); }

// =====================================================================================================
// ReduceHip — int sum of a column (reduce/reduce.cpp:27-98); expected = std::accumulate(..., 0) (:21).
ReduceHip::ReduceHip() : Dwarf("ReduceHip") {}
void ReduceHip::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  banner("ReduceHip");
  DevBuf<int32_t> src(n), out(1);
  db_ok(dbhip_gen_uniform_u32(reinterpret_cast<uint32_t *>(src.get()), n, 42, 0, 1, 10000, nullptr), "gen");
  hip_ok(hipDeviceSynchronize(), "sync");
  const bool host_check = n <= validate_limit();
  int32_t expected = 0;
  if (host_check) {
    const auto h = src.to_host(n);
    uint32_t acc = 0;  // accumulate with defined wrap-around; equals the int sum wherever that is defined
    for (int32_t v : h) acc += static_cast<uint32_t>(v);
    expected = static_cast<int32_t>(acc);
  } else {  // the 64-bit key sum of the sortedness check's kernel, truncated: another code path over the same column
    CheckWords chk;
    db_ok(dbhip_check_sorted_u32(reinterpret_cast<uint32_t *>(src.get()), n, 0, chk.dev(), nullptr), "dbhip_check_sorted_u32");
    expected = static_cast<int32_t>(static_cast<uint32_t>(chk.get()[2]));
  }
  Events ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    auto result = std::make_unique<Result>();
    const auto host_start = clk::now();
    hip_ok(hipEventRecord(ev.a, nullptr), "event");
    db_ok(dbhip_reduce_sum_i32(src.get(), n, out.get(), nullptr), "dbhip_reduce_sum_i32");
    hip_ok(hipEventRecord(ev.b, nullptr), "event");
    int32_t host_out = 0;
    hip_ok(hipMemcpy(&host_out, out.get(), sizeof(host_out), hipMemcpyDeviceToHost), "D2H");  // syncs
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->kernel_time = ev.elapsed();
    result->bytes = n * sizeof(int32_t);
    if (inject_fault()) host_out ^= 1;
    if (host_out != expected) {
      std::cerr << "Incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result(size_param(n), std::move(result));
  }
}
void ReduceHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void ReduceHip::init(const RunOptions &opts) { common_init(*this, opts); }

// =====================================================================================================
// NestedLoopJoinHip — join/nested_join.cpp:10-110: n x n cell matrix on the device, compacted on the host in
// cell order (:81-90), compared with the a-major/b-minor nested loop of join_helpers::seq_join.
NestedLoopJoinHip::NestedLoopJoinHip() : Dwarf("NestedLoopJoinHip") {}
void NestedLoopJoinHip::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  banner("NestedLoopJoinHip");
  if (n > (static_cast<size_t>(1) << 15)) fail("NestedLoopJoinHip: the n x n cell matrix is limited to n <= 32768");
  const size_t cells = n * n;
  DevBuf<uint32_t> ak(n), av(n), bk(n), bv(n), ok_(cells), o1(cells), o2(cells);
  db_ok(dbhip_gen_uniform_u32(ak.get(), n, 42, 0, 1, 10000, nullptr), "gen");
  db_ok(dbhip_gen_uniform_u32(av.get(), n, 43, 0, 1, 10000, nullptr), "gen");
  db_ok(dbhip_gen_uniform_u32(bk.get(), n, 44, 0, 1, 10000, nullptr), "gen");
  db_ok(dbhip_gen_uniform_u32(bv.get(), n, 45, 0, 1, 10000, nullptr), "gen");
  hip_ok(hipDeviceSynchronize(), "sync");
  const bool validate = cells <= validate_limit();
  using Row = std::array<uint32_t, 3>;
  std::vector<Row> expected;
  if (validate) {  // join_helpers.hpp:86-104
    const auto hak = ak.to_host(n), hav = av.to_host(n), hbk = bk.to_host(n), hbv = bv.to_host(n);
    for (size_t i = 0; i < n; ++i)
      for (size_t j = 0; j < n; ++j)
        if (hak[i] == hbk[j]) expected.push_back({hak[i], hav[i], hbv[j]});
  }
  Events ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    auto result = std::make_unique<Result>();
    const auto host_start = clk::now();
    hip_ok(hipEventRecord(ev.a, nullptr), "event");
    db_ok(dbhip_nested_join_u32(ak.get(), av.get(), bk.get(), bv.get(), n, n, ok_.get(), o1.get(), o2.get(), nullptr),
          "dbhip_nested_join_u32");
    hip_ok(hipEventRecord(ev.b, nullptr), "event");
    hip_ok(hipStreamSynchronize(nullptr), "sync");
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->kernel_time = ev.elapsed();
    result->bytes = 12 * cells;
    if (validate) {
      const auto hk = ok_.to_host(cells), h1 = o1.to_host(cells), h2 = o2.to_host(cells);
      std::vector<Row> got;
      for (size_t c = 0; c < cells; ++c)
        if (hk[c] != 0u) got.push_back({hk[c], h1[c], h2[c]});
      if (got != expected) {
        std::cerr << "Incorrect results" << std::endl;
        result->valid = false;
      }
    }
    meter.add_result(size_param(n), std::move(result));
  }
}
void NestedLoopJoinHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void NestedLoopJoinHip::init(const RunOptions &opts) { common_init(*this, opts); }

// =====================================================================================================
// PartitionedJoinHip — the radix-partitioned hash join of SURVEY 8(e) behind the Dwarf hook, one process driving
// `--gpus P` ranks (no reference counterpart; JoinOmnisci semantics, join/join_omnisci.cpp:49-118).
// `buf_size` rows per relation IN TOTAL; rank r owns the contiguous shard [r*n/P, (r+1)*n/P) of both key columns,
// generated in place on its GPU.  Per iteration, phase by phase (every phase ends with a sync of all ranks, like the
// reference's .wait() after every submit):
//   partition  dbhip_pjoin_partition_u32 on both shards -> bucket-major (key, global row id) pairs + counts
//   exchange   bucket d of every rank goes to rank d.  Ranks on distinct GPUs: ONE RCCL group of ncclSend/ncclRecv
//              for all four columns (every GPU talks to every peer at once: one xGMI link per pair, no ring).
//              More ranks than GPUs (rehearsal on one GPU, DWARF_BENCH_PJOIN_EXCHANGE=copy forces it): the same
//              transfers as hipMemcpyPeerAsync pushes.
//   build      dbhip_join_build_pairs_u32 on the received pairs: the id buffer holds GLOBAL build row ids
//   probe      dbhip_join_probe_u32; results stay sharded by key hash: (probe global row id, pos, cnt) + ids.
// The python path (dwarf_bench_amd/pjoin.py, one process per GPU, torch.distributed) pipelines these phases; here
// they are timed separately.  HashJoinResult: build_time = partition + exchange + build, probe_time = probe.
#include <rccl/rccl.h>

namespace {

void nccl_ok(ncclResult_t r, const char *what) {
  if (r != ncclSuccess) fail(std::string(what) + ": " + ncclGetErrorString(r));
}

struct DevMem {  // hipMalloc on a given device
  int dev = 0;
  void *p = nullptr;
  DevMem() = default;
  DevMem(int device, size_t bytes) : dev(device) {
    hip_ok(hipSetDevice(dev), "hipSetDevice");
    hip_ok(hipMalloc(&p, std::max<size_t>(bytes, 16)), "hipMalloc");
  }
  DevMem(DevMem &&o) noexcept : dev(o.dev), p(o.p) { o.p = nullptr; }
  DevMem &operator=(DevMem &&o) noexcept {
    release();
    dev = o.dev;
    p = o.p;
    o.p = nullptr;
    return *this;
  }
  DevMem(const DevMem &) = delete;
  DevMem &operator=(const DevMem &) = delete;
  ~DevMem() { release(); }
  void release() {
    if (p) {
      (void)hipSetDevice(dev);
      (void)hipFree(p);
      p = nullptr;
    }
  }
  template <class T>
  T *as() const { return static_cast<T *>(p); }
};

struct PjRank {
  int device = 0;
  hipStream_t stream = nullptr;
  size_t lo = 0, n_local = 0;
  DevMem build, probe, pk_r, pr_r, pk_s, pr_s, cnt_dev, part_ws, chk;
  size_t part_ws_bytes = 0;
  std::vector<uint64_t> send_r, send_s;  // rows for every destination rank
  size_t recv_r = 0, recv_s = 0;
  DevMem rk, rr, sk, sr, join_ws, ids, pos, cnt;
  size_t join_ws_bytes = 0;
};

template <class T>
std::vector<T> d2h(const void *src, size_t count, int dev) {
  std::vector<T> h(count);
  hip_ok(hipSetDevice(dev), "hipSetDevice");
  if (count) hip_ok(hipMemcpy(h.data(), src, count * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H");
  return h;
}

}  // namespace

PartitionedJoinHip::PartitionedJoinHip() : Dwarf("PartitionedJoinHip") {}
void PartitionedJoinHip::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  banner("PartitionedJoinHip");
  const unsigned P = static_cast<unsigned>(opts.devices ? opts.devices : 1);
  if (P > 256) fail("PartitionedJoinHip: at most 256 ranks");
  if (n > 0xFFFFFFFFull) fail("PartitionedJoinHip: global row ids must fit 32 bits");
  int ndev = 0;
  hip_ok(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
  const char *force = std::getenv("DWARF_BENCH_PJOIN_EXCHANGE");
  const bool use_rccl = static_cast<int>(P) <= ndev && !(force && std::string(force) == "copy");
  std::cout << "PartitionedJoinHip: " << P << " rank(s) on " << std::min<int>(P, ndev) << " GPU(s), exchange by "
            << (use_rccl ? "RCCL send/recv group" : "hipMemcpyPeerAsync") << "\n";
  int home = 0;
  hip_ok(hipGetDevice(&home), "hipGetDevice");

  std::vector<PjRank> ranks(P);
  const uint32_t key_hi = static_cast<uint32_t>(n ? n - 1 : 0);  // keys uniform in [0, n): SURVEY 8(d) join regime
  for (unsigned r = 0; r < P; ++r) {
    PjRank &k = ranks[r];
    k.device = static_cast<int>(r) % ndev;
    hip_ok(hipSetDevice(k.device), "hipSetDevice");
    hip_ok(hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking), "hipStreamCreate");
    const size_t per = n / P;
    k.lo = r * per;
    k.n_local = (r == P - 1) ? n - k.lo : per;
    const size_t col = k.n_local * sizeof(uint32_t);
    k.build = DevMem(k.device, col);
    k.probe = DevMem(k.device, col);
    k.pk_r = DevMem(k.device, col);
    k.pr_r = DevMem(k.device, col);
    k.pk_s = DevMem(k.device, col);
    k.pr_s = DevMem(k.device, col);
    k.cnt_dev = DevMem(k.device, 2 * P * sizeof(uint64_t));
    k.chk = DevMem(k.device, 8 * sizeof(int32_t));
    k.part_ws_bytes = dbhip_pjoin_partition_workspace_bytes(k.n_local, P);
    k.part_ws = DevMem(k.device, k.part_ws_bytes);
    db_ok(dbhip_gen_uniform_u32(k.build.as<uint32_t>(), k.n_local, 42, k.lo, 0, key_hi, k.stream), "gen build");
    db_ok(dbhip_gen_uniform_u32(k.probe.as<uint32_t>(), k.n_local, 43, k.lo, 0, key_hi, k.stream), "gen probe");
  }
  auto sync_all = [&] {
    for (PjRank &k : ranks) {
      hip_ok(hipSetDevice(k.device), "hipSetDevice");
      hip_ok(hipStreamSynchronize(k.stream), "hipStreamSynchronize");
    }
  };
  auto partition = [&] {
    for (PjRank &k : ranks) {
      hip_ok(hipSetDevice(k.device), "hipSetDevice");
      uint64_t *c = k.cnt_dev.as<uint64_t>();
      db_ok(dbhip_pjoin_partition_u32(k.build.as<uint32_t>(), k.n_local, k.lo, P, k.pk_r.as<uint32_t>(),
                                      k.pr_r.as<uint32_t>(), c, k.part_ws.p, k.part_ws_bytes, k.stream),
            "dbhip_pjoin_partition_u32");
      db_ok(dbhip_pjoin_partition_u32(k.probe.as<uint32_t>(), k.n_local, k.lo, P, k.pk_s.as<uint32_t>(),
                                      k.pr_s.as<uint32_t>(), c + P, k.part_ws.p, k.part_ws_bytes, k.stream),
            "dbhip_pjoin_partition_u32");
    }
    sync_all();
    for (PjRank &k : ranks) {  // the P x P count matrix, row = sender (an all-gather in a multi-process setting)
      const auto c = d2h<uint64_t>(k.cnt_dev.p, 2 * P, k.device);
      k.send_r.assign(c.begin(), c.begin() + P);
      k.send_s.assign(c.begin() + P, c.end());
    }
    for (unsigned r = 0; r < P; ++r) {
      ranks[r].recv_r = ranks[r].recv_s = 0;
      for (unsigned q = 0; q < P; ++q) {
        ranks[r].recv_r += ranks[q].send_r[r];
        ranks[r].recv_s += ranks[q].send_s[r];
      }
    }
  };

  // planning pass (untimed): learn the receive sizes, allocate receive buffers, tables and outputs once
  partition();
  for (PjRank &k : ranks) {
    k.rk = DevMem(k.device, k.recv_r * 4);
    k.rr = DevMem(k.device, k.recv_r * 4);
    k.sk = DevMem(k.device, k.recv_s * 4);
    k.sr = DevMem(k.device, k.recv_s * 4);
    k.join_ws_bytes = dbhip_join_workspace_bytes(k.recv_r);
    k.join_ws = DevMem(k.device, k.join_ws_bytes);
    k.ids = DevMem(k.device, k.recv_r * 4);
    k.pos = DevMem(k.device, k.recv_s * 4);
    k.cnt = DevMem(k.device, k.recv_s * 4);
  }
  std::vector<ncclComm_t> comms;
  if (use_rccl) {
    std::vector<int> devs(P);
    for (unsigned r = 0; r < P; ++r) devs[r] = ranks[r].device;
    comms.resize(P);
    nccl_ok(ncclCommInitAll(comms.data(), static_cast<int>(P), devs.data()), "ncclCommInitAll");
  } else {
    for (PjRank &a : ranks)
      for (PjRank &b : ranks)
        if (a.device != b.device) {
          hip_ok(hipSetDevice(a.device), "hipSetDevice");
          const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) hip_ok(e, "hipDeviceEnablePeerAccess");
          (void)hipGetLastError();
        }
  }
  auto exchange = [&] {
    // offsets: sender side = prefix over destinations, receiver side = prefix over senders
    std::vector<std::vector<uint64_t>> soff_r(P, std::vector<uint64_t>(P + 1, 0)), soff_s = soff_r, roff_r = soff_r,
                                       roff_s = soff_r;
    for (unsigned r = 0; r < P; ++r)
      for (unsigned q = 0; q < P; ++q) {
        soff_r[r][q + 1] = soff_r[r][q] + ranks[r].send_r[q];
        soff_s[r][q + 1] = soff_s[r][q] + ranks[r].send_s[q];
        roff_r[r][q + 1] = roff_r[r][q] + ranks[q].send_r[r];
        roff_s[r][q + 1] = roff_s[r][q] + ranks[q].send_s[r];
      }
    if (use_rccl) nccl_ok(ncclGroupStart(), "ncclGroupStart");
    for (unsigned r = 0; r < P; ++r) {
      PjRank &me = ranks[r];
      hip_ok(hipSetDevice(me.device), "hipSetDevice");
      for (unsigned q = 0; q < P; ++q) {
        PjRank &peer = ranks[q];
        struct Col {
          const uint32_t *src;
          uint32_t *my_recv, *peer_recv;
          uint64_t send_off, send_cnt, recv_off, recv_cnt, peer_recv_off;
        };
        const Col cols[4] = {
            {me.pk_r.as<uint32_t>(), me.rk.as<uint32_t>(), peer.rk.as<uint32_t>(), soff_r[r][q], me.send_r[q], roff_r[r][q],
             peer.send_r[r], roff_r[q][r]},
            {me.pr_r.as<uint32_t>(), me.rr.as<uint32_t>(), peer.rr.as<uint32_t>(), soff_r[r][q], me.send_r[q], roff_r[r][q],
             peer.send_r[r], roff_r[q][r]},
            {me.pk_s.as<uint32_t>(), me.sk.as<uint32_t>(), peer.sk.as<uint32_t>(), soff_s[r][q], me.send_s[q], roff_s[r][q],
             peer.send_s[r], roff_s[q][r]},
            {me.pr_s.as<uint32_t>(), me.sr.as<uint32_t>(), peer.sr.as<uint32_t>(), soff_s[r][q], me.send_s[q], roff_s[r][q],
             peer.send_s[r], roff_s[q][r]},
        };
        for (const Col &c : cols) {
          if (use_rccl) {
            // pieces of at most 2^28 elements (1 GiB): one ncclSend/ncclRecv of 2^29 uint32 (2 GiB) was measured to
            // deliver garbage without any error (RCCL 2.27.7); both sides cut their segment the same way, so the
            // k-th send to a peer still meets the k-th receive from it
            constexpr uint64_t kPiece = 1ull << 28;
            for (uint64_t o = 0; o < c.send_cnt; o += kPiece)
              nccl_ok(ncclSend(c.src + c.send_off + o, std::min(kPiece, c.send_cnt - o), ncclUint32, static_cast<int>(q),
                               comms[r], me.stream), "ncclSend");
            for (uint64_t o = 0; o < c.recv_cnt; o += kPiece)
              nccl_ok(ncclRecv(c.my_recv + c.recv_off + o, std::min(kPiece, c.recv_cnt - o), ncclUint32,
                               static_cast<int>(q), comms[r], me.stream), "ncclRecv");
          } else if (c.send_cnt) {
            hip_ok(hipMemcpyPeerAsync(c.peer_recv + c.peer_recv_off, peer.device, c.src + c.send_off, me.device,
                                      c.send_cnt * sizeof(uint32_t), me.stream),
                   "hipMemcpyPeerAsync");
          }
        }
      }
    }
    if (use_rccl) nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
    sync_all();
  };

  // host copy of the global columns for validation
  const bool validate = n <= validate_limit();
  std::vector<uint32_t> build_all, probe_all;
  std::unordered_map<uint32_t, uint32_t> key_count;
  if (validate) {
    sync_all();
    for (PjRank &k : ranks) {
      const auto b = d2h<uint32_t>(k.build.p, k.n_local, k.device), p = d2h<uint32_t>(k.probe.p, k.n_local, k.device);
      build_all.insert(build_all.end(), b.begin(), b.end());
      probe_all.insert(probe_all.end(), p.begin(), p.end());
    }
    for (uint32_t k : build_all) ++key_count[k];
  }

  for (size_t it = 0; it < opts.iterations; ++it) {
    auto result = std::make_unique<HashJoinResult>();
    sync_all();
    const auto t0 = clk::now();
    partition();
    const auto t1 = clk::now();
    exchange();
    const auto t2 = clk::now();
    for (PjRank &k : ranks) {
      hip_ok(hipSetDevice(k.device), "hipSetDevice");
      db_ok(dbhip_join_build_pairs_u32(k.rk.as<uint32_t>(), k.rr.as<uint32_t>(), k.recv_r, k.ids.as<uint32_t>(), k.join_ws.p,
                                       k.join_ws_bytes, k.stream),
            "dbhip_join_build_pairs_u32");
    }
    sync_all();
    const auto t3 = clk::now();
    for (PjRank &k : ranks) {
      hip_ok(hipSetDevice(k.device), "hipSetDevice");
      db_ok(dbhip_join_probe_u32(k.sk.as<uint32_t>(), k.recv_s, k.join_ws.p, k.recv_r, k.pos.as<uint32_t>(),
                                 k.cnt.as<uint32_t>(), k.stream),
            "dbhip_join_probe_u32");
    }
    sync_all();
    const auto t4 = clk::now();
    result->host_time = t4 - t0;
    result->build_time = t3 - t0;
    result->probe_time = t4 - t3;
    result->kernel_time = t4 - t0;  // phases are host-timed between syncs; no single device timeline spans the ranks
    std::cout << "Partition time: " << Duration(t1 - t0).count() << " us\nExchange time: " << Duration(t2 - t1).count()
              << " us\nLocal build time: " << Duration(t3 - t2).count() << " us\n";
    for (PjRank &k : ranks) {
      hip_ok(hipSetDevice(k.device), "hipSetDevice");
      check_status(k.join_ws.p, "PartitionedJoinHip");
      check_status(k.part_ws.p, "PartitionedJoinHip (partition)");
    }
    {
      // always on, at every size: the exchange must conserve the four columns — wrap-around sums of everything sent
      // equal the sums of everything received (device-side reduce, ~0.2 ms per GiB)
      int32_t sent[4] = {0, 0, 0, 0}, got[4] = {0, 0, 0, 0};
      for (PjRank &k : ranks) {
        hip_ok(hipSetDevice(k.device), "hipSetDevice");
        int32_t *scratch = k.chk.as<int32_t>();
        const void *cols[8] = {k.pk_r.p, k.pr_r.p, k.pk_s.p, k.pr_s.p, k.rk.p, k.rr.p, k.sk.p, k.sr.p};
        const size_t lens[8] = {k.n_local, k.n_local, k.n_local, k.n_local, k.recv_r, k.recv_r, k.recv_s, k.recv_s};
        for (int c = 0; c < 8; ++c)
          db_ok(dbhip_reduce_sum_i32(static_cast<const int32_t *>(cols[c]), lens[c], scratch + c, k.stream),
                "dbhip_reduce_sum_i32");
        hip_ok(hipStreamSynchronize(k.stream), "hipStreamSynchronize");
        const auto h = d2h<int32_t>(scratch, 8, k.device);
        for (int c = 0; c < 4; ++c) {
          sent[c] = static_cast<int32_t>(static_cast<uint32_t>(sent[c]) + static_cast<uint32_t>(h[c]));
          got[c] = static_cast<int32_t>(static_cast<uint32_t>(got[c]) + static_cast<uint32_t>(h[4 + c]));
        }
      }
      for (int c = 0; c < 4; ++c)
        if (sent[c] != got[c]) {
          std::cerr << "Incorrect results (exchange did not conserve column " << c << ")" << std::endl;
          result->valid = false;
        }
    }
    if (validate) {
      bool ok = true;
      std::vector<char> seen(n, 0);
      size_t delivered = 0;
      for (PjRank &k : ranks) {
        const auto rid = d2h<uint32_t>(k.sr.p, k.recv_s, k.device), key = d2h<uint32_t>(k.sk.p, k.recv_s, k.device),
                   hpos = d2h<uint32_t>(k.pos.p, k.recv_s, k.device), hcnt = d2h<uint32_t>(k.cnt.p, k.recv_s, k.device),
                   hids = d2h<uint32_t>(k.ids.p, k.recv_r, k.device);
        for (size_t i = 0; i < k.recv_s && ok; ++i) {
          ok = rid[i] < n && !seen[rid[i]] && probe_all[rid[i]] == key[i];  // every probe row arrives once, intact
          if (!ok) break;
          seen[rid[i]] = 1;
          ++delivered;
          const auto f = key_count.find(key[i]);
          const uint32_t want = f == key_count.end() ? 0u : f->second;
          ok = hcnt[i] == want;  // all build rows with this key were routed to the same rank
          for (uint32_t j = 0; ok && j < hcnt[i]; j += (hcnt[i] > 64 ? hcnt[i] / 64 : 1))
            ok = hpos[i] + j < k.recv_r && hids[hpos[i] + j] < n && build_all[hids[hpos[i] + j]] == key[i];
        }
      }
      if (!ok || delivered != n) {
        std::cerr << "Incorrect results" << std::endl;
        result->valid = false;
      }
    }
    meter.add_result(size_param(n), std::move(result));
  }
  for (ncclComm_t c : comms) (void)ncclCommDestroy(c);
  for (PjRank &k : ranks) {
    (void)hipSetDevice(k.device);
    (void)hipStreamDestroy(k.stream);
  }
  ranks.clear();
  hip_ok(hipSetDevice(home), "hipSetDevice");
}
void PartitionedJoinHip::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
void PartitionedJoinHip::init(const RunOptions &opts) { common_init(*this, opts); }
