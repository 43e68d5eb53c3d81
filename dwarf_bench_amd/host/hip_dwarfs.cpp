// hip_dwarfs.cpp — host side of the `...Hip` dwarfs (see hip_dwarfs.hpp).
//
// Every _run follows the reference's shape: generate the inputs once per size, then per iteration
// time the device work between steady_clock stamps (host_time: launch + sync, the reference's
// figure) and between hipEvents (kernel_time), validate, meter.add_result({{"buf_size", n}}, result).
// Differences, all deliberate:
//   * inputs are generated ON the device by the counter-based generators of libdbhip (deterministic
//     seeds instead of std::random_device, common/common.hpp:34-35) and stay resident: host_time does
//     not contain the reference's H2D/D2H of whole columns (scan/scan.cpp:108-120);
//   * validation uses the same host algorithms the reference dwarfs use (std::copy_if, std::sort, the
//     expected_GroupBy loop, per-key match counts) and is ALWAYS on — the reference only validates
//     DPLScan/Radix/GroupBy/JoinOmnisci in Debug builds — but is skipped above
//     DWARF_BENCH_VALIDATE_MAX elements (default 2^24) where a host check would dominate the run;
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
  const bool validate = n <= validate_limit();
  if (validate) {  // scan/scan.cpp:12-17 expected_out_lt
    const std::vector<int32_t> host = src.to_host(n);
    std::copy_if(host.begin(), host.end(), std::back_inserter(expected),
                 [](int v) { return v < filter_value; });
  }
  Events ev;
  for (size_t it = 0; it < opts.iterations; ++it) {
    auto result = std::make_unique<Result>();
    const auto host_start = clk::now();
    hip_ok(hipEventRecord(ev.a, nullptr), "event");
    db_ok(dbhip_copy_if_lt_i32(src.get(), n, filter_value, out.get(), out_size.get(), ws.get(), ws_bytes, nullptr),
          "dbhip_copy_if_lt_i32");
    hip_ok(hipEventRecord(ev.b, nullptr), "event");
    uint64_t count = 0;
    hip_ok(hipMemcpy(&count, out_size.get(), sizeof(count), hipMemcpyDeviceToHost), "out_size D2H");  // syncs
    const auto host_end = clk::now();
    result->host_time = host_end - host_start;
    result->kernel_time = ev.elapsed();
    result->bytes = n * sizeof(int32_t) + count * sizeof(int32_t);
    check_status(ws.get(), who);
    if (validate) {
      if (count != expected.size() || out.to_host(count) != expected) {
        std::cerr << "incorrect results" << std::endl;
        result->valid = false;
      }
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
  const bool validate = n <= validate_limit();
  std::vector<int32_t> expected;
  if (validate) {  // sort/radix.cpp:8-12
    expected = src.to_host(n);
    std::sort(expected.begin(), expected.end());
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
    if (validate && keys.to_host(n) != expected) {
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
  const bool validate = n <= validate_limit();
  std::vector<uint32_t> expected(groups, 0);
  if (validate) {  // groupby/groupby.cpp:8-19 expected_GroupBy with f = +
    const auto hk = keys.to_host(n);
    const auto hv = vals.to_host(n);
    for (size_t i = 0; i < n; ++i) expected[hk[i]] = expected[hk[i]] + hv[i];
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
    if (validate && out.to_host(groups) != expected) {
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
  const bool validate = n <= validate_limit();
  std::vector<uint32_t> ha, hb;
  std::unordered_map<uint32_t, uint32_t> key_count;
  if (validate) {
    ha = a.to_host(n);
    hb = b.to_host(n);
    for (uint32_t k : ha) ++key_count[k];
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
    if (validate) {
      // join/join_omnisci.cpp:31-45 are_equal: size per probe row + every returned id really matches
      const auto hpos = pos.to_host(n), hcnt = cnt.to_host(n), hids = ids.to_host(n);
      bool ok = true;
      for (size_t i = 0; i < n && ok; ++i) {
        const auto f = key_count.find(hb[i]);
        const uint32_t want = f == key_count.end() ? 0u : f->second;
        ok = hcnt[i] == want;
        for (uint32_t j = 0; ok && j < hcnt[i]; j += (hcnt[i] > 64 ? hcnt[i] / 64 : 1))
          ok = hpos[i] + j < n && hids[hpos[i] + j] < n && ha[hids[hpos[i] + j]] == hb[i];
      }
      if (!ok) {
        std::cerr << "Incorrect results" << std::endl;
        result->valid = false;
      }
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
  const bool validate = n <= validate_limit();
  std::unordered_map<uint32_t, uint32_t> a_payload;
  std::vector<uint32_t> hbk, hbv;
  if (validate) {
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
    if (validate) {
      // same table as seq_join would produce (join.cpp:27-28, :133): unique keys -> per probe row
      const auto hk = ok_.to_host(n), h1 = o1.to_host(n), h2 = o2.to_host(n);
      bool ok = true;
      for (size_t i = 0; i < n && ok; ++i) {
        const auto f = a_payload.find(hbk[i]);
        if (f == a_payload.end())
          ok = hk[i] == 0xFFFFFFFFu && h1[i] == 0xFFFFFFFFu && h2[i] == 0xFFFFFFFFu;
        else
          ok = hk[i] == hbk[i] && h1[i] == f->second && h2[i] == hbv[i];
      }
      if (!ok) {
        std::cerr << "Incorrect results" << std::endl;
        result->valid = false;
      }
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
  const bool validate = n <= validate_limit();
  std::vector<uint32_t> expected(groups, 0);
  if (validate) {
    const auto hk = keys.to_host(n);
    const auto hv = vals.to_host(n);
    for (size_t i = 0; i < n; ++i) expected[hk[i]] = expected[hk[i]] + hv[i];
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
    if (validate && out.to_host(groups) != expected) {
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
// ReduceHip — int sum of a column (reduce/reduce.cpp:27-98); expected = std::accumulate(..., 0) (:21).
ReduceHip::ReduceHip() : Dwarf("ReduceHip") {}
void ReduceHip::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  banner("ReduceHip");
  DevBuf<int32_t> src(n), out(1);
  db_ok(dbhip_gen_uniform_u32(reinterpret_cast<uint32_t *>(src.get()), n, 42, 0, 1, 10000, nullptr), "gen");
  hip_ok(hipDeviceSynchronize(), "sync");
  const bool validate = n <= validate_limit();
  int32_t expected = 0;
  if (validate) {
    const auto h = src.to_host(n);
    uint32_t acc = 0;  // accumulate with defined wrap-around; equals the int sum wherever that is defined
    for (int32_t v : h) acc += static_cast<uint32_t>(v);
    expected = static_cast<int32_t>(acc);
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
    if (validate && host_out != expected) {
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
