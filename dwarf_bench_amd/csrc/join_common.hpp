// join_common.hpp — geometry shared by join.hip (the C entry points, the small-input unique-key table, the
// bitmask-claimed table) and join_lds.hip (radix-partitioned build with LDS sub-tables, every size).  Everything is a pure function of n_build, so build
// and probe agree without reading anything back from the device.
#pragma once
#include "dbhip_common.hpp"

namespace dbhip {

constexpr unsigned kJlSubSlots = 4096;  // slots of one LDS sub-table (48 KiB of LDS with counts/positions)
constexpr unsigned kJlSubMask = kJlSubSlots - 1;
constexpr unsigned kJlRowsPerPart = 2048;           // expected rows per partition (load factor <= 0.5)
constexpr size_t kJlMinRows = static_cast<size_t>(1) << 16;  // unique-key join only: below, the CAS table of join.hip
constexpr size_t kJlMaxRows = static_cast<size_t>(1) << 31;  // above: 2^20 partitions would overfill

// The unique-key payload join keeps a plain CAS table in HBM below kJlMinRows build rows (fewer launches); the
// one-to-many join takes the partitioned path at every size.  DBHIP_JOIN_PATH=lds|hbm overrides for experiments.
inline bool jl_use_ujoin(size_t n_build) {
  static const int force = [] {
    const char *e = getenv("DBHIP_JOIN_PATH");
    return !e ? 0 : (e[0] == 'l' ? 1 : (e[0] == 'h' ? 2 : 0));
  }();
  if (force == 2) return false;
  if (force == 1) return n_build > 0 && n_build <= kJlMaxRows;
  return n_build >= kJlMinRows && n_build <= kJlMaxRows;
}

struct JlLayout {
  unsigned parts, k1, k2, log2_k2;
  size_t table_off, keys_a_off, rids_a_off, keys_b_off, rids_b_off, meta_off, meta_bytes, total;
};

inline JlLayout jl_layout(size_t n) {
  JlLayout L;
  unsigned lg = 0;
  while ((static_cast<size_t>(kJlRowsPerPart) << lg) < n && lg < 20) ++lg;
  L.parts = 1u << lg;
  if (L.parts <= 1024) {  // one scatter level handles up to 1024 buckets
    L.log2_k2 = 0;
  } else {
    L.log2_k2 = lg / 2;
  }
  L.k2 = 1u << L.log2_k2;
  L.k1 = L.parts / L.k2;
  const size_t col = align_up((n ? n : 1) * sizeof(unsigned), kWsAlign);
  L.table_off = kWsHeader;
  // 8-byte slots {key, first id position} + one sentinel slot after the last sub-table
  L.keys_a_off = align_up(L.table_off + (static_cast<size_t>(L.parts) * kJlSubSlots + 1) * 8, kWsAlign);
  L.rids_a_off = L.keys_a_off + col;
  L.keys_b_off = L.rids_a_off + col;
  L.rids_b_off = L.keys_b_off + (L.k2 > 1 ? col : 0);
  L.meta_off = L.rids_b_off + (L.k2 > 1 ? col : 0);
  L.meta_bytes = sizeof(unsigned long long) * ((2 * 64 + 2) * static_cast<size_t>(L.k1) + 2 + 3 * static_cast<size_t>(L.parts) + 1);  // 64 = kJlGroups
  L.total = align_up(L.meta_off + L.meta_bytes, kWsAlign);
  return L;
}

int join_lds_build(const unsigned *build_keys, const unsigned *row_ids, size_t n, unsigned *ids, void *workspace,
                   hipStream_t s, const DeviceInfo &dev);
size_t jl_partition_workspace_bytes(unsigned parts);
int jl_partition(const unsigned *keys, size_t n, unsigned long long first_row, unsigned parts, unsigned *out_keys,
                 unsigned *out_rids, unsigned long long *out_counts, void *workspace, hipStream_t s,
                 const DeviceInfo &dev);
int jl_route_check(const unsigned *keys, size_t n, unsigned parts, unsigned rank, unsigned long long *result,
                   hipStream_t s, const DeviceInfo &dev);
int ujoin_lds_build(const unsigned *build_keys, const unsigned *build_vals, size_t n, void *workspace, hipStream_t s,
                    const DeviceInfo &dev);
int ujoin_lds_probe(const unsigned *probe_keys, const unsigned *probe_vals, size_t n_probe, const void *workspace,
                    size_t n_build, unsigned *out_key, unsigned *out_bval, unsigned *out_pval, hipStream_t s,
                    const DeviceInfo &dev);
int join_lds_probe(const unsigned *probe_keys, size_t n_probe, const void *workspace, size_t n_build,
                   unsigned *out_pos, unsigned *out_cnt, hipStream_t s, const DeviceInfo &dev);

}  // namespace dbhip
