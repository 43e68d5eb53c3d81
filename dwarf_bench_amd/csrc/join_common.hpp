// join_common.hpp — geometry shared by join.hip (the C entry points, the small-input unique-key table, the
// bitmask-claimed table) and join_lds.hip (radix-partitioned build with LDS sub-tables, every size).  Everything is a pure function of n_build, so build
// and probe agree without reading anything back from the device.
#pragma once
#include "dbhip_common.hpp"

namespace dbhip {

// Sub-table geometry (compile-time knobs for experiments).  A partition is expected to hold kJlRowsPerPart rows
// (Poisson: +-6 sigma = +-270 at 2048) and owns kJlSubSlots slots of an LDS-resident open-addressing table: any
// number of rows; a partition with more than kJlSubSlots DISTINCT keys (keys constructed against the hash) is built in the
// spill pool instead (below).  3072 slots for 2048 rows: all-distinct keys (the unique-key
// join) load the table to 0.67, the uniform one-to-many regime (63 % distinct) to 0.42; the table costs 12 bytes per
// build row in HBM (4096 slots, the first geometry: 16).  Measured at 2^26 rows, whole build / probe: see DESIGN.md.
#ifndef DBHIP_JL_SUB_SLOTS
#define DBHIP_JL_SUB_SLOTS 3072
#endif
#ifndef DBHIP_JL_ROWS_PER_PART
#define DBHIP_JL_ROWS_PER_PART 2048
#endif
constexpr unsigned kJlSubSlots = DBHIP_JL_SUB_SLOTS;          // slots of one LDS sub-table (a multiple of 512)
constexpr unsigned kJlRowsPerPart = DBHIP_JL_ROWS_PER_PART;   // expected rows per partition
static_assert(kJlSubSlots % 512 == 0 && kJlSubSlots <= 65536, "sub-table size");
constexpr size_t kJlMinRows = static_cast<size_t>(1) << 16;  // unique-key join only: below, the CAS table of join.hip
constexpr size_t kJlMaxRows = static_cast<size_t>(1) << 31;  // 32-bit row ids and id positions

// slot of a key inside its sub-table, from the mixed hash h = fmix32(key): the partition is the HIGH log2(parts)
// bits of h (up to 20), so the slot takes the top 16 bits of h * golden-ratio — every low bit of h reaches them, the
// bits that are constant inside one partition do not pin them — range-reduced by multiply-shift; linear probing
// wraps at kJlSubSlots
__host__ __device__ __forceinline__ unsigned jl_home_slot(unsigned h) {
  return (((h * 0x9E3779B1u) >> 16) * kJlSubSlots) >> 16;
}
__host__ __device__ __forceinline__ unsigned jl_next_slot(unsigned s) { return s + 1 == kJlSubSlots ? 0u : s + 1; }

// One-to-many table slot = {key, first id position | (count field << pos_bits)}: the id positions of a build of n
// rows need pos_bits = bit width of n; the bits above hold min(count - 1, escape) with escape = all ones.  A probe
// that finds a smaller field has the count without a second access; the escape value (and pos_bits == 32) sends it
// to the right-hand neighbour, whose first position ends this slot's id range (positions run on through empty slots
// and from one sub-table to the next, closed by one sentinel slot).
__host__ __device__ __forceinline__ unsigned jl_pos_bits(size_t n_build) {
  unsigned b = 1;
  while (b < 32 && (static_cast<size_t>(1) << b) <= n_build) ++b;
  return b;
}

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

#ifndef DBHIP_JL_K2_BIAS
#define DBHIP_JL_K2_BIAS 0
#endif
// A GIANT partition: one hot key (or a few) gives a partition far more rows than one workgroup should walk alone —
// every other row of 2^24 carrying one key made the build 22.5 ms against 0.3 ms.  A partition above jl_giant_rows(n)
// rows is left out by the per-partition build and counted / filled by all workgroups together, slice by slice
// (jl_giant_count / jl_giant_ids in join_lds.hip).  There are fewer than n / jl_giant_rows(n) of them: the scratch
// below (list, finished-slice counters, per giant the key counts and the id cursors of its sub-table's slots) is sized
// for that, so no giant is ever left to the slow path.  Below 2^18 build rows the two extra launches would cost more
// than a giant can (a partition of 2^17 rows: 0.3 ms): the path is off, jl_max_giants = 0.
__host__ __device__ __forceinline__ size_t jl_giant_rows(size_t n) {
  const size_t t = n / 1024;
  return t < 32768 ? static_cast<size_t>(32768) : t;
}
inline unsigned jl_max_giants(size_t n) {
  return n < (static_cast<size_t>(1) << 18) ? 0u : static_cast<unsigned>(n / jl_giant_rows(n) + 1);
}
// scratch: count, pad[3] | part[max] | done[max] | counts[max][kJlSubSlots] | cursors[max][kJlSubSlots]; the first
// jl_giant_header_bytes are cleared with the partitioner's meta words, a giant's counts by the workgroup that lists it
inline size_t jl_giant_header_bytes(unsigned max_giants) { return max_giants ? 16 + 8 * static_cast<size_t>(max_giants) : 0; }
inline size_t jl_giant_bytes(unsigned max_giants, bool with_tables = false) {
  if (!max_giants) return 0;
  const size_t tables = with_tables ? static_cast<size_t>(max_giants) * (kJlSubSlots + 1) * 8 : 0;  // the radix join publishes no table: its giants' sub-tables live here (+ a sentinel slot each)
  return align_up(jl_giant_header_bytes(max_giants), 16) + static_cast<size_t>(max_giants) * kJlSubSlots * 8 + tables;
}
// the radix join's list also takes partitions whose PROBE side is that large
inline unsigned jr_max_giants(size_t n_build, size_t n_probe) {
  if (n_build < (static_cast<size_t>(1) << 18) && n_probe < (static_cast<size_t>(1) << 18)) return 0u;
  return static_cast<unsigned>(n_build / jl_giant_rows(n_build) + n_probe / jl_giant_rows(n_probe) + 2);
}
constexpr unsigned kJlMaxGiantList = 2052;  // >= 2 * (1024 + 1) + 1: the giant kernels' LDS list

// SPILLED partitions (round 4): a partition with more DISTINCT keys than its sub-table has slots.  The mixing hash puts
// 2048 +- 45 keys into a partition, so only keys constructed against the hash get there — but the reference's table takes
// any input (ht_size = 2 * distinct, join/join_omnisci.cpp:69-70; omnisci_hashtable.hpp:80-108 probes until it finds a
// slot), so this library does too: such a partition is published as an empty sub-table, listed, and built by ONE workgroup
// into an open-addressing table of its own in HBM (memory-side atomics: slow, correct for any input; jl_spill_partition
// in join_lds.hip), and probes of its keys go there through a per-partition directory.  Scratch, behind the giants':
//   dir[parts] {first slot + 1 | 0, slots}  |  list[max]  |  keys[pool] | pos[pool] | cnt[pool]
// A spilled partition of R rows takes R + R/2 + 64 slots of the pool; there are fewer than n / kJlSubSlots + 2 of them.
// Header words of the workspace: [34] listed partitions, [35] pool slots taken.
constexpr unsigned kJlHdrTicket = 32, kJlHdrLeft = 33, kJlHdrSpilled = 34, kJlHdrSpillPool = 35;
__host__ __device__ __forceinline__ unsigned jl_max_spill(size_t n) { return static_cast<unsigned>(n / kJlSubSlots + 2); }
__host__ __device__ __forceinline__ size_t jl_spill_cap(size_t rows) { return rows + rows / 2 + 64; }
__host__ __device__ __forceinline__ size_t jl_spill_pool_slots(size_t n) { return n + n / 2 + 64 * static_cast<size_t>(jl_max_spill(n)); }
__host__ __device__ __forceinline__ size_t jl_spill_list_words(size_t n) { return (static_cast<size_t>(jl_max_spill(n)) + 3) & ~static_cast<size_t>(3); }
inline size_t jl_spill_bytes(unsigned parts, size_t n) {
  return align_up(8 * static_cast<size_t>(parts) + 4 * jl_spill_list_words(n) + 12 * jl_spill_pool_slots(n), kWsAlign);
}

struct JlLayout {
  unsigned parts, k1, k2, log2_k2, max_giants;
  size_t table_off, keys_a_off, rids_a_off, keys_b_off, rids_b_off, meta_off, meta_bytes, giant_off, spill_off, total;
};

// The radix join's fused build + probe kernel likes its partitions emptier than the build kernel does (2^26 x 2^26, rows
// per partition / fused kernel / whole radix join: 2048 / 908 us / 2103 us, 1920 / 798 / 2024, 1792 / 779 / 2006,
// 1536 / 782 / 2013; the build that publishes its tables gets slower instead: 1053 -> 1065 -> 1100 us)
#ifndef DBHIP_JR_ROWS_PER_PART
#define DBHIP_JR_ROWS_PER_PART 1792
#endif
constexpr unsigned kJrRowsPerPart = DBHIP_JR_ROWS_PER_PART;
static_assert(kJrRowsPerPart <= kJlRowsPerPart, "the build kernel caches kJlRowsPerPart + 1/8 rows of a partition");

inline JlLayout jl_layout(size_t n, size_t rows_per_part = kJlRowsPerPart) {
  JlLayout L;
  // parts = ceil(n / kJlRowsPerPart) rounded up to a multiple of the level-1 fan-out k2 (a power of two: level 1 takes
  // the low bits of the partition id, level 0 the rest — any number k1 <= 1024 of buckets; the partition id itself is a
  // multiply-shift of the hash and takes any range).  Until late in round 3 parts was the next POWER of two: one row
  // more than 2^26 meant 65536 half-empty partitions — build 1245 us against 1029, twice the table.
  size_t want = (n + rows_per_part - 1) / rows_per_part;
  if (want == 0) want = 1;
  if (want > (static_cast<size_t>(1) << 20)) want = static_cast<size_t>(1) << 20;  // 2^20 partitions at most
  unsigned lg = 0;
  while ((static_cast<size_t>(1) << lg) < want) ++lg;
  if (want <= 1024) {  // one scatter level handles up to 1024 buckets
    L.log2_k2 = 0;
  } else {
    // split of the partition bits between the two scatter levels, by floor(log2(parts)): between two powers of two
    // level 0 takes the extra buckets (37504 partitions as 293 x 128: one side of 2^26 rows 615 us; as 147 x 256: 639)
    const unsigned lgs = (static_cast<size_t>(1) << lg) != want ? lg - 1 : lg;
    L.log2_k2 = (lgs + DBHIP_JL_K2_BIAS) / 2;
  }
  L.k2 = 1u << L.log2_k2;
  L.k1 = static_cast<unsigned>((want + L.k2 - 1) / L.k2);
  while (L.k1 > 1024) {  // level 0 (one workgroup of 1024 threads owns the bucket offsets) takes at most 1024 buckets
    ++L.log2_k2;
    L.k2 <<= 1;
    L.k1 = static_cast<unsigned>((want + L.k2 - 1) / L.k2);
  }
  L.parts = L.k1 * L.k2;
  const size_t col = align_up((n ? n : 1) * sizeof(unsigned), kWsAlign);
  L.table_off = kWsHeader;
  // 8-byte slots {key, first id position} + one sentinel slot after the last sub-table
  L.keys_a_off = align_up(L.table_off + (static_cast<size_t>(L.parts) * kJlSubSlots + 1) * 8, kWsAlign);
  L.rids_a_off = L.keys_a_off + col;
  L.keys_b_off = L.rids_a_off + col;
  L.rids_b_off = L.keys_b_off + (L.k2 > 1 ? col : 0);
  L.meta_off = L.rids_b_off + (L.k2 > 1 ? col : 0);
  L.meta_bytes = sizeof(unsigned long long) * ((2 * 64 + 2) * static_cast<size_t>(L.k1) + 2 + 3 * static_cast<size_t>(L.parts) + 1);  // 64 = kJlGroups
  L.max_giants = jl_max_giants(n);
  L.giant_off = align_up(L.meta_off + L.meta_bytes, kWsAlign);
  L.spill_off = align_up(L.giant_off + jl_giant_bytes(L.max_giants), kWsAlign);
  L.total = L.spill_off + jl_spill_bytes(L.parts, n);
  return L;
}

int join_lds_build(const unsigned *build_keys, const unsigned *row_ids, size_t n, unsigned *ids, void *workspace,
                   hipStream_t s, const DeviceInfo &dev);
size_t jl_partition_workspace_bytes(unsigned parts);
int jl_partition(const unsigned *keys, size_t n, unsigned long long first_row, unsigned parts, unsigned *out_keys,
                 unsigned *out_rids, unsigned long long *out_counts, void *workspace, hipStream_t s,
                 const DeviceInfo &dev);
size_t join_radix_workspace_bytes(size_t n_build, size_t n_probe);
int join_radix_partition(int probe_side, const unsigned *keys, const unsigned *row_ids, size_t n, size_t n_build,
                         size_t n_probe, void *workspace, hipStream_t s, const DeviceInfo &dev);
int join_radix_match(size_t n_build, size_t n_probe, unsigned *ids, unsigned *out_rid, unsigned *out_pos, unsigned *out_cnt,
                     void *workspace, hipStream_t s, const DeviceInfo &dev);
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
