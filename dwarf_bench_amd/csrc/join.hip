// join.hip — dwarf 4: entry points of the hash joins for gfx950, the small-input path of the unique-key join and the
// bitmask-claimed table.
//
// 4a  one-to-many join, JoinOmnisci semantics (common/dpcpp/omnisci_hashtable.hpp:58-261 <-
//     join/join_omnisci.cpp:74-88): distinct-key table, per-key match count, exclusive scan -> position, ids
//     grouped by key, probe -> {offset into ids, count}.  Every size runs the radix-partitioned build with
//     LDS-resident sub-tables of join_lds.hip.  (The first version kept the table in HBM — atomicCAS insert + count,
//     a single-pass look-back scan over the slot counters, id fill: 19.3 ms at 2^26 x 2^26 against 2.9 ms, and not
//     faster at any small size either, 2^8..2^16 rows: 34-46 us against 30-43 us — and was removed together with
//     the look-back code, so nothing in this library spins on another workgroup any more.)
//     Table layout is not contractual: logical outputs are count per probe row + the set of ids.
// 4b  unique-key payload join, Join semantics (join/join.cpp:60-104): below 2^16 build rows a CAS-claimed
//     {key, payload} table in HBM (four launches: 24-42 us where the partitioned build needs 31-49 us), above it the
//     partitioned build of join_lds.hip; probe writes (key, build_val, probe_val) at the probe row or the
//     0xFFFFFFFF sentinel.
#include "dbhip_common.hpp"
#include "join_common.hpp"

namespace dbhip {
namespace {

constexpr unsigned kEmpty = 0xFFFFFFFFu;
constexpr int kJoinThreads = 256;
inline size_t join_capacity(size_t n_build) {
  size_t cap = 4096;
  while (cap < 2 * n_build) cap <<= 1;  // load factor <= 0.5 (reference: ht_size = 2 * distinct)
  return cap;
}

inline unsigned grid_for(size_t n, const DeviceInfo &dev, int per_cu) {
  const size_t want = (n + kJoinThreads - 1) / kJoinThreads;
  const size_t cap = static_cast<size_t>(dev.cus) * per_cu;
  return static_cast<unsigned>(want < cap ? (want ? want : 1) : cap);
}

// find the slot holding `key`, or the first empty slot of its probe chain
__device__ __forceinline__ unsigned find_slot(const unsigned *__restrict__ keys, unsigned key,
                                              unsigned mask, bool *found) {
  unsigned s = fmix32(key) & mask;
  for (unsigned tries = 0; tries <= mask && key != kEmpty; ++tries) {
    const unsigned k = keys[s];
    if (k == key) {
      *found = true;
      return s;
    }
    if (k == kEmpty) break;
    s = (s + 1) & mask;
  }
  *found = false;
  return s;
}

// ---- 4b unique-key payload join -------------------------------------------------------------------
struct UjoinHeader {
  unsigned status;
  unsigned pad0;
  unsigned long long capacity;
  unsigned pad[60];
};
static_assert(sizeof(UjoinHeader) == kWsHeader, "workspace header size");

__global__ __launch_bounds__(kJoinThreads) void ujoin_build_kernel(const unsigned *__restrict__ bkeys,
                                                                   const unsigned *__restrict__ bvals,
                                                                   size_t n, unsigned *keys,
                                                                   unsigned *__restrict__ vals,
                                                                   unsigned mask, UjoinHeader *hdr) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kJoinThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kJoinThreads + threadIdx.x; i < n; i += stride) {
    const unsigned key = bkeys[i];
    unsigned s = fmix32(key) & mask;
    unsigned tries = 0;
    if (key == kEmpty) {  // the sentinel is not a key: flag it, drop the row
      atomicOr(&hdr->status, DBHIP_DEV_KEY_RANGE);
      continue;
    }
    while (true) {
      const unsigned old = atomicCAS(&keys[s], kEmpty, key);
      if (old == kEmpty) {  // claimed: the payload store is read only by the next launch
        vals[s] = bvals[i];
        break;
      }
      if (old == key) break;  // duplicate build key: first claim wins (keys are unique by contract)
      s = (s + 1) & mask;
      if (++tries > mask) {
        atomicOr(&hdr->status, DBHIP_DEV_TABLE_FULL);
        break;
      }
    }
  }
}

__global__ __launch_bounds__(kJoinThreads) void ujoin_probe_kernel(
    const unsigned *__restrict__ pkeys, const unsigned *__restrict__ pvals, size_t n,
    const unsigned *__restrict__ keys, const unsigned *__restrict__ vals, unsigned mask,
    unsigned *__restrict__ out_key, unsigned *__restrict__ out_bval, unsigned *__restrict__ out_pval) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kJoinThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kJoinThreads + threadIdx.x; i < n; i += stride) {
    const unsigned key = pkeys[i];
    bool found;
    const unsigned s = find_slot(keys, key, mask, &found);
    out_key[i] = found ? key : kEmpty;
    out_bval[i] = found ? vals[s] : kEmpty;
    out_pval[i] = found ? pvals[i] : kEmpty;
  }
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

extern "C" size_t dbhip_join_workspace_bytes(size_t n_build) { return jl_layout(n_build).total; }

namespace {
// shared by dbhip_join_build_u32 (row ids = 0..n-1) and dbhip_join_build_pairs_u32 (caller's row ids)
int join_build_impl(const uint32_t *build_keys, const uint32_t *row_ids, size_t n_build, uint32_t *ids,
                    void *workspace, size_t workspace_bytes, dbhip_stream_t stream) {
  if (n_build && (!build_keys || !ids)) return DBHIP_EINVAL;
  if (n_build > kJlMaxRows) return DBHIP_EINVAL;  // 32-bit ids / positions, 2^20 partitions of 2048 rows
  if (!ws_ok(workspace, workspace_bytes, dbhip_join_workspace_bytes(n_build))) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  return join_lds_build(build_keys, row_ids, n_build, ids, workspace, as_stream(stream), dev);
}
}  // namespace

extern "C" int dbhip_join_build_u32(const uint32_t *build_keys, size_t n_build, uint32_t *ids,
                                    void *workspace, size_t workspace_bytes, dbhip_stream_t stream) {
  return join_build_impl(build_keys, nullptr, n_build, ids, workspace, workspace_bytes, stream);
}

extern "C" int dbhip_join_build_pairs_u32(const uint32_t *build_keys, const uint32_t *build_row_ids,
                                          size_t n_build, uint32_t *ids, void *workspace,
                                          size_t workspace_bytes, dbhip_stream_t stream) {
  if (n_build && !build_row_ids) return DBHIP_EINVAL;
  return join_build_impl(build_keys, build_row_ids, n_build, ids, workspace, workspace_bytes, stream);
}

extern "C" int dbhip_join_probe_u32(const uint32_t *probe_keys, size_t n_probe, const void *workspace,
                                    size_t n_build, uint32_t *out_pos, uint32_t *out_count,
                                    dbhip_stream_t stream) {
  if (n_probe == 0) return DBHIP_OK;
  if (!probe_keys || !workspace || !out_pos || !out_count || n_build > kJlMaxRows) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  // the table geometry is a pure function of the build size: no read-back, no synchronisation
  return join_lds_probe(probe_keys, n_probe, workspace, n_build, out_pos, out_count, as_stream(stream), dev);
}

// ---- radix join (both sides partitioned alike, fused LDS build + probe): results in the probe side's partition order
extern "C" size_t dbhip_join_radix_workspace_bytes(size_t n_build, size_t n_probe) {
  return join_radix_workspace_bytes(n_build, n_probe);
}

extern "C" int dbhip_join_radix_partition_u32(int probe_side, const uint32_t *keys, const uint32_t *row_ids, size_t n,
                                              size_t n_build, size_t n_probe, void *workspace, size_t workspace_bytes,
                                              dbhip_stream_t stream) {
  if (n != (probe_side ? n_probe : n_build) || (n && !keys)) return DBHIP_EINVAL;
  if (n_build > kJlMaxRows || n_probe > 0xFFFFFFFFull) return DBHIP_EINVAL;
  if (!ws_ok(workspace, workspace_bytes, join_radix_workspace_bytes(n_build, n_probe))) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  return join_radix_partition(probe_side ? 1 : 0, keys, row_ids, n, n_build, n_probe, workspace, as_stream(stream), dev);
}

extern "C" int dbhip_join_radix_match_u32(size_t n_build, size_t n_probe, uint32_t *ids, uint32_t *out_probe_row_ids,
                                          uint32_t *out_pos, uint32_t *out_count, void *workspace, size_t workspace_bytes,
                                          dbhip_stream_t stream) {
  if ((n_build && !ids) || (n_probe && (!out_probe_row_ids || !out_pos || !out_count))) return DBHIP_EINVAL;
  if (n_build > kJlMaxRows || n_probe > 0xFFFFFFFFull) return DBHIP_EINVAL;
  if (!ws_ok(workspace, workspace_bytes, join_radix_workspace_bytes(n_build, n_probe))) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  return join_radix_match(n_build, n_probe, ids, out_probe_row_ids, out_pos, out_count, workspace, as_stream(stream), dev);
}

extern "C" int dbhip_join_radix_u32(const uint32_t *build_keys, const uint32_t *build_row_ids, size_t n_build,
                                    const uint32_t *probe_keys, const uint32_t *probe_row_ids, size_t n_probe, uint32_t *ids,
                                    uint32_t *out_probe_row_ids, uint32_t *out_pos, uint32_t *out_count, void *workspace,
                                    size_t workspace_bytes, dbhip_stream_t stream) {
  int rc = dbhip_join_radix_partition_u32(0, build_keys, build_row_ids, n_build, n_build, n_probe, workspace,
                                          workspace_bytes, stream);
  if (rc == 0)
    rc = dbhip_join_radix_partition_u32(1, probe_keys, probe_row_ids, n_probe, n_build, n_probe, workspace, workspace_bytes,
                                        stream);
  if (rc == 0)
    rc = dbhip_join_radix_match_u32(n_build, n_probe, ids, out_probe_row_ids, out_pos, out_count, workspace,
                                    workspace_bytes, stream);
  return rc;
}

extern "C" size_t dbhip_ujoin_workspace_bytes(size_t n_build) {
  if (jl_use_ujoin(n_build)) return jl_layout(n_build).total;  // radix-partitioned build, LDS sub-tables (join_lds.hip)
  return align_up(kWsHeader + 2 * join_capacity(n_build) * sizeof(unsigned), kWsAlign);
}

extern "C" int dbhip_ujoin_build_u32(const uint32_t *build_keys, const uint32_t *build_vals,
                                     size_t n_build, void *workspace, size_t workspace_bytes,
                                     dbhip_stream_t stream) {
  if (n_build && (!build_keys || !build_vals)) return DBHIP_EINVAL;
  if (!ws_ok(workspace, workspace_bytes, dbhip_ujoin_workspace_bytes(n_build))) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  if (jl_use_ujoin(n_build)) return ujoin_lds_build(build_keys, build_vals, n_build, workspace, s, dev);
  const size_t cap = join_capacity(n_build);
  char *base = static_cast<char *>(workspace);
  UjoinHeader *hdr = reinterpret_cast<UjoinHeader *>(base);
  unsigned *keys = reinterpret_cast<unsigned *>(base + kWsHeader);
  unsigned *vals = keys + cap;
  hipError_t e = fill_async(base, 0, kWsHeader, s);
  if (e == hipSuccess) e = fill_async(keys, 0xFF, cap * sizeof(unsigned), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n_build)
    hipLaunchKernelGGL(ujoin_build_kernel, dim3(grid_for(n_build, dev, 8)), dim3(kJoinThreads), 0, s,
                       build_keys, build_vals, n_build, keys, vals, static_cast<unsigned>(cap - 1), hdr);
  return launch_status();
}

extern "C" int dbhip_ujoin_probe_u32(const uint32_t *probe_keys, const uint32_t *probe_vals,
                                     size_t n_probe, const void *workspace, size_t n_build,
                                     uint32_t *out_key, uint32_t *out_build_val,
                                     uint32_t *out_probe_val, dbhip_stream_t stream) {
  if (n_probe == 0) return DBHIP_OK;
  if (!probe_keys || !probe_vals || !workspace || !out_key || !out_build_val || !out_probe_val)
    return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  if (jl_use_ujoin(n_build))
    return ujoin_lds_probe(probe_keys, probe_vals, n_probe, workspace, n_build, out_key, out_build_val, out_probe_val,
                           as_stream(stream), dev);
  const size_t cap = join_capacity(n_build);
  const char *base = static_cast<const char *>(workspace);
  const unsigned *keys = reinterpret_cast<const unsigned *>(base + kWsHeader);
  const unsigned *vals = keys + cap;
  hipLaunchKernelGGL(ujoin_probe_kernel, dim3(grid_for(n_probe, dev, 8)), dim3(kJoinThreads), 0,
                     as_stream(stream), probe_keys, probe_vals, n_probe, keys, vals,
                     static_cast<unsigned>(cap - 1), out_key, out_build_val, out_probe_val);
  return launch_status();
}

// =====================================================================================================
// Bitmask-claimed open-addressing table: HIP counterpart of SimpleNonOwningHashTable
// (common/dpcpp/hashtable.hpp:5-93), the table behind the reference's Join and HashBuild dwarfs.
// Same algorithm: a row claims a free slot by fetch_or on the occupancy word of its hash position,
// skipping runs of occupied slots with ctz(~(present >> minor)) (hashtable.hpp:70-92); key and payload
// are then stored plainly (read only by a later launch); lookups walk while the occupancy bit is set
// (hashtable.hpp:23-62).  Duplicate keys occupy separate slots, as in the reference.  Hashers are the
// reference's: key % size (StaticSimpleHasher/SimpleHasher, hashfunctions.hpp:33-49) or
// MurmurHash3_x86_32(key, seed) % size (hashfunctions.hpp:64-137).
// =====================================================================================================
namespace dbhip {
namespace {

struct BmLayout {
  size_t keys_off, vals_off, mask_off, mask_words, total;
};
inline BmLayout bm_layout(size_t size) {
  BmLayout L;
  L.keys_off = kWsHeader;
  L.vals_off = L.keys_off + size * sizeof(unsigned);
  L.mask_off = align_up(L.vals_off + size * sizeof(unsigned), 16);
  L.mask_words = (size + 31) / 32;  // join.cpp:31 ceil(ht_size / 32)
  L.total = align_up(L.mask_off + L.mask_words * sizeof(unsigned), kWsAlign);
  return L;
}

__device__ __forceinline__ unsigned rotl32(unsigned x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ unsigned bm_hash(unsigned key, int kind, unsigned seed, unsigned size) {
  if (kind == 0) return key % size;
  unsigned k1 = key * 0xcc9e2d51u;  // one 4-byte block, no tail (hashfunctions.hpp:94-133, _len = 4)
  k1 = rotl32(k1, 15) * 0x1b873593u;
  unsigned h1 = seed ^ k1;
  h1 = rotl32(h1, 13) * 5u + 0xe6546b64u;
  h1 ^= 4u;
  return fmix32(h1) % size;
}

// One departure from hashtable.hpp:70-92, not visible where the reference is well defined: `minor += occupied`
// may step onto a slot >= size in a ragged last word, which the reference would then claim and write out of
// bounds; here that step wraps to word 0 like any other end of table.  A table with no free slot makes the
// reference spin forever; here the walk gives up after two laps' worth of visits and raises DBHIP_DEV_TABLE_FULL.
// (Measured and rejected: peeking the word with a plain or sc1 load before the fetch_or to skip full words
// without an atomic — 1.5-4x slower at every size, for unique and for duplicate-heavy keys alike.)
__device__ __forceinline__ void bm_insert(unsigned key, unsigned val, unsigned size, unsigned mask_words,
                                          int kind, unsigned seed, unsigned *keys, unsigned *vals,
                                          unsigned *bitmask, unsigned *status) {
  const unsigned at = bm_hash(key, kind, seed, size);
  unsigned major = at / 32u, minor = at % 32u;
  unsigned pos;
  unsigned long long budget = 64ull * mask_words + 64;  // visits: at most 32 per word per lap
  while (true) {  // update_bitmask, hashtable.hpp:70-92
    if (budget == 0) {
      atomicOr(status, static_cast<unsigned>(DBHIP_DEV_TABLE_FULL));
      return;
    }
    --budget;
    if (major * 32u + minor >= size) {  // ragged last word: nothing claimable from here on
      major = 0;
      minor = 0;
      continue;
    }
    const unsigned bit = 1u << minor;
    const unsigned present = atomicOr(&bitmask[major], bit);
    if (!(present & bit)) {
      pos = major * 32u + minor;
      break;
    }
    const unsigned inv = ~(present >> minor);
    const unsigned occupied = inv ? static_cast<unsigned>(__builtin_ctz(inv)) : 32u;
    if (occupied + minor >= 32u) {
      major = (major + 1) % mask_words;
      minor = 0;
    } else {
      minor += occupied;
    }
  }
  keys[pos] = key;  // hashtable.hpp:16-18
  vals[pos] = val;
}

__global__ __launch_bounds__(kJoinThreads) void bm_insert_kernel(const unsigned *__restrict__ in_keys,
                                                                 const unsigned *__restrict__ in_vals, size_t n,
                                                                 unsigned size, unsigned mask_words, int kind,
                                                                 unsigned seed, unsigned *keys, unsigned *vals,
                                                                 unsigned *bitmask, unsigned *status,
                                                                 int serial) {
  if (serial) {  // one work-item inserts in order: reproduces the reference tests' slot layouts
    if (blockIdx.x == 0 && threadIdx.x == 0)
      for (size_t i = 0; i < n; ++i) bm_insert(in_keys[i], in_vals[i], size, mask_words, kind, seed, keys, vals, bitmask, status);
    return;
  }
  const size_t stride = static_cast<size_t>(gridDim.x) * kJoinThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kJoinThreads + threadIdx.x; i < n; i += stride)
    bm_insert(in_keys[i], in_vals[i], size, mask_words, kind, seed, keys, vals, bitmask, status);
}

__global__ __launch_bounds__(kJoinThreads) void bm_lookup_kernel(const unsigned *__restrict__ q, size_t n,
                                                                 unsigned size, int kind, unsigned seed,
                                                                 const unsigned *__restrict__ keys,
                                                                 const unsigned *__restrict__ vals,
                                                                 const unsigned *__restrict__ bitmask,
                                                                 unsigned *__restrict__ out_vals,
                                                                 unsigned *__restrict__ out_found) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kJoinThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kJoinThreads + threadIdx.x; i < n; i += stride) {
    const unsigned key = q[i];
    unsigned pos = bm_hash(key, kind, seed, size);
    const unsigned start = pos;
    unsigned found = 0, val = 0;
    bool present = (bitmask[pos / 32u] >> (pos % 32u)) & 1u;
    while (present) {  // hashtable.hpp:23-40
      if (keys[pos] == key) {
        found = 1;
        val = vals[pos];
        break;
      }
      pos = (pos + 1) % size;
      if (pos == start) break;
      present = (bitmask[pos / 32u] >> (pos % 32u)) & 1u;
    }
    if (out_vals) out_vals[i] = val;
    if (out_found) out_found[i] = found;
  }
}

}  // namespace
}  // namespace dbhip

extern "C" size_t dbhip_bitmask_table_workspace_bytes(size_t table_size) {
  return table_size ? dbhip::bm_layout(table_size).total : 0;
}

extern "C" int dbhip_bitmask_table_reset(void *workspace, size_t workspace_bytes, size_t table_size,
                                         dbhip_stream_t stream) {
  using namespace dbhip;
  if (table_size == 0 || table_size > 0xFFFFFFFFull) return DBHIP_EINVAL;
  const BmLayout L = bm_layout(table_size);
  if (!ws_ok(workspace, workspace_bytes, L.total)) return DBHIP_EWORKSPACE;
  char *base = static_cast<char *>(workspace);
  hipStream_t s = as_stream(stream);
  hipError_t e = fill_async(base, 0, kWsHeader, s);
  if (e == hipSuccess) e = fill_async(base + L.keys_off, 0xFF, table_size * sizeof(unsigned), s);  // join.cpp:37
  if (e == hipSuccess) e = fill_async(base + L.vals_off, 0, table_size * sizeof(unsigned), s);
  if (e == hipSuccess) e = fill_async(base + L.mask_off, 0, L.mask_words * sizeof(unsigned), s);
  return static_cast<int>(e);
}

extern "C" int dbhip_bitmask_table_insert_u32(const uint32_t *keys, const uint32_t *vals, size_t n, void *workspace,
                                              size_t workspace_bytes, size_t table_size, int hash_kind,
                                              uint32_t seed, int serial, dbhip_stream_t stream) {
  using namespace dbhip;
  if (n == 0) return DBHIP_OK;
  if (!keys || !vals || table_size == 0 || table_size > 0xFFFFFFFFull || (hash_kind != 0 && hash_kind != 1))
    return DBHIP_EINVAL;
  const BmLayout L = bm_layout(table_size);
  if (!ws_ok(workspace, workspace_bytes, L.total)) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  char *base = static_cast<char *>(workspace);
  hipLaunchKernelGGL(bm_insert_kernel, dim3(serial ? 1 : grid_for(n, dev, 8)), dim3(kJoinThreads), 0, as_stream(stream),
                     keys, vals, n, static_cast<unsigned>(table_size), static_cast<unsigned>(L.mask_words), hash_kind,
                     seed, reinterpret_cast<unsigned *>(base + L.keys_off), reinterpret_cast<unsigned *>(base + L.vals_off),
                     reinterpret_cast<unsigned *>(base + L.mask_off), reinterpret_cast<unsigned *>(base), serial);
  return launch_status();
}

extern "C" int dbhip_bitmask_table_lookup_u32(const uint32_t *keys, size_t n, const void *workspace, size_t table_size,
                                              int hash_kind, uint32_t seed, uint32_t *out_vals, uint32_t *out_found,
                                              dbhip_stream_t stream) {
  using namespace dbhip;
  if (n == 0) return DBHIP_OK;
  if (!keys || !workspace || table_size == 0 || table_size > 0xFFFFFFFFull || (hash_kind != 0 && hash_kind != 1))
    return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  const BmLayout L = bm_layout(table_size);
  const char *base = static_cast<const char *>(workspace);
  hipLaunchKernelGGL(bm_lookup_kernel, dim3(grid_for(n, dev, 8)), dim3(kJoinThreads), 0, as_stream(stream), keys, n,
                     static_cast<unsigned>(table_size), hash_kind, seed,
                     reinterpret_cast<const unsigned *>(base + L.keys_off),
                     reinterpret_cast<const unsigned *>(base + L.vals_off),
                     reinterpret_cast<const unsigned *>(base + L.mask_off), out_vals, out_found);
  return launch_status();
}
