// groupby.hip — dwarf 3: GROUP BY key, SUM(val) into the dense output[key] for gfx950.
//
// Replaces GroupBy's two kernels (groupby/groupby.cpp:58-93: a global open-addressing table of n
// slots hit with one CAS + one fetch_add per row, then n redundant lookups + atomic stores) with the
// privatised design the reference itself sketches in GroupByLocal (groupby/groupby_local.cpp:58-112:
// one private table per executor + merge): here the private table is the workgroup's LDS.
// Logical result identical to expected_GroupBy (groupby/groupby.cpp:8-19), uint32 wrap-around sums.
//
//   gb_aggregate  persistent workgroups stream (key, val) with 16-byte-per-lane loads and add into an
//                 LDS table indexed by key (identity hash = the reference's SimpleHasher key % groups,
//                 hashfunctions.hpp:43-49; keys are < groups by the dense-output contract, so a slot
//                 never collides).  A table holds at most 32768 groups (128 KiB of the 160 KiB LDS):
//                 for more groups the key space is cut into R ranges and workgroup (x, r) aggregates
//                 only range r of chunk x — partner workgroups read the same rows at about the same
//                 time, so the second read is served by L2 / Infinity Cache, not HBM.  Few groups:
//                 the table is replicated across lanes (odd stride) to spread same-address ds_add.
//                 More than 32768 groups (round 3): TWO groups per LDS word, 16 bits each — a 128 KiB
//                 table then holds 65536 groups and BASELINE's 2^16-group configuration reads every row
//                 once instead of twice.  The word is one 32-bit accumulator of L + 65536 * H (L, H: the
//                 sums of the even and the odd group) fed by RETURNING ds_add: from the returned value a
//                 lane sees exactly whether its add carried out of the low half (K16 events) or out of
//                 bit 31 (K32 events), so L = low half + 65536 * K16 and H = high half + 65536 * K32 - K16
//                 (mod 2^32) whatever the interleaving.  The events are rare for value ranges like the
//                 reference's [1, 10000] (a workgroup sees about four rows per group) and go, with the
//                 part of a value above 16 bits, to a spill table in global memory by memory-side
//                 atomics; gb_reduce adds it.  Values that carry on most rows make this path slow
//                 (two global atomics per carry), never wrong.
//   gb_reduce     sums the per-workgroup partial tables into output[] (plain coalesced loads, no
//                 global atomics: memory-side atomics are ~5x slower than stores on this chip).
//
// Algorithmic HBM bytes: 8*n (keys + vals) + 4*groups; partial tables add slots*groups*4*2.
#include "dbhip_common.hpp"

namespace dbhip {
namespace {

constexpr int kGbMaxLdsGroups = 32768;  // 128 KiB table of 32-bit sums
constexpr int kGbMaxPackedGroups = 2 * kGbMaxLdsGroups;  // the same 128 KiB with two 16-bit partial sums per word
constexpr int kGbBigThreads = 1024;     // one workgroup per CU when the table is large
#ifndef DBHIP_GB_VEC
#define DBHIP_GB_VEC 2
#endif
#ifndef DBHIP_GB_PIPE
#define DBHIP_GB_PIPE 1
#endif
constexpr int kGbVecPerIter = DBHIP_GB_VEC;  // uint4 key + uint4 val loads in flight per lane per step (4 or 8: a few
                                             // us either way for one key range, 123 -> 146 us for two: the partner
                                             // workgroups drift apart and lose the shared read)

struct GbHeader {
  unsigned status;
  unsigned pad[63];
};
static_assert(sizeof(GbHeader) == kWsHeader, "workspace header size");

struct GbGeometry {
  unsigned ranges;       // R key ranges
  unsigned range_groups; // groups per range (last may be short)
  unsigned replicas;     // lane-replicated copies of the table (few groups)
  unsigned rep_stride;   // words between copies (odd when replicated)
  unsigned threads;      // workgroup size
  unsigned chunk_slots;  // workgroups per range = partial tables per range
  unsigned lds_words;
  unsigned packed;       // two groups per LDS word (more than kGbMaxLdsGroups groups)
  unsigned part_words;   // words of one partial table
};

inline GbGeometry gb_geometry(uint32_t groups, int cus) {
  GbGeometry g;
  g.packed = groups > static_cast<uint32_t>(kGbMaxLdsGroups) ? 1u : 0u;
#ifdef DBHIP_GB_NO_PACKED  // A/B knob: round 2's two-ranges-of-32-bit-sums path
  g.packed = 0;
#endif
  const uint32_t cap = g.packed ? kGbMaxPackedGroups : kGbMaxLdsGroups;
  g.ranges = (groups + cap - 1) / cap;
  if (g.ranges == 0) g.ranges = 1;
  g.range_groups = (groups + g.ranges - 1) / g.ranges;
  if (g.range_groups == 0) g.range_groups = 1;
  if (g.packed) g.range_groups = (g.range_groups + 1u) & ~1u;  // a word's two groups belong to one range
  // Few groups: same-address ds_add serialises, so the table is replicated and lane l adds into copy
  // l % replicas.  The copies are an ODD number of words apart: a stride that is a multiple of the 32
  // LDS banks (64 groups!) would put the same key of every copy on one bank and undo the spreading.
  g.replicas = 32;
  while (g.replicas > 1 && (g.range_groups | 1u) * g.replicas > static_cast<unsigned>(kGbMaxLdsGroups)) g.replicas /= 2;
  g.rep_stride = g.replicas > 1 ? (g.range_groups | 1u) : g.range_groups;
  g.lds_words = g.rep_stride * g.replicas;
  if (g.packed) {
    g.replicas = 1;
    g.rep_stride = g.range_groups / 2;
    g.lds_words = g.range_groups / 2;
  }
  g.part_words = g.packed ? g.range_groups / 2 : g.range_groups;
  // one 16-wave workgroup per CU for every table size: a few hundred partial tables keep gb_reduce short
  g.threads = kGbBigThreads;
  unsigned total = static_cast<unsigned>(cus);
  g.chunk_slots = total / g.ranges;
  if (g.chunk_slots == 0) g.chunk_slots = 1;
  return g;
}

// the rare path of the packed table: carries and the part of a value above 16 bits go to the spill table
__device__ __noinline__ void gb_spill(unsigned *spill, unsigned groups, unsigned key, unsigned val, bool carry32, bool carry16) {
  const unsigned even = key & ~1u, odd = key | 1u;
  if (val >> 16) atomicAdd(&spill[key], val & 0xFFFF0000u);
  if (carry16) {  // out of the low half: 65536 more for the even group, and the odd group's half holds one too many
    atomicAdd(&spill[even], 65536u);
    if (odd < groups) atomicAdd(&spill[odd], 0xFFFFFFFFu);
  }
  if (carry32 && odd < groups) atomicAdd(&spill[odd], 65536u);  // out of bit 31
}

template <int THREADS, bool kShared, bool PACKED>
__global__ __launch_bounds__(THREADS) void gb_aggregate_kernel(
    const u32x4 *__restrict__ keys4, const u32x4 *__restrict__ vals4, const unsigned *__restrict__ keys,
    const unsigned *__restrict__ vals, size_t n, unsigned groups, GbGeometry geo,
    unsigned *__restrict__ partials, unsigned *spill, GbHeader *hdr) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_table[];
  const unsigned tid = threadIdx.x;
  // blocks b and b+8 tend to share an XCD (round-robin dispatch): give them the same rows and
  // different key ranges so the partner's read hits the XCD's L2.  Placement only affects speed.
  const unsigned b = blockIdx.x;
  unsigned slot, range;
  if (geo.ranges > 1 && (gridDim.x % (8 * geo.ranges)) == 0) {
    range = (b / 8) % geo.ranges;
    slot = (b % 8) + 8 * (b / (8 * geo.ranges));
  } else {
    range = b % geo.ranges;
    slot = b / geo.ranges;
  }
  const unsigned lo = range * geo.range_groups;
  const unsigned long long hi64 = static_cast<unsigned long long>(lo) + geo.range_groups;  // may pass 2^32
  const unsigned hi_excl = hi64 < groups ? static_cast<unsigned>(hi64) : groups;
  const unsigned span = hi_excl > lo ? hi_excl - lo : 0;
  const unsigned rep_off = (tid % geo.replicas) * geo.rep_stride;

  {  // clear the table: 16-byte LDS stores (the array is 16-byte aligned), the odd words at the end singly
    u32x4 *t4 = reinterpret_cast<u32x4 *>(s_table);
    for (unsigned i = tid; i < geo.lds_words / 4; i += THREADS) t4[i] = u32x4{0u, 0u, 0u, 0u};
    for (unsigned i = (geo.lds_words & ~3u) + tid; i < geo.lds_words; i += THREADS) s_table[i] = 0;
  }
  __syncthreads();

  bool bad_key = false;
  const size_t n4 = n / 4;
  // (workgroups take every chunk_slots-th step of the columns; one contiguous slab per workgroup instead, as the reduce
  //  kernel walks its input, measured the same within the +-3 us between runs, with 2 and with 4 loads per column in flight)
  const size_t step = static_cast<size_t>(geo.chunk_slots) * THREADS * kGbVecPerIter;
  // software pipeline: the loads of step i+1 are issued before the LDS atomics of step i (DBHIP_GB_PIPE=0 compiles
  // the plain load-then-add loop for A/B timing)
  auto load_step = [&](size_t base, u32x4 (&k)[kGbVecPerIter], u32x4 (&v)[kGbVecPerIter]) {
#pragma unroll
    for (int u = 0; u < kGbVecPerIter; ++u) {
      const size_t i = base + static_cast<size_t>(u) * THREADS + tid;
      if (i < n4) {
        if (kShared) {  // rows are read again by the partner workgroup: let them live in L2
          k[u] = keys4[i];
          v[u] = vals4[i];
        } else {  // read once
          k[u] = __builtin_nontemporal_load(keys4 + i);
          v[u] = __builtin_nontemporal_load(vals4 + i);
        }
      } else {
        k[u] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        v[u] = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto add_packed = [&](unsigned key, unsigned rel, unsigned val) {
    const unsigned sh = (rel & 1u) << 4;
    const unsigned a = (val & 0xFFFFu) << sh;
    const unsigned old = atomicAdd(&s_table[rel >> 1], a);  // returning: the lane sees what its add did
    const bool carry32 = old + a < old;
    const bool carry16 = sh == 0 && (old & 0xFFFFu) + a > 0xFFFFu;
    if (carry32 || carry16 || (val >> 16)) gb_spill(spill, groups, key, val, carry32, carry16);
  };
  auto add_step = [&](size_t base, const u32x4 (&k)[kGbVecPerIter], const u32x4 (&v)[kGbVecPerIter]) {
#pragma unroll
    for (int u = 0; u < kGbVecPerIter; ++u) {
      const unsigned kk[4] = {k[u].x, k[u].y, k[u].z, k[u].w};
      const unsigned vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      const bool live = base + static_cast<size_t>(u) * THREADS + tid < n4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned rel = kk[c] - lo;
        if (PACKED) {
          if (rel < span) add_packed(kk[c], rel, vv[c]);
        } else if (rel < span) {
          atomicAdd(&s_table[rep_off + rel], vv[c]);
        }
        bad_key |= live && kk[c] >= groups;
      }
    }
  };
  const size_t base0 = static_cast<size_t>(slot) * THREADS * kGbVecPerIter, n4_end = n4;
#if DBHIP_GB_PIPE
  u32x4 ka[kGbVecPerIter], va[kGbVecPerIter], kb[kGbVecPerIter], vb[kGbVecPerIter];
  if (base0 < n4_end) load_step(base0, ka, va);
  for (size_t base = base0; base < n4_end; base += 2 * step) {
    const bool has_b = base + step < n4_end;
    if (has_b) load_step(base + step, kb, vb);
    add_step(base, ka, va);
    if (!has_b) break;
    if (base + 2 * step < n4_end) load_step(base + 2 * step, ka, va);
    add_step(base + step, kb, vb);
  }
#else
  for (size_t base = base0; base < n4_end; base += step) {
    u32x4 k[kGbVecPerIter], v[kGbVecPerIter];
    load_step(base, k, v);
    add_step(base, k, v);
  }
#endif
  // the n % 4 tail rows: first workgroup of every range
  if (slot == 0 && tid < (n & 3)) {
    const unsigned kk = keys[n4 * 4 + tid], vv = vals[n4 * 4 + tid];
    const unsigned rel = kk - lo;
    if (PACKED) {
      if (rel < span) add_packed(kk, rel, vv);
    } else if (rel < span) {
      atomicAdd(&s_table[rep_off + rel], vv);
    }
    bad_key |= kk >= groups;
  }
  if (bad_key && range == 0) atomicOr(&hdr->status, DBHIP_DEV_KEY_RANGE);
  __syncthreads();

  // partial table of this workgroup: partials[range][slot][part_words] (packed: the LDS words as they are)
  unsigned *dst = partials + (static_cast<size_t>(range) * geo.chunk_slots + slot) * geo.part_words;
  if (PACKED) {  // 16-byte non-temporal stores: the table is written once and read once, by gb_reduce
    const unsigned w4 = geo.part_words / 4;  // (dst is 16-byte aligned: part_words is even and the tables start on 256-byte lines when it is a multiple of 4)
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
      const u32x4 *s4 = reinterpret_cast<const u32x4 *>(s_table);
      u32x4 *d4 = reinterpret_cast<u32x4 *>(dst);
      for (unsigned w = tid; w < w4; w += THREADS) __builtin_nontemporal_store(s4[w], d4 + w);
      for (unsigned w = w4 * 4 + tid; w < geo.part_words; w += THREADS) dst[w] = s_table[w];
    } else {
      for (unsigned w = tid; w < geo.part_words; w += THREADS) dst[w] = s_table[w];
    }
    return;
  }
  for (unsigned g = tid; g < geo.range_groups; g += THREADS) {
    unsigned sum = 0;
    for (unsigned r = 0; r < geo.replicas; ++r) sum += s_table[r * geo.rep_stride + g];
    dst[g] = sum;
  }
}

__global__ __launch_bounds__(256) void gb_reduce_kernel(const unsigned *__restrict__ partials,
                                                        GbGeometry geo, unsigned groups,
                                                        unsigned *__restrict__ out) {
  // 64 consecutive groups per workgroup (one 256-B line per partial table), the four waves split the
  // partial tables between them; 8 independent loads in flight per lane
  __shared__ unsigned s_sum[4][kWave];
  const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const unsigned g = blockIdx.x * kWave + lane;
  unsigned sum = 0;
  if (g < groups) {
    const unsigned range = g / geo.range_groups, rel = g % geo.range_groups;
    const unsigned *p = partials + static_cast<size_t>(range) * geo.chunk_slots * geo.range_groups + rel;
    unsigned s = wave;
    for (; s + 28 < geo.chunk_slots; s += 32) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[static_cast<size_t>(s + 4 * u) * geo.range_groups];
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; s < geo.chunk_slots; s += 4) sum += p[static_cast<size_t>(s) * geo.range_groups];
  }
  s_sum[wave][lane] = sum;
  __syncthreads();
  if (wave == 0 && g < groups) out[g] = s_sum[0][lane] + s_sum[1][lane] + s_sum[2][lane] + s_sum[3][lane];
}

// packed partial tables: one thread per WORD = two groups; the spill table holds the carries and the upper value bits
// (sixteen waves per workgroup split the partial tables: with four, 32 MiB of tables were read by 8 waves per CU with 8
//  loads each in flight — 9.0 us, latency-bound)
constexpr int kGbRedWaves = 16;
__global__ __launch_bounds__(kGbRedWaves * kWave) void gb_reduce_packed_kernel(const unsigned *__restrict__ partials,
                                                                               const unsigned *__restrict__ spill, GbGeometry geo,
                                                                               unsigned groups, unsigned *__restrict__ out) {
  __shared__ unsigned s_lo[kGbRedWaves][kWave], s_hi[kGbRedWaves][kWave];
  const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const unsigned word = blockIdx.x * kWave + lane;  // over ranges * part_words
  const unsigned range = word / geo.part_words, rel = word % geo.part_words;
  const unsigned g0 = range * geo.range_groups + 2 * rel;
  unsigned lo = 0, hi = 0;
  if (range < geo.ranges && g0 < groups) {
    const unsigned *p = partials + static_cast<size_t>(range) * geo.chunk_slots * geo.part_words + rel;
    unsigned s = wave;
    for (; s + 7 * kGbRedWaves < geo.chunk_slots; s += 8 * kGbRedWaves) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + static_cast<size_t>(s + kGbRedWaves * u) * geo.part_words);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        lo += v[u] & 0xFFFFu;
        hi += v[u] >> 16;
      }
    }
    for (; s < geo.chunk_slots; s += kGbRedWaves) {
      const unsigned v = p[static_cast<size_t>(s) * geo.part_words];
      lo += v & 0xFFFFu;
      hi += v >> 16;
    }
  }
  s_lo[wave][lane] = lo;
  s_hi[wave][lane] = hi;
  __syncthreads();
  if (wave == 0 && range < geo.ranges && g0 < groups) {
    unsigned tl = spill[g0], th = 0;
#pragma unroll
    for (int w = 0; w < kGbRedWaves; ++w) {
      tl += s_lo[w][lane];
      th += s_hi[w][lane];
    }
    out[g0] = tl;
    if (g0 + 1 < groups && 2 * rel + 1 < geo.range_groups) out[g0 + 1] = th + spill[g0 + 1];
  }
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

namespace {
// workspace: header | spill[groups] (packed tables only; cleared with the header) | partial tables
size_t gb_spill_bytes(const GbGeometry &geo, uint32_t groups) {
  return geo.packed ? align_up(static_cast<size_t>(groups) * sizeof(unsigned), kWsAlign) : 0;
}
// geometry for a launch: `max_tables` (0 = no limit) caps the number of private tables per key range —
// the reference's `executors` knob of GroupByLocal (groupby/groupby_local.cpp:27, :58-83)
GbGeometry gb_launch_geometry(uint32_t groups, uint32_t max_tables, int cus) {
  GbGeometry geo = gb_geometry(groups, cus < 256 ? cus : 256);
  if (max_tables && max_tables < geo.chunk_slots) geo.chunk_slots = max_tables;
  return geo;
}

int gb_partial(const uint32_t *keys, const uint32_t *vals, size_t n, uint32_t groups, uint32_t max_tables,
               void *workspace, size_t workspace_bytes, dbhip_stream_t stream) {
  if (groups == 0) {
    if (n) return DBHIP_EINVAL;
    if (workspace && ws_ok(workspace, workspace_bytes, kWsHeader))  // a clean status word even when nothing runs
      return static_cast<int>(fill_async(workspace, 0, kWsHeader, as_stream(stream)));
    return DBHIP_OK;
  }
  if (n && (!keys || !vals)) return DBHIP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(keys) | reinterpret_cast<uintptr_t>(vals)) & 15u) return DBHIP_EINVAL;  // dbhip.h: 16-byte aligned
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  const GbGeometry geo = gb_launch_geometry(groups, max_tables, dev.cus);
  const size_t partial_words = static_cast<size_t>(geo.ranges) * geo.chunk_slots * geo.part_words;
  const size_t spill_bytes = gb_spill_bytes(geo, groups);
  if (!ws_ok(workspace, workspace_bytes, kWsHeader + spill_bytes + partial_words * sizeof(unsigned))) return DBHIP_EWORKSPACE;
  hipStream_t s = as_stream(stream);
  hipError_t e = fill_async(workspace, 0, kWsHeader + spill_bytes, s);
  if (e != hipSuccess) return static_cast<int>(e);
  GbHeader *hdr = static_cast<GbHeader *>(workspace);
  unsigned *spill = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + kWsHeader);
  unsigned *partials = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + kWsHeader + spill_bytes);
  const unsigned grid = geo.ranges * geo.chunk_slots;
  const size_t lds = static_cast<size_t>(geo.lds_words) * sizeof(unsigned);
  const u32x4 *k4 = reinterpret_cast<const u32x4 *>(keys), *v4 = reinterpret_cast<const u32x4 *>(vals);
  auto launch = [&](auto kernel) -> int {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (kGbMaxLdsGroups + 32) * 4);
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kGbBigThreads), lds, s, k4, v4, keys, vals, n, groups, geo, partials, spill,
                       hdr);
    return launch_status();
  };
  // measured: plain (L2-allocating) loads win from 4 readers per row on, nt below
  if (geo.packed) return geo.ranges > 2 ? launch(gb_aggregate_kernel<kGbBigThreads, true, true>)
                                        : launch(gb_aggregate_kernel<kGbBigThreads, false, true>);
  return geo.ranges > 2 ? launch(gb_aggregate_kernel<kGbBigThreads, true, false>)
                        : launch(gb_aggregate_kernel<kGbBigThreads, false, false>);
}

int gb_merge(uint32_t groups, uint32_t max_tables, uint32_t *out, const void *workspace, dbhip_stream_t stream) {
  if (groups == 0) return DBHIP_OK;
  if (!out || !workspace) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  const GbGeometry geo = gb_launch_geometry(groups, max_tables, dev.cus);
  const size_t spill_bytes = gb_spill_bytes(geo, groups);
  const unsigned *spill = reinterpret_cast<const unsigned *>(static_cast<const char *>(workspace) + kWsHeader);
  const unsigned *partials = reinterpret_cast<const unsigned *>(static_cast<const char *>(workspace) + kWsHeader + spill_bytes);
  if (geo.packed) {
    const size_t words = static_cast<size_t>(geo.ranges) * geo.part_words;
    hipLaunchKernelGGL(gb_reduce_packed_kernel, dim3(static_cast<unsigned>((words + kWave - 1) / kWave)),
                       dim3(kGbRedWaves * kWave), 0, as_stream(stream), partials, spill, geo, groups, out);
  } else {
    hipLaunchKernelGGL(gb_reduce_kernel, dim3((groups + kWave - 1) / kWave), dim3(256), 0, as_stream(stream), partials,
                       geo, groups, out);
  }
  return launch_status();
}
}  // namespace

extern "C" size_t dbhip_groupby_sum_u32_workspace_bytes(size_t n, uint32_t groups) {
  (void)n;
  // sized for the largest device this library targets (256 CUs) so the query needs no device
  const GbGeometry g = gb_geometry(groups ? groups : 1, 256);
  const size_t partial_words = static_cast<size_t>(g.ranges) * g.chunk_slots * g.part_words;
  return align_up(kWsHeader + gb_spill_bytes(g, groups ? groups : 1) + partial_words * sizeof(unsigned), kWsAlign);
}

extern "C" int dbhip_groupby_sum_u32(const uint32_t *keys, const uint32_t *vals, size_t n,
                                     uint32_t groups, uint32_t *out, void *workspace,
                                     size_t workspace_bytes, dbhip_stream_t stream) {
  if (groups && !out) return DBHIP_EINVAL;
  const int rc = gb_partial(keys, vals, n, groups, 0, workspace, workspace_bytes, stream);
  return rc != 0 ? rc : gb_merge(groups, 0, out, workspace, stream);
}

// the two phases separately (GroupByLocal reports them separately, groupby/groupby_local.cpp:115-119)
extern "C" int dbhip_groupby_partial_u32(const uint32_t *keys, const uint32_t *vals, size_t n, uint32_t groups,
                                         uint32_t max_private_tables, void *workspace, size_t workspace_bytes,
                                         dbhip_stream_t stream) {
  return gb_partial(keys, vals, n, groups, max_private_tables, workspace, workspace_bytes, stream);
}

extern "C" int dbhip_groupby_merge_u32(uint32_t groups, uint32_t max_private_tables, uint32_t *out,
                                       const void *workspace, dbhip_stream_t stream) {
  return gb_merge(groups, max_private_tables, out, workspace, stream);
}
