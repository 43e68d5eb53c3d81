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
//                 never collides).  A table of 32-bit sums holds at most 32768 groups (128 KiB of the 160 KiB
//                 LDS).  Few groups: the table is replicated across lanes (odd stride) to spread same-address
//                 ds_add.  MORE than 32768 groups, two ways, chosen by the kernel itself (gb_aggregate_big):
//                 * WIDE: the key space is cut into R ranges of <= 32768 groups and workgroup (x, r) aggregates only
//                   range r of chunk x — partner workgroups read the same rows at about the same time, so the second
//                   read is served by L2 / Infinity Cache, not HBM.  Any values, any skew; every row read R times.
//                 * PACKED (round 3): TWO groups per LDS word, 16 bits each, so 128 KiB hold 65536 groups and
//                   BASELINE's 2^16-group configuration reads every row ONCE.  The word is one 32-bit accumulator
//                   of L + 65536 * H (L, H: the sums of the even and the odd group) fed by RETURNING ds_add: from
//                   the returned value a lane sees exactly whether its add carried out of the low half (K16
//                   events) or out of bit 31 (K32 events), whatever the interleaving, so L = low half + 65536 * K16
//                   and H = high half + 65536 * K32 - K16 (mod 2^32).  The events are counted in LDS too — two
//                   4-bit counters per word behind the table (32 KiB: with the table the whole 160 KiB) bumped by
//                   a compare-and-swap loop that saturates at 15 — and only a saturated counter and the part of a
//                   value above 16 bits go to a spill table in global memory (memory-side atomics, ~40 ns each).
//                   That is exact for every input and fast while a workgroup's share of a group stays below
//                   ~10^6 and values below 65536; beyond that it falls off a cliff (measured before the LDS
//                   counters and the mode choice existed: 2^28 rows 1024 us against 423 for WIDE, values up to
//                   60000 1431 against 119, full-range values 4325 against 120).  So every workgroup first looks
//                   at the SAME 12288 rows of the columns (start, middle, end: identical decision everywhere, no
//                   communication): the largest and the mean value, the largest multiplicity of a key, and with
//                   the rows a workgroup will see per group it takes PACKED only if no sampled value reaches
//                   65536, no key is hot and the expected partial sum stays below 12 * 65536; the mode goes into
//                   the workspace header for gb_reduce.  A column whose unsampled rows break the prediction is
//                   still summed exactly, through the saturated counters' global path.
//   gb_reduce     sums the per-workgroup partial tables into output[] (plain coalesced loads, no
//                 global atomics: memory-side atomics are ~5x slower than stores on this chip).
//
// Algorithmic HBM bytes: 8*n (keys + vals) + 4*groups; partial tables add slots*groups*4*2.
#include <cstdlib>

#include "dbhip_common.hpp"

namespace dbhip {
namespace {

constexpr int kGbMaxLdsGroups = 32768;  // 128 KiB table of 32-bit sums
constexpr int kGbMaxPackedGroups = 2 * kGbMaxLdsGroups;  // the same 128 KiB with two 16-bit partial sums per word
constexpr int kGbCarryWords = kGbMaxLdsGroups / 4;       // one byte of carry counters per table word: 32 KiB
constexpr int kGbBigThreads = 1024;     // one workgroup per CU when the table is large
#ifndef DBHIP_GB_CROWD
#define DBHIP_GB_CROWD 24
#endif
constexpr int kGbCrowd = DBHIP_GB_CROWD;  // lanes of a wave on one group from which they are summed before the LDS add
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
  unsigned mode;  // more than kGbMaxLdsGroups groups: what gb_aggregate_big chose (kGbModePacked / kGbModeWide)
  unsigned pad[62];
};
static_assert(sizeof(GbHeader) == kWsHeader, "workspace header size");
constexpr unsigned kGbModePacked = 1, kGbModeWide = 2;

struct GbGeometry {
  unsigned ranges;       // R key ranges
  unsigned range_groups; // groups per range (last may be short)
  unsigned replicas;     // lane-replicated copies of the table (few groups)
  unsigned rep_stride;   // words between copies (odd when replicated)
  unsigned threads;      // workgroup size
  unsigned chunk_slots;  // workgroups per range = partial tables per range
  unsigned lds_words;
  unsigned packed;       // two groups per LDS word
  unsigned part_words;   // words of one partial table
};

inline GbGeometry gb_geometry(uint32_t groups, int cus, bool packed) {
  GbGeometry g;
  g.packed = packed ? 1u : 0u;
  const uint32_t cap = g.packed ? kGbMaxPackedGroups : kGbMaxLdsGroups;
  g.ranges = (groups + cap - 1) / cap;
  if (g.ranges == 0) g.ranges = 1;
  g.range_groups = (groups + g.ranges - 1) / g.ranges;
  if (g.range_groups == 0) g.range_groups = 1;
  if (g.packed) g.range_groups = (g.range_groups + 7u) & ~7u;  // whole carry words; a word's two groups belong to one range
  // Few groups: same-address ds_add serialises, so the table is replicated and lane l adds into copy
  // l % replicas.  The copies are an ODD number of words apart: a stride that is a multiple of the 32
  // LDS banks (64 groups!) would put the same key of every copy on one bank and undo the spreading.
  g.replicas = 32;
  while (g.replicas > 1 && (g.range_groups | 1u) * g.replicas > static_cast<unsigned>(kGbMaxLdsGroups)) g.replicas /= 2;
  g.rep_stride = g.replicas > 1 ? (g.range_groups | 1u) : g.range_groups;
  g.lds_words = g.rep_stride * g.replicas;
  g.part_words = g.range_groups;
  if (g.packed) {
    g.replicas = 1;
    g.rep_stride = g.range_groups / 2;
    g.lds_words = g.range_groups / 2;
    g.part_words = g.range_groups / 2;  // the table words (the carry counters are settled in the spill table at the flush)
  }
  // one 16-wave workgroup per CU for every table size: a few hundred partial tables keep gb_reduce short
  g.threads = kGbBigThreads;
  unsigned total = static_cast<unsigned>(cus);
  g.chunk_slots = total / g.ranges;
  if (g.chunk_slots == 0) g.chunk_slots = 1;
  return g;
}

// the rare paths of the packed table: the part of a value above 16 bits, and k16 / k32 carry events that the LDS
// counters handed back (sixteen at a time), go to the spill table in global memory
__device__ __noinline__ void gb_spill_global(unsigned *spill, unsigned groups, unsigned key, unsigned wide_bits, unsigned k32,
                                             unsigned k16) {
  const unsigned even = key & ~1u, odd = key | 1u;
  if (wide_bits) atomicAdd(&spill[key], wide_bits);
  if (k16) {  // out of the low half: 65536 more each for the even group, and the odd group's half holds one too many each
    atomicAdd(&spill[even], k16 << 16);
    if (odd < groups) atomicAdd(&spill[odd], 0u - k16);
  }
  if (k32 && odd < groups) atomicAdd(&spill[odd], k32 << 16);  // out of bit 31
}
// one more carry event of table word `w` (which = 0: out of the low half, 1: out of bit 31) in its 4-bit LDS counter.
// A counter at 15 is DRAINED instead: set back to 0, and the caller settles sixteen events (its own and the fifteen
// counted) with one global operation — a hot group costs one memory-side atomic per sixteen carries, not one per carry.
// Returns 0 (counted) or 16 (drained: settle sixteen events globally).
__device__ __forceinline__ unsigned gb_bump_carry(unsigned *s_carry, unsigned w, unsigned which) {
  unsigned *cw = s_carry + (w >> 2);
  const unsigned sh = ((w & 3u) << 3) + (which << 2);
  unsigned old = *cw;
  while (true) {
    const bool full = ((old >> sh) & 15u) == 15u;
    const unsigned want = full ? old & ~(15u << sh) : old + (1u << sh);
    const unsigned prev = atomicCAS(cw, old, want);
    if (prev == old) return full ? 16u : 0u;
    old = prev;
  }
}

// The aggregation of one workgroup.  PACKED: s_table = range_groups / 2 table words followed by range_groups / 8 carry
// words; otherwise lds_words 32-bit sums.  shared_rows: the rows are read again by partner workgroups (plain loads, let
// them live in L2); otherwise non-temporal loads.
template <int THREADS, bool PACKED, bool kShared>
__device__ __forceinline__ void gb_aggregate_body(const u32x4 *__restrict__ keys4, const u32x4 *__restrict__ vals4,
                                                  const unsigned *__restrict__ keys, const unsigned *__restrict__ vals, size_t n,
                                                  unsigned groups, const GbGeometry &geo,
                                                  unsigned *__restrict__ partials, unsigned *spill, GbHeader *hdr,
                                                  unsigned *s_table) {
  const unsigned tid = threadIdx.x;
  // blocks b and b+8 tend to share an XCD (round-robin dispatch): give them the same rows and
  // different key ranges so the partner's read hits the XCD's L2.  Placement only affects speed.
  const unsigned b = blockIdx.x;
  if (b >= geo.ranges * geo.chunk_slots) return;  // (the big kernel's grid covers both modes)
  unsigned slot, range;
  if (geo.ranges > 1 && (gridDim.x % (8 * geo.ranges)) == 0 && gridDim.x == geo.ranges * geo.chunk_slots) {
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
  unsigned *s_carry = s_table + geo.lds_words;  // PACKED only
  const unsigned clear_words = PACKED ? geo.lds_words + geo.range_groups / 8 : geo.lds_words;

  {  // clear the table: 16-byte LDS stores (the array is 16-byte aligned), the odd words at the end singly
    u32x4 *t4 = reinterpret_cast<u32x4 *>(s_table);
    for (unsigned i = tid; i < clear_words / 4; i += THREADS) t4[i] = u32x4{0u, 0u, 0u, 0u};
    for (unsigned i = (clear_words & ~3u) + tid; i < clear_words; i += THREADS) s_table[i] = 0;
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
    if (carry32 || carry16 || (val >> 16)) {  // rare
      const unsigned k16 = carry16 ? gb_bump_carry(s_carry, rel >> 1, 0u) : 0u;
      const unsigned k32 = carry32 ? gb_bump_carry(s_carry, rel >> 1, 1u) : 0u;
      if (k16 || k32 || (val >> 16)) gb_spill_global(spill, groups, key, val & 0xFFFF0000u, k32, k16);
    }
  };
  // Rows of ONE group in many lanes of a wave (a hot key, keys that come in runs, sorted keys): the LDS serves the lanes of
  // an atomic that meet on one word one after the other.  Up to 16 lanes per word that hides behind the HBM reads (one key
  // in all 2^26 rows, 4096 groups in 4 lane copies: 97 us against 94 for uniform keys), 32 and 64 do not (16384 and 32768
  // groups, one copy: 244-251 us against 100-107; 65536 groups 471 against 111).  So where the table has fewer than four
  // lane copies every row of keys asks how many lanes share the first lane's group, and kGbCrowd or more are summed
  // across the wave and added once (one key in all rows: 107-114 us; the test on uniform keys: nothing measurable; with
  // the test on tables of 16 copies, or crowds from 8 lanes: 92 -> 104 us, the wave sum costs more than the queue).
  const bool crowd_guard = !PACKED && geo.replicas <= 2;
  unsigned crowd_misses = 0, crowd_tick = 0;  // steps in a row that found no crowd; steps since
  auto add_step = [&](size_t base, const u32x4 (&k)[kGbVecPerIter], const u32x4 (&v)[kGbVecPerIter]) {
    // The question is asked of a step's FIRST row of keys — a hot or clustered key shows in every row alike — and only
    // of every fourth step once four steps in a row have said no; a step that finds a crowd there asks it of each of its
    // eight rows.  Asked of every row it cost the multi-range tables, whose workgroups walk every row
    // of the columns with four or five instructions each, half their speed on uniform keys (2^24 rows into 100000
    // groups: 57 -> 87 us); once per step still 20-30 % with four and eight ranges.
    bool step_guard = false;
    if (crowd_guard && (crowd_misses < 4 || (++crowd_tick & 3u) == 0)) {  // (uniform; every lane of the wave is here)
      const unsigned rel0 = k[0].x - lo;
      const bool in0 = rel0 < span;
      const unsigned long long act = __ballot(in0);
      const unsigned first = __builtin_amdgcn_readlane(rel0, act ? __builtin_ctzll(act) : 0);
      step_guard = __builtin_popcountll(__ballot(in0 && rel0 == first)) >= kGbCrowd;
      crowd_misses = step_guard ? 0u : crowd_misses + 1u;
    }
    if (!step_guard) {  // (uniform) the step as it always was: straight-line code, eight independent LDS adds
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
      return;
    }
    // a step with a crowd in its first row (never PACKED): every row sums its crowd across the wave and adds it once
    // (with the question inside the loop above — one branch per row, never taken — the multi-range tables lost 20-30 %
    //  on uniform keys however rarely the question was asked: the eight adds of a step were no longer issued together)
#pragma unroll
    for (int u = 0; u < kGbVecPerIter; ++u) {
      const unsigned kk[4] = {k[u].x, k[u].y, k[u].z, k[u].w};
      const unsigned vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      const bool live = base + static_cast<size_t>(u) * THREADS + tid < n4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned rel = kk[c] - lo;
        bad_key |= live && kk[c] >= groups;
        const bool in = rel < span;
        const unsigned long long act = __ballot(in);
        const unsigned first = __builtin_amdgcn_readlane(rel, act ? __builtin_ctzll(act) : 0);
        const bool same = in && rel == first;
        const unsigned long long crowd = __ballot(same);
        if (__builtin_popcountll(crowd) >= kGbCrowd) {
          const unsigned sum = wave_reduce_add(same ? vv[c] : 0u);
          if ((tid & (kWave - 1)) == static_cast<unsigned>(__builtin_ctzll(crowd))) atomicAdd(&s_table[rep_off + first], sum);
          if (in && !same) atomicAdd(&s_table[rep_off + rel], vv[c]);
        } else if (in) {
          atomicAdd(&s_table[rep_off + rel], vv[c]);
        }
      }
    }
  };
  const size_t base0 = static_cast<size_t>(slot) * THREADS * kGbVecPerIter;
#if DBHIP_GB_PIPE
  u32x4 ka[kGbVecPerIter], va[kGbVecPerIter], kb[kGbVecPerIter], vb[kGbVecPerIter];
  // (Round 4, measured and removed: the row steps of a one-range table handed out by TICKET — batches of four steps, the
  //  next batch asked for one batch ahead by thread 0 and passed round through the workgroup's own partial-table slot in
  //  global memory, since the packed table leaves no LDS word free.  Same box, 2^26 rows, three interleaved runs: 65536
  //  groups 116.0-117.3 us against 115.8-116.8 static; 32768 groups 106.5-108.9 against 105.3-107.0; 1024 groups 101.5-102.1
  //  against 95.6-97.9; 64 groups 100.3-101.8 against 92.9-98.7.  What ticketing gave the join's build kernels — workgroups
  //  that are not equally fast — is not what this stream loses time to.)
  {
  if (base0 < n4) load_step(base0, ka, va);
  for (size_t base = base0; base < n4; base += 2 * step) {
    const bool has_b = base + step < n4;
    if (has_b) load_step(base + step, kb, vb);
    add_step(base, ka, va);
    if (!has_b) break;
    if (base + 2 * step < n4) load_step(base + 2 * step, ka, va);
    add_step(base + step, kb, vb);
  }
  }
#else
  for (size_t base = base0; base < n4; base += step) {
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

  // partial table of this workgroup: partials[range][slot][part_words] (packed: the table's words as they are)
  unsigned *dst = partials + (static_cast<size_t>(range) * geo.chunk_slots + slot) * geo.part_words;
  if (PACKED) {
    // the carry counters first: a word that carried (about one in a thousand where this mode is chosen) settles its events
    // in the spill table, so the partial table stays the 128 KiB of words (with one carry byte per word in the partial
    // tables gb_reduce read 40 MiB with two loads per table: 11.3 us against 7.3)
    for (unsigned ci = tid; ci < geo.range_groups / 8; ci += THREADS) {
      const unsigned cw = s_carry[ci];
      if (cw != 0)
        for (unsigned b4 = 0; b4 < 4; ++b4) {
          const unsigned byte = (cw >> (8 * b4)) & 0xFFu;
          if (byte) gb_spill_global(spill, groups, lo + 2 * (4 * ci + b4), 0u, byte >> 4, byte & 15u);
        }
    }
    // 16-byte non-temporal stores: the table is written once and read once, by gb_reduce
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
      const unsigned w4 = geo.part_words / 4;
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

// at most kGbMaxLdsGroups groups: one table of 32-bit sums per workgroup
template <int THREADS>
__global__ __launch_bounds__(THREADS) void gb_aggregate_kernel(
    const u32x4 *__restrict__ keys4, const u32x4 *__restrict__ vals4, const unsigned *__restrict__ keys,
    const unsigned *__restrict__ vals, size_t n, unsigned groups, GbGeometry geo,
    unsigned *__restrict__ partials, GbHeader *hdr) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_table[];
  gb_aggregate_body<THREADS, false, false>(keys4, vals4, keys, vals, n, groups, geo, partials, nullptr, hdr, s_table);
}

// more than kGbMaxLdsGroups groups: every workgroup looks at the same sample of the columns, takes the same decision
// (see the file header) and runs the packed or the wide aggregation; may_pack = 0: the device cannot give a workgroup
// the 160 KiB the packed table and its carry counters need (or DBHIP_GB_PACKED=0)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void gb_aggregate_big_kernel(
    const u32x4 *__restrict__ keys4, const u32x4 *__restrict__ vals4, const unsigned *__restrict__ keys,
    const unsigned *__restrict__ vals, size_t n, unsigned groups, GbGeometry geo_packed, GbGeometry geo_wide, unsigned may_pack,
    unsigned *__restrict__ partials, unsigned *spill, GbHeader *hdr) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_table[];
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1);
  const size_t n4 = n / 4;
  // The decision, by wave 0 of every workgroup on the SAME 768 rows (three windows of 64 vectors: start, middle, end of
  // the columns — identical everywhere, no communication; the first version had all sixteen waves look at 12288 rows:
  // 12-15 us of every call, 256 workgroups on the same lines).  Scratch: the carry counters' LDS words.
  unsigned *s_dec = s_table + kGbMaxLdsGroups;          // [0, 1024): multiplicities of the sampled keys by key % 1024
  unsigned &s_flag = s_table[kGbMaxLdsGroups + 1024];   // the decision, for the other waves
  // may_pack == 0: the launch carries only the wide table's LDS (kGbMaxLdsGroups + 32 words) — s_dec and s_flag do not
  // exist then and are not touched (the argument is uniform over the grid, so are the barriers below)
  if (may_pack == 0) {
  } else if (may_pack == 2) {  // DBHIP_GB_PACKED=force (tests: every packed path whatever the sample would say)
    if (tid == 0) s_flag = 1;
  } else if (n4 < 3) {
    if (tid == 0) s_flag = 0;
  } else if (tid < kWave) {
    for (unsigned i = lane; i < 1024; i += kWave) s_dec[i] = 0;
    const size_t at[3] = {0, n4 / 2, n4 > kWave ? n4 - kWave : 0};
    unsigned sk[12], vmax = 0, vsum = 0, live_rows = 0;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const size_t i = at[p] + lane;
      const bool live = i < n4;
      const u32x4 k = live ? keys4[i] : u32x4{0u, 0u, 0u, 0u}, v = live ? vals4[i] : u32x4{0u, 0u, 0u, 0u};
      sk[4 * p + 0] = k.x, sk[4 * p + 1] = k.y, sk[4 * p + 2] = k.z, sk[4 * p + 3] = k.w;
      if (live) {
        const unsigned m0 = v.x > v.y ? v.x : v.y, m1 = v.z > v.w ? v.z : v.w;
        const unsigned m = m0 > m1 ? m0 : m1;
        vmax = vmax > m ? vmax : m;
        vsum += (v.x >> 8) + (v.y >> 8) + (v.z >> 8) + (v.w >> 8);  // in units of 256: 768 values cannot overflow
        live_rows += 4;
      }
    }
    // (one wave: its LDS operations execute in program order — no barrier between the three steps)
#pragma unroll
    for (int q = 0; q < 12; ++q)
      if (at[q / 4] + lane < n4) atomicAdd(&s_dec[sk[q] & 1023u], 1u);
    unsigned kmax = 0;
#pragma unroll
    for (int q = 0; q < 12; ++q)
      if (at[q / 4] + lane < n4) {
        const unsigned m = s_dec[sk[q] & 1023u];
        kmax = kmax > m ? kmax : m;
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned ov = __shfl_xor(vmax, off, kWave), ok = __shfl_xor(kmax, off, kWave);
      vmax = vmax > ov ? vmax : ov;
      kmax = kmax > ok ? kmax : ok;
    }
    vsum = wave_reduce_add(vsum);
    live_rows = wave_reduce_add(live_rows);
    // rows a workgroup sees per group if the keys are uniform (lambda), their mean value mu: a partial sum has mean
    // lambda * mu and a variance below lambda * vmax * mu.  PACKED wants carries to be rare PER ROW — a wave whose lanes
    // carry takes the slow branch as a whole (2^27 rows, lambda = 8: 242 us against 213 for WIDE) — so the mean plus
    // three standard deviations must stay inside the 16 bits; no value may be wider than 16 bits; and no key may be hot
    // or clustered (768 uniform keys over >= 32768 groups hit one of 1024 counters 0.75 times on average: a counter at 8
    // is a key with ~1 % of the rows, or keys that come in runs — a workgroup would see hundreds of rows of one group).
    const float mu = 256.0f * static_cast<float>(vsum) / static_cast<float>(live_rows ? live_rows : 1u);
    const float lambda = static_cast<float>(n) / (static_cast<float>(geo_packed.chunk_slots) * static_cast<float>(geo_packed.range_groups));
    const float top = lambda * mu + 3.0f * __builtin_sqrtf(lambda * static_cast<float>(vmax) * (mu + 1.0f));
    // and the input must be long enough for the saved second read to pay for the larger partial tables (table + carry
    // counters, two loads per table in gb_reduce): 2^24 rows 39 us packed against 34.5 us wide, 2^26 rows 103 against 110
    const bool ok = n >= (static_cast<size_t>(1) << 25) && vmax < 65536u && kmax < 8u && top < 65536.0f;
    if (lane == 0) s_flag = ok ? 1u : 0u;
  }
  bool packed = false;
  if (may_pack != 0) {
    __syncthreads();
    packed = s_flag != 0;
    __syncthreads();  // the flag is one of the words the aggregation clears next
  }
  if (blockIdx.x == 0 && tid == 0) hdr->mode = packed ? kGbModePacked : kGbModeWide;
  // (measured: plain, L2-allocating loads win from 4 readers per row on, non-temporal ones below)
  if (packed) {
    if (geo_packed.ranges > 2)
      gb_aggregate_body<THREADS, true, true>(keys4, vals4, keys, vals, n, groups, geo_packed, partials, spill, hdr, s_table);
    else
      gb_aggregate_body<THREADS, true, false>(keys4, vals4, keys, vals, n, groups, geo_packed, partials, spill, hdr, s_table);
  } else {
    if (geo_wide.ranges > 2)
      gb_aggregate_body<THREADS, false, true>(keys4, vals4, keys, vals, n, groups, geo_wide, partials, spill, hdr, s_table);
    else
      gb_aggregate_body<THREADS, false, false>(keys4, vals4, keys, vals, n, groups, geo_wide, partials, spill, hdr, s_table);
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

// More than kGbMaxLdsGroups groups: the partial tables are in the layout of the mode gb_aggregate_big chose (header).
// One workgroup of sixteen waves per 128 consecutive groups; the waves split the partial tables (with four waves, 32 MiB
// of tables were read by 8 waves per CU with 8 loads each in flight: 9.0 us, latency-bound; sixteen: 7.3 us).
//   packed: lane = table word = two groups; even += low halves, odd += high halves, plus the spill table (which holds
//           the carries: 65536 * K16 for the even group, 65536 * K32 - K16 for the odd one)
//   wide:   waves 0-7 take the first 64 groups, waves 8-15 the next 64; 32-bit sums
constexpr int kGbRedWaves = 16;
__global__ __launch_bounds__(kGbRedWaves * kWave) void gb_reduce_big_kernel(const unsigned *__restrict__ partials,
                                                                            const unsigned *__restrict__ spill,
                                                                            const GbHeader *hdr, GbGeometry geo_packed,
                                                                            GbGeometry geo_wide, unsigned groups,
                                                                            unsigned *__restrict__ out) {
  __shared__ unsigned s_lo[kGbRedWaves][kWave], s_hi[kGbRedWaves][kWave];
  const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (hdr->mode == kGbModePacked) {
    const GbGeometry &geo = geo_packed;
    const unsigned table_words = geo.range_groups / 2;
    const unsigned word = blockIdx.x * kWave + lane;  // over ranges * table_words
    const unsigned range = word / table_words, rel = word % table_words;
    const unsigned g0 = range * geo.range_groups + 2 * rel;
    unsigned lo = 0, hi = 0;
    if (range < geo.ranges && g0 < groups) {
      const unsigned *p = partials + static_cast<size_t>(range) * geo.chunk_slots * geo.part_words + rel;
      unsigned s = wave;
      for (; s + 7 * kGbRedWaves < geo.chunk_slots; s += 8 * kGbRedWaves) {
        unsigned w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load(p + static_cast<size_t>(s + kGbRedWaves * u) * geo.part_words);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          lo += w[u] & 0xFFFFu;
          hi += w[u] >> 16;
        }
      }
      for (; s < geo.chunk_slots; s += kGbRedWaves) {
        const unsigned w = p[static_cast<size_t>(s) * geo.part_words];
        lo += w & 0xFFFFu;
        hi += w >> 16;
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
      if (g0 + 1 < groups) out[g0 + 1] = th + spill[g0 + 1];
    }
    return;
  }
  const GbGeometry &geo = geo_wide;
  const unsigned half = wave / (kGbRedWaves / 2), sub = wave % (kGbRedWaves / 2);
  const unsigned g = blockIdx.x * 2 * kWave + half * kWave + lane;
  unsigned sum = 0;
  if (g < groups) {
    const unsigned range = g / geo.range_groups, rel = g % geo.range_groups;
    const unsigned *p = partials + static_cast<size_t>(range) * geo.chunk_slots * geo.part_words + rel;
    unsigned s = sub;
    for (; s + 56 < geo.chunk_slots; s += 64) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[static_cast<size_t>(s + 8 * u) * geo.part_words];
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; s < geo.chunk_slots; s += 8) sum += p[static_cast<size_t>(s) * geo.part_words];
  }
  s_lo[wave][lane] = sum;
  __syncthreads();
  if (sub == 0 && g < groups) {
    unsigned t = spill[g];  // (zero in this mode: cleared with the header)
#pragma unroll
    for (int w = 0; w < kGbRedWaves / 2; ++w) t += s_lo[half * (kGbRedWaves / 2) + w][lane];
    out[g] = t;
  }
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

namespace {
// geometry for a launch: `max_tables` (0 = no limit) caps the number of private tables per key range —
// the reference's `executors` knob of GroupByLocal (groupby/groupby_local.cpp:27, :58-83)
GbGeometry gb_launch_geometry(uint32_t groups, uint32_t max_tables, int cus, bool packed) {
  GbGeometry geo = gb_geometry(groups, cus < 256 ? cus : 256, packed);
  if (max_tables && max_tables < geo.chunk_slots) geo.chunk_slots = max_tables;
  return geo;
}
bool gb_big(uint32_t groups) { return groups > static_cast<uint32_t>(kGbMaxLdsGroups); }
// workspace of the big path: header | spill[groups] | partial tables (the larger of the two modes' layouts)
size_t gb_spill_bytes(uint32_t groups) {
  return gb_big(groups) ? align_up(static_cast<size_t>(groups) * sizeof(unsigned), kWsAlign) : 0;
}
size_t gb_partial_words(uint32_t groups, uint32_t max_tables, int cus) {
  const GbGeometry w = gb_launch_geometry(groups, max_tables, cus, false);
  size_t words = static_cast<size_t>(w.ranges) * w.chunk_slots * w.part_words;
  if (gb_big(groups)) {
    const GbGeometry p = gb_launch_geometry(groups, max_tables, cus, true);
    const size_t pw = static_cast<size_t>(p.ranges) * p.chunk_slots * p.part_words;
    words = words > pw ? words : pw;
  }
  return words;
}
// 1: a workgroup of this device can have the packed table and its carry counters (160 KiB of LDS), the kernel decides
// by its sample; 0: never packed (smaller LDS, or DBHIP_GB_PACKED=0: A/B timing and an escape hatch); 2: always packed
// (DBHIP_GB_PACKED=force: tests)
unsigned gb_device_can_pack() {
  static const unsigned mode = [] {
    const char *e = getenv("DBHIP_GB_PACKED");
    if (e && e[0] == '0') return 0u;
    int dev = 0, lds = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0u;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return 0u;
    if (static_cast<size_t>(lds) < static_cast<size_t>(kGbMaxLdsGroups + kGbCarryWords) * sizeof(unsigned)) return 0u;
    return e && e[0] == 'f' ? 2u : 1u;
  }();
  return mode;
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
  const size_t spill_bytes = gb_spill_bytes(groups);
  const size_t partial_words = gb_partial_words(groups, max_tables, dev.cus);
  if (!ws_ok(workspace, workspace_bytes, kWsHeader + spill_bytes + partial_words * sizeof(unsigned))) return DBHIP_EWORKSPACE;
  hipStream_t s = as_stream(stream);
  hipError_t e = fill_async(workspace, 0, kWsHeader + spill_bytes, s);
  if (e != hipSuccess) return static_cast<int>(e);
  GbHeader *hdr = static_cast<GbHeader *>(workspace);
  unsigned *spill = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + kWsHeader);
  unsigned *partials = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + kWsHeader + spill_bytes);
  const u32x4 *k4 = reinterpret_cast<const u32x4 *>(keys), *v4 = reinterpret_cast<const u32x4 *>(vals);
  if (!gb_big(groups)) {
    const GbGeometry geo = gb_launch_geometry(groups, max_tables, dev.cus, false);
    const size_t lds = static_cast<size_t>(geo.lds_words) * sizeof(unsigned);
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(gb_aggregate_kernel<kGbBigThreads>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (kGbMaxLdsGroups + 32) * 4);
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL((gb_aggregate_kernel<kGbBigThreads>), dim3(geo.ranges * geo.chunk_slots), dim3(kGbBigThreads), lds, s, k4,
                       v4, keys, vals, n, groups, geo, partials, hdr);
    return launch_status();
  }
  const GbGeometry gp = gb_launch_geometry(groups, max_tables, dev.cus, true);
  const GbGeometry gw = gb_launch_geometry(groups, max_tables, dev.cus, false);
  const unsigned may_pack = gb_device_can_pack();
  const size_t lds = (may_pack ? static_cast<size_t>(kGbMaxLdsGroups + kGbCarryWords) : static_cast<size_t>(kGbMaxLdsGroups + 32)) * sizeof(unsigned);
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(gb_aggregate_big_kernel<kGbBigThreads>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  if (e != hipSuccess) return static_cast<int>(e);
  const unsigned grid_p = gp.ranges * gp.chunk_slots, grid_w = gw.ranges * gw.chunk_slots;
  hipLaunchKernelGGL((gb_aggregate_big_kernel<kGbBigThreads>), dim3(grid_p > grid_w ? grid_p : grid_w), dim3(kGbBigThreads), lds, s,
                     k4, v4, keys, vals, n, groups, gp, gw, may_pack, partials, spill, hdr);
  return launch_status();
}

int gb_merge(uint32_t groups, uint32_t max_tables, uint32_t *out, const void *workspace, dbhip_stream_t stream) {
  if (groups == 0) return DBHIP_OK;
  if (!out || !workspace) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  const size_t spill_bytes = gb_spill_bytes(groups);
  const unsigned *spill = reinterpret_cast<const unsigned *>(static_cast<const char *>(workspace) + kWsHeader);
  const unsigned *partials = reinterpret_cast<const unsigned *>(static_cast<const char *>(workspace) + kWsHeader + spill_bytes);
  if (gb_big(groups)) {
    const GbGeometry gp = gb_launch_geometry(groups, max_tables, dev.cus, true);
    const GbGeometry gw = gb_launch_geometry(groups, max_tables, dev.cus, false);
    // one workgroup per 128 groups in either mode (packed: 64 table words; a range may end inside a workgroup)
    const size_t words = static_cast<size_t>(gp.ranges) * (gp.range_groups / 2);
    const size_t grid_p = (words + kWave - 1) / kWave, grid_w = (static_cast<size_t>(groups) + 2 * kWave - 1) / (2 * kWave);
    hipLaunchKernelGGL(gb_reduce_big_kernel, dim3(static_cast<unsigned>(grid_p > grid_w ? grid_p : grid_w)),
                       dim3(kGbRedWaves * kWave), 0, as_stream(stream), partials, spill,
                       static_cast<const GbHeader *>(workspace), gp, gw, groups, out);
  } else {
    const GbGeometry geo = gb_launch_geometry(groups, max_tables, dev.cus, false);
    hipLaunchKernelGGL(gb_reduce_kernel, dim3((groups + kWave - 1) / kWave), dim3(256), 0, as_stream(stream), partials,
                       geo, groups, out);
  }
  return launch_status();
}
}  // namespace

extern "C" size_t dbhip_groupby_sum_u32_workspace_bytes(size_t n, uint32_t groups) {
  (void)n;
  // sized for the largest device this library targets (256 CUs) so the query needs no device
  const uint32_t g = groups ? groups : 1;
  return align_up(kWsHeader + gb_spill_bytes(g) + gb_partial_words(g, 0, 256) * sizeof(unsigned), kWsAlign);
}

extern "C" int dbhip_groupby_sum_u32(const uint32_t *keys, const uint32_t *vals, size_t n,
                                     uint32_t groups, uint32_t *out, void *workspace,
                                     size_t workspace_bytes, dbhip_stream_t stream) {
  if (groups && !out) return DBHIP_EINVAL;
  // Short columns get as many private tables as they have 8192-row steps for (a workgroup's step: 1024 lanes x two
  // 16-byte loads per column), not one per CU: every table is cleared, written out and read back whatever it received —
  // 2^13 rows into 32768 groups took 31 us, 256 tables of 128 KiB for one step's worth of rows.
  const size_t steps = (n + 8191) / 8192;
  const uint32_t tables = steps >= 256 ? 0u : static_cast<uint32_t>(steps ? steps : 1);
  const int rc = gb_partial(keys, vals, n, groups, tables, workspace, workspace_bytes, stream);
  return rc != 0 ? rc : gb_merge(groups, tables, out, workspace, stream);
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
