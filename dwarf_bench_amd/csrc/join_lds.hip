// join_lds.hip — dwarf 4a at every size (and dwarf 4b from 2^16 build rows): radix-partitioned build with
// LDS-resident sub-tables, single-gather probe.  OmniSci semantics: distinct-key table, per-key count, exclusive scan -> position, ids grouped by key, probe ->
// {offset, count} (common/dpcpp/omnisci_hashtable.hpp:58-261).
//
// Why: on MI355X a table in HBM costs one memory-side atomic per step (20-27 G/s random, tools/ubench) — three per
// build row — and three 4-byte gathers per probe row (53 G/s).  Random LDS atomics run at ~4,000 G/s chip-wide
// (profiles/r04_ubench.txt).  So:
//   build  1. partition the build column into K = ceil(n / 2048) partitions (join_common.hpp jl_layout) by the mixed hash,
//             in one or two levels of <= 1024-way scatter (jl_hist / jl_offsets / jl_scatter: LDS counts, one global
//             reservation per bucket per tile, runs of (key, row id) pairs written contiguously; tiles of 4096 rows,
//             of 16384 where a level has 512+ buckets: JlShape; both histograms from ONE read of the keys up to 81920
//             partitions, above that level 0 leaves every row's level-1 bucket as a 16-bit column for the level-1
//             histogram to read instead of the pairs);
//          2. persistent workgroups — as many as are RESIDENT, partitions dealt by ticket — walk the partitions
//             (jl_build_kernel): 3072-slot sub-table in LDS (kJlSubSlots) — plain read of the key's slot, ds_cmpst on
//             an empty one, ONE returning ds_add whose old value is the row's rank inside its key group, LDS exclusive
//             scan -> positions, the fill takes slot and rank of each row from registers kept since the claim, then
//             the sub-table is written out as 8-byte slots {key, first id position | count field} — EVERY slot, empty
//             ones included, carries the position the exclusive scan reached there, and positions run on from one
//             sub-table to the next (the global table is the concatenation of the sub-tables plus one sentinel slot
//             {empty, n}), so a key's match count is ALSO first_position[slot + 1] - first_position[slot]; the bits
//             of the position word that n does not need hold min(count - 1, all ones) (join_common.hpp): the table
//             costs 12 bytes per build row;
//   probe  ONE 8-byte gather per probe row whenever the count fits its field (a 16-byte read of slot and neighbour
//          crossed a 64-byte line for every eighth row: one more memory request); the all-ones field sends the row to
//          the right-hand neighbour.  Partition from the high hash bits, linear probing inside the sub-table, outputs
//          written coalesced in row order.
// A partition may hold any number of rows: one far above its expected size — a hot key — is shared by all workgroups
// ("giant partitions" below).  It may hold any number of DISTINCT keys too: 2048 +- 45 per sigma are expected, a
// sub-table has 3072 slots, and a partition that needs more (keys constructed against the hash) is built in a table of
// its own in HBM by one workgroup (jl_spill_partition; until round 4 this raised DBHIP_DEV_TABLE_FULL) — the join
// answers every input the reference's table answers (omnisci_hashtable.hpp:80-108, ht_size = 2 * distinct).
#include "dbhip_common.hpp"
#include "join_common.hpp"

namespace dbhip {
namespace {

constexpr unsigned kEmptyKey = 0xFFFFFFFFu;
#ifndef DBHIP_JL_THREADS
#define DBHIP_JL_THREADS 512
#endif
constexpr int kJlThreads = DBHIP_JL_THREADS;  // scatter / histogram / probe workgroups
#ifndef DBHIP_JL_KPT
#define DBHIP_JL_KPT 8
#endif
constexpr int kJlKpt = DBHIP_JL_KPT;
constexpr int kJlTile = kJlThreads * kJlKpt;  // 4096 rows per scatter tile, 36 KiB of LDS: four 512-thread workgroups
                                              // per CU.  Measured at 2^26 rows (build, us): 512x8 1361, 512x16 1423,
                                              // 1024x8 1390, 512x4 1442, 256x8 1499, 512x32 1687
#ifndef DBHIP_JL_BUILD_THREADS
#define DBHIP_JL_BUILD_THREADS 512
#endif
constexpr int kJlBuildThreads = DBHIP_JL_BUILD_THREADS;  // per-partition build workgroup; 2^26 rows: 512 -> 1398 us,
                                                        // 256 -> 1514 us, 1024 -> 1465 us (whole build); round 4, resident
                                                        // ticketed grid: build 920 / 938 / 1148 us and the radix join's fused
                                                        // kernel 605 / 700 / 1285 us with 512 / 256 / 1024 threads

__device__ __forceinline__ unsigned jl_pid(unsigned key, unsigned parts) {
  return static_cast<unsigned>((static_cast<unsigned long long>(fmix32(key)) * parts) >> 32);
}
// Destination RANK of the multi-GPU partitioner: a second, independent hash.  It must not be the high bits of
// fmix32(key) again: a rank only receives keys of one rank bucket, and its local build (jl_pid above) would then
// find all of them in 1/P of its partitions — P times overfull sub-tables (at P = 8 more keys than slots: the spill path).
__device__ __forceinline__ unsigned jl_rank_of(unsigned key, unsigned parts) {
  return static_cast<unsigned>((static_cast<unsigned long long>(fmix32(key * 0x9E3779B1u + 0x7F4A7C15u)) * parts) >> 32);
}
template <bool RANK>
__device__ __forceinline__ unsigned jl_pid_sel(unsigned key, unsigned parts) {
  return RANK ? jl_rank_of(key, parts) : jl_pid(key, parts);
}

// ---- level 0: histogram per (tile group, bucket) --------------------------------------------------
// The column's 4096-row tiles are cut into kJlGroups contiguous groups; every group owns a private slice
// of every bucket (its rows' share), so the scatter's reservations on one cursor come from 1/64 of
// the tiles: 16384 tiles bumping the SAME 128 cursors serialise on the memory-side atomic unit
// (measured: 544 us for a 768 MiB scatter).
constexpr unsigned kJlGroups = 64;
constexpr unsigned kJlHistWgPerGroup = 32;

// rows of a tile group: a whole number of the level-0 scatter's tiles (`tile` rows each — the scatter comes in three
// tile shapes, see JlShape), so histogram and scatter agree on which rows are group g's
__host__ __device__ __forceinline__ size_t jl_group_rows(size_t n, unsigned tile) {
  const size_t tiles = (n + tile - 1) / tile;
  return (tiles + kJlGroups - 1) / kJlGroups * tile;
}

template <bool RANK>
__global__ __launch_bounds__(kJlThreads) void jl_hist0_kernel(const unsigned *__restrict__ keys, size_t n, size_t group_rows,
                                                              unsigned parts, unsigned k2_shift,
                                                              unsigned k1, unsigned long long *counts_g) {
  extern __shared__ unsigned s_hist[];
  const unsigned group = blockIdx.x / kJlHistWgPerGroup, w = blockIdx.x % kJlHistWgPerGroup;
  const size_t lo = static_cast<size_t>(group) * group_rows;
  size_t hi = lo + group_rows;
  hi = hi < n ? hi : n;
  if (lo >= hi) return;
  for (unsigned i = threadIdx.x; i < k1; i += kJlThreads) s_hist[i] = 0;
  __syncthreads();
  for (size_t i = lo + static_cast<size_t>(w) * 4 * kJlThreads + threadIdx.x; i < hi;
       i += static_cast<size_t>(kJlHistWgPerGroup) * 4 * kJlThreads) {  // four independent loads per lane per step
    unsigned k[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) k[j] = i + j * kJlThreads < hi ? keys[i + j * kJlThreads] : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i + j * kJlThreads < hi) atomicAdd(&s_hist[jl_pid_sel<RANK>(k[j], parts) >> k2_shift], 1u);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < k1; i += kJlThreads)
    if (s_hist[i]) atomicAdd(&counts_g[static_cast<size_t>(group) * k1 + i], static_cast<unsigned long long>(s_hist[i]));
}

// ---- both levels' histograms in ONE read of the keys (parts <= 32768: the counters fit 128 KiB of LDS) --------------
// Workgroup (group g, w) counts the FINAL partition of every row of its share of group g in an LDS histogram of `parts`
// bins and stores it, plainly, as its own row of wgcnt[][]; jl_hist_reduce sums the rows: per (group, level-0 bucket)
// for the level-0 cursors and per partition for level 1.  Replaces jl_hist0 + jl_hist1: the second used to re-read the
// level-0 output (8 bytes per row: 113 us of the 2^26-row build).
constexpr unsigned kJlFusedWgPerGroup = 4;   // 64 groups x 4 = 256 workgroups of 1024 threads: one per CU
constexpr unsigned kJlFusedThreads = 1024;
#ifndef DBHIP_JL_FUSED_MAX_PARTS
#define DBHIP_JL_FUSED_MAX_PARTS 32768
#endif
constexpr unsigned kJlFusedMaxParts = DBHIP_JL_FUSED_MAX_PARTS;  // 0 disables the fused histogram (A/B timing)
// two 16-bit counters per LDS word (jl_hist_fused16_kernel): as many partitions as the CU's 160 KiB hold — 2^27 rows
// and a quarter more (a rank of the 8-GPU join receives 2^27 rows +- a few thousand: 65537+ partitions)
constexpr unsigned kJlFused16MaxParts = 80 * 1024;
#ifndef DBHIP_JL_HIST_LOADS
#define DBHIP_JL_HIST_LOADS 4
#endif
constexpr int kJlHistLoads = DBHIP_JL_HIST_LOADS;  // 16-byte key loads in flight per lane of the fused16 histogram

__global__ __launch_bounds__(kJlFusedThreads) void jl_hist_fused_kernel(const unsigned *__restrict__ keys, size_t n, size_t group_rows,
                                                                        unsigned parts, unsigned *__restrict__ wgcnt) {
  extern __shared__ unsigned s_hist[];
  const unsigned group = blockIdx.x / kJlFusedWgPerGroup, w = blockIdx.x % kJlFusedWgPerGroup;
  for (unsigned i = threadIdx.x; i < parts; i += kJlFusedThreads) s_hist[i] = 0;
  __syncthreads();
  const size_t lo = static_cast<size_t>(group) * group_rows;
  size_t hi = lo + group_rows;
  hi = hi < n ? hi : n;
  if (lo < hi && (reinterpret_cast<uintptr_t>(keys + lo) & 15u) == 0) {
    // 16-byte loads, four in flight per lane (4-byte loads kept 16 KiB per CU in flight: 79 us for 256 MiB of keys)
    const u32x4 *k4 = reinterpret_cast<const u32x4 *>(keys + lo);
    const size_t n4 = (hi - lo) / 4;
    for (size_t i = static_cast<size_t>(w) * 4 * kJlFusedThreads + threadIdx.x; i < n4;
         i += static_cast<size_t>(kJlFusedWgPerGroup) * 4 * kJlFusedThreads) {
      u32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = i + j * kJlFusedThreads < n4 ? k4[i + j * kJlFusedThreads] : u32x4{0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j * kJlFusedThreads < n4) {
          atomicAdd(&s_hist[jl_pid(v[j].x, parts)], 1u);
          atomicAdd(&s_hist[jl_pid(v[j].y, parts)], 1u);
          atomicAdd(&s_hist[jl_pid(v[j].z, parts)], 1u);
          atomicAdd(&s_hist[jl_pid(v[j].w, parts)], 1u);
        }
    }
    if (w == 0 && lo + n4 * 4 + threadIdx.x < hi) atomicAdd(&s_hist[jl_pid(keys[lo + n4 * 4 + threadIdx.x], parts)], 1u);
  } else {
    for (size_t i = lo + static_cast<size_t>(w) * 4 * kJlFusedThreads + threadIdx.x; i < hi;
         i += static_cast<size_t>(kJlFusedWgPerGroup) * 4 * kJlFusedThreads) {  // four independent loads per lane per step
      unsigned k[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) k[j] = i + j * kJlFusedThreads < hi ? keys[i + j * kJlFusedThreads] : 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j * kJlFusedThreads < hi) atomicAdd(&s_hist[jl_pid(k[j], parts)], 1u);
    }
  }
  __syncthreads();
  unsigned *mine = wgcnt + static_cast<size_t>(blockIdx.x) * parts;
  for (unsigned i = threadIdx.x; i < parts; i += kJlFusedThreads) mine[i] = s_hist[i];
}

// ---- the same for 32768 < parts <= 65536 (2^27-row shards: what every rank of the 8-GPU join partitions) ----------------
// 65536 32-bit counters do not fit the LDS; 16 bits are enough for a workgroup's share of a partition (2^27 rows / 256
// workgroups / 65536 partitions = 8 rows) unless the input is heavily skewed.  TWO counters per LDS word, fed by
// RETURNING ds_add: from the returned value a lane sees exactly when its increment carried out of the low half (the
// even partition's counter wrapped and the odd one's now holds one too many) or out of bit 31 (the odd one wrapped) and
// settles that in the global accumulators the reduce kernel adds to — correct for any input, and free for every input
// that is not pathological.  The workgroup's row of wgcnt is the LDS image: parts / 2 words.
__global__ __launch_bounds__(kJlFusedThreads) void jl_hist_fused16_kernel(const unsigned *__restrict__ keys, size_t n, size_t group_rows,
                                                                          unsigned parts, unsigned log2_k2, unsigned k1,
                                                                          unsigned *__restrict__ wgcnt,
                                                                          unsigned long long *counts0g,
                                                                          unsigned long long *counts1) {
  extern __shared__ unsigned s_hist[];
  const unsigned group = blockIdx.x / kJlFusedWgPerGroup, w = blockIdx.x % kJlFusedWgPerGroup;
  const unsigned words = parts / 2;
  for (unsigned i = threadIdx.x; i < words; i += kJlFusedThreads) s_hist[i] = 0;
  __syncthreads();
  auto count = [&](unsigned key) {
    const unsigned p = jl_pid(key, parts);
    const unsigned inc = 1u << ((p & 1u) << 4);
    const unsigned old = atomicAdd(&s_hist[p >> 1], inc);
    const bool carry16 = (p & 1u) == 0 && (old & 0xFFFFu) == 0xFFFFu, carry32 = old + inc < old;
    if (carry16 || carry32) {  // (more than 65535 rows of this workgroup's share in one partition)
      const unsigned even = p & ~1u, odd = p | 1u;
      unsigned long long *g0 = counts0g + static_cast<size_t>(group) * k1;
      if (carry16) {
        atomicAdd(&counts1[even], 65536ull);
        atomicAdd(&g0[even >> log2_k2], 65536ull);
        atomicAdd(&counts1[odd], ~0ull);  // minus one: the carry landed in the odd partition's half
        atomicAdd(&g0[odd >> log2_k2], ~0ull);
      }
      if (carry32) {
        atomicAdd(&counts1[odd], 65536ull);
        atomicAdd(&g0[odd >> log2_k2], 65536ull);
      }
    }
  };
  const size_t lo = static_cast<size_t>(group) * group_rows;
  size_t hi = lo + group_rows;
  hi = hi < n ? hi : n;
  if (lo < hi && (reinterpret_cast<uintptr_t>(keys + lo) & 15u) == 0) {
    const u32x4 *k4 = reinterpret_cast<const u32x4 *>(keys + lo);
    const size_t n4 = (hi - lo) / 4;
    // (round 4, measured and dropped: the next step's loads in flight while this step's keys are counted, two register
    //  sets as in the group-by — 80.7 -> 99.6 us for the 256 MiB of keys of a 2^26-row side; eight loads per lane in
    //  flight instead of four: partition of one side 572 -> 590 us, two: the same as four)
    for (size_t i = static_cast<size_t>(w) * kJlHistLoads * kJlFusedThreads + threadIdx.x; i < n4;
         i += static_cast<size_t>(kJlFusedWgPerGroup) * kJlHistLoads * kJlFusedThreads) {
      u32x4 v[kJlHistLoads];
#pragma unroll
      for (int j = 0; j < kJlHistLoads; ++j) v[j] = i + j * kJlFusedThreads < n4 ? k4[i + j * kJlFusedThreads] : u32x4{0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < kJlHistLoads; ++j)
        if (i + j * kJlFusedThreads < n4) {
          count(v[j].x);
          count(v[j].y);
          count(v[j].z);
          count(v[j].w);
        }
    }
    if (w == 0 && lo + n4 * 4 + threadIdx.x < hi) count(keys[lo + n4 * 4 + threadIdx.x]);
  } else {
    for (size_t i = lo + static_cast<size_t>(w) * kJlFusedThreads + threadIdx.x; i < hi;
         i += static_cast<size_t>(kJlFusedWgPerGroup) * kJlFusedThreads)
      count(keys[i]);
  }
  __syncthreads();
  unsigned *mine = wgcnt + static_cast<size_t>(blockIdx.x) * words;
  for (unsigned i = threadIdx.x; i < words; i += kJlFusedThreads) mine[i] = s_hist[i];
}

// reduce of the packed rows; ADDS to counts1 / counts0g (zeroed with the metadata, and possibly holding the carries
// the histogram kernel settled): one thread per WORD (two partitions) for the column sums, one wave per (row, bucket)
// for the level-0 counts — the sum of both halves of a bucket's words
__global__ __launch_bounds__(256) void jl_hist_reduce16_kernel(const unsigned *__restrict__ wgcnt, unsigned parts, unsigned k1,
                                                               unsigned k2, unsigned long long *counts0g,
                                                               unsigned long long *counts1) {
  constexpr unsigned kRows = kJlGroups * kJlFusedWgPerGroup;
  const unsigned words = parts / 2, col_blocks = (words + 255) / 256;
  if (blockIdx.x < col_blocks) {
    const unsigned wd = blockIdx.x * 256 + threadIdx.x;
    if (wd >= words) return;
    unsigned long long lo = 0, hi = 0;
    for (unsigned r0 = 0; r0 < kRows; r0 += 8) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = wgcnt[static_cast<size_t>(r0 + u) * words + wd];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        lo += v[u] & 0xFFFFu;
        hi += v[u] >> 16;
      }
    }
    counts1[2 * wd] += lo;  // (the histogram kernel has finished: no one else touches these words now)
    counts1[2 * wd + 1] += hi;
    return;
  }
  const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const size_t item = static_cast<size_t>(blockIdx.x - col_blocks) * 4 + wave;  // (row, bucket)
  if (item >= static_cast<size_t>(kRows) * k1) return;
  const unsigned row = static_cast<unsigned>(item / k1), bucket = static_cast<unsigned>(item % k1);
  const unsigned *src = wgcnt + static_cast<size_t>(row) * words + static_cast<size_t>(bucket) * (k2 / 2);
  unsigned mine = 0;
  for (unsigned sub = lane; sub < k2 / 2; sub += kWave) mine += (src[sub] & 0xFFFFu) + (src[sub] >> 16);
  mine = wave_reduce_add(mine);
  if (lane == kWave - 1 && mine)
    atomicAdd(&counts0g[static_cast<size_t>(row / kJlFusedWgPerGroup) * k1 + bucket], static_cast<unsigned long long>(mine));
}

// counts1[p] = rows of partition p (column sums of wgcnt, one thread per partition: the first parts/256 workgroups),
// counts0g[g][b] = rows of group g in level-0 bucket b (one WAVE per (workgroup row, bucket): k2 contiguous counters,
// added to the zeroed counts0g with one atomic per wave: the remaining workgroups)
__global__ __launch_bounds__(256) void jl_hist_reduce_kernel(const unsigned *__restrict__ wgcnt, unsigned parts, unsigned k1,
                                                             unsigned k2, unsigned long long *counts0g,
                                                             unsigned long long *counts1) {
  constexpr unsigned kRows = kJlGroups * kJlFusedWgPerGroup;
  const unsigned col_blocks = (parts + 255) / 256;
  if (blockIdx.x < col_blocks) {
    const unsigned p = blockIdx.x * 256 + threadIdx.x;
    if (p >= parts) return;
    unsigned long long sum = 0;
    for (unsigned r0 = 0; r0 < kRows; r0 += 8) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = wgcnt[static_cast<size_t>(r0 + u) * parts + p];
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    counts1[p] = sum;
    return;
  }
  const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const size_t item = static_cast<size_t>(blockIdx.x - col_blocks) * 4 + wave;  // (row, bucket)
  if (item >= static_cast<size_t>(kRows) * k1) return;
  const unsigned row = static_cast<unsigned>(item / k1), bucket = static_cast<unsigned>(item % k1);
  const unsigned *src = wgcnt + static_cast<size_t>(row) * parts + static_cast<size_t>(bucket) * k2;
  unsigned mine = 0;
  for (unsigned sub = lane; sub < k2; sub += kWave) mine += src[sub];
  mine = wave_reduce_add(mine);
  if (lane == kWave - 1 && mine)
    atomicAdd(&counts0g[static_cast<size_t>(row / kJlFusedWgPerGroup) * k1 + bucket], static_cast<unsigned long long>(mine));
}

// bucket starts, per-group cursors and the tile index of every bucket (for the 1-D grid of level 1).
// One workgroup, thread b owns bucket b (k1 <= 1024).
__global__ __launch_bounds__(1024) void jl_offsets0_kernel(const unsigned long long *__restrict__ counts_g,
                                                           unsigned k1, unsigned tile1, unsigned long long *cursors_g,
                                                           unsigned long long *starts, unsigned long long *tile_starts,
                                                           unsigned long long *totals_out) {
  __shared__ unsigned long long s_tot[1024], s_start[1025], s_tstart[1025];
  __shared__ unsigned long long s_wrow[16], s_wtile[16];
  const unsigned b = threadIdx.x, lane = b & (kWave - 1), wave = b / kWave;
  unsigned long long tot = 0;
  if (b < k1)
    for (unsigned g = 0; g < kJlGroups; ++g) tot += counts_g[static_cast<size_t>(g) * k1 + b];
  s_tot[b] = tot;
  // exclusive prefix over the buckets of rows and of level-1 tiles: wave scans + a 16-entry pass
  const unsigned long long tl = b < k1 ? (tot + tile1 - 1) / tile1 : 0ull;  // tiles of the level-1 scatter (tile1 rows each)
  unsigned long long ir = tot, it = tl;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const unsigned long long pr = __shfl_up(ir, off, kWave), pt = __shfl_up(it, off, kWave);
    if (lane >= static_cast<unsigned>(off)) {
      ir += pr;
      it += pt;
    }
  }
  if (lane == kWave - 1) {
    s_wrow[wave] = ir;
    s_wtile[wave] = it;
  }
  __syncthreads();
  unsigned long long base_r = 0, base_t = 0;
  for (unsigned w = 0; w < wave; ++w) {
    base_r += s_wrow[w];
    base_t += s_wtile[w];
  }
  s_start[b] = base_r + ir - tot;
  s_tstart[b] = base_t + it - tl;
  if (b == 1023) {
    s_start[1024] = base_r + ir;
    s_tstart[1024] = base_t + it;
  }
  __syncthreads();
  if (b < k1) {
    starts[b] = s_start[b];
    tile_starts[b] = s_tstart[b];
    if (totals_out) totals_out[b] = s_tot[b];
    unsigned long long run = s_start[b];
    for (unsigned g = 0; g < kJlGroups; ++g) {
      cursors_g[static_cast<size_t>(g) * k1 + b] = run;
      run += counts_g[static_cast<size_t>(g) * k1 + b];
    }
  }
  if (b == 0) {
    starts[k1] = s_start[k1];
    tile_starts[k1] = s_tstart[k1];
  }
}

// level 1: bucket b's k2 sub-buckets live inside [starts0[b], starts0[b+1]).  One workgroup per bucket.
__global__ __launch_bounds__(kJlThreads) void jl_offsets1_kernel(const unsigned long long *__restrict__ counts1,
                                                                 const unsigned long long *__restrict__ starts0,
                                                                 unsigned k1, unsigned k2, unsigned long long *starts1,
                                                                 unsigned long long *cursors1) {
  __shared__ unsigned s_wsum[kJlThreads / kWave];
  const unsigned b = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const unsigned per = (k2 + kJlThreads - 1) / kJlThreads;  // <= 4
  unsigned c[4] = {0, 0, 0, 0}, mine = 0;
#pragma unroll
  for (unsigned u = 0; u < 4; ++u) {
    const unsigned sidx = tid * per + u;
    if (u < per && sidx < k2) c[u] = static_cast<unsigned>(counts1[static_cast<size_t>(b) * k2 + sidx]);
    mine += c[u];
  }
  const unsigned incl = wave_inclusive_scan(mine);
  if (lane == kWave - 1) s_wsum[wave] = incl;
  __syncthreads();
  unsigned long long run = starts0[b] + incl - mine;
  for (unsigned w = 0; w < wave; ++w) run += s_wsum[w];
#pragma unroll
  for (unsigned u = 0; u < 4; ++u) {
    const unsigned sidx = tid * per + u;
    if (u < per && sidx < k2) {
      starts1[static_cast<size_t>(b) * k2 + sidx] = run;
      cursors1[static_cast<size_t>(b) * k2 + sidx] = run;
      run += c[u];
    }
  }
  if (b == k1 - 1 && tid == 0) starts1[static_cast<size_t>(k1) * k2] = starts0[k1];
}

// Scatter of one 4096-row tile into `nb` (<= 1024) buckets, staged through LDS so that the global
// writes are runs: rows are ranked inside their bucket with LDS atomics, the tile is re-ordered by
// bucket in LDS, every bucket's run gets ONE global reservation, and consecutive lanes then write
// consecutive addresses of a run.  LEVEL selects how the bucket is recomputed from the key on the way
// out (0: pid >> arg, 1: pid & arg).  dest[j] == nb marks an invalid (out-of-range) row.
// LDS: cnt[nb] | excl[nb] | base[nb] (u64) | keys[4096] | rids[4096] | 4 wave sums.
constexpr size_t jl_scatter_lds_bytes(unsigned nb, unsigned tile = kJlTile, unsigned threads = kJlThreads) {
  return static_cast<size_t>(nb) * 16 + 2 * static_cast<size_t>(tile) * sizeof(unsigned) + sizeof(unsigned) * (threads / kWave);
}
struct JlNoHook {
  __device__ __forceinline__ void operator()() const {}
};
// before_stores(): called once, right before the tile's global stores are issued (the level-0 kernel waits there for
// the next tile's prefetched keys: see jl_scatter0_kernel)
template <int LEVEL, int THREADS, int KPT, bool RANK = false, bool DIGITS = false, class Hook = JlNoHook>
__device__ __forceinline__ void jl_scatter_tile(const unsigned (&key)[KPT], const unsigned (&rid)[KPT],
                                                const unsigned (&dest)[KPT], unsigned nb, unsigned parts,
                                                unsigned arg, unsigned long long *cursors,
                                                unsigned *__restrict__ out_keys, unsigned *__restrict__ out_rids,
                                                unsigned *s_mem, Hook before_stores = Hook()) {
  unsigned long long *s_base = reinterpret_cast<unsigned long long *>(s_mem);  // 8-byte aligned first
  unsigned *s_cnt = s_mem + 2 * nb;
  unsigned *s_excl = s_cnt + nb;
  unsigned *s_keys = s_excl + nb;
  unsigned *s_rids = s_keys + (THREADS * KPT);
  unsigned *s_wsum = s_rids + (THREADS * KPT);
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;

  for (unsigned i = tid; i < nb; i += THREADS) s_cnt[i] = 0;
  __syncthreads();
  unsigned rank[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) rank[j] = dest[j] < nb ? atomicAdd(&s_cnt[dest[j]], 1u) : 0u;
  __syncthreads();
  // exclusive scan of the bucket counts (nb <= 1024: up to 4 consecutive buckets per thread)
  const unsigned per = (nb + THREADS - 1) / THREADS;
  unsigned c[4] = {0, 0, 0, 0}, mine = 0;
#pragma unroll
  for (unsigned u = 0; u < 4; ++u) {
    const unsigned b = tid * per + u;
    if (u < per && b < nb) c[u] = s_cnt[b];
    mine += c[u];
  }
  const unsigned incl = wave_inclusive_scan(mine);
  if (lane == kWave - 1) s_wsum[wave] = incl;
  __syncthreads();
  unsigned run = incl - mine;
  for (unsigned w = 0; w < wave; ++w) run += s_wsum[w];
  unsigned total = 0;
#pragma unroll
  for (int w = 0; w < (THREADS / kWave); ++w) total += s_wsum[w];
#pragma unroll
  for (unsigned u = 0; u < 4; ++u) {
    const unsigned b = tid * per + u;
    if (u < per && b < nb) {
      s_excl[b] = run;
      // (one returning global atomic per bucket per tile; replacing them by a precomputed offset in a timing
      //  experiment did not make the kernel faster: the reservations are not what bounds it)
      s_base[b] = c[u] ? atomicAdd(&cursors[b], static_cast<unsigned long long>(c[u])) : 0ull;
      run += c[u];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    if (dest[j] < nb) {
      const unsigned p = s_excl[dest[j]] + rank[j];
      s_keys[p] = key[j];
      s_rids[p] = rid[j];
    }
  }
  __syncthreads();
  before_stores();
  for (unsigned p = tid; p < total; p += THREADS) {
    const unsigned k = s_keys[p];
    const unsigned pid = jl_pid_sel<RANK>(k, parts);
    const unsigned d = LEVEL == 0 ? pid >> arg : pid & arg;
    const size_t slot = s_base[d] + (p - s_excl[d]);
    if (DIGITS) {  // pairs + the row's level-1 bucket as a 16-bit column of its own (behind `out_rids`): what the level-1
                   // histogram reads instead of the pairs, 2 bytes per row for 8 (jl_hist1d_kernel)
      reinterpret_cast<u32x2 *>(out_keys)[slot] = u32x2{k, s_rids[p]};
      reinterpret_cast<unsigned short *>(out_rids)[slot] = static_cast<unsigned short>(pid & ((1u << arg) - 1u));
    } else if (out_rids) {  // two columns (the rank-level partition: its outputs go into an all-to-all as they are)
      out_keys[slot] = k;
      out_rids[slot] = s_rids[p];
    } else {  // one array of (key, row id) pairs: one 8-byte store per row, a run of r rows is 8r contiguous bytes
      // (plain stores: the runs of neighbouring tiles meet in L2; non-temporal stores here made a partition side of
      //  2^26 rows 778 us instead of 602)
      reinterpret_cast<u32x2 *>(out_keys)[slot] = u32x2{k, s_rids[p]};
    }
  }
  __syncthreads();  // LDS is reused by the next tile
}

// level-0 scatter: (key, row id) pairs bucket-major; row id = index (or row_ids[index] when given)
// THREADS x KPT rows per tile (JlShape); RIDS: row ids come as a column (the received pairs of the multi-GPU join) —
// a template parameter so that the other callers do not carry the prefetched row-id registers
#ifndef DBHIP_JL_SC0_WPE
#define DBHIP_JL_SC0_WPE 6
#endif
template <bool RANK, bool RIDS, int THREADS, int KPT, bool DIGITS = false>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS == 512 && !RIDS ? DBHIP_JL_SC0_WPE : 1))) void jl_scatter0_kernel(const unsigned *__restrict__ keys,
                                                                 const unsigned *__restrict__ row_ids,
                                                                 unsigned long long first_row, size_t n,
                                                                 unsigned parts, unsigned k2_shift, unsigned k1,
                                                                 unsigned long long *cursors,
                                                                 unsigned *__restrict__ out_keys,
                                                                 unsigned *__restrict__ out_rids) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_mem[];
  const size_t tiles = (n + (THREADS * KPT) - 1) / (THREADS * KPT);
  // XCD-aware tile order (speed only, any order is correct): workgroups are dealt to the 8 XCDs round-robin by
  // blockIdx, so XCD x = blockIdx % 8 takes the tile groups g with g % 8 == x.  A (group, bucket) write frontier is
  // then advanced by ONE XCD, whose L2 merges the partial lines of consecutive runs before they leave for memory
  // (WRITE_SIZE 770 MB for 537 MB stored when every XCD touched every frontier; 338 -> 310 us at 2^26 rows).
  // The same slicing of the level-1 scatter (buckets b % 8 == x per XCD, persistent grid) measured no faster.
  const size_t tpg = jl_group_rows(n, THREADS * KPT) / (THREADS * KPT);
  const unsigned xcd = blockIdx.x % 8u, slot = blockIdx.x / 8u, per_xcd = gridDim.x / 8u;  // host: grid % 8 == 0
  const size_t locals = (kJlGroups / 8) * tpg;
  // tile of the workgroup's `local`-th step, or `tiles` when that step has none (the ragged end of the last group)
  auto tile_of = [&](size_t local, size_t *group) -> size_t {
    *group = (local / tpg) * 8 + xcd;
    const size_t tile = *group * tpg + local % tpg;
    return local < locals && tile < tiles ? tile : tiles;
  };
  auto load_tile = [&](size_t tile, unsigned (&k)[KPT], unsigned (&r)[KPT]) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const size_t idx = tile * (THREADS * KPT) + static_cast<size_t>(j) * THREADS + threadIdx.x;
      const bool valid = tile < tiles && idx < n;
      k[j] = valid ? keys[idx] : 0u;
      r[j] = RIDS && valid ? row_ids[idx] : 0u;
    }
  };
  // The next tile's keys are requested before the current tile's LDS work and waited for right before the current
  // tile's stores go out (vmcnt counts a wave's loads and stores in issue order: waiting for loads at the top of the
  // next step would also wait for every store of this one).  They cross the loop in registers moved by a v_mov the
  // compiler cannot see through — a loop-carried register that a load defined is waited for with vmcnt(0) at first use.
  unsigned ckey[KPT], crid[KPT];
  size_t group = 0, tile = tile_of(slot, &group);
  load_tile(tile, ckey, crid);
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    asm volatile("v_mov_b32 %0, %0" : "+v"(ckey[j]));
    if (RIDS) asm volatile("v_mov_b32 %0, %0" : "+v"(crid[j]));
  }
  for (size_t local = slot; local < locals; local += per_xcd) {
    size_t ngroup = 0;
    const size_t ntile = tile_of(local + per_xcd, &ngroup);
    unsigned nkey[KPT], nrid[KPT], mkey[KPT], mrid[KPT];
    load_tile(ntile, nkey, nrid);
    auto wait_next = [&]() {
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        asm volatile("v_mov_b32 %0, %1" : "=v"(mkey[j]) : "v"(nkey[j]));
        if (RIDS) asm volatile("v_mov_b32 %0, %1" : "=v"(mrid[j]) : "v"(nrid[j]));
        else mrid[j] = 0u;
      }
    };
    if (tile < tiles) {  // uniform over the workgroup
      const size_t base = tile * (THREADS * KPT);
      unsigned rid[KPT], dest[KPT];
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const size_t idx = base + static_cast<size_t>(j) * THREADS + threadIdx.x;
        const bool valid = idx < n;
        rid[j] = valid ? (RIDS ? crid[j] : static_cast<unsigned>(first_row + idx)) : 0u;
        dest[j] = valid ? jl_pid_sel<RANK>(ckey[j], parts) >> k2_shift : k1;
      }
      // this tile bumps only its group's cursors
      jl_scatter_tile<0, THREADS, KPT, RANK, DIGITS>(ckey, rid, dest, k1, parts, k2_shift, cursors + group * k1, out_keys, out_rids, s_mem, wait_next);
    } else {
      wait_next();
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      ckey[j] = mkey[j];
      crid[j] = mrid[j];
    }
    tile = ntile;
    group = ngroup;
  }
}

// (bucket, tile-in-bucket) of virtual tile `vt` by binary search over tile_starts[0..k1]
__device__ __forceinline__ bool jl_locate(const unsigned long long *__restrict__ tile_starts, unsigned k1,
                                          unsigned long long vt, unsigned *bucket, unsigned long long *tile) {
  if (vt >= tile_starts[k1]) return false;
  unsigned lo = 0, hi = k1;  // find largest b with tile_starts[b] <= vt
  while (hi - lo > 1) {
    const unsigned mid = (lo + hi) / 2;
    if (tile_starts[mid] <= vt) lo = mid; else hi = mid;
  }
  *bucket = lo;
  *tile = vt - tile_starts[lo];
  return true;
}

// level-1 histogram: kJlHist1WgPerBucket workgroups stride over one level-0 bucket (four independent loads per
// lane per step), so a bucket's k2 counters see 16 flushes instead of one per scatter tile
constexpr unsigned kJlHist1WgPerBucket = 16;

__global__ __launch_bounds__(kJlThreads) void jl_hist1_kernel(const u32x2 *__restrict__ rows,
                                                              const unsigned long long *__restrict__ starts0,
                                                              unsigned parts, unsigned k2,
                                                              unsigned long long *counts1) {
  extern __shared__ unsigned s_hist[];
  const unsigned bucket = blockIdx.x / kJlHist1WgPerBucket, w = blockIdx.x % kJlHist1WgPerBucket;
  const size_t lo = starts0[bucket], hi = starts0[bucket + 1];
  if (lo + static_cast<size_t>(w) * 4 * kJlThreads >= hi) return;
  for (unsigned i = threadIdx.x; i < k2; i += kJlThreads) s_hist[i] = 0;
  __syncthreads();
  for (size_t i = lo + static_cast<size_t>(w) * 4 * kJlThreads + threadIdx.x; i < hi;
       i += static_cast<size_t>(kJlHist1WgPerBucket) * 4 * kJlThreads) {
    unsigned k[4];  // the level-0 output is (key, row id) pairs: the histogram reads them whole (8 bytes per row)
#pragma unroll
    for (int j = 0; j < 4; ++j) k[j] = i + j * kJlThreads < hi ? rows[i + j * kJlThreads].x : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i + j * kJlThreads < hi) atomicAdd(&s_hist[jl_pid(k[j], parts) & (k2 - 1)], 1u);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < k2; i += kJlThreads)
    if (s_hist[i]) atomicAdd(&counts1[static_cast<size_t>(bucket) * k2 + i], static_cast<unsigned long long>(s_hist[i]));
}

// The same histogram from the 16-bit level-1 bucket column the level-0 scatter wrote beside its pairs (jl_scatter_tile
// DIGITS; the 16384-row shape, i.e. 2^28 rows and more): 2 bytes per row instead of 8, and no hash.  Eight digits per
// 16-byte load over the aligned middle of the bucket's range, the ragged ends one digit per lane.
__global__ __launch_bounds__(kJlThreads) void jl_hist1d_kernel(const unsigned short *__restrict__ digits,
                                                               const unsigned long long *__restrict__ starts0, unsigned k2,
                                                               unsigned long long *counts1) {
  extern __shared__ unsigned s_hist[];
  const unsigned bucket = blockIdx.x / kJlHist1WgPerBucket, w = blockIdx.x % kJlHist1WgPerBucket;
  const size_t lo = starts0[bucket], hi = starts0[bucket + 1];
  if (lo >= hi) return;
  for (unsigned i = threadIdx.x; i < k2; i += kJlThreads) s_hist[i] = 0;
  __syncthreads();
  const unsigned mask = k2 - 1;
  // head [lo, a) and tail [b, hi) one digit per lane (first workgroup of the bucket), [a, b) in whole 16-byte vectors
  size_t a = (lo + 7) & ~static_cast<size_t>(7);
  if (a > hi) a = hi;
  size_t b = hi & ~static_cast<size_t>(7);
  if (b < a) b = a;
  if (w == 0) {
    for (size_t i = lo + threadIdx.x; i < a; i += kJlThreads) atomicAdd(&s_hist[digits[i] & mask], 1u);
    for (size_t i = b + threadIdx.x; i < hi; i += kJlThreads) atomicAdd(&s_hist[digits[i] & mask], 1u);
  }
  const u32x4 *vec = reinterpret_cast<const u32x4 *>(digits);
  const size_t va = a / 8, vb = b / 8;
  for (size_t v = va + static_cast<size_t>(w) * 2 * kJlThreads + threadIdx.x; v < vb;
       v += static_cast<size_t>(kJlHist1WgPerBucket) * 2 * kJlThreads) {
    u32x4 x[2];
    const bool second = v + kJlThreads < vb;
    x[0] = vec[v];
    x[1] = second ? vec[v + kJlThreads] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j == 1 && !second) break;
      const unsigned word[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        atomicAdd(&s_hist[word[q] & mask], 1u);
        atomicAdd(&s_hist[(word[q] >> 16) & mask], 1u);
      }
    }
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < k2; i += kJlThreads)
    if (s_hist[i]) atomicAdd(&counts1[static_cast<size_t>(bucket) * k2 + i], static_cast<unsigned long long>(s_hist[i]));
}

// Level-1 scatter: a PERSISTENT grid of the resident workgroups over the "virtual tiles" (every level-0 bucket cut into
// tiles of THREADS x KPT rows; tile_starts[b] = index of bucket b's first tile).  XCD x walks the x-th eighth of the
// virtual tiles, its workgroups interleaved (workgroup j: tiles j, j + per, ...), so the tiles in flight on an XCD
// belong to one or two level-0 buckets, whose k2 write frontiers then meet in ONE L2; a workgroup finds its next tile
// by stepping on from the current one (the bucket changes every few steps: one or two loads), requests its rows
// before the current tile's LDS work and waits for them right before the current tile's stores (the vmcnt rule of
// jl_scatter0_kernel).  Until round 4 this was one tile per workgroup, each starting with a binary search for its
// tile (eight to ten dependent loads) and then its row loads: radix join 2^22 / 2^24 / 2^26 / 2^27 rows 212 / 514 / 1737 /
// 3240 us -> 205 / 496 / 1671 / 3144 (4096-row tiles), 2^30 rows 31.3 -> 26.8 ms (8192-row tiles; 16384-row ones 28.2:
// their 32 prefetched words per lane no longer fit the 128 VGPRs of a 1024-thread workgroup).
template <int THREADS, int KPT>
__global__ __launch_bounds__(THREADS) void jl_scatter1p_kernel(const u32x2 *__restrict__ rows,
                                                               const unsigned long long *__restrict__ starts0,
                                                               const unsigned long long *__restrict__ tile_starts,
                                                               unsigned parts, unsigned k1, unsigned k2,
                                                               unsigned long long *cursors1, u32x2 *__restrict__ out_pairs) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_mem[];
  constexpr unsigned kTile = THREADS * KPT;
  const unsigned long long total = tile_starts[k1];
  const unsigned xcd = blockIdx.x % 8u, j = blockIdx.x / 8u, per = gridDim.x / 8u;  // host: gridDim.x % 8 == 0
  const unsigned long long per_xcd = (total + 7) / 8;
  const unsigned long long vend = (xcd + 1) * per_xcd < total ? (xcd + 1) * per_xcd : total;
  unsigned long long vt = xcd * per_xcd + j;
  if (vt >= vend) return;
  unsigned bucket;
  unsigned long long tile;
  if (!jl_locate(tile_starts, k1, vt, &bucket, &tile)) return;
  auto load_tile = [&](unsigned b, unsigned long long t, unsigned (&k)[KPT], unsigned (&r)[KPT]) {
    const size_t lo = starts0[b] + t * kTile, hi = starts0[b + 1];
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      const size_t idx = lo + static_cast<size_t>(q) * THREADS + threadIdx.x;
      const u32x2 row = idx < hi ? rows[idx] : u32x2{0u, 0u};  // (idx < lo + kTile by construction)
      k[q] = row.x;
      r[q] = row.y;
    }
  };
  unsigned ckey[KPT], crid[KPT];
  load_tile(bucket, tile, ckey, crid);
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    asm volatile("v_mov_b32 %0, %0" : "+v"(ckey[q]));
    asm volatile("v_mov_b32 %0, %0" : "+v"(crid[q]));
  }
  while (true) {
    // the next tile of this workgroup: step on from the current bucket
    const unsigned long long nvt = vt + per;
    const bool more = nvt < vend;
    unsigned nbucket = bucket;
    if (more)
      while (nbucket + 1 < k1 && nvt >= tile_starts[nbucket + 1]) ++nbucket;
    const unsigned long long ntile = more ? nvt - tile_starts[nbucket] : 0ull;
    unsigned nkey[KPT], nrid[KPT], mkey[KPT], mrid[KPT];
    if (more) {
      load_tile(nbucket, ntile, nkey, nrid);
    } else {
#pragma unroll
      for (int q = 0; q < KPT; ++q) nkey[q] = nrid[q] = 0u;
    }
    auto wait_next = [&]() {
#pragma unroll
      for (int q = 0; q < KPT; ++q) {
        asm volatile("v_mov_b32 %0, %1" : "=v"(mkey[q]) : "v"(nkey[q]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(mrid[q]) : "v"(nrid[q]));
      }
    };
    {
      const size_t lo = starts0[bucket] + tile * kTile, hi = starts0[bucket + 1];
      unsigned dest[KPT];
#pragma unroll
      for (int q = 0; q < KPT; ++q) {
        const size_t idx = lo + static_cast<size_t>(q) * THREADS + threadIdx.x;
        dest[q] = idx < hi ? jl_pid(ckey[q], parts) & (k2 - 1) : k2;
      }
      jl_scatter_tile<1, THREADS, KPT, false>(ckey, crid, dest, k2, parts, k2 - 1, cursors1 + static_cast<size_t>(bucket) * k2,
                                              reinterpret_cast<unsigned *>(out_pairs), nullptr, s_mem, wait_next);
    }
    if (!more) break;
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      ckey[q] = mkey[q];
      crid[q] = mrid[q];
    }
    vt = nvt;
    bucket = nbucket;
    tile = ntile;
  }
}

// ---- per-partition build in LDS --------------------------------------------------------------------
// A partition's rows come either as two columns (the level-0 output, when one level suffices) or as (key, row id)
// pairs (the level-1 output): prids == nullptr means pkeys points at pairs.
__device__ __forceinline__ u32x2 jl_row(const unsigned *__restrict__ pkeys, const unsigned *__restrict__ prids,
                                        size_t i) {
  (void)prids;  // both scatter levels write (key, row id) pairs
  return reinterpret_cast<const u32x2 *>(pkeys)[i];
}

// Slot of `key` in the LDS key array `lk`: a plain read first — a key that is already there (a duplicate row) and
// every step of a collision chain need no atomic — ds_cmpst only on a slot read as empty.  Returns kJlSubSlots for
// the sentinel key (flagged in the status word) and for a full sub-table — more distinct keys than slots: *full (an LDS
// word of the caller) is set, the caller hands the partition to the spill path.
__device__ __forceinline__ unsigned jl_claim(unsigned *lk, unsigned key, unsigned *status, unsigned *full) {
  if (key == kEmptyKey) {  // the sentinel is not a key (join/join_omnisci.cpp:52): flag it, drop the row
    atomicOr(status, DBHIP_DEV_KEY_RANGE);
    return kJlSubSlots;
  }
  unsigned s = jl_home_slot(fmix32(key));
  for (unsigned tries = 0; tries < kJlSubSlots; ++tries) {
    unsigned k = lk[s];
    if (k == kEmptyKey) k = atomicCAS(&lk[s], kEmptyKey, key);
    if (k == kEmptyKey || k == key) return s;
    s = jl_next_slot(s);
  }
  *full = 1u;
  return kJlSubSlots;
}

// ---- lanes of a wave that meet on one LDS counter -----------------------------------------------------------------
// A hot key is many rows of a wave on one counter of the sub-table, and the LDS serves the lanes of one atomic that hit
// the same word one after the other.  jl_take: every live lane takes the next value of its counter (UP: returns the old
// value, the counter grows; down: returns old - 1, the counter shrinks); when kJlCrowd lanes or more share the first live
// lane's counter, one of them adds for all and the others derive their values from its result.  `id` names the counter
// (lanes with the same id pass the same pointer).  Every lane of the wave must call it together.
constexpr unsigned kJlCrowd = 16;
template <bool UP>
__device__ __forceinline__ unsigned jl_take(unsigned *counter, unsigned id, bool live) {
  const unsigned lane = threadIdx.x & (kWave - 1);
  const unsigned long long act = __ballot(live);
  if (act == 0) return 0u;  // (uniform)
  const unsigned first = __builtin_amdgcn_readlane(id, __builtin_ctzll(act));
  const bool same = live && id == first;
  const unsigned long long crowd = __ballot(same);
  const unsigned c = static_cast<unsigned>(__builtin_popcountll(crowd));
  if (c >= kJlCrowd) {  // (uniform)
    const unsigned leader = static_cast<unsigned>(__builtin_ctzll(crowd));
    unsigned old = 0;
    if (lane == leader) old = UP ? atomicAdd(counter, c) : atomicSub(counter, c);
    old = __builtin_amdgcn_readlane(old, leader);
    if (same) return UP ? old + mbcnt(crowd) : old - 1u - mbcnt(crowd);
  }
  if (!live) return 0u;
  return UP ? atomicAdd(counter, 1u) : atomicSub(counter, 1u) - 1u;
}

// Barriers of the build kernel order LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the
// next partition's prefetched rows and for every id / table store still on its way to memory (DBHIP_JL_FULL_BARRIER=1
// restores it for A/B timing).
#ifdef DBHIP_JL_FULL_BARRIER
#define JL_BUILD_BARRIER() __syncthreads()
#else
#define JL_BUILD_BARRIER() wg_barrier_lds_only()
#endif
// kMatch = false: the build of the one-to-many join (sub-tables published to HBM for a later probe launch).
// kMatch = true: the radix join — the probe side was partitioned with the SAME geometry, so partition p of S meets
// the sub-table of partition p of R while it is still in LDS: no table is written, no random access leaves the CU;
// results go out in S's partition order together with the probe row ids.
struct JlMatchArgs {
  const u32x2 *spairs;               // probe side, partition-major (key, row id) pairs
  const unsigned long long *sstarts; // parts + 1 offsets
  unsigned *out_rid, *out_pos, *out_cnt;
};
// The giant partitions' scratch (join_common.hpp: jl_giant_bytes): max == 0 switches the path off.
struct JlGiants {
  unsigned *base;           // count, pad[3] | part[max] | done[max] | counts[max][kJlSubSlots] | cursors[max][kJlSubSlots] | tables
  unsigned max;
  unsigned long long rows;  // a partition with more rows than this is a giant
  unsigned long long probe_rows;  // radix join: ... or with more probe rows than this
  unsigned with_tables;     // radix join: the giants' sub-tables are in the scratch (kJlSubSlots + 1 slots each), not in `table`
  __host__ __device__ u32x2 *scratch_table(unsigned g) const {
    return reinterpret_cast<u32x2 *>(cursors(max)) + static_cast<size_t>(g) * (kJlSubSlots + 1);
  }
  __host__ __device__ u32x2 *sub_table(u32x2 *table, unsigned g, size_t part) const {
    if (with_tables) return scratch_table(g);
    return table + part * kJlSubSlots;
  }
  __host__ __device__ unsigned *count() const { return base; }
  __host__ __device__ unsigned *part() const { return base + 4; }
  __host__ __device__ unsigned *done() const { return base + 4 + max; }
  __host__ __device__ unsigned *counts(unsigned g) const {
    return base + ((4 + 2 * static_cast<size_t>(max) + 3) & ~static_cast<size_t>(3)) + static_cast<size_t>(g) * kJlSubSlots;
  }
  __host__ __device__ unsigned *cursors(unsigned g) const { return counts(max) + static_cast<size_t>(g) * kJlSubSlots; }
};
// The spilled partitions' scratch (join_common.hpp).
struct JlSpill {
  unsigned *area;  // dir[parts] {first slot + 1 | 0, slots} | list[max_list] | keys[pool] | pos[pool] | cnt[pool]
  unsigned parts, max_list, list_words;
  unsigned long long pool;
  __host__ __device__ u32x2 *dir() const { return reinterpret_cast<u32x2 *>(area); }
  __host__ __device__ unsigned *list() const { return area + 2 * static_cast<size_t>(parts); }
  __host__ __device__ unsigned *keys() const { return list() + list_words; }
  __host__ __device__ unsigned *pos() const { return keys() + pool; }
  __host__ __device__ unsigned *cnt() const { return pos() + pool; }
};
// (a pure function of the build's partition and row counts: the build kernels take only the area's address — their scalar
//  registers are all in use — and derive the rest where the rare path needs it)
__host__ __device__ __forceinline__ JlSpill jl_spill_of(void *area, unsigned parts, size_t n) {
  return JlSpill{static_cast<unsigned *>(area), parts, jl_max_spill(n), static_cast<unsigned>(jl_spill_list_words(n)),
                 static_cast<unsigned long long>(jl_spill_pool_slots(n))};
}
__device__ __forceinline__ unsigned jl_spill_home(unsigned key, unsigned cap) {
  return static_cast<unsigned>((static_cast<unsigned long long>(fmix32(key * 0x9E3779B1u ^ 0x7F4A7C15u)) * cap) >> 32);
}
__device__ __forceinline__ unsigned jl_ld(const unsigned *p) {  // a read that no stale line of this CU's L1 can answer
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// {first id position, count} of `key` in the spilled partition's table [base, base + cap), {0, 0} if it is not there
__device__ __forceinline__ void jl_spill_lookup(const JlSpill &sp, unsigned base, unsigned cap, unsigned key, unsigned *pos_out,
                                                unsigned *cnt_out) {
  *pos_out = *cnt_out = 0;
  if (key == kEmptyKey) return;
  const unsigned *keys = sp.keys() + base;
  unsigned s = jl_spill_home(key, cap);
  for (unsigned tries = 0; tries < cap; ++tries) {
    const unsigned k = jl_ld(&keys[s]);
    if (k == key) {
      *pos_out = jl_ld(&sp.pos()[base + s]);
      *cnt_out = jl_ld(&sp.cnt()[base + s]);
      return;
    }
    if (k == kEmptyKey) return;
    s = s + 1 == cap ? 0u : s + 1;
  }
}
// One workgroup (kJlGiantThreads = kJlBuildThreads threads, all of them call this together) builds the table of partition
// `part` — rows [lo, hi) of the partition-major pairs — in the pool and writes the partition's ids[lo, hi); radix join
// (match.spairs != nullptr): it then answers the partition's probe rows.  Correct for any rows; not written to be fast:
// every step is a memory-side atomic (a hot key of such a partition is served at one atomic per ~11 ns).
// `sub` (one-to-many build, a giant that spilled): the partition's sub-table of the published table, which jl_giant_count
// left half-claimed — it is rewritten as an empty, MARKED one (jl_probe_sub; the build kernel publishes a spilled
// partition of its own that way itself).
__device__ __noinline__ void jl_spill_partition(const JlSpill sp, const u32x2 *__restrict__ rows, size_t lo, size_t hi, unsigned part,
                                                unsigned *__restrict__ ids, unsigned *status, const JlMatchArgs match, size_t slo,
                                                size_t shi, u32x2 *sub, unsigned pos_bits) {
  constexpr unsigned kT = kJlBuildThreads;
  __shared__ unsigned s_base, s_w[kT / kWave];
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const unsigned cap = static_cast<unsigned>(jl_spill_cap(hi - lo));
  __syncthreads();
  if (tid == 0) s_base = atomicAdd(status + kJlHdrSpillPool, cap);
  __syncthreads();
  const unsigned base = s_base;
  if (static_cast<unsigned long long>(base) + cap > sp.pool) {  // (cannot happen: the pool holds every partition that can spill)
    if (tid == 0) atomicOr(status, DBHIP_DEV_TABLE_FULL);
    return;
  }
  unsigned *keys = sp.keys() + base, *pos = sp.pos() + base, *cnt = sp.cnt() + base;
  for (unsigned i = tid; i < cap; i += kT) {
    keys[i] = kEmptyKey;
    cnt[i] = 0;
  }
  __threadfence();
  __syncthreads();
  // 1. distinct keys and their row counts
  for (size_t i = lo + tid; i < hi; i += kT) {
    const unsigned key = rows[i].x;
    if (key == kEmptyKey) {  // the sentinel is not a key (join/join_omnisci.cpp:52): flagged, the row is dropped
      atomicOr(status, DBHIP_DEV_KEY_RANGE);
      continue;
    }
    unsigned s = jl_spill_home(key, cap);
    while (true) {  // (cap > rows >= distinct keys: an empty slot exists)
      const unsigned k = atomicCAS(&keys[s], kEmptyKey, key);
      if (k == kEmptyKey || k == key) break;
      s = s + 1 == cap ? 0u : s + 1;
    }
    atomicAdd(&cnt[s], 1u);
  }
  __threadfence();
  __syncthreads();
  // 2. first id position of every key: exclusive scan of the counts in slot order, from the partition's first position
  unsigned run = static_cast<unsigned>(lo);
  for (unsigned c0 = 0; c0 < cap; c0 += kT) {  // (uniform)
    const unsigned i = c0 + tid;
    const unsigned c = i < cap && jl_ld(&keys[i]) != kEmptyKey ? jl_ld(&cnt[i]) : 0u;
    const unsigned incl = wave_inclusive_scan(c);
    if (lane == kWave - 1) s_w[wave] = incl;
    __syncthreads();
    unsigned before = 0, total = 0;
    for (unsigned w = 0; w < kT / kWave; ++w) {
      before += w < wave ? s_w[w] : 0u;
      total += s_w[w];
    }
    if (i < cap) pos[i] = run + before + incl - c;
    run += total;
    __syncthreads();
  }
  __threadfence();
  __syncthreads();
  // 3. ids: every row takes the next place of its key's range (the position word is the cursor; step 4 puts it back)
  for (size_t i = lo + tid; i < hi; i += kT) {
    const u32x2 row = rows[i];
    if (row.x == kEmptyKey) continue;
    unsigned s = jl_spill_home(row.x, cap);
    while (jl_ld(&keys[s]) != row.x) s = s + 1 == cap ? 0u : s + 1;
    ids[atomicAdd(&pos[s], 1u)] = row.y;
  }
  __threadfence();
  __syncthreads();
  for (unsigned i = tid; i < cap; i += kT)
    if (jl_ld(&keys[i]) != kEmptyKey) pos[i] = jl_ld(&pos[i]) - jl_ld(&cnt[i]);
  __threadfence();
  __syncthreads();
  if (tid == 0) sp.dir()[part] = u32x2{base + 1u, cap};
  if (sub != nullptr) {
    const unsigned word = static_cast<unsigned>(lo) | (pos_bits < 32 ? 1u << pos_bits : 0u);
    for (unsigned i = tid; i < kJlSubSlots; i += kT) sub[i] = u32x2{kEmptyKey, word};
  }
  // 4. radix join: the partition's probe rows
  if (match.spairs != nullptr)
    for (size_t j = slo + tid; j < shi; j += kT) {
      const u32x2 row = match.spairs[j];
      unsigned p, c;
      jl_spill_lookup(sp, base, cap, row.x, &p, &c);
      match.out_rid[j] = row.y;
      match.out_pos[j] = p;
      match.out_cnt[j] = c;
    }
  __syncthreads();
}
// a workgroup whose partition does not fit its sub-table lists it for the spill path (thread 0)
__device__ __forceinline__ void jl_spill_list(const JlSpill &sp, unsigned part, unsigned *status) {
  const unsigned g = atomicAdd(status + kJlHdrSpilled, 1u);
  if (g < sp.max_list) sp.list()[g] = part;
  else atomicOr(status, DBHIP_DEV_TABLE_FULL);  // (cannot happen: fewer partitions can spill than the list holds)
}

// kInline: the workgroup that finds its partition overfull builds the spill table itself, at once (small inputs: no
// launch behind this one looks at the list); otherwise it lists the partition for the tail of jl_giant_ids_kernel.
template <bool kMatch, bool kInline>
__global__ __launch_bounds__(kJlBuildThreads) __attribute__((amdgpu_waves_per_eu(kInline ? 4 : 6))) void jl_build_kernel(const unsigned *__restrict__ pkeys,
                                                                   const unsigned *__restrict__ prids,
                                                                   const unsigned long long *__restrict__ starts,
                                                                   u32x2 *__restrict__ table, unsigned parts,
                                                                   unsigned n_rows, unsigned pos_bits,
                                                                   unsigned *__restrict__ ids, unsigned *status,
                                                                   JlMatchArgs match, JlGiants giants, unsigned *spill_area) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_lds[];
  unsigned *lk = s_lds;                // keys
  unsigned *lc = s_lds + kJlSubSlots;  // counts in step 1; the scan turns the same words into positions:
  unsigned *lp = lc;                   // first id position of the slot, bumped to its end by the fill.
  // (two arrays instead of three: 24 KiB per workgroup at 3072 slots)
  __shared__ unsigned s_wsum[kJlBuildThreads / kWave];
  __shared__ unsigned s_end;  // where the last slot's id range ends = first position behind the partition's counted rows
  __shared__ unsigned s_ticket;  // the ticket thread 0 took for the partition after the next one
  __shared__ unsigned s_full;    // a row of this partition found the sub-table full of other keys
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  // Partitions by ticket (round 4).  A static deal (partition p, p + grid, ...) ends when the SLOWEST workgroup has walked
  // its share, and workgroups are not equally fast (CUs share L2 slices and memory channels unevenly): the fused kernel
  // took 669 us for the 37504 partitions of 2^26 rows, 49 per workgroup, and 8903 us / 16 = 556 us per 2^26 rows at 2^30,
  // 781 per workgroup, where the differences average out.  The first partition is blockIdx, every further one comes from
  // a counter in the workspace header (status[32]; status[33] counts the workgroups that have left: the last one zeroes
  // both for the next launch).  The ticket is taken one step ahead of the prefetch, by thread 0, and handed over
  // through LDS where the prefetched rows are waited for — a returning atomic counts in vmcnt like a load.
  unsigned *const ticket_word = status + kJlHdrTicket, *const left_word = status + kJlHdrLeft;
  // a partition holds ~kJlRowsPerPart rows (+ 6 sigma of a Poisson count): rows per thread whose slot and row id
  // stay in registers between steps 1 and 3, and rows per thread loaded one partition ahead
  constexpr unsigned kRowsPerPart = kMatch ? kJrRowsPerPart : kJlRowsPerPart;  // (the radix join partitions for fewer rows)
  constexpr int kJlCached = static_cast<int>((kRowsPerPart + kRowsPerPart / 8 + kJlBuildThreads - 1) / kJlBuildThreads);
  constexpr int kJlPre = kJlCached;  // every cached row comes from the prefetch: a row loaded inside the claim phase made
                                     // the compiler wait (s_waitcnt vmcnt(0)) for the prefetch issued just before it too
  static_assert(static_cast<unsigned>(kJlCached) * kJlBuildThreads <= kJlSubSlots, "a partition without overflow rows fits the id staging area");

  // Persistent workgroups walk the partitions with a stride of the grid; a workgroup is a chain of dependent
  // phases (load, claim, scan, fill, publish), so the NEXT partition's rows are requested before the current
  // partition's LDS work starts and arrive while it runs.
  unsigned part = blockIdx.x;
  if (part >= parts) return;
  unsigned lo = static_cast<unsigned>(starts[part]), hi = static_cast<unsigned>(starts[part + 1]);  // (32-bit: n <= 2^31 rows; the kernel is short of scalar registers)
  unsigned npart = parts;  // the partition after this one
  {
    if (tid == 0) s_ticket = atomicAdd(ticket_word, 1u);
    __syncthreads();
    npart = static_cast<unsigned>(gridDim.x) + s_ticket;
    __syncthreads();
  }
  u32x2 carry[kJlPre];  // the current partition's rows, in registers that no load is pending on
#pragma unroll
  for (int r = 0; r < kJlPre; ++r) {
    const unsigned i = lo + tid + static_cast<unsigned>(r) * kJlBuildThreads;
    carry[r] = i < hi ? jl_row(pkeys, prids, i) : u32x2{0u, 0u};
  }
#pragma unroll
  for (int r = 0; r < kJlPre; ++r) {  // (see below: the loop must not carry a register that a load defined)
    asm volatile("v_mov_b32 %0, %0" : "+v"(carry[r].x));
    asm volatile("v_mov_b32 %0, %0" : "+v"(carry[r].y));
  }
  while (true) {
    // the offsets this step needs, requested together: the next partition's rows and (radix join) this partition's probe rows
    unsigned nlo = 0, nhi = 0, slo = 0, shi = 0;
    unsigned tk = 0;
    if (npart < parts) {
      nlo = static_cast<unsigned>(starts[npart]);
      nhi = static_cast<unsigned>(starts[npart + 1]);
      if (tid == 0) tk = atomicAdd(ticket_word, 1u);  // for the step after the next (consumed below, with the prefetch)
    }
    if (kMatch) {
      slo = static_cast<unsigned>(match.sstarts[part]);
      shi = static_cast<unsigned>(match.sstarts[part + 1]);
    }
    if (giants.max != 0 && (hi - lo > giants.rows || (kMatch && shi - slo > giants.probe_rows))) {  // (uniform)
      // a giant partition (join_common.hpp): listed for the jl_giant_* kernels, which all workgroups share.  Build: it is
      // published here as an EMPTY sub-table — the state their claims of its slots start from.  Radix join: nothing of
      // it is done here, neither side; its sub-table will be one of the scratch tables, emptied here.
      if (tid == 0) s_end = atomicAdd(giants.count(), 1u);
      JL_BUILD_BARRIER();
      const unsigned slot = s_end;
      JL_BUILD_BARRIER();  // (s_end is written again below)
      if (slot < giants.max) {  // always: the list is sized for every partition that can be this large
        if (tid == 0) giants.part()[slot] = static_cast<unsigned>(part);
        unsigned *gc = giants.counts(slot);
        for (unsigned i = tid; i < kJlSubSlots; i += kJlBuildThreads) gc[i] = 0;
        if (kMatch) {
          u32x2 *sub = giants.scratch_table(slot);
          for (unsigned i = tid; i < kJlSubSlots + 1; i += kJlBuildThreads) sub[i] = u32x2{kEmptyKey, 0u};
          shi = slo;
        }
        hi = lo;
      }
    }
    for (unsigned i = tid; i < kJlSubSlots; i += kJlBuildThreads) {
      lk[i] = kEmptyKey;
      lc[i] = 0;
    }
    if (tid == 0) {
      s_full = 0;
      reinterpret_cast<u32x2 *>(spill_area)[part] = u32x2{0u, 0u};  // "not spilled" (every partition passes here exactly once per build)
    }
    u32x2 cur[kJlPre];
#pragma unroll
    for (int r = 0; r < kJlPre; ++r) cur[r] = carry[r];
    u32x2 pre[kJlPre];  // the next partition's rows, in flight
#pragma unroll
    for (int r = 0; r < kJlPre; ++r) pre[r] = u32x2{0u, 0u};
    if (npart < parts) {
#pragma unroll
      for (int r = 0; r < kJlPre; ++r) {
        const unsigned i = nlo + tid + static_cast<unsigned>(r) * kJlBuildThreads;
        if (i < nhi) pre[r] = jl_row(pkeys, prids, i);
      }
    }
    // radix join: this partition's first probe rows (four consecutive ones per lane) are requested now and arrive
    // while the sub-table is built
    u32x2 srow[4] = {u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}};
    if (kMatch) {
      const unsigned j0 = slo + 4 * static_cast<unsigned>(tid);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j0 + q < shi) srow[q] = match.spairs[j0 + q];
    }
    JL_BUILD_BARRIER();
    // 1. claim the key's slot and take the row's rank inside its key group.  LDS atomics are what bounds this kernel
    //    (measured with the output stores compiled out: 400 of its 550 us at 2^26 rows), so a row costs two of them,
    //    not three: a plain read of the slot first (a duplicate of an already published key and every step of a
    //    collision chain need no ds_cmpst), ds_cmpst only on a slot read as empty, and ONE returning ds_add whose
    //    old value is the row's rank — the fill then needs no second atomic.  Two attempts at overlapping the LDS round
    //    trips of a lane's rows both measured SLOWER: a loop of rounds (every row one probe step per round: 1583 vs 1186 us
    //    for the 2^26 build) and a staged form (all first reads, then all first ds_cmpst, stragglers one by one, then all
    //    rank atomics: 1194 vs 1110 us; the fused match kernel 940 vs 870 us) — in-kernel stamps put this phase at a
    //    median 3500 of a 6500-cycle partition step with a 12000-cycle 90th percentile: the CU's LDS pipeline, shared
    //    by four workgroups of random-address traffic, is what the rows queue on, not their own dependency chain.
    unsigned c_slot[kJlCached], c_rid[kJlCached], c_rank[kJlCached];
#pragma unroll
    for (int r = 0; r < kJlCached; ++r) {
      const unsigned i = lo + tid + static_cast<unsigned>(r) * kJlBuildThreads;
      c_slot[r] = kJlSubSlots;  // "no row"
      c_rid[r] = 0;
      c_rank[r] = 0;
      if (i < hi) {
        const u32x2 row = r < kJlPre ? cur[r < kJlPre ? r : 0] : jl_row(pkeys, prids, i);
        c_rid[r] = row.y;
        const unsigned s = jl_claim(lk, row.x, status, &s_full);
        if (s < kJlSubSlots) {
          c_slot[r] = s;
          c_rank[r] = atomicAdd(&lc[s], 1u);
        }
      }
    }
    // a partition far above its expected size: the rows beyond the cached ones are counted AFTER every cached row has
    // its rank (so the cached rows of a key hold the ranks 0 .. k-1 and these rows own the ranks behind them)
    const bool overflow = hi - lo > static_cast<unsigned>(kJlCached) * kJlBuildThreads;  // uniform over the workgroup
    if (overflow) {
      // (four rows per lane in flight and crowds of one key added by one lane, round 3: a partition of 2^16 rows of one
      //  key took a workgroup 2.6 ns per row with one load in flight and every lane's atomic on the same word)
      JL_BUILD_BARRIER();
      for (unsigned i0 = lo + static_cast<unsigned>(kJlCached) * kJlBuildThreads; i0 < hi; i0 += 4 * static_cast<unsigned>(kJlBuildThreads)) {
        u32x2 row[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned i = i0 + static_cast<unsigned>(q) * kJlBuildThreads + tid;
          row[q] = i < hi ? jl_row(pkeys, prids, i) : u32x2{0u, 0u};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool have = i0 + static_cast<unsigned>(q) * kJlBuildThreads + tid < hi;
          const unsigned s = have ? jl_claim(lk, row[q].x, status, &s_full) : kJlSubSlots;
          (void)jl_take<true>(&lc[s < kJlSubSlots ? s : 0u], s, s < kJlSubSlots);
        }
      }
    }
    JL_BUILD_BARRIER();
    const bool spilled = s_full != 0u;
    if (spilled) {  // (uniform) more distinct keys than slots: the spill path builds this partition; here it becomes an
                    // empty one — cleared counters scan to "every position = lo", which is what gets published; the rows'
                    // stale slot / rank registers reach nothing (no counted rows are written out; guards below)
      const JlSpill spill = jl_spill_of(spill_area, parts, n_rows);
      if (kInline) {
        jl_spill_partition(spill, reinterpret_cast<const u32x2 *>(pkeys), lo, hi, static_cast<unsigned>(part), ids, status, match,
                           slo, shi, nullptr, pos_bits);
        if (tid == 0) atomicAdd(status + kJlHdrSpilled, 1u);
      } else if (tid == 0) {
        jl_spill_list(spill, static_cast<unsigned>(part), status);
      }
      // (a build of exactly 2^31 rows has no count field to mark a spilled sub-table with: that one size still fails loudly)
      if (!kMatch && pos_bits >= 32 && tid == 0) atomicOr(status, DBHIP_DEV_TABLE_FULL);
      JL_BUILD_BARRIER();
      for (unsigned i = tid; i < kJlSubSlots; i += kJlBuildThreads) {
        lk[i] = kEmptyKey;
        lc[i] = 0;
      }
      JL_BUILD_BARRIER();
    }
    // 2. exclusive scan of the counts -> first id position of every slot
    constexpr unsigned kPer = kJlSubSlots / kJlBuildThreads;
    unsigned c[kPer], mine = 0;
#pragma unroll
    for (unsigned j = 0; j < kPer; ++j) {
      c[j] = lc[tid * kPer + j];
      mine += c[j];
    }
    const unsigned incl = wave_inclusive_scan(mine);
    if (lane == kWave - 1) s_wsum[wave] = incl;
    JL_BUILD_BARRIER();
    unsigned run = static_cast<unsigned>(lo) + incl - mine;
    for (unsigned w = 0; w < wave; ++w) run += s_wsum[w];
    // every thread has its counts in registers (the barrier above): overwrite them with the positions
#pragma unroll
    for (unsigned j = 0; j < kPer; ++j) {
      lp[tid * kPer + j] = run;
      run += c[j];
    }
    if (tid == kJlBuildThreads - 1) s_end = run;
    JL_BUILD_BARRIER();
    // vmcnt counts loads and stores of a wave in issue order: the next partition's rows (requested at the top of this
    // iteration) are waited for HERE, while nothing but those loads is outstanding — at the top of the next iteration
    // the wait would also cover every store issued below, i.e. a full store drain per partition
    // (the rows move on through a v_mov the compiler cannot see through: a loop-carried register that a load defined
    //  is waited for with vmcnt(0) at the loop head whatever happened in between)
    u32x2 nxt[kJlPre];
#pragma unroll
    for (int r = 0; r < kJlPre; ++r) {
      asm volatile("v_mov_b32 %0, %1" : "=v"(nxt[r].x) : "v"(pre[r].x));
      asm volatile("v_mov_b32 %0, %1" : "=v"(nxt[r].y) : "v"(pre[r].y));
    }
    if (tid == 0 && npart < parts) {  // (read by everyone behind the barrier that closes this step)
      unsigned got;
      asm volatile("v_mov_b32 %0, %1" : "=v"(got) : "v"(tk));
      s_ticket = got;
    }
    if (!kMatch) {
    // 3. publish the sub-table: {key, first position | count field} for every slot.  Positions are an exclusive scan
    //    in slot order, so slot i ends where slot i+1 starts (the last one at s_end).  Output stores are what this
    //    kernel waits for (per-instruction issue cost: with the stores compiled out it ran 150 us shorter at 2^26
    //    rows), so every store instruction carries 16 bytes per lane: two slots here, four ids below.  Written once,
    //    read by another launch: non-temporal.
    u32x4 *dst = reinterpret_cast<u32x4 *>(table + static_cast<size_t>(part) * kJlSubSlots);  // 16-byte aligned: kJlSubSlots is even
    const unsigned cnt_esc = pos_bits < 32 ? (1u << (32 - pos_bits)) - 1u : 0u;
    for (unsigned i = tid; i < kJlSubSlots / 2; i += kJlBuildThreads) {
      const unsigned p0 = lp[2 * i], p1 = lp[2 * i + 1], p2 = 2 * i + 2 < kJlSubSlots ? lp[2 * i + 2] : s_end;
      const unsigned c0 = p1 - p0, c1 = p2 - p1;
      unsigned f0 = c0 ? c0 - 1 : 0u, f1 = c1 ? c1 - 1 : 0u;
      f0 = f0 < cnt_esc ? f0 : cnt_esc;
      f1 = f1 < cnt_esc ? f1 : cnt_esc;
      if (spilled) f0 = f1 = 1u;  // (uniform) the mark on a spilled partition's empty slots: see jl_probe_sub
      const unsigned w0 = pos_bits < 32 ? (p0 | (f0 << pos_bits)) : p0, w1 = pos_bits < 32 ? (p1 | (f1 << pos_bits)) : p1;
      __builtin_nontemporal_store(u32x4{lk[2 * i], w0, lk[2 * i + 1], w1}, dst + i);
    }
    if (part + 1 == parts && tid == 0) table[static_cast<size_t>(parts) * kJlSubSlots] = u32x2{kEmptyKey, n_rows};  // sentinel
    } else {
    // 3'. probe: the rows of S's partition `part` against the sub-table in LDS — plain LDS reads; the results
    //     {probe row id, first id position, count} leave coalesced in S's partition order
      //     FOUR consecutive rows per lane: one 16-byte store per output column and lane (4-byte stores made this
      //     kernel store-issue-bound: 914 us at 2^26 x 2^26 rows)
      typedef unsigned u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
      const unsigned jfirst = slo + 4 * static_cast<unsigned>(tid);
      if (!spilled)  // (a spilled partition's probe rows are answered by jl_spill_partition)
      for (unsigned j0 = jfirst; j0 < shi; j0 += 4 * static_cast<unsigned>(kJlBuildThreads)) {
        unsigned rid[4], pos[4], cnt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          rid[q] = pos[q] = cnt[q] = 0;
          if (j0 + q < shi) {
            const u32x2 row = j0 == jfirst ? srow[q] : match.spairs[j0 + q];
            const unsigned key = row.x;
            rid[q] = row.y;
            unsigned sl = jl_home_slot(fmix32(key));
            for (unsigned tries = 0; tries < kJlSubSlots && key != kEmptyKey; ++tries) {
              const unsigned k = lk[sl];
              if (k == key) {
                pos[q] = lp[sl];
                cnt[q] = (sl + 1 < kJlSubSlots ? lp[sl + 1] : s_end) - pos[q];
                break;
              }
              if (k == kEmptyKey) break;
              sl = jl_next_slot(sl);
            }
          }
        }
        if (j0 + 3 < shi) {
          *reinterpret_cast<u32x4_a4 *>(match.out_rid + j0) = u32x4{rid[0], rid[1], rid[2], rid[3]};
          *reinterpret_cast<u32x4_a4 *>(match.out_pos + j0) = u32x4{pos[0], pos[1], pos[2], pos[3]};
          *reinterpret_cast<u32x4_a4 *>(match.out_cnt + j0) = u32x4{cnt[0], cnt[1], cnt[2], cnt[3]};
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (j0 + q < shi) {
              match.out_rid[j0 + q] = rid[q];
              match.out_pos[j0 + q] = pos[q];
              match.out_cnt[j0 + q] = cnt[q];
            }
        }
      }
    }
    // 4. fill: id position = first position of the slot + rank of the row (slot / rank / row id from the registers of
    //    step 1: a plain LDS read, no atomic).  The partition's ids are ONE contiguous range [lo, s_end): they are
    //    staged in LDS — in the key array, free once it is published — and leave as 16-byte stores.  A partition far
    //    above its expected size (rows beyond the cached ones, `overflow`) stores every id directly instead.
    unsigned c_pos[kJlCached];
#pragma unroll
    for (int r = 0; r < kJlCached; ++r) c_pos[r] = c_slot[r] < kJlSubSlots ? lp[c_slot[r]] + c_rank[r] : 0xFFFFFFFFu;
    if (!overflow) {
      JL_BUILD_BARRIER();  // every key and position word has been read: the key array becomes the staging area
      unsigned *stage = lk;
      const unsigned lo32 = static_cast<unsigned>(lo);
#pragma unroll
      for (int r = 0; r < kJlCached; ++r)
        if (c_pos[r] != 0xFFFFFFFFu) stage[c_pos[r] - lo32] = c_rid[r];  // rows <= cached rows <= kJlSubSlots (static_assert)
      JL_BUILD_BARRIER();
      const unsigned rows = s_end - lo32;  // counted rows of the partition
      unsigned *out = ids + lo;
      const unsigned head0 = ((16u - (static_cast<unsigned>(reinterpret_cast<uintptr_t>(out)) & 15u)) & 15u) / 4u;  // up to the next 16-byte boundary
      const unsigned head = head0 < rows ? head0 : rows;
      if (tid < head) out[tid] = stage[tid];
      const unsigned body = (rows - head) / 4;
      for (unsigned v = tid; v < body; v += kJlBuildThreads) {
        const unsigned at = head + 4 * v;
        __builtin_nontemporal_store(u32x4{stage[at], stage[at + 1], stage[at + 2], stage[at + 3]},
                                    reinterpret_cast<u32x4 *>(out + at));
      }
      const unsigned tail0 = head + 4 * body;
      if (tid < 4 && tail0 + tid < rows) out[tail0 + tid] = stage[tail0 + tid];
    } else if (!spilled) {
#pragma unroll
      for (int r = 0; r < kJlCached; ++r)
        if (c_pos[r] != 0xFFFFFFFFu) ids[c_pos[r]] = c_rid[r];
      // the uncached rows of a key own the ranks behind its cached rows: they take them from the END of the key's id
      // range downwards, decrementing a cursor that starts at the next slot's first position (for the last slot:
      // s_end) — after a barrier: the publish and the cached rows above still needed those words intact
      JL_BUILD_BARRIER();
      for (unsigned i0 = lo + static_cast<unsigned>(kJlCached) * kJlBuildThreads; i0 < hi; i0 += 4 * static_cast<unsigned>(kJlBuildThreads)) {
        u32x2 row[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned i = i0 + static_cast<unsigned>(q) * kJlBuildThreads + tid;
          row[q] = i < hi ? jl_row(pkeys, prids, i) : u32x2{kEmptyKey, 0u};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned key = row[q].x;
          unsigned s = kJlSubSlots;
          if (key != kEmptyKey) {  // (a missing row reads as the sentinel key)
            s = jl_home_slot(fmix32(key));
            for (unsigned tries = 0; tries < kJlSubSlots && lk[s] != key; ++tries) s = jl_next_slot(s);
            if (lk[s] != key) s = kJlSubSlots;
          }
          const bool live = s < kJlSubSlots;
          unsigned *cursor = live && s + 1 < kJlSubSlots ? &lp[s + 1] : &s_end;
          const unsigned at = jl_take<false>(cursor, s, live);
          if (live) ids[at] = row[q].y;
        }
      }
    }
    if (npart >= parts) break;
    JL_BUILD_BARRIER();  // the LDS arrays and s_wsum are reused by the next partition
    part = npart;
    npart = static_cast<unsigned>(gridDim.x) + s_ticket;
    lo = nlo;
    hi = nhi;
#pragma unroll
    for (int r = 0; r < kJlPre; ++r) carry[r] = nxt[r];
  }
  // every ticket this workgroup asked for has been answered (each was consumed above): the last workgroup to leave
  // resets the two words, so a launch finds them zero without a fill in front of it
  if (tid == 0 && atomicAdd(left_word, 1u) + 1u == gridDim.x) {
    __hip_atomic_store(ticket_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(left_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- giant partitions: all workgroups together, slice by slice ----------------------------------------------------
// jl_build_kernel leaves a partition above giants.rows rows out (an empty sub-table in its place) and lists it.  Such a
// partition is a few hot keys — any number of rows, still at most kJlSubSlots distinct keys — and one workgroup walks
// 0.38 G rows/s.  Two launches over slices of kJlSlice rows of the listed partitions:
//   jl_giant_count  a slice's rows are counted per key in an LDS sub-table (as the build does); every distinct key of the
//                   slice then claims its slot of the partition's sub-table IN HBM (atomicCAS on the empty slots the build
//                   published; same home slot, same linear probing: the probe kernel reads it like any other sub-table)
//                   and adds the slice's count to that slot's counter — one memory-side atomic per slice and key, not per
//                   row.  The workgroup that finishes a giant's LAST slice (a counter per giant) scans the slot counters:
//                   first id position of every slot -> the table's position words and the slot cursors.
//   jl_giant_ids    the slices are counted again the same way; per distinct key ONE returning add on the slot's cursor
//                   reserves the slice's share of the key's id range, then every row takes its place inside the share
//                   from an LDS cursor and stores its row id.
// Lanes of a wave that meet on one LDS counter (that is what a hot key is) are added by one lane: jl_take.
// Nothing here waits for another workgroup.
constexpr unsigned kJlSlice = 8192;        // rows of a slice: 16 per thread
constexpr int kJlGiantThreads = 512;
// the slices of the listed giants as one flat list: s_first[g] = index of giant g's first slice (s_first[ng] = total)
__device__ __forceinline__ unsigned jl_giant_slices(const JlGiants &giants, const unsigned long long *__restrict__ starts,
                                                    unsigned *s_first, unsigned *s_wsum, unsigned *ng_out) {
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  if (giants.max == 0) {  // (uniform) DBHIP_JL_NO_GIANTS=1: no list, this launch only serves the spilled partitions
    *ng_out = 0;
    return 0u;
  }
  unsigned ng = *giants.count();
  ng = ng < giants.max ? ng : giants.max;
  ng = ng < kJlMaxGiantList - 1 ? ng : kJlMaxGiantList - 1;
  *ng_out = ng;
  if (ng == 0) return 0u;
  unsigned total = 0;
  for (unsigned g0 = 0; g0 < ng; g0 += kJlGiantThreads) {  // (uniform trip count)
    const unsigned g = g0 + tid;
    unsigned mine = 0;
    if (g < ng) {
      const unsigned p = giants.part()[g];
      mine = static_cast<unsigned>((starts[p + 1] - starts[p] + kJlSlice - 1) / kJlSlice);
    }
    const unsigned incl = wave_inclusive_scan(mine);
    if (lane == kWave - 1) s_wsum[wave] = incl;
    __syncthreads();
    unsigned before = total;
    for (unsigned w = 0; w < wave; ++w) before += s_wsum[w];
    if (g < ng) s_first[g] = before + incl - mine;
    for (unsigned w = 0; w < kJlGiantThreads / kWave; ++w) total += s_wsum[w];
    __syncthreads();
  }
  if (tid == 0) s_first[ng] = total;
  __syncthreads();
  return total;
}
// giant of flat slice `item` (largest g with s_first[g] <= item)
__device__ __forceinline__ unsigned jl_giant_of(const unsigned *s_first, unsigned ng, unsigned item) {
  unsigned lo = 0, hi = ng;
  while (hi - lo > 1) {
    const unsigned mid = (lo + hi) / 2;
    if (s_first[mid] <= item) lo = mid; else hi = mid;
  }
  return lo;
}
// count the rows [a, b) per key in the LDS sub-table lk / lc (cleared here)
__device__ __forceinline__ void jl_slice_count(const u32x2 *__restrict__ rows, size_t a, size_t b, unsigned *lk, unsigned *lc,
                                               unsigned *status, unsigned *full) {
  const unsigned tid = threadIdx.x;
  for (unsigned i = tid; i < kJlSubSlots; i += kJlGiantThreads) {
    lk[i] = kEmptyKey;
    lc[i] = 0;
  }
  __syncthreads();
  for (size_t i0 = a; i0 < b; i0 += 4 * static_cast<size_t>(kJlGiantThreads)) {  // (uniform) four rows per lane in flight
    u32x2 r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t i = i0 + static_cast<size_t>(q) * kJlGiantThreads + tid;
      r[q] = i < b ? rows[i] : u32x2{kEmptyKey, 0u};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool have = i0 + static_cast<size_t>(q) * kJlGiantThreads + tid < b;
      const unsigned sl = have ? jl_claim(lk, r[q].x, status, full) : kJlSubSlots;
      (void)jl_take<true>(&lc[sl < kJlSubSlots ? sl : 0u], sl, sl < kJlSubSlots);
    }
  }
  __syncthreads();
}
// slot of `key` in the giant's sub-table in HBM; claim = true: take the first empty slot of its chain if it is not there
// yet.  kJlSubSlots: the sub-table is full of other keys (*full is set: the giant goes to the spill path) / the key is
// not there.
__device__ __forceinline__ unsigned jl_giant_slot(u32x2 *sub, unsigned key, bool claim, unsigned *full) {
  unsigned s = jl_home_slot(fmix32(key));
  for (unsigned tries = 0; tries < kJlSubSlots; ++tries) {
    unsigned *kw = reinterpret_cast<unsigned *>(sub + s);
    unsigned k = __builtin_nontemporal_load(kw);
    if (k == kEmptyKey) {
      if (!claim) return kJlSubSlots;
      k = atomicCAS(kw, kEmptyKey, key);
      if (k == kEmptyKey) return s;
    }
    if (k == key) return s;
    s = jl_next_slot(s);
  }
  if (claim) *full = 1u;
  return kJlSubSlots;
}

__global__ __launch_bounds__(kJlGiantThreads) void jl_giant_count_kernel(const u32x2 *__restrict__ rows,
                                                                         const unsigned long long *__restrict__ starts,
                                                                         u32x2 *table, unsigned pos_bits, JlGiants giants,
                                                                         unsigned *status, JlSpill spill) {
  __shared__ unsigned lk[kJlSubSlots], lc[kJlSubSlots];
  __shared__ unsigned s_first[kJlMaxGiantList], s_wsum[kJlGiantThreads / kWave], s_last, s_full;
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  unsigned ng;
  const unsigned items = jl_giant_slices(giants, starts, s_first, s_wsum, &ng);
  for (unsigned item = blockIdx.x; item < items; item += gridDim.x) {
    const unsigned g = jl_giant_of(s_first, ng, item);
    const unsigned part = giants.part()[g];
    const size_t lo = starts[part], hi = starts[part + 1];
    const size_t a = lo + static_cast<size_t>(item - s_first[g]) * kJlSlice;
    const size_t b = a + kJlSlice < hi ? a + kJlSlice : hi;
    if (tid == 0) s_full = 0;
    jl_slice_count(rows, a, b, lk, lc, status, &s_full);
    u32x2 *sub = giants.sub_table(table, g, part);
    unsigned *gcount = giants.counts(g);
    for (unsigned i = tid; i < kJlSubSlots && !s_full; i += kJlGiantThreads) {
      const unsigned c = lc[i];
      if (c) {
        const unsigned gs = jl_giant_slot(sub, lk[i], true, &s_full);
        if (gs < kJlSubSlots) {
          // (scope and order spelled out: a device-scope read-modify-write performed at the memory side; what orders it
          //  against the tick below is that the lane holds its answer before the barrier)
          const unsigned before = __hip_atomic_fetch_add(&gcount[gs], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("" ::"v"(before));  // the lane waits for the atomic's answer: the add has been performed
        }
      }
    }
    // The last slice of a giant to get here turns the slot counters into positions.  Counter adds, tick and the reads
    // below are all memory-side atomics, every lane has its adds' answers before the barrier, the tick comes after it:
    // no fence.  (With __threadfence() on either side of the tick — a write-back and an invalidation of the XCD's whole
    // L2 each — this kernel took 2.1 ms for 500 giants of 2^16 rows: 9 slices each, every one stalling its XCD twice.)
    __syncthreads();
    // more distinct keys than a sub-table has slots (in this slice, or in the giant as a whole): the giant goes to the spill
    // path, listed once — the directory word doubles as the guard; what the other slices still count is never read
    if (tid == 0 && s_full && atomicCAS(spill.area + 2 * static_cast<size_t>(part), 0u, 0xFFFFFFFFu) == 0u) jl_spill_list(spill, part, status);
    if (tid == 0)
      s_last = __hip_atomic_fetch_add(&giants.done()[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == s_first[g + 1] - s_first[g] ? 1u : 0u;
    __syncthreads();
    if (s_last) {  // (uniform)
      constexpr unsigned kPer = kJlSubSlots / kJlGiantThreads;
      unsigned c[kPer], mine = 0;
#pragma unroll
      for (unsigned j = 0; j < kPer; ++j) {
        c[j] = __hip_atomic_fetch_add(&gcount[tid * kPer + j], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a read that no stale cache line can answer)
        mine += c[j];
      }
      const unsigned incl = wave_inclusive_scan(mine);
      if (lane == kWave - 1) s_wsum[wave] = incl;
      __syncthreads();
      unsigned run = static_cast<unsigned>(lo) + incl - mine;
      for (unsigned w = 0; w < wave; ++w) run += s_wsum[w];
      const unsigned cnt_esc = pos_bits < 32 ? (1u << (32 - pos_bits)) - 1u : 0u;
      unsigned *cursor = giants.cursors(g);
#pragma unroll
      for (unsigned j = 0; j < kPer; ++j) {
        const unsigned slot = tid * kPer + j;
        unsigned f = c[j] ? c[j] - 1 : 0u;
        f = f < cnt_esc ? f : cnt_esc;
        cursor[slot] = run;
        reinterpret_cast<unsigned *>(sub + slot)[1] = pos_bits < 32 ? (run | (f << pos_bits)) : run;
        run += c[j];
      }
      if (giants.with_tables && tid == kJlGiantThreads - 1) sub[kJlSubSlots] = u32x2{kEmptyKey, run};  // the scratch table's sentinel
    }
    __syncthreads();  // lk / lc / s_wsum / s_last are reused by the next slice
  }
}

__device__ __forceinline__ void jl_probe_sub(unsigned key, unsigned h, const u32x2 *__restrict__ sub, unsigned pos_bits,
                                             unsigned pos_mask, unsigned cnt_esc, unsigned *pos_out, unsigned *cnt_out,
                                             unsigned *marked);
// (radix join, match.spairs != nullptr: the same launch then runs the listed partitions' PROBE rows against their scratch
//  sub-tables, in slices of kJlSlice rows over all workgroups — they need the tables' position words, which
//  jl_giant_count finished, not the ids; one 8-byte gather per row, a hot key's slot one cache line for everybody;
//  results at the row's place in the probe side's partition order, like the fused kernel's)
__global__ __launch_bounds__(kJlGiantThreads) void jl_giant_ids_kernel(const u32x2 *__restrict__ rows,
                                                                        const unsigned long long *__restrict__ starts,
                                                                        u32x2 *table, JlGiants giants,
                                                                        unsigned *__restrict__ ids, unsigned *status,
                                                                        JlMatchArgs match, unsigned pos_bits, JlSpill spill) {
  __shared__ unsigned lk[kJlSubSlots], lc[kJlSubSlots];
  __shared__ unsigned s_first[kJlMaxGiantList], s_wsum[kJlGiantThreads / kWave], s_full;
  const unsigned tid = threadIdx.x;
  unsigned ng;
  const unsigned items = jl_giant_slices(giants, starts, s_first, s_wsum, &ng);
  for (unsigned item = blockIdx.x; item < items; item += gridDim.x) {
    const unsigned g = jl_giant_of(s_first, ng, item);
    const unsigned part = giants.part()[g];
    const size_t lo = starts[part], hi = starts[part + 1];
    const size_t a = lo + static_cast<size_t>(item - s_first[g]) * kJlSlice;
    const size_t b = a + kJlSlice < hi ? a + kJlSlice : hi;
    if (spill.dir()[part].x != 0u) continue;  // (uniform) a giant that went to the spill path: built at the end of this kernel
    jl_slice_count(rows, a, b, lk, lc, status, &s_full);
    u32x2 *sub = giants.sub_table(table, g, part);
    unsigned *cursor = giants.cursors(g);
    // the slice's share of every key's id range: lc[i] becomes the first position of the share
    for (unsigned i = tid; i < kJlSubSlots; i += kJlGiantThreads) {
      const unsigned c = lc[i];
      if (c) {
        const unsigned gs = jl_giant_slot(sub, lk[i], false, &s_full);
        lc[i] = gs < kJlSubSlots ? atomicAdd(&cursor[gs], c) : 0xFFFFFFFFu;  // (not there: only behind a full table, flagged)
      }
    }
    __syncthreads();
    for (size_t i0 = a; i0 < b; i0 += 4 * static_cast<size_t>(kJlGiantThreads)) {  // (uniform)
      u32x2 r[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t i = i0 + static_cast<size_t>(q) * kJlGiantThreads + tid;
        r[q] = i < b ? rows[i] : u32x2{kEmptyKey, 0u};
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned key = r[q].x;
        unsigned sl = kJlSubSlots;
        if (i0 + static_cast<size_t>(q) * kJlGiantThreads + tid < b && key != kEmptyKey) {
          unsigned t = jl_home_slot(fmix32(key));
          for (unsigned tries = 0; tries < kJlSubSlots; ++tries) {
            const unsigned k = lk[t];
            if (k == key) {
              sl = t;
              break;
            }
            if (k == kEmptyKey) break;
            t = jl_next_slot(t);
          }
        }
        const bool live = sl < kJlSubSlots;
        const unsigned pos = jl_take<true>(&lc[live ? sl : 0u], sl, live);
        if (live && pos - static_cast<unsigned>(lo) < static_cast<unsigned>(hi - lo)) ids[pos] = r[q].y;  // (outside: only behind a full table)
      }
    }
    __syncthreads();  // lk / lc are reused by the next slice
  }
  // the spilled partitions (listed by jl_build_kernel and jl_giant_count_kernel): one workgroup each
  {
    unsigned nsp = status[kJlHdrSpilled];
    nsp = nsp < spill.max_list ? nsp : spill.max_list;
    for (unsigned g = blockIdx.x; g < nsp; g += gridDim.x) {  // (uniform)
      const unsigned part = spill.list()[g];
      const size_t slo = match.spairs ? match.sstarts[part] : 0, shi = match.spairs ? match.sstarts[part + 1] : 0;
      jl_spill_partition(spill, rows, starts[part], starts[part + 1], part, ids, status, match, slo, shi,
                         table != nullptr ? table + static_cast<size_t>(part) * kJlSubSlots : nullptr, pos_bits);
    }
  }
  if (match.spairs == nullptr) return;
  __syncthreads();
  const unsigned pitems = jl_giant_slices(giants, match.sstarts, s_first, s_wsum, &ng);
  const unsigned pos_mask = pos_bits < 32 ? (1u << pos_bits) - 1u : 0xFFFFFFFFu;
  const unsigned cnt_esc = pos_bits < 32 ? (1u << (32 - pos_bits)) - 1u : 0u;
  for (unsigned item = blockIdx.x; item < pitems; item += gridDim.x) {
    const unsigned g = jl_giant_of(s_first, ng, item);
    const unsigned part = giants.part()[g];
    const size_t slo = match.sstarts[part], shi = match.sstarts[part + 1];
    const size_t a = slo + static_cast<size_t>(item - s_first[g]) * kJlSlice;
    const size_t b = a + kJlSlice < shi ? a + kJlSlice : shi;
    if (spill.dir()[part].x != 0u) continue;  // (uniform) spilled: its probe rows were answered by jl_spill_partition
    const u32x2 *sub = giants.scratch_table(g);
    for (size_t j = a + tid; j < b; j += kJlGiantThreads) {
      const u32x2 row = match.spairs[j];
      unsigned pos, cnt;
      unsigned marked = 0;
      jl_probe_sub(row.x, fmix32(row.x), sub, pos_bits, pos_mask, cnt_esc, &pos, &cnt, &marked);
      match.out_rid[j] = row.y;
      match.out_pos[j] = pos;
      match.out_cnt[j] = cnt;
    }
  }
}

// one probe row: slot {key, first position | count field} by linear probing inside the key's sub-table.
// *marked |= the count field of the EMPTY slot a miss ended on: zero in every sub-table the build kernels publish
// normally, non-zero in the sub-table of a partition whose keys live in the spill pool (jl_spill_partition) — the probe
// learns that from the slot it reads anyway (a flag word in the workspace header, read once per wave, made the same
// loop 7-11 % slower: 1557 -> 1670-1730 us at 2^26 rows on the same box, the branch never taken).
__device__ __forceinline__ void jl_probe_sub(unsigned key, unsigned h, const u32x2 *__restrict__ sub, unsigned pos_bits,
                                             unsigned pos_mask, unsigned cnt_esc, unsigned *pos_out, unsigned *cnt_out,
                                             unsigned *marked);
__device__ __forceinline__ void jl_probe_row(unsigned key, const u32x2 *__restrict__ table, unsigned parts, unsigned pos_bits,
                                             unsigned pos_mask, unsigned cnt_esc, unsigned *pos_out, unsigned *cnt_out,
                                             unsigned *marked) {
  const unsigned h = fmix32(key);
  jl_probe_sub(key, h, table + static_cast<size_t>((static_cast<unsigned long long>(h) * parts) >> 32) * kJlSubSlots, pos_bits,
               pos_mask, cnt_esc, pos_out, cnt_out, marked);
}
// the same inside a given sub-table (h = fmix32(key))
__device__ __forceinline__ void jl_probe_sub(unsigned key, unsigned h, const u32x2 *__restrict__ sub, unsigned pos_bits,
                                             unsigned pos_mask, unsigned cnt_esc, unsigned *pos_out, unsigned *cnt_out,
                                             unsigned *marked) {
  unsigned s = jl_home_slot(h), pos = 0, cnt = 0;
  for (unsigned tries = 0; tries < kJlSubSlots && key != kEmptyKey; ++tries) {  // the sentinel never matches
    const u32x2 e = sub[s];
    if (e.x == key) {
      pos = e.y & pos_mask;
      const unsigned field = pos_bits < 32 ? e.y >> pos_bits : 0u;
      // s + 1 may be the first slot of the next sub-table (or the sentinel): positions run on across them
      cnt = field < cnt_esc ? field + 1u : (sub[s + 1].y & pos_mask) - pos;
      break;
    }
    if (e.x == kEmptyKey) {
      *marked |= pos_bits < 32 ? e.y >> pos_bits : 0u;
      break;
    }
    s = jl_next_slot(s);
  }
  *pos_out = pos;
  *cnt_out = cnt;
}

__global__ __launch_bounds__(kJlThreads) void jl_probe_kernel(const unsigned *__restrict__ probe, size_t n,
                                                              const u32x2 *__restrict__ table, unsigned parts,
                                                              unsigned pos_bits, unsigned *__restrict__ out_pos,
                                                              unsigned *__restrict__ out_cnt, JlSpill spill) {
  // The probe is pure memory latency: one random 8-byte gather per row = the slot {key, first position | count
  // field}; the right-hand neighbour is read only when the field holds the escape value.  (The first layout read
  // slot and neighbour with one 16-byte load at 8-byte alignment: every eighth row crossed a 64-byte line, i.e. one
  // more memory request.  A variant with FOUR consecutive rows per lane — one 16-byte key load, four gathers in flight,
  // one 16-byte store per output column — measured SLOWER, 1852 vs 1434 us at 2^26 rows: the random-access path
  // saturates sooner; removed in round 4.)
  const unsigned pos_mask = pos_bits < 32 ? (1u << pos_bits) - 1u : 0xFFFFFFFFu;
  const unsigned cnt_esc = pos_bits < 32 ? (1u << (32 - pos_bits)) - 1u : 0u;
  const size_t stride = static_cast<size_t>(gridDim.x) * kJlThreads;
  unsigned marked = 0;  // some row of this thread ended on an empty slot of a spilled partition's sub-table
  for (size_t i = static_cast<size_t>(blockIdx.x) * kJlThreads + threadIdx.x; i < n; i += stride)
    jl_probe_row(probe[i], table, parts, pos_bits, pos_mask, cnt_esc, out_pos + i, out_cnt + i, &marked);
  // Spilled partitions (never on hashed keys): a sub-table whose EMPTY slots carry a non-zero count field belongs to a
  // partition whose keys are in the spill pool; a thread that met one goes over ITS rows again and answers those of
  // spilled partitions from the pool (a hit cannot happen in such a sub-table: it holds no keys)
  if (marked == 0u) return;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kJlThreads + threadIdx.x; i < n; i += stride) {
    const unsigned key = probe[i];
    const u32x2 d = spill.dir()[jl_pid(key, parts)];
    if (d.x == 0u) continue;
    unsigned p, c;
    jl_spill_lookup(spill, d.x - 1u, d.y, key, &p, &c);
    out_pos[i] = p;
    out_cnt[i] = c;
  }
}

// ---- unique-key payload join (dwarf 4b, join/join.cpp:60-131) over the same partitions -----------------------
// One workgroup per partition: ds_cmpst claims the key's slot, the claimer stores its payload next to it (a
// duplicate build key keeps the first claimer's payload: keys are unique by contract, join/join.cpp:13-16), the
// sub-table goes out as 8-byte slots {key, payload}.  Probe: one 8-byte gather per step.
// A partition of more than kJlSubSlots (unique) keys — keys constructed against the hash — goes to the spill pool as well
// (round 4): the workgroup builds an open-addressing {key -> payload} table for it there (jl_spill_partition's pool and
// directory; the position array holds the payloads), and publishes the sub-table as an empty one whose slots carry
// payload 0 instead of the sentinel: the mark the probe looks for on a miss.
__device__ __noinline__ void jl_uspill_partition(const JlSpill sp, const u32x2 *__restrict__ rows, size_t lo, size_t hi, unsigned part,
                                                 unsigned *status) {
  constexpr unsigned kT = kJlBuildThreads;
  __shared__ unsigned s_base;
  const unsigned tid = threadIdx.x;
  const unsigned cap = static_cast<unsigned>(jl_spill_cap(hi - lo));
  __syncthreads();
  if (tid == 0) s_base = atomicAdd(status + kJlHdrSpillPool, cap);
  __syncthreads();
  const unsigned base = s_base;
  if (static_cast<unsigned long long>(base) + cap > sp.pool) {  // (cannot happen: the pool holds every partition that can spill)
    if (tid == 0) atomicOr(status, DBHIP_DEV_TABLE_FULL);
    return;
  }
  unsigned *keys = sp.keys() + base, *vals = sp.pos() + base;
  for (unsigned i = tid; i < cap; i += kT) keys[i] = kEmptyKey;
  __threadfence();
  __syncthreads();
  for (size_t i = lo + tid; i < hi; i += kT) {
    const u32x2 row = rows[i];
    if (row.x == kEmptyKey) {
      atomicOr(status, DBHIP_DEV_KEY_RANGE);
      continue;
    }
    unsigned sl = jl_spill_home(row.x, cap);
    while (true) {  // (cap > rows: an empty slot exists)
      const unsigned k = atomicCAS(&keys[sl], kEmptyKey, row.x);
      if (k == kEmptyKey) {  // the claimer's payload stays (a duplicate build key keeps the first claimer's: join.cpp:13-16)
        __hip_atomic_store(&vals[sl], row.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      if (k == row.x) break;
      sl = sl + 1 == cap ? 0u : sl + 1;
    }
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) sp.dir()[part] = u32x2{base + 1u, cap};
}

__global__ __launch_bounds__(kJlBuildThreads) void jl_ubuild_kernel(const unsigned *__restrict__ pkeys,
                                                                    const unsigned *__restrict__ prids,
                                                                    const unsigned long long *__restrict__ starts,
                                                                    u32x2 *__restrict__ table, unsigned *status, JlSpill spill) {
  __shared__ unsigned lk[kJlSubSlots];
  __shared__ unsigned lv[kJlSubSlots];
  __shared__ unsigned s_full;
  const unsigned tid = threadIdx.x;
  const size_t part = blockIdx.x;
  const size_t lo = starts[part], hi = starts[part + 1];
  for (unsigned i = tid; i < kJlSubSlots; i += kJlBuildThreads) {
    lk[i] = kEmptyKey;
    lv[i] = kEmptyKey;
  }
  if (tid == 0) {
    s_full = 0;
    spill.dir()[part] = u32x2{0u, 0u};  // "not spilled"
  }
  __syncthreads();
  for (size_t i = lo + tid; i < hi; i += kJlBuildThreads) {
    const u32x2 row = jl_row(pkeys, prids, i);
    const unsigned key = row.x;
    unsigned s = jl_home_slot(fmix32(key));
    if (key == kEmptyKey) {  // the sentinel is not a key: flag it, drop the row
      atomicOr(status, DBHIP_DEV_KEY_RANGE);
      continue;
    }
    for (unsigned tries = 0;; ++tries) {
      const unsigned old = atomicCAS(&lk[s], kEmptyKey, key);
      if (old == kEmptyKey) {
        lv[s] = row.y;
        break;
      }
      if (old == key) break;
      s = jl_next_slot(s);
      if (tries + 1 >= kJlSubSlots) {  // more keys than slots: the partition goes to the spill pool
        s_full = 1u;
        break;
      }
    }
  }
  __syncthreads();
  u32x2 *dst = table + part * kJlSubSlots;
  if (s_full) {  // (uniform)
    jl_uspill_partition(spill, reinterpret_cast<const u32x2 *>(pkeys), lo, hi, static_cast<unsigned>(part), status);
    for (unsigned i = tid; i < kJlSubSlots; i += kJlBuildThreads) __builtin_nontemporal_store(u32x2{kEmptyKey, 0u}, dst + i);
    return;
  }
  for (unsigned i = tid; i < kJlSubSlots; i += kJlBuildThreads) __builtin_nontemporal_store(u32x2{lk[i], lv[i]}, dst + i);
}

__global__ __launch_bounds__(kJlThreads) void jl_uprobe_kernel(const unsigned *__restrict__ pkeys,
                                                               const unsigned *__restrict__ pvals, size_t n,
                                                               const u32x2 *__restrict__ table, unsigned parts,
                                                               unsigned *__restrict__ out_key,
                                                               unsigned *__restrict__ out_bval,
                                                               unsigned *__restrict__ out_pval, JlSpill spill) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kJlThreads;
  unsigned marked = 0;  // a miss of this thread ended on an empty slot of a spilled partition's sub-table (payload 0, not the sentinel)
  for (size_t i = static_cast<size_t>(blockIdx.x) * kJlThreads + threadIdx.x; i < n; i += stride) {
    const unsigned key = pkeys[i];
    const unsigned h = fmix32(key);
    const u32x2 *sub = table + static_cast<size_t>((static_cast<unsigned long long>(h) * parts) >> 32) * kJlSubSlots;
    unsigned s = jl_home_slot(h), bval = kEmptyKey;
    bool found = false;
    for (unsigned tries = 0; tries < kJlSubSlots && key != kEmptyKey; ++tries) {
      const u32x2 e = sub[s];
      if (e.x == key) {
        found = true;
        bval = e.y;
        break;
      }
      if (e.x == kEmptyKey) {
        marked |= e.y != kEmptyKey ? 1u : 0u;
        break;
      }
      s = jl_next_slot(s);
    }
    // join.cpp:41-43, :96-101: sentinels where the probe row has no partner
    out_key[i] = found ? key : kEmptyKey;
    out_bval[i] = bval;
    out_pval[i] = found ? pvals[i] : kEmptyKey;
  }
  if (marked == 0u) return;
  // a thread that met a spilled partition goes over ITS rows again and answers those of spilled partitions from the pool
  for (size_t i = static_cast<size_t>(blockIdx.x) * kJlThreads + threadIdx.x; i < n; i += stride) {
    const unsigned key = pkeys[i];
    if (key == kEmptyKey) continue;
    const u32x2 d = spill.dir()[jl_pid(key, parts)];
    if (d.x == 0u) continue;
    const unsigned base = d.x - 1u, cap = d.y;
    unsigned sl = jl_spill_home(key, cap);
    for (unsigned tries = 0; tries < cap; ++tries) {
      const unsigned k = spill.keys()[base + sl];
      if (k == key) {
        out_key[i] = key;
        out_bval[i] = spill.pos()[base + sl];
        out_pval[i] = pvals[i];
        break;
      }
      if (k == kEmptyKey) break;
      sl = sl + 1 == cap ? 0u : sl + 1;
    }
  }
}

// grid of the level-0 scatter: a multiple of 8 (one slice of workgroups per XCD)
inline unsigned jl_scatter0_grid(size_t tiles, size_t cap) {
  static const int forced = [] { const char *e = getenv("DBHIP_JL_SC0_WGS"); return e ? atoi(e) : 0; }();  // experiment knob: workgroups per CU
  if (forced >= 1 && forced <= 32) cap = static_cast<size_t>(forced) * 256;  // (knob: per CU of a 256-CU chip)
  size_t g = tiles < cap ? tiles : cap;
  g = (g + 7) / 8 * 8;
  return static_cast<unsigned>(g ? g : 8);
}

inline unsigned jl_grid(size_t items, const DeviceInfo &dev, int per_cu) {
  const size_t want = (items + kJlThreads - 1) / kJlThreads;
  const size_t cap = static_cast<size_t>(dev.cus) * per_cu;
  return static_cast<unsigned>(want < cap ? (want ? want : 1) : cap);
}


// ---- tile shapes of the two scatter levels ---------------------------------------------------------------------------
// 0: 512 threads x 8 rows = 4096-row tiles (36 KiB + 16 B per bucket of LDS) — the shape every size up to 2^27 rows was
//    tuned on;  1: 1024 x 8 = 8192 rows;  2 (level 0 only): 1024 x 16 = 16384 rows (128 KiB of LDS: one workgroup per CU).
// A tile of T rows into nb buckets writes runs of T / nb rows and takes one returning global atomic per bucket: with the
// 586 x 1024 buckets of a 2^30-row side a 4096-row tile writes 56- and 32-byte runs and one atomic per 7 / 4 rows (level 0
// at 2.8 TB/s, level 1 at 2.6, against 4.0 / 3.9 at 2^26 rows with 293 x 128 buckets).  DBHIP_JL_T0 / DBHIP_JL_T1 force a
// shape (experiments; T1 = 2 reads as 1).
struct JlShape {
  int t0, t1;
};
inline unsigned jl_shape_rows(int id) { return id == 0 ? 4096u : id == 1 ? 8192u : 16384u; }
inline unsigned jl_shape_threads(int id) { return id == 0 ? 512u : 1024u; }
inline int jl_env_shape(const char *name) {
  const char *e = getenv(name);
  return e && e[0] >= '0' && e[0] <= '2' && !e[1] ? e[0] - '0' : -1;
}
inline JlShape jl_shape_for(size_t n, unsigned k1, unsigned k2) {
  static const int f0 = jl_env_shape("DBHIP_JL_T0"), f1 = jl_env_shape("DBHIP_JL_T1");
  (void)n;
  // Measured (radix join, us, t0/t1; same box per size).  Level 1 one tile per workgroup, before it became persistent:
  // 2^26 rows (293 x 128 buckets) 0/0 1838, 1/1 1862, 2/2 2039; 2^27 (293 x 256) 0/0 3406, 1/1 3502, 2/2 3805; 2^28 (586 x 256)
  // 0/0 7776, 1/0 7521, 2/0 7261, 2/1 7425, 2/2 8089; 2^29 (586 x 512) 0/0 16214, 2/0 15225, 1/1 15196, 2/1 14567, 2/2 15836;
  // 2^30 (586 x 1024) 0/0 35256, 2/0 33994, 0/2 33265, 1/2 32852, 2/2 32558.  Persistent level 1: 2^26 x/0 1671, x/1 1703;
  // 2^27 x/0 3144, x/1 3157; 2^28 2/0 6829, 2/1 6769, 0/1 7181, 1/1 7007; 2^30 2/0 29565, 2/1 26830, 1/1 27381, 0/1 27871.
  JlShape sh{0, 0};
  if (k1 >= 512) sh.t0 = 2;
  if (k2 >= 512) sh.t1 = 1;
  if (f0 >= 0) sh.t0 = f0;
  if (f1 >= 0) sh.t1 = f1 > 1 ? 1 : f1;
  return sh;
}

// resident workgroups per CU of `kernel` with `lds` bytes of dynamic LDS: asked of the runtime once per (kernel, lds) and
// host thread, not on every launch
inline int jl_resident_blocks(const void *kernel, int threads, size_t lds) {
  thread_local const void *last_kernel = nullptr;
  thread_local size_t last_lds = 0;
  thread_local int last_blocks = 0;
  if (kernel != last_kernel || lds != last_lds || last_blocks < 1) {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, threads, lds) != hipSuccess || blocks < 1) {
      (void)hipGetLastError();
      blocks = 1;
    }
    last_kernel = kernel;
    last_lds = lds;
    last_blocks = blocks;
  }
  return last_blocks;
}

template <bool RANK, bool RIDS, int THREADS, int KPT, bool DIGITS = false>
hipError_t jl_launch_scatter0_shape(const DeviceInfo &dev, hipStream_t s, const unsigned *keys, const unsigned *row_ids,
                                    unsigned long long first_row, size_t n, unsigned parts, unsigned k2_shift, unsigned k1,
                                    unsigned long long *cursors, unsigned *out_keys, unsigned *out_rids) {
  constexpr unsigned kTile = THREADS * KPT;
  const size_t lds = jl_scatter_lds_bytes(k1, kTile, THREADS);
  auto kernel = jl_scatter0_kernel<RANK, RIDS, THREADS, KPT, DIGITS>;
  if (lds > 48 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  const size_t tiles = (n + kTile - 1) / kTile;
  // persistent grid of the workgroups that are resident (round 4: the 4096-row shape ran with eight per CU where its
  // registers allowed two; held to 80 VGPRs — amdgpu_waves_per_eu(6) — three are, and a grid of exactly those measured
  // 1.5-2 % of the radix join at 2^26 rows: 1759-1777 -> 1726-1740 us on the same box)
  const size_t per_cu = static_cast<size_t>(jl_resident_blocks(reinterpret_cast<const void *>(kernel), THREADS, lds));
  hipLaunchKernelGGL(kernel, dim3(jl_scatter0_grid(tiles, static_cast<size_t>(dev.cus) * per_cu)), dim3(THREADS), lds, s, keys,
                     row_ids, first_row, n, parts, k2_shift, k1, cursors, out_keys, out_rids);
  return hipSuccess;
}
template <bool RANK>
hipError_t jl_launch_scatter0(int shape, const DeviceInfo &dev, hipStream_t s, const unsigned *keys, const unsigned *row_ids,
                              unsigned long long first_row, size_t n, unsigned parts, unsigned k2_shift, unsigned k1,
                              unsigned long long *cursors, unsigned *out_keys, unsigned *out_rids, bool digits = false) {
  // digits: pairs into out_keys AND every row's level-1 bucket as a 16-bit column behind out_rids
  if (digits) {
    if (RANK) return hipErrorInvalidValue;
#define JL_SC0D(RIDS, T, K) \
  jl_launch_scatter0_shape<false, RIDS, T, K, true>(dev, s, keys, row_ids, first_row, n, parts, k2_shift, k1, cursors, out_keys, out_rids)
    if (row_ids) return shape == 0 ? JL_SC0D(true, 512, 8) : shape == 1 ? JL_SC0D(true, 1024, 8) : JL_SC0D(true, 1024, 16);
    return shape == 0 ? JL_SC0D(false, 512, 8) : shape == 1 ? JL_SC0D(false, 1024, 8) : JL_SC0D(false, 1024, 16);
#undef JL_SC0D
  }
#define JL_SC0(RIDS, T, K) \
  jl_launch_scatter0_shape<RANK, RIDS, T, K>(dev, s, keys, row_ids, first_row, n, parts, k2_shift, k1, cursors, out_keys, out_rids)
  if (row_ids) {
    if (RANK) return hipErrorInvalidValue;  // (the rank-level partition numbers its rows itself)
    return shape == 0 ? JL_SC0(!RANK, 512, 8) : shape == 1 ? JL_SC0(!RANK, 1024, 8) : JL_SC0(!RANK, 1024, 16);
  }
  return shape == 0 ? JL_SC0(false, 512, 8) : shape == 1 ? JL_SC0(false, 1024, 8) : JL_SC0(false, 1024, 16);
#undef JL_SC0
}

template <int THREADS, int KPT>
hipError_t jl_launch_scatter1_shape(hipStream_t s, size_t n, const u32x2 *rows, const unsigned long long *starts0,
                                    const unsigned long long *tstarts0, unsigned parts, unsigned k1, unsigned k2,
                                    unsigned long long *cursors1, u32x2 *out, const DeviceInfo &dev) {
  constexpr unsigned kTile = THREADS * KPT;
  const size_t lds = jl_scatter_lds_bytes(k2, kTile, THREADS);
  auto kernel = jl_scatter1p_kernel<THREADS, KPT>;
  if (lds > 48 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  const int blocks = jl_resident_blocks(reinterpret_cast<const void *>(kernel), THREADS, lds);
  const size_t vtiles = ((n + kTile - 1) / kTile + k1 + 7) / 8 * 8;  // every bucket's last tile may be ragged
  size_t grid = static_cast<size_t>(dev.cus) * blocks / 8 * 8;       // the resident workgroups, a whole number per XCD
  if (grid < 8) grid = 8;
  if (grid > vtiles) grid = vtiles;
  hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(grid)), dim3(THREADS), lds, s, rows, starts0, tstarts0, parts, k1, k2,
                     cursors1, out);
  return hipSuccess;
}
inline hipError_t jl_launch_scatter1(int shape, hipStream_t s, size_t n, const u32x2 *rows, const unsigned long long *starts0,
                                     const unsigned long long *tstarts0, unsigned parts, unsigned k1, unsigned k2,
                                     unsigned long long *cursors1, u32x2 *out, const DeviceInfo &dev) {
  return shape == 0 ? jl_launch_scatter1_shape<512, 8>(s, n, rows, starts0, tstarts0, parts, k1, k2, cursors1, out, dev)
                    : jl_launch_scatter1_shape<1024, 8>(s, n, rows, starts0, tstarts0, parts, k1, k2, cursors1, out, dev);
}

}  // namespace

namespace {
struct JlPartitioned {
  const unsigned *keys, *rids;             // partition-major rows: (key, row id) pairs behind `keys`, rids == nullptr
  const unsigned long long *starts;        // parts + 1 offsets
  u32x2 *table;
  unsigned *status;
};

// The one or two scatter levels shared by all joins, for a column of n rows and a partition geometry that may come
// from ANOTHER column (the radix join partitions the probe side by the build side's geometry): level-0 output in
// rows_a, level-1 output (k2 > 1) in rows_b, offsets in `meta` (jl_meta_bytes(k1, parts) bytes).
int jl_partition_side(const unsigned *keys, const unsigned *row_ids, size_t n, unsigned parts, unsigned k1, unsigned k2,
                      unsigned log2_k2, u32x2 *rows_a, u32x2 *rows_b, unsigned long long *meta, size_t meta_bytes,
                      hipStream_t s, const DeviceInfo &dev, const unsigned **out_pairs,
                      const unsigned long long **out_starts) {
  // meta: counts0g[G*k1] | cursors0g[G*k1] | starts0[k1+1] | tile_starts0[k1+1] | counts1[K] | starts1[K+1] | cursors1[K]
  unsigned long long *counts0 = meta;
  unsigned long long *cursors0 = counts0 + static_cast<size_t>(kJlGroups) * k1;
  unsigned long long *starts0 = cursors0 + static_cast<size_t>(kJlGroups) * k1;
  unsigned long long *tstarts0 = starts0 + k1 + 1;
  unsigned long long *counts1 = tstarts0 + k1 + 1;
  unsigned long long *starts1 = counts1 + parts;
  unsigned long long *cursors1 = starts1 + parts + 1;

  const hipError_t e = fill_async(meta, 0, meta_bytes, s);
  if (e != hipSuccess) return static_cast<int>(e);
  // scratch of the fused histogram: 256 rows of `parts` counters, in the level-1 output region while it is still unused
  // (it holds 8n bytes; 1 KiB per partition is enough whenever a partition averages >= 128 rows)
  unsigned *fused_scratch = (k2 > 1 && n * 8 >= static_cast<size_t>(kJlGroups) * kJlFusedWgPerGroup * parts * sizeof(unsigned))
                                ? reinterpret_cast<unsigned *>(rows_b) : nullptr;

  // both levels write (key, row id) as ONE 8-byte element: a run of r rows is 8r contiguous bytes instead of two
  // runs of 4r (the scatters are bound by partially written lines: level 1 went 330 -> 254 us at 2^26 rows when it
  // switched, level 0 followed once the level-1 histogram read pairs instead of a keys-only column)
  const unsigned k2_shift = log2_k2;
  const JlShape shape = jl_shape_for(n, k1, k2);
  const size_t group_rows = jl_group_rows(n, jl_shape_rows(shape.t0));
  // two levels and at most 32768 partitions: both histograms from one read of the keys (wgcnt scratch: the level-1
  // output region, written only later by the level-1 scatter)
  // (8192..32768 partitions = 2^24..2^26 rows: level below it the two plain histograms are as fast, 2^22 rows: 77 vs 80 us)
  const bool fused = k2 > 1 && parts >= 8192 && parts <= kJlFusedMaxParts && fused_scratch != nullptr;
  // 32768 < parts <= 81920 (2^27-row shards and a quarter more): the same with two 16-bit counters per LDS word (jl_hist_fused16_kernel)
#ifdef DBHIP_JL_NO_FUSED16  // A/B knob: the two plain histograms for these sizes, as in round 2
  const bool fused16 = false;
#else
  const bool fused16 = !fused && k2 > 1 && kJlFusedMaxParts != 0 && parts > kJlFusedMaxParts && parts <= kJlFused16MaxParts &&
                       fused_scratch != nullptr;
#endif
  if (fused16) {
    const size_t lds = static_cast<size_t>(parts / 2) * sizeof(unsigned);
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(jl_hist_fused16_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (ea != hipSuccess) return static_cast<int>(ea);
    hipLaunchKernelGGL(jl_hist_fused16_kernel, dim3(kJlGroups * kJlFusedWgPerGroup), dim3(kJlFusedThreads), lds, s, keys, n,
                       group_rows, parts, log2_k2, k1, fused_scratch, counts0, counts1);
    const unsigned red_grid = (parts / 2 + 255) / 256 + (kJlGroups * kJlFusedWgPerGroup * k1 + 3) / 4;
    hipLaunchKernelGGL(jl_hist_reduce16_kernel, dim3(red_grid), dim3(256), 0, s, fused_scratch, parts, k1, k2, counts0, counts1);
  } else if (fused) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(jl_hist_fused_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(parts * sizeof(unsigned)));
    if (ea != hipSuccess) return static_cast<int>(ea);
    hipLaunchKernelGGL(jl_hist_fused_kernel, dim3(kJlGroups * kJlFusedWgPerGroup), dim3(kJlFusedThreads),
                       parts * sizeof(unsigned), s, keys, n, group_rows, parts, fused_scratch);
    const unsigned red_grid = (parts + 255) / 256 + (kJlGroups * kJlFusedWgPerGroup * k1 + 3) / 4;
    hipLaunchKernelGGL(jl_hist_reduce_kernel, dim3(red_grid), dim3(256), 0, s, fused_scratch, parts, k1, k2, counts0, counts1);
  } else {
    hipLaunchKernelGGL(jl_hist0_kernel<false>, dim3(kJlGroups * kJlHistWgPerGroup), dim3(kJlThreads),
                       k1 * sizeof(unsigned), s, keys, n, group_rows, parts, k2_shift, k1, counts0);
  }
  hipLaunchKernelGGL(jl_offsets0_kernel, dim3(1), dim3(1024), 0, s, counts0, k1, jl_shape_rows(shape.t1), cursors0, starts0,
                     tstarts0, static_cast<unsigned long long *>(nullptr));
  // A level-1 histogram of its own (more partitions than the fused ones count: sides of more than 1.47e8 rows): level 0
  // writes every row's level-1 bucket as a 16-bit column into the level-1 output region — unused until the level-1
  // scatter writes it, like the fused histograms' scratch — and the histogram reads those 2 bytes per row instead of
  // the 8-byte pairs: 2^30 x 2^30 26.7-26.8 -> 25.5-25.6 ms, same box (DBHIP_JL_DIGITS=0: the pairs, for A/B runs).
  // Not below 8192 partitions, where the plain histograms run as well: those sides are a few hundred us as they are.
  static const bool digits_on = [] { const char *v = getenv("DBHIP_JL_DIGITS"); return !(v && v[0] == '0'); }();
  const bool digits = digits_on && k2 > 1 && k2 <= 65536 && !fused && !fused16 && parts > kJlFused16MaxParts;
  {
    const hipError_t es = jl_launch_scatter0<false>(shape.t0, dev, s, keys, row_ids, 0ull, n, parts, k2_shift, k1, cursors0,
                                                    reinterpret_cast<unsigned *>(rows_a),
                                                    digits ? reinterpret_cast<unsigned *>(rows_b) : static_cast<unsigned *>(nullptr), digits);
    if (es != hipSuccess) return static_cast<int>(es);
  }
  *out_pairs = reinterpret_cast<const unsigned *>(rows_a);
  *out_starts = starts0;
  if (k2 > 1) {
    if (digits)
      hipLaunchKernelGGL(jl_hist1d_kernel, dim3(k1 * kJlHist1WgPerBucket), dim3(kJlThreads), k2 * sizeof(unsigned), s,
                         reinterpret_cast<const unsigned short *>(rows_b), starts0, k2, counts1);
    else if (!fused && !fused16)
      hipLaunchKernelGGL(jl_hist1_kernel, dim3(k1 * kJlHist1WgPerBucket), dim3(kJlThreads), k2 * sizeof(unsigned), s,
                         rows_a, starts0, parts, k2, counts1);
    hipLaunchKernelGGL(jl_offsets1_kernel, dim3(k1), dim3(kJlThreads), 0, s, counts1, starts0, k1, k2, starts1,
                       cursors1);
    // (one tile per workgroup; a persistent grid with the next tile's rows prefetched — what helps the level-0
    //  scatter — measured the same here: a workgroup that ends after its stores never waits for them.  Round 3: a
    //  precomputed {bucket, tile} map in place of the workgroup's binary search over tile_starts — eight dependent
    //  loads in front of its row loads — measured the same as well (partition of 2^26 rows 610 vs 615 us), and so did
    //  the tile shapes 512x16 / 512x4 / 1024x4 once more (723 / 661 / 699 us against 610).)
    const hipError_t e1 = jl_launch_scatter1(shape.t1, s, n, rows_a, starts0, tstarts0, parts, k1, k2, cursors1, rows_b, dev);
    if (e1 != hipSuccess) return static_cast<int>(e1);
    *out_pairs = reinterpret_cast<const unsigned *>(rows_b);
    *out_starts = starts1;
  }
  return launch_status();
}

// the build side of the one-to-many / unique-key joins, laid out by jl_layout(n): fills `out`
int jl_partition_rows(const unsigned *build_keys, const unsigned *row_ids, size_t n, void *workspace, hipStream_t s,
                      const DeviceInfo &dev, const JlLayout &L, JlPartitioned *out) {
  char *base = static_cast<char *>(workspace);
  const hipError_t e = fill_async(base, 0, kWsHeader, s);
  if (e != hipSuccess) return static_cast<int>(e);
  out->rids = nullptr;
  out->table = reinterpret_cast<u32x2 *>(base + L.table_off);
  out->status = reinterpret_cast<unsigned *>(base);
  return jl_partition_side(build_keys, row_ids, n, L.parts, L.k1, L.k2, L.log2_k2,
                           reinterpret_cast<u32x2 *>(base + L.keys_a_off), reinterpret_cast<u32x2 *>(base + L.keys_b_off),
                           reinterpret_cast<unsigned long long *>(base + L.meta_off), L.meta_bytes, s, dev, &out->keys,
                           &out->starts);
}

size_t jl_build_lds_bytes() { return 2 * static_cast<size_t>(kJlSubSlots) * sizeof(unsigned); }
bool jl_no_giants() {  // DBHIP_JL_NO_GIANTS=1: every partition through the per-partition kernel, whatever its size (A/B timing)
  static const bool off = [] { const char *v = getenv("DBHIP_JL_NO_GIANTS"); return v && v[0] == '1'; }();
  return off;
}
// Resident workgroups per CU of a persistent kernel, as the runtime computes it from the kernel's REGISTERS as well as
// its LDS (round 4: the build kernels were launched with four 512-thread workgroups per CU — what their 24 KiB of LDS
// allow — while their 73-79 VGPRs allow six waves per SIMD, i.e. three: a quarter of the statically dealt partitions
// belonged to workgroups that only started when the first ones had finished).  `env`: experiment knob, workgroups per CU.
template <class Kernel>
unsigned jl_resident_per_cu(Kernel kernel, int threads, size_t lds, const char *env, unsigned fallback) {
  const char *e = getenv(env);
  const int forced = e ? atoi(e) : 0;
  if (forced >= 1 && forced <= 32) return static_cast<unsigned>(forced);
  int blocks = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, threads, lds) != hipSuccess || blocks < 1) {
    (void)hipGetLastError();
    return fallback;
  }
  return static_cast<unsigned>(blocks);
}
template <bool kMatch>
unsigned jl_build_grid(unsigned parts, const DeviceInfo &dev) {
  static const unsigned per_cu = jl_resident_per_cu(jl_build_kernel<kMatch, false>, kJlBuildThreads, jl_build_lds_bytes(),
                                                    kMatch ? "DBHIP_JL_MATCH_WGS" : "DBHIP_JL_BUILD_WGS", 3u);
  const size_t cap = static_cast<size_t>(dev.cus) * per_cu;
  return static_cast<unsigned>(parts < cap ? parts : cap);
}
}  // namespace

// ---- host side (called from join.hip's entry points) ---------------------------------------------------
int join_lds_build(const unsigned *build_keys, const unsigned *row_ids, size_t n, unsigned *ids, void *workspace,
                   hipStream_t s, const DeviceInfo &dev) {
  const JlLayout L = jl_layout(n);
  JlPartitioned p;
  const int rc = jl_partition_rows(build_keys, row_ids, n, workspace, s, dev, L, &p);
  if (rc != 0) return rc;
  const size_t build_lds = jl_build_lds_bytes();
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(jl_build_kernel<false, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(build_lds));
  if (e != hipSuccess) return static_cast<int>(e);
  // giant partitions (join_common.hpp): listed by the build kernel, counted and filled by two launches of their own
  JlGiants giants{nullptr, 0u, ~0ull, ~0ull, 0u};
  if (L.max_giants && !jl_no_giants()) {
    giants.base = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + L.giant_off);
    giants.max = L.max_giants;
    giants.rows = jl_giant_rows(n);
    const hipError_t eg = fill_async(giants.base, 0, jl_giant_header_bytes(L.max_giants), s);
    if (eg != hipSuccess) return static_cast<int>(eg);
  }
  // spilled partitions (more distinct keys than a sub-table has slots): from the size where the giants' launches exist
  // they are listed and built at the end of jl_giant_ids_kernel; below it the build kernel's workgroup does it at once
  const JlSpill spill = jl_spill_of(static_cast<char *>(workspace) + L.spill_off, L.parts, n);
  const JlMatchArgs no_match{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (L.max_giants) {
    hipLaunchKernelGGL((jl_build_kernel<false, false>), dim3(jl_build_grid<false>(L.parts, dev)), dim3(kJlBuildThreads), build_lds, s,
                       p.keys, p.rids, p.starts, p.table, L.parts, static_cast<unsigned>(n), jl_pos_bits(n), ids, p.status,
                       no_match, giants, spill.area);
    const unsigned grid = static_cast<unsigned>(dev.cus) * (2048 / kJlGiantThreads);
    if (giants.max)
      hipLaunchKernelGGL(jl_giant_count_kernel, dim3(grid), dim3(kJlGiantThreads), 0, s, reinterpret_cast<const u32x2 *>(p.keys),
                         p.starts, p.table, jl_pos_bits(n), giants, p.status, spill);
    hipLaunchKernelGGL(jl_giant_ids_kernel, dim3(grid), dim3(kJlGiantThreads), 0, s, reinterpret_cast<const u32x2 *>(p.keys),
                       p.starts, p.table, giants, ids, p.status, no_match, jl_pos_bits(n), spill);
  } else {
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(jl_build_kernel<false, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(build_lds));
    if (e2 != hipSuccess) return static_cast<int>(e2);
    hipLaunchKernelGGL((jl_build_kernel<false, true>), dim3(jl_build_grid<false>(L.parts, dev)), dim3(kJlBuildThreads), build_lds, s,
                       p.keys, p.rids, p.starts, p.table, L.parts, static_cast<unsigned>(n), jl_pos_bits(n), ids, p.status,
                       no_match, giants, spill.area);
  }
  return launch_status();
}

// ---- radix join: both sides partitioned with the build side's geometry, one fused build + probe launch -----------
// workspace: header | build side: pairs a, pairs b, meta | probe side: pairs a, pairs b, meta
namespace {
struct JrLayout {
  unsigned parts, k1, k2, log2_k2, max_giants;
  size_t meta_bytes, b_a, b_b, b_meta, p_a, p_b, p_meta, giant_off, spill_off, total;
};
JrLayout jr_layout(size_t n_build, size_t n_probe) {
  const JlLayout G = jl_layout(n_build, kJrRowsPerPart);
  JrLayout L;
  L.parts = G.parts; L.k1 = G.k1; L.k2 = G.k2; L.log2_k2 = G.log2_k2;
  L.meta_bytes = G.meta_bytes;
  const size_t cb = align_up((n_build ? n_build : 1) * 8, kWsAlign), cp = align_up((n_probe ? n_probe : 1) * 8, kWsAlign);
  const size_t mb = align_up(L.meta_bytes, kWsAlign);
  L.b_a = kWsHeader;
  L.b_b = L.b_a + cb;
  L.b_meta = L.b_b + (L.k2 > 1 ? cb : 0);
  L.p_a = L.b_meta + mb;
  L.p_b = L.p_a + cp;
  L.p_meta = L.p_b + (L.k2 > 1 ? cp : 0);
  L.max_giants = jr_max_giants(n_build, n_probe);
  L.giant_off = L.p_meta + mb;
  L.spill_off = align_up(L.giant_off + jl_giant_bytes(L.max_giants, true), kWsAlign);
  L.total = L.spill_off + jl_spill_bytes(L.parts, n_build);
  return L;
}
// where a partitioned side ended up is a pure function of the sizes: no state is kept between the calls
void jr_side(const JrLayout &L, char *base, bool probe, const unsigned **pairs, const unsigned long long **starts) {
  const size_t a = probe ? L.p_a : L.b_a, b = probe ? L.p_b : L.b_b, m = probe ? L.p_meta : L.b_meta;
  unsigned long long *meta = reinterpret_cast<unsigned long long *>(base + m);
  unsigned long long *starts0 = meta + 2 * static_cast<size_t>(kJlGroups) * L.k1;
  unsigned long long *starts1 = starts0 + 2 * (static_cast<size_t>(L.k1) + 1) + L.parts;
  *pairs = reinterpret_cast<const unsigned *>(base + (L.k2 > 1 ? b : a));
  *starts = L.k2 > 1 ? starts1 : starts0;
}
}  // namespace

size_t join_radix_workspace_bytes(size_t n_build, size_t n_probe) { return jr_layout(n_build, n_probe).total; }

int join_radix_partition(int probe_side, const unsigned *keys, const unsigned *row_ids, size_t n, size_t n_build,
                         size_t n_probe, void *workspace, hipStream_t s, const DeviceInfo &dev) {
  const JrLayout L = jr_layout(n_build, n_probe);
  char *base = static_cast<char *>(workspace);
  if (!probe_side) {  // the build side's call opens a join: it clears the status word
    const hipError_t e = fill_async(base, 0, kWsHeader, s);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  const unsigned *pairs;
  const unsigned long long *starts;
  return jl_partition_side(keys, row_ids, n, L.parts, L.k1, L.k2, L.log2_k2,
                           reinterpret_cast<u32x2 *>(base + (probe_side ? L.p_a : L.b_a)),
                           reinterpret_cast<u32x2 *>(base + (probe_side ? L.p_b : L.b_b)),
                           reinterpret_cast<unsigned long long *>(base + (probe_side ? L.p_meta : L.b_meta)), L.meta_bytes, s,
                           dev, &pairs, &starts);
}

int join_radix_match(size_t n_build, size_t n_probe, unsigned *ids, unsigned *out_rid, unsigned *out_pos, unsigned *out_cnt,
                     void *workspace, hipStream_t s, const DeviceInfo &dev) {
  const JrLayout L = jr_layout(n_build, n_probe);
  char *base = static_cast<char *>(workspace);
  const unsigned *bp, *pp;
  const unsigned long long *bs, *ps;
  jr_side(L, base, false, &bp, &bs);
  jr_side(L, base, true, &pp, &ps);
  const size_t build_lds = jl_build_lds_bytes();
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(jl_build_kernel<true, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(build_lds));
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(jl_build_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(build_lds));
  if (e != hipSuccess) return static_cast<int>(e);
  // partitions with a giant build OR probe side (join_common.hpp): left out by the fused kernel, listed, then built into
  // scratch sub-tables (jl_giant_count / jl_giant_ids) and probed (second half of jl_giant_ids) by all workgroups together
  JlGiants giants{nullptr, 0u, ~0ull, ~0ull, 0u};
  if (L.max_giants && !jl_no_giants()) {
    giants.base = reinterpret_cast<unsigned *>(base + L.giant_off);
    giants.max = L.max_giants;
    giants.rows = jl_giant_rows(n_build);
    giants.probe_rows = jl_giant_rows(n_probe);
    giants.with_tables = 1u;
    const hipError_t eg = fill_async(giants.base, 0, jl_giant_header_bytes(L.max_giants), s);
    if (eg != hipSuccess) return static_cast<int>(eg);
  }
  const JlMatchArgs match{reinterpret_cast<const u32x2 *>(pp), ps, out_rid, out_pos, out_cnt};
  unsigned *status = reinterpret_cast<unsigned *>(base);
  const JlSpill spill = jl_spill_of(base + L.spill_off, L.parts, n_build);
  if (L.max_giants) {
    hipLaunchKernelGGL((jl_build_kernel<true, false>), dim3(jl_build_grid<true>(L.parts, dev)), dim3(kJlBuildThreads), build_lds, s, bp,
                       static_cast<const unsigned *>(nullptr), bs, static_cast<u32x2 *>(nullptr), L.parts,
                       static_cast<unsigned>(n_build), jl_pos_bits(n_build), ids, status, match, giants, spill.area);
    const unsigned grid = static_cast<unsigned>(dev.cus) * (2048 / kJlGiantThreads);
    if (giants.max)
      hipLaunchKernelGGL(jl_giant_count_kernel, dim3(grid), dim3(kJlGiantThreads), 0, s, reinterpret_cast<const u32x2 *>(bp), bs,
                         static_cast<u32x2 *>(nullptr), jl_pos_bits(n_build), giants, status, spill);
    hipLaunchKernelGGL(jl_giant_ids_kernel, dim3(grid), dim3(kJlGiantThreads), 0, s, reinterpret_cast<const u32x2 *>(bp), bs,
                       static_cast<u32x2 *>(nullptr), giants, ids, status, match, jl_pos_bits(n_build), spill);
  } else {
    hipLaunchKernelGGL((jl_build_kernel<true, true>), dim3(jl_build_grid<true>(L.parts, dev)), dim3(kJlBuildThreads), build_lds, s, bp,
                       static_cast<const unsigned *>(nullptr), bs, static_cast<u32x2 *>(nullptr), L.parts,
                       static_cast<unsigned>(n_build), jl_pos_bits(n_build), ids, status, match, giants, spill.area);
  }
  return launch_status();
}

// unique-key payload join: `build_vals` ride through the partition in the row-id column
int ujoin_lds_build(const unsigned *build_keys, const unsigned *build_vals, size_t n, void *workspace, hipStream_t s,
                    const DeviceInfo &dev) {
  const JlLayout L = jl_layout(n);
  JlPartitioned p;
  const int rc = jl_partition_rows(build_keys, build_vals, n, workspace, s, dev, L, &p);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(jl_ubuild_kernel, dim3(L.parts), dim3(kJlBuildThreads), 0, s, p.keys, p.rids, p.starts, p.table,
                     p.status, jl_spill_of(static_cast<char *>(workspace) + L.spill_off, L.parts, n));
  return launch_status();
}

int ujoin_lds_probe(const unsigned *probe_keys, const unsigned *probe_vals, size_t n_probe, const void *workspace,
                    size_t n_build, unsigned *out_key, unsigned *out_bval, unsigned *out_pval, hipStream_t s,
                    const DeviceInfo &dev) {
  const JlLayout L = jl_layout(n_build);
  const u32x2 *table = reinterpret_cast<const u32x2 *>(static_cast<const char *>(workspace) + L.table_off);
  static const int uprobe_wgs = static_cast<int>(jl_resident_per_cu(jl_uprobe_kernel, kJlThreads, 0, "DBHIP_JL_UPROBE_WGS", 4u));
  hipLaunchKernelGGL(jl_uprobe_kernel, dim3(jl_grid(n_probe, dev, uprobe_wgs)), dim3(kJlThreads), 0, s, probe_keys, probe_vals,
                     n_probe, table, L.parts, out_key, out_bval, out_pval,
                     jl_spill_of(const_cast<char *>(static_cast<const char *>(workspace)) + L.spill_off, L.parts, n_build));
  return launch_status();
}

int join_lds_probe(const unsigned *probe_keys, size_t n_probe, const void *workspace, size_t n_build,
                   unsigned *out_pos, unsigned *out_cnt, hipStream_t s, const DeviceInfo &dev) {
  const JlLayout L = jl_layout(n_build);
  const u32x2 *table = reinterpret_cast<const u32x2 *>(static_cast<const char *>(workspace) + L.table_off);
  // Exactly the workgroups that are resident (four of 512 threads per CU), every thread striding over its rows: with
  // eight per CU — two rounds of workgroups — the kernel's time depended on how the second round happened to fill in
  // (round 4, same box, 2^26 rows: 2 per CU 1767-1838 us, 4: 1553-1572, 6: 1692-1713, 8: 1566 for one build of this
  // kernel and 1708 for another whose loop was the same instructions; 16: 1566 / 1625).  DBHIP_JL_PROBE_WGS: experiments.
  static const int wgs_per_cu = static_cast<int>(jl_resident_per_cu(jl_probe_kernel, kJlThreads, 0, "DBHIP_JL_PROBE_WGS", 4u));
  char *ws = const_cast<char *>(static_cast<const char *>(workspace));
  hipLaunchKernelGGL(jl_probe_kernel, dim3(jl_grid(n_probe, dev, wgs_per_cu)), dim3(kJlThreads), 0, s, probe_keys, n_probe,
                     table, L.parts, jl_pos_bits(n_build), out_pos, out_cnt, jl_spill_of(ws + L.spill_off, L.parts, n_build));
  return launch_status();
}

// how many of `keys` do NOT belong to bucket `rank` of `parts` under the rank hash (validator of the exchange's routing)
__global__ __launch_bounds__(kJlThreads) void jl_route_check_kernel(const unsigned *__restrict__ keys, size_t n,
                                                                    unsigned parts, unsigned rank,
                                                                    unsigned long long *result) {
  __shared__ unsigned s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  const size_t stride = static_cast<size_t>(gridDim.x) * kJlThreads;
  unsigned bad = 0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kJlThreads + threadIdx.x; i < n; i += stride)
    bad += jl_rank_of(keys[i], parts) != rank;
  bad = wave_reduce_add(bad);
  if ((threadIdx.x & (kWave - 1)) == kWave - 1 && bad) atomicAdd(&s_bad, bad);
  __syncthreads();
  if (threadIdx.x == 0 && s_bad) atomicAdd(result, static_cast<unsigned long long>(s_bad));
}

int jl_route_check(const unsigned *keys, size_t n, unsigned parts, unsigned rank, unsigned long long *result,
                   hipStream_t s, const DeviceInfo &dev) {
  const hipError_t e = fill_async(result, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n == 0) return DBHIP_OK;
  hipLaunchKernelGGL(jl_route_check_kernel, dim3(jl_grid(n, dev, 8)), dim3(kJlThreads), 0, s, keys, n, parts, rank, result);
  return launch_status();
}

// ---- stand-alone level-0 partition (multi-GPU join: bucket = destination rank) ------------------------
// 2^27 rows into 8 buckets: histogram 158 us + scatter 566 us (into 2 buckets: 285 + 700 us — the LDS atomics of a
// wave land on very few addresses).  Counting and ranking by ballot in wave-uniform registers instead (16 unrolled
// bucket tests per key) was measured at 324 + 794 us and dropped.
size_t jl_partition_workspace_bytes(unsigned parts) {
  return align_up(kWsHeader + sizeof(unsigned long long) * ((2 * static_cast<size_t>(kJlGroups) + 2) * parts + 2),
                  kWsAlign);
}

int jl_partition(const unsigned *keys, size_t n, unsigned long long first_row, unsigned parts, unsigned *out_keys,
                 unsigned *out_rids, unsigned long long *out_counts, void *workspace, hipStream_t s,
                 const DeviceInfo &dev) {
  char *base = static_cast<char *>(workspace);
  unsigned long long *meta = reinterpret_cast<unsigned long long *>(base + kWsHeader);
  unsigned long long *counts0 = meta;
  unsigned long long *cursors0 = counts0 + static_cast<size_t>(kJlGroups) * parts;
  unsigned long long *starts0 = cursors0 + static_cast<size_t>(kJlGroups) * parts;
  unsigned long long *tstarts0 = starts0 + parts + 1;
  hipError_t e = fill_async(base, 0, jl_partition_workspace_bytes(parts), s);
  if (e != hipSuccess) return static_cast<int>(e);
  hipLaunchKernelGGL(jl_hist0_kernel<true>, dim3(kJlGroups * kJlHistWgPerGroup), dim3(kJlThreads),
                     parts * sizeof(unsigned), s, keys, n, jl_group_rows(n, kJlTile), parts, 0u, parts, counts0);
  hipLaunchKernelGGL(jl_offsets0_kernel, dim3(1), dim3(1024), 0, s, counts0, parts, static_cast<unsigned>(kJlTile), cursors0,
                     starts0, tstarts0, out_counts);
  if (n) {
    e = jl_launch_scatter0<true>(0, dev, s, keys, static_cast<const unsigned *>(nullptr), first_row, n, parts, 0u, parts,
                                 cursors0, out_keys, out_rids);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  return launch_status();
}

}  // namespace dbhip
