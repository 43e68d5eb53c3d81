// radix.hip — dwarf 2: LSD radix sort of 32-bit keys for gfx950.
//
// Replaces oneDPL's std::sort(device_policy) behind Radix/RadixCuda (dpl_wrapper.hpp:35-39 <-
// sort/radix.cpp:34); result identical to std::sort (sort/radix.cpp:8-12).
//
// Digit width BITS = 8 (tuned) or 4 (the configuration named in BASELINE.json).  Work is cut into CHUNKS of consecutive
// 8192-key tiles; a pass needs, for every chunk, the number of keys per digit (-> where the chunk's keys of that digit
// go), then scatters chunk by chunk with no communication between workgroups:
//   rs_chunk_scatter  one workgroup per chunk, tile by tile: each wave ranks its 1024 keys stably (one returning LDS
//                     atomic per key, or BITS ballots per key — CDNA has no match instruction), the tile is re-ordered
//                     by digit through LDS and written out in digit order, so consecutive lanes hit consecutive
//                     addresses; the chunk's running per-digit offsets live in registers of the digit-owner threads.
// What differs between the digit widths is where the per-chunk counts come from:
//   8-bit   rs_upfront (one read of the keys: digit 0's counts per chunk + which key bits vary at all) and, for every
//           later pass, rs_chunk_hist (a second read of the pass's input) -> rs_chunk_scan (one workgroup per digit:
//           exclusive scan of its row, row total) -> scatter (adds the digit base: exclusive scan of the 256 totals).
//   4-bit   READ ONCE PER PASS (round 3): the scatter of pass k counts, while it writes a key to its final position,
//           the key's NEXT digit d' into an LDS table indexed by (destination chunk, d, d') — a chunk's keys of digit d
//           are one contiguous run of the output, which lies in at most two of the next pass's (position-defined)
//           chunks, so the table has 2 x 16 x 16 counters — and adds the table to the next pass's count matrix with
//           512 coalesced global atomics per workgroup.  Per pass: rs_scan4 (ONE workgroup: offsets of every chunk and
//           digit, clears the matrix the scatter is about to fill) -> scatter.  With 8-bit digits the same table would
//           have 2 x 256 x 256 counters per chunk and about one key per counter: no aggregation, one global atomic per
//           key — the 8-bit sort keeps its second read.
// Passes whose digit bits do not vary over the input (OR of the keys & OR of their complements, taken by rs_upfront)
// are skipped by every kernel on that device-side word: no host round trip, no plan kernel; each kernel derives the
// ping-pong parity of its pass from the same word.  rs_finalize copies tmp -> keys after an odd number of passes.
// (A single-pass Onesweep with decoupled look-back was measured in round 1: 106 us per 8-bit pass at 2^24 keys,
// dominated by look-back waits between tiles in flight; same finding as in scan.hip.)
//
// Bytes: 8-bit 4N + 8N + 3 * 12N; 4-bit 4N + P * 8N, P <= 8; at 2^24 keys both ping-pong buffers (128 MiB) live in
// the 256 MiB Infinity Cache.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "dbhip_common.hpp"

namespace dbhip {
namespace {

#ifndef DBHIP_RS_THREADS
#define DBHIP_RS_THREADS 512
#endif
constexpr int kRsThreads = DBHIP_RS_THREADS;
constexpr int kRsWaves = kRsThreads / kWave;
// Tile shape of the scatter kernel (compile-time knobs for experiments).  Measured at 2^24 full-range keys, 8-bit:
// 16 keys/lane at 4 waves/SIMD (128 VGPRs) 356 us; 8 keys/lane at 6 waves/SIMD (80 VGPRs) 379 us; 8 at 8 (spills)
// 443 us; 16 at 5 (spills) 553 us — more resident waves with narrower tiles do not pay.  Workgroup width (tile =
// threads x keys/lane): 256 threads 356 us, 512 threads (8192-key tiles, 32-key digit runs) 336 us, 1024 threads 426 us.
// After the match-mask rewrite (kernel VALU-bound, 297 us) the shapes 512x16, 512x8 and 256x16 are within 3 %.
#ifndef DBHIP_RS_KPT
#define DBHIP_RS_KPT 16
#endif
#ifndef DBHIP_RS_WPE
#define DBHIP_RS_WPE 4
#endif
constexpr int kRsKpt = DBHIP_RS_KPT;             // keys per lane per tile
constexpr int kRsWaveKeys = kWave * kRsKpt;      // 1024 contiguous keys per wave
constexpr int kRsTile = kRsWaveKeys * kRsWaves;  // 8192 keys
constexpr int kRsMaxRadix = 256;
#ifndef DBHIP_RS_CHECK_MOD
#define DBHIP_RS_CHECK_MOD 5  // the order check looks at the 512-key bands k = 0, 5, 10, 15 of a tile (experiments: 1 = all, 99 = first)
#endif
#ifndef DBHIP_RS_CHUNKS
#define DBHIP_RS_CHUNKS 2048
#endif
constexpr size_t kRsTargetChunks = DBHIP_RS_CHUNKS;  // chunks per pass (>= 8 per CU for balance)
constexpr size_t kRsFusedScanChunks = 32;        // 8-bit: up to this many chunks the scatter sums its own prefix (2^16 keys: 89 -> 80 us; at 128 chunks it costs 17 us)
constexpr size_t kRs4MaxChunks = kRsTargetChunks / 2;  // 4-bit: at most this many chunks, one thread of rs_scan4 each
static_assert(kRs4MaxChunks <= 1024, "rs_scan4 is one workgroup with one thread per chunk");

struct RsHeader {
  unsigned status, pad0;
  // low word: OR of the keys, high word: OR of their complements (both start at 0 with the cleared header).  A key bit
  // takes both values somewhere in the input iff it is set in both words; a pass is skipped iff none of its digit's
  // bits does.
  unsigned long long or_nor;
  unsigned pad[60];
};
static_assert(sizeof(RsHeader) == kWsHeader, "workspace header size");
__device__ __forceinline__ unsigned rs_varying(const RsHeader *hdr) {
  const unsigned long long w = hdr->or_nor;
  return static_cast<unsigned>(w) & static_cast<unsigned>(w >> 32);
}

// what every kernel derives from the header word for ITS pass
template <int BITS>
__device__ __forceinline__ bool rs_varies(unsigned varying, int pass) {
  return ((varying >> (pass * BITS)) & ((1u << BITS) - 1u)) != 0u;
}
template <int BITS>
__device__ __forceinline__ unsigned rs_parity(unsigned varying, int pass) {  // passes executed before `pass`, mod 2
  unsigned par = 0;
  for (int q = 0; q < pass; ++q) par ^= rs_varies<BITS>(varying, q) ? 1u : 0u;
  return par;
}
template <int BITS>
__device__ __forceinline__ int rs_next_pass(unsigned varying, int pass) {  // next executed pass after `pass`, or -1
  for (int q = pass + 1; q < 32 / BITS; ++q)
    if (rs_varies<BITS>(varying, q)) return q;
  return -1;
}

// workspace, 8-bit: header | totals[4][256] | counts[256][chunks] (turned into offsets in place by rs_chunk_scan)
//            4-bit: header | cnt[2][chunks][16] (the matrix of the current pass and the one the scatter fills) | off[chunks][16]
// (Measured and dropped for 4-bit digits in round 1: per-pass sums of the chunk counts over groups of 64 chunks, added by
// the histogram workgroups with 16 global atomics each, so that the scatter could find its offsets without a scan
// kernel — the 2048 workgroups hammer the same 32 cache lines of sums: rs_chunk_hist 12.5 -> 28 us.)
constexpr size_t kRsTotalsOff = kWsHeader;
constexpr size_t kRsCountsOff = kRsTotalsOff + sizeof(unsigned) * 4 * kRsMaxRadix;  // 8-bit
constexpr size_t kRs4CntOff = kWsHeader;                                            // 4-bit

struct RsGeometry {
  size_t tiles, tiles_per_chunk, chunks;
};
inline RsGeometry rs_geometry(size_t n, int bits) {
  RsGeometry g;
  // 4-bit digits: half as many, twice as long chunks (2^24 keys: 378 -> 363 us; 8-bit digits: no difference)
  const size_t target = bits == 4 ? kRs4MaxChunks : kRsTargetChunks;
  g.tiles = (n + kRsTile - 1) / kRsTile;
  g.tiles_per_chunk = (g.tiles + target - 1) / target;
  if (g.tiles_per_chunk == 0) g.tiles_per_chunk = 1;
  g.chunks = (g.tiles + g.tiles_per_chunk - 1) / g.tiles_per_chunk;
  if (g.chunks == 0) g.chunks = 1;
  return g;
}

// lanes of the wave whose digit equals mine: BITS ballots (no match instruction on CDNA).  Per bit and 32-lane
// half: sel = bit ? ballot : ~ballot = ~(ballot ^ (bit ? ~0 : 0)), one v_xnor_b32 on the sign-extended bit
// (v_bfe_i32) folded with the running mask into one v_bitop3_b32 — 4 vector instructions per bit (v_bfe_i32, v_cmp,
// 2 x v_bitop3) where the obvious `bit ? bal : ~bal` on 64-bit values compiled to 9 (the scatter kernel is
// VALU-bound, so this is its critical path).
struct LaneMask {
  unsigned lo, hi;
};
template <int BITS>
__device__ __forceinline__ LaneMask match_digit(unsigned d) {
  LaneMask m{~0u, ~0u};
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    const unsigned nb = static_cast<unsigned>(__builtin_amdgcn_sbfe(d, b, 1));  // bit b of d as 0 / 0xFFFFFFFF
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(static_cast<int>(nb) < 0);
    // m & ~(bal ^ nb) in one three-input boolean op per half (v_bitop3_b32, truth table 0x90 over (m, bal, nb))
    m.lo = __builtin_amdgcn_bitop3_b32(m.lo, static_cast<unsigned>(bal), nb, 0x90);
    m.hi = __builtin_amdgcn_bitop3_b32(m.hi, static_cast<unsigned>(bal >> 32), nb, 0x90);
  }
  return m;
}
// number of set mask bits below my lane / in the whole mask
__device__ __forceinline__ unsigned lanes_before(LaneMask m) {
  return __builtin_amdgcn_mbcnt_hi(m.hi, __builtin_amdgcn_mbcnt_lo(m.lo, 0u));
}
__device__ __forceinline__ unsigned lanes_in(LaneMask m) { return __builtin_popcount(m.lo) + __builtin_popcount(m.hi); }

// consecutive chunks per histogram workgroup: 4 with 8-bit digits (2^24 keys: rs_chunk_hist 15.9 -> 13.0 us, the sort
// 217 -> 208 us; 2: 216, 8: 222), 1 with 4-bit digits (16 counts per chunk are one 64-byte row: nothing to gain)
template <int BITS>
constexpr int rs_hist_cpw() { return BITS == 8 ? 4 : 1; }

// Byte histogram of the tiles [t0, t1) of the input — consecutive chunks of `tiles_per_chunk` tiles each, the first one
// counted into s_bins[0], the next into s_bins[1], ... (LDS, cleared by the caller): one ds_add per key on the byte that
// holds the pass's digit.  The next tile's four 16-byte loads per lane are in flight while the current tile is counted
// (one load in flight per lane made the up-front read latency-bound: 28 us for 64 MiB that come from HBM, against 13 us
// for the per-pass histograms whose input lies in the Infinity Cache).  A byte that is the same in the whole wave (upper
// bytes of small keys: the reference's [1,10000] data) would serialise 64 same-address ds_add: one lane adds the lot.
__device__ __forceinline__ void rs_count4(const u32x4 v, int byte_shift, unsigned xor_mask, unsigned *bins, unsigned &my_or,
                                          unsigned &my_nor) {
  my_or |= v.x | v.y | v.z | v.w;
  my_nor |= ~(v.x & v.y & v.z & v.w);
  const unsigned b0 = ((v.x ^ xor_mask) >> byte_shift) & 255u, b1 = ((v.y ^ xor_mask) >> byte_shift) & 255u,
                 b2 = ((v.z ^ xor_mask) >> byte_shift) & 255u, b3 = ((v.w ^ xor_mask) >> byte_shift) & 255u;
  const unsigned first = __builtin_amdgcn_readfirstlane(b0);
  const bool same = b0 == first && b1 == first && b2 == first && b3 == first;
  if (__ballot(same) == __ballot(true)) {
    if (lane_id() == 0) atomicAdd(&bins[first], 4u * kWave);  // (only reached with all 64 lanes active: full tiles)
  } else {
    atomicAdd(&bins[b0], 1u);
    atomicAdd(&bins[b1], 1u);
    atomicAdd(&bins[b2], 1u);
    atomicAdd(&bins[b3], 1u);
  }
}
__device__ __forceinline__ void rs_count_tiles(const unsigned *__restrict__ src, size_t n, size_t t0, size_t t1,
                                               size_t tiles_per_chunk, int byte_shift, unsigned xor_mask,
                                               unsigned (*s_bins)[256], unsigned &my_or, unsigned &my_nor) {
  constexpr int kVec = kRsTile / 4 / kRsThreads;  // 16-byte loads per lane and tile
  const size_t full_end = n / kRsTile < t1 ? n / kRsTile : t1;  // tiles below this index are complete
  u32x4 cur[kVec], nxt[kVec];
  if (t0 < full_end) {
    const u32x4 *k4 = reinterpret_cast<const u32x4 *>(src + t0 * kRsTile);  // tile starts are 32 KiB multiples: aligned
#pragma unroll
    for (int j = 0; j < kVec; ++j) cur[j] = k4[threadIdx.x + j * kRsThreads];
  }
#pragma unroll 1
  for (size_t t = t0; t < full_end; ++t) {
    if (t + 1 < full_end) {
      const u32x4 *k4 = reinterpret_cast<const u32x4 *>(src + (t + 1) * kRsTile);
#pragma unroll
      for (int j = 0; j < kVec; ++j) nxt[j] = k4[threadIdx.x + j * kRsThreads];
    }
    unsigned *bins = s_bins[(t - t0) / tiles_per_chunk];
#pragma unroll
    for (int j = 0; j < kVec; ++j) rs_count4(cur[j], byte_shift, xor_mask, bins, my_or, my_nor);
#pragma unroll
    for (int j = 0; j < kVec; ++j) cur[j] = nxt[j];
  }
  if (full_end < t1) {  // the input's ragged last tile
    unsigned *bins = s_bins[(full_end - t0) / tiles_per_chunk];
    for (size_t i = full_end * kRsTile + threadIdx.x; i < n; i += kRsThreads) {
      const unsigned k = src[i];
      my_or |= k;
      my_nor |= ~k;
      atomicAdd(&bins[((k ^ xor_mask) >> byte_shift) & 255u], 1u);
    }
  }
}

// Store the digit counts of the kCpw chunks a histogram workgroup has counted (s_bins[cc][256] byte bins).
//   8-bit: counts[d][chunk], the kCpw counts of a digit side by side: one 16-byte store per digit when the row allows
//          it (with one chunk per workgroup every count was a 4-byte store into a line of its own: 512 K partial-line
//          writes per pass at 2^24 keys, 16 MiB written back for a 2 MiB matrix);
//   4-bit: cnt[chunk][16]: the nibble's counts are sums of the byte bins over the other nibble.
template <int BITS>
__device__ __forceinline__ void rs_store_chunk_counts(unsigned (*s_bins)[256], unsigned *counts, size_t chunk0,
                                                      size_t num_chunks, int high_nibble) {
  constexpr int kCpw = rs_hist_cpw<BITS>();
  if (BITS == 8) {
    const bool vec = kCpw == 4 && chunk0 + 4 <= num_chunks && (num_chunks & 3) == 0;
    for (int d = threadIdx.x; d < 256; d += kRsThreads) {
      unsigned *row = counts + static_cast<size_t>(d) * num_chunks + chunk0;
      if (vec) {
        *reinterpret_cast<u32x4 *>(row) = u32x4{s_bins[0][d], s_bins[1 % kCpw][d], s_bins[2 % kCpw][d], s_bins[3 % kCpw][d]};
      } else {
        for (int cc = 0; cc < kCpw && chunk0 + cc < num_chunks; ++cc) row[cc] = s_bins[cc][d];
      }
    }
  } else if (threadIdx.x < 16) {
    unsigned c = 0;
#pragma unroll
    for (int o = 0; o < 16; ++o) c += s_bins[0][high_nibble ? threadIdx.x * 16 + o : o * 16 + threadIdx.x];
    counts[chunk0 * 16 + threadIdx.x] = c;
  }
}

// ---- one read of the keys up front: digit 0's counts of every chunk, and which key bits vary at all ---------------
// (Rounds 1-2 took the digit totals of ALL passes here — four ds_add per key on byte bins, 28.7 us at 2^24 keys against
// ~13 us for one; the totals now come out of the per-pass scan and every skip decision out of the two OR words.)
template <int BITS>
__global__ __launch_bounds__(kRsThreads) void rs_upfront_kernel(const unsigned *__restrict__ keys, size_t n, unsigned xor_mask,
                                                                RsHeader *hdr, unsigned *__restrict__ counts0,
                                                                size_t tiles_per_chunk, size_t num_chunks) {
  // The two OR words reach the header through ONE 64-bit atomic per workgroup, issued after the workgroup's FIRST group of
  // chunks (for most inputs every varying bit has shown up by then) so that it drains while the workgroup streams on, and
  // once more at the end only if later keys added a bit: same-address global atomics are served one per ~11 ns, and two
  // per workgroup at the end of 1024 workgroups made this kernel 34 us instead of 14.
  constexpr int kCpw = rs_hist_cpw<BITS>();
  __shared__ unsigned s_bins[kCpw][256];
  __shared__ unsigned s_or, s_nor;
  if (threadIdx.x == 0) s_or = s_nor = 0;
  unsigned my_or = 0, my_nor = 0;
  unsigned long long published = 0;  // thread 0: what this workgroup has already added to the header
  const size_t total_tiles = (n + kRsTile - 1) / kRsTile;
  const size_t groups = (num_chunks + kCpw - 1) / kCpw;
  for (size_t group = blockIdx.x; group < groups; group += gridDim.x) {
    const bool first = group == blockIdx.x;
    for (int i = threadIdx.x; i < kCpw * 256; i += kRsThreads) (&s_bins[0][0])[i] = 0;
    __syncthreads();
    const size_t t0 = group * kCpw * tiles_per_chunk;
    const size_t t1 = t0 + kCpw * tiles_per_chunk < total_tiles ? t0 + kCpw * tiles_per_chunk : total_tiles;
    rs_count_tiles(keys, n, t0, t1, tiles_per_chunk, 0, xor_mask, s_bins, my_or, my_nor);
    if (first) {
      if (my_or & ~s_or) atomicOr(&s_or, my_or);
      if (my_nor & ~s_nor) atomicOr(&s_nor, my_nor);
    }
    __syncthreads();
    rs_store_chunk_counts<BITS>(s_bins, counts0, group * kCpw, num_chunks, 0);
    if (first && threadIdx.x == 0) {
      published = static_cast<unsigned long long>(s_or) | (static_cast<unsigned long long>(s_nor) << 32);
      if (published) atomicOr(&hdr->or_nor, published);
    }
    __syncthreads();
  }
  if (my_or & ~s_or) atomicOr(&s_or, my_or);
  if (my_nor & ~s_nor) atomicOr(&s_nor, my_nor);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long all = static_cast<unsigned long long>(s_or) | (static_cast<unsigned long long>(s_nor) << 32);
    if (all & ~published) atomicOr(&hdr->or_nor, all);
  }
}

// ---- digit counts of every chunk by a second read of the pass's input -------------------------------------------
// 8-bit: every pass after the first.  4-bit: launched once after rs_upfront and does something only when digit 0 turned
// out constant (FIRST = true: it counts the first digit that varies — the rest of the 4-bit pipeline gets its counts
// from the previous pass's scatter).
template <int BITS, bool FIRST>
__global__ __launch_bounds__(kRsThreads) void rs_chunk_hist_kernel(const unsigned *keys, const unsigned *tmp, size_t n, int pass,
                                                                   unsigned xor_mask, const RsHeader *hdr, unsigned *counts,
                                                                   size_t tiles_per_chunk, size_t num_chunks) {
  constexpr int kCpw = rs_hist_cpw<BITS>();
  __shared__ unsigned s_bins[kCpw][256];
  const unsigned varying = rs_varying(hdr);
  if (FIRST) {
    if (rs_varies<BITS>(varying, 0)) return;  // the common case: rs_upfront counted the right digit
    pass = rs_next_pass<BITS>(varying, 0);
    if (pass < 0) return;  // all keys equal: nothing to sort
  } else if (!rs_varies<BITS>(varying, pass)) {
    return;
  }
  const unsigned *__restrict__ src = rs_parity<BITS>(varying, pass) ? tmp : keys;
  const int byte_shift = (pass * BITS) & ~7;
  const size_t total_tiles = (n + kRsTile - 1) / kRsTile;
  const size_t groups = (num_chunks + kCpw - 1) / kCpw;
  unsigned unused_or = 0, unused_nor = 0;
  for (size_t group = blockIdx.x; group < groups; group += gridDim.x) {
    for (int i = threadIdx.x; i < kCpw * 256; i += kRsThreads) (&s_bins[0][0])[i] = 0;
    __syncthreads();
    const size_t t0 = group * kCpw * tiles_per_chunk;
    const size_t t1 = t0 + kCpw * tiles_per_chunk < total_tiles ? t0 + kCpw * tiles_per_chunk : total_tiles;
    rs_count_tiles(src, n, t0, t1, tiles_per_chunk, byte_shift, xor_mask, s_bins, unused_or, unused_nor);
    __syncthreads();
    rs_store_chunk_counts<BITS>(s_bins, counts, group * kCpw, num_chunks, BITS == 4 ? (pass & 1) : 0);
    __syncthreads();
  }
}

// ---- 8-bit, per pass: counts[d][*] -> start of digit d in every chunk relative to the digit's base; row total ----------
__global__ __launch_bounds__(kRsThreads) void rs_chunk_scan_kernel(int pass, const RsHeader *hdr, unsigned *__restrict__ totals,
                                                                   unsigned *counts, size_t num_chunks) {
  // one workgroup per digit, ONE sweep: every thread takes `per` consecutive chunk counts (<= 8: at most 4096 chunks),
  // wave scan of the thread sums, wave sums through LDS (a loop of 512-chunk rounds with three barriers each took 5 us
  // for 2048 chunks, most of it barrier and LDS latency)
  constexpr unsigned kMaxPer = 8;
  static_assert(kRsTargetChunks <= static_cast<size_t>(kMaxPer) * kRsThreads, "chunks per scan workgroup");
  __shared__ unsigned s_wsum[kRsWaves];
  if (!rs_varies<8>(rs_varying(hdr), pass)) return;
  const unsigned d = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  unsigned *row = counts + static_cast<size_t>(d) * num_chunks;
  const unsigned per = static_cast<unsigned>((num_chunks + kRsThreads - 1) / kRsThreads);
  const size_t first = static_cast<size_t>(tid) * per;
  unsigned v[kMaxPer], mine = 0;
#pragma unroll
  for (unsigned j = 0; j < kMaxPer; ++j) {
    v[j] = (j < per && first + j < num_chunks) ? row[first + j] : 0u;
    mine += v[j];
  }
  const unsigned incl = wave_inclusive_scan(mine);
  if (lane == kWave - 1) s_wsum[wave] = incl;
  __syncthreads();
  unsigned run = incl - mine, all = 0;
  for (unsigned w = 0; w < kRsWaves; ++w) {
    const unsigned s = s_wsum[w];
    if (w < wave) run += s;
    all += s;
  }
  if (tid == 0) totals[pass * kRsMaxRadix + d] = all;
#pragma unroll
  for (unsigned j = 0; j < kMaxPer; ++j) {
    if (j < per && first + j < num_chunks) row[first + j] = run;
    run += v[j];
  }
}

// ---- 4-bit, per pass: ONE workgroup turns cnt[chunk][16] into off[chunk][16] (global start of the chunk's keys of every
// digit) and clears the matrix the pass's scatter accumulates the next pass's counts into -----------------------------
__global__ __launch_bounds__(1024) void rs_scan4_kernel(int pass, const RsHeader *hdr, unsigned *cnt2, unsigned *__restrict__ off,
                                                        size_t num_chunks) {
  constexpr int kW = 1024 / kWave;
  __shared__ unsigned s_w[kW][17];  // per-wave digit sums, then exclusive over the waves
  __shared__ unsigned s_tot[16], s_base[16];
  static_assert(kW == 16, "the cross-wave scan below is one 16-lane DPP row per digit");
  // (requesting the rows of BOTH matrices together with the header word that says which one is current, to save a
  //  dependent round trip, was measured: 7.8 -> 9.0 us.  About 4.5 us of this kernel's time is the write-back of what
  //  the scatter before it left dirty in the L2s: a kernel that only reads the header and returns takes 4.8 us here.)
  const unsigned varying = rs_varying(hdr);
  if (!rs_varies<4>(varying, pass)) return;
  const unsigned par = rs_parity<4>(varying, pass);
  const u32x4 *__restrict__ cur = reinterpret_cast<const u32x4 *>(cnt2 + static_cast<size_t>(par) * num_chunks * 16);
  u32x4 *next = reinterpret_cast<u32x4 *>(cnt2 + static_cast<size_t>(par ^ 1u) * num_chunks * 16);
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  unsigned v[16], incl[16];
  const bool live = tid < num_chunks;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32x4 x = live ? cur[static_cast<size_t>(tid) * 4 + q] : u32x4{0, 0, 0, 0};
    v[q * 4 + 0] = x.x, v[q * 4 + 1] = x.y, v[q * 4 + 2] = x.z, v[q * 4 + 3] = x.w;
    if (live) next[static_cast<size_t>(tid) * 4 + q] = u32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    incl[d] = wave_inclusive_scan(v[d]);
    if (lane == kWave - 1) s_w[wave][d] = incl[d];
  }
  __syncthreads();
  if (tid < 256) {  // lane = (digit, wave) with the wave in the low four bits: one 16-lane DPP row per digit
    const unsigned d = tid >> 4, w = tid & 15u;
    const unsigned t = s_w[w][d];
    unsigned x = t;
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, false);  // row_shr:1 (stays inside the row)
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, false);
    s_w[w][d] = x - t;
    if (w == 15u) s_tot[d] = x;
  }
  __syncthreads();
  if (tid < 16) {  // exclusive scan of the 16 digit totals
    const unsigned run = s_tot[tid];
    unsigned x = run;
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, false);
    s_base[tid] = x - run;
  }
  __syncthreads();
  if (live) {
    u32x4 *o = reinterpret_cast<u32x4 *>(off) + static_cast<size_t>(tid) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned r[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = q * 4 + e;
        r[e] = s_base[d] + s_w[wave][d] + incl[d] - v[d];
      }
      o[q] = u32x4{r[0], r[1], r[2], r[3]};
    }
  }
}

// ---- ranking by LDS atomics: the property it rests on, and how it is watched -------------------------------------------
// rank = atomicAdd(&count[wave][digit], 1) replaces the BITS ballots of match_digit (8-bit digits: ~55 of the scatter's
// ~100 vector instructions per 64 keys; the kernel is VALU-bound) by one ds_add_rtn_u32.  An LSD sort needs a STABLE
// rank.  Between instructions the LDS keeps a wave's operations in issue order; INSIDE one instruction the rank is
// stable iff lanes that hit the same counter get their return values in ascending lane order.  gfx950 does that (the
// LDS serialises the lanes of a conflicting access lowest lane first) but the ISA manual does not promise it.  Three
// guards: (1) the atomic ranking is the default only on an allow-listed architecture (gfx950); (2) EVERY tile of EVERY
// call checks the invariant the ranking exists for — after k stable passes a tile re-ordered by digit k is sorted by
// its low (k+1) digits, so thread p compares its key with its left neighbour's under that mask while it writes the
// tile out; a violation sets DBHIP_DEV_RANK_ORDER in the workspace's status word (it also catches a damaged earlier
// pass); (3) dbhip_radix_sort_prepare() runs the kernel below — every lane compares what the atomic returned with the
// count the ballots predict, over dense, sparse, skewed and partially masked digit patterns on every CU — and pins
// the ranking to what it saw.  The sort entry points themselves never synchronise (DBHIP_RS_RANK=ballot|atomic overrides).
__global__ __launch_bounds__(kRsThreads) void rs_rank_selftest_kernel(unsigned *mismatches) {
  __shared__ unsigned s_cnt[kRsWaves][256];
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  for (int i = tid; i < kRsWaves * 256; i += kRsThreads) (&s_cnt[0][0])[i] = 0;
  __syncthreads();
  unsigned bad = 0;
  for (unsigned it = 0; it < 128; ++it) {
    unsigned h = (blockIdx.x * 977u + wave * 131u + it) * 0x9E3779B1u + lane * 0x85EBCA6Bu;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    const unsigned spread = it & 7u;  // 0: one counter for the whole wave ... 7: 256 counters
    const unsigned d = spread == 0 ? (it >> 3) & 255u : (h >> 8) & ((2u << spread) - 1u);
    const bool valid = (it & 8u) == 0 || (h & 3u) != 0;  // every other group of 8: a quarter of the lanes masked off
    LaneMask m = match_digit<8>(d);
    const unsigned long long v = __ballot(valid);
    m.lo &= static_cast<unsigned>(v);
    m.hi &= static_cast<unsigned>(v >> 32);
    const unsigned expect = s_cnt[wave][d] + lanes_before(m);
    if (valid) {
      const unsigned got = atomicAdd(&s_cnt[wave][d], 1u);
      bad += got != expect;
    }
  }
  if (bad) atomicAdd(mismatches, bad);
}

std::atomic<int> g_rank_verdict[64];  // per device, set by dbhip_radix_sort_prepare: 0 not run, 1 atomics, 2 ballots
int rank_forced() {
  static const int forced = [] {
    const char *e = std::getenv("DBHIP_RS_RANK");
    return !e ? 0 : (std::strcmp(e, "atomic") == 0 ? 1 : (std::strcmp(e, "ballot") == 0 ? 2 : 0));
  }();
  return forced;
}
// true: the scatter ranks by LDS atomics on the current device.  No device work, no synchronisation.
bool rank_by_lds_atomics() {
  const int forced = rank_forced();
  if (forced) return forced == 1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  const int known = g_rank_verdict[dev].load(std::memory_order_acquire);
  if (known) return known == 1;
  return current_device_info().gfx950;
}
unsigned rank_fault_injection() {  // test hook: one swapped pair inside a digit run of the first tile, every pass
  static const unsigned on = [] { const char *e = std::getenv("DBHIP_RS_INJECT_UNSTABLE"); return e && e[0] == '1' ? 1u : 0u; }();
  return on;
}

// ---- per pass: stable scatter of every chunk ---------------------------------------------------------------------
// LDS of the scatter kernel, handed to the tile routine
template <int BITS>
struct RsLds {
  unsigned (*cnt)[1 << BITS];  // per-wave digit counts, then wave-exclusive offsets
  unsigned *dexcl;             // tile-local exclusive offset of each digit
  unsigned *goff;              // global offset of a digit minus its local offset
  unsigned *wsum, *wsum2;
  unsigned *keys;
  unsigned *bnd;               // 4-bit: first output position of digit d that belongs to the SECOND destination chunk
  unsigned *thr;               // 4-bit, per tile: first LDS slot of digit d whose key goes to the second chunk
  unsigned *h2;                // 4-bit: [16][16][2] next-digit counts per (digit, next digit, first / second destination chunk)
};

// One tile of the scatter: stable rank inside each wave, digit offsets across waves, re-order through LDS, write out
// in digit order.  FULL = the tile holds kRsTile keys: no per-key bounds checks (the kernel is VALU-bound — about 100
// vector instructions per 64 keys with ballots, 80 % of the issue slots at 2^24 keys by the SQ counters).
// shift2 >= 0 (4-bit only): count the digit at shift2 of every key against its destination chunk (see the file header).
// base_from (8-bit only, the workgroup's first tile): `running` is still relative to the digit's base; the base = the
// exclusive scan of the digits' totals (total_d) rides on the tile's own scan and barrier.
// Returns nonzero if the tile, re-ordered by its digit, was not sorted by its low (shift + BITS) bits.
template <int BITS, bool FULL, bool ARANK>
__device__ __forceinline__ unsigned rs_scatter_tile(const unsigned *__restrict__ src, unsigned *__restrict__ dst,
                                                    size_t tile_base, unsigned valid_in_tile, int shift, int shift2,
                                                    unsigned xor_mask, unsigned low_mask, bool inject, bool base_from,
                                                    unsigned total_d, unsigned &running, const RsLds<BITS> &L) {
  constexpr int kRadix = 1 << BITS;
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const unsigned wave_first = wave * kRsWaveKeys + lane;
  unsigned key[kRsKpt];
#pragma unroll
  for (int j = 0; j < kRsKpt; ++j) {
    const unsigned idx = wave_first + j * kWave;
    key[j] = (FULL || idx < valid_in_tile) ? src[tile_base + idx] : 0xFFFFFFFFu;
  }
  for (int i = tid; i < kRsWaves * kRadix; i += kRsThreads) (&L.cnt[0][0])[i] = 0;
  __syncthreads();

  // ---- stable rank of every key among the keys of its wave with the same digit
  unsigned rank[kRsKpt];
#pragma unroll
  for (int j = 0; j < kRsKpt; ++j) {
    const bool valid = FULL || wave_first + j * kWave < valid_in_tile;
    const unsigned d = ((key[j] ^ xor_mask) >> shift) & (kRadix - 1);
    if (ARANK) {  // one returning LDS atomic: see rs_rank_selftest_kernel for why this is a stable rank
      rank[j] = valid ? atomicAdd(&L.cnt[wave][d], 1u) : 0u;
      continue;
    }
    LaneMask m = match_digit<BITS>(d);
    if (!FULL) {
      const unsigned long long v = __ballot(valid);
      m.lo &= static_cast<unsigned>(v);
      m.hi &= static_cast<unsigned>(v >> 32);
    }
    const unsigned prior = lanes_before(m);
    const unsigned c = L.cnt[wave][d];  // same address inside a digit group: LDS broadcast
    rank[j] = c + prior;
    if (valid && prior == 0) L.cnt[wave][d] = c + lanes_in(m);  // group leader
  }
  __syncthreads();

  // ---- digit owners: counts across waves -> wave-exclusive offsets, tile totals
  unsigned tile_count = 0;
  if (tid < kRadix) {
#pragma unroll
    for (int w = 0; w < kRsWaves; ++w) {
      const unsigned c = L.cnt[w][tid];
      L.cnt[w][tid] = tile_count;
      tile_count += c;
    }
  }
  const unsigned incl = wave_inclusive_scan(tile_count);
  unsigned incl_t = 0;
  if (BITS == 8 && base_from) incl_t = wave_inclusive_scan(total_d);
  if (lane == kWave - 1) {
    L.wsum[wave] = incl;
    if (BITS == 8 && base_from) L.wsum2[wave] = incl_t;
  }
  __syncthreads();
  unsigned dexcl = incl - tile_count;
  for (unsigned w = 0; w < wave; ++w) dexcl += L.wsum[w];
  if (BITS == 8 && base_from) {
    unsigned base = incl_t - total_d;
    for (unsigned w = 0; w < wave; ++w) base += L.wsum2[w];
    running += base;
  }
  if (tid < kRadix) {
    L.dexcl[tid] = dexcl;
    L.goff[tid] = running - dexcl;
    if (BITS == 4) {  // keys of this digit with a run index >= bnd - running go to the second destination chunk
      const unsigned b = L.bnd[tid];
      const unsigned room = b > running ? b - running : 0u;
      L.thr[tid] = dexcl + (room < tile_count ? room : tile_count);
    }
    running += tile_count;
  }
  __syncthreads();

  // ---- re-order the tile by digit in LDS; 4-bit: count the key's next digit against its destination chunk here, where
  // the lanes of a wave hold keys of all digits (in the write-out loop a wave holds ONE digit's run: 64 lanes on 16
  // counters, 4-way same-address and bank conflicts on every instruction: the scatter took 40 us instead of 27)
#pragma unroll
  for (int j = 0; j < kRsKpt; ++j) {
    if (FULL || wave_first + j * kWave < valid_in_tile) {
      const unsigned kx = key[j] ^ xor_mask;
      const unsigned d = (kx >> shift) & (kRadix - 1);
      const unsigned slot = L.dexcl[d] + L.cnt[wave][d] + rank[j];
      L.keys[slot] = key[j];
      if (BITS == 4 && shift2 >= 0) {
        const unsigned d2 = (kx >> shift2) & (kRadix - 1);
        atomicAdd(&L.h2[((d << 4 | d2) << 1) + (slot >= L.thr[d] ? 1u : 0u)], 1u);
      }
    }
  }
  __syncthreads();
  if (inject) {  // test hook (uniform): swap the first neighbours that share the digit and differ below it
    if (tid == 0)
      for (unsigned p = 0; p + 1 < valid_in_tile && p < 4096; ++p) {
        const unsigned a = L.keys[p] ^ xor_mask, b = L.keys[p + 1] ^ xor_mask;
        if (((a ^ b) >> shift & (kRadix - 1)) == 0 && ((a ^ b) & low_mask) != 0) {
          const unsigned t = L.keys[p];
          L.keys[p] = L.keys[p + 1];
          L.keys[p + 1] = t;
          break;
        }
      }
    __syncthreads();
  }

  // ---- write out in digit order: consecutive lanes -> consecutive addresses inside a digit run
  unsigned bad = 0;
#pragma unroll
  for (int k = 0; k < kRsKpt; ++k) {
    const unsigned p = k * kRsThreads + tid;
    if (FULL || p < valid_in_tile) {
      const unsigned kk = L.keys[p];
      const unsigned kx = kk ^ xor_mask;
      // the tile as it now lies in LDS must be sorted by its low (shift + BITS) bits (stable ranking, here and in
      // every earlier pass): my left neighbour's may not exceed mine.  The neighbour's key comes over DPP (wave_shr:1;
      // lane 0 keeps the 0).  Checked on four of the sixteen 512-key bands of the tile: three vector instructions per
      // checked key in a VALU-bound kernel — all sixteen cost 10 % of the pass (28.1 -> 31.4 us at 2^24 keys), an
      // unstable rank is a property of the hardware and shows in any band.
      if (DBHIP_RS_CHECK_MOD != 0 && k % (DBHIP_RS_CHECK_MOD ? DBHIP_RS_CHECK_MOD : 1) == 0) {
        const unsigned mine = kx & low_mask;
        const unsigned left = __builtin_amdgcn_update_dpp(0u, mine, 0x138, 0xf, 0xf, false);
        bad |= left > mine ? 1u : 0u;
      }
      const unsigned d = (kx >> shift) & (kRadix - 1);
      dst[L.goff[d] + p] = kk;
    }
  }
  __syncthreads();
  return bad;
}

template <int BITS, bool ARANK>
__global__ __launch_bounds__(kRsThreads, DBHIP_RS_WPE) void rs_chunk_scatter_kernel(unsigned *keys, unsigned *tmp, size_t n, int pass,
                                                                         unsigned xor_mask, RsHeader *hdr,
                                                                         const unsigned *__restrict__ offsets,
                                                                         const unsigned *__restrict__ totals,
                                                                         unsigned *next_counts, unsigned fused_scan,
                                                                         unsigned inject, size_t tiles_per_chunk,
                                                                         size_t num_chunks) {
  // 8-bit: offsets = counts[256][chunks] after rs_chunk_scan (fused_scan: still the raw counts, and this kernel sums its
  //        own prefix and the totals — few chunks: the separate scan kernel would only add a dependent launch, ~5 us each
  //        at small sizes), totals = this pass's row totals.
  // 4-bit: offsets = off[chunks][16] of rs_scan4 (global positions), next_counts = cnt2 (both matrices).
  constexpr int kRadix = 1 << BITS;
  __shared__ unsigned s_cnt[kRsWaves][kRadix];
  __shared__ unsigned s_dexcl[kRadix];
  __shared__ unsigned s_goff[kRadix];
  __shared__ unsigned s_wsum[kRsWaves], s_wsum2[kRsWaves];
  __shared__ unsigned s_keys[kRsTile];
  __shared__ unsigned s_bnd[BITS == 4 ? 16 : 1], s_c0[BITS == 4 ? 16 : 1], s_thr[BITS == 4 ? 16 : 1];
  __shared__ unsigned s_h2[BITS == 4 ? 512 : 1];
  const RsLds<BITS> L{s_cnt, s_dexcl, s_goff, s_wsum, s_wsum2, s_keys, s_bnd, s_thr, s_h2};

#if defined(DBHIP_RS_DIAG) && (DBHIP_RS_DIAG & 4)
  const unsigned varying = 0xFFFFFFFFu;
  if (hdr->status == 12345u) return;
#else
  const unsigned varying = rs_varying(hdr);
#endif
  if (!rs_varies<BITS>(varying, pass)) return;  // uniform over the grid
  const unsigned par = rs_parity<BITS>(varying, pass);
  const unsigned *__restrict__ src = par ? tmp : keys;
  unsigned *__restrict__ dst = par ? keys : tmp;
  const int shift = pass * BITS;
  const int next = BITS == 4 ? rs_next_pass<BITS>(varying, pass) : -1;
  const int shift2 = next >= 0 ? next * BITS : -1;
  const unsigned low_mask = shift + BITS >= 32 ? 0xFFFFFFFFu : (1u << (shift + BITS)) - 1u;

  const unsigned tid = threadIdx.x;
  // XCD-aware chunk order (speed only): workgroups go to the 8 XCDs round-robin by blockIdx; XCD x takes the x-th
  // eighth of the chunks, so neighbouring chunks — whose digit runs are neighbours in the output and share the
  // partial lines at their ends — are written through the same L2.
  const size_t per_xcd = (num_chunks + 7) / 8;
  const size_t chunk = (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u;
  if (chunk >= num_chunks || blockIdx.x / 8u >= per_xcd) return;
  const size_t first_tile = chunk * tiles_per_chunk;
  const size_t total_tiles = (n + kRsTile - 1) / kRsTile;
  size_t last_tile = first_tile + tiles_per_chunk;
  last_tile = last_tile < total_tiles ? last_tile : total_tiles;
  // digit owners keep the chunk's running global offset of their digit in a register
  unsigned running = 0, total = 0;
  if (BITS == 8) {
    if (tid < kRadix) {
      const unsigned *row = offsets + static_cast<size_t>(tid) * num_chunks;
      if (fused_scan) {
        for (size_t c = 0; c < num_chunks; ++c) {
          const unsigned v = row[c];
          if (c < chunk) running += v;
          total += v;
        }
      } else {
        running = row[chunk];
#if !(defined(DBHIP_RS_DIAG) && (DBHIP_RS_DIAG & 2))
        total = totals[pass * kRsMaxRadix + tid];
#endif
      }
    }
    // (the digit base = exclusive scan of the 256 totals is added inside the first tile: done here it put a dependent
    //  global load and two barriers in front of the tile's key loads, 28 -> 35 us per pass at 2^24 keys)
  } else {
    const size_t chunk_keys = tiles_per_chunk * kRsTile;
    if (tid < kRadix) {
      running = offsets[chunk * kRadix + tid];
      const unsigned c0 = static_cast<unsigned>(running / chunk_keys);
      const unsigned long long bnd = static_cast<unsigned long long>(c0 + 1u) * chunk_keys;
      s_c0[tid] = c0;
      s_bnd[tid] = bnd > 0xFFFFFFFFull ? 0xFFFFFFFFu : static_cast<unsigned>(bnd);
    }
    for (int i = tid; i < 512; i += kRsThreads) s_h2[i] = 0;
    // (the first tile's barriers order these LDS writes before their first use in its write-out loop)
  }

  // (prefetching the next tile's keys into a second register set was measured: it needs 3 waves/SIMD
  //  instead of 4 to avoid spills and came out 7 % slower at 2^24 keys.  Round 2 repeated it the way that pays in
  //  the join's level-0 scatter — persistent grid of 2-8 workgroups per CU, next tile's keys and digit offsets waited
  //  for right before the current tile's stores, carried across the loop through an opaque v_mov so that no vmcnt(0)
  //  sits at the loop head — with 16 and 8 keys per lane at 3-6 waves per SIMD: 288-360 us against 272 us for one
  //  chunk per workgroup.  This kernel is not waiting for its loads.)
  unsigned bad = 0;
  for (size_t tile = first_tile; tile < last_tile; ++tile) {
    const size_t tile_base = tile * kRsTile;
    const unsigned valid_in_tile = static_cast<unsigned>(n - tile_base < kRsTile ? n - tile_base : kRsTile);
#if defined(DBHIP_RS_DIAG) && (DBHIP_RS_DIAG & 1)
    const bool inj = false;
#else
    const bool inj = inject != 0 && tile == 0;
#endif
#if defined(DBHIP_RS_DIAG) && (DBHIP_RS_DIAG & 2)
    const bool first = false;
#else
    const bool first = BITS == 8 && tile == first_tile;
#endif
    if (valid_in_tile == kRsTile)  // every tile but the input's last one
      bad |= rs_scatter_tile<BITS, true, ARANK>(src, dst, tile_base, valid_in_tile, shift, shift2, xor_mask, low_mask, inj,
                                                first, total, running, L);
    else
      bad |= rs_scatter_tile<BITS, false, ARANK>(src, dst, tile_base, valid_in_tile, shift, shift2, xor_mask, low_mask, inj,
                                                 first, total, running, L);
  }
  if (bad) atomicOr(&hdr->status, DBHIP_DEV_RANK_ORDER);
  if (BITS == 4 && shift2 >= 0) {
    // the next pass's counts: table entry (d, d', w) belongs to chunk c0[d] + w of the next pass (the tiles' last barrier
    // has ordered the table's atomics before these reads); 32 consecutive lanes add to two runs of 64 contiguous bytes
    unsigned *nc = next_counts + static_cast<size_t>(par ^ 1u) * num_chunks * 16;
    for (unsigned i = tid; i < 512; i += kRsThreads) {
      const unsigned v = s_h2[i];
      const size_t c = static_cast<size_t>(s_c0[i >> 5]) + (i & 1u);
      if (v && c < num_chunks) atomicAdd(&nc[c * 16 + ((i >> 1) & 15u)], v);
    }
  }
}

// ---- n <= one tile: the whole sort in ONE workgroup and one launch ------------------------------------------
// (the reference's small sweeps and dwarf tests run 128..65536 keys, where the ~16 launches of the general path
// are all that is measured).  Keys stay in registers between passes; every pass ranks them exactly as the
// scatter kernel does and re-orders them through LDS; passes whose digit is constant over the input are skipped
// on a workgroup-uniform vote; the result is written back to `keys`.
template <int BITS, bool ARANK>
__global__ __launch_bounds__(kRsThreads, DBHIP_RS_WPE) void rs_single_tile_kernel(unsigned *keys, unsigned n, unsigned xor_mask,
                                                                                  unsigned *status) {
  constexpr int kRadix = 1 << BITS;
  constexpr int kPasses = 32 / BITS;
  __shared__ unsigned s_cnt[kRsWaves][kRadix];
  __shared__ unsigned s_dexcl[kRadix];
  __shared__ unsigned s_wsum[kRsWaves];
  __shared__ unsigned s_keys[kRsTile];
  __shared__ unsigned s_or, s_and;
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const unsigned wave_first = wave * kRsWaveKeys + lane;

  unsigned key[kRsKpt];
  unsigned my_or = 0, my_and = ~0u;
#pragma unroll
  for (int j = 0; j < kRsKpt; ++j) {
    const unsigned idx = wave_first + j * kWave;
    key[j] = idx < n ? keys[idx] : 0xFFFFFFFFu;
    if (idx < n) {
      my_or |= key[j];
      my_and &= key[j];
    }
  }
  if (tid == 0) {
    s_or = 0;
    s_and = ~0u;
  }
  __syncthreads();
  atomicOr(&s_or, my_or);
  atomicAnd(&s_and, my_and);
  __syncthreads();
  const unsigned varying = s_or ^ s_and;  // bits that differ somewhere in the input

  unsigned bad = 0;
  for (int pass = 0; pass < kPasses; ++pass) {
    const int shift = pass * BITS;
    if (((varying >> shift) & (kRadix - 1)) == 0) continue;  // constant digit: the pass is the identity
    const unsigned low_mask = shift + BITS >= 32 ? 0xFFFFFFFFu : (1u << (shift + BITS)) - 1u;
    for (int i = tid; i < kRsWaves * kRadix; i += kRsThreads) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    unsigned rank[kRsKpt];
#pragma unroll
    for (int j = 0; j < kRsKpt; ++j) {
      const bool valid = wave_first + j * kWave < n;
      const unsigned d = ((key[j] ^ xor_mask) >> shift) & (kRadix - 1);
      if (ARANK) {
        rank[j] = valid ? atomicAdd(&s_cnt[wave][d], 1u) : 0u;
        continue;
      }
      LaneMask m = match_digit<BITS>(d);
      const unsigned long long v = __ballot(valid);
      m.lo &= static_cast<unsigned>(v);
      m.hi &= static_cast<unsigned>(v >> 32);
      const unsigned prior = lanes_before(m);
      const unsigned c = s_cnt[wave][d];
      rank[j] = c + prior;
      if (valid && prior == 0) s_cnt[wave][d] = c + lanes_in(m);
    }
    __syncthreads();
    unsigned tile_count = 0;
    if (tid < kRadix) {
#pragma unroll
      for (int w = 0; w < kRsWaves; ++w) {
        const unsigned c = s_cnt[w][tid];
        s_cnt[w][tid] = tile_count;
        tile_count += c;
      }
    }
    const unsigned incl = wave_inclusive_scan(tile_count);
    if (lane == kWave - 1) s_wsum[wave] = incl;
    __syncthreads();
    unsigned dexcl = incl - tile_count;
    for (unsigned w = 0; w < wave; ++w) dexcl += s_wsum[w];
    if (tid < kRadix) s_dexcl[tid] = dexcl;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kRsKpt; ++j) {
      if (wave_first + j * kWave < n) {
        const unsigned d = ((key[j] ^ xor_mask) >> shift) & (kRadix - 1);
        s_keys[s_dexcl[d] + s_cnt[wave][d] + rank[j]] = key[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kRsKpt; ++j) {
      const unsigned idx = wave_first + j * kWave;
      key[j] = idx < n ? s_keys[idx] : 0xFFFFFFFFu;
      if (idx < n && idx > 0)  // the same order check as in rs_scatter_tile
        bad |= ((s_keys[idx - 1] ^ xor_mask) & low_mask) > ((key[j] ^ xor_mask) & low_mask) ? 1u : 0u;
    }
    __syncthreads();
  }
  if (bad) atomicOr(status, DBHIP_DEV_RANK_ORDER);
#pragma unroll
  for (int j = 0; j < kRsKpt; ++j) {
    const unsigned idx = wave_first + j * kWave;
    if (idx < n) keys[idx] = key[j];
  }
}

template <int BITS>
__global__ __launch_bounds__(kRsThreads) void rs_finalize_kernel(unsigned *__restrict__ keys, const unsigned *__restrict__ tmp,
                                                                 size_t n, const RsHeader *hdr) {
  if (!rs_parity<BITS>(rs_varying(hdr), 32 / BITS)) return;  // an even number of passes ran
  const size_t stride = static_cast<size_t>(gridDim.x) * kRsThreads;
  const size_t n4 = n / 4;
  const u32x4 *s4 = reinterpret_cast<const u32x4 *>(tmp);
  u32x4 *d4 = reinterpret_cast<u32x4 *>(keys);
  for (size_t i = static_cast<size_t>(blockIdx.x) * kRsThreads + threadIdx.x; i < n4; i += stride)
    d4[i] = s4[i];
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) keys[n4 * 4 + threadIdx.x] = tmp[n4 * 4 + threadIdx.x];
}

template <int BITS>
size_t rs_workspace_bytes(const RsGeometry &g) {
  return align_up(BITS == 8 ? kRsCountsOff + sizeof(unsigned) * 256 * g.chunks
                            : kRs4CntOff + sizeof(unsigned) * 16 * g.chunks * 3,
                  kWsAlign);
}

template <int BITS>
int radix_sort_impl(unsigned *keys, unsigned *tmp, size_t n, unsigned xor_mask, void *workspace,
                    hipStream_t s, const DeviceInfo &dev) {
  constexpr int kPasses = 32 / BITS;
  const RsGeometry g = rs_geometry(n, BITS);
  char *base = static_cast<char *>(workspace);
  RsHeader *hdr = reinterpret_cast<RsHeader *>(base);

  const bool arank = rank_by_lds_atomics();
  const unsigned inject = rank_fault_injection();
  if (n <= static_cast<size_t>(kRsTile)) {  // one tile: one workgroup, one launch (+ the status word)
    const hipError_t e0 = fill_async(workspace, 0, kWsHeader, s);
    if (e0 != hipSuccess) return static_cast<int>(e0);
    if (arank)
      hipLaunchKernelGGL((rs_single_tile_kernel<BITS, true>), dim3(1), dim3(kRsThreads), 0, s, keys,
                         static_cast<unsigned>(n), xor_mask, &hdr->status);
    else
      hipLaunchKernelGGL((rs_single_tile_kernel<BITS, false>), dim3(1), dim3(kRsThreads), 0, s, keys,
                         static_cast<unsigned>(n), xor_mask, &hdr->status);
    return launch_status();
  }
  hipError_t e = fill_async(workspace, 0, kWsHeader, s);
  if (e != hipSuccess) return static_cast<int>(e);

  const size_t want = (n / 4 + kRsThreads - 1) / kRsThreads;
  const size_t fcap = static_cast<size_t>(dev.cus) * 2;  // (after an even number of passes its workgroups return at once)
  const unsigned fgrid = static_cast<unsigned>(want < fcap ? (want ? want : 1) : fcap);
  constexpr int kCpw = rs_hist_cpw<BITS>();
  const size_t hist_groups = (g.chunks + kCpw - 1) / kCpw;
  const size_t ucap = static_cast<size_t>(dev.cus) * 2;  // persistent: one header atomic per workgroup (see rs_upfront_kernel)
  const unsigned hist_grid = static_cast<unsigned>(hist_groups < ucap ? hist_groups : ucap);
  const unsigned cgrid = static_cast<unsigned>(g.chunks);
  const unsigned sgrid = (cgrid + 7) / 8 * 8;
  if (BITS == 8) {
    unsigned *totals = reinterpret_cast<unsigned *>(base + kRsTotalsOff);
    unsigned *counts = reinterpret_cast<unsigned *>(base + kRsCountsOff);
    const unsigned fused_scan = g.chunks <= kRsFusedScanChunks ? 1u : 0u;
    hipLaunchKernelGGL((rs_upfront_kernel<BITS>), dim3(hist_grid), dim3(kRsThreads), 0, s, keys, n, xor_mask, hdr, counts,
                       g.tiles_per_chunk, g.chunks);
    for (int p = 0; p < kPasses; ++p) {
      if (p > 0)  // pass 0's chunk counts came with the up-front read
        hipLaunchKernelGGL((rs_chunk_hist_kernel<BITS, false>), dim3(static_cast<unsigned>(hist_groups)), dim3(kRsThreads), 0, s,
                           keys, tmp, n, p, xor_mask, hdr, counts, g.tiles_per_chunk, g.chunks);
      if (!fused_scan)
        hipLaunchKernelGGL(rs_chunk_scan_kernel, dim3(256), dim3(kRsThreads), 0, s, p, hdr, totals, counts, g.chunks);
      if (arank)
        hipLaunchKernelGGL((rs_chunk_scatter_kernel<BITS, true>), dim3(sgrid), dim3(kRsThreads), 0, s, keys, tmp, n, p,
                           xor_mask, hdr, counts, totals, static_cast<unsigned *>(nullptr), fused_scan, inject,
                           g.tiles_per_chunk, g.chunks);
      else
        hipLaunchKernelGGL((rs_chunk_scatter_kernel<BITS, false>), dim3(sgrid), dim3(kRsThreads), 0, s, keys, tmp, n, p,
                           xor_mask, hdr, counts, totals, static_cast<unsigned *>(nullptr), fused_scan, inject,
                           g.tiles_per_chunk, g.chunks);
    }
  } else {
    unsigned *cnt2 = reinterpret_cast<unsigned *>(base + kRs4CntOff);
    unsigned *off = cnt2 + 2 * 16 * g.chunks;
    hipLaunchKernelGGL((rs_upfront_kernel<BITS>), dim3(hist_grid), dim3(kRsThreads), 0, s, keys, n, xor_mask, hdr, cnt2,
                       g.tiles_per_chunk, g.chunks);
    // (does something only when digit 0 is constant; a grid of one workgroup per CU returns in ~2 us, 1024 take ~5)
    hipLaunchKernelGGL((rs_chunk_hist_kernel<BITS, true>), dim3(cgrid < static_cast<unsigned>(dev.cus) ? cgrid : dev.cus),
                       dim3(kRsThreads), 0, s, keys, tmp, n, 0, xor_mask, hdr, cnt2, g.tiles_per_chunk, g.chunks);
    for (int p = 0; p < kPasses; ++p) {
      hipLaunchKernelGGL(rs_scan4_kernel, dim3(1), dim3(1024), 0, s, p, hdr, cnt2, off, g.chunks);
      if (arank)
        hipLaunchKernelGGL((rs_chunk_scatter_kernel<BITS, true>), dim3(sgrid), dim3(kRsThreads), 0, s, keys, tmp, n, p,
                           xor_mask, hdr, off, static_cast<const unsigned *>(nullptr), cnt2, 0u, inject, g.tiles_per_chunk,
                           g.chunks);
      else
        hipLaunchKernelGGL((rs_chunk_scatter_kernel<BITS, false>), dim3(sgrid), dim3(kRsThreads), 0, s, keys, tmp, n, p,
                           xor_mask, hdr, off, static_cast<const unsigned *>(nullptr), cnt2, 0u, inject, g.tiles_per_chunk,
                           g.chunks);
    }
  }
  hipLaunchKernelGGL((rs_finalize_kernel<BITS>), dim3(fgrid), dim3(kRsThreads), 0, s, keys, tmp, n, hdr);
  return launch_status();
}

int radix_sort_entry(unsigned *keys, unsigned *tmp, size_t n, int radix_bits, unsigned xor_mask,
                     void *workspace, size_t workspace_bytes, dbhip_stream_t stream) {
  if (radix_bits != 4 && radix_bits != 8) return DBHIP_EINVAL;
  if (n >= (1ull << 32)) return DBHIP_EINVAL;  // 32-bit offsets
  if (n == 0) {  // nothing to sort; a workspace that was passed still gets a clean status word
    if (workspace && ws_ok(workspace, workspace_bytes, kWsHeader))
      return static_cast<int>(fill_async(workspace, 0, kWsHeader, as_stream(stream)));
    return DBHIP_OK;
  }
  if (!keys || !tmp) return DBHIP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(keys) | reinterpret_cast<uintptr_t>(tmp)) & 15u) return DBHIP_EINVAL;  // dbhip.h: 16-byte aligned
  if (!ws_ok(workspace, workspace_bytes, dbhip_radix_sort_workspace_bytes(n, radix_bits)))
    return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  return radix_bits == 8
             ? radix_sort_impl<8>(keys, tmp, n, xor_mask, workspace, as_stream(stream), dev)
             : radix_sort_impl<4>(keys, tmp, n, xor_mask, workspace, as_stream(stream), dev);
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

extern "C" size_t dbhip_radix_sort_workspace_bytes(size_t n, int radix_bits) {
  if (radix_bits != 4 && radix_bits != 8) return 0;
  const RsGeometry g = rs_geometry(n, radix_bits);
  return radix_bits == 8 ? rs_workspace_bytes<8>(g) : rs_workspace_bytes<4>(g);
}

extern "C" int dbhip_radix_sort_rank_mode(void) {
  int dev = 0;
  if (!rank_forced() && (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)) return -1;
  return rank_by_lds_atomics() ? 1 : 0;
}

extern "C" int dbhip_radix_sort_prepare(dbhip_stream_t stream) {
  hipStream_t s = as_stream(stream);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DBHIP_ENODEVICE;
  const DeviceInfo &info = current_device_info();
  if (!info.ok) return DBHIP_ENODEVICE;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return DBHIP_EINVAL;
  unsigned *scratch = nullptr;
  unsigned host = 1;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&scratch), kWsAlign);
  if (e != hipSuccess) return static_cast<int>(e);
  e = fill_async(scratch, 0, sizeof(unsigned), s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(rs_rank_selftest_kernel, dim3(info.cus * 2), dim3(kRsThreads), 0, s, scratch);  // ~50 us
    e = hipMemcpyAsync(&host, scratch, sizeof(host), hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(scratch);
  if (e != hipSuccess) return static_cast<int>(e);
  g_rank_verdict[dev].store(host == 0 ? 1 : 2, std::memory_order_release);
  return rank_by_lds_atomics() ? 1 : 0;
}

extern "C" int dbhip_radix_sort_u32(uint32_t *keys, uint32_t *tmp, size_t n, int radix_bits,
                                    void *workspace, size_t workspace_bytes, dbhip_stream_t stream) {
  return radix_sort_entry(keys, tmp, n, radix_bits, 0u, workspace, workspace_bytes, stream);
}

extern "C" int dbhip_radix_sort_i32(int32_t *keys, int32_t *tmp, size_t n, int radix_bits,
                                    void *workspace, size_t workspace_bytes, dbhip_stream_t stream) {
  // signed order = unsigned order with the sign bit flipped (applied on the fly, keys unchanged)
  return radix_sort_entry(reinterpret_cast<unsigned *>(keys), reinterpret_cast<unsigned *>(tmp), n,
                          radix_bits, 0x80000000u, workspace, workspace_bytes, stream);
}
