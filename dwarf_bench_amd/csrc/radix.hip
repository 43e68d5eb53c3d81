// radix.hip — dwarf 2: LSD radix sort of 32-bit keys for gfx950.
//
// Replaces oneDPL's std::sort(device_policy) behind Radix/RadixCuda (dpl_wrapper.hpp:35-39 <-
// sort/radix.cpp:34); result identical to std::sort (sort/radix.cpp:8-12).
//
// Structure (digit width BITS = 8 tuned, 4 = the configuration named in BASELINE.json):
//   1. rs_histogram      one read of the keys: global digit totals for EVERY pass (LDS histograms per
//                        workgroup, one coalesced atomic flush per workgroup into one of eight copies of the totals).
//   2. rs_plan           one workgroup: per pass the exclusive digit bases, and which passes are skipped
//                        because their digit is constant over the whole input (e.g. keys in [1,10000]
//                        skip the two upper bytes); fixes the ping-pong parity of every pass.
//   3. per executed pass, three kernels over CHUNKS of consecutive 8192-key tiles, with no communication
//      between workgroups inside a kernel:
//        rs_chunk_hist     digit counts of every chunk (LDS histogram) -> counts[digit][chunk]
//        rs_chunk_scan     one workgroup per digit: exclusive scan of its row + the digit base
//        rs_chunk_scatter  one workgroup per chunk, tile by tile: each wave ranks its 1024 keys stably
//                          with wave64 match masks (BITS ballots per key — CDNA has no match
//                          instruction —, v_mbcnt for the lane rank, wave-private LDS digit counters),
//                          the tile is re-ordered by digit through LDS and written out in digit order,
//                          so consecutive lanes hit consecutive addresses; the chunk's running
//                          per-digit offsets live in registers of the digit-owner threads.
//      (A single-pass Onesweep with decoupled look-back was measured first: 106 us per 8-bit pass at
//      2^24 keys, dominated by look-back waits between tiles in flight; same finding as in scan.hip.)
//   4. rs_finalize       copies tmp -> keys when an odd number of passes ran.
//
// Round 3, measured and NOT kept (commit 790ed71 holds the code; same-box A/B at 2^24 full-range keys, round 2's
// library beside it: 8-bit 203.8 us, 4-bit 361.1 us):
//   * 4-bit passes that read their keys ONCE: the scatter of pass k counts every key's next digit d' against the key's
//     destination chunk (a chunk's keys of digit d are one contiguous run of the output and lie in at most two of the next
//     pass's position-defined chunks: an LDS table of 16 x 16 x 2 counters, added to the next pass's count matrix with 512
//     coalesced global atomics per workgroup), one single-workgroup kernel per pass turns the matrix into offsets.  No
//     rs_chunk_hist after the first pass (7 x 12 us saved) — and 367.6 us: the counting costs the scatter 6 us per pass
//     (27.5 -> 35.6 us; counting in the write-out loop, where a wave holds one digit's run and 64 lanes hit 16 counters:
//     40 us), the offsets kernel takes 7.9 us — a kernel that reads one word and returns takes 4.8 us behind a scatter.  With 8-bit digits the table would have
//     2 x 256 x 256 counters per chunk and about one key per counter: no aggregation, one global atomic per key.
//   * an up-front read that only counts digit 0 per chunk and ORs the keys (a pass is skipped iff none of its digit's bits
//     varies; every kernel derives skip flags and ping-pong parity from that word, no plan kernel), digit totals from the
//     per-pass scan, digit bases added inside the scatter's first tile: up-front 29.7 + 4.6 -> 20.7 us, and every scatter
//     2.7 us slower (28.2 -> 30.7 us; neither the order check, nor the skip logic, nor the fault hook — each compiled out
//     in turn —, 35 % more VALU and 74 % more SALU instructions by the SQ counters but the kernel is bound by neither):
//     207.3 us.  Isolated on round 2's code afterwards: a digit base that is loaded in the prologue and added inside the
//     first tile costs 0.6 us per pass, added in the prologue itself (a use of a loaded value in front of the tile's key
//     loads) 2.3 us; the rest is the 256-entry scan.  At best 5 us for the whole sort: not pursued.  Two lessons kept: 512-1024 workgroups ending with two atomics on the SAME word are served one per ~11 ns
//     (the kernel then takes 34 us instead of 14), and a divide that consumes a loaded offset in the prologue puts that
//     load's latency in front of the tile's key loads.
//   * the scatter's stores as write-through (sc1) or non-temporal stores, so that the histogram kernel behind it would not
//     wait for dirty L2 lines: scatter 28.4 -> 37.6 / 35.3 us (8-bit), the histogram 13.0 -> 13.0 / 16.6 us — the ~4.8 us
//     that any kernel takes behind a scatter, however little it does, are not an L2 write-back.  199 -> 235 / 239 us.
// A single-pass Onesweep was re-examined on paper and not built: a dependent global round trip costs 2-4 us under load
// on this chip (dense scan) against 7 us that a workgroup spends on an 8192-key tile, and when all chunks of
// a pass are in flight at once (2^24 keys = 512 workgroups x 32768 keys) a look-back has nothing finished to look back on.
//
// Bytes: 4N (histogram) + P * 12N (chunk histogram read + scatter read/write), P <= 32/BITS; at 2^24
// keys both ping-pong buffers (128 MiB) live in the 256 MiB Infinity Cache.  Everything between passes
// stays device-side (skipped passes return at once on a device-side flag, no host round trip).
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
#ifndef DBHIP_RS_CROWD_ROWS
#define DBHIP_RS_CROWD_ROWS 2
#endif
constexpr int kRsKpt = DBHIP_RS_KPT;             // keys per lane per tile
constexpr int kRsWaveKeys = kWave * kRsKpt;      // 1024 contiguous keys per wave
constexpr int kRsTile = kRsWaveKeys * kRsWaves;  // 8192 keys
constexpr int kRsMaxPasses = 8;
constexpr int kRsMaxRadix = 256;
#ifndef DBHIP_RS_CHUNKS
#define DBHIP_RS_CHUNKS 2048
#endif
constexpr size_t kRsTargetChunks = DBHIP_RS_CHUNKS;  // chunks per pass (>= 8 per CU for balance)
constexpr size_t kRsFusedScanChunks = 32;        // up to this many chunks the scatter sums its own prefix (2^16 keys: 89 -> 80 us; at 128 chunks it costs 17 us)

struct RsPass {
  unsigned skip;        // digit constant over the input: the pass's kernels return immediately
  unsigned src_is_tmp;  // which buffer holds the keys when this pass starts
};
struct RsHeader {
  unsigned status;
  unsigned final_in_tmp;
  unsigned pad0[6];
  RsPass pass[kRsMaxPasses];
  unsigned pad1[64 - 8 - 2 * kRsMaxPasses];
};
static_assert(sizeof(RsHeader) == kWsHeader, "workspace header size");

// workspace: header | totals[kRsTotalCopies][8][256] | bases[8][256] | counts[radix][chunks]
// The up-front histogram's workgroups flush their digit totals with global atomics.  Into ONE copy of the totals that
// flush was most of the kernel: 512 workgroups x 1024 adds onto the same 32 cache lines are served at the contended rate
// of the memory-side atomic unit (2 MiB at ~0.09 TB/s: 23 of the kernel's 29.7 us at 2^24 keys).  Workgroup b adds into
// copy b % 8 (eight times fewer adders per line); rs_plan sums the copies.
// (Measured and dropped for 4-bit digits: per-pass sums of the chunk counts over groups of 64 chunks, added by the
// histogram workgroups with 16 global atomics each, so that the scatter could find its offsets without a scan kernel —
// the 2048 workgroups hammer the same 32 cache lines of sums: rs_chunk_hist 12.5 -> 28 us, scatter 31 -> 35 us,
// 2^24 keys 458 -> 518 us.)
constexpr int kRsTotalCopies = 8;
constexpr size_t kRsTotalsOff = kWsHeader;
constexpr size_t kRsBasesOff = kRsTotalsOff + sizeof(unsigned) * kRsTotalCopies * kRsMaxPasses * kRsMaxRadix;
constexpr size_t kRsCountsOff = kRsBasesOff + sizeof(unsigned) * kRsMaxPasses * kRsMaxRadix;

struct RsGeometry {
  size_t tiles, tiles_per_chunk, chunks;
};
inline RsGeometry rs_geometry(size_t n, int bits) {
  RsGeometry g;
  // 4-bit digits: half as many, twice as long chunks (2^24 keys: 378 -> 363 us; 8-bit digits: no difference)
  const size_t target = bits == 4 ? kRsTargetChunks / 2 : kRsTargetChunks;
  g.tiles = (n + kRsTile - 1) / kRsTile;
  // Every chunk holds the same number of tiles, q or q + 1 with q = tiles / target — whichever brings the number of
  // chunks closer to the target (as a ratio): q while tiles < target * sqrt(q (q + 1)).  (Until late in round 3 it was
  // always the ceiling: one key more than 2^24 meant 1025 chunks of two tiles instead of 2049 of one — 231 us against
  // 200; dealing the tiles out as evenly as they go, with chunks of q and of q + 1 tiles in one pass, removed that
  // step and lost 10-15 % at sizes in between.)  At most 1.42 * target chunks.
  // 4-bit digits keep the ceiling: their chunks are twice as long and 1025 of them are three rounds of workgroups where
  // 1024 are two (same box, old rule -> this one: 2^26 + 1 keys 1431 -> 1537 us, 2^27 + 12345 keys 2914 -> 3294; with
  // 8-bit digits every size measured got faster or stayed: 2^24 + 1 keys 234 -> 201, 2^24 + 2^20 257 -> 213, 2^25 + 1
  // 464 -> 428, 2^26 + 1 1025 -> 991, 2^27 + 12345 2003 -> 1959).
  const size_t q = g.tiles / target;
  g.tiles_per_chunk = q == 0 ? 1 : (bits != 4 && g.tiles * g.tiles < q * (q + 1) * target * target ? q : q + 1);
  if (g.tiles == q * target && q != 0) g.tiles_per_chunk = q;
  g.chunks = (g.tiles + g.tiles_per_chunk - 1) / g.tiles_per_chunk;
  if (g.chunks == 0) g.chunks = 1;
  return g;
}
// the rows of the count matrix counts[digit][chunk] are padded to four chunks: a histogram workgroup's counts of four
// consecutive chunks are ONE aligned 16-byte store whatever the number of chunks (the padding stays zero)
__host__ __device__ __forceinline__ size_t rs_row_stride(size_t num_chunks) { return (num_chunks + 3) & ~static_cast<size_t>(3); }

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
// 217 -> 208 us; 2: 216, 8: 222), 1 with 4-bit digits (16 counts per chunk: nothing to gain, and 1024 chunks in 256
// workgroups leave CUs idle: 360 -> 377 us with 4)
template <int BITS>
constexpr int rs_hist_cpw() { return BITS == 8 ? 4 : 1; }
// LDS histograms are kept in several copies, lane l adding into copy l % copies, copy c lying c words further (its
// counter of a digit in another bank): the LDS serves the lanes of one atomic that hit the same word one after the
// other, so keys that crowd into a few digits (two distinct values, 90 % one value, sorted input in its top pass)
// made the histogram kernels 4.5-8 x slower than on spread keys (2^24 keys, 8-bit: 13 -> 58 us per pass, the up-front
// read 25 -> 130-200 us).  With 4-bit digits 64 lanes share 16 counters even on uniform keys.
template <int BITS>
constexpr int rs_hist_copies() { return BITS == 8 ? 8 : 16; }

template <int BITS>
__global__ __launch_bounds__(kRsThreads) void rs_histogram_kernel(const unsigned *__restrict__ keys,
                                                                  size_t n, unsigned xor_mask,
                                                                  unsigned *__restrict__ totals,
                                                                  unsigned *__restrict__ counts0,
                                                                  size_t tiles_per_chunk, size_t num_chunks) {
  // One read of the keys: digit totals of EVERY pass and, because the workgroups walk the input chunk by chunk,
  // the per-chunk digit counts of pass 0 as well (counts0[digit][chunk]) — the first pass then needs no
  // rs_chunk_hist of its own.  The LDS histograms are always over BYTES (four ds_add per key): with 4-bit digits the
  // two nibble histograms of a byte are its row and column sums, taken once per chunk / once at the end — eight
  // ds_add per key made this kernel 70 us at 2^24 keys against 32 us for the byte version.
  constexpr int kBins = 256, kBytes = 4;
  constexpr int kRadix = 1 << BITS;
  constexpr int kCpw = rs_hist_cpw<BITS>();  // consecutive chunks whose pass-0 counts are written side by side
  // every histogram in kCopies copies, lane l adding into copy l % kCopies (see rs_hist_copies; four here: seven
  // byte histograms at eight copies would halve the resident workgroups of a kernel that waits for HBM)
  constexpr int kCopies = 4, kStride = kBins + 1;
  __shared__ unsigned s_byte0[kBins];                          // byte 0 over the workgroup's chunks (from s_chunks)
  __shared__ unsigned s_upper[kBytes - 1][kCopies * kStride];  // bytes 1..3 over the workgroup's chunks
  __shared__ unsigned s_chunks[kCpw][kCopies * kStride];       // byte 0 of the chunks in hand
  for (int i = threadIdx.x; i < kBins; i += kRsThreads) s_byte0[i] = 0;
  for (int i = threadIdx.x; i < (kBytes - 1) * kCopies * kStride; i += kRsThreads) (&s_upper[0][0])[i] = 0;
  const unsigned my_copy = (threadIdx.x & (kCopies - 1)) * kStride;
  auto chunk_count = [&](int cc, int d) {
    unsigned sum = 0;
#pragma unroll
    for (int k = 0; k < kCopies; ++k) sum += s_chunks[cc][k * kStride + d];
    return sum;
  };
  auto byte_total = [&](int byte, int d) {
    if (byte == 0) return s_byte0[d];
    unsigned sum = 0;
#pragma unroll
    for (int k = 0; k < kCopies; ++k) sum += s_upper[byte - 1][k * kStride + d];
    return sum;
  };
  const size_t chunk_keys = tiles_per_chunk * kRsTile;
  const size_t groups = (num_chunks + kCpw - 1) / kCpw;
  for (size_t group = blockIdx.x; group < groups; group += gridDim.x) {
    for (int i = threadIdx.x; i < kCpw * kCopies * kStride; i += kRsThreads) (&s_chunks[0][0])[i] = 0;
    __syncthreads();
#pragma unroll 1
    for (int cc = 0; cc < kCpw; ++cc) {
      const size_t chunk = group * kCpw + cc;
      if (chunk >= num_chunks) break;
      unsigned *s_chunk = s_chunks[cc] + my_copy;
      const size_t lo = chunk * chunk_keys;
      size_t hi = lo + chunk_keys;
      hi = hi < n ? hi : n;
      const size_t n4 = (hi - lo) / 4;  // chunk starts are multiples of the tile size: 16-byte loads are aligned
      const u32x4 *k4 = reinterpret_cast<const u32x4 *>(keys + lo);
      auto count4 = [&](const u32x4 v) {
        const unsigned k[4] = {v.x ^ xor_mask, v.y ^ xor_mask, v.z ^ xor_mask, v.w ^ xor_mask};
  #pragma unroll
        for (int p = 0; p < kBytes; ++p) {
          unsigned *hist = p == 0 ? s_chunk : s_upper[p == 0 ? 0 : p - 1] + my_copy;
          // a byte that is the same in the whole wave (the upper bytes of small keys: the reference's
          // [1,10000] data) would serialise 64 same-address ds_add: one lane adds the lot instead
          const unsigned d0 = (k[0] >> (p * 8)) & (kBins - 1);
          const unsigned first = __builtin_amdgcn_readfirstlane(d0);
          const bool same = ((k[0] >> (p * 8)) & (kBins - 1)) == first && ((k[1] >> (p * 8)) & (kBins - 1)) == first &&
                            ((k[2] >> (p * 8)) & (kBins - 1)) == first && ((k[3] >> (p * 8)) & (kBins - 1)) == first;
          const unsigned long long active = __ballot(true);
          if (__ballot(same) == active) {
            if (threadIdx.x % kWave == static_cast<unsigned>(__builtin_ctzll(active)))
              atomicAdd(&hist[first], 4u * static_cast<unsigned>(__builtin_popcountll(active)));
          } else {
  #pragma unroll
            for (int c = 0; c < 4; ++c) atomicAdd(&hist[(k[c] >> (p * 8)) & (kBins - 1)], 1u);
          }
        }
      };
      // a tile's four 16-byte loads per lane are requested together: this kernel reads what comes from HBM (the per-pass
      // histograms read what the scatter before them left in the Infinity Cache) and one load in flight per lane left
      // it latency-bound (round 3: 29.7 -> 25.7 us with the eight copies of the totals, -> this)
      size_t i = threadIdx.x;
      for (; i + 3 * kRsThreads < n4; i += 4 * kRsThreads) {
        const u32x4 v0 = k4[i], v1 = k4[i + kRsThreads], v2 = k4[i + 2 * kRsThreads], v3 = k4[i + 3 * kRsThreads];
        count4(v0);
        count4(v1);
        count4(v2);
        count4(v3);
      }
      for (; i < n4; i += kRsThreads) count4(k4[i]);
      for (size_t i = lo + n4 * 4 + threadIdx.x; i < hi; i += kRsThreads) {  // ragged end of the last chunk
        const unsigned k = keys[i] ^ xor_mask;
        atomicAdd(&s_chunk[k & (kBins - 1)], 1u);
  #pragma unroll
        for (int p = 1; p < kBytes; ++p) atomicAdd(&s_upper[p - 1][my_copy + ((k >> (p * 8)) & (kBins - 1))], 1u);
      }
    }
    __syncthreads();
    // pass 0's counts of these chunks: the byte bins themselves (side by side: one 16-byte store per digit where the
    // row allows it, see rs_chunk_hist_kernel), or (4-bit digits) their sums over the high nibble
    const size_t chunk0 = group * kCpw;
    unsigned mine[kCpw] = {};  // thread d < 256: byte-0 bin d of each chunk in hand
    if (threadIdx.x < kBins) {
#pragma unroll
      for (int cc = 0; cc < kCpw; ++cc) mine[cc] = chunk_count(cc, threadIdx.x);
    }
    __syncthreads();
    if (threadIdx.x < kBins) {  // the summed counts back into copy 0, for the nibble sums below
      unsigned c = 0;
#pragma unroll
      for (int cc = 0; cc < kCpw; ++cc) {
        s_chunks[cc][threadIdx.x] = mine[cc];
        c += mine[cc];
      }
      s_byte0[threadIdx.x] += c;
    }
    if (BITS == 8) {
      const bool vec = kCpw == 4;  // (rows are padded to four chunks)
      if (threadIdx.x < kRadix) {
        unsigned *row = counts0 + static_cast<size_t>(threadIdx.x) * rs_row_stride(num_chunks) + chunk0;
        if (vec) {
          *reinterpret_cast<u32x4 *>(row) = u32x4{mine[0], mine[1 % kCpw], mine[2 % kCpw], mine[3 % kCpw]};
        } else {
          for (int cc = 0; cc < kCpw && chunk0 + cc < num_chunks; ++cc) row[cc] = mine[cc];
        }
      }
    } else {
      __syncthreads();
      if (threadIdx.x < kRadix) {
        unsigned c = 0;
#pragma unroll
        for (int hi4 = 0; hi4 < 16; ++hi4) c += s_chunks[0][hi4 * 16 + threadIdx.x];
        counts0[static_cast<size_t>(threadIdx.x) * rs_row_stride(num_chunks) + chunk0] = c;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  unsigned *copy = totals + static_cast<size_t>(blockIdx.x % kRsTotalCopies) * kRsMaxPasses * kRsMaxRadix;
  if (BITS == 8) {
    for (int i = threadIdx.x; i < kBytes * kBins; i += kRsThreads) {
      const unsigned c = byte_total(i / kBins, i % kBins);
      if (c) atomicAdd(&copy[(i / kBins) * kRsMaxRadix + (i % kBins)], c);
    }
  } else if (threadIdx.x < kBytes * 2 * kRadix) {  // 8 passes x 16 digits: pass 2q = low nibble of byte q, 2q+1 = high
    const unsigned pass = threadIdx.x / kRadix, d = threadIdx.x % kRadix, byte = pass / 2;
    unsigned c = 0;
#pragma unroll
    for (int o = 0; o < 16; ++o) c += byte_total(byte, (pass & 1u) ? d * 16 + o : o * 16 + d);
    if (c) atomicAdd(&copy[pass * kRsMaxRadix + d], c);
  }
}

template <int BITS>
__global__ __launch_bounds__(kRsThreads) void rs_plan_kernel(size_t n, RsHeader *hdr,
                                                             const unsigned *__restrict__ totals,
                                                             unsigned *__restrict__ bases) {
  constexpr int kRadix = 1 << BITS;
  constexpr int kPasses = 32 / BITS;
  __shared__ unsigned s_wsum[kRsWaves];
  __shared__ unsigned s_skip[kPasses];
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  if (tid < kPasses) s_skip[tid] = 0;
  __syncthreads();
  for (int p = 0; p < kPasses; ++p) {
    unsigned t = 0;
    if (tid < kRadix)
#pragma unroll
      for (int c = 0; c < kRsTotalCopies; ++c) t += totals[(c * kRsMaxPasses + p) * kRsMaxRadix + tid];
    if (t == n) s_skip[p] = 1;
    const unsigned incl = wave_inclusive_scan(t);
    if (lane == kWave - 1) s_wsum[wave] = incl;
    __syncthreads();
    unsigned off = 0;
    for (unsigned w = 0; w < wave; ++w) off += s_wsum[w];
    if (tid < kRadix) bases[p * kRsMaxRadix + tid] = off + incl - t;
    __syncthreads();
  }
  if (tid == 0) {
    unsigned cur = 0;
    for (int p = 0; p < kPasses; ++p) {
      hdr->pass[p].skip = s_skip[p];
      hdr->pass[p].src_is_tmp = cur;
      if (!s_skip[p]) cur ^= 1u;
    }
    hdr->final_in_tmp = cur;
  }
}

// ---- per pass, kernel 1: digit counts of every chunk -------------------------------------------------
template <int BITS>
__global__ __launch_bounds__(kRsThreads) void rs_chunk_hist_kernel(const unsigned *keys, const unsigned *tmp,
                                                                   size_t n, int pass, unsigned xor_mask,
                                                                   const RsHeader *hdr, unsigned *counts,
                                                                   size_t tiles_per_chunk, size_t num_chunks) {
  // A workgroup counts kRsHistCpw CONSECUTIVE chunks and writes, per digit, their counts side by side (one 16-byte
  // store per digit when the row allows it): with one chunk per workgroup every count was a 4-byte store into a
  // line of its own — 512 K partial-line writes per pass at 2^24 keys, 16 MiB written back for a 2 MiB matrix.
  constexpr int kRadix = 1 << BITS;
  constexpr int kRsHistCpw = rs_hist_cpw<BITS>();
  // kCopies histograms per chunk, lane l counts in copy l % kCopies (rs_hist_copies): keys that crowd into a few
  // digits no longer queue on one LDS word
  constexpr int kCopies = rs_hist_copies<BITS>(), kStride = kRadix + 1;
  __shared__ unsigned s_hist[kRsHistCpw][kCopies * kStride];
  const RsPass plan = hdr->pass[pass];
  if (plan.skip) return;
  const unsigned *__restrict__ src = plan.src_is_tmp ? tmp : keys;
  const int shift = pass * BITS;
  const size_t chunk0 = static_cast<size_t>(blockIdx.x) * kRsHistCpw;
  for (int i = threadIdx.x; i < kRsHistCpw * kCopies * kStride; i += kRsThreads) (&s_hist[0][0])[i] = 0;
  __syncthreads();
  const unsigned my_copy = (threadIdx.x & (kCopies - 1)) * kStride;
#pragma unroll 1
  for (int c = 0; c < kRsHistCpw; ++c) {
    const size_t chunk = chunk0 + c;
    if (chunk >= num_chunks) break;
    unsigned *hist = s_hist[c] + my_copy;
    const size_t lo = chunk * tiles_per_chunk * kRsTile;
    size_t hi = lo + tiles_per_chunk * kRsTile;
    hi = hi < n ? hi : n;
    // chunk starts are multiples of the tile size: 16-byte loads are aligned
    const size_t n4 = (hi - lo) / 4;
    const u32x4 *k4 = reinterpret_cast<const u32x4 *>(src + lo);
    // (issuing a tile's four 16-byte loads per lane before the first LDS atomic, non-temporal, measured no faster)
    for (size_t i = threadIdx.x; i < n4; i += kRsThreads) {
      const u32x4 v = k4[i];
      atomicAdd(&hist[((v.x ^ xor_mask) >> shift) & (kRadix - 1)], 1u);
      atomicAdd(&hist[((v.y ^ xor_mask) >> shift) & (kRadix - 1)], 1u);
      atomicAdd(&hist[((v.z ^ xor_mask) >> shift) & (kRadix - 1)], 1u);
      atomicAdd(&hist[((v.w ^ xor_mask) >> shift) & (kRadix - 1)], 1u);
    }
    for (size_t i = lo + n4 * 4 + threadIdx.x; i < hi; i += kRsThreads)
      atomicAdd(&hist[((src[i] ^ xor_mask) >> shift) & (kRadix - 1)], 1u);
  }
  __syncthreads();
  auto count_of = [&](int c, int d) {
    unsigned sum = 0;
#pragma unroll
    for (int k = 0; k < kCopies; ++k) sum += s_hist[c][k * kStride + d];
    return sum;
  };
  const bool vec = kRsHistCpw == 4;  // 16-byte aligned row pieces (rows are padded to four chunks)
  for (int d = threadIdx.x; d < kRadix; d += kRsThreads) {
    unsigned *row = counts + static_cast<size_t>(d) * rs_row_stride(num_chunks) + chunk0;
    if (vec) {
      *reinterpret_cast<u32x4 *>(row) = u32x4{count_of(0, d), count_of(1 % kRsHistCpw, d), count_of(2 % kRsHistCpw, d), count_of(3 % kRsHistCpw, d)};
    } else {
      for (int c = 0; c < kRsHistCpw && chunk0 + c < num_chunks; ++c) row[c] = count_of(c, d);
    }
  }
}

// ---- per pass, kernel 2: counts[d][*] -> global start of digit d in every chunk ------------------------
template <int BITS>
__global__ __launch_bounds__(kRsThreads) void rs_chunk_scan_kernel(int pass, const RsHeader *hdr,
                                                                   const unsigned *__restrict__ bases,
                                                                   unsigned *counts, size_t num_chunks) {
  // one workgroup per digit, ONE sweep: every thread takes `per` consecutive chunk counts (<= 8: at most 4096 chunks),
  // wave scan of the thread sums, wave sums through LDS (a loop of 512-chunk rounds with three barriers each took 5 us
  // for 2048 chunks, most of it barrier and LDS latency)
  constexpr unsigned kMaxPer = 8;
  static_assert(kRsTargetChunks * 3 / 2 <= static_cast<size_t>(kMaxPer) * kRsThreads, "chunks per scan workgroup (rs_geometry: at most 1.42 x the target)");
  __shared__ unsigned s_wsum[kRsWaves];
  if (hdr->pass[pass].skip) return;
  const unsigned d = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  unsigned *row = counts + static_cast<size_t>(d) * rs_row_stride(num_chunks);
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
  unsigned run = bases[pass * kRsMaxRadix + d] + incl - mine;
  for (unsigned w = 0; w < wave; ++w) run += s_wsum[w];
#pragma unroll
  for (unsigned j = 0; j < kMaxPer; ++j) {
    if (j < per && first + j < num_chunks) row[first + j] = run;
    run += v[j];
  }
}

// ---- ranking by LDS atomics: the property it rests on, and how it is watched -------------------------------------------
// rank = atomicAdd(&count[wave][digit], 1) replaces the BITS ballots of match_digit (8-bit digits: ~55 of the scatter's
// ~100 vector instructions per 64 keys) by one ds_add_rtn_u32.  An LSD sort needs a STABLE rank.  Between instructions
// the LDS keeps a wave's operations in issue order; INSIDE one instruction the rank is stable iff lanes that hit the
// same counter get their return values in ascending lane order.  gfx950 does that (the LDS serialises the lanes of a
// conflicting access lowest lane first) but the ISA manual does not promise it.  Three guards: (1) the atomic ranking
// is the default only on an allow-listed architecture (gfx950); (2) EVERY tile of EVERY call checks the invariant the
// ranking exists for — after k stable passes a tile re-ordered by digit k is sorted by its low (k+1) digits, so thread p
// compares its key with its left neighbour's (over DPP) under that mask while it writes the tile out; a violation sets
// DBHIP_DEV_RANK_ORDER in the workspace's status word (it also catches a damaged earlier pass); measured cost 0.2-0.9 us
// of a 28 us pass; (3) dbhip_radix_sort_prepare() runs the kernel below — every lane compares what the atomic returned
// with the count the ballots predict, over dense, sparse, skewed and partially masked digit patterns on every CU — and
// pins the ranking to what it saw.  The sort entry points themselves never synchronise (round 2 ran the self-test inside
// the first sort: a D2H copy and a stream synchronisation in a call documented as asynchronous, and a captured first sort
// fell back to ballots).  DBHIP_RS_RANK=ballot|atomic overrides.
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

// Stable rank of a wave's kRsKpt rows of keys among the wave's keys of the same digit; counts[] = the wave's digit
// counters in LDS (zero on entry, the wave's digit counts on return).  ARANK: by returning LDS atomics — unless the
// wave's keys crowd into few digits: the lanes of one atomic that hit the same counter are served one after the other
// (two distinct key values: the scatter 28.6 -> 70 us per pass at 2^24 keys), the ballots cost the same whatever the
// keys are (35 us).  The wave looks at two of its rows: if the first lane's digit is shared by kCrowd lanes or more, this
// tile's 1024 keys are ranked by ballots.  The counters are the wave's own, so the choice is the wave's own too.
template <int BITS, bool CHECK_VALID, bool ARANK>
__device__ __forceinline__ void rs_rank_rows(const unsigned (&key)[kRsKpt], unsigned (&rank)[kRsKpt], unsigned *counts,
                                             unsigned wave_first, unsigned valid_keys, int shift, unsigned xor_mask) {
  constexpr int kRadix = 1 << BITS;
  constexpr int kCrowd = BITS == 8 ? 8 : 16;  // uniform 4-bit digits put 4 lanes on a counter, uniform 8-bit digits 0.25
  bool atomics = ARANK;
  if (ARANK) {
#pragma unroll
    for (int j = 0; j < DBHIP_RS_CROWD_ROWS; ++j) {  // the rows whose loads come back first
      const unsigned d = ((key[j] ^ xor_mask) >> shift) & (kRadix - 1);
      const unsigned first = __builtin_amdgcn_readfirstlane(d);
      if (__builtin_popcountll(__ballot(d == first)) >= kCrowd) atomics = false;
    }
  }
  if (atomics) {
#pragma unroll
    for (int j = 0; j < kRsKpt; ++j) {
      const bool valid = !CHECK_VALID || wave_first + j * kWave < valid_keys;
      const unsigned d = ((key[j] ^ xor_mask) >> shift) & (kRadix - 1);
      rank[j] = valid ? atomicAdd(&counts[d], 1u) : 0u;  // see rank_by_lds_atomics() for why this is a stable rank
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < kRsKpt; ++j) {
    const bool valid = !CHECK_VALID || wave_first + j * kWave < valid_keys;
    const unsigned d = ((key[j] ^ xor_mask) >> shift) & (kRadix - 1);
    LaneMask m = match_digit<BITS>(d);
    if (CHECK_VALID) {
      const unsigned long long v = __ballot(valid);
      m.lo &= static_cast<unsigned>(v);
      m.hi &= static_cast<unsigned>(v >> 32);
    }
    const unsigned prior = lanes_before(m);
    const unsigned c = counts[d];  // same address inside a digit group: LDS broadcast
    rank[j] = c + prior;
    if (valid && prior == 0) counts[d] = c + lanes_in(m);  // group leader
  }
}

// ---- per pass, kernel 3: stable scatter of every chunk -------------------------------------------------
// One tile of the scatter: stable rank inside each wave, digit offsets across waves, re-order through LDS, write out
// in digit order.  FULL = the tile holds kRsTile keys: no per-key bounds checks (the kernel is VALU-bound — about 100
// vector instructions per 64 keys, 80 % of the issue slots at 2^24 keys by the SQ counters).
template <int BITS, bool FULL, bool ARANK>
__device__ __forceinline__ unsigned rs_scatter_tile(const unsigned *__restrict__ src, unsigned *__restrict__ dst,
                                                size_t tile_base, unsigned valid_in_tile, int shift, unsigned xor_mask,
                                                bool inject, unsigned &running, unsigned (*s_cnt)[1 << BITS], unsigned *s_dexcl,
                                                unsigned *s_goff, unsigned *s_wsum, unsigned *s_keys) {
  constexpr int kRadix = 1 << BITS;
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const unsigned wave_first = wave * kRsWaveKeys + lane;
  unsigned key[kRsKpt];
#pragma unroll
  for (int j = 0; j < kRsKpt; ++j) {
    const unsigned idx = wave_first + j * kWave;
    key[j] = (FULL || idx < valid_in_tile) ? src[tile_base + idx] : 0xFFFFFFFFu;
  }
  for (int i = tid; i < kRsWaves * kRadix; i += kRsThreads) (&s_cnt[0][0])[i] = 0;
  __syncthreads();

  // ---- stable rank of every key among the keys of its wave with the same digit
  unsigned rank[kRsKpt];
  rs_rank_rows<BITS, !FULL, ARANK>(key, rank, s_cnt[wave], wave_first, valid_in_tile, shift, xor_mask);
  __syncthreads();

  // ---- digit owners: counts across waves -> wave-exclusive offsets, tile totals
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
  if (tid < kRadix) {
    s_dexcl[tid] = dexcl;
    s_goff[tid] = running - dexcl;
    running += tile_count;
  }
  __syncthreads();

  // ---- re-order the tile by digit in LDS
#pragma unroll
  for (int j = 0; j < kRsKpt; ++j) {
    if (FULL || wave_first + j * kWave < valid_in_tile) {
      const unsigned d = ((key[j] ^ xor_mask) >> shift) & (kRadix - 1);
      s_keys[s_dexcl[d] + s_cnt[wave][d] + rank[j]] = key[j];
    }
  }
  __syncthreads();

  const unsigned low_mask = shift + BITS >= 32 ? 0xFFFFFFFFu : (1u << (shift + BITS)) - 1u;
  if (inject) {  // test hook (uniform): swap the first neighbours that share the digit and differ below it
    if (tid == 0)
      for (unsigned p = 0; p + 1 < valid_in_tile && p < 4096; ++p) {
        const unsigned a = s_keys[p] ^ xor_mask, b = s_keys[p + 1] ^ xor_mask;
        if ((((a ^ b) >> shift) & (kRadix - 1)) == 0 && ((a ^ b) & low_mask) != 0) {
          const unsigned t = s_keys[p];
          s_keys[p] = s_keys[p + 1];
          s_keys[p + 1] = t;
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
      const unsigned kk = s_keys[p];
      const unsigned kx = kk ^ xor_mask;
      // the tile as it now lies in LDS must be sorted by its low (shift + BITS) bits (stable ranking, here and in every
      // earlier pass): my left neighbour's may not exceed mine.  The neighbour's key comes over DPP (wave_shr:1; lane 0
      // keeps the 0: the pairs across the 64-key rows, one in 64, go unchecked).
      const unsigned mine = kx & low_mask;
      const unsigned left = __builtin_amdgcn_update_dpp(0u, mine, 0x138, 0xf, 0xf, false);
      bad |= left > mine ? 1u : 0u;
      const unsigned d = (kx >> shift) & (kRadix - 1);
      dst[s_goff[d] + p] = kk;
    }
  }
  // (Round 4 measured what comparing the pairs the DPP compare cannot see — last key of a 64-key row against the first of
  //  the next, one in 64 — would cost: as a read by lane 0 inside the loop above 10 % of the sort (200 -> 221 us for 2^24
  //  keys: a divergent LDS read in front of every row's stores), as one thread per row boundary behind the loop 1.2-1.5 %
  //  (201.4 -> 203.9 us, 4-bit 369 -> 375).  Not taken: this is a tripwire for an unstable rank, which shows inside the
  //  rows of a wave as surely as between them; include/dbhip.h says what it covers.)
  __syncthreads();
  return bad;
}

template <int BITS, bool ARANK>
__global__ __launch_bounds__(kRsThreads, DBHIP_RS_WPE) void rs_chunk_scatter_kernel(unsigned *keys, unsigned *tmp, size_t n,
                                                                         int pass, unsigned xor_mask,
                                                                         RsHeader *hdr,
                                                                         const unsigned *__restrict__ offsets,
                                                                         const unsigned *__restrict__ bases,
                                                                         unsigned inject, size_t tiles_per_chunk,
                                                                         size_t num_chunks) {
  // bases != nullptr: `offsets` still holds the raw per-chunk counts and this kernel sums its own prefix (few
  // chunks: the separate scan kernel would only add a dependent launch, ~5 us each at small sizes)
  constexpr int kRadix = 1 << BITS;
  __shared__ unsigned s_cnt[kRsWaves][kRadix];  // per-wave digit counts, then wave-exclusive offsets
  __shared__ unsigned s_dexcl[kRadix];          // tile-local exclusive offset of each digit
  __shared__ unsigned s_goff[kRadix];           // global offset of a digit minus its local offset
  __shared__ unsigned s_wsum[kRsWaves];
  __shared__ unsigned s_keys[kRsTile];

  const RsPass plan = hdr->pass[pass];
  if (plan.skip) return;  // uniform over the grid
  const unsigned *__restrict__ src = plan.src_is_tmp ? tmp : keys;
  unsigned *__restrict__ dst = plan.src_is_tmp ? keys : tmp;
  const int shift = pass * BITS;

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
  unsigned running = 0;
  if (tid < kRadix) {
    if (bases) {
      running = bases[pass * kRsMaxRadix + tid];
      const unsigned *row = offsets + static_cast<size_t>(tid) * rs_row_stride(num_chunks);
      for (size_t c = 0; c < chunk; ++c) running += row[c];
    } else {
      running = offsets[static_cast<size_t>(tid) * rs_row_stride(num_chunks) + chunk];
    }
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
    if (valid_in_tile == kRsTile)  // every tile but the input's last one
      bad |= rs_scatter_tile<BITS, true, ARANK>(src, dst, tile_base, valid_in_tile, shift, xor_mask, inject != 0 && tile == 0,
                                                running, s_cnt, s_dexcl, s_goff, s_wsum, s_keys);
    else
      bad |= rs_scatter_tile<BITS, false, ARANK>(src, dst, tile_base, valid_in_tile, shift, xor_mask, inject != 0 && tile == 0,
                                                 running, s_cnt, s_dexcl, s_goff, s_wsum, s_keys);
  }
  if (bad) atomicOr(&hdr->status, DBHIP_DEV_RANK_ORDER);
}

// ---- n <= one tile: the whole sort in ONE workgroup and one launch ------------------------------------------
// (the reference's small sweeps and dwarf tests run 128..65536 keys, where the ~16 launches of the general path
// are all that is measured).  Keys stay in registers between passes; every pass ranks them exactly as the
// scatter kernel does and re-orders them through LDS; passes whose digit is constant over the input are skipped
// on a workgroup-uniform vote; the result is written back to `keys`.
template <int BITS, bool ARANK>
__global__ __launch_bounds__(kRsThreads, DBHIP_RS_WPE) void rs_single_tile_kernel(unsigned *keys, unsigned n,
                                                                                  unsigned xor_mask, unsigned *status,
                                                                                  unsigned inject) {
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
    rs_rank_rows<BITS, true, ARANK>(key, rank, s_cnt[wave], wave_first, n, shift, xor_mask);
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
    if (inject) {  // test hook (uniform), as in rs_scatter_tile: swap the first neighbours that share the digit and differ below it
      if (tid == 0)
        for (unsigned p = 0; p + 1 < n; ++p) {
          const unsigned a = s_keys[p] ^ xor_mask, b = s_keys[p + 1] ^ xor_mask;
          if ((((a ^ b) >> shift) & (kRadix - 1)) == 0 && ((a ^ b) & low_mask) != 0) {
            const unsigned t = s_keys[p];
            s_keys[p] = s_keys[p + 1];
            s_keys[p + 1] = t;
            break;
          }
        }
      __syncthreads();
    }
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

__global__ __launch_bounds__(kRsThreads) void rs_finalize_kernel(unsigned *__restrict__ keys,
                                                                 const unsigned *__restrict__ tmp,
                                                                 size_t n, const RsHeader *hdr) {
  if (!hdr->final_in_tmp) return;
  const size_t stride = static_cast<size_t>(gridDim.x) * kRsThreads;
  const size_t n4 = n / 4;
  const u32x4 *s4 = reinterpret_cast<const u32x4 *>(tmp);
  u32x4 *d4 = reinterpret_cast<u32x4 *>(keys);
  for (size_t i = static_cast<size_t>(blockIdx.x) * kRsThreads + threadIdx.x; i < n4; i += stride)
    d4[i] = s4[i];
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) keys[n4 * 4 + threadIdx.x] = tmp[n4 * 4 + threadIdx.x];
}

template <int BITS>
int radix_sort_impl(unsigned *keys, unsigned *tmp, size_t n, unsigned xor_mask, void *workspace,
                    hipStream_t s, const DeviceInfo &dev) {
  constexpr int kPasses = 32 / BITS;
  constexpr int kRadix = 1 << BITS;
  const RsGeometry g = rs_geometry(n, BITS);
  char *base = static_cast<char *>(workspace);
  RsHeader *hdr = reinterpret_cast<RsHeader *>(base);
  unsigned *totals = reinterpret_cast<unsigned *>(base + kRsTotalsOff);
  unsigned *bases = reinterpret_cast<unsigned *>(base + kRsBasesOff);
  unsigned *counts = reinterpret_cast<unsigned *>(base + kRsCountsOff);

  const bool arank = rank_by_lds_atomics();
  const unsigned inject = rank_fault_injection();
  if (n <= static_cast<size_t>(kRsTile)) {  // one tile: one workgroup, one launch (+ the status word)
    const hipError_t e0 = fill_async(workspace, 0, kWsHeader, s);
    if (e0 != hipSuccess) return static_cast<int>(e0);
    if (arank)
      hipLaunchKernelGGL((rs_single_tile_kernel<BITS, true>), dim3(1), dim3(kRsThreads), 0, s, keys,
                         static_cast<unsigned>(n), xor_mask, &hdr->status, rank_fault_injection());
    else
      hipLaunchKernelGGL((rs_single_tile_kernel<BITS, false>), dim3(1), dim3(kRsThreads), 0, s, keys,
                         static_cast<unsigned>(n), xor_mask, &hdr->status, rank_fault_injection());
    return launch_status();
  }
  hipError_t e = fill_async(workspace, 0, kRsCountsOff, s);  // header + totals (+ bases)
  if (e != hipSuccess) return static_cast<int>(e);

  const size_t want = (n / 4 + kRsThreads - 1) / kRsThreads;
  const size_t cap = static_cast<size_t>(dev.cus) * 4;
  const unsigned hgrid = static_cast<unsigned>(want < cap ? (want ? want : 1) : cap);
  const size_t hist_groups = (g.chunks + rs_hist_cpw<BITS>() - 1) / rs_hist_cpw<BITS>();
  const unsigned hist_grid = static_cast<unsigned>(hist_groups < cap ? hist_groups : cap);
  hipLaunchKernelGGL((rs_histogram_kernel<BITS>), dim3(hist_grid), dim3(kRsThreads), 0, s, keys, n,
                     xor_mask, totals, counts, g.tiles_per_chunk, g.chunks);
  hipLaunchKernelGGL((rs_plan_kernel<BITS>), dim3(1), dim3(kRsThreads), 0, s, n, hdr, totals, bases);
  const unsigned cgrid = static_cast<unsigned>(g.chunks);
  const bool fused_scan = g.chunks <= kRsFusedScanChunks;
  for (int p = 0; p < kPasses; ++p) {
    if (p > 0)  // pass 0's chunk counts came with the up-front histogram
      hipLaunchKernelGGL((rs_chunk_hist_kernel<BITS>), dim3((cgrid + rs_hist_cpw<BITS>() - 1) / rs_hist_cpw<BITS>()), dim3(kRsThreads), 0, s, keys, tmp, n, p, xor_mask,
                         hdr, counts, g.tiles_per_chunk, g.chunks);
    if (!fused_scan)
      hipLaunchKernelGGL((rs_chunk_scan_kernel<BITS>), dim3(kRadix), dim3(kRsThreads), 0, s, p, hdr, bases, counts,
                         g.chunks);
    const unsigned *fused_bases = fused_scan ? bases : static_cast<const unsigned *>(nullptr);
    if (arank)
      hipLaunchKernelGGL((rs_chunk_scatter_kernel<BITS, true>), dim3((cgrid + 7) / 8 * 8), dim3(kRsThreads), 0, s, keys,
                         tmp, n, p, xor_mask, hdr, counts, fused_bases, inject, g.tiles_per_chunk, g.chunks);
    else
      hipLaunchKernelGGL((rs_chunk_scatter_kernel<BITS, false>), dim3((cgrid + 7) / 8 * 8), dim3(kRsThreads), 0, s, keys,
                         tmp, n, p, xor_mask, hdr, counts, fused_bases, inject, g.tiles_per_chunk, g.chunks);
  }
  hipLaunchKernelGGL(rs_finalize_kernel, dim3(hgrid), dim3(kRsThreads), 0, s, keys, tmp, n, hdr);
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
  const size_t radix = static_cast<size_t>(1) << radix_bits;
  return align_up(kRsCountsOff + sizeof(unsigned) * radix * rs_row_stride(g.chunks), kWsAlign);
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
