// scan.hip — dwarf 1: stable stream compaction out = [x in src : x < filter] for gfx950.
//
// Replaces the reference kernel simple_two_pass_scan (scan/scan.cl:3-42: chunked count -> serial
// prefix by work-item 0 -> chunked write, which reads src twice with uncoalesced per-lane chunks)
// and the oneDPL copy_if behind DPLScan (dpl_wrapper.hpp:27-33).  Same logical result
// (scan/scan.cpp:12-17 expected_out_lt), different machine mapping:
//
//   * ONE pass over src.  Every wave reads 2048 contiguous elements of a tile with eight
//     16-byte-per-lane non-temporal loads (each wave instruction = 1 KiB contiguous), keeps them in
//     registers, counts, and writes its matches at an offset inside the chunk's staging slot.
//   * In-wave ranks come from the compare masks themselves: v_cmp -> 64-bit ballot in SGPRs,
//     s_bcnt1 for totals, v_mbcnt_lo/hi for the lane-exclusive prefix.
//   * dbhip_copy_if_lt_i32: no workgroup ever waits on another one (see below): no tickets, no look-back, no
//     spin, nothing to time out; every call is two launches whatever the size.
//   * dbhip_copy_if_lt_dense_i32 (dense predicates, second half of this file): one launch, matches written once at
//     their final positions; a chunk waits for the match count in front of it (ticketed chunks, bounded wait).
//
// Algorithmic HBM bytes: 4*n read + 4*out_size written.
#include <climits>

#include "dbhip_common.hpp"
#include "handoff.hpp"

namespace dbhip {
namespace {

constexpr int kScanVpt = 8;                           // 16-byte loads per lane per tile
constexpr int kScanWaveElems = kWave * kScanVpt * 4;  // 2048 contiguous elements per wave

struct ScanWs {
  unsigned status;  // DBHIP_DEV_* bits (the two-launch path has no device-side failure; the dense path can time out)
  unsigned pad0;
  unsigned long long ticket;  // dense path: next chunk to hand out
  unsigned pad[60];
};
static_assert(sizeof(ScanWs) == kWsHeader, "workspace header size");

template <int WAVES, bool kAligned, bool kNontemporal>
__device__ __forceinline__ void load_wave_tile(i32x4 (&v)[kScanVpt], const int *__restrict__ src,
                                               size_t n, size_t tile, unsigned wave, unsigned lane) {
  constexpr size_t kTile = static_cast<size_t>(kScanWaveElems) * WAVES;
  const size_t wave_base = tile * kTile + static_cast<size_t>(wave) * kScanWaveElems;
  if (kAligned && tile * kTile + kTile <= n) {
    const i32x4 *p = reinterpret_cast<const i32x4 *>(src + wave_base) + lane;
#pragma unroll
    for (int k = 0; k < kScanVpt; ++k)
      v[k] = kNontemporal ? __builtin_nontemporal_load(p + k * kWave) : p[k * kWave];
  } else {
#pragma unroll
    for (int k = 0; k < kScanVpt; ++k) {
      const size_t e = wave_base + (static_cast<size_t>(k) * kWave + lane) * 4;
      v[k].x = e + 0 < n ? src[e + 0] : INT_MAX;  // INT_MAX never satisfies x < filter
      v[k].y = e + 1 < n ? src[e + 1] : INT_MAX;
      v[k].z = e + 2 < n ? src[e + 2] : INT_MAX;
      v[k].w = e + 3 < n ? src[e + 3] : INT_MAX;
    }
  }
}

// matches of one wave: the compare IS the ballot (v_cmp -> SGPR pair), totals are scalar popcounts
__device__ __forceinline__ unsigned count_wave_matches(const i32x4 (&v)[kScanVpt], int filter) {
  unsigned total = 0;
#pragma unroll
  for (int k = 0; k < kScanVpt; ++k) {
    total += __builtin_popcountll(__ballot(v[k].x < filter));
    total += __builtin_popcountll(__ballot(v[k].y < filter));
    total += __builtin_popcountll(__ballot(v[k].z < filter));
    total += __builtin_popcountll(__ballot(v[k].w < filter));
  }
  return total;
}

// writes the wave's matches, in source order, to dst_wave[0 .. wave_total)
// (dst_wave may point to global memory or to the LDS staging buffer)
__device__ __forceinline__ void emit_wave_matches(const i32x4 (&v)[kScanVpt], int filter,
                                                  int *dst_wave) {
  unsigned row_base = 0;
#pragma unroll
  for (int k = 0; k < kScanVpt; ++k) {
    const bool m0 = v[k].x < filter, m1 = v[k].y < filter, m2 = v[k].z < filter, m3 = v[k].w < filter;
    const unsigned long long b0 = __ballot(m0), b1 = __ballot(m1), b2 = __ballot(m2), b3 = __ballot(m3);
    if ((b0 | b1 | b2 | b3) == 0) continue;  // scalar branch: most rows are empty at low selectivity
    int *dst = dst_wave + row_base + mbcnt(b0) + mbcnt(b1) + mbcnt(b2) + mbcnt(b3);
    if (m0) *dst++ = v[k].x;
    if (m1) *dst++ = v[k].y;
    if (m2) *dst++ = v[k].z;
    if (m3) *dst++ = v[k].w;
    row_base += __builtin_popcountll(b0) + __builtin_popcountll(b1) + __builtin_popcountll(b2) +
                __builtin_popcountll(b3);
  }
}

// ---- emission through LDS (dense path) --------------------------------------------------------------------------
// Straight from the registers a tile costs up to 32 dword stores per lane whose lanes land one match-count apart: at
// dense selectivities every instruction touches ~16 partial lines and the store path, not HBM, sets the pace (s = 1:
// 2.8 TB/s of writes).  The dense path packs the matches of four rows (4 KiB per wave) into a wave-private LDS strip
// at the destination's alignment (strip index = destination index mod 4); every full 16-byte group then leaves with
// one aligned x4 store (1 KiB contiguous per wave instruction), the at most 3 + 3 edge elements go one by one.
// Packing is branch-free: a lane without a match in a component writes that component to its own trash word (one
// ds_write per component for the whole wave, no exec juggling) — the masked form costs ~100 instructions per row,
// as long as the row's share of the stream; this one ~45.  (Straight to global memory the masked form stays the
// right one: a dummy store would be HBM traffic.)
constexpr int kStripRows = 4;
constexpr int kStripTrash = kWave * kStripRows * 4 + 8;  // first of the 64 per-lane trash words
constexpr int kStageStride = kStripTrash + kWave;        // ints per wave strip; a multiple of 4: strips stay 16-byte aligned

__device__ __forceinline__ unsigned mbcnt_acc(unsigned long long mask, unsigned acc) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), acc));
}

// matches of rows [K0, K1) of v, in source order, to strip_rows[0 .. count); returns count
template <int K0, int K1, int N>
__device__ __forceinline__ unsigned emit_wave_rows_lds(const i32x4 (&v)[N], int filter, int *strip_rows, int *trash) {
  unsigned row_base = 0;
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const bool m0 = v[k].x < filter, m1 = v[k].y < filter, m2 = v[k].z < filter, m3 = v[k].w < filter;
    const unsigned long long b0 = __ballot(m0), b1 = __ballot(m1), b2 = __ballot(m2), b3 = __ballot(m3);
    if ((b0 | b1 | b2 | b3) == 0) continue;  // scalar branch
    const unsigned a0 = mbcnt_acc(b3, mbcnt_acc(b2, mbcnt_acc(b1, mbcnt_acc(b0, row_base))));
    const unsigned a1 = a0 + (m0 ? 1u : 0u), a2 = a1 + (m1 ? 1u : 0u), a3 = a2 + (m2 ? 1u : 0u);
    *(m0 ? strip_rows + a0 : trash) = v[k].x;
    *(m1 ? strip_rows + a1 : trash) = v[k].y;
    *(m2 ? strip_rows + a2 : trash) = v[k].z;
    *(m3 ? strip_rows + a3 : trash) = v[k].w;
    row_base += __builtin_popcountll(b0) + __builtin_popcountll(b1) + __builtin_popcountll(b2) +
                __builtin_popcountll(b3);
  }
  return row_base;
}

// strip[shift .. shift + run) -> dst[0 .. run), where shift = (dst / 4 bytes) mod 4.  LDS executes a wave's operations
// in order, so write -> read -> overwrite needs no barrier beyond the compiler fences.
__device__ __forceinline__ void flush_strip(const int *strip, unsigned shift, unsigned run, int *dst, unsigned lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int *g0 = dst - shift;  // 16-byte aligned; strip[i] belongs at g0[i] for shift <= i < shift + run
  const unsigned end = shift + run, first_full = (shift + 3u) >> 2, last_full = end >> 2;
  for (unsigned g = first_full + lane; g < last_full; g += kWave)
    __builtin_nontemporal_store(*reinterpret_cast<const i32x4 *>(strip + 4 * g), reinterpret_cast<i32x4 *>(g0 + 4 * g));
  // edges: the part of group 0 in front of the first full group, and what follows the last full group (a run that
  // ends inside group 0 has no second edge)
  if (lane < 4u) {
    if (lane >= shift && lane < end && lane < 4u * first_full) g0[lane] = strip[lane];
  } else if (lane < 8u) {
    const unsigned i = 4u * last_full + lane - 4u;
    if (last_full >= first_full && i < end) g0[i] = strip[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the strip is rewritten next
  __builtin_amdgcn_wave_barrier();
}

// =================================================================================================
// Chunked path, two kernels, NO communication between workgroups while streaming.
//
// Why not a single-pass decoupled look-back (measured history in DESIGN.md 4.1): with all 256 CUs streaming,
// ~30 MiB of loads are in flight and any read that must come from memory — a look-back poll included — takes
// 3.5-4.5 us, as long as a CU needs for a whole 128 KiB tile; every variant built landed between 255 and 340 us
// at 2^28 against 160 us for the bare stream.  So the dependency is removed instead of hidden:
//   scan_chunk_kernel  chunks (1 MiB for large inputs) are dealt by blockIdx — nothing here waits on another
//                      workgroup, so no ticket and no zeroed header are needed: the call is two launches.
//                      A 16-wave workgroup streams its chunk tile by tile (128 KiB, double-buffered registers,
//                      next tile's loads issued right after the count, a barrier that does not drain vmcnt),
//                      ranks matches with ballots/mbcnt and writes them to the chunk's own slot of a staging
//                      buffer at chunk-local offsets; one count per chunk.
//   scan_move_kernel   one workgroup per chunk: prefix of the chunk counts (<= 4 KiB, read from L2),
//                      then a coalesced copy staging -> out at the global offset; writes out_size and the
//                      (always clean) status word.
// HBM bytes: 4n + 4*matches (stream) + 8*matches (move); at the reference's selectivity (4e-4) the
// move is ~0.1 % of the traffic.  Needs an n-element staging buffer in the workspace.
// =================================================================================================
constexpr int kChWaves = 16;
constexpr int kChThreads = kChWaves * kWave;
constexpr int kChTile = kChWaves * kScanWaveElems;  // 32768 elements = 128 KiB
constexpr int kChMaxTilesPerChunk = 8;  // 1 MiB chunks for large inputs; fewer tiles per chunk when
                                        // the input would otherwise give less than ~4 chunks per CU

struct ChunkLayout {
  size_t chunks, chunk_elems, counts_off, staging_off, total;
  unsigned tiles_per_chunk;
};
inline ChunkLayout chunk_layout(size_t n) {
  ChunkLayout L;
  const size_t tiles = (n + kChTile - 1) / kChTile;
  size_t tpc = tiles / 1024;  // aim at >= 1024 chunks (4 per CU on 256 CUs)
  tpc = tpc < 1 ? 1 : (tpc > kChMaxTilesPerChunk ? kChMaxTilesPerChunk : tpc);
  L.tiles_per_chunk = static_cast<unsigned>(tpc);
  L.chunk_elems = tpc * kChTile;
  L.chunks = (n + L.chunk_elems - 1) / L.chunk_elems;
  L.counts_off = kWsHeader;
  L.staging_off = align_up(L.counts_off + (L.chunks ? L.chunks : 1) * sizeof(unsigned), kWsAlign);
  L.total = align_up(L.staging_off + n * sizeof(int), kWsAlign);
  return L;
}

// one tile of the chunk: count `cur`, prefetch the next tile into `nxt`, emit matches
template <bool kAligned, bool kNontemporal>
__device__ __forceinline__ unsigned chunk_step(i32x4 (&cur)[kScanVpt], i32x4 (&nxt)[kScanVpt],
                                               const int *__restrict__ src, size_t n, int filter,
                                               int *__restrict__ dst_chunk, unsigned running,
                                               size_t tile, bool has_next, unsigned par,
                                               unsigned (*s_wave_total)[kChWaves], unsigned wave,
                                               unsigned lane) {
  const unsigned wave_total = count_wave_matches(cur, filter);
  if (lane == 0) s_wave_total[par][wave] = wave_total;
  // keep the CU streaming: the next tile's loads go out before anything else happens
  if (has_next) load_wave_tile<kChWaves, kAligned, kNontemporal>(nxt, src, n, tile + 1, wave, lane);
  wg_barrier_lds_only();  // NOT __syncthreads(): its vmcnt(0) would wait for the prefetch
  unsigned wave_excl = 0, tile_total = 0;
#pragma unroll
  for (int w = 0; w < kChWaves; ++w) {
    const unsigned t = s_wave_total[par][w];
    wave_excl += w < static_cast<int>(wave) ? t : 0u;
    tile_total += t;
  }
  if (wave_total != 0) emit_wave_matches(cur, filter, dst_chunk + running + wave_excl);
  return running + tile_total;
}

template <bool kAligned, bool kNontemporal>
__global__ __launch_bounds__(kChThreads) void scan_chunk_kernel(const int *__restrict__ src, size_t n,
                                                                int filter, int *__restrict__ staging,
                                                                unsigned *__restrict__ counts, size_t num_chunks,
                                                                unsigned tiles_per_chunk) {
  // two slots alternate with the tile parity: one barrier per tile is enough
  __shared__ unsigned s_wave_total[2][kChWaves];
  const unsigned lane = threadIdx.x & (kWave - 1);
  const unsigned wave = threadIdx.x / kWave;

  // Chunks are dealt by blockIdx, not by ticket: this kernel never waits on another workgroup, so residency does
  // not matter for progress, and without a ticket counter the call needs no zeroed header — one dependent launch
  // (the fill) less in front of the stream.
  for (size_t chunk = blockIdx.x; chunk < num_chunks; chunk += gridDim.x) {
    __syncthreads();  // the previous chunk's last tile may still be reading the s_wave_total slot tile 0 writes
    const size_t chunk_elems = static_cast<size_t>(tiles_per_chunk) * kChTile;
    const size_t first_tile = chunk * tiles_per_chunk;
    const size_t elems = n - chunk * chunk_elems < chunk_elems ? n - chunk * chunk_elems : chunk_elems;
    const unsigned tiles = static_cast<unsigned>((elems + kChTile - 1) / kChTile);
    int *dst_chunk = staging + chunk * chunk_elems;

    i32x4 a[kScanVpt], b[kScanVpt];
    load_wave_tile<kChWaves, kAligned, kNontemporal>(a, src, n, first_tile, wave, lane);
    unsigned running = 0;
    for (unsigned t = 0; t < tiles; t += 2) {
      running = chunk_step<kAligned, kNontemporal>(a, b, src, n, filter, dst_chunk, running, first_tile + t,
                                                   t + 1 < tiles, 0u, s_wave_total, wave, lane);
      if (t + 1 >= tiles) break;
      running = chunk_step<kAligned, kNontemporal>(b, a, src, n, filter, dst_chunk, running,
                                                   first_tile + t + 1, t + 2 < tiles, 1u, s_wave_total,
                                                   wave, lane);
    }
    if (threadIdx.x == 0) counts[chunk] = running;
  }
}

__global__ __launch_bounds__(256) void scan_move_kernel(const int *__restrict__ staging,
                                                        const unsigned *__restrict__ counts,
                                                        int *__restrict__ out,
                                                        unsigned long long *__restrict__ out_size,
                                                        size_t num_chunks, size_t chunk_elems, ScanWs *ws) {
  __shared__ unsigned long long s_part[256 / kWave];
  const size_t chunk = blockIdx.x;
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  unsigned long long mine = 0;
  for (size_t i = tid; i < chunk; i += 256) mine += counts[i];
  mine = wave_reduce_add_u64(mine);
  if (lane == 0) s_part[wave] = mine;
  __syncthreads();
  unsigned long long prefix = 0;
#pragma unroll
  for (int w = 0; w < 256 / kWave; ++w) prefix += s_part[w];
  const unsigned m = counts[chunk];
  const int *src = staging + chunk * chunk_elems;  // 16-byte aligned: chunk_elems is a multiple of the tile
  int *dst = out + prefix;
  // 16 bytes per lane: the destination decides the alignment (elements up to its next 16-byte boundary go one by
  // one), the source is then read at whatever alignment that leaves (4-byte aligned 16-byte loads are legal on
  // global memory); at dense selectivities this kernel moves as many bytes as the stream itself
  const unsigned head0 = ((16u - (static_cast<unsigned>(reinterpret_cast<uintptr_t>(dst)) & 15u)) & 15u) / 4u;
  const unsigned head = head0 < m ? head0 : m;
  if (tid < head) dst[tid] = src[tid];
  const unsigned body = (m - head) / 4;
  typedef int i32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
  for (unsigned v = tid; v < body; v += 256) {
    const i32x4 x = *reinterpret_cast<const i32x4_a4 *>(src + head + 4 * v);
    __builtin_nontemporal_store(x, reinterpret_cast<i32x4 *>(dst + head + 4 * v));
  }
  const unsigned tail0 = head + 4 * body;
  if (tid < 4 && tail0 + tid < m) dst[tail0 + tid] = src[tail0 + tid];
  if (chunk == num_chunks - 1 && tid == 0) {
    *out_size = prefix + m;
    ws->status = DBHIP_DEV_OK;  // nothing on this path can fail on the device; the header is not cleared up front
  }
}


// =================================================================================================
// Dense predicates (dbhip_copy_if_lt_dense_i32): final positions at FIRST write, most of the input read ONCE.
//
// The two-launch path above moves 12 bytes per match (staging write, staging read, final write) and is bound by
// exactly that once the predicate is dense (s = 0.5: ~500 us for 1.61 GB of algorithmic bytes).  Final positions at
// first write need the number of matches in front of a chunk before the chunk's matches leave the CU, i.e. a
// hand-off between workgroups, and a place for the chunk to wait meanwhile.  Here that place is the register file:
//   * an 8-wave workgroup (two per CU) takes a 128 KiB chunk by ticket; every wave owns 16 contiguous rows (1 KiB
//     each) of it.  The last 12 rows stay in registers (48 VGPRs) from the count to the emission; the first 4 rows
//     are counted, dropped and requested again right before the hand-off, so that this second read (on-die: the
//     lines were read microseconds ago) is what the wave has in flight while it waits.  Fabric bytes 5n + 4m.
//   * hand-off: chunk-granular decoupled look-back over 8-byte {state, value} granules (agent-scope relaxed
//     atomics), one wave, 64 granules per poll, every granule on a 128-byte line of its own.
//   * emission through the LDS strips above, 16-byte non-temporal stores at the final positions.
// Chunks go by TICKET, taken when the workgroup starts on the chunk: a chunk's predecessors are then always held by
// workgroups that are already running, so every wait terminates whatever else shares the GPU; waits are
// time-bounded anyway (2 s: DBHIP_DEV_SPIN_TIMEOUT).
// What was measured on the way (2^28 rows, s = 0.5 / s = 0.01, same box; DESIGN.md 4.1 has the table):
//   chunk read twice (count pass, look-back, emit pass; 512 KiB chunks)         392 / 318 us — bound by 8n + 4m on the
//                                                                               fabric: the second read is served on
//                                                                               die but costs its full fabric time
//   whole chunk in registers, 4-wave workgroups x 2 per CU, 192 KiB chunks      431 / 397 us — hops of 17-23 us
//   packed granules (eight per line): 16- / 8- / 4-wave workgroups              311-325 / 245-257, 322 / 255, 349 / 302 us
//   ... ticket of the next chunk requested one chunk ahead                      345 / 261 us — a reserved chunk that
//                                                                               nobody loads yet stalls its successors
//   ... 256 granules per poll                                                   342 / 274 us
//   (round 3, this kernel) next ticket taken right after the hand-off, before   s = 0.1: 260 us against 246, 0.5: 321
//   the emission, instead of at the top of the next chunk                       against 297 — the same stall, shorter
//   (round 3, this kernel) the four counted rows simply KEPT in their            s = 0.1: 243 against 247.5, 0.5: 311 against
//   registers (no second read at all: the registers are reloaded anyway)        299, 1.0: 378 against 373 — the second read is
//                                                                               not what the kernel waits for, and having it
//                                                                               in flight helps the emission-heavy cases
//   one granule per line: 16- / 8- (this kernel) / 4-wave workgroups            316-318 / 250, 300 / 239, 334 / 320 us
// Slower than the two-launch path for sparse predicates (hand-off per chunk, a quarter of the input read twice):
// callers choose — ops.CopyIfLt and the TwoPassScan dwarf switch on the selectivity of the previous call (> 0.1).
// =================================================================================================
constexpr int kKpStream = 4, kKpKeep = 12, kKpRows = kKpStream + kKpKeep;  // rows of a wave read twice / kept
constexpr int kKpWaveElems = kKpRows * kWave * 4;                           // contiguous elements per wave
constexpr int kKpWaves = 8;                                                 // two workgroups per CU
constexpr int kKpChunk = kKpWaves * kKpWaveElems;                           // 32768 elements = 128 KiB
static_assert(kKpStream % kStripRows == 0 && kKpKeep % kStripRows == 0, "rows are emitted four at a time");

template <int N, bool kFast, bool kNontemporal>
__device__ __forceinline__ void kp_load_rows(i32x4 (&v)[N], const int *__restrict__ src, size_t n, size_t first,
                                             unsigned lane) {
  if (kFast) {
    const i32x4 *p = reinterpret_cast<const i32x4 *>(src + first) + lane;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = kNontemporal ? __builtin_nontemporal_load(p + k * kWave) : p[k * kWave];
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const size_t e = first + (static_cast<size_t>(k) * kWave + lane) * 4;
      v[k].x = e + 0 < n ? src[e + 0] : INT_MAX;  // INT_MAX never satisfies x < filter
      v[k].y = e + 1 < n ? src[e + 1] : INT_MAX;
      v[k].z = e + 2 < n ? src[e + 2] : INT_MAX;
      v[k].w = e + 3 < n ? src[e + 3] : INT_MAX;
    }
  }
}
template <int N>
__device__ __forceinline__ unsigned kp_count_rows(const i32x4 (&v)[N], int filter) {
  unsigned total = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    total += __builtin_popcountll(__ballot(v[k].x < filter));
    total += __builtin_popcountll(__ballot(v[k].y < filter));
    total += __builtin_popcountll(__ballot(v[k].z < filter));
    total += __builtin_popcountll(__ballot(v[k].w < filter));
  }
  return total;
}
// rows [4G, 4G + 4) of v -> dst; returns the position behind them
template <int G, int N>
__device__ __forceinline__ int *kp_emit_group(const i32x4 (&v)[N], int filter, int *dst, int *strip, unsigned lane) {
  const unsigned shift = static_cast<unsigned>(reinterpret_cast<uintptr_t>(dst) >> 2) & 3u;
  const unsigned r = emit_wave_rows_lds<G * kStripRows, (G + 1) * kStripRows, N>(v, filter, strip + shift,
                                                                                  strip + kStripTrash + lane);
  if (r != 0) flush_strip(strip, shift, r, dst, lane);
  return dst + r;
}
template <int N>
__device__ __forceinline__ void kp_emit_rows(const i32x4 (&v)[N], int filter, int *dst, int *strip, unsigned lane) {
  static_assert(N == 4 || N == 8 || N == 12, "groups of four rows");
  dst = kp_emit_group<0, N>(v, filter, dst, strip, lane);
  if (N >= 8) dst = kp_emit_group<(N >= 8 ? 1 : 0), N>(v, filter, dst, strip, lane);
  if (N >= 12) dst = kp_emit_group<(N >= 12 ? 2 : 0), N>(v, filter, dst, strip, lane);
}

template <bool kAligned>
__global__ __launch_bounds__(kKpWaves * kWave) void scan_dense_kernel(const int *__restrict__ src, size_t n, int filter,
                                                                int *__restrict__ out,
                                                                unsigned long long *__restrict__ out_size, ScanWs *ws,
                                                                unsigned long long *granules, size_t num_chunks) {
  __shared__ unsigned s_cnt[kKpWaves];
  __shared__ unsigned long long s_chunk, s_excl;
  __shared__ __attribute__((aligned(16))) int s_strip[kKpWaves][kStageStride];
  const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  int *strip = s_strip[wave];
  while (true) {
    __syncthreads();  // s_chunk / s_cnt / s_excl of the previous chunk are no longer read
    if (threadIdx.x == 0) s_chunk = __hip_atomic_fetch_add(&ws->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const size_t chunk = s_chunk;
    if (chunk >= num_chunks) return;  // uniform
    const size_t first = chunk * kKpChunk + static_cast<size_t>(wave) * kKpWaveElems;
    const bool fast = kAligned && first + kKpWaveElems <= n;  // one decision for the wave's whole slice
    i32x4 st[kKpStream], kp[kKpKeep];
    unsigned c_st, c_kp;
    if (fast) {
      kp_load_rows<kKpStream, true, false>(st, src, n, first, lane);  // plain: read again below
      kp_load_rows<kKpKeep, true, true>(kp, src, n, first + kKpStream * kWave * 4, lane);
      c_st = kp_count_rows(st, filter);
      if (c_st != 0) kp_load_rows<kKpStream, true, true>(st, src, n, first, lane);  // in flight across the hand-off
      c_kp = kp_count_rows(kp, filter);
    } else {  // the last chunk, or a source that is not 16-byte aligned: bounds-checked, everything stays in registers
      kp_load_rows<kKpStream, false, false>(st, src, n, first, lane);
      kp_load_rows<kKpKeep, false, false>(kp, src, n, first + kKpStream * kWave * 4, lane);
      c_st = kp_count_rows(st, filter);
      c_kp = kp_count_rows(kp, filter);
    }
    if (lane == 0) s_cnt[wave] = c_st + c_kp;
    wg_barrier_lds_only();  // NOT __syncthreads(): its vmcnt(0) would wait for the second read
    unsigned wave_excl = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kKpWaves; ++w) {
      const unsigned c = s_cnt[w];
      wave_excl += w < static_cast<int>(wave) ? c : 0u;
      total += c;
    }
    if (wave == 0) {  // publish the aggregate, look back, publish the inclusive prefix
      const unsigned long long excl = chunk_handoff(granules, chunk, total, lane, &ws->status);
      if (lane == 0) {
        s_excl = excl;
        if (chunk == num_chunks - 1) *out_size = excl + total;
      }
    }
    wg_barrier_lds_only();
    int *dst = out + s_excl + wave_excl;
    if (c_kp != 0) kp_emit_rows(kp, filter, dst + c_st, strip, lane);  // the kept rows first: st may still be landing
    if (c_st != 0) kp_emit_rows(st, filter, dst, strip, lane);
  }
}

inline int env_int(const char *name, int lo, int hi, int dflt) {
  const char *e = getenv(name);
  if (!e) return dflt;
  const int x = atoi(e);
  return (x >= lo && x <= hi) ? x : dflt;
}
inline bool scan_nontemporal() {
  static const int v = env_int("DBHIP_SCAN_NT", 0, 1, 1);
  return v != 0;
}
template <bool kAligned, bool kNontemporal>
int launch_chunked(const int *src, size_t n, int filter, int *out, unsigned long long *osz, void *workspace,
                   const DeviceInfo &dev, hipStream_t s) {
  const ChunkLayout L = chunk_layout(n);
  char *base = static_cast<char *>(workspace);
  ScanWs *ws = reinterpret_cast<ScanWs *>(base);
  unsigned *counts = reinterpret_cast<unsigned *>(base + L.counts_off);
  int *staging = reinterpret_cast<int *>(base + L.staging_off);
  const size_t cap = static_cast<size_t>(dev.cus);  // one 16-wave workgroup per CU
  const unsigned grid = static_cast<unsigned>(L.chunks < cap ? L.chunks : cap);
  hipLaunchKernelGGL((scan_chunk_kernel<kAligned, kNontemporal>), dim3(grid), dim3(kChThreads), 0, s, src, n,
                     filter, staging, counts, L.chunks, L.tiles_per_chunk);
  hipLaunchKernelGGL(scan_move_kernel, dim3(static_cast<unsigned>(L.chunks)), dim3(256), 0, s, staging,
                     counts, out, osz, L.chunks, L.chunk_elems, ws);
  return launch_status();
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

extern "C" size_t dbhip_copy_if_lt_i32_workspace_bytes(size_t n) {
  return chunk_layout(n ? n : 1).total;  // header | counts[chunks] | staging[n]
}

extern "C" int dbhip_copy_if_lt_i32(const int32_t *src, size_t n, int32_t filter_value,
                                    int32_t *out, uint64_t *out_size, void *workspace,
                                    size_t workspace_bytes, dbhip_stream_t stream) {
  if (!out_size || (n && (!src || !out))) return DBHIP_EINVAL;
  if (n >= (1ull << 61)) return DBHIP_EINVAL;
  if (!ws_ok(workspace, workspace_bytes, dbhip_copy_if_lt_i32_workspace_bytes(n)))
    return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    hipError_t e = fill_async(workspace, 0, kWsHeader, s);
    if (e == hipSuccess) e = fill_async(out_size, 0, sizeof(uint64_t), s);
    return static_cast<int>(e);
  }
  unsigned long long *osz = reinterpret_cast<unsigned long long *>(out_size);
  const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
  if (!aligned) return launch_chunked<false, false>(src, n, filter_value, out, osz, workspace, dev, s);
  if (scan_nontemporal()) return launch_chunked<true, true>(src, n, filter_value, out, osz, workspace, dev, s);
  return launch_chunked<true, false>(src, n, filter_value, out, osz, workspace, dev, s);
}

// Dense predicates: one launch (+ the fill that zeroes ticket and granules), chunk-granular hand-off (see above).
// Same arguments, same results and the same workspace size as dbhip_copy_if_lt_i32.
extern "C" int dbhip_copy_if_lt_dense_i32(const int32_t *src, size_t n, int32_t filter_value, int32_t *out,
                                          uint64_t *out_size, void *workspace, size_t workspace_bytes,
                                          dbhip_stream_t stream) {
  if (!out_size || (n && (!src || !out))) return DBHIP_EINVAL;
  if (n >= (1ull << 61)) return DBHIP_EINVAL;
  if (!ws_ok(workspace, workspace_bytes, dbhip_copy_if_lt_i32_workspace_bytes(n))) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  const size_t chunks = (n + kKpChunk - 1) / kKpChunk;
  // granules live in the staging area of the two-launch path (n * 4 bytes: far more than 128 bytes per 128 KiB chunk)
  const ChunkLayout L = chunk_layout(n ? n : 1);
  char *base = static_cast<char *>(workspace);
  unsigned long long *granules = reinterpret_cast<unsigned long long *>(base + L.staging_off);
  if (n == 0) {
    hipError_t e0 = fill_async(workspace, 0, kWsHeader, s);
    if (e0 == hipSuccess) e0 = fill_async(out_size, 0, sizeof(uint64_t), s);
    return static_cast<int>(e0);
  }
  const size_t granule_bytes = chunks * kGranuleStride * sizeof(unsigned long long);
  if (L.staging_off + granule_bytes > L.total) return DBHIP_EWORKSPACE;
  // one fill for the header (status, ticket), the unused chunk counts and the granules behind them
  const hipError_t e = fill_async(workspace, 0, L.staging_off + granule_bytes, s);
  if (e != hipSuccess) return static_cast<int>(e);
  const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
  unsigned long long *osz = reinterpret_cast<unsigned long long *>(out_size);
  ScanWs *hdr = reinterpret_cast<ScanWs *>(base);
  const size_t cap = static_cast<size_t>(dev.cus) * (16 / kKpWaves);  // as many as fit: 16 waves per CU at 120 VGPRs
  const unsigned grid = static_cast<unsigned>(chunks < cap ? chunks : cap);
  if (aligned)
    hipLaunchKernelGGL(scan_dense_kernel<true>, dim3(grid), dim3(kKpWaves * kWave), 0, s, src, n, filter_value, out, osz, hdr,
                       granules, chunks);
  else
    hipLaunchKernelGGL(scan_dense_kernel<false>, dim3(grid), dim3(kKpWaves * kWave), 0, s, src, n, filter_value, out, osz, hdr,
                       granules, chunks);
  return launch_status();
}
