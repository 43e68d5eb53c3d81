// scan.hip — dwarf 1: stable stream compaction out = [x in src : x < filter] for gfx950.
//
// Replaces the reference kernel simple_two_pass_scan (scan/scan.cl:3-42: chunked count -> serial
// prefix by work-item 0 -> chunked write, which reads src twice with uncoalesced per-lane chunks)
// and the oneDPL copy_if behind DPLScan (dpl_wrapper.hpp:27-33).  Same logical result
// (scan/scan.cpp:12-17 expected_out_lt), different machine mapping:
//
//   * ONE pass over src.  Every wave reads 2048 contiguous elements of a tile with eight
//     16-byte-per-lane non-temporal loads (each wave instruction = 1 KiB contiguous), keeps them in
//     registers, counts, and emits the matches once the tile's global offset is known.
//   * Tiles are handed out by a ticket counter (one returning atomic per tile), NOT by blockIdx: a
//     tile index is only ever held by a workgroup that is running, so a look-back can never wait on
//     a workgroup that is not resident (other kernels may share the GPU), and tiles start in ticket
//     order, which keeps look-back distances short.
//   * The offset comes from a decoupled look-back over one 8-byte {state, value} granule per tile
//     (lookback.hpp): agent-scope relaxed atomics, flag and value in one store, no fences, nothing
//     depends on workgroup->XCD placement or dispatch order.
//   * In-wave ranks come from the compare masks themselves: v_cmp -> 64-bit ballot in SGPRs,
//     s_bcnt1 for totals, v_mbcnt_lo/hi for the lane-exclusive prefix.
//
// Two kernels: a simple one (4 waves, 32 KiB tiles) for small inputs and the LDS-staged,
// deferred-write one below for large inputs.
//
// Algorithmic HBM bytes: 4*n read + 4*out_size written (+ ~24 B of granule/ticket traffic per tile).
#include <climits>

#include "dbhip_common.hpp"
#include "lookback.hpp"

namespace dbhip {
namespace {

constexpr int kScanVpt = 8;                           // 16-byte loads per lane per tile
constexpr int kScanWaveElems = kWave * kScanVpt * 4;  // 2048 contiguous elements per wave
constexpr int kScanSmallWaves = 4;                    // 32 KiB tiles
constexpr int kScanSmallTile = kScanWaveElems * kScanSmallWaves;
constexpr size_t kNoTile = ~static_cast<size_t>(0);

struct ScanWs {
  unsigned status;  // DBHIP_DEV_* bits
  unsigned pad0;
  unsigned long long ticket;  // next tile index to hand out
  unsigned pad[60];
  // followed by one 8-byte granule per tile
};
static_assert(sizeof(ScanWs) == kWsHeader, "workspace header size");

__device__ __forceinline__ unsigned long long take_tickets(ScanWs *ws, unsigned long long count) {
  return __hip_atomic_fetch_add(&ws->ticket, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// loads the 2048 elements of wave `wave` of tile `tile` (tile size = kScanWaveElems * WAVES)
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

// =================================================================================================
// Small inputs: one tile at a time per workgroup, immediate look-back, direct writes.
// =================================================================================================
template <bool kAligned, bool kNontemporal>
__global__ __launch_bounds__(kScanSmallWaves *kWave) void copy_if_lt_kernel(
    const int *__restrict__ src, size_t n, int filter, int *__restrict__ out,
    unsigned long long *__restrict__ out_size, ScanWs *ws, size_t num_tiles) {
  __shared__ unsigned s_wave_total[kScanSmallWaves];
  __shared__ unsigned long long s_tile_excl;
  __shared__ unsigned long long s_tile;

  unsigned long long *granules = reinterpret_cast<unsigned long long *>(ws + 1);
  const unsigned lane = threadIdx.x & (kWave - 1);
  const unsigned wave = threadIdx.x / kWave;

  while (true) {
    if (threadIdx.x == 0) s_tile = take_tickets(ws, 1);
    __syncthreads();
    const size_t tile = s_tile;
    if (tile >= num_tiles) return;  // uniform

    i32x4 v[kScanVpt];
    load_wave_tile<kScanSmallWaves, kAligned, kNontemporal>(v, src, n, tile, wave, lane);
    const unsigned wave_total = count_wave_matches(v, filter);
    if (lane == 0) s_wave_total[wave] = wave_total;
    __syncthreads();

    unsigned wave_excl = 0, tile_total = 0;
#pragma unroll
    for (int w = 0; w < kScanSmallWaves; ++w) {
      const unsigned t = s_wave_total[w];
      wave_excl += w < static_cast<int>(wave) ? t : 0u;
      tile_total += t;
    }
    if (wave == 0) {
      unsigned long long excl = 0;
      if (tile == 0) {
        if (lane == 0) st_agent(granules, kLb64Inclusive | tile_total);
      } else {
        if (lane == 0) st_agent(granules + tile, kLb64Aggregate | tile_total);
        excl = lookback_wave64(granules, tile, lane, &ws->status);
        if (lane == 0) st_agent(granules + tile, kLb64Inclusive | ((excl + tile_total) & kLb64Value));
      }
      if (lane == 0) {
        s_tile_excl = excl;
        if (tile == num_tiles - 1) *out_size = excl + tile_total;
      }
    }
    __syncthreads();
    if (wave_total != 0) emit_wave_matches(v, filter, out + s_tile_excl + wave_excl);
    // s_tile / s_wave_total / s_tile_excl are rewritten only after the next loop-top barrier
  }
}

// =================================================================================================
// Large inputs: LDS-staged, deferred-write variant.
//
// With every CU streaming, ~30 MiB of loads are in flight chip-wide, so ANY read that has to come
// from memory — a look-back poll included — takes 3.5-4.5 us (queue depth / bandwidth), about the
// time one CU needs for a whole 120 KiB tile.  A tile that waits for its own offset before writing
// therefore runs at half speed.  Here nothing on the streaming path ever waits for a poll:
//   * 15 streaming waves per workgroup read/count 30720-element tiles (double-buffered registers,
//     next tile's loads issued right after the count, barriers that do not drain vmcnt) and compact
//     the matches of tile i into an LDS staging ring in tile-local order;
//   * wave 0 is a control wave.  Per step it publishes tile i's aggregate, issues the 1024-tile window
//     poll for tile i-1 (every lower tile was counted at least a step ago, so the window is complete)
//     and consumes the poll it issued one step earlier for tile i-2, whose round trip has long
//     finished; tickets are drawn the same way, one step ahead of their use;
//   * tile i-2 is then flushed from the ring to out[] with fully coalesced stores while tile i+1 is
//     in flight.  The ring (120 KiB) holds two consecutive tiles' matches; if tile i-1 and tile i do
//     not fit together (selectivity > 50 %) tile i-1 is resolved with a blocking look-back and
//     flushed in the same step — correct at any selectivity, deferred whenever it fits.
// =================================================================================================
constexpr int kStWaves = 16;
constexpr int kStStream = kStWaves - 1;
constexpr int kStThreads = kStWaves * kWave;
constexpr int kStTile = kStStream * kScanWaveElems;  // 30720 elements = 120 KiB
constexpr unsigned kStRing = kStTile;                // staging ring capacity in elements
constexpr unsigned kStReplicas = 16;                 // copies of the granule array (<= 64)

// entries between replicas: a whole number of 4 KiB pages plus an odd 4352 B, so that the hot
// windows of different replicas do not alias onto the same memory channels
inline size_t st_replica_stride(size_t tiles) {
  return align_up((kLbPad + tiles) * sizeof(unsigned), 4096) / sizeof(unsigned) + 1088;
}

// zero status/ticket/granules and preset the kLbPad "tiles before 0" of every replica
__global__ __launch_bounds__(256) void scan_init_staged_kernel(ScanWs *ws, unsigned rep_stride) {
  unsigned *w = reinterpret_cast<unsigned *>(ws);
  const unsigned total = kWsHeader / 4 + kStReplicas * rep_stride;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    unsigned v = 0;
    if (i >= kWsHeader / 4 && (i - kWsHeader / 4) % rep_stride < kLbPad) v = kLb32Inclusive;
    w[i] = v;
  }
}

struct StLds {
  int ring[kStRing];
  unsigned totals[2][kStWaves];
  unsigned long long excl[2];  // [0]: offset of the tile flushed in the deferred slot, [1]: forced slot
  unsigned long long ticket[2];
};

// Diagnostic build only (-DDBHIP_SCAN_PROFILE): the control wave of one workgroup accumulates the
// time between stamps into spare words of the workspace header (words 8..).
#ifdef DBHIP_SCAN_PROFILE
#define DBP_STAMP(k)                                                                       \
  do {                                                                                     \
    if (control && lane == 0) {                                                            \
      const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                    \
      if (c.prof_on) atomicAdd(&ws->pad[8 + (k)], static_cast<unsigned>(now_ - c.prof_t)); \
      c.prof_t = now_;                                                                     \
    }                                                                                      \
  } while (0)
#else
#define DBP_STAMP(k) \
  do {               \
  } while (0)
#endif

// A tile whose matches sit in the ring, not yet written out.  Uniform over the workgroup: every wave
// keeps its own copy in scalar registers.
struct StPending {
  size_t tile;   // kNoTile = none
  unsigned cnt;  // matches
  unsigned off;  // start in the ring
};

// State of the control wave that lives across steps.
struct StControl {
  LbWindow win;              // window polled during the previous step
  unsigned long long drawn;  // ticket drawn during the previous step
#ifdef DBHIP_SCAN_PROFILE
  unsigned long long prof_t;
  bool prof_on;
#endif
};

// staging: like emit_wave_matches, into the ring (indices wrap at kStRing)
__device__ __forceinline__ void stage_wave_matches(const i32x4 (&v)[kScanVpt], int filter, int *ring,
                                                   unsigned start) {
  unsigned row_base = 0;
#pragma unroll
  for (int k = 0; k < kScanVpt; ++k) {
    const bool m0 = v[k].x < filter, m1 = v[k].y < filter, m2 = v[k].z < filter, m3 = v[k].w < filter;
    const unsigned long long b0 = __ballot(m0), b1 = __ballot(m1), b2 = __ballot(m2), b3 = __ballot(m3);
    if ((b0 | b1 | b2 | b3) == 0) continue;
    unsigned idx = start + row_base + mbcnt(b0) + mbcnt(b1) + mbcnt(b2) + mbcnt(b3);
    idx = idx >= kStRing ? idx - kStRing : idx;
    if (m0) { ring[idx] = v[k].x; idx = idx + 1 == kStRing ? 0 : idx + 1; }
    if (m1) { ring[idx] = v[k].y; idx = idx + 1 == kStRing ? 0 : idx + 1; }
    if (m2) { ring[idx] = v[k].z; idx = idx + 1 == kStRing ? 0 : idx + 1; }
    if (m3) { ring[idx] = v[k].w; }
    row_base += __builtin_popcountll(b0) + __builtin_popcountll(b1) + __builtin_popcountll(b2) +
                __builtin_popcountll(b3);
  }
}

__device__ __forceinline__ void flush_pending(const StPending &p, const int *ring, int *__restrict__ out,
                                              unsigned long long excl) {
  int *dst = out + excl;
  for (unsigned j = threadIdx.x - kWave; j < p.cnt; j += kStStream * kWave) {
    unsigned idx = p.off + j;
    idx = idx >= kStRing ? idx - kStRing : idx;
    dst[j] = ring[idx];
  }
}

// control wave: exclusive prefix of pending tile p (window already polled if have_window), publish
// its INCLUSIVE granule, hand the offset to the streaming waves through LDS slot `slot`
__device__ __forceinline__ void resolve_pending(const StPending &p, StLds &L, int slot, ScanWs *ws,
                                                unsigned *g32, unsigned rep_stride,
                                                __amdgpu_buffer_rsrc_t rsrc, size_t num_tiles,
                                                unsigned long long *out_size, unsigned lane,
                                                LbWindow &win, bool have_window) {
  unsigned excl = 0;
  if (p.tile != 0) {
    excl = lookback1024(rsrc, static_cast<long long>(kLbPad + p.tile), lane, win, have_window, &ws->status);
    if (lane < kStReplicas)  // one store instruction, one lane per replica
      st_agent(g32 + lane * rep_stride + kLbPad + p.tile, kLb32Inclusive | ((excl + p.cnt) & kLb32Value));
  }
  if (lane == 0) {
    L.excl[slot] = excl;
    if (p.tile == num_tiles - 1) *out_size = static_cast<unsigned long long>(excl) + p.cnt;
  }
}

// One step.  `tile` (already in `cur`) is counted and staged, `next_tile` is prefetched into `nxt`,
// p1 (staged last step) gets its window polled, p2 (staged two steps ago) is resolved and flushed.
// PAR alternates 0/1 between consecutive steps.  Returns the ticket for the step after next.
//
// Order inside a step — the memory queue of a CU is a FIFO and its loaded latency is ~4 us:
//   count | control wave's polls/publish/ticket (queue empty: they leave at once) | prefetch of the
//   next tile (keeps the CU streaming) | resolve p2 from the window polled a whole step ago |
//   flush p2 | stage tile i.
template <bool kAligned, bool kNontemporal, unsigned PAR>
__device__ __forceinline__ size_t staged_step(i32x4 (&cur)[kScanVpt], i32x4 (&nxt)[kScanVpt], StLds &L,
                                              const int *__restrict__ src, size_t n, int filter,
                                              int *__restrict__ out, unsigned long long *out_size,
                                              ScanWs *ws, unsigned *g32, unsigned rep_stride,
                                              __amdgpu_buffer_rsrc_t rsrc, size_t num_tiles, size_t tile,
                                              size_t next_tile, StPending &p1, StPending &p2,
                                              unsigned wave, unsigned lane, StControl &c) {
  constexpr unsigned par = PAR;
  const bool control = wave == 0;
  const unsigned swave = wave - 1;
  const bool has_tile = tile < num_tiles;  // false in the epilogue steps (drain of p1/p2)
  DBP_STAMP(0);

  // ---- phase 1: streaming waves count tile i; the control wave hands over last step's ticket
  unsigned wave_total = 0;
  if (!control) {
    if (has_tile) {
      wave_total = count_wave_matches(cur, filter);
      if (lane == 0) L.totals[par][swave] = wave_total;
    }
  } else {
    if (lane == 0) L.ticket[par] = c.drawn;
    DBP_STAMP(5);
  }
  wg_barrier_lds_only();
  DBP_STAMP(1);

  // ---- phase 2 (control wave): everything issued during the previous step has retired by now
  //      (the ticket, its youngest operation, was just consumed), so p2 is resolved FIRST from the
  //      window polled a whole step ago — no wait — and only then the new operations enter the
  //      (empty) queue: window poll for p1, tile i's aggregate, next ticket.  Consuming in issue
  //      order matters: the compiler's vmcnt bookkeeping cannot skip over younger operations
  //      across loop iterations and would wait for the fresh polls too.
  unsigned wave_excl = 0, tile_total = 0;
  if (has_tile) {
#pragma unroll
    for (int w = 0; w < kStStream; ++w) {
      const unsigned t = L.totals[par][w];
      wave_excl += w < static_cast<int>(swave) ? t : 0u;
      tile_total += t;
    }
  }
  const size_t after_next = L.ticket[par];
  const bool have_p1 = p1.tile != kNoTile, have_p2 = p2.tile != kNoTile;
  unsigned my_off = have_p1 ? p1.off + p1.cnt : 0u;
  my_off = my_off >= kStRing ? my_off - kStRing : my_off;
  // p1 and tile i must share the ring; if they cannot, p1 is resolved and flushed in this step too
  const bool force_p1 = has_tile && have_p1 && p1.cnt + tile_total > kStRing;
  if (control) {
    if (have_p2)
      resolve_pending(p2, L, 0, ws, g32, rep_stride, rsrc, num_tiles, out_size, lane, c.win, true);
    DBP_STAMP(6);
    if (have_p1 && p1.tile != 0) lb1024_issue(rsrc, static_cast<long long>(kLbPad + p1.tile), lane, c.win);
    if (has_tile && lane < kStReplicas)  // one store instruction, one lane per replica
      st_agent(g32 + lane * rep_stride + kLbPad + tile,
               (tile == 0 ? kLb32Inclusive : kLb32Aggregate) | tile_total);
    c.drawn = kNoTile;
    if (has_tile && after_next < num_tiles && lane == 0) c.drawn = take_tickets(ws, 1);
  }
  wg_barrier_lds_only();
  DBP_STAMP(2);

  // ---- phase 3: streaming waves start tile i+1 (the barriers do not drain vmcnt: these loads stay
  //      in flight through everything below) and flush p2 from the ring with coalesced stores
  if (!control) {
    if (has_tile && next_tile < num_tiles)
      load_wave_tile<kStStream, kAligned, kNontemporal>(nxt, src, n, next_tile, swave, lane);
    if (have_p2) flush_pending(p2, L.ring, out, L.excl[0]);
  }
  if (force_p1) {  // uniform: the ring cannot hold p1 and tile i, resolve (blocking) and flush p1 now
    if (control) resolve_pending(p1, L, 1, ws, g32, rep_stride, rsrc, num_tiles, out_size, lane, c.win, true);
    wg_barrier_lds_only();
    if (!control) flush_pending(p1, L.ring, out, L.excl[1]);
  }
  // bookkeeping (uniform): what stays pending after this step
  if (force_p1) {
    p2.tile = kNoTile;
  } else {
    p2 = p1;
  }
  p1.tile = has_tile ? tile : kNoTile;
  p1.cnt = tile_total;
  p1.off = my_off;
  if (!has_tile) return after_next;
  wg_barrier_lds_only();
  DBP_STAMP(3);

  // ---- phase 4: streaming waves stage tile i into the ring
  if (!control && wave_total != 0) stage_wave_matches(cur, filter, L.ring, my_off + wave_excl);
  return after_next;
}

template <bool kAligned, bool kNontemporal>
__global__ __launch_bounds__(kStThreads) void copy_if_lt_staged_kernel(
    const int *__restrict__ src, size_t n, int filter, int *__restrict__ out,
    unsigned long long *__restrict__ out_size, ScanWs *ws, size_t num_tiles, unsigned rep_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  StLds &L = *reinterpret_cast<StLds *>(s_raw);
  // kStReplicas copies of [kLbPad preset entries][num_tiles granules]; every tile is published to all
  // of them, this workgroup polls only one: 256 CUs polling the same 4 KiB would overload the few
  // HBM channels that hold it (and stall every stream that crosses those channels)
  unsigned *g32 = reinterpret_cast<unsigned *>(ws + 1);
  const unsigned replica = blockIdx.x % kStReplicas;
  const __amdgpu_buffer_rsrc_t rsrc =
      lb_make_rsrc(g32 + replica * rep_stride, kLbPad + static_cast<unsigned>(num_tiles));
  const unsigned lane = threadIdx.x & (kWave - 1);
  const unsigned wave = threadIdx.x / kWave;

  // Three SEPARATE draws: tickets interleave with the other workgroups' draws, so tile order follows
  // the order in which tiles will actually be counted (this one now, the next a step later, ...).
  if (threadIdx.x == 0) {
    L.ticket[0] = take_tickets(ws, 1);
    L.ticket[1] = take_tickets(ws, 1);
    L.excl[0] = take_tickets(ws, 1);
  }
  __syncthreads();
  size_t tile = L.ticket[0];
  size_t next = L.ticket[1];
  const size_t third = L.excl[0];
  if (tile >= num_tiles) return;
  __syncthreads();  // everyone has read the slots before step 0 overwrites them

  i32x4 a[kScanVpt], b[kScanVpt];
  StControl c;
  c.drawn = third;
#ifdef DBHIP_SCAN_PROFILE
  c.prof_t = __builtin_amdgcn_s_memrealtime();
  c.prof_on = tile == 100;
#endif
#pragma unroll
  for (int j = 0; j < 4; ++j) c.win.q[j] = u32x4{0u, 0u, 0u, 0u};
  StPending p1 = {kNoTile, 0u, 0u}, p2 = {kNoTile, 0u, 0u};
  if (wave != 0) load_wave_tile<kStStream, kAligned, kNontemporal>(a, src, n, tile, wave - 1, lane);
  while (true) {
    size_t nn = staged_step<kAligned, kNontemporal, 0u>(a, b, L, src, n, filter, out, out_size, ws, g32,
                                                        rep_stride, rsrc, num_tiles, tile, next, p1, p2,
                                                        wave, lane, c);
    if (tile >= num_tiles && p1.tile == kNoTile && p2.tile == kNoTile) break;  // drained
    tile = next;
    next = nn;
    nn = staged_step<kAligned, kNontemporal, 1u>(b, a, L, src, n, filter, out, out_size, ws, g32,
                                                 rep_stride, rsrc, num_tiles, tile, next, p1, p2, wave,
                                                 lane, c);
    if (tile >= num_tiles && p1.tile == kNoTile && p2.tile == kNoTile) break;
    tile = next;
    next = nn;
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
// elements from which the staged kernel is used (>= 4 of its tiles per CU at 256 CUs)
inline size_t scan_big_threshold() {
  static const int lg = env_int("DBHIP_SCAN_BIG_LOG2", 10, 40, 25);
  return static_cast<size_t>(1) << lg;
}

template <bool kAligned, bool kNontemporal>
int launch_small(const int *src, size_t n, int filter, int *out, unsigned long long *osz, ScanWs *ws,
                 const DeviceInfo &dev, size_t tiles, hipStream_t s) {
  // tickets make residency irrelevant for correctness: fill the chip
  const size_t cap = static_cast<size_t>(dev.cus) * env_int("DBHIP_SCAN_BLOCKS_PER_CU", 1, 8, 6);
  const unsigned grid = static_cast<unsigned>(tiles < cap ? tiles : cap);
  hipLaunchKernelGGL((copy_if_lt_kernel<kAligned, kNontemporal>), dim3(grid),
                     dim3(kScanSmallWaves * kWave), 0, s, src, n, filter, out, osz, ws, tiles);
  return launch_status();
}

template <bool kAligned, bool kNontemporal>
int launch_staged(const int *src, size_t n, int filter, int *out, unsigned long long *osz, ScanWs *ws,
                  const DeviceInfo &dev, size_t tiles, hipStream_t s) {
  auto kernel = copy_if_lt_staged_kernel<kAligned, kNontemporal>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(StLds));
  if (e != hipSuccess) return static_cast<int>(e);
  // one workgroup per CU (the staging buffer takes 120 KiB of the CU's LDS)
  const size_t cap = static_cast<size_t>(dev.cus);
  const unsigned grid = static_cast<unsigned>(tiles < cap ? tiles : cap);
  const unsigned rep_stride = static_cast<unsigned>(st_replica_stride(tiles));
  hipLaunchKernelGGL(scan_init_staged_kernel, dim3(64), dim3(256), 0, s, ws, rep_stride);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kStThreads), sizeof(StLds), s, src, n, filter, out, osz, ws,
                     tiles, rep_stride);
  return launch_status();
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

extern "C" size_t dbhip_copy_if_lt_i32_workspace_bytes(size_t n) {
  // 8-byte granules of the finer (32 KiB) tiling; it also bounds the staged kernel's
  // kLbPad + n/30720 four-byte granules for every n it is used for
  const size_t tiles = (n + kScanSmallTile - 1) / kScanSmallTile;
  const size_t small_bytes = (tiles ? tiles : 1) * sizeof(unsigned long long);
  const size_t staged_bytes = kStReplicas * st_replica_stride((n + kStTile - 1) / kStTile) * sizeof(unsigned);
  return align_up(kWsHeader + (small_bytes > staged_bytes ? small_bytes : staged_bytes), kWsAlign);
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
    hipError_t e = hipMemsetAsync(workspace, 0, kWsHeader, s);
    if (e == hipSuccess) e = hipMemsetAsync(out_size, 0, sizeof(uint64_t), s);
    return static_cast<int>(e);
  }
  // the staged kernel's 4-byte granules hold prefix sums < 2^30
  const bool staged = n >= scan_big_threshold() && n < (1ull << 30);
  const size_t tile = staged ? kStTile : kScanSmallTile;
  const size_t tiles = (n + tile - 1) / tile;
  // ticket counter, status word and granules must be zero before every launch
  if (!staged) {  // (the staged path clears its replicated granules with its own init kernel)
    hipError_t e = hipMemsetAsync(workspace, 0, align_up(kWsHeader + tiles * sizeof(unsigned long long), 16), s);
    if (e != hipSuccess) return static_cast<int>(e);
  }

  ScanWs *ws = static_cast<ScanWs *>(workspace);
  unsigned long long *osz = reinterpret_cast<unsigned long long *>(out_size);
  const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
  const bool nt = scan_nontemporal();
  if (staged) {
    if (!aligned) return launch_staged<false, false>(src, n, filter_value, out, osz, ws, dev, tiles, s);
    if (nt) return launch_staged<true, true>(src, n, filter_value, out, osz, ws, dev, tiles, s);
    return launch_staged<true, false>(src, n, filter_value, out, osz, ws, dev, tiles, s);
  }
  if (!aligned) return launch_small<false, false>(src, n, filter_value, out, osz, ws, dev, tiles, s);
  if (nt) return launch_small<true, true>(src, n, filter_value, out, osz, ws, dev, tiles, s);
  return launch_small<true, false>(src, n, filter_value, out, osz, ws, dev, tiles, s);
}
