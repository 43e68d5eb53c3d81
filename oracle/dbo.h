/*
 * dbo.h — CPU oracle: a plain-C restatement of the reference's algorithms for the four dwarf paths.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (libdbhip.so, libdbench.so, the dwarf_bench_amd Python package) never does.
 *
 * Pinning: checked against (a) every golden vector the reference's own tests hold for these paths
 * (tests/golden/reference_kats.json, transcribed from /root/reference/tests), and (b) the reference
 * sources that compile here with plain g++ (oracle/_ref: join_helpers.hpp seq_join,
 * hashfunctions.hpp Murmur3 / Polynomial / Simple hashers) via tests/golden/make_golden.py.
 * The device paths of the reference (SYCL / oneDPL / OpenCL-CPU) are unbuildable in this image.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef DBO_H
#define DBO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- deterministic data (CPU twin of dbhip_gen_*; replaces helpers::make_random's random_device,
 *      common/common.hpp:31-40) -------------------------------------------------------------- */
uint64_t dbo_mix64(uint64_t seed, uint64_t i);
void dbo_gen_uniform_u32(uint32_t *out, size_t n, uint64_t seed, uint64_t first_index, uint32_t lo,
                         uint32_t hi);
void dbo_gen_unique_sorted_u32(uint32_t *out, size_t n, uint64_t seed, uint64_t first_index);

/* ---- scan ------------------------------------------------------------------------------------ */
/* scan/scan.cpp:12-17 expected_out_lt: std::copy_if(x < filter).  Returns the count. */
size_t dbo_copy_if_lt_i32(const int32_t *src, size_t n, int32_t filter, int32_t *out);
/* scan/scan.cl:3-42 simple_two_pass_scan restated as three barrier-separated phases over `tnum`
 * work-items (8 on CPU, scan/scan.cpp:65-70).  Faithful: the n % tnum tail is dropped (scan.cl:11).
 * prefix has tnum+1 entries.  threads>1 runs phases 1 and 3 on that many pthreads. */
void dbo_two_pass_scan_i32(const int32_t *src, size_t n, int32_t filter, int32_t *out,
                           int32_t *out_size, int32_t *prefix, int tnum, int threads);
/* generalisation used as the CPU baseline: same three phases, T chunks, tail included */
size_t dbo_chunked_scan_i32(const int32_t *src, size_t n, int32_t filter, int32_t *out, int threads);
/* tests/scan_tests.cpp:14-21 prefix_sum_scalar (exclusive) */
void dbo_prefix_sum_exclusive_i32(const int32_t *in, size_t n, int32_t *out);

/* ---- sort ------------------------------------------------------------------------------------ */
/* sort/radix.cpp:8-12 expected_out: std::sort ascending */
void dbo_sort_i32(int32_t *keys, size_t n);
void dbo_sort_u32(uint32_t *keys, size_t n);
/* CPU baseline: parallel LSD radix sort, 8-bit digits (stands in for oneDPL/TBB parallel sort,
 * dpl_wrapper.hpp:35-39, sort/tbbsort.cpp:22) */
void dbo_radix_sort_u32_mt(uint32_t *keys, uint32_t *tmp, size_t n, int threads);

/* ---- hashers (common/dpcpp/hashfunctions.hpp) -------------------------------------------- */
uint32_t dbo_polynomial_hash(uint32_t v, int p, size_t sz);           /* :3-31  */
uint32_t dbo_simple_hash(uint32_t v, size_t sz);                      /* :43-49 */
uint32_t dbo_murmur3_x86_32(uint32_t key, uint32_t seed);             /* :64-137, len = 4, before % sz */

/* ---- group-by -------------------------------------------------------------------------------- */
/* groupby/groupby.cpp:8-19 expected_GroupBy with f = + (uint32 wrap-around) */
void dbo_groupby_sum_u32(const uint32_t *keys, const uint32_t *vals, size_t n, uint32_t groups,
                         uint32_t *out);
/* groupby/groupby.cpp:58-93 through NonOwningHashTableNonBitmask::add / at
 * (common/dpcpp/hashtable.hpp:107-153): table of `table_size` slots, PolynomialHasher(p),
 * CAS(empty->key) + fetch_add, then per-row lookup into dense out[key].  threads>1 uses atomics.
 * Returns 0, or -1 if the table filled up. */
int dbo_groupby_hash_u32(const uint32_t *keys, const uint32_t *vals, size_t n, uint32_t groups,
                         size_t table_size, int p, uint32_t *out, int threads);
/* groupby/groupby_local.cpp:58-112: `executors` private LinearHashtables (SimpleHasher) + serial merge */
void dbo_groupby_local_u32(const uint32_t *keys, const uint32_t *vals, size_t n, uint32_t groups,
                           size_t executors, uint32_t *out, int threads);

/* ---- one-to-many join (common/dpcpp/omnisci_hashtable.hpp) ---------------------------------- */
typedef struct {
  size_t ht_size;
  uint32_t *ht;  /* keys, empty = 0xFFFFFFFF           (:58-77)  */
  size_t *cnt;   /* per-slot match count               (:223-248) */
  size_t *pos;   /* exclusive scan of cnt              (:250-261) */
  size_t *ids;   /* build row ids grouped by slot      (:115-146) */
  size_t n_build;
} dbo_join_table;
size_t dbo_count_distinct_u32(const uint32_t *v, size_t n);            /* join/join_omnisci.cpp:10-13 */
/* build_table + build_id_buffer with SimpleHasher(ht_size) = key % ht_size
 * (join/join_omnisci.cpp:69-82).  threads>1 uses atomics like the SYCL kernels. */
int dbo_join_build(dbo_join_table *t, const uint32_t *keys, size_t n, size_t ht_size, int threads);
/* lookup (:149-192): per probe row offset into ids and count; a miss leaves {0,0} */
void dbo_join_probe(const dbo_join_table *t, const uint32_t *probe, size_t n, size_t *out_pos,
                    size_t *out_cnt, int threads);
void dbo_join_free(dbo_join_table *t);
/* join/join_omnisci.cpp:15-29 build_join_id_buffer: brute force, per probe row the match count and,
 * if ids_out != NULL, the ascending build row ids (concatenated; offsets in off_out[n_probe+1]) */
void dbo_join_bruteforce(const uint32_t *a, size_t na, const uint32_t *b, size_t nb, size_t *cnt_out,
                         size_t *off_out, size_t *ids_out);

/* ---- unique-key payload join ---------------------------------------------------------------- */
/* reduce/reduce.cpp:10-22 expected_out: std::accumulate(v.begin(), v.end(), 0) in int — restated with
 * unsigned wrap-around so the result is defined for every input (the reference asserts its inputs
 * cannot overflow, :13-19; on those inputs the two agree).
 * join/nested_join.cpp:52-66: the dense cell matrix, pre-filled with (0, 0xFFFFFFFF, 0xFFFFFFFF) (:30-32). */
int32_t dbo_reduce_sum_i32(const int32_t *src, size_t n);
void dbo_nested_join_u32(const uint32_t *a_keys, const uint32_t *a_vals, size_t na, const uint32_t *b_keys,
                         const uint32_t *b_vals, size_t nb, uint32_t *out_key, uint32_t *out_v1,
                         uint32_t *out_v2);

/* join/join_helpers/join_helpers.hpp:86-104 seq_join (a-major, b-minor).  Returns rows written
 * (outputs sized na*nb worst case by the caller, or NULL to only count). */
size_t dbo_seq_join_u32(const uint32_t *a_keys, const uint32_t *a_vals, size_t na,
                        const uint32_t *b_keys, const uint32_t *b_vals, size_t nb, uint32_t *out_key,
                        uint32_t *out_v1, uint32_t *out_v2);
/* common/dpcpp/hashtable.hpp:5-93 SimpleNonOwningHashTable: bitmask claim (fetch_or + ctz) insert,
 * at/has.  hash_kind: 0 = StaticSimpleHasher (v % size), 1 = Murmur3(seed) % size. */
typedef struct {
  size_t size, bitmask_sz;
  uint32_t *keys, *vals, *bitmask;
  int hash_kind;
  uint32_t seed;
} dbo_bitmask_table;
int dbo_bitmask_table_init(dbo_bitmask_table *t, size_t size, int hash_kind, uint32_t seed);
uint32_t dbo_bitmask_table_insert(dbo_bitmask_table *t, uint32_t key, uint32_t val); /* returns slot */
int dbo_bitmask_table_at(const dbo_bitmask_table *t, uint32_t key, uint32_t *val);   /* 1 = found */
void dbo_bitmask_table_free(dbo_bitmask_table *t);
/* join/join.cpp:60-131: build over (a_keys,a_vals), probe with b; per probe row outputs with the
 * 0xFFFFFFFF sentinel on a miss (join.cpp:41-43, :95-100).  ht_size = 2*na (join.cpp:30). */
int dbo_ujoin_u32(const uint32_t *a_keys, const uint32_t *a_vals, size_t na, const uint32_t *b_keys,
                  const uint32_t *b_vals, size_t nb, uint32_t seed, uint32_t *out_key,
                  uint32_t *out_build_val, uint32_t *out_probe_val);

#ifdef __cplusplus
}
#endif
#endif
