/*
 * dbhip.h — C ABI of libdbhip.so, the MI355X (gfx950 / CDNA4) backend for dwarf_bench's
 * data-parallel dwarf kernels.
 *
 * This is the drop-in seam behind the reference's Dwarf::run() hook (common/dwarf.hpp:15-16):
 * every `...Hip` dwarf (dwarf_bench_amd/host/) and every test / bench driver calls exactly these
 * entry points.  Plain pointers and sizes only: all data pointers are DEVICE pointers (hipMalloc'd
 * or any allocator handing out device memory, e.g. torch), `stream` is a hipStream_t passed as
 * void*, and every call is asynchronous on that stream.  Nothing here allocates, frees or
 * synchronises: scratch memory is caller-provided through the `*_workspace_bytes` queries, so a
 * call sequence can be captured into a hipGraph.  (One exception, named as such below:
 * dbhip_radix_sort_prepare, an optional calibration call that is never needed inside a capture.)
 *
 * Return value: 0 on success, a negative DBHIP_E* code for argument errors detected on the host,
 * or a positive hipError_t if a launch failed.  Device-side failures (an out-of-range group key,
 * a full hash table) are reported through the status word that
 * lives at the start of the workspace: read it back with dbhip_workspace_status() after the
 * stream has been synchronised.
 *
 * Reference interfaces replaced (file:line in kurapov-peter/dwarf_bench):
 *   dbhip_copy_if_lt_i32      scan/scan.cl:3-42 (kernel simple_two_pass_scan), scan/scan.cpp:107-128,
 *                             common/dpcpp/dpl_wrapper/dpl_wrapper.hpp:27-33 (copy_if) <- scan/dplscan.cpp:43
 *   dbhip_radix_sort_*        dpl_wrapper.hpp:35-39 (sort) <- sort/radix.cpp:34
 *   dbhip_groupby_sum_u32     groupby/groupby.cpp:58-93 (hash_build + hash_build_check),
 *                             common/dpcpp/hashtable.hpp:136-153, groupby/groupby_local.cpp:58-112
 *   dbhip_join_*              common/dpcpp/omnisci_hashtable.hpp:58-261 <- join/join_omnisci.cpp:74-88
 *   dbhip_ujoin_*             join/join.cpp:60-104, common/dpcpp/hashtable.hpp:5-93,
 *                             common/dpcpp/hashfunctions.hpp:64-137 (MurmurHash3_x86_32)
 *   dbhip_groupby_partial/merge_u32  groupby/groupby_local.cpp:52-112 (the two timed phases of GroupByLocal)
 *   dbhip_bitmask_table_*     common/dpcpp/hashtable.hpp:5-93 (SimpleNonOwningHashTable) <- hash/hash_build.cpp:8-98,
 *                             join/join.cpp:30-38, tests/hash_table_tests.cpp
 *   dbhip_reduce_sum_i32      reduce/reduce.cpp:27-88
 *   dbhip_nested_join_u32     join/nested_join.cpp:52-66
 *   dbhip_pjoin_*             no reference counterpart (multi-GPU radix-partitioned join)
 *   dbhip_gen_*               common/common.hpp:31-40, common/common.cpp:7-20 (data generators)
 *   dbhip_exclusive_scan_u32  scan/scan.cl:44-66, tests/scan_tests.cpp:14-21, dpl_wrapper.hpp:18-25 (exclusive_scan)
 *   dbhip_check_*             the dwarfs' own result checks (scan/scan.cpp:157-164, sort/radix.cpp:46-52,
 *                             groupby/groupby.cpp:95-103, join/join_omnisci.cpp:31-45, join/join.cpp:133-137)
 */
#ifndef DBHIP_H
#define DBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DBHIP_VERSION 1

/* host-side argument errors (negative); positive values are hipError_t */
#define DBHIP_OK 0
#define DBHIP_EINVAL (-1)     /* null pointer / bad size / bad parameter */
#define DBHIP_EWORKSPACE (-2) /* workspace too small or misaligned (needs 256-byte alignment) */
#define DBHIP_ENODEVICE (-3)  /* no HIP device / not a gfx950-class device */

/* device-side status word values (dbhip_workspace_status) */
#define DBHIP_DEV_OK 0u
#define DBHIP_DEV_SPIN_TIMEOUT 1u  /* dbhip_copy_if_lt_dense_i32 and the single-launch path of dbhip_exclusive_scan_u32:
                                      a chunk waited 2 s for its predecessors (never seen); the output is then wrong */
#define DBHIP_DEV_KEY_RANGE 2u     /* group key >= groups_count, or the 0xFFFFFFFF sentinel as a join build key */
#define DBHIP_DEV_TABLE_FULL 4u    /* open-addressing table wrapped without finding a slot: the bitmask-claimed table (a
                                      table that really is full: the reference spins forever there) and the small-input
                                      unique-key table.  NOT the LDS-partitioned joins (dbhip_join_build* / _radix_* /
                                      dbhip_ujoin_* from 2^16 rows): a partition with more distinct keys than its 3072-slot
                                      LDS sub-table holds is built in a spill table of its own, any keys join (the one
                                      exception: a one-to-many build of exactly 2^31 rows has no spare bit to mark such a
                                      sub-table and still raises this) */
#define DBHIP_DEV_RANK_ORDER 8u    /* radix sort: a tile re-ordered by the pass's digit was not sorted by its lower digits —
                                      the ranking was not stable (or an earlier pass was damaged); the output is wrong */

typedef void *dbhip_stream_t; /* hipStream_t */

/* ---- library / device ------------------------------------------------------------------- */
int dbhip_version(void);
/* Fills name (<=len bytes), compute units and wavefront size of HIP device `device`. */
int dbhip_device_info(int device, char *name, size_t len, int *compute_units, int *wave_size);
/* Asynchronously copies the status word of a workspace to *host_status (pinned or pageable),
 * then synchronises `stream`.  Convenience for hosts that have no other D2H path. */
int dbhip_workspace_status(const void *workspace, uint32_t *host_status, dbhip_stream_t stream);

/* ---- deterministic synthetic data (counter-based; oracle/dbo.c dbo_gen_uniform_u32 / dbo_gen_unique_sorted_u32 are the CPU twins) ---------
 * element i of the logical column = lo + mix64(seed, first_index + i) % (hi - lo + 1)            */
int dbhip_gen_uniform_u32(uint32_t *out, size_t n, uint64_t seed, uint64_t first_index,
                          uint32_t lo, uint32_t hi, dbhip_stream_t stream);
/* the same column read at given positions: out[i] = lo + mix64(seed, indices[i]) % (hi - lo + 1) (a key column
 * regenerated in the order of a list of row ids: the partitioned join's checks) */
int dbhip_gen_uniform_at_u32(uint32_t *out, const uint32_t *indices, size_t n, uint64_t seed, uint32_t lo, uint32_t hi,
                             dbhip_stream_t stream);
/* unique ascending keys in [0, 10*N): element i = 10*(first_index+i) + mix64(seed, first_index+i) % 10
 * (same shape as helpers::make_unique_random, common/common.cpp:7-20) */
int dbhip_gen_unique_sorted_u32(uint32_t *out, size_t n, uint64_t seed, uint64_t first_index,
                                dbhip_stream_t stream);

/* ---- dwarf 1: scan / stream compaction --------------------------------------------------------
 * out[0..*out_size) = [x in src : x < filter_value] in source order (stable), *out_size = count.
 * src is read ONCE: a chunked kernel stages each chunk's matches in the workspace (which therefore holds n
 * elements) and a second small kernel moves them to their final offsets; no workgroup waits on another.  `out` needs room for n elements in the worst case.
 * out_size is a DEVICE pointer to one uint64.                                                    */
size_t dbhip_copy_if_lt_i32_workspace_bytes(size_t n);
int dbhip_copy_if_lt_i32(const int32_t *src, size_t n, int32_t filter_value, int32_t *out,
                         uint64_t *out_size, void *workspace, size_t workspace_bytes,
                         dbhip_stream_t stream);

/* The same compaction for DENSE predicates (more than about a tenth of the rows match): one launch, every match
 * written once at its final position (HBM bytes 4n + 4*out_size instead of 4n + 12*out_size).  A workgroup keeps
 * three quarters of its 128 KiB chunk in registers while it waits for the number of matches in front of the chunk
 * and reads the remaining quarter a second time (on-die) during that wait.  Chunks are taken by ticket, so every
 * wait is on a workgroup that is already running; waits are bounded (DBHIP_DEV_SPIN_TIMEOUT).
 * Same arguments, results and workspace size as dbhip_copy_if_lt_i32; slower than it for sparse predicates.        */
int dbhip_copy_if_lt_dense_i32(const int32_t *src, size_t n, int32_t filter_value, int32_t *out,
                               uint64_t *out_size, void *workspace, size_t workspace_bytes,
                               dbhip_stream_t stream);

/* ---- dwarf 2: LSD radix sort ---------------------------------------------------------------
 * Ascending sort of n 32-bit keys.  keys is sorted IN PLACE; tmp is an n-element ping-pong buffer.
 * radix_bits in {4, 8}: digit width of every pass (4 = the configuration named in BASELINE.json,
 * 8 = the tuned variant).  Passes whose digit is constant over the whole input are skipped.
 * keys and tmp must be 16-byte aligned (DBHIP_EINVAL otherwise).                                   */
size_t dbhip_radix_sort_workspace_bytes(size_t n, int radix_bits);
int dbhip_radix_sort_u32(uint32_t *keys, uint32_t *tmp, size_t n, int radix_bits, void *workspace,
                         size_t workspace_bytes, dbhip_stream_t stream);
/* signed order (the reference sorts `int`, sort/radix.cpp:8-12) */
int dbhip_radix_sort_i32(int32_t *keys, int32_t *tmp, size_t n, int radix_bits, void *workspace,
                         size_t workspace_bytes, dbhip_stream_t stream);
/* How the scatter ranks keys on the current device: 1 = one returning LDS atomic per key, 0 = wave ballots, -1 = no
 * device.  The atomic ranking is stable only if same-address lanes of one ds_add_rtn are served in lane order; gfx950
 * does that, the ISA manual does not promise it.  It is therefore (a) the default on gfx950 only, (b) watched by every
 * tile of every sort through the invariant it exists for: inside each run of 64 keys (a wave's row) of a tile re-ordered
 * by the pass's digit, every key is compared with its left neighbour under the mask of the digits sorted so far
 * (DBHIP_DEV_RANK_ORDER in the status word if a pair is out of order: about 2 % of the sort's time).  The pairs that
 * straddle two rows (one in 64) or two tiles are NOT compared: a tripwire for an unstable rank or a damaged earlier
 * pass, not a proof of sortedness.  And (c) pinned by dbhip_radix_sort_prepare.  The
 * same ranking is used inside and outside graph captures.  DBHIP_RS_RANK=ballot|atomic overrides.                 */
int dbhip_radix_sort_rank_mode(void);
/* OPTIONAL calibration, the one call of this library that allocates (a scratch word) and SYNCHRONISES `stream`: runs
 * a device-side self-test of the property above on every CU (~50 us) and pins the ranking of the current device to
 * what it saw (atomics where no lane ever disagreed with the ballots' prediction, ballots otherwise).  Returns the
 * resulting rank mode (1 / 0), DBHIP_EINVAL while `stream` is being captured, or an error code.  The sort entry
 * points never call it and never synchronise.                                                                    */
int dbhip_radix_sort_prepare(dbhip_stream_t stream);

/* ---- dwarf 3: group-by hash aggregate, SUM ------------------------------------------------------
 * out[g] = sum of vals[i] over rows with keys[i] == g (uint32 wrap-around), g in [0, groups).
 * Keys must be < groups (the reference's dense output[key] contract, groupby/groupby.cpp:88-91);
 * a larger key sets DBHIP_DEV_KEY_RANGE and is ignored.  keys and vals must be 16-byte aligned
 * (DBHIP_EINVAL otherwise); the scan, reduce and exclusive-scan entry points take any 4-byte aligned column. */
size_t dbhip_groupby_sum_u32_workspace_bytes(size_t n, uint32_t groups);
int dbhip_groupby_sum_u32(const uint32_t *keys, const uint32_t *vals, size_t n, uint32_t groups,
                          uint32_t *out, void *workspace, size_t workspace_bytes,
                          dbhip_stream_t stream);

/* The two phases on their own (GroupByLocal reports group_by_time and reduction_time separately,
 * groupby/groupby_local.cpp:115-119): partial = private per-workgroup LDS tables written to the workspace,
 * merge = their sum into out[].  max_private_tables caps the number of private tables per key range
 * (the reference's `executors`); 0 = one per compute unit.  Same workspace size as the fused call.      */
int dbhip_groupby_partial_u32(const uint32_t *keys, const uint32_t *vals, size_t n, uint32_t groups,
                              uint32_t max_private_tables, void *workspace, size_t workspace_bytes,
                              dbhip_stream_t stream);
int dbhip_groupby_merge_u32(uint32_t groups, uint32_t max_private_tables, uint32_t *out, const void *workspace,
                            dbhip_stream_t stream);

/* ---- dwarf 4a: one-to-many hash join (JoinOmnisci semantics) -----------------------------------
 * Build: hash table over the DISTINCT build keys, per-key match count, exclusive scan -> position,
 * ids[pos .. pos+count) = build row indices carrying that key (order inside a key's range is not
 * defined, as in the reference).  Probe: per probe row i, out_count[i] = number of build rows with
 * the same key and out_pos[i] = offset of their ids (0/0 on a miss).
 * 0xFFFFFFFF is the empty-slot sentinel and must not occur as a key
 * (join/join_omnisci.cpp:52): a build row carrying it sets DBHIP_DEV_KEY_RANGE and is dropped, a probe row
 * carrying it gets 0/0.  The table lives in the workspace between build and probe; the
 * probe takes n_build again because the table geometry is a pure function of it.
 * A key may repeat any number of times: the rows of a hot key are shared by all workgroups (from 2^18 build rows
 * on; a build in which every other row carries one key takes 1.3-1.6 x the time of uniform keys).  ANY keys join,
 * as with the reference's table (common/dpcpp/omnisci_hashtable.hpp:80-108, ht_size = 2 * distinct keys,
 * join/join_omnisci.cpp:69-70): keys that crowd one partition of the internal radix partitioning beyond its LDS
 * sub-table (constructed against the hash: hashed keys never do) are joined through a spill table — slowly,
 * correctly.  The workspace is 46 bytes per build row (12 table + 16 partition scratch + 18 spill pool).       */
size_t dbhip_join_workspace_bytes(size_t n_build);
int dbhip_join_build_u32(const uint32_t *build_keys, size_t n_build, uint32_t *ids, void *workspace,
                         size_t workspace_bytes, dbhip_stream_t stream);
/* same build over (key, row id) pairs: ids[] receives build_row_ids[] values instead of 0..n-1
 * (the partitioned multi-GPU join builds on received pairs carrying GLOBAL row ids) */
int dbhip_join_build_pairs_u32(const uint32_t *build_keys, const uint32_t *build_row_ids, size_t n_build,
                               uint32_t *ids, void *workspace, size_t workspace_bytes,
                               dbhip_stream_t stream);
int dbhip_join_probe_u32(const uint32_t *probe_keys, size_t n_probe, const void *workspace,
                         size_t n_build, uint32_t *out_pos, uint32_t *out_count,
                         dbhip_stream_t stream);

/* ---- dwarf 4a, radix join: the same one-to-many join for callers that do not need the results in probe-row order
 * (the multi-GPU join, whose results stay sharded anyway).  Both sides are partitioned with the BUILD side's geometry
 * (a pure function of n_build), then ONE launch builds every partition's sub-table in LDS and probes it with the
 * probe rows of the same partition while it is still there: no table in HBM, no random access leaves a CU.
 * Outputs: ids[] grouped by key as above; out_probe_row_ids / out_pos / out_count hold, in the probe side's PARTITION
 * order, each probe row's id (probe_row_ids[i], or i when probe_row_ids is NULL), the offset of its ids and their
 * number.  The three steps are separate entry points so that a host can overlap them with other work (the build
 * side's partition call opens a join: it clears the status word and must come first); dbhip_join_radix_u32 runs all
 * three.  Same contract as above (the sentinel key is flagged; any other keys join).                              */
size_t dbhip_join_radix_workspace_bytes(size_t n_build, size_t n_probe);
int dbhip_join_radix_partition_u32(int probe_side, const uint32_t *keys, const uint32_t *row_ids, size_t n,
                                   size_t n_build, size_t n_probe, void *workspace, size_t workspace_bytes,
                                   dbhip_stream_t stream);
int dbhip_join_radix_match_u32(size_t n_build, size_t n_probe, uint32_t *ids, uint32_t *out_probe_row_ids,
                               uint32_t *out_pos, uint32_t *out_count, void *workspace, size_t workspace_bytes,
                               dbhip_stream_t stream);
int dbhip_join_radix_u32(const uint32_t *build_keys, const uint32_t *build_row_ids, size_t n_build,
                         const uint32_t *probe_keys, const uint32_t *probe_row_ids, size_t n_probe, uint32_t *ids,
                         uint32_t *out_probe_row_ids, uint32_t *out_pos, uint32_t *out_count, void *workspace,
                         size_t workspace_bytes, dbhip_stream_t stream);

/* The reference's answer record (JoinOneToMany<global_ptr<size_t>>, common/dpcpp/omnisci_hashtable.hpp:12-17: pointer
 * into the id buffer + number of ids) for callers that want that shape instead of two 32-bit columns:
 * answers[i] = {ids + out_pos[i], out_count[i]} (a miss: {ids, 0}).  16 bytes per probe row.          */
typedef struct dbhip_join_one_to_many {
  const uint32_t *vals;
  size_t size;
} dbhip_join_one_to_many;
int dbhip_join_answers_u32(const uint32_t *ids, const uint32_t *out_pos, const uint32_t *out_count, size_t n_probe,
                           dbhip_join_one_to_many *answers, dbhip_stream_t stream);

/* ---- dwarf 4b: unique-key join carrying payloads (Join semantics, join/join.cpp:60-104) ---------
 * Build keys are unique.  For probe row i: on a hit out_key[i] = key, out_build_val[i] = payload of
 * the build row, out_probe_val[i] = probe payload; on a miss all three are left at 0xFFFFFFFF
 * (the caller's sentinel fill, join/join.cpp:41-43 — this call writes the sentinel itself).       */
size_t dbhip_ujoin_workspace_bytes(size_t n_build);
int dbhip_ujoin_build_u32(const uint32_t *build_keys, const uint32_t *build_vals, size_t n_build,
                          void *workspace, size_t workspace_bytes, dbhip_stream_t stream);
int dbhip_ujoin_probe_u32(const uint32_t *probe_keys, const uint32_t *probe_vals, size_t n_probe,
                          const void *workspace, size_t n_build, uint32_t *out_key,
                          uint32_t *out_build_val, uint32_t *out_probe_val, dbhip_stream_t stream);

/* ---- bitmask-claimed table: HIP counterpart of SimpleNonOwningHashTable (common/dpcpp/hashtable.hpp:5-93),
 * the table of the reference's Join / HashBuild dwarfs: slots claimed by fetch_or on an occupancy bitmask
 * + ctz over occupied runs; duplicate keys take separate slots.  hash_kind 0 = key % table_size
 * (StaticSimpleHasher / SimpleHasher), 1 = MurmurHash3_x86_32(key, seed) % table_size
 * (hashfunctions.hpp:64-137).  serial != 0 inserts with ONE work-item in input order (reproduces the slot
 * layouts the reference's hash_table_tests expect).  Workspace: header | keys[size] | vals[size] | bitmask. */
size_t dbhip_bitmask_table_workspace_bytes(size_t table_size);
int dbhip_bitmask_table_reset(void *workspace, size_t workspace_bytes, size_t table_size, dbhip_stream_t stream);
int dbhip_bitmask_table_insert_u32(const uint32_t *keys, const uint32_t *vals, size_t n, void *workspace,
                                   size_t workspace_bytes, size_t table_size, int hash_kind, uint32_t seed,
                                   int serial, dbhip_stream_t stream);
int dbhip_bitmask_table_lookup_u32(const uint32_t *keys, size_t n, const void *workspace, size_t table_size,
                                   int hash_kind, uint32_t seed, uint32_t *out_vals, uint32_t *out_found,
                                   dbhip_stream_t stream);

/* ---- multi-GPU radix-partitioned join: device pieces (no reference counterpart) ---------
 * Partition a local column shard into `parts` (1..1024) destination buckets by a mixed hash of
 * the key (independent of the hash the local join partitions by): out_keys / out_row_ids are bucket-major (bucket d occupies
 * [sum(out_counts[0..d)), +out_counts[d])), out_row_ids[i] = first_row_id + local index of the key
 * (global row ids must fit 32 bits), out_counts is a DEVICE array of `parts` uint64.  The exchange
 * between GPUs is the host's job (RCCL all-to-all); the local join on the received pairs is
 * dbhip_join_build_u32 / dbhip_join_probe_u32, and dbhip_gather_u32 (out[i] = table[idx[i]]) turns
 * its build-row indices into global row ids.                                                      */
size_t dbhip_pjoin_partition_workspace_bytes(size_t n, uint32_t parts);
int dbhip_pjoin_partition_u32(const uint32_t *keys, size_t n, uint64_t first_row_id, uint32_t parts,
                              uint32_t *out_keys, uint32_t *out_row_ids, uint64_t *out_counts,
                              void *workspace, size_t workspace_bytes, dbhip_stream_t stream);
int dbhip_gather_u32(const uint32_t *table, const uint32_t *idx, size_t n, uint32_t *out,
                     dbhip_stream_t stream);
/* validator of the exchange's routing: result[0] (DEVICE uint64) = number of keys that dbhip_pjoin_partition_u32
 * would NOT put into bucket `rank` of `parts` (0 on a rank that received only its own keys)            */
int dbhip_check_pjoin_route_u32(const uint32_t *keys, size_t n, uint32_t parts, uint32_t rank, uint64_t *result,
                                dbhip_stream_t stream);

/* ---- the two small dwarfs that complete the taxonomy ---------------------------
 * dbhip_reduce_sum_i32: *out = sum of src[0..n) with int32 wrap-around (reduce/reduce.cpp:27-88, the
 *   oneAPI plus<> reduction into an int; the reference's oracle is std::accumulate(..., 0), :10-22).
 * dbhip_nested_join_u32: the dense n_a x n_b cell matrix of join/nested_join.cpp:52-66, row-major
 *   over (a row, b row): a matching cell holds (key, a_val, b_val), every other cell holds the
 *   reference's "empty" markers (key 0, values 0xFFFFFFFF, nested_join.cpp:30-32).  The kernel writes
 *   all cells, so the three outputs need no initialisation.  n_a <= 1,048,560.                       */
int dbhip_reduce_sum_i32(const int32_t *src, size_t n, int32_t *out, dbhip_stream_t stream);
int dbhip_nested_join_u32(const uint32_t *a_keys, const uint32_t *a_vals, const uint32_t *b_keys,
                          const uint32_t *b_vals, size_t n_a, size_t n_b, uint32_t *out_key,
                          uint32_t *out_val1, uint32_t *out_val2, dbhip_stream_t stream);

/* ---- exclusive prefix sum -------------------------------------------------------
 * dst[0] = init, dst[i] = init + src[0] + ... + src[i-1], uint32 wrap-around: the semantics of the reference's
 * prefix_sum_scalar / prefix_local_test (tests/scan_tests.cpp:14-21, :46-51, scan/scan.cl:44-66) and of
 * oneDPL exclusive_scan behind DPLWrapper::exclusive_scan (common/dpcpp/dpl_wrapper/dpl_wrapper.hpp:18-25,
 * used by common/dpcpp/omnisci_hashtable.hpp:252-254).  dst may alias src.
 * 16-byte aligned src and dst: ONE launch, every element read once and written once (128 KiB chunks kept in registers
 * across a ticketed chunk-to-chunk hand-off; bounded wait: DBHIP_DEV_SPIN_TIMEOUT); other alignments: three launches. */
size_t dbhip_exclusive_scan_u32_workspace_bytes(size_t n);
int dbhip_exclusive_scan_u32(const uint32_t *src, size_t n, uint32_t init, uint32_t *dst, void *workspace,
                             size_t workspace_bytes, dbhip_stream_t stream);

/* ---- device-side validators: Result::valid of the `...Hip` dwarfs above the host-check size -------------
 * Each is an algorithm independent of the kernel it checks; `result` is a DEVICE array of uint64 words,
 * zeroed by the call.  Reference checks replaced: scan/scan.cpp:157-164 (out == std::copy_if),
 * sort/radix.cpp:46-52 (== std::sort), groupby/groupby.cpp:95-103 (== expected_GroupBy),
 * join/join_omnisci.cpp:31-45 (are_equal: size per probe row + membership of every id),
 * join/join.cpp:133-137 (== seq_join).
 *   fingerprint_lt_i32  result[0] = order-sensitive fingerprint of the subsequence x < filter_value,
 *                       result[1] = its length.  copy_if is right iff the pair computed over src equals the
 *                       pair computed over out[0..out_size) (every element of out passes the same filter).
 *   sorted_u32          result[0] = number of i with key[i] > key[i+1] (signed_order != 0: as int32),
 *                       result[1], result[2] = commutative multiset fingerprint (compare with the input's).
 *   weighted_sum_u32    result[0], result[1] = sum vals[i] * w(keys[i]) mod 2^32 for two weight functions;
 *                       keys == NULL means keys[i] = i (the dense group-by output).
 *   permutation_u32     result[0] = number of entries >= n or seen before (0 iff ids is a permutation of 0..n-1).
 *   join_u32            sorted_build_keys = the build column sorted ascending.  result[0] = number of probe rows
 *                       whose count differs from the key's multiplicity in the build column, whose id range
 *                       leaves the id buffer, or whose first / last / one pseudo-random id does not carry the key;
 *                       result[1] = sum of all counts.  build_keys != NULL: ids are build row indices and the key
 *                       of id is build_keys[id]; build_keys == NULL: ids are global row ids of a generated column
 *                       and the key of id is gen_lo + mix64(gen_seed, id) % (gen_hi - gen_lo + 1).
 *   ujoin_u32           build keys unique and sorted ascending (what dbhip_gen_unique_sorted_u32 produces):
 *                       result[0] = probe rows whose (key, build payload, probe payload) triple or sentinels are
 *                       wrong, result[1] = number of hits.
 *   gen_uniform_u32     result[0] = number of i with values[i] != lo + mix64(seed, index_i) % (hi - lo + 1),
 *                       index_i = indices ? indices[i] : first_index + i.                              */
size_t dbhip_check_fingerprint_workspace_bytes(size_t n);
int dbhip_check_fingerprint_lt_i32(const int32_t *src, size_t n, int32_t filter_value, uint64_t *result,
                                   void *workspace, size_t workspace_bytes, dbhip_stream_t stream);
int dbhip_check_sorted_u32(const uint32_t *keys, size_t n, int signed_order, uint64_t *result,
                           dbhip_stream_t stream);
int dbhip_check_weighted_sum_u32(const uint32_t *keys, const uint32_t *vals, size_t n, uint64_t *result,
                                 dbhip_stream_t stream);
size_t dbhip_check_permutation_workspace_bytes(size_t n);
int dbhip_check_permutation_u32(const uint32_t *ids, size_t n, uint64_t *result, void *workspace,
                                size_t workspace_bytes, dbhip_stream_t stream);
int dbhip_check_join_u32(const uint32_t *sorted_build_keys, size_t n_build, const uint32_t *probe_keys,
                         size_t n_probe, const uint32_t *out_pos, const uint32_t *out_count, const uint32_t *ids,
                         const uint32_t *build_keys, uint64_t gen_seed, uint32_t gen_lo, uint32_t gen_hi,
                         uint64_t *result, dbhip_stream_t stream);
int dbhip_check_ujoin_u32(const uint32_t *sorted_build_keys, const uint32_t *build_vals, size_t n_build,
                          const uint32_t *probe_keys, const uint32_t *probe_vals, size_t n_probe,
                          const uint32_t *out_key, const uint32_t *out_build_val, const uint32_t *out_probe_val,
                          uint64_t *result, dbhip_stream_t stream);
int dbhip_check_gen_uniform_u32(const uint32_t *values, const uint32_t *indices, size_t n, uint64_t seed,
                                uint64_t first_index, uint32_t lo, uint32_t hi, uint64_t *result,
                                dbhip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DBHIP_H */
