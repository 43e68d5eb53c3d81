// pjoin_engine.hpp — host engine of the radix-partitioned multi-GPU hash join (no reference counterpart:
// the reference is single-device).  JoinOmnisci semantics (join/join_omnisci.cpp:49-118) over key columns sharded
// across `world` ranks, one rank per GPU:
//
//     compute stream :  partition R |            partition S | local partition R' | local partition S' + match
//     exchange stream:          counts R | exchange R | counts S | exchange S |
// (local join = the radix join of include/dbhip.h; with Options::radix_local = false: build R | probe S)
//
// Per rank two HIP streams and events between them; the only host waits inside a step are the two tiny count
// gathers (the receive sizes must be host integers) and the final sync.  The exchange is ONE RCCL group of
// ncclSend/ncclRecv per relation (every GPU talks to every peer at once: one xGMI link per pair, no ring), the
// P x P count matrix travels by ncclAllGather.  Two ways to host the ranks:
//   * one process drives all ranks (the `PartitionedJoinHip --gpus P` dwarf): ncclCommInitAll, every RCCL call of the
//     local ranks inside one ncclGroupStart/End; with more ranks than GPUs (rehearsal on one GPU) the ranks share
//     devices and the exchange is the same transfers as hipMemcpyPeerAsync pushes;
//   * one process per GPU (bench.py under torch.distributed.run, through the C entry points at the end of this
//     header): ncclCommInitRank with an id created by rank 0 and handed round by the launcher.
// Results stay sharded by key hash: per rank (probe global row id, position, count) + the id buffer of GLOBAL build
// row ids.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace pjoin {

struct Options {
  unsigned world = 1;          // ranks of the join
  bool all_local = true;       // this process drives every rank; false: exactly one (`rank` on `device`)
  unsigned rank = 0;
  int device = 0;
  const void *nccl_id = nullptr;  // all_local == false: the 128-byte ncclUniqueId every rank passes
  bool force_copy = false;     // all_local: exchange by hipMemcpyPeerAsync even where RCCL could be used
  bool direct_single = false;  // world == 1: plain local join without partition / exchange
  bool radix_local = true;     // local join = the radix join (both received sides partitioned alike, fused LDS build +
                               // probe, no table in HBM); false: build + row-ordered probe (the first implementation)
  uint64_t build_seed = 42, probe_seed = 43;  // columns: key_i = mix64(seed, i) % n_total (SURVEY 8d join regime)
  // SUB-JOINS (round 4): every rank's rows are cut by one or two more bits of the rank hash into `sub_joins` (1, 2 or 4)
  // independent joins — equal keys share all hash bits — whose exchanges follow each other on the links while the local
  // join of the one before runs: the work that can only start behind the LAST exchange (local partition of the last
  // relation to arrive + match) shrinks to 1 / sub_joins of what it is for one join.  0: DWARF_BENCH_PJOIN_SUBJOINS or 2.
  unsigned sub_joins = 0;
};

struct StepTimes {  // microseconds; phases are device-event spans (max over the local ranks) and overlap by design
  double total = 0, partition = 0, exchange = 0, build = 0, probe = 0;
  double until_build_done = 0;  // host clock: step start -> the slowest local rank's build finished
  // the two send/recv groups on their own (exchange stream events): R's pairs, then S's pairs — `exchange` above also
  // spans the wait for partition S and the gather of S's counts between them
  double exchange_r = 0, exchange_s = 0;
};

struct CheckReport {  // sums over the local ranks; a multi-process caller adds them up over its ranks
  uint64_t bad_pairs = 0;    // received (key, row id) pairs that are not what the generator produced for that row
  uint64_t bad_route = 0;    // received keys that hash to another rank
  uint64_t bad_rows = 0;     // probe rows with a wrong count / id range / ids not carrying the key
  uint64_t matches = 0;      // sum of all counts
  uint64_t recv_build = 0, recv_probe = 0, sent_rows = 0;
  uint32_t sent_sum[4] = {0, 0, 0, 0}, recv_sum[4] = {0, 0, 0, 0};  // wrap-around column sums (conservation)
};

class Engine {
 public:
  Engine(size_t n_total, const Options &opts);
  ~Engine();
  Engine(const Engine &) = delete;
  Engine &operator=(const Engine &) = delete;

  // untimed: one partition pass to learn the receive sizes, then every buffer of the steady state
  void plan();
  // one pipelined join over all ranks; returns when every local rank has finished
  StepTimes step();
  // after a step: conservation sums, generator / routing / per-row checks on the device (outside the timed region)
  CheckReport check();
  // conservation over ALL ranks: all-reduces the sums when the ranks live in several processes
  bool conserved(const CheckReport &local);

  unsigned world() const;
  unsigned sub_joins() const;  // what Options::sub_joins came to (1 in the direct one-GPU mode)
  unsigned local_ranks() const;
  bool uses_rccl() const;
  // ranks RCCL itself reports for the first local rank's communicator (ncclCommCount; 0 without RCCL): what a reader of
  // a multi-GPU bench line needs to see that the exchange really ran over `world` ranks
  unsigned rccl_ranks_seen() const;
  int device_of(unsigned local_index) const;
  size_t n_total() const;
  // host copies of one local rank's results (validation of small runs)
  struct HostShard {
    std::vector<uint32_t> probe_row_ids, probe_keys, pos, cnt, ids;
  };
  HostShard download(unsigned local_index) const;
  std::vector<uint32_t> download_column(unsigned local_index, bool build) const;  // the rank's input shard
  void corrupt_one_count();  // fault injection for tests: flips one bit of one count of the first local rank

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace pjoin

// ---- C entry points for a one-process-per-GPU launcher (bench.py over ctypes) -------------------------------------
extern "C" {
// rank 0: a fresh ncclUniqueId (128 bytes) to hand to every rank
int dbench_pjoin_unique_id(char *out128);
// every rank: device = local GPU index.  world == 1 needs no id.  Returns a handle or NULL (message on stderr).
void *dbench_pjoin_create(uint64_t n_total, unsigned rank, unsigned world, int device, const char *id128,
                          int direct_single);
// one join; times_us[6] = total, partition, exchange, build, probe, until_build_done.  0 on success.
int dbench_pjoin_step(void *handle, double *times_us);
// the same with the caller's capacity: up to 8 values (..., exchange_r, exchange_s); returns the number written (> 0) or < 0
int dbench_pjoin_step_n(void *handle, double *times_us, unsigned capacity);
// words[4] = ranks RCCL reports for this rank's communicator (ncclCommCount; 0 = no RCCL), world, this rank's HIP device,
// local ranks; name (<= len bytes) = that device's name.  0 on success.
int dbench_pjoin_info(void *handle, unsigned *words, char *name, unsigned long len);
// after a step; words[16]: bad_pairs, bad_route, bad_rows, matches, recv_build, recv_probe, sent_rows,
// conserved (over ALL ranks: collective call), sent_sum[4], recv_sum[4].  0 on success.
int dbench_pjoin_check(void *handle, uint64_t *words);
void dbench_pjoin_destroy(void *handle);
}
