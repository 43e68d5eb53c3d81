/*
 * dbo.c — CPU oracle (see dbo.h).  TEST INFRASTRUCTURE ONLY: never linked or loaded by the product.
 * Plain C11 + pthreads; __atomic builtins where the reference's SYCL kernels use sycl::atomic.
 */
#include "dbo.h"

#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define DBO_EMPTY 0xFFFFFFFFu
#define DBO_MAX_THREADS 256

/* ------------------------------------------------------------------------------------------ */
/* tiny fork-join helper: fn(tid, nthreads, ctx) on `threads` pthreads (inline when threads <= 1) */
typedef void (*dbo_fn)(int tid, int nthreads, void *ctx);
typedef struct {
  dbo_fn fn;
  int tid, n;
  void *ctx;
} dbo_task;
static void *dbo_tramp(void *p) {
  dbo_task *t = (dbo_task *)p;
  t->fn(t->tid, t->n, t->ctx);
  return NULL;
}
static void dbo_parallel(int threads, dbo_fn fn, void *ctx) {
  if (threads <= 1) {
    fn(0, 1, ctx);
    return;
  }
  if (threads > DBO_MAX_THREADS) threads = DBO_MAX_THREADS;
  pthread_t th[DBO_MAX_THREADS];
  dbo_task tk[DBO_MAX_THREADS];
  for (int i = 0; i < threads; ++i) {
    tk[i].fn = fn;
    tk[i].tid = i;
    tk[i].n = threads;
    tk[i].ctx = ctx;
    pthread_create(&th[i], NULL, dbo_tramp, &tk[i]);
  }
  for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
}
static void dbo_range(size_t n, int tid, int nt, size_t *lo, size_t *hi) {
  size_t per = (n + (size_t)nt - 1) / (size_t)nt;
  *lo = per * (size_t)tid;
  *hi = *lo + per;
  if (*lo > n) *lo = n;
  if (*hi > n) *hi = n;
}

/* ------------------------------------------------------------------------------------------ */
/* data generators — must stay bit-identical with dbhip::mix64 (csrc/dbhip_common.hpp) */
uint64_t dbo_mix64(uint64_t seed, uint64_t i) {
  uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
void dbo_gen_uniform_u32(uint32_t *out, size_t n, uint64_t seed, uint64_t first, uint32_t lo,
                         uint32_t hi) {
  const uint64_t span = (uint64_t)hi - lo + 1;
  for (size_t i = 0; i < n; ++i) out[i] = lo + (uint32_t)(dbo_mix64(seed, first + i) % span);
}
void dbo_gen_unique_sorted_u32(uint32_t *out, size_t n, uint64_t seed, uint64_t first) {
  for (size_t i = 0; i < n; ++i)
    out[i] = (uint32_t)(10ull * (first + i) + dbo_mix64(seed, first + i) % 10ull);
}

/* ------------------------------------------------------------------------------------------ */
/* scan */
size_t dbo_copy_if_lt_i32(const int32_t *src, size_t n, int32_t filter, int32_t *out) {
  size_t k = 0; /* scan/scan.cpp:14-15 std::copy_if + back_inserter */
  for (size_t i = 0; i < n; ++i)
    if (src[i] < filter) out[k++] = src[i];
  return k;
}

typedef struct {
  const int32_t *src;
  size_t n;
  int32_t filter;
  int32_t *out;
  int32_t *prefix;     /* int prefix, reference layout */
  size_t *prefix64;    /* size_t prefix, generalised variant */
  int tnum;
  size_t work;         /* elements per work-item */
  int include_tail;
} scan_ctx;

static void scan_phase1(int tid, int nt, void *p) { /* scan.cl:13-19 */
  scan_ctx *c = (scan_ctx *)p;
  for (int id = tid; id < c->tnum; id += nt) {
    size_t b = (size_t)id * c->work, e = b + c->work;
    if (c->include_tail && (id == c->tnum - 1 || e > c->n)) e = (id == c->tnum - 1) ? c->n : e;
    if (b > c->n) b = c->n;
    if (e > c->n) e = c->n;
    size_t sz = 0;
    for (size_t i = b; i < e; ++i)
      if (c->src[i] < c->filter) sz++;
    if (c->prefix) c->prefix[id + 1] = (int32_t)sz;
    if (c->prefix64) c->prefix64[id + 1] = sz;
  }
}
static void scan_phase3(int tid, int nt, void *p) { /* scan.cl:33-41 */
  scan_ctx *c = (scan_ctx *)p;
  for (int id = tid; id < c->tnum; id += nt) {
    size_t b = (size_t)id * c->work, e = b + c->work;
    if (c->include_tail && (id == c->tnum - 1 || e > c->n)) e = (id == c->tnum - 1) ? c->n : e;
    if (b > c->n) b = c->n;
    if (e > c->n) e = c->n;
    size_t out_idx = c->prefix ? (size_t)c->prefix[id] : c->prefix64[id];
    size_t idx = 0;
    for (size_t i = b; i < e; ++i)
      if (c->src[i] < c->filter) c->out[out_idx + idx++] = c->src[i];
  }
}

void dbo_two_pass_scan_i32(const int32_t *src, size_t n, int32_t filter, int32_t *out,
                           int32_t *out_size, int32_t *prefix, int tnum, int threads) {
  scan_ctx c = {src, n, filter, out, prefix, NULL, tnum, n / (size_t)tnum, 0}; /* scan.cl:11 */
  dbo_parallel(threads, scan_phase1, &c);
  prefix[0] = 0; /* scan.cl:23-30: work-item 0, serial inclusive sum */
  for (int i = 1; i <= tnum; ++i) prefix[i] += prefix[i - 1];
  *out_size = prefix[tnum];
  dbo_parallel(threads, scan_phase3, &c);
}

size_t dbo_chunked_scan_i32(const int32_t *src, size_t n, int32_t filter, int32_t *out,
                            int threads) {
  int tnum = threads < 1 ? 1 : threads;
  size_t *prefix = (size_t *)calloc((size_t)tnum + 1, sizeof(size_t));
  scan_ctx c = {src, n, filter, out, NULL, prefix, tnum, (n + (size_t)tnum - 1) / (size_t)tnum, 1};
  dbo_parallel(threads, scan_phase1, &c);
  for (int i = 1; i <= tnum; ++i) prefix[i] += prefix[i - 1];
  size_t total = prefix[tnum];
  dbo_parallel(threads, scan_phase3, &c);
  free(prefix);
  return total;
}

void dbo_prefix_sum_exclusive_i32(const int32_t *in, size_t n, int32_t *out) {
  int32_t run = 0; /* tests/scan_tests.cpp:15-19: out[i+1] = out[i] + v[i], last dropped */
  for (size_t i = 0; i < n; ++i) {
    out[i] = run;
    run += in[i];
  }
}

/* ------------------------------------------------------------------------------------------ */
/* sort */
static int cmp_i32(const void *a, const void *b) {
  int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
  return (x > y) - (x < y);
}
static int cmp_u32(const void *a, const void *b) {
  uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return (x > y) - (x < y);
}
void dbo_sort_i32(int32_t *keys, size_t n) { qsort(keys, n, sizeof(int32_t), cmp_i32); }
void dbo_sort_u32(uint32_t *keys, size_t n) { qsort(keys, n, sizeof(uint32_t), cmp_u32); }

typedef struct {
  uint32_t *src, *dst;
  size_t n;
  int shift, nt;
  size_t *hist; /* [nt][256] */
} radix_ctx;
static void radix_hist(int tid, int nt, void *p) {
  radix_ctx *c = (radix_ctx *)p;
  size_t lo, hi;
  dbo_range(c->n, tid, nt, &lo, &hi);
  size_t *h = c->hist + (size_t)tid * 256;
  memset(h, 0, 256 * sizeof(size_t));
  for (size_t i = lo; i < hi; ++i) h[(c->src[i] >> c->shift) & 255u]++;
}
static void radix_scatter(int tid, int nt, void *p) {
  radix_ctx *c = (radix_ctx *)p;
  size_t lo, hi;
  dbo_range(c->n, tid, nt, &lo, &hi);
  size_t *h = c->hist + (size_t)tid * 256;
  for (size_t i = lo; i < hi; ++i) c->dst[h[(c->src[i] >> c->shift) & 255u]++] = c->src[i];
}
void dbo_radix_sort_u32_mt(uint32_t *keys, uint32_t *tmp, size_t n, int threads) {
  if (threads < 1) threads = 1;
  if (threads > DBO_MAX_THREADS) threads = DBO_MAX_THREADS;
  size_t *hist = (size_t *)malloc((size_t)threads * 256 * sizeof(size_t));
  radix_ctx c = {keys, tmp, n, 0, threads, hist};
  for (int pass = 0; pass < 4; ++pass) {
    c.shift = pass * 8;
    dbo_parallel(threads, radix_hist, &c);
    size_t run = 0; /* digit-major, then thread: stable */
    for (int d = 0; d < 256; ++d)
      for (int t = 0; t < threads; ++t) {
        size_t v = hist[(size_t)t * 256 + d];
        hist[(size_t)t * 256 + d] = run;
        run += v;
      }
    dbo_parallel(threads, radix_scatter, &c);
    uint32_t *s = c.src;
    c.src = c.dst;
    c.dst = s;
  }
  free(hist); /* 4 passes: result is back in keys */
}

/* ------------------------------------------------------------------------------------------ */
/* hashers (common/dpcpp/hashfunctions.hpp) */
uint32_t dbo_polynomial_hash(uint32_t v, int p, size_t sz) { /* :13-24, int arithmetic as written */
  uint32_t v_copy = v;
  int res = 0;
  int pow_p = p;
  while (v_copy > 0) {
    /* res += ((v_copy % 10) * pow_p) % _sz  — the product is int*uint32 -> unsigned, % size_t */
    res += (int)((size_t)((v_copy % 10u) * (unsigned)pow_p) % sz);
    res = (int)((size_t)res % sz);
    pow_p = (int)((unsigned)pow_p * (unsigned)p); /* wraps like the reference's int overflow */
    v_copy /= 10u;
  }
  return (uint32_t)res;
}
uint32_t dbo_simple_hash(uint32_t v, size_t sz) { return (uint32_t)(v % sz); } /* :43-49 */

static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t dbo_murmur3_x86_32(uint32_t key, uint32_t seed) { /* :94-136 with _len = 4 */
  uint32_t h1 = seed;
  uint32_t k1 = key; /* one 4-byte block, little-endian load of the key bytes (:109) */
  k1 *= 0xcc9e2d51u;
  k1 = rotl32(k1, 15);
  k1 *= 0x1b873593u;
  h1 ^= k1;
  h1 = rotl32(h1, 13);
  h1 = h1 * 5u + 0xe6546b64u;
  /* no tail for len = 4 (:123-133) */
  h1 ^= 4u;       /* :135 h1 ^= _len */
  h1 ^= h1 >> 16; /* fmix32 :81-89 */
  h1 *= 0x85ebca6bu;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35u;
  h1 ^= h1 >> 16;
  return h1;
}

/* ------------------------------------------------------------------------------------------ */
/* group-by */
void dbo_groupby_sum_u32(const uint32_t *keys, const uint32_t *vals, size_t n, uint32_t groups,
                         uint32_t *out) {
  memset(out, 0, (size_t)groups * sizeof(uint32_t)); /* groupby.cpp:11 */
  for (size_t i = 0; i < n; ++i) out[keys[i]] = out[keys[i]] + vals[i]; /* :14-16 */
}

typedef struct {
  const uint32_t *keys, *vals;
  size_t n, table_size;
  int p;
  uint32_t *tk, *tv, *out;
  int failed;
} gb_ctx;
static void gb_build(int tid, int nt, void *q) { /* hashtable.hpp:136-153 add_update */
  gb_ctx *c = (gb_ctx *)q;
  size_t lo, hi;
  dbo_range(c->n, tid, nt, &lo, &hi);
  for (size_t i = lo; i < hi; ++i) {
    const uint32_t key = c->keys[i];
    const uint32_t h = dbo_polynomial_hash(key, c->p, c->table_size);
    uint32_t at = h;
    for (;;) {
      uint32_t expected = DBO_EMPTY;
      int ok = __atomic_compare_exchange_n(&c->tk[at], &expected, key, 0, __ATOMIC_RELAXED,
                                           __ATOMIC_RELAXED);
      if (ok || expected == key) {
        __atomic_fetch_add(&c->tv[at], c->vals[i], __ATOMIC_RELAXED);
        break;
      }
      at = (uint32_t)((at + 1) % c->table_size);
      if (at == h) {
        c->failed = 1;
        break;
      }
    }
  }
}
static void gb_check(int tid, int nt, void *q) { /* groupby.cpp:82-92 + hashtable.hpp:107-124 at */
  gb_ctx *c = (gb_ctx *)q;
  size_t lo, hi;
  dbo_range(c->n, tid, nt, &lo, &hi);
  for (size_t i = lo; i < hi; ++i) {
    const uint32_t key = c->keys[i];
    const uint32_t h = dbo_polynomial_hash(key, c->p, c->table_size);
    uint32_t pos = h, val = 0;
    int present = c->tk[pos] != DBO_EMPTY;
    while (present) {
      if (c->tk[pos] == key) {
        val = c->tv[pos];
        break;
      }
      pos = (uint32_t)((pos + 1) % c->table_size);
      if (pos == h) break;
      present = c->tk[pos] != DBO_EMPTY;
    }
    __atomic_store_n(&c->out[key], val, __ATOMIC_RELAXED);
  }
}
int dbo_groupby_hash_u32(const uint32_t *keys, const uint32_t *vals, size_t n, uint32_t groups,
                         size_t table_size, int p, uint32_t *out, int threads) {
  gb_ctx c = {keys, vals, n, table_size, p, NULL, NULL, out, 0};
  c.tk = (uint32_t *)malloc(table_size * sizeof(uint32_t));
  c.tv = (uint32_t *)calloc(table_size, sizeof(uint32_t));
  memset(c.tk, 0xFF, table_size * sizeof(uint32_t)); /* groupby.cpp:48-49 */
  memset(out, 0, (size_t)groups * sizeof(uint32_t)); /* :50 */
  dbo_parallel(threads, gb_build, &c);
  dbo_parallel(threads, gb_check, &c);
  free(c.tk);
  free(c.tv);
  return c.failed ? -1 : 0;
}

typedef struct {
  const uint32_t *keys, *vals;
  size_t n, executors, work;
  uint32_t groups;
  uint32_t *tk, *tv;
} gbl_ctx;
static void gbl_build(int tid, int nt, void *q) { /* groupby_local.cpp:58-83 + hashtable.hpp:215-232 */
  gbl_ctx *c = (gbl_ctx *)q;
  for (size_t ex = (size_t)tid; ex < c->executors; ex += (size_t)nt) {
    uint32_t *k = c->tk + ex * c->groups, *v = c->tv + ex * c->groups;
    for (size_t i = c->work * ex; i < c->work * (ex + 1) && i < c->n; ++i) {
      const uint32_t key = c->keys[i], h = key % c->groups;
      uint32_t at = h;
      for (;;) {
        if (k[at] == DBO_EMPTY) k[at] = key;
        if (k[at] == key) {
          v[at] += c->vals[i];
          break;
        }
        at = (at + 1) % c->groups;
        if (at == h) break;
      }
    }
  }
}
void dbo_groupby_local_u32(const uint32_t *keys, const uint32_t *vals, size_t n, uint32_t groups,
                           size_t executors, uint32_t *out, int threads) {
  gbl_ctx c = {keys, vals, n, executors, 0, groups, NULL, NULL};
  /* groupby_local.cpp:56 std::ceil((float)buf_size / executors) */
  c.work = (size_t)(((float)n / (float)executors) + 0.999999f);
  if (c.work * executors < n) c.work = (n + executors - 1) / executors;
  c.tk = (uint32_t *)malloc(executors * groups * sizeof(uint32_t));
  c.tv = (uint32_t *)calloc(executors * groups, sizeof(uint32_t));
  memset(c.tk, 0xFF, executors * groups * sizeof(uint32_t));
  dbo_parallel(threads, gbl_build, &c);
  memset(out, 0, (size_t)groups * sizeof(uint32_t));
  for (size_t ex = 0; ex < executors; ++ex) { /* :96-111 serial collect via ht.at(j) */
    const uint32_t *k = c.tk + ex * groups, *v = c.tv + ex * groups;
    for (uint32_t j = 0; j < groups; ++j) {
      uint32_t pos = j % groups;
      const uint32_t h = pos;
      int present = k[pos] != DBO_EMPTY;
      while (present) {
        if (k[pos] == j) {
          out[j] += v[pos];
          break;
        }
        pos = (pos + 1) % groups;
        if (pos == h) break;
        present = k[pos] != DBO_EMPTY;
      }
    }
  }
  free(c.tk);
  free(c.tv);
}

/* ------------------------------------------------------------------------------------------ */
/* one-to-many join */
size_t dbo_count_distinct_u32(const uint32_t *v, size_t n) {
  if (!n) return 0;
  uint32_t *s = (uint32_t *)malloc(n * sizeof(uint32_t));
  memcpy(s, v, n * sizeof(uint32_t));
  qsort(s, n, sizeof(uint32_t), cmp_u32);
  size_t d = 1;
  for (size_t i = 1; i < n; ++i) d += s[i] != s[i - 1];
  free(s);
  return d;
}

typedef struct {
  dbo_join_table *t;
  const uint32_t *keys;
  size_t n;
  int phase;
  size_t *out_pos, *out_cnt;
} join_ctx;

/* find the slot holding `key` (omnisci_hashtable.hpp:229-244 probing order); SIZE_MAX if absent */
static size_t join_find(const dbo_join_table *t, uint32_t key) {
  const size_t h = key % t->ht_size;
  if (t->ht[h] == key) return h;
  size_t hp = (h + 1) % t->ht_size;
  while (hp != h) {
    if (t->ht[hp] == key) return hp;
    hp = (hp + 1) % t->ht_size;
  }
  return (size_t)-1;
}
static void join_phase(int tid, int nt, void *q) {
  join_ctx *c = (join_ctx *)q;
  dbo_join_table *t = c->t;
  size_t lo, hi;
  dbo_range(c->n, tid, nt, &lo, &hi);
  for (size_t i = lo; i < hi; ++i) {
    const uint32_t key = c->keys[i];
    if (c->phase == 0) { /* build_table :80-108 */
      const size_t h = key % t->ht_size;
      size_t hp = h;
      do {
        uint32_t expected = DBO_EMPTY;
        int ok = __atomic_compare_exchange_n(&t->ht[hp], &expected, key, 0, __ATOMIC_RELAXED,
                                             __ATOMIC_RELAXED);
        if (ok || expected == key) break;
        hp = (hp + 1) % t->ht_size;
      } while (hp != h);
    } else if (c->phase == 1) { /* build_count_buffer :223-248 */
      size_t s = join_find(t, key);
      if (s != (size_t)-1) __atomic_fetch_add(&t->cnt[s], 1, __ATOMIC_RELAXED);
    } else if (c->phase == 2) { /* build_id_buffer kernel :115-146 */
      size_t s = join_find(t, key);
      if (s != (size_t)-1) {
        size_t off = __atomic_fetch_add(&t->cnt[s], 1, __ATOMIC_RELAXED);
        t->ids[t->pos[s] + off] = i;
      }
    } else { /* lookup :149-192 */
      const size_t h = key % t->ht_size;
      size_t id = h;
      int found = 1;
      if (t->ht[h] != key) {
        size_t hp = (h + 1) % t->ht_size;
        for (;;) {
          if (t->ht[hp] == key) {
            id = hp;
            break;
          }
          if (hp == h || t->ht[hp] == DBO_EMPTY) {
            found = 0;
            break;
          }
          hp = (hp + 1) % t->ht_size;
        }
      }
      if (found) {
        c->out_pos[i] = t->pos[id];
        c->out_cnt[i] = t->cnt[id];
      } else { /* default-constructed JoinOneToMany {nullptr, 0} */
        c->out_pos[i] = 0;
        c->out_cnt[i] = 0;
      }
    }
  }
}
int dbo_join_build(dbo_join_table *t, const uint32_t *keys, size_t n, size_t ht_size, int threads) {
  memset(t, 0, sizeof(*t));
  t->ht_size = ht_size;
  t->n_build = n;
  t->ht = (uint32_t *)malloc(ht_size * sizeof(uint32_t));
  t->cnt = (size_t *)calloc(ht_size, sizeof(size_t));
  t->pos = (size_t *)calloc(ht_size, sizeof(size_t));
  t->ids = (size_t *)calloc(n ? n : 1, sizeof(size_t));
  if (!t->ht || !t->cnt || !t->pos || !t->ids) return -1;
  memset(t->ht, 0xFF, ht_size * sizeof(uint32_t)); /* ctor kernel :58-77 */
  join_ctx c = {t, keys, n, 0, NULL, NULL};
  dbo_parallel(threads, join_phase, &c);
  c.phase = 1;
  dbo_parallel(threads, join_phase, &c);
  size_t run = 0; /* build_pos_buffer :250-261: exclusive scan, then cnt zeroed on the host */
  for (size_t i = 0; i < ht_size; ++i) {
    t->pos[i] = run;
    run += t->cnt[i];
    t->cnt[i] = 0;
  }
  c.phase = 2;
  dbo_parallel(threads, join_phase, &c);
  return 0;
}
void dbo_join_probe(const dbo_join_table *t, const uint32_t *probe, size_t n, size_t *out_pos,
                    size_t *out_cnt, int threads) {
  join_ctx c = {(dbo_join_table *)t, probe, n, 3, out_pos, out_cnt};
  dbo_parallel(threads, join_phase, &c);
}
void dbo_join_free(dbo_join_table *t) {
  free(t->ht);
  free(t->cnt);
  free(t->pos);
  free(t->ids);
  memset(t, 0, sizeof(*t));
}
void dbo_join_bruteforce(const uint32_t *a, size_t na, const uint32_t *b, size_t nb, size_t *cnt_out,
                         size_t *off_out, size_t *ids_out) {
  size_t run = 0;
  for (size_t i = 0; i < nb; ++i) { /* join_omnisci.cpp:19-27 */
    size_t c = 0;
    if (off_out) off_out[i] = run;
    for (size_t j = 0; j < na; ++j)
      if (a[j] == b[i]) {
        if (ids_out) ids_out[run + c] = j;
        c++;
      }
    cnt_out[i] = c;
    run += c;
  }
  if (off_out) off_out[nb] = run;
}

/* ------------------------------------------------------------------------------------------ */
/* reduce + nested-loop join */
int32_t dbo_reduce_sum_i32(const int32_t *src, size_t n) {
  uint32_t acc = 0; /* reduce.cpp:21 accumulate from 0 */
  for (size_t i = 0; i < n; ++i) acc += (uint32_t)src[i];
  return (int32_t)acc;
}

void dbo_nested_join_u32(const uint32_t *a_keys, const uint32_t *a_vals, size_t na, const uint32_t *b_keys,
                         const uint32_t *b_vals, size_t nb, uint32_t *out_key, uint32_t *out_v1,
                         uint32_t *out_v2) {
  for (size_t c = 0; c < na * nb; ++c) { /* nested_join.cpp:30-32 */
    out_key[c] = 0;
    out_v1[c] = 0xFFFFFFFFu;
    out_v2[c] = 0xFFFFFFFFu;
  }
  for (size_t it = 0; it < na; ++it) { /* :56-66, one work-item per A row */
    const uint32_t key = a_keys[it], val = a_vals[it];
    for (size_t i = 0; i < nb; ++i)
      if (b_keys[i] == key) {
        out_key[it * nb + i] = key;
        out_v1[it * nb + i] = val;
        out_v2[it * nb + i] = b_vals[i];
      }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* unique-key payload join */
size_t dbo_seq_join_u32(const uint32_t *a_keys, const uint32_t *a_vals, size_t na,
                        const uint32_t *b_keys, const uint32_t *b_vals, size_t nb, uint32_t *out_key,
                        uint32_t *out_v1, uint32_t *out_v2) {
  size_t k = 0;
  for (size_t i = 0; i < na; ++i)
    for (size_t j = 0; j < nb; ++j)
      if (a_keys[i] == b_keys[j]) { /* join_helpers.hpp:94-102 */
        if (out_key) {
          out_key[k] = a_keys[i];
          out_v1[k] = a_vals[i];
          out_v2[k] = b_vals[j];
        }
        k++;
      }
  return k;
}

static uint32_t bm_hash(const dbo_bitmask_table *t, uint32_t key) {
  return t->hash_kind == 0 ? (uint32_t)(key % t->size)
                           : (uint32_t)(dbo_murmur3_x86_32(key, t->seed) % t->size);
}
int dbo_bitmask_table_init(dbo_bitmask_table *t, size_t size, int hash_kind, uint32_t seed) {
  t->size = size;
  t->bitmask_sz = (size + 31) / 32; /* join.cpp:31 ceil(ht_size / 32) */
  t->keys = (uint32_t *)malloc(size * sizeof(uint32_t));
  t->vals = (uint32_t *)calloc(size, sizeof(uint32_t));
  t->bitmask = (uint32_t *)calloc(t->bitmask_sz, sizeof(uint32_t));
  t->hash_kind = hash_kind;
  t->seed = seed;
  if (!t->keys || !t->vals || !t->bitmask) return -1;
  memset(t->keys, 0xFF, size * sizeof(uint32_t)); /* join.cpp:37 */
  return 0;
}
uint32_t dbo_bitmask_table_insert(dbo_bitmask_table *t, uint32_t key, uint32_t val) {
  const uint32_t elem_sz = 32; /* hashtable.hpp:68 */
  uint32_t at = bm_hash(t, key);
  uint32_t major = at / elem_sz; /* update_bitmask :70-92 */
  uint32_t minor = at % elem_sz;
  uint32_t pos;
  for (;;) {
    if ((size_t)major * elem_sz + minor >= t->size) { /* ragged last word: the reference would claim a slot
                                                       * >= size here (out of bounds); wrap instead */
      major = 0;
      minor = 0;
      continue;
    }
    uint32_t mask = 1u << minor;
    uint32_t present = __atomic_fetch_or(&t->bitmask[major], mask, __ATOMIC_RELAXED);
    if (!(present & mask)) {
      pos = major * elem_sz + minor;
      break;
    }
    uint32_t inv = ~(present >> minor);
    uint32_t occupied = inv ? (uint32_t)__builtin_ctz(inv) : 32u;
    if (occupied + minor >= elem_sz) {
      major = (uint32_t)((major + 1) % t->bitmask_sz);
      minor = 0;
    } else {
      minor += occupied;
    }
  }
  t->keys[pos] = key; /* :16-18 */
  t->vals[pos] = val;
  return pos;
}
int dbo_bitmask_table_at(const dbo_bitmask_table *t, uint32_t key, uint32_t *val) {
  uint32_t pos = bm_hash(t, key); /* :23-40 */
  const uint32_t start = pos;
  int present = (t->bitmask[pos / 32] & (1u << (pos % 32))) != 0;
  while (present) {
    if (t->keys[pos] == key) {
      if (val) *val = t->vals[pos];
      return 1;
    }
    pos = (uint32_t)((pos + 1) % t->size);
    if (pos == start) break;
    present = (t->bitmask[pos / 32] & (1u << (pos % 32))) != 0;
  }
  return 0;
}
void dbo_bitmask_table_free(dbo_bitmask_table *t) {
  free(t->keys);
  free(t->vals);
  free(t->bitmask);
  memset(t, 0, sizeof(*t));
}
int dbo_ujoin_u32(const uint32_t *a_keys, const uint32_t *a_vals, size_t na, const uint32_t *b_keys,
                  const uint32_t *b_vals, size_t nb, uint32_t seed, uint32_t *out_key,
                  uint32_t *out_build_val, uint32_t *out_probe_val) {
  dbo_bitmask_table t;
  if (dbo_bitmask_table_init(&t, na ? na * 2 : 1, 1, seed)) return -1; /* join.cpp:30-32 */
  for (size_t i = 0; i < na; ++i) dbo_bitmask_table_insert(&t, a_keys[i], a_vals[i]); /* :65-76 */
  for (size_t i = 0; i < nb; ++i) { /* :90-101 */
    uint32_t v;
    out_key[i] = out_build_val[i] = out_probe_val[i] = DBO_EMPTY; /* :41-43 */
    if (dbo_bitmask_table_at(&t, b_keys[i], &v)) {
      out_key[i] = b_keys[i];
      out_build_val[i] = v;
      out_probe_val[i] = b_vals[i];
    }
  }
  dbo_bitmask_table_free(&t);
  return 0;
}
