/* ORACLE -- TEST INFRASTRUCTURE ONLY (see glx_oracle.h).
 *
 * Plain-C restatement of the reference algorithms, one serial loop per
 * request exactly like the reference.  Build: oracle/Makefile
 * (-O2 -ffp-contract=off so float expressions round like the reference's
 * x86-64 SSE build, no fused multiply-add).
 */
#include "glx_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ RNG -- */
/* Philox4x32-10 (Salmon et al., SC'11).  The reference draws from an
 * unseedable thread_local mt19937 (random_sampler.cc:46-47); the glx seeding
 * contract replaces only that entropy source, never the formulas around it. */
void glxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Contract: key = (seed lo, seed hi); counter = (j >> 1, row, cc lo, cc hi);
 * draw j takes words {2(j&1), 2(j&1)+1} of the block as lo, hi. */
uint64_t glxo_draw64(uint64_t seed, uint64_t call_counter, uint32_t row, uint32_t j) {
  uint32_t ctr[4] = {j >> 1, row, (uint32_t)call_counter, (uint32_t)(call_counter >> 32)};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t o[4];
  glxo_philox4x32_10(ctr, key, o);
  uint32_t w = (j & 1u) * 2u;
  return ((uint64_t)o[w + 1] << 32) | o[w];
}

/* Bounded integer in [0, n): high 64 bits of u * n (bias <= n / 2^64). */
static inline uint64_t bounded(uint64_t u, uint64_t n) {
  return (uint64_t)(((unsigned __int128)u * n) >> 64);
}

/* ---------------------------------------------------------- id -> row ---- */
typedef struct {
  int64_t* keys;
  int64_t* vals;
  uint64_t mask;
} idmap;

static uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

static void idmap_build(idmap* m, const int64_t* ids, int64_t V) {
  uint64_t cap = 16;
  while (cap < (uint64_t)V * 2) cap <<= 1;
  m->mask = cap - 1;
  m->keys = (int64_t*)malloc(cap * sizeof(int64_t));
  m->vals = (int64_t*)malloc(cap * sizeof(int64_t));
  for (uint64_t i = 0; i < cap; ++i) m->vals[i] = -1;
  for (int64_t r = 0; r < V; ++r) {
    uint64_t h = mix64((uint64_t)ids[r]) & m->mask;
    /* first insertion wins, like unordered_map::insert (auto_indexing.cc:21-24) */
    while (m->vals[h] != -1 && m->keys[h] != ids[r]) h = (h + 1) & m->mask;
    if (m->vals[h] == -1) { m->keys[h] = ids[r]; m->vals[h] = r; }
  }
}

static int64_t idmap_get(const idmap* m, int64_t id) {
  uint64_t h = mix64((uint64_t)id) & m->mask;
  while (m->vals[h] != -1) {
    if (m->keys[h] == id) return m->vals[h];
    h = (h + 1) & m->mask;
  }
  return -1;
}

static void idmap_free(idmap* m) { free(m->keys); free(m->vals); }

static inline int64_t row_of(const int64_t* ids, const idmap* m, int64_t V, int64_t id) {
  if (!ids) return (id >= 0 && id < V) ? id : -1;
  return idmap_get(m, id);
}

/* ---------------------------------------------------------------- alias -- */
/* AliasMethod::Build, alias_method.cc:57-107, one CSR row at a time.  low /
 * high sets are LIFO stacks; `sum` accumulates in double and is narrowed to
 * float (std::accumulate(..., 0.0) assigned to a float, :73). */
static void alias_build_row(const float* dist, int32_t count, float* probs, int32_t* alias,
                            int32_t* low_set, int32_t* high_set) {
  if (count == 0) return;
  float avg_prob = (float)(1.0 / count);
  double acc = 0.0;
  for (int32_t i = 0; i < count; ++i) acc += dist[i];
  float sum = (float)acc;
  int32_t low_num = 0, high_num = 0;
  for (int32_t i = 0; i < count; ++i) {
    alias[i] = i;
    float prob = dist[i] / sum;
    probs[i] = prob * count;
    if (prob < avg_prob) {
      low_set[low_num++] = i;
    } else if (prob > avg_prob) {
      high_set[high_num++] = i;
    }
  }
  while (low_num > 0 && high_num > 0) {
    int32_t low_idx = low_set[--low_num];
    int32_t high_idx = high_set[--high_num];
    probs[high_idx] = probs[high_idx] - 1 + probs[low_idx];
    alias[low_idx] = high_idx;
    if (probs[high_idx] < 1.0) {
      low_set[low_num++] = high_idx;
    } else if (probs[high_idx] > 1.0) {
      high_set[high_num++] = high_idx;
    }
  }
  while (low_num > 0) probs[low_set[--low_num]] = 1.0f;
  while (high_num > 0) probs[high_set[--high_num]] = 1.0f;
}

void glxo_alias_build(const int64_t* row_ptr, const float* weight, int64_t V, float* prob_out,
                      int32_t* alias_out) {
  int64_t maxdeg = 0;
  for (int64_t r = 0; r < V; ++r) {
    int64_t d = row_ptr[r + 1] - row_ptr[r];
    if (d > maxdeg) maxdeg = d;
  }
  int32_t* low = (int32_t*)malloc((size_t)(maxdeg + 1) * sizeof(int32_t));
  int32_t* high = (int32_t*)malloc((size_t)(maxdeg + 1) * sizeof(int32_t));
  for (int64_t r = 0; r < V; ++r) {
    int64_t s = row_ptr[r];
    alias_build_row(weight + s, (int32_t)(row_ptr[r + 1] - s), prob_out + s, alias_out + s, low,
                    high);
  }
  free(low);
  free(high);
}

/* ----------------------------------------------------------- row sorting -- */
typedef struct { int64_t col, eid; float w; int64_t pos; } sort_rec;

static int cmp_weight_desc(const void* a, const void* b) {
  const sort_rec* x = (const sort_rec*)a;
  const sort_rec* y = (const sort_rec*)b;
  if (x->w > y->w) return -1;
  if (x->w < y->w) return 1;
  return (x->pos > y->pos) - (x->pos < y->pos);
}

void glxo_sort_rows_by_weight_desc(const int64_t* row_ptr, int64_t V, int64_t* col, int64_t* eid,
                                   float* weight) {
  int64_t maxdeg = 0;
  for (int64_t r = 0; r < V; ++r) {
    int64_t d = row_ptr[r + 1] - row_ptr[r];
    if (d > maxdeg) maxdeg = d;
  }
  sort_rec* buf = (sort_rec*)malloc((size_t)(maxdeg + 1) * sizeof(sort_rec));
  for (int64_t r = 0; r < V; ++r) {
    int64_t s = row_ptr[r], d = row_ptr[r + 1] - s;
    for (int64_t i = 0; i < d; ++i) {
      buf[i].col = col[s + i]; buf[i].eid = eid[s + i]; buf[i].w = weight[s + i]; buf[i].pos = i;
    }
    qsort(buf, (size_t)d, sizeof(sort_rec), cmp_weight_desc);
    for (int64_t i = 0; i < d; ++i) {
      col[s + i] = buf[i].col; eid[s + i] = buf[i].eid; weight[s + i] = buf[i].w;
    }
  }
  free(buf);
}

/* -------------------------------------------------------------- samplers -- */
static int g_reference_cost_model = 0;
void glxo_set_reference_cost_model(int on) { g_reference_cost_model = on; }

/* ---- L1 entropy: the reference's own random variates ------------------------------------------------------------
 * The reference draws from one thread_local std::mt19937 per sampler translation unit -- random_sampler.cc:46-47,
 * random_without_replacement_sampler.cc:53-54, alias_method.cc:114-115 (shared by EdgeWeightSampler and
 * InDegreeSampler) -- seeded once per thread from std::random_device and consumed request after request through
 * libstdc++'s uniform_int_distribution<int>, std::shuffle and uniform_real_distribution<double>.  With
 * glxo_set_reference_entropy(1, seed) glxo_sample takes its variates the same way, so that its outputs can be compared
 * with oracle/_ref (the reference's own code, random_device pinned to `seed`) DRAW FOR DRAW instead of by
 * distribution: everything in the row algorithms but the entropy source -- row lookup, default fill, the alias
 * compare on the float-cast variate, both padders -- is then pinned exactly.  Restated from the published algorithms:
 * MT19937 (Matsumoto & Nishimura 1998; its outputs are fixed by the C++ standard, [rand.predef]); Lemire's nearly
 * divisionless bounded integers (arXiv:1805.10941) as libstdc++ 11 applies them to a 32-bit engine
 * (bits/uniform_int_dist.h, _S_nd); its std::shuffle, which draws two swap positions from one variate while
 * n * n <= 2^32 - 1 (bits/stl_algo.h); generate_canonical<double, 53> = (lo + hi * 2^32) / 2^64 (bits/random.tcc).
 * The distributions are implementation-defined: this matches libstdc++ 11.4, the library oracle/_ref is built with.
 * The contract path (Philox, one stream per row) is untouched by this switch and stays the HIP kernels' target. */
typedef struct {
  uint32_t mt[624];
  int at;
  int seeded;
} ref_engine;

static int g_ref_entropy = 0;
static uint32_t g_ref_seed = 0;
/* one engine per reference source file that owns one: 0 random_sampler.cc, 1 random_without_replacement_sampler.cc,
 * 2 alias_method.cc (EdgeWeight / InDegree samplers, the in-degree / node-weight negative samplers, node2vec),
 * 3 random_negative_sampler.cc, 4 random_walk.cc (DeepWalk) */
enum { ENG_RANDOM = 0, ENG_RWOR = 1, ENG_ALIAS = 2, ENG_RANDOM_NEGATIVE = 3, ENG_WALK = 4, ENG_COUNT = 5 };
static ref_engine g_ref_eng[ENG_COUNT];

void glxo_set_reference_entropy(int on, uint32_t seed) {
  g_ref_entropy = on;
  g_ref_seed = seed;
  for (int e = 0; e < ENG_COUNT; ++e) g_ref_eng[e].seeded = 0; /* "a fresh thread": every engine seeds at its first use */
}

static uint32_t ref_next(ref_engine* g) {
  if (!g->seeded) { /* std::mt19937 engine(rd()) */
    g->mt[0] = g_ref_seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->at = 624;
    g->seeded = 1;
  }
  if (g->at >= 624) {
    for (int i = 0; i < 624; ++i) {
      const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->at = 0;
  }
  uint32_t y = g->mt[g->at++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

/* uniform_int_distribution{0, range - 1} over a 32-bit engine, 1 <= range <= 2^32 - 1 */
static uint32_t ref_uniform_below(ref_engine* g, uint32_t range) {
  uint64_t product = (uint64_t)ref_next(g) * range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (uint32_t)(0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)ref_next(g) * range;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32);
}

/* std::shuffle(a, a + n, engine) for n >= 1 */
static void ref_shuffle(ref_engine* g, int64_t* a, int64_t n) {
  const uint64_t urngrange = 0xffffffffu, un = (uint64_t)n;
  int64_t i = 1, t;
  if (urngrange / un >= un) {
    if ((un % 2) == 0) {
      const uint32_t pos = ref_uniform_below(g, 2);
      t = a[i]; a[i] = a[pos]; a[pos] = t;
      ++i;
    }
    while (i != n) {
      const uint64_t swap_range = (uint64_t)i + 1;
      const uint32_t x = ref_uniform_below(g, (uint32_t)(swap_range * (swap_range + 1)));
      const uint64_t p0 = x / (swap_range + 1), p1 = x % (swap_range + 1);
      t = a[i]; a[i] = a[p0]; a[p0] = t;
      ++i;
      t = a[i]; a[i] = a[p1]; a[p1] = t;
      ++i;
    }
    return;
  }
  for (; i != n; ++i) { /* one position per variate: uniform_int_distribution{0, i} */
    const uint32_t pos = ref_uniform_below(g, (uint32_t)i + 1u);
    t = a[i]; a[i] = a[pos]; a[pos] = t;
  }
}

/* uniform_real_distribution<double>{0, b}(engine) */
static double ref_uniform_real(ref_engine* g, double b) {
  double sum = 0.0, tmp = 1.0;
  for (int k = 0; k < 2; ++k) {
    sum += (double)ref_next(g) * tmp;
    tmp *= 4294967296.0;
  }
  double ret = sum / tmp;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret * (b - 0.0) + 0.0;
}

/* The two places entropy enters a row algorithm.  u = the contract's 64-bit draw for this position (ignored under the
 * reference's entropy, whose engines are sequential). */
static inline int64_t int_variate(int eng, uint64_t u, uint64_t n) { /* uniform in [0, n) */
  return g_ref_entropy ? (int64_t)ref_uniform_below(&g_ref_eng[eng], (uint32_t)n) : (int64_t)bounded(u, n);
}
static inline float alias_variate(uint64_t u, double b) { /* AliasMethod::Sample's `float rand` in [0, b] */
  const double rd = g_ref_entropy ? ref_uniform_real(&g_ref_eng[ENG_ALIAS], b) : ((double)(u >> 11) * 0x1.0p-53) * b;
  return (float)rd;
}

static void fill_default(int64_t* nbr, int64_t* eid, int32_t k, int64_t def) {
  /* SamplingResponse::FillWith, sampling_request.cc:279-290 */
  for (int32_t j = 0; j < k; ++j) { nbr[j] = def; eid[j] = -1; }
}

/* Pad with an index list (indices == NULL means identity of length n_idx).
 * CircularPadder::Pad circular_padder.h:36-66; ReplicatePadder::Pad
 * replicate_padder.h:37-56 -- which ignores the VALUES of the index list
 * (cursor = idx) and is clamped here to the row length instead of reading out
 * of bounds. */
static void pad_row(const int64_t* rn, const int64_t* re, int64_t deg, const int64_t* indices,
                    int64_t n_idx, int32_t k, int padding_mode, int64_t def, int64_t* nbr,
                    int64_t* eid) {
  if (padding_mode == GLXO_PAD_CIRCULAR) {
    if (n_idx == 0) { fill_default(nbr, eid, k, def); return; }
    for (int32_t j = 0; j < k; ++j) {
      int64_t cursor = j % n_idx;
      if (indices) cursor = indices[cursor];
      nbr[j] = rn[cursor];
      eid[j] = re[cursor];
    }
  } else {
    int64_t size = n_idx < k ? n_idx : k;
    if (size > deg) size = deg;
    for (int64_t j = 0; j < size; ++j) { nbr[j] = rn[j]; eid[j] = re[j]; }
    for (int64_t j = size; j < k; ++j) { nbr[j] = def; eid[j] = -1; }
  }
}

int glxo_sample(const glxo_graph* g, int op, const int64_t* src, const int64_t* rng_rows,
                int32_t batch, int32_t k,
                int padding_mode, int64_t default_neighbor_id, uint64_t seed,
                uint64_t call_counter, int64_t* nbr_out, int64_t* eid_out) {
  if (op < GLXO_RANDOM || op > GLXO_IN_DEGREE) return 3;
  if (op == GLXO_IN_DEGREE && (!g->indeg_prob || !g->indeg_alias)) return 3;
  if (op == GLXO_EDGE_WEIGHT && !g_reference_cost_model && (!g->alias_prob || !g->alias_idx)) return 3;
  if (op == GLXO_EDGE_WEIGHT && g_reference_cost_model && !g->weight) return 3;
  idmap m;
  if (g->ids) idmap_build(&m, g->ids, g->V);
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
  int64_t* perm = NULL;
  int64_t perm_cap = 0;
  for (int32_t i = 0; i < batch; ++i) {
    int64_t* nbr = nbr_out + (int64_t)i * k;
    int64_t* eid = eid_out + (int64_t)i * k;
    const uint32_t rr = rng_rows ? (uint32_t)rng_rows[i] : (uint32_t)i;
    int64_t row = row_of(g->ids, &m, g->V, src[i]);
    int64_t start = row < 0 ? 0 : g->row_ptr[row];
    int64_t deg = row < 0 ? 0 : g->row_ptr[row + 1] - start;
    if (deg == 0) { fill_default(nbr, eid, k, default_neighbor_id); continue; }
    const int64_t* rn = g->col + start;
    const int64_t* re = g->eid + start;
    switch (op) {
      case GLXO_RANDOM:
        /* random_sampler.cc:61-71 without a filter: k draws in [0, deg). */
        for (int32_t j = 0; j < k; ++j) {
          int64_t d = int_variate(ENG_RANDOM, glxo_draw64(seed, call_counter, rr, (uint32_t)j), (uint64_t)deg);
          nbr[j] = rn[d];
          eid[j] = re[d];
        }
        break;
      case GLXO_RANDOM_WITHOUT_REPLACEMENT: {
        /* random_without_replacement_sampler.cc:57-68: shuffle(iota(deg)) then
         * Pad.  Contract: forward Fisher-Yates, step j swaps a[j] with
         * a[j + bounded(draw_j, deg - j)]; only the first min(k, deg) steps
         * can influence the padded output, so only those are run. */
        if (padding_mode != GLXO_PAD_CIRCULAR && !g_ref_entropy) {
          pad_row(rn, re, deg, NULL, deg, k, padding_mode, default_neighbor_id, nbr, eid);
          break;
        }
        if (deg > perm_cap) {
          free(perm);
          perm_cap = deg * 2;
          perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)perm_cap);
        }
        for (int64_t t = 0; t < deg; ++t) perm[t] = t;
        if (g_ref_entropy) {
          /* the reference shuffles the WHOLE index list whatever k and the padder are (and so advances its engine) */
          ref_shuffle(&g_ref_eng[ENG_RWOR], perm, deg);
          pad_row(rn, re, deg, perm, deg, k, padding_mode, default_neighbor_id, nbr, eid);
          break;
        }
        int64_t msteps = deg < k ? deg : k;
        for (int64_t j = 0; j < msteps; ++j) {
          int64_t r = j + (int64_t)bounded(
                              glxo_draw64(seed, call_counter, rr, (uint32_t)j),
                              (uint64_t)(deg - j));
          int64_t t = perm[j]; perm[j] = perm[r]; perm[r] = t;
        }
        /* circular pad over the full permutation (indices_->size() == deg);
         * slots j < k only ever index perm[j % deg] with j % deg < msteps. */
        pad_row(rn, re, deg, perm, deg, k, padding_mode, default_neighbor_id, nbr, eid);
        break;
      }
      case GLXO_IN_DEGREE:
      case GLXO_EDGE_WEIGHT: {
        /* alias_method.cc:109-124: rand = float(U[0, deg-1)); idx = int(rand);
         * ret = probs[idx] <= rand - idx ? alias[idx] : idx. */
        const float* probs = g->alias_prob ? g->alias_prob + start : NULL;
        const int32_t* alias = g->alias_idx ? g->alias_idx + start : NULL;
        if (op == GLXO_IN_DEGREE) {
          probs = g->indeg_prob + start;
          alias = g->indeg_alias + start;
        }
        float* tmp_p = NULL;
        int32_t* tmp_a = NULL;
        if (g_reference_cost_model && op == GLXO_EDGE_WEIGHT) {
          /* the reference's per-row, per-request AliasMethod(&edge_weights) */
          tmp_p = (float*)malloc(sizeof(float) * (size_t)deg);
          tmp_a = (int32_t*)malloc(sizeof(int32_t) * (size_t)deg * 3);
          alias_build_row(g->weight + start, (int32_t)deg, tmp_p, tmp_a, tmp_a + deg, tmp_a + 2 * deg);
          probs = tmp_p;
          alias = tmp_a;
        }
        for (int32_t j = 0; j < k; ++j) {
          float rnd = alias_variate(glxo_draw64(seed, call_counter, rr, (uint32_t)j), (double)(deg - 1));
          int32_t ix = (int32_t)rnd;
          idx[j] = (probs[ix] <= (rnd - ix)) ? alias[ix] : ix;
        }
        pad_row(rn, re, deg, idx, k, k, padding_mode, default_neighbor_id, nbr, eid);
        free(tmp_p);
        free(tmp_a);
        break;
      }
      case GLXO_TOPK:
        /* topk_sampler.cc:53-61: iota(deg) then Pad (rows pre-sorted by weight). */
        pad_row(rn, re, deg, NULL, deg, k, padding_mode, default_neighbor_id, nbr, eid);
        break;
    }
  }
  free(idx);
  free(perm);
  if (g->ids) idmap_free(&m);
  return 0;
}

void glxo_in_degree_weights(const int64_t* col, int64_t E, float* w_out) {
  /* distinct neighbour ids -> occurrence counts (TopoStatics::Add, topo_statics.cc:33-60) */
  idmap m;
  uint64_t cap = 16;
  while (cap < (uint64_t)E * 2) cap <<= 1;
  m.mask = cap - 1;
  m.keys = (int64_t*)malloc(cap * sizeof(int64_t));
  m.vals = (int64_t*)malloc(cap * sizeof(int64_t));
  for (uint64_t i = 0; i < cap; ++i) m.vals[i] = -1;
  for (int64_t e = 0; e < E; ++e) {
    uint64_t h = mix64((uint64_t)col[e]) & m.mask;
    while (m.vals[h] != -1 && m.keys[h] != col[e]) h = (h + 1) & m.mask;
    if (m.vals[h] == -1) { m.keys[h] = col[e]; m.vals[h] = 0; }
    m.vals[h]++;
  }
  for (int64_t e = 0; e < E; ++e) w_out[e] = (float)(int32_t)idmap_get(&m, col[e]);
  idmap_free(&m);
}

int64_t glxo_sample_full(const glxo_graph* g, const int64_t* src, int32_t batch, int32_t max_limit,
                         int32_t* degrees_out, int64_t* nbr_out, int64_t* eid_out, int64_t cap) {
  idmap m;
  if (g->ids) idmap_build(&m, g->ids, g->V);
  int64_t total = 0;
  for (int32_t i = 0; i < batch; ++i) {
    int64_t row = row_of(g->ids, &m, g->V, src[i]);
    int64_t start = row < 0 ? 0 : g->row_ptr[row];
    int64_t deg = row < 0 ? 0 : g->row_ptr[row + 1] - start;
    /* GetTruncatedSize, full_sampler.cc:89-96 */
    int64_t take = (max_limit > 0 && max_limit < deg) ? max_limit : deg;
    degrees_out[i] = (int32_t)take;
    for (int64_t j = 0; j < take; ++j) {
      if (total < cap) { nbr_out[total] = g->col[start + j]; eid_out[total] = g->eid[start + j]; }
      ++total;
    }
  }
  if (g->ids) idmap_free(&m);
  return total;
}

/* ------------------------------------------------------------ random walk -- */
static int32_t node2vec_weights(const glxo_graph* g, const idmap* m, int64_t cur, int64_t parent, int has_parent_nbrs,
                                float p, float q, int32_t F, float default_weight, float* w_out, int64_t* start_out) {
  int64_t row = row_of(g->ids, m, g->V, cur);
  if (row < 0) return 0;
  int64_t s = g->row_ptr[row], d = g->row_ptr[row + 1] - s;
  int32_t n = (int32_t)(d < F ? d : F);
  *start_out = s;
  int64_t ps = 0;
  int32_t pn = 0;
  if (has_parent_nbrs) {
    int64_t prow = row_of(g->ids, m, g->V, parent);
    if (prow >= 0) {
      ps = g->row_ptr[prow];
      int64_t pd = g->row_ptr[prow + 1] - ps;
      pn = (int32_t)(pd < F ? pd : F);
    }
  }
  for (int32_t x = 0; x < n; ++x) {
    float ew = g->weight ? g->weight[s + x] : default_weight;
    if (g->col[s + x] == parent) {
      w_out[x] = ew * 1.0 / (p + 1e-6);  /* random_walk.cc:247-248 */
    } else {
      int32_t y = 0;
      for (; y < pn; ++y) {
        if (g->col[ps + y] == g->col[s + x]) { w_out[x] = ew; break; }
      }
      if (y == pn) w_out[x] = ew * 1.0 / (q + 1e-6);
    }
  }
  return n;
}

/* the same weights with the parent's neighbour list handed in (what WeightedRandomWalkKernel is given: a window
 * [start, start + size) of the request's concatenated lists, random_walk.cc:228-262) */
static int32_t node2vec_window_weights(const glxo_graph* g, const idmap* m, int64_t cur, int64_t parent, const int64_t* pl,
                                       int32_t pn, float p, float q, int32_t F, float default_weight, float* w_out,
                                       int64_t* start_out) {
  int64_t row = row_of(g->ids, m, g->V, cur);
  if (row < 0) return 0;
  int64_t s = g->row_ptr[row], d = g->row_ptr[row + 1] - s;
  int32_t n = (int32_t)(d < F ? d : F);
  *start_out = s;
  for (int32_t x = 0; x < n; ++x) {
    float ew = g->weight ? g->weight[s + x] : default_weight;
    if (g->col[s + x] == parent) {
      w_out[x] = ew * 1.0 / (p + 1e-6);
    } else {
      int32_t y = 0;
      for (; y < pn; ++y) {
        if (pl[y] == g->col[s + x]) { w_out[x] = ew; break; }
      }
      if (y == pn) w_out[x] = ew * 1.0 / (q + 1e-6);
    }
  }
  return n;
}

void glxo_node2vec_weights(const glxo_graph* g, int64_t cur, int64_t parent, int has_parent_nbrs, float p, float q,
                           int32_t full_nbr_num, float default_weight, float* w_out, int32_t* n_out) {
  idmap m;
  if (g->ids) idmap_build(&m, g->ids, g->V);
  int64_t s = 0;
  *n_out = node2vec_weights(g, &m, cur, parent, has_parent_nbrs, p, q, full_nbr_num, default_weight, w_out, &s);
  if (g->ids) idmap_free(&m);
}

int glxo_random_walk(const glxo_graph* g, const int64_t* seeds, int32_t batch, int32_t walk_len, float p, float q,
                     int32_t full_nbr_num, float default_weight, int64_t default_neighbor_id, uint64_t seed,
                     uint64_t call_counter, int64_t* walks_out) {
  /* RandomWalkRequest::IsDeepWalk, random_walk_request.cc:152-160 */
  const int deep = fabsf(p - 1.0f) < 32 * FLT_EPSILON && fabsf(q - 1.0f) < 32 * FLT_EPSILON;
  if (!deep && full_nbr_num < 1) return 3;
  idmap m;
  if (g->ids) idmap_build(&m, g->ids, g->V);
  const int32_t F = full_nbr_num > 0 ? full_nbr_num : 1;
  float* w = (float*)malloc(sizeof(float) * (size_t)F * 2);
  int32_t* tab = (int32_t*)malloc(sizeof(int32_t) * (size_t)F * 3);
  /* node2vec: the parents' neighbour lists travel as ONE concatenated array (the previous step's answer), walked with a
   * cursor that random_walk.cc:214-226 advances only for walkers whose current vertex has out-edges: behind a stuck
   * walker every later walker's window starts too early by that walker's list length.  plist / pseg: the lists of the
   * previous step's current vertices, walker by walker. */
  int64_t* plist = deep ? NULL : (int64_t*)malloc(sizeof(int64_t) * ((size_t)batch * (size_t)F + 1));
  int64_t* pnext = deep ? NULL : (int64_t*)malloc(sizeof(int64_t) * ((size_t)batch * (size_t)F + 1));
  int32_t* pseg = deep ? NULL : (int32_t*)calloc((size_t)batch + 1, sizeof(int32_t));
  int32_t* pseg_next = deep ? NULL : (int32_t*)calloc((size_t)batch + 1, sizeof(int32_t));
  for (int32_t t = 0; t < walk_len; ++t) {
    int64_t cursor = 0, filled = 0;
    for (int32_t i = 0; i < batch; ++i) {
      int64_t* walk = walks_out + (int64_t)i * walk_len;
      const int64_t cur = t == 0 ? seeds[i] : walk[t - 1];
      const int64_t parent = t <= 1 ? seeds[i] : walk[t - 2];
      const uint64_t u = glxo_draw64(seed, call_counter + (uint64_t)t, (uint32_t)i, 0);
      if (deep) {
        int64_t row = row_of(g->ids, &m, g->V, cur);
        int64_t s = row < 0 ? 0 : g->row_ptr[row];
        int64_t d = row < 0 ? 0 : g->row_ptr[row + 1] - s;
        walk[t] = d == 0 ? default_neighbor_id : g->col[s + int_variate(ENG_WALK, u, (uint64_t)d)];
        continue;
      }
      int64_t s = 0;
      int32_t n = node2vec_window_weights(g, &m, cur, parent, plist + cursor, t > 0 ? pseg[i] : 0, p, q, F, default_weight, w, &s);
      /* this step's answer carries the current vertex's first min(deg, F) neighbours: the next step's parent lists */
      pseg_next[i] = n;
      for (int32_t x = 0; x < n; ++x) pnext[filled + x] = g->col[s + x];
      filled += n;
      if (n == 0) { walk[t] = default_neighbor_id; continue; } /* ... and the cursor stays where it is */
      if (t > 0) cursor += pseg[i];
      float* probs = w + F;
      alias_build_row(w, n, probs, tab, tab + F, tab + 2 * F);
      float rnd = alias_variate(u, (double)(n - 1));
      int32_t ix = (int32_t)rnd;
      walk[t] = g->col[s + ((probs[ix] <= (rnd - ix)) ? tab[ix] : ix)];
    }
    if (!deep) {
      int64_t* tl = plist; plist = pnext; pnext = tl;
      int32_t* ts = pseg; pseg = pseg_next; pseg_next = ts;
    }
  }
  free(w);
  free(tab);
  free(plist); free(pnext); free(pseg); free(pseg_next);
  if (g->ids) idmap_free(&m);
  return 0;
}

/* ---------------------------------------------------------------- filters -- */
static int64_t filter_field(const glxo_filter* f, const int64_t* row_nbr, const int64_t* row_ts, int32_t idx) {
  /* Filter::GetFieldFunc, filter.cc:122-150 */
  if (f->field == GLXO_FIELD_ID) return row_nbr[idx];
  if (f->field == GLXO_FIELD_TIMESTAMP) return row_ts ? row_ts[idx] : f->default_timestamp;
  return -1;
}

static int filter_hit(const glxo_filter* f, int32_t batch_idx, const int64_t* row_nbr, const int64_t* row_ts,
                      int32_t idx) {
  /* Filter::Hit + GetFilterFunc, filter.cc:98-107,152-193 */
  int64_t v = filter_field(f, row_nbr, row_ts, idx);
  if (f->type == GLXO_FILTER_EQUAL) return v == f->values[batch_idx];
  if (f->type == GLXO_FILTER_LARGER_THAN) return v > f->values[batch_idx];
  return 0;
}

static int32_t filter_find_kth(const glxo_filter* f, const int64_t* row_nbr, const int64_t* row_ts, int32_t n) {
  /* Filter::FindkthLargest with batch_share_idx = 0, filter.cc:196-229 */
  int32_t start = 0, end = n - 1, mid = 0;
  const int64_t filter = f->values[0];
  if (end == 0) return -1;
  while (end >= start) {
    mid = start + (end - start) / 2;
    int64_t v = filter_field(f, row_nbr, row_ts, mid);
    if (v == filter) return mid;
    if (v > filter) end = mid - 1; else start = mid + 1;
  }
  if (filter_field(f, row_nbr, row_ts, mid) < filter) mid += 1;
  return mid;
}

int32_t glxo_filter_act_on(const glxo_filter* f, int32_t batch_idx, const int64_t* row_nbr,
                           const int64_t* row_ts, int32_t n, int32_t* indices) {
  for (int32_t t = 0; t < n; ++t) indices[t] = t;
  if (f->field == GLXO_FIELD_TIMESTAMP && f->type == GLXO_FILTER_LARGER_THAN) {
    int32_t k = filter_find_kth(f, row_nbr, row_ts, n);
    if (k < 0) k = 0;
    for (int32_t a = 0, b = k - 1; a < b; ++a, --b) { int32_t t = indices[a]; indices[a] = indices[b]; indices[b] = t; }
    return k;
  }
  int32_t l = 0, r = n - 1;
  while (l <= r) {
    while (filter_hit(f, batch_idx, row_nbr, row_ts, indices[l]) && (l <= r)) {
      int32_t t = indices[l]; indices[l] = indices[r]; indices[r] = t;
      --r;
    }
    ++l;
  }
  return r + 1;
}

int glxo_sample_filtered(const glxo_graph* g, int op, const int64_t* src, const int64_t* rng_rows,
                         int32_t batch, int32_t k, int padding_mode, int64_t default_neighbor_id,
                         uint64_t seed, uint64_t call_counter, const glxo_filter* f, int64_t* nbr_out,
                         int64_t* eid_out) {
  if (!f || f->type == GLXO_FILTER_NONE)
    return glxo_sample(g, op, src, rng_rows, batch, k, padding_mode, default_neighbor_id, seed, call_counter,
                       nbr_out, eid_out);
  if (op < GLXO_RANDOM || op > GLXO_IN_DEGREE || !f->values) return 3;
  if (op == GLXO_EDGE_WEIGHT && !g->weight) return 3;
  if (op == GLXO_IN_DEGREE && !f->indeg_weight) return 3;
  idmap m;
  if (g->ids) idmap_build(&m, g->ids, g->V);
  int64_t maxdeg = 1;
  for (int64_t r = 0; r < g->V; ++r)
    if (g->row_ptr[r + 1] - g->row_ptr[r] > maxdeg) maxdeg = g->row_ptr[r + 1] - g->row_ptr[r];
  int32_t* res = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxdeg);
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(maxdeg > k ? maxdeg : (k > 0 ? k : 1)));
  float* dist = (float*)malloc(sizeof(float) * (size_t)maxdeg * 2);
  int32_t* tab = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxdeg * 3);
  int32_t retry = f->retry_times;
  for (int32_t i = 0; i < batch; ++i) {
    int64_t* nbr = nbr_out + (int64_t)i * k;
    int64_t* eid = eid_out + (int64_t)i * k;
    const uint32_t rr = rng_rows ? (uint32_t)rng_rows[i] : (uint32_t)i;
    int64_t row = row_of(g->ids, &m, g->V, src[i]);
    int64_t start = row < 0 ? 0 : g->row_ptr[row];
    int32_t deg = row < 0 ? 0 : (int32_t)(g->row_ptr[row + 1] - start);
    if (deg == 0) { fill_default(nbr, eid, k, default_neighbor_id); continue; }
    const int64_t* rn = g->col + start;
    const int64_t* re = g->eid + start;
    const int64_t* rt = f->ts_slot ? f->ts_slot + start : NULL;
    if (op == GLXO_RANDOM) {
      /* random_sampler.cc:58-71: HitAll -> default row; else rejection with the retry budget */
      int all = 1;
      for (int32_t t = 0; t < deg && all; ++t) all &= filter_hit(f, i, rn, rt, t);
      if (all) { fill_default(nbr, eid, k, default_neighbor_id); continue; }
      for (int32_t j = 0; j < k; ++j) {
        for (uint32_t a = 0;; ++a) {
          int32_t d = (int32_t)int_variate(ENG_RANDOM, glxo_draw64(seed, call_counter, rr, (uint32_t)j + a * (uint32_t)k),
                                           (uint64_t)deg);
          if (!filter_hit(f, i, rn, rt, d) || --retry < 0) {
            nbr[j] = rn[d];
            eid[j] = re[d];
            retry = f->retry_times;
            break;
          }
        }
      }
      continue;
    }
    int32_t cnt = glxo_filter_act_on(f, i, rn, rt, deg, res);
    if (op == GLXO_TOPK) {
      for (int32_t t = 0; t < cnt; ++t) idx[t] = res[t];
      pad_row(rn, re, deg, idx, cnt, k, padding_mode, default_neighbor_id, nbr, eid);
    } else if (op == GLXO_RANDOM_WITHOUT_REPLACEMENT) {
      /* shuffle(reserved) under the contract's forward Fisher-Yates, then Pad */
      for (int32_t t = 0; t < cnt; ++t) idx[t] = res[t];
      if (g_ref_entropy) {
        if (cnt > 0) ref_shuffle(&g_ref_eng[ENG_RWOR], idx, cnt); /* std::shuffle of an empty range draws nothing */
      } else if (padding_mode == GLXO_PAD_CIRCULAR) {
        int32_t steps = cnt < k ? cnt : k;
        for (int32_t j = 0; j < steps; ++j) {
          int64_t r = j + (int64_t)bounded(glxo_draw64(seed, call_counter, rr, (uint32_t)j), (uint64_t)(cnt - j));
          int64_t t = idx[j]; idx[j] = idx[r]; idx[r] = t;
        }
      }
      pad_row(rn, re, deg, idx, cnt, k, padding_mode, default_neighbor_id, nbr, eid);
    } else {
      /* SampleFromIndices: alias table over the reserved neighbours' weights, k draws,
       * mapped back through the reserved list; an empty list stays empty */
      if (cnt == 0) {
        pad_row(rn, re, deg, idx, 0, k, padding_mode, default_neighbor_id, nbr, eid);
        continue;
      }
      const float* w = op == GLXO_EDGE_WEIGHT ? g->weight + start : f->indeg_weight + start;
      for (int32_t t = 0; t < cnt; ++t) dist[t] = w[res[t]];
      float* probs = dist + maxdeg;
      alias_build_row(dist, cnt, probs, tab, tab + maxdeg, tab + 2 * maxdeg);
      for (int32_t j = 0; j < k; ++j) {
        float rnd = alias_variate(glxo_draw64(seed, call_counter, rr, (uint32_t)j), (double)(cnt - 1));
        int32_t ix = (int32_t)rnd;
        idx[j] = res[(probs[ix] <= (rnd - ix)) ? tab[ix] : ix];
      }
      pad_row(rn, re, deg, idx, k, k, padding_mode, default_neighbor_id, nbr, eid);
    }
  }
  free(res); free(idx); free(dist); free(tab);
  if (g->ids) idmap_free(&m);
  return 0;
}

int64_t glxo_sample_full_filtered(const glxo_graph* g, const int64_t* src, int32_t batch, int32_t max_limit,
                                  int padding_mode, int64_t default_neighbor_id, const glxo_filter* f,
                                  int32_t* degrees_out, int64_t* nbr_out, int64_t* eid_out, int64_t cap) {
  if (!f || f->type == GLXO_FILTER_NONE)
    return glxo_sample_full(g, src, batch, max_limit, degrees_out, nbr_out, eid_out, cap);
  idmap m;
  if (g->ids) idmap_build(&m, g->ids, g->V);
  int64_t maxdeg = 1;
  for (int64_t r = 0; r < g->V; ++r)
    if (g->row_ptr[r + 1] - g->row_ptr[r] > maxdeg) maxdeg = g->row_ptr[r + 1] - g->row_ptr[r];
  int32_t* res = (int32_t*)malloc(sizeof(int32_t) * (size_t)maxdeg);
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)maxdeg);
  int64_t* tn = (int64_t*)malloc(sizeof(int64_t) * (size_t)maxdeg * 2);
  int64_t total = 0;
  for (int32_t i = 0; i < batch; ++i) {
    int64_t row = row_of(g->ids, &m, g->V, src[i]);
    int64_t start = row < 0 ? 0 : g->row_ptr[row];
    int32_t deg = row < 0 ? 0 : (int32_t)(g->row_ptr[row + 1] - start);
    int32_t take = (max_limit > 0 && max_limit < deg) ? max_limit : deg;
    degrees_out[i] = take;
    if (deg == 0) continue;
    const int64_t* rt = f->ts_slot ? f->ts_slot + start : NULL;
    int32_t cnt = glxo_filter_act_on(f, i, g->col + start, rt, deg, res);
    for (int32_t t = 0; t < cnt; ++t) idx[t] = res[t];
    pad_row(g->col + start, g->eid + start, deg, idx, cnt, take, padding_mode, default_neighbor_id, tn, tn + maxdeg);
    for (int32_t j = 0; j < take; ++j) {
      if (total < cap) { nbr_out[total] = tn[j]; eid_out[total] = tn[maxdeg + j]; }
      ++total;
    }
  }
  free(res); free(idx); free(tn);
  if (g->ids) idmap_free(&m);
  return total;
}

/* ------------------------------------------------------------ aggregators -- */
int glxo_aggregate(const float* feats, int64_t V, int32_t dim, const int64_t* ids, int op,
                   const int64_t* node_ids, const int32_t* segment_ids, int32_t num_ids,
                   int32_t num_segments, float default_attr, float* emb_out, int32_t* cnt_out) {
  if (op < GLXO_SUM || op > GLXO_PROD) return 3;
  idmap m;
  if (ids) idmap_build(&m, ids, V);
  float* defrow = (float*)malloc(sizeof(float) * (size_t)(dim > 0 ? dim : 1));
  for (int32_t i = 0; i < dim; ++i) defrow[i] = default_attr;
  int32_t cursor = 0; /* AggregatingRequest::cursor_, aggregating_request.cc:86-105 */
  for (int32_t s = 0; s < num_segments; ++s) {
    float* emb = emb_out + (int64_t)s * dim;
    /* InitFunc: aggregator.cc:61-65; max_aggregator.cc:26-30 (FLT_MIN_10_EXP =
     * -37); min_aggregator.cc:26-30; prod_aggregator.cc. */
    float init = 0.0f;
    if (op == GLXO_MAX) init = (float)FLT_MIN_10_EXP;
    if (op == GLXO_MIN) init = FLT_MAX;
    if (op == GLXO_PROD) init = 1.0f;
    for (int32_t i = 0; i < dim; ++i) emb[i] = init;
    int32_t n = 0;
    while (cursor < num_ids && segment_ids[cursor] == s) {
      int64_t row = row_of(ids, &m, V, node_ids[cursor]);
      const float* a = row < 0 ? defrow : feats + row * (int64_t)dim;
      ++cursor;
      switch (op) {
        case GLXO_SUM:
        case GLXO_MEAN:
          for (int32_t i = 0; i < dim; ++i) emb[i] = emb[i] + a[i];
          break;
        case GLXO_MAX: /* std::max(l, r) == (l < r) ? r : l */
          for (int32_t i = 0; i < dim; ++i) emb[i] = (emb[i] < a[i]) ? a[i] : emb[i];
          break;
        case GLXO_MIN: /* std::min(l, r) == (r < l) ? r : l */
          for (int32_t i = 0; i < dim; ++i) emb[i] = (a[i] < emb[i]) ? a[i] : emb[i];
          break;
        case GLXO_PROD:
          for (int32_t i = 0; i < dim; ++i) emb[i] = emb[i] * a[i];
          break;
      }
      ++n;
    }
    /* FinalFunc: aggregator.cc:74-86; mean_aggregator.cc:45-61. */
    if (n == 0) {
      for (int32_t i = 0; i < dim; ++i) emb[i] = default_attr;
    } else if (op == GLXO_MEAN) {
      for (int32_t i = 0; i < dim; ++i) emb[i] = emb[i] / n;
    }
    cnt_out[s] = n;
  }
  free(defrow);
  if (ids) idmap_free(&m);
  return 0;
}

/* ------------------------------------------------------ partition / stitch -- */
void glxo_partition(const int64_t* ids, int64_t n, int32_t P, int64_t* order_out,
                    int64_t* counts_out) {
  for (int32_t p = 0; p < P; ++p) counts_out[p] = 0;
  for (int64_t i = 0; i < n; ++i) counts_out[llabs(ids[i]) % P]++;
  int64_t* off = (int64_t*)malloc(sizeof(int64_t) * (size_t)P);
  int64_t acc = 0;
  for (int32_t p = 0; p < P; ++p) { off[p] = acc; acc += counts_out[p]; }
  for (int64_t i = 0; i < n; ++i) order_out[off[llabs(ids[i]) % P]++] = i;
  free(off);
}

void glxo_stitch_i64(const int64_t* shard_major, const int64_t* order, int64_t n, int32_t width,
                     int64_t* out) {
  for (int64_t i = 0; i < n; ++i)
    memcpy(out + order[i] * width, shard_major + i * width, sizeof(int64_t) * (size_t)width);
}

int glxo_aggregate_stitch(int op, int32_t P, const float* parts, const int32_t* cnts,
                          int32_t num_segments, int32_t dim, float default_attr, int reference_fold,
                          float* emb_out, int32_t* cnt_out) {
  if (op < GLXO_SUM || op > GLXO_PROD) return 3;
  float init = 0.0f; /* InitFunc, as in glxo_aggregate */
  if (op == GLXO_MAX) init = (float)FLT_MIN_10_EXP;
  if (op == GLXO_MIN) init = FLT_MAX;
  if (op == GLXO_PROD) init = 1.0f;
  for (int32_t s = 0; s < num_segments; ++s) {
    float* emb = emb_out + (int64_t)s * dim;
    for (int32_t i = 0; i < dim; ++i) emb[i] = init;
    int32_t total = 0;
    for (int32_t p = 0; p < P; ++p) { /* shards->Next(): ascending shard id */
      const int32_t c = cnts[(int64_t)p * num_segments + s];
      const float* a = parts + ((int64_t)p * num_segments + s) * dim;
      if (c == 0 && !reference_fold) continue;
      switch (op) {
        case GLXO_SUM:
          for (int32_t i = 0; i < dim; ++i) emb[i] = emb[i] + a[i];
          break;
        case GLXO_MEAN: /* left += right * segments[i] (int -> float) */
          for (int32_t i = 0; i < dim; ++i) {
            const float w = a[i] * (float)c;
            emb[i] = emb[i] + w;
          }
          break;
        case GLXO_MAX:
          for (int32_t i = 0; i < dim; ++i) emb[i] = (emb[i] < a[i]) ? a[i] : emb[i];
          break;
        case GLXO_MIN:
          for (int32_t i = 0; i < dim; ++i) emb[i] = (a[i] < emb[i]) ? a[i] : emb[i];
          break;
        case GLXO_PROD:
          for (int32_t i = 0; i < dim; ++i) emb[i] = emb[i] * a[i];
          break;
      }
      total += c;
    }
    if (total == 0) {
      for (int32_t i = 0; i < dim; ++i) emb[i] = default_attr;
    } else if (op == GLXO_MEAN) {
      for (int32_t i = 0; i < dim; ++i) emb[i] = emb[i] / (float)total;
    }
    cnt_out[s] = total;
  }
  return 0;
}

int64_t glxo_dst_statics(const int64_t* col, const int64_t* eid, int64_t E, int64_t* ids_out,
                         int32_t* in_degrees_out) {
  /* replay the insertions in edge-id order */
  int64_t* by_eid = (int64_t*)malloc(sizeof(int64_t) * (size_t)(E > 0 ? E : 1));
  int64_t max_eid = -1;
  for (int64_t i = 0; i < E; ++i) {
    if (eid[i] > max_eid) max_eid = eid[i];
  }
  int64_t* slot_of = (int64_t*)malloc(sizeof(int64_t) * (size_t)(max_eid + 2));
  for (int64_t i = 0; i <= max_eid; ++i) slot_of[i] = -1;
  for (int64_t i = 0; i < E; ++i) slot_of[eid[i]] = i;
  int64_t n = 0;
  for (int64_t e = 0; e <= max_eid; ++e) {
    if (slot_of[e] >= 0) by_eid[n++] = col[slot_of[e]];
  }
  int64_t U = 0;
  /* dst id -> index: the role of dst_indexing_ */
  int64_t cap = 16;
  while (cap < 2 * n + 2) cap <<= 1;
  int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap);
  int32_t* vals = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
  for (int64_t i = 0; i < cap; ++i) vals[i] = -1;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t d = by_eid[i];
    uint64_t h = (uint64_t)d * 0x9E3779B97F4A7C15ull;
    int64_t at = (int64_t)(h >> 7) & (cap - 1);
    while (vals[at] >= 0 && keys[at] != d) at = (at + 1) & (cap - 1);
    if (vals[at] < 0) { /* new coming */
      keys[at] = d;
      vals[at] = (int32_t)U;
      ids_out[U] = d;
      in_degrees_out[U] = 1;
      ++U;
    } else { /* has appeared before */
      in_degrees_out[vals[at]]++;
    }
  }
  free(keys);
  free(vals);
  free(slot_of);
  free(by_eid);
  return U;
}

int glxo_negative_sample(const int64_t* ids, int64_t U, const float* prob, const int32_t* alias, int exclude,
                         const glxo_graph* g, const int64_t* src, int32_t batch, int32_t count,
                         int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter, int64_t* out) {
  if (exclude < 0 || exclude > 2) return 3;
  if (U == 0) { /* res->FillWith(DefaultNeighborId, -1) */
    for (int64_t i = 0; i < (int64_t)batch * count; ++i) out[i] = default_neighbor_id;
    return 0;
  }
  idmap m;
  if (exclude == 1 && g->ids) idmap_build(&m, g->ids, g->V);
  int32_t* indices = (int32_t*)malloc(sizeof(int32_t) * (size_t)(count > 0 ? count : 1));
  /* exclude == 2 (NodeWeightNegativeSampler): the set of the request's own ids is ONE object for all rows
   * (node_weight_negative_sampler.cc:68), so the sets.clear() of the first row that exhausts its retries (:80) frees
   * every later row from it too.  exclude == 1 builds a set per row (in_degree_negative_sampler.cc:70-73). */
  int batch_set_alive = 1;
  for (int32_t i = 0; i < batch; ++i) {
    /* the exclusion set of this row */
    const int64_t* ex = NULL;
    int64_t exn = 0;
    if (exclude == 1) {
      int64_t row = row_of(g->ids, &m, g->V, src[i]);
      if (row >= 0) {
        ex = g->col + g->row_ptr[row];
        exn = g->row_ptr[row + 1] - g->row_ptr[row];
      }
    } else if (exclude == 2) {
      ex = src;
      exn = batch;
    }
    int set_active = exclude == 2 ? batch_set_alive : exclude != 0;
    int32_t taken = 0, cursor = 0, blk = 0;
    int32_t retry_times = 3 + 1; /* kRetryTimes + 1 */
    if (exclude == 0) { /* one block, every candidate is taken */
      for (int32_t j = 0; j < count; ++j) {
        uint64_t u = glxo_draw64(seed, call_counter, (uint32_t)i, (uint32_t)j);
        int64_t ix;
        if (prob) {
          float rnd = alias_variate(u, (double)(U - 1));
          int32_t k = (int32_t)rnd;
          ix = (prob[k] <= (rnd - k)) ? alias[k] : k;
        } else {
          ix = int_variate(ENG_RANDOM_NEGATIVE, u, (uint64_t)U);
        }
        out[(int64_t)i * count + j] = ids[ix];
      }
      continue;
    }
    while (taken < count && retry_times >= 0) {
      cursor %= count;
      if (cursor == 0) {
        for (int32_t j = 0; j < count; ++j) { /* am->Sample(n, indices) */
          uint64_t u = glxo_draw64(seed, call_counter, (uint32_t)i, (uint32_t)(blk * count + j));
          if (prob) {
            float rnd = alias_variate(u, (double)(U - 1));
            int32_t k = (int32_t)rnd;
            indices[j] = (prob[k] <= (rnd - k)) ? alias[k] : k;
          } else {
            indices[j] = (int32_t)int_variate(ENG_RANDOM_NEGATIVE, u, (uint64_t)U);
          }
        }
        ++blk;
        if (--retry_times <= 0) { /* sets.clear() */
          set_active = 0;
          if (exclude == 2) batch_set_alive = 0;
        }
      }
      int64_t item = ids[indices[cursor++]];
      int found = 0;
      if (set_active) {
        for (int64_t e = 0; e < exn; ++e) {
          if (ex[e] == item) {
            found = 1;
            break;
          }
        }
      }
      if (!found) out[(int64_t)i * count + taken++] = item;
    }
  }
  free(indices);
  if (exclude == 1 && g->ids) idmap_free(&m);
  return 0;
}

void glxo_sort_rows_by_timestamp_asc(const int64_t* row_ptr, int64_t V, int64_t* col, int64_t* eid,
                                     int64_t* ts_slot, float* weight) {
  for (int64_t r = 0; r < V; ++r) {
    const int64_t s = row_ptr[r], e = row_ptr[r + 1];
    /* stable insertion sort: rows are short in the tests this oracle serves */
    for (int64_t i = s + 1; i < e; ++i) {
      const int64_t t = ts_slot[i], c = col[i], d = eid[i];
      const float w = weight ? weight[i] : 0.0f;
      int64_t j = i - 1;
      while (j >= s && ts_slot[j] > t) {
        ts_slot[j + 1] = ts_slot[j];
        col[j + 1] = col[j];
        eid[j + 1] = eid[j];
        if (weight) weight[j + 1] = weight[j];
        --j;
      }
      ts_slot[j + 1] = t;
      col[j + 1] = c;
      eid[j + 1] = d;
      if (weight) weight[j + 1] = w;
    }
  }
}

/* ------------------------------------------------------------- sub-graph -- */
/* SubGraphSampler::InduceSubGraph (subgraph/subgraph_sampler.cc:34-95) on FullSampler's response rows of `nodes`
 * (offsets[n + 1], nbr, eid): for node i a map neighbour id -> edge id where a later slot overwrites an earlier one
 * (:60-64); then for every j in list order with nodes[j] in the map: AppendEdge(i, j, eid), AppendEdge(j, i, eid)
 * (:66-70).  Returns the number of entries (2 per match); writes the first `cap`. */
int64_t glxo_subgraph_induce(const int64_t* nodes, int32_t n, const int64_t* offsets, const int64_t* nbr, const int64_t* eid,
                             int32_t* row_out, int32_t* col_out, int64_t* eid_out, int64_t cap) {
  int64_t total = 0;
  for (int32_t i = 0; i < n; ++i) {
    const int64_t o0 = offsets[i], deg = offsets[i + 1] - o0;
    for (int32_t j = 0; j < n; ++j) {
      int64_t found = -1;
      for (int64_t k = 0; k < deg; ++k) {
        if (nbr[o0 + k] == nodes[j]) found = k; /* the last slot wins, as node2edge[nbrs[k]] = edge_ids[k] does */
      }
      if (found < 0) continue;
      if (total + 1 < cap) {
        row_out[total] = i; col_out[total] = j; eid_out[total] = eid[o0 + found];
        row_out[total + 1] = j; col_out[total + 1] = i; eid_out[total + 1] = eid[o0 + found];
      }
      total += 2;
    }
  }
  return total;
}

/* BFSShortestPath (subgraph/subgraph_utils.cc:36-57) over the induced edges with node `skip` removed; unreachable =
 * INT32_MAX.  dist_out[n]. */
static void bfs_without(int32_t n, const int32_t* row, const int32_t* col, int64_t m, int32_t skip, int32_t start,
                        int32_t* dist_out) {
  int32_t* queue = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  for (int32_t i = 0; i < n; ++i) dist_out[i] = INT32_MAX;
  int32_t head = 0, tail = 0;
  dist_out[start] = 0;
  queue[tail++] = start;
  while (head < tail) {
    const int32_t s = queue[head++];
    for (int64_t e = 0; e < m; ++e) {
      if (row[e] != s || row[e] == skip || col[e] == skip) continue;
      if (dist_out[col[e]] == INT32_MAX && col[e] != start) {
        dist_out[col[e]] = dist_out[s] + 1;
        queue[tail++] = col[e];
      }
    }
  }
  free(queue);
}

/* The need_dist half of InduceSubGraph (:71-93): src = node 0, dst = node 1; dist_to_dst = BFS from dst in the
 * graph without src (then dist_to_dst[src] = 0), dist_to_src = BFS from src without dst (dist_to_src[dst] = 0). */
void glxo_subgraph_dist(int32_t n, const int32_t* row, const int32_t* col, int64_t m, int32_t* dist_to_src,
                        int32_t* dist_to_dst) {
  if (n < 2) {
    for (int32_t i = 0; i < n; ++i) { dist_to_src[i] = 0; dist_to_dst[i] = 0; }
    return;
  }
  bfs_without(n, row, col, m, /*skip=*/0, /*start=*/1, dist_to_dst);
  bfs_without(n, row, col, m, /*skip=*/1, /*start=*/0, dist_to_src);
  dist_to_dst[0] = 0;
  dist_to_src[1] = 0;
}

/* ------------------------------------------- ConditionalNegativeSampler -- */
/* core/operator/sampler/conditional_negative_sampler.cc:37-161 over condition_table.cc:65-148 and
 * attribute_nodes_map.h:74-127, under the glx seeding contract.
 *
 * Candidates ids[U] with weights[U] (NULL = 1.0f each: the "random" strategy's ConditionTable has no weights and its
 * default AliasMethod(ids.Size()) is built over ones, alias_method.cc:36-39).  Column c of the condition table groups
 * the candidates by cand_keys[c * U + u] (the attribute value as an int64 key: an int attribute, a float's bits, a
 * string's dictionary id); every group has its own alias table over the members' weights (CreateAM).  Request row i
 * carries dst_keys[i * ncols + c] (GLXO_NO_KEY matches no group: "when there is no this attr at all, just skip").
 *
 * The exclusion set lives ACROSS the rows of a request (nbr_set is declared before the row loop and never cleared,
 * conditional_negative_sampler.cc:105-123): batch_share = all dst ids up front; otherwise row i adds src i's neighbours
 * and dst i before it samples.  unique also adds every accepted id.
 *
 * Per row, per column c: num_c = (int32)(count * props[c]); AttributeNodesMap::Sample draws blocks of num_c alias
 * indices, at most `retry` blocks, walks a block in order and accepts what is not in the set -- but its loop condition
 * re-tests retry_times > 0 after the LAST block has been drawn, so of block `retry` only the first entry is looked at
 * (attribute_nodes_map.h:109-125).  Draw d of row i is word d of the stream (seed, call_counter, i); column c, block b,
 * position j is draw retry * (num_0 + .. + num_{c-1}) + b * num_c + j.
 *
 * The reference then means to fill the row up to `count` from the default alias table (:128-152), but computes how many
 * it has from the STATIC response shape (res->GetShape().size - idx * num = (batch - idx) * num >= num), so that loop
 * never runs and a row that came up short leaves the response tensor short and misaligned.  Like the out-of-bounds read
 * of SURVEY 8(a).3 this is not reproduced: the fill loop runs as written -- blocks of `count` default draws starting at
 * draw retry * (num_0 + ..); retry + 1 blocks against the set, then the set is dropped for good (nbr_set.clear(), :140)
 * and up to two more blocks are taken; whatever is still missing is default_neighbor_id.  Where the reference's response
 * is complete (every column found its num_c and they add up to count) the two agree in distribution. */
typedef struct { int64_t* keys; uint8_t* used; uint64_t mask; } i64set;
static void i64set_init(i64set* s, uint64_t cap_hint) {
  uint64_t cap = 16;
  while (cap < 2 * cap_hint + 2) cap <<= 1;
  s->keys = (int64_t*)malloc(sizeof(int64_t) * cap);
  s->used = (uint8_t*)calloc(cap, 1);
  s->mask = cap - 1;
}
static void i64set_clear(i64set* s) { memset(s->used, 0, (size_t)(s->mask + 1)); }
static void i64set_free(i64set* s) { free(s->keys); free(s->used); }
static int i64set_has(const i64set* s, int64_t v) {
  uint64_t h = mix64((uint64_t)v) & s->mask;
  while (s->used[h]) {
    if (s->keys[h] == v) return 1;
    h = (h + 1) & s->mask;
  }
  return 0;
}
static void i64set_add(i64set* s, int64_t v) {
  uint64_t h = mix64((uint64_t)v) & s->mask;
  while (s->used[h]) {
    if (s->keys[h] == v) return;
    h = (h + 1) & s->mask;
  }
  s->used[h] = 1;
  s->keys[h] = v;
}

typedef struct { int64_t key; int64_t pos; } keypos;
static int cmp_keypos(const void* a, const void* b) {
  const keypos* x = (const keypos*)a; const keypos* y = (const keypos*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}

static int32_t alias_draw(uint64_t u, int64_t n, const float* prob, const int32_t* alias) {
  /* AliasMethod::Sample (alias_method.cc:117-121) */
  float rnd = alias_variate(u, (double)(n - 1));
  int32_t k = (int32_t)rnd;
  return (prob[k] <= (rnd - k)) ? alias[k] : k;
}

int glxo_cond_negative_sample(const int64_t* ids, const float* weights, int64_t U, int32_t ncols, const int64_t* cand_keys,
                              const float* props, const glxo_graph* g, const int64_t* src, const int64_t* dst,
                              const int64_t* dst_keys, int32_t batch, int32_t count, int batch_share, int unique,
                              int32_t retry, int64_t default_neighbor_id, uint64_t seed, uint64_t call_counter,
                              int64_t* out, int32_t* filled_by_columns_out) {
  if (batch <= 0 || count <= 0) return 0;
  int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(U > 0 ? U : 1)); /* low / high stacks of Build */
  /* condition table: per column, members sorted by (key, candidate position); groups = runs of equal keys */
  keypos** order = (keypos**)malloc(sizeof(keypos*) * (size_t)(ncols > 0 ? ncols : 1));
  float** gprob = (float**)malloc(sizeof(float*) * (size_t)(ncols > 0 ? ncols : 1));
  int32_t** galias = (int32_t**)malloc(sizeof(int32_t*) * (size_t)(ncols > 0 ? ncols : 1));
  for (int32_t c = 0; c < ncols; ++c) {
    order[c] = (keypos*)malloc(sizeof(keypos) * (size_t)(U > 0 ? U : 1));
    gprob[c] = (float*)malloc(sizeof(float) * (size_t)(U > 0 ? U : 1));
    galias[c] = (int32_t*)malloc(sizeof(int32_t) * (size_t)(U > 0 ? U : 1));
    for (int64_t u = 0; u < U; ++u) { order[c][u].key = cand_keys[(int64_t)c * U + u]; order[c][u].pos = u; }
    qsort(order[c], (size_t)U, sizeof(keypos), cmp_keypos);
    float* w = (float*)malloc(sizeof(float) * (size_t)(U > 0 ? U : 1));
    for (int64_t a = 0; a < U;) {
      int64_t b = a;
      while (b < U && order[c][b].key == order[c][a].key) ++b;
      for (int64_t t = a; t < b; ++t) w[t - a] = weights ? weights[order[c][t].pos] : 1.0f;
      alias_build_row(w, (int32_t)(b - a), gprob[c] + a, galias[c] + a, stack, stack + U);
      a = b;
    }
    free(w);
  }
  /* default alias table over all candidates */
  float* dprob = (float*)malloc(sizeof(float) * (size_t)(U > 0 ? U : 1));
  int32_t* dalias = (int32_t*)malloc(sizeof(int32_t) * (size_t)(U > 0 ? U : 1));
  {
    float* w = (float*)malloc(sizeof(float) * (size_t)(U > 0 ? U : 1));
    for (int64_t u = 0; u < U; ++u) w[u] = weights ? weights[u] : 1.0f;
    if (U > 0) alias_build_row(w, (int32_t)U, dprob, dalias, stack, stack + U);
    free(w);
  }
  idmap m;
  if (g && g->ids) idmap_build(&m, g->ids, g->V);
  uint64_t cap = (uint64_t)batch * (uint64_t)(count + 1);
  if (g) {
    for (int32_t i = 0; i < batch; ++i) {
      int64_t row = row_of(g->ids, &m, g->V, src[i]);
      if (row >= 0) cap += (uint64_t)(g->row_ptr[row + 1] - g->row_ptr[row]);
    }
  }
  i64set S;
  i64set_init(&S, cap);
  if (batch_share) for (int32_t i = 0; i < batch; ++i) i64set_add(&S, dst[i]);
  int32_t* num_c = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ncols > 0 ? ncols : 1));
  int32_t* block_ix = NULL;
  int32_t block_cap = 0;
  for (int32_t i = 0; i < batch; ++i) {
    if (!batch_share) {
      if (g) {
        int64_t row = row_of(g->ids, &m, g->V, src[i]);
        if (row >= 0) for (int64_t e = g->row_ptr[row]; e < g->row_ptr[row + 1]; ++e) i64set_add(&S, g->col[e]);
      }
      i64set_add(&S, dst[i]);
    }
    int64_t* orow = out + (int64_t)i * count;
    int32_t taken = 0;
    uint32_t base = 0;
    for (int32_t c = 0; c < ncols; ++c) {
      const int32_t n = (int32_t)((float)count * props[c]);
      num_c[c] = n;
      if (n <= 0) continue;
      /* the group of this row's key */
      const int64_t key = dst_keys[(int64_t)i * ncols + c];
      int64_t a = -1, b = -1;
      if (key != GLXO_NO_KEY) {
        int64_t lo = 0, hi = U;
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (order[c][mid].key < key) lo = mid + 1; else hi = mid; }
        if (lo < U && order[c][lo].key == key) { a = lo; b = lo; while (b < U && order[c][b].key == key) ++b; }
      }
      if (a >= 0) {
        int32_t got = 0;
        for (int32_t blk = 0; blk < retry && got < n; ++blk) {
          const int32_t look = (blk == retry - 1) ? 1 : n; /* the last block: its first entry only */
          if (g_ref_entropy) {
            /* am->Sample(num, indices) draws the WHOLE block before its entries are looked at (attribute_nodes_map.h:
             * 111-116): a sequential engine moves on by n variates per started block, used or not.  (Under the contract
             * every position has its own draw, so unused ones simply are not computed.) */
            if (n > block_cap) {
              block_cap = n;
              block_ix = (int32_t*)realloc(block_ix, sizeof(int32_t) * (size_t)block_cap);
            }
            for (int32_t j = 0; j < n; ++j) block_ix[j] = alias_draw(0, b - a, gprob[c] + a, galias[c] + a);
          }
          for (int32_t j = 0; j < look && got < n; ++j) {
            uint64_t u = glxo_draw64(seed, call_counter, (uint32_t)i, base + (uint32_t)(blk * n + j));
            int32_t ix = g_ref_entropy ? block_ix[j] : alias_draw(u, b - a, gprob[c] + a, galias[c] + a);
            int64_t item = ids[order[c][a + ix].pos];
            if (!i64set_has(&S, item)) {
              if (taken < count) orow[taken] = item;
              ++taken; ++got;
              if (unique) i64set_add(&S, item);
            }
          }
        }
      }
      base += (uint32_t)retry * (uint32_t)n;
    }
    if (taken > count) taken = count; /* props adding up to more than 1: the row is cut at count */
    if (filled_by_columns_out) filled_by_columns_out[i] = taken; /* == count: the reference's response row is complete too */
    /* default sampling (:128-152, as written) */
    if (U > 0) {
      int32_t retry_times = retry + 1, blk = 0;
      int last = 0;
      while (taken < count && !last) {
        if (--retry_times <= 0) i64set_clear(&S);
        if (retry_times < 0) last = 1; /* the loop condition fails after the first entry of this block */
        const int32_t look = last ? 1 : count;
        for (int32_t j = 0; j < look && taken < count; ++j) {
          uint64_t u = glxo_draw64(seed, call_counter, (uint32_t)i, base + (uint32_t)(blk * count + j));
          int64_t item = ids[alias_draw(u, U, dprob, dalias)];
          if (!i64set_has(&S, item)) {
            orow[taken++] = item;
            if (unique) i64set_add(&S, item);
          }
        }
        ++blk;
      }
    }
    for (; taken < count; ++taken) orow[taken] = default_neighbor_id;
  }
  free(num_c);
  free(block_ix);
  i64set_free(&S);
  if (g && g->ids) idmap_free(&m);
  free(dprob); free(dalias); free(stack);
  for (int32_t c = 0; c < ncols; ++c) { free(order[c]); free(gprob[c]); free(galias[c]); }
  free(order); free(gprob); free(galias);
  return 0;
}
