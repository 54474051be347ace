/*
 * bh_oracle.c — CPU restatement of the reference's per-step Barnes-Hut path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED — see bh_oracle.h for both statements.
 *
 * Every function cites the lines of /root/reference/nbody_v5_bench.cu ("ref:") it
 * follows.  Where the literal reference is degenerate the
 * *intended* recurrence is restated and the deviation is named.
 *
 * Floating point: all force/integrate arithmetic is the reference's source text
 * evaluated in IEEE binary32 with NO contraction (build with -ffp-contract=off);
 * sqrtf and '/' are correctly rounded.  The GPU "strict_fp" kernels use the same
 * sequence of operations and must agree bit for bit.
 */
#include "bh_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#else
static double omp_get_wtime(void) { return 0.0; }
static int omp_get_max_threads(void) { return 1; }
#endif

void bho_default_params(bho_params* p) {
  p->G = 0.5f;          /* ref:14 */
  p->theta = 0.5f;      /* ref:15 */
  p->dt = 0.02f;        /* ref:16 */
  p->eps2 = 50.0f;      /* ref:17 */
  p->max_speed = 500.0f;/* ref:18 */
  p->leaf_cap = 1;      /* ref:100-124: a slot holds one body, a second one splits it */
  p->max_depth = 21;
  p->key_bits = 63;
  p->compress = 1;
  p->key_curve = 0;
}

int bho_max_threads(void) { return omp_get_max_threads(); }

/* ------------------------------------------------------------------ bbox */
/* ref:134-156: serial min/max from sentinels +-1e10, cube anchored at the min corner */
static void bbox_strided(const float* x, const float* y, const float* z, int stride, int n,
                         float bounds[6]) {
  float minX = 1e10f, minY = 1e10f, minZ = 1e10f, maxX = -1e10f, maxY = -1e10f, maxZ = -1e10f;
  for (int i = 0; i < n; i++) {
    size_t j = (size_t)i * stride;
    minX = fminf(minX, x[j]);
    minY = fminf(minY, y[j]);
    minZ = fminf(minZ, z[j]);
    maxX = fmaxf(maxX, x[j]);
    maxY = fmaxf(maxY, y[j]);
    maxZ = fmaxf(maxZ, z[j]);
  }
  float size = fmaxf(maxX - minX, fmaxf(maxY - minY, maxZ - minZ)); /* ref:148 */
  bounds[0] = minX;
  bounds[1] = minY;
  bounds[2] = minZ;
  bounds[3] = minX + size;
  bounds[4] = minY + size;
  bounds[5] = minZ + size;
}

void bho_bbox(const float* x, const float* y, const float* z, int n, float bounds[6]) {
  bbox_strided(x, y, z, 1, n, bounds);
}

/* ------------------------------------------------------------------ keys */
/* ref:42-49 */
static uint32_t expand_bits10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

/* ref:51-63, literal */
void bho_morton30(const float* x, const float* y, const float* z, const float bounds[6], int n,
                  uint32_t* codes, int32_t* indices) {
  float minX = bounds[0], minY = bounds[1], minZ = bounds[2];
  float size = fmaxf(bounds[3] - bounds[0], 1.0f);
  for (int i = 0; i < n; i++) {
    uint32_t xx = (uint32_t)((x[i] - minX) / size * 1023.0f);
    uint32_t yy = (uint32_t)((y[i] - minY) / size * 1023.0f);
    uint32_t zz = (uint32_t)((z[i] - minZ) / size * 1023.0f);
    codes[i] = (expand_bits10(xx) << 2) | (expand_bits10(yy) << 1) | expand_bits10(zz);
    if (indices) indices[i] = i;
  }
}

float bho_root_edge(const float bounds[6]) { return fmaxf(bounds[3] - bounds[0], 1.0f); /* ref:55 */ }

/* bit-by-bit interleave (deliberately NOT the magic-mask form the GPU uses):
   bit j of q lands at bit 3j */
static uint64_t spread3(uint32_t q, int bits) {
  uint64_t r = 0;
  for (int j = 0; j < bits; j++) r |= (uint64_t)((q >> j) & 1u) << (3 * j);
  return r;
}

/* Hilbert numbering of the 2^21-per-axis cell grid (bh_params.key_curve = 1; NOT in the reference, which uses the
   Morton / Z order — the cells and therefore the octree are the same, only their order along the key axis
   changes).  J. Skilling, "Programming the Hilbert curve", AIP Conf. Proc. 707 (2004), restated in its original
   array form with explicit loops over the dimensions (the GPU has the three axes unrolled in registers):
   AxesToTranspose turns cell coordinates into the "transposed" index, whose bit-interleave — X[0] supplying the
   most significant bit of each triple — is the position along the curve. */
#define HB 21
static void axes_to_transpose(uint32_t X[3]) {
  const uint32_t M = 1u << (HB - 1);
  for (uint32_t Q = M; Q > 1; Q >>= 1) { /* inverse undo of the excess work */
    const uint32_t P = Q - 1;
    for (int i = 0; i < 3; i++) {
      if (X[i] & Q) {
        X[0] ^= P; /* invert the low bits of X[0] */
      } else {     /* exchange the low bits of X[0] and X[i] */
        const uint32_t t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  for (int i = 1; i < 3; i++) X[i] ^= X[i - 1]; /* Gray encode */
  uint32_t t = 0;
  for (uint32_t Q = M; Q > 1; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1;
  for (int i = 0; i < 3; i++) X[i] ^= t;
}
static void transpose_to_axes(uint32_t X[3]) {
  const uint32_t N = 2u << (HB - 1);
  uint32_t t = X[2] >> 1; /* Gray decode by H ^ (H / 2) */
  for (int i = 2; i > 0; i--) X[i] ^= X[i - 1];
  X[0] ^= t;
  for (uint32_t Q = 2; Q != N; Q <<= 1) { /* undo the excess work */
    const uint32_t P = Q - 1;
    for (int i = 2; i >= 0; i--) {
      if (X[i] & Q) {
        X[0] ^= P;
      } else {
        const uint32_t u = (X[0] ^ X[i]) & P;
        X[0] ^= u;
        X[i] ^= u;
      }
    }
  }
}
uint64_t bho_hilbert_index(uint32_t x, uint32_t y, uint32_t z) {
  uint32_t X[3] = {x, y, z};
  axes_to_transpose(X);
  return (spread3(X[0], HB) << 2) | (spread3(X[1], HB) << 1) | spread3(X[2], HB);
}
void bho_hilbert_cell(uint64_t index, uint32_t xyz[3]) {
  uint32_t X[3] = {0, 0, 0};
  for (int j = 0; j < HB; j++) {
    X[0] |= (uint32_t)((index >> (3 * j + 2)) & 1u) << j;
    X[1] |= (uint32_t)((index >> (3 * j + 1)) & 1u) << j;
    X[2] |= (uint32_t)((index >> (3 * j)) & 1u) << j;
  }
  transpose_to_axes(X);
  xyz[0] = X[0]; xyz[1] = X[1]; xyz[2] = X[2];
}

/* key_bits 30: the reference quantisation (x1023, ref:56-58) -> identical to bho_morton30.
   key_bits 63: same normalisation, 21 bits/axis, scale 2^21 with an explicit clamp, so a
   cell of the key grid is exactly a cell of the repeated midpoint halving (ref:96-99,114-119).
   x is the most significant bit of each octal digit (ref:61). */
static void keys_strided(const float* x, const float* y, const float* z, int stride,
                         const float bounds[6], int n, int key_bits, int key_curve, uint64_t* keys) {
  const int b = key_bits / 3;
  const float minX = bounds[0], minY = bounds[1], minZ = bounds[2];
  const float size = fmaxf(bounds[3] - bounds[0], 1.0f);
  const float scale = (b == 10) ? 1023.0f : 2097152.0f;
  const uint32_t qmax = (1u << b) - 1u;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    size_t j = (size_t)i * stride;
    uint32_t xx = (uint32_t)((x[j] - minX) / size * scale);
    uint32_t yy = (uint32_t)((y[j] - minY) / size * scale);
    uint32_t zz = (uint32_t)((z[j] - minZ) / size * scale);
    if (xx > qmax) xx = qmax;
    if (yy > qmax) yy = qmax;
    if (zz > qmax) zz = qmax;
    keys[i] = (key_curve == 1 && b == HB) ? bho_hilbert_index(xx, yy, zz)
                                          : (spread3(xx, b) << 2) | (spread3(yy, b) << 1) | spread3(zz, b);
  }
}

void bho_keys(const float* x, const float* y, const float* z, const float bounds[6], int n,
              int key_bits, int key_curve, uint64_t* keys) {
  keys_strided(x, y, z, 1, bounds, n, key_bits, key_curve, keys);
}

/* ------------------------------------------------------------------ sort */
/* ref:262-264 thrust::sort_by_key(keys, indices): ascending, stable.  Restated as a
   top-down merge sort on (key, index) pairs (stable because ties take the left run). */
typedef struct {
  uint64_t k;
  int32_t v;
} kv_t;

static void merge_runs(const kv_t* a, int na, const kv_t* b, int nb, kv_t* out) {
  int i = 0, j = 0, o = 0;
  while (i < na && j < nb) out[o++] = (b[j].k < a[i].k) ? b[j++] : a[i++];
  while (i < na) out[o++] = a[i++];
  while (j < nb) out[o++] = b[j++];
}

static void msort(kv_t* a, kv_t* tmp, int n, int depth) {
  if (n <= 32) { /* insertion sort, stable */
    for (int i = 1; i < n; i++) {
      kv_t t = a[i];
      int j = i - 1;
      while (j >= 0 && a[j].k > t.k) {
        a[j + 1] = a[j];
        j--;
      }
      a[j + 1] = t;
    }
    return;
  }
  int h = n / 2;
#pragma omp task if (depth < 4) shared(a, tmp)
  msort(a, tmp, h, depth + 1);
#pragma omp task if (depth < 4) shared(a, tmp)
  msort(a + h, tmp + h, n - h, depth + 1);
#pragma omp taskwait
  merge_runs(a, h, a + h, n - h, tmp);
  memcpy(a, tmp, (size_t)n * sizeof(kv_t));
}

void bho_sort(const uint64_t* keys, int n, uint64_t* sorted_keys, int32_t* perm) {
  kv_t* a = (kv_t*)malloc((size_t)n * sizeof(kv_t));
  kv_t* t = (kv_t*)malloc((size_t)n * sizeof(kv_t));
  for (int i = 0; i < n; i++) {
    a[i].k = keys[i];
    a[i].v = i;
  }
#pragma omp parallel
#pragma omp single
  msort(a, t, n, 0);
  for (int i = 0; i < n; i++) {
    sorted_keys[i] = a[i].k;
    perm[i] = a[i].v;
  }
  free(a);
  free(t);
}

/* ------------------------------------------------------------------ build */
/* Intended tree of ref:83-132 with defects D3-D5 removed:
   the canonical octree over the bbox cube — a cell is subdivided iff it holds more
   than leaf_cap bodies and its level < max_depth; octant membership is the key digit
   (== the midpoint tests ref:96-99 up to rounding at cell faces); bodies in an unsplit
   cell are kept (the reference drops them at depth 25, ref:93,130).

   compress = 0: every such cell is emitted, including chains of cells with a single
                 non-empty octant (what repeated insertion ref:106-124 produces).
   compress = 1: a single-octant cell is replaced by its first descendant that branches (or
                 by the depth-capped leaf at level max_depth).  Same (mass, COM), smaller edge:
                 under `s/dist < theta` the chain is accepted iff that descendant is, so every
                 body sees exactly the same set of accepted cells and bodies; in pre-order
                 traversal the sum is even bit-identical (tests/test_oracle.py checks both).
                 This is the engine's tree: cells <= n-1 and records <= 2n for any input.

   Layout: entry 0 = root; an internal cell's children are consecutive entries in ascending
   digit order.  Child blocks are ordered by the cell's first child boundary (index of the
   first body of its second non-empty octant) when compress = 1 — the engine's order — and in
   depth-first pre-order when compress = 0. */
typedef struct {
  int lo, hi, level;      /* body range, level whose digit splits it */
  int nchild;
  int clo[8], chi[8];     /* child ranges, ascending digit */
  int ccell[8];           /* index of the child's cell, or -1 if the child is a leaf */
  int clevel[8];          /* level recorded for a leaf child (edge = s0 * 2^-level) */
  int sortkey, block;
} ocell;

typedef struct {
  const uint64_t* k;
  int B, cap, D, compress;
  ocell* cells;
  int ncells, cellcap;
  int max_level;
} bctx;

static int digit_at(const bctx* c, int j, int level) {
  return (int)((c->k[j] >> (3 * (c->B - 1 - level))) & 7u);
}

/* level at which [lo,hi) first has two different digits, scanning level by level; B if none */
static int branch_level(const bctx* c, int lo, int hi, int from) {
  for (int L = from; L < c->B; L++)
    if (digit_at(c, lo, L) != digit_at(c, hi - 1, L)) return L; /* keys sorted: ends differ iff any differ */
  return c->B;
}

/* classify the cell [lo,hi) that sits at `level` below its parent.
   returns 1 if internal (*split = level whose digit splits it), else 0 (*split = leaf level) */
static int classify(const bctx* c, int lo, int hi, int level, int* split) {
  const int n = hi - lo;
  *split = level;
  if (n == 1) return 0;
  if (n <= c->cap) return 0;
  if (!c->compress) return level < c->D;
  int Lb = branch_level(c, lo, hi, level);
  if (Lb >= c->D) {
    *split = c->D;
    return 0;
  }
  *split = Lb;
  return 1;
}

static int discover(bctx* c, int lo, int hi, int level) {
  if (c->ncells == c->cellcap) {
    c->cellcap = c->cellcap * 2 + 16;
    c->cells = (ocell*)realloc(c->cells, (size_t)c->cellcap * sizeof(ocell));
  }
  const int me = c->ncells++;
  {
    ocell* q = &c->cells[me];
    q->lo = lo; q->hi = hi; q->level = level; q->nchild = 0;
    /* linear scan of the digit at this level (keys sorted => digits non-decreasing) */
    int start = lo, g = digit_at(c, lo, level);
    for (int j = lo + 1; j <= hi; j++) {
      int gj = (j < hi) ? digit_at(c, j, level) : 8;
      if (gj != g) {
        q->clo[q->nchild] = start;
        q->chi[q->nchild] = j;
        q->nchild++;
        start = j;
        g = gj;
      }
    }
    q->sortkey = c->compress ? (q->nchild > 1 ? q->clo[1] : lo) : me;
    if (level + 1 > c->max_level) c->max_level = level + 1;
  }
  const int nchild = c->cells[me].nchild;
  for (int i = 0; i < nchild; i++) {
    int split;
    const int clo = c->cells[me].clo[i], chi = c->cells[me].chi[i];
    int internal = classify(c, clo, chi, level + 1, &split);
    int child = internal ? discover(c, clo, chi, split) : -1; /* may realloc c->cells */
    c->cells[me].ccell[i] = child;
    c->cells[me].clevel[i] = split;
  }
  return me;
}

static int cmp_cell(const void* a, const void* b) {
  const ocell* x = *(const ocell* const*)a;
  const ocell* y = *(const ocell* const*)b;
  return (x->sortkey > y->sortkey) - (x->sortkey < y->sortkey);
}

static void write_entry(bho_node* r, int32_t* er_lo, int32_t* er_hi, int e, int lo, int hi,
                        const ocell* cell /* NULL for a leaf */, int level, float s0) {
  r[e].x = r[e].y = r[e].z = r[e].m = 0.0f;
  er_lo[e] = lo;
  er_hi[e] = hi;
  if (cell) {
    r[e].kind = BHO_KIND_INTERNAL;
    r[e].first = cell->block;
    r[e].count = cell->nchild;
    r[e].s = ldexpf(s0, -cell->level);
  } else if (hi - lo == 1) {
    r[e].kind = BHO_KIND_BODY;
    r[e].first = lo;
    r[e].count = 1;
    r[e].s = -1.0f; /* negative edge: accepted by any theta >= 0 (ref:208 `idx < n` intent, D1) */
  } else {
    r[e].kind = BHO_KIND_MULTI;
    r[e].first = lo;
    r[e].count = hi - lo;
    r[e].s = ldexpf(s0, -level);
  }
}

static void write_pad(bho_node* rec, int32_t* er_lo, int32_t* er_hi, int e) {
  memset(&rec[e], 0, sizeof(rec[e]));
  rec[e].kind = BHO_KIND_PAD;
  er_lo[e] = er_hi[e] = 0;
}

int bho_build(const uint64_t* sorted_keys, int n, const bho_params* p, float s0, bho_node* rec,
              int32_t* er_lo, int32_t* er_hi, int capacity, int* n_internal, int* max_level) {
  bctx c;
  c.k = sorted_keys;
  c.B = p->key_bits / 3;
  c.cap = p->leaf_cap < 1 ? 1 : p->leaf_cap;
  c.D = p->max_depth < c.B ? p->max_depth : c.B;
  c.compress = p->compress;
  c.cells = NULL;
  c.ncells = c.cellcap = 0;
  c.max_level = 0;
  if (capacity < 2) return -1;
  int split;
  int root_internal = classify(&c, 0, n, 0, &split);
  int root = root_internal ? discover(&c, 0, n, split) : -1;
  /* child blocks in sortkey order.  Layout (include/bh.h): entry 0 = root, child blocks start at EVEN
     entries (64-byte boundaries of the 32-byte records), so entry 1 and the entry after every block of
     an odd number of children are padding (kind BHO_KIND_PAD, all fields zero) */
  ocell** order = (ocell**)malloc((size_t)(c.ncells + 1) * sizeof(ocell*));
  for (int i = 0; i < c.ncells; i++) order[i] = &c.cells[i];
  qsort(order, (size_t)c.ncells, sizeof(ocell*), cmp_cell);
  int next = 2;
  for (int i = 0; i < c.ncells; i++) {
    order[i]->block = next;
    next += (order[i]->nchild + 1) & ~1;
  }
  int E = next;
  if (E > capacity) {
    free(order);
    free(c.cells);
    return -1;
  }
  write_entry(rec, er_lo, er_hi, 0, 0, n, root >= 0 ? &c.cells[root] : NULL, split, s0);
  write_pad(rec, er_lo, er_hi, 1);
  for (int i = 0; i < c.ncells; i++) {
    const ocell* q = &c.cells[i];
    for (int j = 0; j < q->nchild; j++)
      write_entry(rec, er_lo, er_hi, q->block + j, q->clo[j], q->chi[j],
                  q->ccell[j] >= 0 ? &c.cells[q->ccell[j]] : NULL, q->clevel[j], s0);
    if (q->nchild & 1) write_pad(rec, er_lo, er_hi, q->block + q->nchild);
  }
  if (n_internal) *n_internal = c.ncells;
  if (max_level) *max_level = c.max_level;
  free(order);
  free(c.cells);
  return E;
}

/* ------------------------------------------------------------------ COM */
/* ref:158-189: mass = sum m_b, com = sum m_b p_b / mass when mass > 1e-6 (else the raw
   sum is left).  The reference accumulates with fp32 atomics in a non-deterministic
   order (D11); restated as an fp64 sum over the cell's bodies rounded once to fp32. */
void bho_com(bho_node* rec, const int32_t* er_lo, const int32_t* er_hi, int n_entries,
             const float* xyzm) {
#pragma omp parallel for schedule(dynamic, 1024)
  for (int e = 0; e < n_entries; e++) {
    bho_node* r = &rec[e];
    const int lo = er_lo[e], hi = er_hi[e];
    if (r->kind == BHO_KIND_PAD) continue; /* padding entry: stays all zero */
    if (r->kind == BHO_KIND_BODY) {
      r->x = xyzm[4 * (size_t)lo + 0];
      r->y = xyzm[4 * (size_t)lo + 1];
      r->z = xyzm[4 * (size_t)lo + 2];
      r->m = xyzm[4 * (size_t)lo + 3];
      continue;
    }
    double M = 0, sx = 0, sy = 0, sz = 0;
    for (int b = lo; b < hi; b++) {
      const float* q = &xyzm[4 * (size_t)b];
      double m = (double)q[3];
      M += m;
      sx += m * (double)q[0];
      sy += m * (double)q[1];
      sz += m * (double)q[2];
    }
    float mass = (float)M;
    r->m = mass;
    if (mass > 1e-6f) { /* ref:180 */
      r->x = (float)(sx / M);
      r->y = (float)(sy / M);
      r->z = (float)(sz / M);
    } else {
      r->x = (float)sx;
      r->y = (float)sy;
      r->z = (float)sz;
    }
  }
}

/* ------------------------------------------------------------------ force */
typedef struct {
  const bho_node* rec;
  const float* xyzm;
  float px, py, pz;
  float G, theta, eps2;
  float ax, ay, az;
  uint32_t V, O, P;
} fctx;

/* ref:205-213: d = com - p; dist = sqrtf(d.d + SOFTENING); f = G*mass/(dist^3); a += f*d */
#define BHO_DIST(c, cx, cy, cz)                     \
  float dx = (cx) - (c)->px;                        \
  float dy = (cy) - (c)->py;                        \
  float dz = (cz) - (c)->pz;                        \
  float d2 = dx * dx + dy * dy + dz * dz;           \
  float dist = sqrtf(d2 + (c)->eps2);

#define BHO_ACCUM(c, mass)                                      \
  {                                                             \
    float f = (c)->G * (mass) / (dist * dist * dist);           \
    (c)->ax += f * dx;                                          \
    (c)->ay += f * dy;                                          \
    (c)->az += f * dz;                                          \
  }

static void multi_leaf(fctx* c, const bho_node* r) {
  /* bodies of an unsplit cell interact directly, ascending sorted index (D2 intent:
     a body child uses its own pos/mass) */
  for (int b = r->first; b < r->first + r->count; b++) {
    const float* q = &c->xyzm[4 * (size_t)b];
    if (q[3] <= 0.0f) continue; /* ref:203 */
    BHO_DIST(c, q[0], q[1], q[2]);
    BHO_ACCUM(c, q[3]);
    c->P++;
  }
}

static void walk_pre(fctx* c, int e) {
  const bho_node* r = &c->rec[e];
  if (r->m <= 0.0f) return; /* ref:203 */
  BHO_DIST(c, r->x, r->y, r->z);
  if (r->kind == BHO_KIND_BODY) { /* ref:208 `idx < n` intent: a body is always accepted */
    BHO_ACCUM(c, r->m);
    c->P++;
    return;
  }
  c->V++;
  if (r->s / dist < c->theta) { /* ref:208 */
    BHO_ACCUM(c, r->m);
    return;
  }
  c->O++;
  if (r->kind == BHO_KIND_INTERNAL) {
    for (int k = 0; k < r->count; k++) walk_pre(c, r->first + k);
  } else {
    multi_leaf(c, r);
  }
}

/* the same pre-order walk without recursion (bench.py's CPU baseline): a frame is the unvisited rest of a
   child block, so the visit order — hence every sum — is identical to walk_pre's */
static void walk_pre_iter(fctx* c) {
  int nxt[64], left[64], sp = 1; /* one frame per level: depth <= 21 + root */
  nxt[0] = 0; left[0] = 1;
  while (sp > 0) {
    const int e = nxt[sp - 1]++;
    if (--left[sp - 1] == 0) sp--;
    const bho_node* r = &c->rec[e];
    if (r->m <= 0.0f) continue; /* ref:203 */
    BHO_DIST(c, r->x, r->y, r->z);
    if (r->kind == BHO_KIND_BODY) {
      BHO_ACCUM(c, r->m);
      c->P++;
      continue;
    }
    c->V++;
    if (r->s / dist < c->theta) { /* ref:208 */
      BHO_ACCUM(c, r->m);
      continue;
    }
    c->O++;
    if (r->kind == BHO_KIND_INTERNAL) {
      nxt[sp] = r->first; left[sp] = r->count; sp++;
    } else {
      multi_leaf(c, r);
    }
  }
}

static void walk_batched(fctx* c, int first, int count) {
  int opened[8], no = 0;
  for (int k = 0; k < count; k++) {
    const bho_node* r = &c->rec[first + k];
    if (r->m <= 0.0f) continue;
    BHO_DIST(c, r->x, r->y, r->z);
    if (r->kind == BHO_KIND_BODY) {
      BHO_ACCUM(c, r->m);
      c->P++;
      continue;
    }
    c->V++;
    if (r->s / dist < c->theta) {
      BHO_ACCUM(c, r->m);
      continue;
    }
    c->O++;
    if (r->kind == BHO_KIND_MULTI)
      multi_leaf(c, r);
    else
      opened[no++] = first + k;
  }
  for (int i = no - 1; i >= 0; i--) walk_batched(c, c->rec[opened[i]].first, c->rec[opened[i]].count);
}

void bho_force(const bho_node* rec, const float* xyzm, int lo, int hi, const bho_params* p, int order,
               float* acc4, uint32_t* V, uint32_t* O, uint32_t* P, int nthreads) {
  if (nthreads < 1) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
  for (int i = lo; i < hi; i++) {
    fctx c;
    c.rec = rec;
    c.xyzm = xyzm;
    c.px = xyzm[4 * (size_t)i + 0]; /* ref:196 */
    c.py = xyzm[4 * (size_t)i + 1];
    c.pz = xyzm[4 * (size_t)i + 2];
    c.G = p->G;
    c.theta = p->theta;
    c.eps2 = p->eps2;
    c.ax = c.ay = c.az = 0.0f;
    c.V = c.O = c.P = 0;
    if (order == BHO_ORDER_BATCHED)
      walk_batched(&c, 0, 1);
    else if (order == BHO_ORDER_PREORDER_RECURSIVE)
      walk_pre(&c, 0); /* ref:198 stack = {root} */
    else
      walk_pre_iter(&c);
    acc4[4 * (size_t)i + 0] = c.ax; /* ref:222-224 */
    acc4[4 * (size_t)i + 1] = c.ay;
    acc4[4 * (size_t)i + 2] = c.az;
    acc4[4 * (size_t)i + 3] = 0.0f;
    if (V) V[i] = c.V;
    if (O) O[i] = c.O;
    if (P) P[i] = c.P;
  }
}

/* ------------------------------------------------------------------ integrate */
/* ref:227-249, source text, no contraction */
void bho_integrate(float* xyzm, float* vel3, const float* acc4, int n, const bho_params* p) {
  const float DT = p->dt, MAX_SPEED = p->max_speed;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    float* q = &xyzm[4 * (size_t)i];
    float* w = &vel3[3 * (size_t)i];
    const float* a = &acc4[4 * (size_t)i];
    float vx = w[0] + a[0] * DT;
    float vy = w[1] + a[1] * DT;
    float vz = w[2] + a[2] * DT;
    float speedSq = vx * vx + vy * vy + vz * vz;
    if (speedSq > MAX_SPEED * MAX_SPEED) {
      float scale = MAX_SPEED / sqrtf(speedSq);
      vx *= scale;
      vy *= scale;
      vz *= scale;
    }
    w[0] = vx;
    w[1] = vy;
    w[2] = vz;
    q[0] += vx * DT;
    q[1] += vy * DT;
    q[2] += vz * DT;
  }
}

/* ------------------------------------------------------------------ fp64 direct sum */
void bho_direct_f64(const float* xyzm, int n, int lo, int hi, float G, float eps2, double* acc3,
                    int nthreads) {
  if (nthreads < 1) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int i = lo; i < hi; i++) {
    double px = xyzm[4 * (size_t)i], py = xyzm[4 * (size_t)i + 1], pz = xyzm[4 * (size_t)i + 2];
    double ax = 0, ay = 0, az = 0;
    for (int j = 0; j < n; j++) {
      double dx = xyzm[4 * (size_t)j] - px, dy = xyzm[4 * (size_t)j + 1] - py,
             dz = xyzm[4 * (size_t)j + 2] - pz;
      double r2 = dx * dx + dy * dy + dz * dz + (double)eps2;
      double f = (double)G * (double)xyzm[4 * (size_t)j + 3] / (r2 * sqrt(r2));
      ax += f * dx;
      ay += f * dy;
      az += f * dz;
    }
    acc3[3 * (size_t)(i - lo) + 0] = ax;
    acc3[3 * (size_t)(i - lo) + 1] = ay;
    acc3[3 * (size_t)(i - lo) + 2] = az;
  }
}

/* ------------------------------------------------------------------ whole step */
struct bho_state {
  int n;
  bho_params p;
  float *xyzm, *vel3, *acc4;      /* current (Morton-sorted after the first step) order */
  float *xyzm_t, *vel3_t;         /* gather targets */
  int32_t *id, *id_t;             /* caller index of the body in each slot */
  uint64_t *keys, *skeys;
  int32_t* perm;
  bho_node* rec;
  int32_t *er_lo, *er_hi;
  int capacity, n_entries, n_internal, max_level;
  uint32_t *V, *O, *P;
  uint64_t tV, tO, tP;
  double t[7];
  float bounds[6];
};

bho_state* bho_create(int n, const bho_params* p) {
  bho_state* s = (bho_state*)calloc(1, sizeof(bho_state));
  s->n = n;
  s->p = *p;
  size_t N = (size_t)n;
  s->xyzm = (float*)calloc(4 * N, 4);
  s->xyzm_t = (float*)calloc(4 * N, 4);
  s->vel3 = (float*)calloc(3 * N, 4);
  s->vel3_t = (float*)calloc(3 * N, 4);
  s->acc4 = (float*)calloc(4 * N, 4);
  s->id = (int32_t*)calloc(N, 4);
  s->id_t = (int32_t*)calloc(N, 4);
  s->keys = (uint64_t*)calloc(N, 8);
  s->skeys = (uint64_t*)calloc(N, 8);
  s->perm = (int32_t*)calloc(N, 4);
  s->capacity = 3 * n + 8;
  s->rec = (bho_node*)calloc((size_t)s->capacity, sizeof(bho_node));
  s->er_lo = (int32_t*)calloc((size_t)s->capacity, 4);
  s->er_hi = (int32_t*)calloc((size_t)s->capacity, 4);
  s->V = (uint32_t*)calloc(N, 4);
  s->O = (uint32_t*)calloc(N, 4);
  s->P = (uint32_t*)calloc(N, 4);
  return s;
}

void bho_destroy(bho_state* s) {
  if (!s) return;
  free(s->xyzm); free(s->xyzm_t); free(s->vel3); free(s->vel3_t); free(s->acc4);
  free(s->id); free(s->id_t); free(s->keys); free(s->skeys); free(s->perm);
  free(s->rec); free(s->er_lo); free(s->er_hi); free(s->V); free(s->O); free(s->P);
  free(s);
}

void bho_upload(bho_state* s, const float* x, const float* y, const float* z, const float* vx,
                const float* vy, const float* vz, const float* m) {
  for (int i = 0; i < s->n; i++) { /* ref:329-335 */
    s->xyzm[4 * (size_t)i + 0] = x[i];
    s->xyzm[4 * (size_t)i + 1] = y[i];
    s->xyzm[4 * (size_t)i + 2] = z[i];
    s->xyzm[4 * (size_t)i + 3] = m[i];
    s->vel3[3 * (size_t)i + 0] = vx[i];
    s->vel3[3 * (size_t)i + 1] = vy[i];
    s->vel3[3 * (size_t)i + 2] = vz[i];
    s->id[i] = i;
  }
}

/* ref:255-283 stage order.  Unlike the reference (D12) the bodies are physically
   gathered into Morton order each step and stay there; `id` carries the caller index. */
void bho_step(bho_state* s, int order, int nthreads) {
  const int n = s->n;
  double t0 = omp_get_wtime(), t1;
  bbox_strided(s->xyzm, s->xyzm + 1, s->xyzm + 2, 4, n, s->bounds); /* ref:259 */
  t1 = omp_get_wtime(); s->t[0] = t1 - t0; t0 = t1;
  keys_strided(s->xyzm, s->xyzm + 1, s->xyzm + 2, 4, s->bounds, n, s->p.key_bits, s->p.key_curve, s->keys); /* ref:260 */
  t1 = omp_get_wtime(); s->t[1] = t1 - t0; t0 = t1;
  bho_sort(s->keys, n, s->skeys, s->perm); /* ref:262-264 */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    int j = s->perm[i];
    memcpy(&s->xyzm_t[4 * (size_t)i], &s->xyzm[4 * (size_t)j], 16);
    memcpy(&s->vel3_t[3 * (size_t)i], &s->vel3[3 * (size_t)j], 12);
    s->id_t[i] = s->id[j];
  }
  { float* t = s->xyzm; s->xyzm = s->xyzm_t; s->xyzm_t = t; }
  { float* t = s->vel3; s->vel3 = s->vel3_t; s->vel3_t = t; }
  { int32_t* t = s->id; s->id = s->id_t; s->id_t = t; }
  t1 = omp_get_wtime(); s->t[2] = t1 - t0; t0 = t1;
  s->n_entries = bho_build(s->skeys, n, &s->p, bho_root_edge(s->bounds), s->rec, s->er_lo, s->er_hi,
                           s->capacity, &s->n_internal, &s->max_level); /* ref:266-275 */
  t1 = omp_get_wtime(); s->t[3] = t1 - t0; t0 = t1;
  bho_com(s->rec, s->er_lo, s->er_hi, s->n_entries, s->xyzm); /* ref:279-280 */
  t1 = omp_get_wtime(); s->t[4] = t1 - t0; t0 = t1;
  bho_force(s->rec, s->xyzm, 0, n, &s->p, order, s->acc4, s->V, s->O, s->P, nthreads); /* ref:281 */
  t1 = omp_get_wtime(); s->t[5] = t1 - t0; t0 = t1;
  uint64_t tV = 0, tO = 0, tP = 0;
  for (int i = 0; i < n; i++) {
    tV += s->V[i];
    tO += s->O[i];
    tP += s->P[i];
  }
  s->tV = tV; s->tO = tO; s->tP = tP;
  t0 = omp_get_wtime();
  bho_integrate(s->xyzm, s->vel3, s->acc4, n, &s->p); /* ref:282 */
  t1 = omp_get_wtime(); s->t[6] = t1 - t0;
}

void bho_download(const bho_state* s, float* x, float* y, float* z, float* vx, float* vy, float* vz) {
  for (int i = 0; i < s->n; i++) {
    int j = s->id[i];
    if (x) x[j] = s->xyzm[4 * (size_t)i + 0];
    if (y) y[j] = s->xyzm[4 * (size_t)i + 1];
    if (z) z[j] = s->xyzm[4 * (size_t)i + 2];
    if (vx) vx[j] = s->vel3[3 * (size_t)i + 0];
    if (vy) vy[j] = s->vel3[3 * (size_t)i + 1];
    if (vz) vz[j] = s->vel3[3 * (size_t)i + 2];
  }
}

void bho_download_acc(const bho_state* s, float* ax, float* ay, float* az) {
  for (int i = 0; i < s->n; i++) {
    int j = s->id[i];
    ax[j] = s->acc4[4 * (size_t)i + 0];
    ay[j] = s->acc4[4 * (size_t)i + 1];
    az[j] = s->acc4[4 * (size_t)i + 2];
  }
}

void bho_last_counts(const bho_state* s, uint64_t* V, uint64_t* O, uint64_t* P, int* n_internal,
                     int* n_entries, int* max_level) {
  if (V) *V = s->tV;
  if (O) *O = s->tO;
  if (P) *P = s->tP;
  if (n_internal) *n_internal = s->n_internal;
  if (n_entries) *n_entries = s->n_entries;
  if (max_level) *max_level = s->max_level;
}

void bho_last_times(const bho_state* s, double t[7]) { memcpy(t, s->t, sizeof(s->t)); }

/* ------------------------------------------------------------------ analysis helper */
/* Work a group of `group` consecutive sorted bodies does when it walks ONE shared traversal
   (the GPU kernel's scheme): a record is evaluated by the group if any member evaluates it.
   Returns per group: records evaluated, blocks popped, multi-leaf bodies streamed. Not part of
   the recurrence — used by tools/ and tests to size the kernel's lane efficiency. */
typedef struct { const bho_node* rec; const float* xyzm; const bho_params* p; int g0, g1;
                 uint64_t recs, pops, leafbodies; } gctx;

static void group_walk(gctx* c, int first, int count, const uint8_t* active_in) {
  uint8_t want[8][64];
  int opened[8], no = 0;
  const int G = c->g1 - c->g0;
  c->pops++;
  for (int k = 0; k < count; k++) {
    const bho_node* r = &c->rec[first + k];
    if (r->m <= 0.0f) continue;
    c->recs++;
    int any = 0;
    for (int l = 0; l < G; l++) {
      want[no][l] = 0;
      if (!active_in[l]) continue;
      const float* q = &c->xyzm[4 * (size_t)(c->g0 + l)];
      float dx = r->x - q[0], dy = r->y - q[1], dz = r->z - q[2];
      float dist = sqrtf(dx * dx + dy * dy + dz * dz + c->p->eps2);
      if (r->kind != BHO_KIND_BODY && !(r->s / dist < c->p->theta)) { want[no][l] = 1; any = 1; }
    }
    if (any) {
      if (r->kind == BHO_KIND_MULTI) c->leafbodies += (uint64_t)r->count;
      else opened[no++] = first + k;
    }
  }
  /* copy masks: recursion reuses the stack frame arrays */
  for (int i = no - 1; i >= 0; i--) {
    uint8_t m[64];
    memcpy(m, want[i], 64);
    group_walk(c, c->rec[opened[i]].first, c->rec[opened[i]].count, m);
  }
}

void bho_group_stats(const bho_node* rec, const float* xyzm, int n, const bho_params* p, int group,
                     int stride, uint64_t* recs, uint64_t* pops, uint64_t* leafbodies, uint64_t* groups) {
  uint64_t R = 0, Pp = 0, Lb = 0, Gn = 0;
  const int ngroups = (n + group - 1) / group;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : R, Pp, Lb, Gn)
  for (int g = 0; g < ngroups; g += stride) {
    gctx c;
    c.rec = rec; c.xyzm = xyzm; c.p = p;
    c.g0 = g * group;
    c.g1 = c.g0 + group < n ? c.g0 + group : n;
    c.recs = c.pops = c.leafbodies = 0;
    uint8_t act[64];
    memset(act, 1, 64);
    group_walk(&c, 0, 1, act);
    R += c.recs; Pp += c.pops; Lb += c.leafbodies; Gn += 1;
  }
  *recs = R; *pops = Pp; *leafbodies = Lb; *groups = Gn;
}
