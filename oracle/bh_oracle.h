/*
 * bh_oracle.h — CPU restatement of the reference's per-step Barnes-Hut path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (libbh.so, the Python host
 * package, the C++ bench driver) may include, link, import or execute anything
 * under oracle/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (bgcarmin/NBody-Barnes-Hut-CUDA) ships no tests,
 * golden vectors or known-answer fixtures for this path, it is
 * CUDA-only (needs nvcc + Thrust/CUB, absent here) so it cannot be built or run in
 * this image, and its literal force/tree code is degenerate.
 * This oracle therefore restates the reference's *intended* recurrence from its
 * source text; it is cross-checked against an independent fp64 direct sum and
 * analytic cases (tests/test_oracle_*.py) but not against reference outputs.
 *
 * "ref:" = line numbers of /root/reference/nbody_v5_bench.cu.
 */
#ifndef BH_ORACLE_H_
#define BH_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bho_params {
  float G, theta, dt, eps2, max_speed; /* ref:14-18 */
  int32_t leaf_cap, max_depth, key_bits;
  int32_t compress; /* 1: path-compressed octree (the engine's tree); 0: literal chain cells */
  int32_t key_curve; /* numbering of the key grid's cells: 0 = Morton (the reference's, ref:42-63; default),
                        1 = Hilbert (include/bh.h bh_params.key_curve; 63-bit keys only) */
} bho_params;

#define BHO_KIND_BODY 0
#define BHO_KIND_INTERNAL 1
#define BHO_KIND_MULTI 2
#define BHO_KIND_PAD 3 /* padding entry: child blocks start at even entries (include/bh.h BH_KIND_PAD) */

/* same 32-byte record as include/bh.h's bh_node (defined independently here) */
typedef struct bho_node {
  float x, y, z, m;
  float s;
  int32_t first, count, kind;
} bho_node;

#define BHO_ORDER_PREORDER 0 /* depth-first, children ascending, each child finished before the next */
#define BHO_ORDER_PREORDER_RECURSIVE 2 /* the same order by plain recursion (cross-check of the iterative walk) */
#define BHO_ORDER_BATCHED 1  /* per opened cell: evaluate all children ascending (accepted ones and
                                multi-body leaves accumulate at once), then descend into the opened
                                internal children in DESCENDING order — the GPU kernel's order     */

void bho_default_params(bho_params* p);

/* ---- stages on plain arrays ---- */
void bho_bbox(const float* x, const float* y, const float* z, int n, float bounds[6]);              /* ref:134-156 */
void bho_morton30(const float* x, const float* y, const float* z, const float bounds[6], int n,
                  uint32_t* codes, int32_t* indices);                                                /* ref:42-63, literal */
/* key_curve as in bho_params */
void bho_keys(const float* x, const float* y, const float* z, const float bounds[6], int n,
              int key_bits, int key_curve, uint64_t* keys);
/* 21-bit cell coordinates <-> Hilbert cell number (test hooks for the curve's defining properties) */
uint64_t bho_hilbert_index(uint32_t x, uint32_t y, uint32_t z);
void bho_hilbert_cell(uint64_t index, uint32_t xyz[3]);
void bho_sort(const uint64_t* keys, int n, uint64_t* sorted_keys, int32_t* perm);                    /* ref:262-264: stable */
float bho_root_edge(const float bounds[6]);                                                          /* ref:55 fmaxf(b[3]-b[0],1) */
/* returns number of entries, or -1 if capacity is too small */
int bho_build(const uint64_t* sorted_keys, int n, const bho_params* p, float s0, bho_node* rec,
              int32_t* er_lo, int32_t* er_hi, int capacity, int* n_internal, int* max_level);        /* ref:83-132 intent */
void bho_com(bho_node* rec, const int32_t* er_lo, const int32_t* er_hi, int n_entries,
             const float* xyzm);                                                                      /* ref:158-189 */
/* accelerations of sorted bodies [lo,hi) -> acc4[4*(i)] for i in [lo,hi); counters may be NULL */
void bho_force(const bho_node* rec, const float* xyzm, int lo, int hi, const bho_params* p, int order,
               float* acc4, uint32_t* V, uint32_t* O, uint32_t* P, int nthreads);                     /* ref:191-225 intent */
void bho_integrate(float* xyzm, float* vel3, const float* acc4, int n, const bho_params* p);          /* ref:227-249 */
/* fp64 direct O(n^2) sum with the same softened kernel (physics check) */
void bho_direct_f64(const float* xyzm, int n, int lo, int hi, float G, float eps2, double* acc3,
                    int nthreads);

/* ---- whole-step state machine mirroring simulationStep (ref:255-283) ---- */
typedef struct bho_state bho_state;
bho_state* bho_create(int n, const bho_params* p);
void bho_destroy(bho_state* s);
void bho_upload(bho_state* s, const float* x, const float* y, const float* z, const float* vx,
                const float* vy, const float* vz, const float* m);
void bho_step(bho_state* s, int order, int nthreads);
void bho_download(const bho_state* s, float* x, float* y, float* z, float* vx, float* vy, float* vz);
void bho_download_acc(const bho_state* s, float* ax, float* ay, float* az);
/* totals of the last bho_step's force stage */
void bho_last_counts(const bho_state* s, uint64_t* V, uint64_t* O, uint64_t* P, int* n_internal,
                     int* n_entries, int* max_level);
/* per-stage wall seconds of the last bho_step: bbox, keys, sort, build, com, force, integrate */
void bho_last_times(const bho_state* s, double t[7]);
int bho_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
