/* design study (not product code): how the 64 lanes of a wave are occupied during the shared traversal,
 * and what a "defer sparse masks" strategy would save.
 *   hist[p]      = record evaluations made with exactly p active lanes (p = 1..64)
 *   For each threshold T in thr[0..nthr): records evaluated by the wave if every open mask with <= T lanes
 *   is NOT pushed (dense[T]) and the (lane, record) pair evaluations those lanes then need on their own
 *   below the deferred cells (pairs[T]).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float x, y, z, m, s; int first, count, kind; } node;
typedef struct { int first, count; uint64_t mask; } ent;

/* per-lane traversal size (record evaluations) of the children block [first, first+count) for body q */
static long lane_walk(const node* rec, const float* q, int first, int count, float theta, float eps2) {
  int st[512], sc[512], sp = 0;
  long evals = 0;
  st[sp] = first; sc[sp++] = count;
  while (sp > 0) {
    --sp;
    const int f = st[sp], c = sc[sp];
    for (int k = 0; k < c; k++) {
      const node* r = &rec[f + k];
      evals++;
      if (r->kind == 0 || r->m <= 0) continue;
      float dx = r->x - q[0], dy = r->y - q[1], dz = r->z - q[2];
      float dist = sqrtf(dx * dx + dy * dy + dz * dz + eps2);
      if (!(r->s / dist < theta) && r->kind == 1) { st[sp] = r->first; sc[sp++] = r->count; }
    }
  }
  return evals;
}

void sim_lanes(const node* rec, const float* xyzm, int n, float theta, float eps2, int stride,
               uint64_t* hist /*[65]*/, const int* thr, int nthr, uint64_t* dense, uint64_t* pairs,
               uint64_t* defers, uint64_t* groups_out, uint64_t* pushes_out) {
  const int group = 64;
  int ngroups = (n + group - 1) / group;
  uint64_t H[65] = {0};
  uint64_t G = 0, PU = 0;
  uint64_t D[16] = {0}, P[16] = {0}, DF[16] = {0};
#pragma omp parallel
  {
    uint64_t h[65] = {0}, d[16] = {0}, p[16] = {0}, df[16] = {0}, g = 0, pu = 0;
#pragma omp for schedule(dynamic, 8)
    for (int gi = 0; gi < ngroups; gi += stride) {
      int g0 = gi * group, g1 = g0 + group < n ? g0 + group : n;
      g++;
      for (int t = -1; t < nthr; t++) {
        const int T = t < 0 ? 0 : thr[t];
        ent st[512]; int sp = 0;
        uint64_t full = (g1 - g0 == 64) ? ~0ull : ((1ull << (g1 - g0)) - 1);
        st[sp++] = (ent){0, 1, full};
        while (sp > 0) {
          ent e = st[--sp];
          for (int k = 0; k < e.count; k++) {
            const node* r = &rec[e.first + k];
            if (t < 0) h[__builtin_popcountll(e.mask)]++; else d[t]++;
            if (r->kind == 0 || r->m <= 0) continue;
            uint64_t open = 0;
            for (int l = 0; l < g1 - g0; l++) {
              if (!((e.mask >> l) & 1)) continue;
              const float* q = &xyzm[4 * (size_t)(g0 + l)];
              float dx = r->x - q[0], dy = r->y - q[1], dz = r->z - q[2];
              float dist = sqrtf(dx * dx + dy * dy + dz * dz + eps2);
              if (!(r->s / dist < theta)) open |= 1ull << l;
            }
            if (open && r->kind == 1) {
              if (t >= 0 && __builtin_popcountll(open) <= T) {
                df[t]++;
                for (int l = 0; l < g1 - g0; l++)
                  if ((open >> l) & 1)
                    p[t] += lane_walk(rec, &xyzm[4 * (size_t)(g0 + l)], r->first, r->count, theta, eps2);
              } else {
                if (t < 0) pu++;
                st[sp++] = (ent){r->first, r->count, open};
              }
            }
          }
        }
      }
    }
#pragma omp critical
    {
      for (int i = 0; i < 65; i++) H[i] += h[i];
      for (int i = 0; i < nthr; i++) { D[i] += d[i]; P[i] += p[i]; DF[i] += df[i]; }
      G += g; PU += pu;
    }
  }
  memcpy(hist, H, sizeof H);
  for (int i = 0; i < nthr; i++) { dense[i] = D[i]; pairs[i] = P[i]; defers[i] = DF[i]; }
  *groups_out = G; *pushes_out = PU;
}
