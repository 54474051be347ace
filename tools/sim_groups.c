/* design study: per-group union size (records a 64-body wave evaluates) for consecutive groups */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float x, y, z, m, s; int first, count, kind; } node;
typedef struct { int first, count; uint64_t mask; } ent;
/* out[g] = records evaluated by group g (bodies [g*group, (g+1)*group)) ; lane_need[g] = sum over lanes of per-lane evals */
void sim_groups(const node* rec, const float* xyzm, int n, float theta, float eps2, int group, int stride,
                int* out, int* lane_need) {
  int ngroups = (n + group - 1) / group;
#pragma omp parallel for schedule(dynamic, 16)
  for (int g = 0; g < ngroups; g += stride) {
    int g0 = g * group, g1 = g0 + group < n ? g0 + group : n;
    ent st[512]; int sp = 0; int R = 0; long need = 0;
    uint64_t full = (g1 - g0 == 64) ? ~0ull : ((1ull << (g1 - g0)) - 1);
    st[sp++] = (ent){0, 1, full};
    while (sp > 0) {
      ent e = st[--sp];
      for (int k = 0; k < e.count; k++) {
        const node* r = &rec[e.first + k];
        R++; need += __builtin_popcountll(e.mask);
        if (r->kind == 0 || r->m <= 0) continue;
        uint64_t open = 0;
        for (int l = 0; l < g1 - g0; l++) {
          if (!((e.mask >> l) & 1)) continue;
          const float* q = &xyzm[4 * (size_t)(g0 + l)];
          float dx = r->x - q[0], dy = r->y - q[1], dz = r->z - q[2];
          float dist = sqrtf(dx * dx + dy * dy + dz * dz + eps2);
          if (!(r->s / dist < theta)) open |= 1ull << l;
        }
        if (open && r->kind == 1) st[sp++] = (ent){r->first, r->count, open};
      }
    }
    out[g] = R; lane_need[g] = (int)(need / (g1 - g0));
  }
}
