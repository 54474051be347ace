/* CPU simulation of the batched wave traversal (design study, not product code):
   per batch take the top E<=8 stack entries (8 record slots each), evaluate every record for
   the group's 64 bodies, push opened internal cells.  Reports batches, slot use, max stack. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float x, y, z, m, s; int first, count, kind; } node;
typedef struct { int first, count; uint64_t mask; } ent;

void sim(const node* rec, const float* xyzm, int n, float theta, float eps2, int group, int stride,
         int Emax, int prefetch, double* out /* batches, records, maxsp, slots, maxsp_all */) {
  double B = 0, R = 0, S = 0, G = 0; int maxsp_all = 0; double sum_maxsp = 0;
  int ngroups = (n + group - 1) / group;
  for (int g = 0; g < ngroups; g += stride) {
    int g0 = g * group, g1 = g0 + group < n ? g0 + group : n;
    static ent st[8192]; int sp = 0, maxsp = 0;
    uint64_t full = (g1 - g0 == 64) ? ~0ull : ((1ull << (g1 - g0)) - 1);
    st[sp++] = (ent){0, 1, full};
    ent pre[8]; int npre = 0;  /* prefetched batch */
    for (;;) {
      ent cur[8]; int ncur = 0;
      if (prefetch) {
        memcpy(cur, pre, sizeof(ent) * npre); ncur = npre;
        npre = 0;
        while (npre < Emax && sp > 0) pre[npre++] = st[--sp];
        if (ncur == 0) { if (npre == 0) break; continue; }
      } else {
        while (ncur < Emax && sp > 0) cur[ncur++] = st[--sp];
        if (ncur == 0) break;
      }
      B++; S += 8 * ncur;
      for (int e = 0; e < ncur; e++)
        for (int k = 0; k < cur[e].count; k++) {
          const node* r = &rec[cur[e].first + k];
          R++;
          if (r->kind == 0 || r->m <= 0) continue;
          uint64_t open = 0;
          for (int l = 0; l < g1 - g0; l++) {
            if (!((cur[e].mask >> l) & 1)) continue;
            const float* q = &xyzm[4 * (size_t)(g0 + l)];
            float dx = r->x - q[0], dy = r->y - q[1], dz = r->z - q[2];
            float dist = sqrtf(dx * dx + dy * dy + dz * dz + eps2);
            if (!(r->s / dist < theta)) open |= 1ull << l;
          }
          if (open && r->kind == 1) { st[sp++] = (ent){r->first, r->count, open}; if (sp > maxsp) maxsp = sp; }
        }
    }
    G++; sum_maxsp += maxsp; if (maxsp > maxsp_all) maxsp_all = maxsp;
  }
  out[0] = B / G; out[1] = R / G; out[2] = sum_maxsp / G; out[3] = S / G; out[4] = maxsp_all;
}
