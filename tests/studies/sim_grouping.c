/* design study (not product code): cost of the wave-shared walk under different body->wave groupings.
 * A group is (start, count <= 64) in the sorted body order; for each group the shared traversal is run and
 *   blocks        child blocks popped
 *   pairs         sum over blocks of ceil(children / 2)            (the product kernel's unit of VALU work)
 *   pairs_half    the same, but a block whose lane mask lies inside one aligned 32-lane half costs
 *                 ceil(pairs / 2), inside one aligned 16-lane quarter ceil(pairs / 4) ("dual evaluation")
 *   lane_evals    (lane, record) evaluations actually needed
 *   recs          records evaluated (sum of children)
 * are accumulated.  sub = 64, 32 or 16 additionally restricts the group to sub bodies per wave with 64/sub
 * record pairs evaluated per instruction (cost per block = ceil(pairs / (64 / sub))).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float x, y, z, m, s; int first, count, kind; } node;
typedef struct { int first, count; uint64_t mask; } ent;

void sim_groups(const node* rec, const float* xyzm, const int* gstart, const int* gcount, int ngroups, float theta,
                float eps2, int sub, uint64_t* out /* [8] */) {
  uint64_t B = 0, P = 0, PH = 0, LE = 0, R = 0, PS = 0;
  const int per = 64 / sub;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : B, P, PH, LE, R, PS)
  for (int gi = 0; gi < ngroups; gi++) {
    const int g0 = gstart[gi], cnt = gcount[gi];
    if (cnt <= 0) continue;
    ent st[1024];
    int sp = 0;
    const uint64_t full = cnt == 64 ? ~0ull : ((1ull << cnt) - 1);
    st[sp++] = (ent){0, 1, full};
    while (sp > 0) {
      const ent e = st[--sp];
      const int pairs = (e.count + 1) / 2;
      B++;
      P += pairs;
      PS += (pairs + per - 1) / per;
      R += e.count;
      LE += (uint64_t)e.count * __builtin_popcountll(e.mask);
      {
        const uint64_t m = e.mask;
        int q = 0;
        for (int k = 0; k < 4; k++)
          if ((m >> (16 * k)) & 0xffffull) q |= 1 << k;
        if (q == 1 || q == 2 || q == 4 || q == 8)
          PH += (pairs + 3) / 4;
        else if ((q & 12) == 0 || (q & 3) == 0)
          PH += (pairs + 1) / 2;
        else
          PH += pairs;
      }
      for (int k = 0; k < e.count; k++) {
        const node* r = &rec[e.first + k];
        if (r->kind == 0 || r->kind == 3 || r->m <= 0) continue;
        uint64_t open = 0;
        for (int l = 0; l < cnt; l++) {
          if (!((e.mask >> l) & 1)) continue;
          const float* q = &xyzm[4 * (size_t)(g0 + l)];
          const float dx = r->x - q[0], dy = r->y - q[1], dz = r->z - q[2];
          const float dist = sqrtf(dx * dx + dy * dy + dz * dz + eps2);
          if (!(r->s / dist < theta)) open |= 1ull << l;
        }
        if (open && r->kind == 1 && sp < 1024) st[sp++] = (ent){r->first, r->count, open};
      }
    }
  }
  out[0] = B; out[1] = P; out[2] = PH; out[3] = LE; out[4] = R; out[5] = PS;
}
