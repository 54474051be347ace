// bh_ic.cpp — synthetic initial conditions (host only, no GPU needed).
//
// The reference seeds srand(42) and draws a rotating thin disc with C rand()
// (nbody_v5_bench.cu:294-308), which is platform dependent.  Both generators
// here use a counter-based splitmix64 stream keyed on (seed, body index, draw index), so a
// body's state does not depend on n, on generation order, or on the rank that generates it.
#include <math.h>
#include <stdint.h>

#include "bh.h"

namespace {

struct Rng {
  uint64_t key;
  uint64_t ctr;
  Rng(uint64_t seed, uint64_t index) : key(seed ^ (index * 0xD1B54A32D192ED03ull)), ctr(0) {}
  // uniform in [0,1), 53 bits
  double u() {
    uint64_t x = key + (++ctr) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
  }
};

const double kPi = 3.14159265358979323846;

}  // namespace

extern "C" {

// Plummer sphere: scale radius a, radius by inverse CDF r = a / sqrt(u^(-2/3) - 1)
// redrawn while r > 10 a, isotropic direction; mass m = 2 + 5u (law of ref:302); speed by the
// Aarseth-Henon-Wielen rejection g(q) = q^2 (1-q^2)^(7/2), v = q sqrt(2 G M / a) (1 + r^2/a^2)^(-1/4).
int bh_ic_plummer(int n, uint64_t seed, float a, float G, float* x, float* y, float* z, float* vx,
                  float* vy, float* vz, float* m) {
  if (n < 1 || !x || !y || !z || !vx || !vy || !vz || !m || !(a > 0.0f)) return BH_ERR_BAD_ARG;
  double M = 0.0;
  for (int i = 0; i < n; i++) {
    Rng g(seed, (uint64_t)i);
    const double mi = 2.0 + 5.0 * g.u();  // draw 1
    m[i] = (float)mi;
    M += (double)m[i];
  }
  const double A = (double)a;
  const double vscale = sqrt(2.0 * (double)G * M / A);
  for (int i = 0; i < n; i++) {
    Rng g(seed, (uint64_t)i);
    (void)g.u();  // draw 1 was the mass
    double r;
    do {
      double u = g.u();
      if (u < 1e-12) u = 1e-12;
      r = A / sqrt(pow(u, -2.0 / 3.0) - 1.0);
    } while (!(r <= 10.0 * A));
    double cz = 1.0 - 2.0 * g.u();
    double ph = 2.0 * kPi * g.u();
    double sz = sqrt(fmax(0.0, 1.0 - cz * cz));
    x[i] = (float)(r * sz * cos(ph));
    y[i] = (float)(r * sz * sin(ph));
    z[i] = (float)(r * cz);
    double q, gq;
    do {
      q = g.u();
      gq = 0.1 * g.u();
    } while (gq > q * q * pow(1.0 - q * q, 3.5));
    const double v = q * vscale * pow(1.0 + (r * r) / (A * A), -0.25);
    cz = 1.0 - 2.0 * g.u();
    ph = 2.0 * kPi * g.u();
    sz = sqrt(fmax(0.0, 1.0 - cz * cz));
    vx[i] = (float)(v * sz * cos(ph));
    vy[i] = (float)(v * sz * sin(ph));
    vz[i] = (float)(v * cz);
  }
  return BH_OK;
}

// The reference's disc by formula (ref:297-307); draw order r, angle, z, mass, vz as in the loop.
int bh_ic_disc(int n, uint64_t seed, float G, float* x, float* y, float* z, float* vx, float* vy,
               float* vz, float* m) {
  if (n < 1 || !x || !y || !z || !vx || !vy || !vz || !m) return BH_ERR_BAD_ARG;
  for (int i = 0; i < n; i++) {
    Rng g(seed, (uint64_t)i);
    const float r = 200.0f + (float)g.u() * 1500.0f;                     // ref:297
    const float a = (float)((double)((float)g.u() * 2.0f) * kPi);        // ref:298
    x[i] = (float)((double)r * cos((double)a));                          // ref:299
    y[i] = (float)((double)r * sin((double)a));                          // ref:300
    z[i] = ((float)g.u() - 0.5f) * (r * 0.05f);                          // ref:301
    m[i] = 2.0f + (float)g.u() * 5.0f;                                   // ref:302
    const float approx_mass_inside = 50000.0f + r * 100.0f;              // ref:303
    const float v_mag = sqrtf(G * approx_mass_inside / r);               // ref:304
    vx[i] = (float)(-sin((double)a) * (double)v_mag);                    // ref:305
    vy[i] = (float)(cos((double)a) * (double)v_mag);                     // ref:306
    vz[i] = ((float)g.u() - 0.5f) * 2.0f;                                // ref:307
  }
  return BH_OK;
}

// The reference BINARY's disc: srand(seed) + the Microsoft C runtime's rand() — the authors' nbody_v5_bench.exe
// is an MSVC build — in the call order of ref:294-308, with the source's own arithmetic
// (`(float)rand() / RAND_MAX` in float, RAND_MAX = 32767; `* 2.0f * M_PI` and the cos/sin products in double).
// MSVC rand(): state = state * 214013 + 2531011; return (state >> 16) & 0x7fff.  With seed 42 and
// bh_params.literal_force = 1 the trajectories can be diffed against the CUDA binary's output.
int bh_ic_disc_msvc(int n, uint32_t seed, float G, float* x, float* y, float* z, float* vx, float* vy,
                    float* vz, float* m) {
  if (n < 1 || !x || !y || !z || !vx || !vy || !vz || !m) return BH_ERR_BAD_ARG;
  uint32_t state = seed;  // srand(42) ref:294
  auto rnd = [&state]() -> float {
    state = state * 214013u + 2531011u;
    return (float)(int)((state >> 16) & 0x7fffu) / 32767.0f;  // (float)rand() / RAND_MAX
  };
  for (int i = 0; i < n; i++) {
    const float r = 200.0f + rnd() * 1500.0f;                            // ref:297
    const float a = (float)((double)(rnd() * 2.0f) * kPi);               // ref:298
    x[i] = (float)((double)r * cos((double)a));                          // ref:299
    y[i] = (float)((double)r * sin((double)a));                          // ref:300
    z[i] = (rnd() - 0.5f) * (r * 0.05f);                                 // ref:301
    m[i] = 2.0f + rnd() * 5.0f;                                          // ref:302
    const float approx_mass_inside = 50000.0f + r * 100.0f;              // ref:303
    const float v_mag = sqrtf(G * approx_mass_inside / r);               // ref:304
    vx[i] = (float)(-sin((double)a) * (double)v_mag);                    // ref:305
    vy[i] = (float)(cos((double)a) * (double)v_mag);                     // ref:306
    vz[i] = (rnd() - 0.5f) * 2.0f;                                       // ref:307
  }
  return BH_OK;
}

}  // extern "C"
