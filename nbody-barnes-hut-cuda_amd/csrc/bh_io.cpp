// bh_io.cpp — state dump / restart (host only).
//
// The reference's v5 generation writes nothing to disk; its older generation (nbody_bh.exe)
// left a final-state text dump whose header is preserved in /root/reference/output_bh.txt:1-4.
// bh_write_text / bh_read_text speak that format (so a dump can be diffed against such a file or
// fed to an external viewer); bh_write_snapshot / bh_read_snapshot are the lossless binary form
// used for checkpoint / resume (positions, velocities AND masses + the parameters).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bh.h"

namespace {
const char kMagic[8] = {'B', 'H', 'S', 'N', 'A', 'P', '0', '1'};
}

extern "C" {

int bh_write_text(const char* path, int n, int steps, float theta, float dt, const float* x,
                  const float* y, const float* z, const float* vx, const float* vy, const float* vz) {
  if (!path || n < 1 || !x || !y || !z || !vx || !vy || !vz) return BH_ERR_BAD_ARG;
  FILE* f = fopen(path, "w");
  if (!f) return BH_ERR_BAD_ARG;
  fprintf(f, "# Barnes-Hut N-Body Simulation Results\n");                       // output_bh.txt:1
  fprintf(f, "# Final positions and velocities after %d steps\n", steps);       // :2
  fprintf(f, "# Bodies: %d, Theta: %.2f, dt: %.3f\n", n, theta, dt);            // :3
  fprintf(f, "# Format: x y z vx vy vz\n");                                     // :4
  for (int i = 0; i < n; i++)
    fprintf(f, "%f %f %f %f %f %f\n", x[i], y[i], z[i], vx[i], vy[i], vz[i]);
  const int bad = ferror(f);
  fclose(f);
  return bad ? BH_ERR_BAD_ARG : BH_OK;
}

int bh_read_text(const char* path, int capacity, int* n_out, int* steps_out, float* x, float* y,
                 float* z, float* vx, float* vy, float* vz) {
  if (!path || !n_out) return BH_ERR_BAD_ARG;
  FILE* f = fopen(path, "r");
  if (!f) return BH_ERR_BAD_ARG;
  char line[512];
  int n = 0, declared = -1, steps = -1;
  int status = BH_OK;
  while (fgets(line, sizeof(line), f)) {
    if (line[0] == '#') {
      int v;
      if (sscanf(line, "# Bodies: %d", &v) == 1) declared = v;
      if (sscanf(line, "# Final positions and velocities after %d steps", &v) == 1) steps = v;
      continue;
    }
    float a[6];
    if (sscanf(line, "%f %f %f %f %f %f", &a[0], &a[1], &a[2], &a[3], &a[4], &a[5]) != 6) continue;
    if (n < capacity && x && y && z && vx && vy && vz) {
      x[n] = a[0]; y[n] = a[1]; z[n] = a[2];
      vx[n] = a[3]; vy[n] = a[4]; vz[n] = a[5];
    } else if (n >= capacity) {
      status = BH_ERR_SMALL_BUFFER;
    }
    n++;
  }
  fclose(f);
  *n_out = n;
  if (steps_out) *steps_out = steps;
  if (declared >= 0 && declared != n) return BH_ERR_BAD_ARG;  // truncated or padded file
  return status;
}

int bh_write_snapshot(const char* path, int n, int steps, const bh_params* p, const float* x,
                      const float* y, const float* z, const float* vx, const float* vy,
                      const float* vz, const float* m) {
  if (!path || n < 1 || !p || !x || !y || !z || !vx || !vy || !vz || !m) return BH_ERR_BAD_ARG;
  FILE* f = fopen(path, "wb");
  if (!f) return BH_ERR_BAD_ARG;
  const int32_t hdr[4] = {n, steps, (int32_t)sizeof(bh_params), BH_ABI_VERSION};
  bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(hdr, sizeof(hdr), 1, f) == 1 &&
            fwrite(p, sizeof(bh_params), 1, f) == 1;
  const float* arr[7] = {x, y, z, vx, vy, vz, m};
  for (int k = 0; ok && k < 7; k++) ok = fwrite(arr[k], sizeof(float), (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok ? BH_OK : BH_ERR_BAD_ARG;
}

int bh_read_snapshot(const char* path, int capacity, int* n_out, int* steps_out, bh_params* p_out,
                     float* x, float* y, float* z, float* vx, float* vy, float* vz, float* m) {
  if (!path || !n_out) return BH_ERR_BAD_ARG;
  FILE* f = fopen(path, "rb");
  if (!f) return BH_ERR_BAD_ARG;
  char magic[8];
  int32_t hdr[4];
  bh_params p;
  bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, kMagic, 8) == 0 &&
            fread(hdr, sizeof(hdr), 1, f) == 1 && hdr[2] == (int32_t)sizeof(bh_params) &&
            fread(&p, sizeof(bh_params), 1, f) == 1;
  if (!ok) {
    fclose(f);
    return BH_ERR_BAD_ARG;
  }
  *n_out = hdr[0];
  if (steps_out) *steps_out = hdr[1];
  if (p_out) *p_out = p;
  if (!x) {  // header query only
    fclose(f);
    return BH_OK;
  }
  if (hdr[0] > capacity) {
    fclose(f);
    return BH_ERR_SMALL_BUFFER;
  }
  float* arr[7] = {x, y, z, vx, vy, vz, m};
  for (int k = 0; ok && k < 7; k++)
    ok = arr[k] && fread(arr[k], sizeof(float), (size_t)hdr[0], f) == (size_t)hdr[0];
  fclose(f);
  return ok ? BH_OK : BH_ERR_BAD_ARG;
}

}  // extern "C"
