// bh_api.hip — host side of the C-ABI declared in include/bh.h.
//
// Mirrors the reference's host code (nbody_v5_bench.cu): the cudaMalloc block :311-326
// (bh_create), the seven H2D copies :329-335 (bh_upload), simulationStep :255-283 (bh_step and
// one entry point per stage, same order) and the cudaFree block :372-387 (bh_destroy).
// Differences by design: an opaque context instead of file-scope globals, error codes instead
// of unchecked calls, no per-step host synchronisation (the reference blocks on a
// 4-byte D2H of nodeCounter every step, :277-278), per-stage hipEvent timers.
// There is NO CPU fallback anywhere in this library: without a HIP device bh_create fails.
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include <algorithm>
#include <vector>

#include "bh_internal.h"

#define BH_HIP(c, call)                       \
  do {                                        \
    hipError_t _e = (call);                   \
    if (_e != hipSuccess) {                   \
      (c)->last_hip = (int)_e;                \
      return BH_ERR_HIP;                      \
    }                                         \
  } while (0)

static_assert(sizeof(bh_node) == 32, "bh_node must be 32 bytes");

template <typename T>
static hipError_t dalloc(T** p, size_t count) {
  return hipMalloc((void**)p, count * sizeof(T) + 256);
}

extern "C" {

int bh_abi_version(void) { return BH_ABI_VERSION; }

int bh_default_params(bh_params* p) {
  if (!p) return BH_ERR_BAD_ARG;
  memset(p, 0, sizeof(*p));
  p->G = 0.5f;            // ref:14
  p->theta = 0.5f;        // ref:15
  p->dt = 0.02f;          // ref:16
  p->eps2 = 50.0f;        // ref:17
  p->max_speed = 500.0f;  // ref:18
  p->leaf_cap = 1;
  p->max_depth = 21;
  p->key_bits = 63;
  p->strict_fp = 0;
  p->key_curve = 1;  // Hilbert order
  p->xcd_mode = 3;   // automatic
  return BH_OK;
}

const char* bh_strerror(int s) {
  switch (s) {
    case BH_OK: return "ok";
    case BH_ERR_BAD_ARG: return "bad argument";
    case BH_ERR_NO_DEVICE: return "no usable HIP device";
    case BH_ERR_HIP: return "HIP runtime error";
    case BH_ERR_OOM: return "out of memory";
    case BH_ERR_POOL_OVERFLOW: return "octree record pool overflow";
    case BH_ERR_ORDER: return "stage called out of order";
    case BH_ERR_SMALL_BUFFER: return "caller buffer too small";
    case BH_ERR_DEVICE_FLAG: return "device-side error flag set (see bh_get_stats().status_flags)";
    case BH_ERR_COMM: return "exchange between ranks failed (bh_comm)";
    case BH_ERR_DOMAIN_LEFT: return "a rank left the domain-decomposed step (collective: every rank returns this)";
    default: return "unknown status";
  }
}

int bh_last_hip_error(const bh_ctx* c) { return c ? c->last_hip : 0; }
int bh_n(const bh_ctx* c) { return c ? c->n : 0; }

static void drop_graphs(bh_ctx* c);

static void free_all(bh_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->dd) {  // the bound record pool is caller-owned
    c->frec = c->frec_own;
    bh_dd_free(c);
  }
  void* ptrs[] = {c->posm[0], c->posm[1], c->velid[0], c->velid[1], c->acc_own, c->stage_buf,
                  c->keys[0], c->keys[1], c->vals[0], c->vals[1], c->hist, c->sw_hist, c->sw_status,
                  c->sw_ticket, c->sp_keys, c->sp_count, c->bbox_partial, c->bounds_next, c->ibox_rows,
                  c->bounds, c->d8, c->ksamp, c->fuse_rows, c->fuse_cnt, c->pa, c->pb, c->pn,
                  c->cb, c->ttot, c->blk_done, c->rec, c->frec, c->er_lo, c->er_hi, c->P, c->info, c->scan_tmp,
                  c->cV, c->cO, c->cP};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (c->evring) {
    for (int i = 0; i < BH_TIMING_RING * 8; i++)
      if (c->evring[i]) (void)hipEventDestroy(c->evring[i]);
    free(c->evring);
  }
  drop_graphs(c);
  if (c->scan_tmp2) (void)hipFree(c->scan_tmp2);
  if (c->host_flags) (void)hipHostFree(c->host_flags);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int bh_create_on_stream(bh_ctx** out, int n, const bh_params* pin, int device, void* hip_stream) {
  if (!out) return BH_ERR_BAD_ARG;
  *out = nullptr;
  bh_params p;
  if (pin) p = *pin; else bh_default_params(&p);
  if (n < 1 || n > (1 << 30) / 2) return BH_ERR_BAD_ARG;
  if (p.key_bits != 63 && p.key_bits != 30) return BH_ERR_BAD_ARG;
  if (!(p.eps2 > 0.0f) || !(p.theta >= 0.0f) || p.leaf_cap < 1 || p.leaf_cap > 64 || p.max_depth < 0 ||
      p.force_variant < 0 || p.force_variant > 1 || p.sort_variant < 0 || p.sort_variant > 3 ||
      p.key_curve < 0 || p.key_curve > 1 || p.xcd_mode < 0 || p.xcd_mode > 3 || p.force_coop < 0 || p.force_coop > 8)
    return BH_ERR_BAD_ARG;
  if (p.key_bits == 30) p.key_curve = 0;  // the reference-literal 30-bit code is a Morton code

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return BH_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return BH_ERR_NO_DEVICE;

  bh_ctx* c = new (std::nothrow) bh_ctx();
  if (!c) return BH_ERR_OOM;
  memset((void*)c, 0, sizeof(*c));
  c->n = n;
  c->p = p;
  c->B = p.key_bits / 3;
  c->D = p.max_depth < c->B ? p.max_depth : c->B;
  c->cap = p.leaf_cap;
  c->device = device;
  c->num_cus = 256;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
      c->num_cus = cus;
  }
  if (hipSetDevice(device) != hipSuccess) {
    delete c;
    return BH_ERR_NO_DEVICE;
  }
  if (hip_stream) {
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return BH_ERR_HIP;
    }
    c->own_stream = true;
  }

  const size_t N = (size_t)n;
  c->rec_cap = BH_REC_CAP(n);
  c->sort_tiles = (n + BH_SORT_TILE - 1) / BH_SORT_TILE;
  // scans run over n (+1) ints, 256*sort_tiles ints and n fp64 quadruples
  size_t scan_n = N + 1;
  if ((size_t)256 * c->sort_tiles > scan_n) scan_n = (size_t)256 * c->sort_tiles;
  c->scan_tmp_bytes = bhk_scan_tmp_bytes((int)scan_n);
  c->scan_cnt_off = bhk_scan_cnt_offset((int)scan_n);

  bool ok = true;
  ok = ok && dalloc(&c->posm[0], N) == hipSuccess && dalloc(&c->posm[1], N) == hipSuccess;
  ok = ok && dalloc(&c->velid[0], N) == hipSuccess && dalloc(&c->velid[1], N) == hipSuccess;
  ok = ok && dalloc(&c->acc_own, N) == hipSuccess;
  c->acc = c->acc_own;
  ok = ok && dalloc(&c->stage_buf, 7 * N) == hipSuccess;
  ok = ok && dalloc(&c->keys[0], N) == hipSuccess && dalloc(&c->keys[1], N) == hipSuccess;
  ok = ok && dalloc(&c->vals[0], N) == hipSuccess && dalloc(&c->vals[1], N) == hipSuccess;
  ok = ok && dalloc(&c->hist, (size_t)256 * c->sort_tiles + 256) == hipSuccess;
  ok = ok && dalloc(&c->sw_hist, (size_t)8 * 256) == hipSuccess;
  ok = ok && dalloc(&c->sw_status, (size_t)8 * c->sort_tiles * 256) == hipSuccess;
  ok = ok && dalloc(&c->sw_ticket, (size_t)16) == hipSuccess;
  ok = ok && dalloc(&c->sp_keys, (size_t)256) == hipSuccess;
  ok = ok && dalloc(&c->sp_count, (size_t)512) == hipSuccess;
  ok = ok && hipMemset(c->sp_count, 0, 512 * sizeof(u32)) == hipSuccess;
  // look-back granules and tickets start at zero once; they are never cleared afterwards
  // (granules carry the sort-call tag, tickets are monotonic)
  ok = ok && hipMemset(c->sw_status, 0, (size_t)8 * c->sort_tiles * 256 * sizeof(u64)) == hipSuccess;
  ok = ok && hipMemset(c->sw_ticket, 0, 16 * sizeof(u32)) == hipSuccess;
  ok = ok && hipMemset(c->sw_hist, 0, 8 * 256 * sizeof(u32)) == hipSuccess;
  ok = ok && dalloc(&c->bbox_partial, (size_t)BH_BBOX_BLOCKS * 6) == hipSuccess;
  ok = ok && dalloc(&c->bounds, 8) == hipSuccess;
  ok = ok && dalloc(&c->bounds_next, 8) == hipSuccess;
  ok = ok && dalloc(&c->ibox_rows, (N / 1024 + 2) * 6) == hipSuccess;  // one row per integrate block (>= 1024 bodies)
  ok = ok && dalloc(&c->d8, N + 1) == hipSuccess;
  ok = ok && dalloc(&c->ksamp, (size_t)2048 + 8) == hipSuccess;
  // waves of a force launch over all bodies: 64 bodies per wave, 32 up to 57,344 bodies, 16 up to 20,480 (force_group)
  c->fuse_waves = (int)std::max<size_t>(N / 64 + 2, std::min<size_t>(N / 16 + 2, 2600));
  ok = ok && dalloc(&c->fuse_rows, ((size_t)c->fuse_waves + c->fuse_waves / 32 + 2) * 6) == hipSuccess;
  ok = ok && dalloc(&c->fuse_cnt, (size_t)c->fuse_waves / 32 + 3) == hipSuccess;
  ok = ok && hipMemset(c->fuse_cnt, 0, ((size_t)c->fuse_waves / 32 + 3) * sizeof(u32)) == hipSuccess;
  ok = ok && dalloc(&c->pa, N) == hipSuccess && dalloc(&c->pb, N) == hipSuccess;
  ok = ok && dalloc(&c->pn, N + 1) == hipSuccess;
  ok = ok && dalloc(&c->cb, N + 4) == hipSuccess;
  ok = ok && dalloc(&c->ttot, 2 * (N / 256 + 2)) == hipSuccess;
  ok = ok && dalloc(&c->blk_done, 2 * BH_BLKDONE_STRIDE(N)) == hipSuccess;
  ok = ok && hipMemset(c->blk_done, 0, 2 * BH_BLKDONE_STRIDE(N) * sizeof(u32)) == hipSuccess;
  c->blk_done2 = c->blk_done ? c->blk_done + BH_BLKDONE_STRIDE(N) : nullptr;
  ok = ok && dalloc(&c->rec, (size_t)c->rec_cap) == hipSuccess;
  ok = ok && dalloc(&c->frec, BH_FREC_POOL(c->rec_cap, N)) == hipSuccess;  // tree digests + body digests
  ok = ok && dalloc(&c->er_lo, (size_t)c->rec_cap) == hipSuccess;
  ok = ok && dalloc(&c->er_hi, (size_t)c->rec_cap) == hipSuccess;
  ok = ok && dalloc(&c->P, N + 1) == hipSuccess;
  ok = ok && dalloc(&c->info, 1) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->host_flags, 4 * sizeof(int)) == hipSuccess;
  ok = ok && hipMalloc(&c->scan_tmp, c->scan_tmp_bytes) == hipSuccess;
  ok = ok && hipMalloc(&c->scan_tmp2, c->scan_tmp_bytes) == hipSuccess;
  ok = ok && hipMemset(c->scan_tmp, 0, c->scan_tmp_bytes) == hipSuccess;
  ok = ok && hipMemset(c->scan_tmp2, 0, c->scan_tmp_bytes) == hipSuccess;
  ok = ok && dalloc(&c->cV, N) == hipSuccess && dalloc(&c->cO, N) == hipSuccess &&
       dalloc(&c->cP, N) == hipSuccess;
  if (!ok) {
    free_all(c);
    return BH_ERR_OOM;
  }
  // the record pool is read up to 3 records past a child block (force kernel): keep it defined
  if (hipMemsetAsync(c->rec, 0, (size_t)c->rec_cap * sizeof(bh_node), c->stream) != hipSuccess ||
      hipMemsetAsync(c->frec, 0, BH_FREC_POOL(c->rec_cap, N) * sizeof(bh_frec), c->stream) != hipSuccess ||
      hipMemsetAsync(c->info, 0, sizeof(bh_devinfo), c->stream) != hipSuccess ||
      hipMemsetAsync(c->acc, 0, N * sizeof(float4), c->stream) != hipSuccess) {
    free_all(c);
    return BH_ERR_HIP;
  }
  *out = c;
  return BH_OK;
}

int bh_create(bh_ctx** out, int n, const bh_params* p, int device) {
  return bh_create_on_stream(out, n, p, device, nullptr);
}

void bh_destroy(bh_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  free_all(c);
}

// plain wait, used by the downloads and bh_get_stats (the caller must be able to read the state and the
// flags themselves while a flag is set)
static int sync_raw(bh_ctx* c) {
  BH_HIP(c, hipStreamSynchronize(c->stream));
  return BH_OK;
}

// The splitter sort met buckets beyond its LDS capacity.  An isolated one is normal (two neighbouring splitter
// bodies that both crossed a high-level cell plane: about once in 60 steps of the 1M Plummer run, a 0.18 ms
// sort); when more than every fourth sort since the last look had one, the input defeats the splitters (many
// equal keys) and the context goes back to the radix passes until the next upload.  Looked at wherever the host
// reads the device info block anyway: bh_sync (every frame of a step loop) and bh_get_stats.
static void note_slow_buckets(bh_ctx* c, int slow_buckets) {
  const int d_slow = slow_buckets - c->slow_seen;
  const long d_sorts = (long)c->sort_calls - (long)c->slow_seen_sorts;
  if (d_slow > 0 && d_sorts > 0 && 4L * d_slow > d_sorts) c->splitter_off = true;
  c->slow_seen = slow_buckets;
  c->slow_seen_sorts = c->sort_calls;
}

int bh_sync(bh_ctx* c) {
  if (!c) return BH_ERR_BAD_ARG;
  // the sticky flags (and the slow-bucket count of the splitter sort) ride on the same synchronisation: 12 bytes
  // of the device info block — flags, redo_waves, slow_buckets — into pinned memory, then one wait
  static_assert(offsetof(bh_devinfo, slow_buckets) == offsetof(bh_devinfo, flags) + 8, "bh_devinfo layout");
  if (c->host_flags) {
    c->host_flags[0] = 0;
    c->host_flags[2] = c->slow_seen;
    BH_HIP(c, hipMemcpyAsync(c->host_flags, &c->info->flags, 3 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  }
  BH_HIP(c, hipStreamSynchronize(c->stream));
  if (c->host_flags) {
    note_slow_buckets(c, c->host_flags[2]);
    if (c->host_flags[0]) return BH_ERR_DEVICE_FLAG;
  }
  return BH_OK;
}

int bh_set_timing(bh_ctx* c, int on) {
  if (!c) return BH_ERR_BAD_ARG;
  if (on && !c->evring) {
    c->evring = (hipEvent_t*)calloc((size_t)BH_TIMING_RING * 8, sizeof(hipEvent_t));
    if (!c->evring) return BH_ERR_OOM;
    for (int i = 0; i < BH_TIMING_RING * 8; i++) BH_HIP(c, hipEventCreate(&c->evring[i]));
  }
  c->timing = on != 0;
  c->timing_mode = (on == 2 || on == 3) ? on : 1;
  c->timed_steps = 0;
  return BH_OK;
}

// ---- upload: ref:329-335 ----
int bh_upload(bh_ctx* c, const float* x, const float* y, const float* z, const float* vx,
              const float* vy, const float* vz, const float* m) {
  if (!c || !x || !y || !z || !vx || !vy || !vz || !m) return BH_ERR_BAD_ARG;
  BH_HIP(c, hipSetDevice(c->device));
  const size_t N = (size_t)c->n, nb = N * sizeof(float);
  const float* src[7] = {x, y, z, vx, vy, vz, m};
  for (int k = 0; k < 7; k++)
    BH_HIP(c, hipMemcpyAsync(c->stage_buf + k * N, src[k], nb, hipMemcpyHostToDevice, c->stream));
  c->cur = 0;
  c->order_hint = false;  // caller order: nothing for the splitter sort to exploit
  c->splitter_off = false;
  c->slow_seen = 0;       // (the counter itself is cleared with the device info block below)
  c->slow_seen_sorts = c->sort_calls;
  c->bounds_next_ok = false;
  BH_HIP(c, bhk_pack(c));
  BH_HIP(c, hipMemsetAsync(c->info, 0, sizeof(bh_devinfo), c->stream));  // clears the sticky flags
  BH_HIP(c, hipMemsetAsync(c->acc, 0, N * sizeof(float4), c->stream));
  BH_HIP(c, hipStreamSynchronize(c->stream));  // host buffers may be reused on return
  c->stage = BH_ST_UPLOADED;
  c->ever = BH_ST_UPLOADED;
  c->steps = 0;
  return BH_OK;
}

// ---- stages ----
#define BH_NEED(c, bit)                          \
  do {                                           \
    if (!(c)) return BH_ERR_BAD_ARG;             \
    if (!((c)->stage & (bit))) return BH_ERR_ORDER; \
  } while (0)
// downloads accept data of a stage that has run at least once since upload
#define BH_NEED_EVER(c, bit)                     \
  do {                                           \
    if (!(c)) return BH_ERR_BAD_ARG;             \
    if (!((c)->ever & (bit))) return BH_ERR_ORDER; \
  } while (0)

int bh_bbox(bh_ctx* c) {
  BH_NEED(c, BH_ST_UPLOADED);
  BH_HIP(c, bhk_bbox(c));
  c->stage = BH_ST_UPLOADED | BH_ST_BBOX;
  c->ever |= BH_ST_BBOX;
  return BH_OK;
}
int bh_morton(bh_ctx* c) {
  BH_NEED(c, BH_ST_BBOX);
  BH_HIP(c, bhk_keys(c));
  c->stage |= BH_ST_MORTON;
  c->ever |= BH_ST_MORTON;
  c->ever &= ~BH_ST_SORT;
  c->key_buf = 0;
  return BH_OK;
}
int bh_sort(bh_ctx* c) {
  BH_NEED(c, BH_ST_MORTON);
  if (c->stage & BH_ST_SORT) return BH_ERR_ORDER;  // sorting twice would permute twice
  // the tree of an earlier digest-only step stays downloadable: make its records canonical before the bodies
  // they refer to are reordered (stage calls only; bh_step rebuilds the tree anyway)
  if ((c->ever & BH_ST_BUILD) && c->rec_proto && c->com_digests) BH_HIP(c, bhk_canonical_records(c));
  c->com_digests = false;
  BH_HIP(c, bhk_sort(c));
  c->stage |= BH_ST_SORT;
  c->ever |= BH_ST_SORT;
  return BH_OK;
}
int bh_build(bh_ctx* c) {
  BH_NEED(c, BH_ST_SORT);
  BH_HIP(c, bhk_build(c));
  c->stage |= BH_ST_BUILD;
  c->ever |= BH_ST_BUILD;
  return BH_OK;
}
int bh_com(bh_ctx* c) {
  BH_NEED(c, BH_ST_BUILD);
  BH_HIP(c, bhk_com(c));
  c->stage |= BH_ST_COM;
  c->ever |= BH_ST_COM;
  return BH_OK;
}
int bh_force_range(bh_ctx* c, int lo, int hi) {
  BH_NEED(c, BH_ST_COM);
  if (lo < 0 || hi > c->n || lo > hi) return BH_ERR_BAD_ARG;
  if (!bhk_force_range_aligned(c, lo)) return BH_ERR_BAD_ARG;  // cooperative walk: slabs start on group boundaries
  BH_HIP(c, bhk_force(c, lo, hi, false));
  c->stage |= BH_ST_FORCE;
  c->ever |= BH_ST_FORCE;
  return BH_OK;
}
int bh_force(bh_ctx* c) {
  if (!c) return BH_ERR_BAD_ARG;
  return bh_force_range(c, 0, c->n);
}
int bh_integrate(bh_ctx* c) {
  BH_NEED(c, BH_ST_FORCE);
  c->bounds_next_ok = false;
  // domain-decomposed step: the kernel also folds this rank's min / max, the next step's X1 payload — unless the
  // step's last force pass has done both already (bh_dd_force)
  if (c->dd && c->dd_integrated) {
    c->dd_integrated = false;
  } else {
    BH_HIP(c, bhk_integrate(c, c->dd != nullptr));
  }
  c->dd_minmax_ok = c->dd != nullptr;
  c->stage = BH_ST_UPLOADED;  // positions changed: bbox..force must be redone
  return BH_OK;
}

int bh_force_walk_stats(bh_ctx* c, bh_walk_stats* out) {
  if (!out) return BH_ERR_BAD_ARG;
  BH_NEED(c, BH_ST_COM);
  if (c->p.strict_fp || c->p.literal_force || c->dd) return BH_ERR_BAD_ARG;  // the default walk only
  return bh_walk_stats_from(c, 0, out);
}

}  // extern "C"

// the counted walk of the context's bodies from pool record `root` (bh_force_walk_stats: 0; bh_dd_walk_stats: a top tree)
int bh_walk_stats_from(bh_ctx* c, int root, bh_walk_stats* out) {
  memset(out, 0, sizeof(*out));
  const size_t W = (size_t)bhk_force_walk_rows(c);
  u32* rows = nullptr;
  BH_HIP(c, hipMalloc((void**)&rows, W * BH_WALK_ROW * sizeof(u32)));
  std::vector<u32> h(W * BH_WALK_ROW);
  hipError_t e = hipMemsetAsync(rows, 0, W * BH_WALK_ROW * sizeof(u32), c->stream);
  if (e == hipSuccess) e = bhk_force_walk_stats(c, rows, root);
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), rows, W * BH_WALK_ROW * sizeof(u32), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(rows);
  if (e != hipSuccess) {
    c->last_hip = (int)e;
    return BH_ERR_HIP;
  }
  std::vector<double> ghz;
  ghz.reserve(W);
  double cyc_max = 0, cyc_sum = 0;
  for (size_t w = 0; w < W; w++) {
    const u32* r = &h[w * BH_WALK_ROW];
    if (r[6] == 0) continue;
    out->waves++;
    out->pairs += r[0];
    out->blocks += r[1];
    out->masked_pairs += r[2];
    out->lane_spills += r[3];
    out->no_taker_pairs += r[7];
    out->fetch_wait_cycles += r[8];
    if (r[5]) ghz.push_back((double)r[4] / ((double)r[5] * 10.0));  // 100 MHz ticks -> ns
    cyc_max = r[4] > cyc_max ? (double)r[4] : cyc_max;
    cyc_sum += (double)r[4];
  }
  if (!ghz.empty()) {
    std::nth_element(ghz.begin(), ghz.begin() + ghz.size() / 2, ghz.end());
    out->clock_ghz = ghz[ghz.size() / 2];
  }
  out->wave_cycles_max = cyc_max;
  out->wave_cycles_mean = out->waves ? cyc_sum / (double)out->waves : 0.0;
  return BH_OK;
}

extern "C" {

int bh_force_launch_trace(bh_ctx* c, uint32_t* rows, int capacity_rows, int* n_rows) {
  if (!rows || !n_rows || capacity_rows < 1) return BH_ERR_BAD_ARG;
  BH_NEED(c, BH_ST_COM);
  *n_rows = 0;
  u32* dev = nullptr;
  BH_HIP(c, hipMalloc((void**)&dev, (size_t)capacity_rows * 16));
  int nr = 0;
  hipError_t e = hipMemsetAsync(dev, 0, (size_t)capacity_rows * 16, c->stream);
  if (e == hipSuccess) e = bhk_force_trace(c, dev, capacity_rows, &nr);
  if (e == hipSuccess && nr > 0) e = hipMemcpyAsync(rows, dev, (size_t)nr * 16, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dev);
  if (e != hipSuccess) {
    c->last_hip = (int)e;
    return BH_ERR_HIP;
  }
  if (nr > 0) {  // the launch stored accelerations like bh_force
    c->stage |= BH_ST_FORCE;
    c->ever |= BH_ST_FORCE;
  }
  *n_rows = nr;
  return BH_OK;
}

int bh_force_count(bh_ctx* c) {
  BH_NEED(c, BH_ST_COM);
  const size_t N = (size_t)c->n;
  BH_HIP(c, bhk_force(c, 0, c->n, true));
  c->stage |= BH_ST_FORCE;
  c->ever |= BH_ST_FORCE;
  u32* h = (u32*)malloc(3 * N * sizeof(u32));
  if (!h) return BH_ERR_OOM;
  hipError_t e = hipMemcpyAsync(h, c->cV, N * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h + N, c->cO, N * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h + 2 * N, c->cP, N * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    free(h);
    c->last_hip = (int)e;
    return BH_ERR_HIP;
  }
  u64 v = 0, o = 0, p = 0;
  for (size_t i = 0; i < N; i++) {
    v += h[i];
    o += h[N + i];
    p += h[2 * N + i];
  }
  free(h);
  c->tV = v; c->tO = o; c->tP = p;
  return BH_OK;
}

// ---- the step: ref:255-283, same stage order, no host sync ----
static int step_launch(bh_ctx* c) {
  // mode 2 records only the pair around the force launch: an event record costs the stream ~7-16 us between
  // two dependent kernels, eight of them ~60 us per 1M-body step (seen as gaps in the rocprofv3 kernel trace);
  // mode 3 records that pair on every 4th step only (14 us per step is 4 % of a 65,536-body step)
  const bool t = c->timing && (c->timing_mode != 3 || (c->steps & 3) == 0);
  hipEvent_t* ev = t ? c->evring + (size_t)(c->timed_steps % BH_TIMING_RING) * 8 : nullptr;
  const bool all = t && c->timing_mode == 1;
#define BH_MARK(i) if (all || (t && ((i) == 5 || (i) == 6))) BH_HIP(c, hipEventRecord(ev[i], c->stream))
  BH_MARK(0);
  if (c->bounds_next_ok) {  // the previous step's integrate already folded the cube of these positions
    float* t = c->bounds;
    c->bounds = c->bounds_next;
    c->bounds_next = t;
    c->bounds_next_ok = false;
  } else {
    BH_HIP(c, bhk_bbox(c));                    // ref:259
  }
  BH_MARK(1);
  BH_HIP(c, bhk_keys(c));                      // ref:260
  c->key_buf = 0;
  BH_MARK(2);
  BH_HIP(c, bhk_sort(c));                      // ref:262-264
  BH_MARK(3);
  // One stream: the COM prefix scan (which needs only the sorted bodies) rides in the launches of the tree build
  // (which needs only the sorted keys) — bhk_build pm_scan.  Rounds 2-4 ran body gather + scan on a second stream
  // beside the build; each hand-over between the streams cost ~7 us of gaps, and with the scan's tiles in the pairs
  // / emit launches one stream measures faster at every size (profiles/r04_final/fork_sweep.txt).
  BH_HIP(c, bhk_build(c, true));               // ref:266-275
  BH_MARK(4);
  // digests only unless something reads the canonical records after this step (strict / literal kernels)
  BH_HIP(c, bhk_com_records(c, c->p.strict_fp || c->p.literal_force || c->dd));  // ref:279-280
  BH_MARK(5);
  // Outside per-stage timing the force launch also integrates (ref:282) and folds the next step's cube (ref:259):
  // force_fast_kernel FUSE.  The timed pair (5, 6) then brackets force + integrate.
  const bool fuse_bbox = !c->dd && c->p.step_graph != 1;
  bool fused = false;
  BH_HIP(c, bhk_force(c, 0, c->n, false, fuse_bbox && !all, &fused));     // ref:281
  BH_MARK(6);
  if (!fused) BH_HIP(c, bhk_integrate(c, fuse_bbox));      // ref:282 (+ ref:259 of the next step)
  c->bounds_next_ok = fuse_bbox;
  BH_MARK(7);
#undef BH_MARK
  if (t) c->timed_steps++;
  c->stage = BH_ST_UPLOADED;
  c->ever |= BH_ST_BBOX | BH_ST_MORTON | BH_ST_SORT | BH_ST_BUILD | BH_ST_COM | BH_ST_FORCE;
  c->steps++;
  return BH_OK;
}

static void drop_graphs(bh_ctx* c) {
  for (int k = 0; k < 2; k++)
    if (c->gexec[k]) {
      (void)hipGraphExecDestroy(c->gexec[k]);
      c->gexec[k] = nullptr;
    }
}

// The reference's step is one function with one synchronisation (ref:255-283); here it is 8 kernels on one
// stream whose arguments never change from step to step except for the ping-pong parity of the body arrays
// (the sort gathers cur -> cur^1): one HIP graph per parity, captured the first time that parity is stepped
// and replayed afterwards.  Not used while per-stage timing is on (event records between the stages), in
// domain-decomposed mode (body count changes) (design-study builds, -DBH_STUDY, also honour BH_NO_GRAPH).  OPT-IN (bh_params.step_graph = 1):
// on ROCm 7.2 the replay measured SLOWER than the plain launches (1.93 vs 1.85 ms at 1M bodies, 0.66 vs 0.57 ms
// at 65,536: graph kernel nodes are dispatched with more packet overhead than back-to-back stream launches).
int bh_step(bh_ctx* c) {
  BH_NEED(c, BH_ST_UPLOADED);
#ifdef BH_STUDY
  static const bool env_off = getenv("BH_NO_GRAPH") != nullptr;
#else
  constexpr bool env_off = false;
#endif
  if (c->timing || c->dd || c->p.step_graph != 1 || env_off || c->graph_failed) return step_launch(c);
  const int par = c->cur;
  if (!c->gexec[par]) {
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      c->graph_failed = true;
      return step_launch(c);
    }
    const int s = step_launch(c);  // records the launches; the host-side state (cur, key_buf, ...) advances as usual
    const hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (s != BH_OK || e != hipSuccess || !g ||
        hipGraphInstantiate(&c->gexec[par], g, nullptr, nullptr, 0) != hipSuccess) {
      if (g) (void)hipGraphDestroy(g);
      c->gexec[par] = nullptr;
      c->graph_failed = true;  // nothing was executed: undo the bookkeeping and take the plain path from now on
      c->cur = par;
      c->steps--;
      c->sort_calls--;
      (void)hipGetLastError();
      return s != BH_OK ? s : step_launch(c);
    }
    (void)hipGraphDestroy(g);
    c->g_keybuf[par] = c->key_buf;
    BH_HIP(c, hipGraphLaunch(c->gexec[par], c->stream));
    return BH_OK;
  }
  BH_HIP(c, hipGraphLaunch(c->gexec[par], c->stream));
  c->cur = par ^ 1;  // the bookkeeping step_launch does next to its launches
  c->key_buf = c->g_keybuf[par];
  c->sort_calls++;
  // what the replayed COM stage leaves behind (bhk_build / bhk_com_records set these next to their launches, which a
  // replay does not run on the host): proto records again unless the engine keeps canonical ones, digests of THIS tree
  c->rec_proto = !(c->p.strict_fp || c->p.literal_force);
  c->com_digests = true;
  c->stage = BH_ST_UPLOADED;
  c->steps++;
  return BH_OK;
}

// ---- downloads ----
static int d2h(bh_ctx* c, void* dst, const void* src, size_t bytes) {
  BH_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  return BH_OK;
}

int bh_download(bh_ctx* c, float* x, float* y, float* z, float* vx, float* vy, float* vz) {
  BH_NEED(c, BH_ST_UPLOADED);
  const size_t N = (size_t)c->n, nb = N * sizeof(float);
  BH_HIP(c, bhk_unpack(c, 0));
  float* dst[6] = {x, y, z, vx, vy, vz};
  for (int k = 0; k < 6; k++)
    if (dst[k]) {
      int s = d2h(c, dst[k], c->stage_buf + k * N, nb);
      if (s) return s;
    }
  return sync_raw(c);
}

int bh_download_acc(bh_ctx* c, float* ax, float* ay, float* az) {
  BH_NEED(c, BH_ST_UPLOADED);
  if (!ax || !ay || !az) return BH_ERR_BAD_ARG;
  const size_t N = (size_t)c->n, nb = N * sizeof(float);
  BH_HIP(c, bhk_unpack(c, 1));
  int s;
  if ((s = d2h(c, ax, c->stage_buf, nb))) return s;
  if ((s = d2h(c, ay, c->stage_buf + N, nb))) return s;
  if ((s = d2h(c, az, c->stage_buf + 2 * N, nb))) return s;
  return sync_raw(c);
}

int bh_download_mass(bh_ctx* c, float* m) {
  BH_NEED(c, BH_ST_UPLOADED);
  if (!m) return BH_ERR_BAD_ARG;
  const size_t N = (size_t)c->n;
  BH_HIP(c, bhk_unpack(c, 0));  // slot 6 of the staging buffer = mass in caller order
  int s = d2h(c, m, c->stage_buf + 6 * N, N * sizeof(float));
  if (s) return s;
  return sync_raw(c);
}

int bh_export_visual(bh_ctx* c, float* pos_xyz, float* col_rgb) {
  BH_NEED(c, BH_ST_UPLOADED);
  if (!pos_xyz || !col_rgb) return BH_ERR_BAD_ARG;
  const size_t N = (size_t)c->n;
  BH_HIP(c, bhk_unpack(c, 3));
  int s;
  if ((s = d2h(c, pos_xyz, c->stage_buf, 3 * N * sizeof(float)))) return s;
  if ((s = d2h(c, col_rgb, c->stage_buf + 3 * N, 3 * N * sizeof(float)))) return s;
  return sync_raw(c);
}

int bh_download_counters(bh_ctx* c, uint32_t* V, uint32_t* O, uint32_t* P) {
  BH_NEED_EVER(c, BH_ST_FORCE);
  if (!V || !O || !P) return BH_ERR_BAD_ARG;
  const size_t N = (size_t)c->n, nb = N * sizeof(u32);
  BH_HIP(c, bhk_unpack(c, 2));
  const u32* s = (const u32*)c->stage_buf;
  int r;
  if ((r = d2h(c, V, s, nb))) return r;
  if ((r = d2h(c, O, s + N, nb))) return r;
  if ((r = d2h(c, P, s + 2 * N, nb))) return r;
  return sync_raw(c);
}

int bh_download_bounds(bh_ctx* c, float bounds[6]) {
  BH_NEED_EVER(c, BH_ST_BBOX);
  if (!bounds) return BH_ERR_BAD_ARG;
  int s = d2h(c, bounds, c->bounds, 6 * sizeof(float));
  if (s) return s;
  return sync_raw(c);
}

int bh_download_keys(bh_ctx* c, uint64_t* keys) {
  BH_NEED_EVER(c, BH_ST_MORTON);
  if (!keys) return BH_ERR_BAD_ARG;
  const int buf = (c->ever & BH_ST_SORT) ? c->key_buf : 0;
  int s = d2h(c, keys, c->keys[buf], (size_t)c->n * sizeof(u64));
  if (s) return s;
  return sync_raw(c);
}

int bh_download_order(bh_ctx* c, int32_t* ids) {
  BH_NEED(c, BH_ST_UPLOADED);
  if (!ids) return BH_ERR_BAD_ARG;
  const size_t N = (size_t)c->n;
  float4* h = (float4*)malloc(N * sizeof(float4));
  if (!h) return BH_ERR_OOM;
  int s = d2h(c, h, c->velid[c->cur], N * sizeof(float4));
  if (!s) s = sync_raw(c);
  if (!s)
    for (size_t i = 0; i < N; i++) memcpy(&ids[i], &h[i].w, 4);
  free(h);
  return s;
}

int bh_download_sorted_bodies(bh_ctx* c, float* xyzm) {
  BH_NEED(c, BH_ST_UPLOADED);
  if (!xyzm) return BH_ERR_BAD_ARG;
  int s = d2h(c, xyzm, c->posm[c->cur], (size_t)c->n * sizeof(float4));
  if (s) return s;
  return sync_raw(c);
}

static int fetch_info(bh_ctx* c, bh_devinfo* h) {
  int s = d2h(c, h, c->info, sizeof(bh_devinfo));
  if (s) return s;
  return sync_raw(c);
}

int bh_download_tree(bh_ctx* c, bh_node* out, int capacity, int* n_entries) {
  BH_NEED_EVER(c, BH_ST_BUILD);
  // bh_step of the default engine writes only the force kernel's digests: the canonical records are then made on
  // demand, here (the step's prefix sums and digests are still in place).  After bh_build alone there is no
  // centre of mass to serve yet.
  if (out && c->rec_proto) {
    if (!c->com_digests) return BH_ERR_ORDER;
    BH_HIP(c, bhk_canonical_records(c));
  }
  bh_devinfo hi;
  int s = fetch_info(c, &hi);
  if (s) return s;
  if (hi.flags & BH_FLAG_POOL_OVERFLOW) return BH_ERR_POOL_OVERFLOW;
  if (n_entries) *n_entries = hi.n_entries;
  if (!out) return BH_OK;
  if (capacity < hi.n_entries) return BH_ERR_SMALL_BUFFER;
  s = d2h(c, out, c->rec, (size_t)hi.n_entries * sizeof(bh_node));
  if (s) return s;
  return sync_raw(c);
}

int bh_get_stats(bh_ctx* c, bh_stats* st) {
  if (!c || !st) return BH_ERR_BAD_ARG;
  memset(st, 0, sizeof(*st));
  bh_devinfo hi;
  int s = fetch_info(c, &hi);
  if (s) return s;
  st->n = c->n;
  st->n_internal = hi.n_internal;
  st->n_entries = hi.n_entries;
  st->max_level = hi.max_level;
  st->status_flags = hi.flags;
  st->steps = c->steps;
  if (c->timing && c->timing_mode >= 2 && c->timed_steps > 0) {
    hipEvent_t* ev = c->evring + (size_t)((c->timed_steps - 1) % BH_TIMING_RING) * 8;
    BH_HIP(c, hipEventElapsedTime(&st->ms_force, ev[5], ev[6]));
  } else if (c->timing && c->timed_steps > 0) {
    hipEvent_t* ev = c->evring + (size_t)((c->timed_steps - 1) % BH_TIMING_RING) * 8;
    float* ms[7] = {&st->ms_bbox, &st->ms_morton, &st->ms_sort, &st->ms_build,
                    &st->ms_com, &st->ms_force, &st->ms_integrate};
    for (int i = 0; i < 7; i++) BH_HIP(c, hipEventElapsedTime(ms[i], ev[i], ev[i + 1]));
    BH_HIP(c, hipEventElapsedTime(&st->ms_step, ev[0], ev[7]));
  }
  st->force_redo_waves = hi.redo_waves;
  st->sort_slow_buckets = hi.slow_buckets;
  note_slow_buckets(c, hi.slow_buckets);
  st->count_V = c->tV;
  st->count_O = c->tO;
  st->count_P = c->tP;
  if (hi.flags & BH_FLAG_POOL_OVERFLOW) return BH_ERR_POOL_OVERFLOW;
  return BH_OK;
}

int bh_timing_history(bh_ctx* c, float* ms_force, float* ms_step, int capacity, int* count) {
  if (!c || !count) return BH_ERR_BAD_ARG;
  *count = 0;
  if (!c->timing || c->timed_steps == 0) return BH_OK;
  int s = sync_raw(c);
  if (s) return s;
  const long have = c->timed_steps < BH_TIMING_RING ? c->timed_steps : BH_TIMING_RING;
  const long take = have < capacity ? have : capacity;
  for (long i = 0; i < take; i++) {
    const long step = c->timed_steps - take + i;
    hipEvent_t* ev = c->evring + (size_t)(step % BH_TIMING_RING) * 8;
    if (ms_force) BH_HIP(c, hipEventElapsedTime(&ms_force[i], ev[5], ev[6]));
    if (ms_step) {
      if (c->timing_mode >= 2) ms_step[i] = 0.0f;  // not recorded in these modes
      else BH_HIP(c, hipEventElapsedTime(&ms_step[i], ev[0], ev[7]));
    }
  }
  *count = (int)take;
  return BH_OK;
}

int bh_bind_acc(bh_ctx* c, void* device_float4_n) {
  if (!c) return BH_ERR_BAD_ARG;
  BH_HIP(c, hipStreamSynchronize(c->stream));
  drop_graphs(c);  // the captured steps write the old buffer
  c->acc = device_float4_n ? (float4*)device_float4_n : c->acc_own;
  return BH_OK;
}

int bh_device_acc(bh_ctx* c, void** dptr, int64_t* bytes) {
  if (!c || !dptr) return BH_ERR_BAD_ARG;
  *dptr = (void*)c->acc;
  if (bytes) *bytes = (int64_t)c->n * (int64_t)sizeof(float4);
  return BH_OK;
}

}  // extern "C"
