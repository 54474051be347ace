// bh_group.hip — the multi-GPU step behind the C-ABI (include/bh.h: bh_comm, bh_rank_*, bh_group; SURVEY §8b
// "bh_create_group / bh_step_group").  The reference is single-GPU: simulationStep() (nbody_v5_bench.cu:255-283)
// and main() (:285-390) are what bh_step_group and a C++ host around it replace for a whole node.
//
//   bh_comm   how bytes travel between ranks: all_gather + all_to_all on a HIP stream.  Three transports:
//               RCCL           ncclAllGather / ncclAllToAll (grouped ncclSend + ncclRecv if the library lacks it) on
//                              the rank's own stream, librccl.so.1 opened on first use;
//               hub            the ranks of one process: device-to-device copies on each rank's stream, ordered
//                              by events and two host barriers per exchange (one-GPU rehearsal; RCCL refuses two
//                              ranks on one device);
//               callbacks      the caller fills the struct (dist.py: torch.distributed, gloo on CPU tensors).
//   bh_rank   one rank: context + eight exchange buffers + the per-step PROTOCOL around the bh_dd_phase_* calls
//             of bh_dd.hip:
//               [X1 all-gather] phase_migrate [X2 all-gather] phase_tree ( + further X2 rounds while some rank
//               still holds emigrants ) [X3 all-gather] phase_let [X4 all-to-all] phase_force ( repeated with a
//               larger stride while some rank's LET did not fit ) phase_end
//             Sizes that can overflow (X2 slots, X4 stride) are functions of all-gathered counts, so every rank
//             takes the same decision without a further exchange.  A rank whose engine call fails keeps taking part
//             in the exchanges of that step with EMPTY payloads and marks every X4 segment it sends (header count
//             -1): all ranks then return BH_ERR_DOMAIN_LEFT from the same step — nobody is stranded in a
//             collective.  The same protocol runs around a scripted engine on host buffers (bh_rank_create_scripted)
//             so that it is tested without a GPU.
//   bh_group  the ranks of one process, one persistent host thread per rank (a rank's calls stay on one thread
//             with its device current).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "bh_internal.h"

namespace {

constexpr int kMaxWorld = 64;
constexpr int kLetMin = 4 + BH_DD_PIECE_CAP;  // header + needs row + piece slots (bh_dd.hip kSegBlocks0)

inline long long round_up(long long v, long long a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------ RCCL, opened on first use
typedef void* nccl_comm_t;
struct nccl_uid {
  char internal[128];
};
struct rccl_api {
  void* h = nullptr;
  int (*GetUniqueId)(nccl_uid*) = nullptr;
  int (*CommInitRank)(nccl_comm_t*, int, nccl_uid, int) = nullptr;
  int (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*CommAbort)(nccl_comm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllToAll)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;  // RCCL extension
  int (*Send)(const void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  bool ok = false;
};
constexpr int kNcclInt8 = 0;  // ncclInt8 / ncclChar

rccl_api* rccl() {
  static rccl_api api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      api.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (api.h) break;
    }
    if (!api.h) return;
#define BH_SYM(field, name) *(void**)(&api.field) = dlsym(api.h, name)
    BH_SYM(GetUniqueId, "ncclGetUniqueId");
    BH_SYM(CommInitRank, "ncclCommInitRank");
    BH_SYM(CommInitAll, "ncclCommInitAll");
    BH_SYM(CommDestroy, "ncclCommDestroy");
    BH_SYM(CommAbort, "ncclCommAbort");
    BH_SYM(AllGather, "ncclAllGather");
    BH_SYM(AllToAll, "ncclAllToAll");
    BH_SYM(Send, "ncclSend");
    BH_SYM(Recv, "ncclRecv");
    BH_SYM(GroupStart, "ncclGroupStart");
    BH_SYM(GroupEnd, "ncclGroupEnd");
#undef BH_SYM
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommInitAll && api.CommDestroy && api.AllGather &&
             api.Send && api.Recv && api.GroupStart && api.GroupEnd;
  });
  return api.ok ? &api : nullptr;
}

struct rccl_user {
  nccl_comm_t comm;
  bool own;
  int world, rank;
};

int rccl_all_gather(void* user, void* recv, const void* send, int64_t bytes, void* stream) {
  rccl_user* u = (rccl_user*)user;
  return rccl()->AllGather(send, recv, (size_t)bytes, kNcclInt8, u->comm, (hipStream_t)stream);
}

int rccl_all_to_all(void* user, void* recv, const void* send, int64_t bytes, void* stream) {
  rccl_user* u = (rccl_user*)user;
  rccl_api* a = rccl();
  if (a->AllToAll) return a->AllToAll(send, recv, (size_t)bytes, kNcclInt8, u->comm, (hipStream_t)stream);
  int s = a->GroupStart();
  for (int q = 0; q < u->world && !s; q++) {
    s = a->Send((const char*)send + (size_t)q * bytes, (size_t)bytes, kNcclInt8, q, u->comm, (hipStream_t)stream);
    if (!s) s = a->Recv((char*)recv + (size_t)q * bytes, (size_t)bytes, kNcclInt8, q, u->comm, (hipStream_t)stream);
  }
  const int e = a->GroupEnd();
  return s ? s : e;
}

int rccl_all_to_all_v(void* user, void* recv, const void* send, int64_t slot, const int64_t* sb, const int64_t* rb,
                      void* stream) {
  rccl_user* u = (rccl_user*)user;
  rccl_api* a = rccl();
  int s = a->GroupStart();
  for (int q = 0; q < u->world && !s; q++) {
    if (sb[q] > 0)
      s = a->Send((const char*)send + (size_t)q * slot, (size_t)sb[q], kNcclInt8, q, u->comm, (hipStream_t)stream);
    if (!s && rb[q] > 0)
      s = a->Recv((char*)recv + (size_t)q * slot, (size_t)rb[q], kNcclInt8, q, u->comm, (hipStream_t)stream);
  }
  const int e = a->GroupEnd();
  return s ? s : e;
}

void rccl_release(void* user) {
  rccl_user* u = (rccl_user*)user;
  if (u->own && u->comm && rccl()) (void)rccl()->CommDestroy(u->comm);
  free(u);
}

int rccl_fill(bh_comm* out, nccl_comm_t comm, bool own, int world, int rank) {
  rccl_user* u = (rccl_user*)calloc(1, sizeof(rccl_user));
  if (!u) return BH_ERR_OOM;
  u->comm = comm;
  u->own = own;
  u->world = world;
  u->rank = rank;
  out->world = world;
  out->rank = rank;
  out->user = u;
  out->all_gather = rccl_all_gather;
  out->all_to_all = rccl_all_to_all;
  out->release = rccl_release;
  out->all_to_all_v = rccl_all_to_all_v;
  return BH_OK;
}

}  // namespace

// ------------------------------------------------------------------ hub: the ranks of one process
struct bh_hub {
  int world;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long gen = 0;
  bool aborted = false;
  const void* src[kMaxWorld];
  hipStream_t stream[kMaxWorld];
  hipEvent_t ev_post[kMaxWorld], ev_done[kMaxWorld];
  int ev_dev[kMaxWorld];
  bool have_ev[kMaxWorld];

  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    const long g = gen;
    if (++arrived == world) {
      arrived = 0;
      gen++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g || aborted; });
    }
    return !aborted;
  }
};

namespace {

struct hub_user {
  bh_hub* hub;
  int rank;
};

// mode 0: all-gather (every rank reads all of rank q's send); 1: all-to-all (chunk `rank` of it)
int hub_exchange(void* user, void* recv, const void* send, int64_t bytes, void* stream_, int mode,
                 const int64_t* recv_bytes = nullptr) {  // recv_bytes: mode 1 with a size per pair (slots `bytes` apart)
  hub_user* u = (hub_user*)user;
  bh_hub* h = u->hub;
  const int r = u->rank, P = h->world;
  hipStream_t st = (hipStream_t)stream_;
  if (!h->have_ev[r]) {  // created by the rank's own thread, its device current
    if (hipEventCreateWithFlags(&h->ev_post[r], hipEventDisableTiming) != hipSuccess) return 1;
    if (hipEventCreateWithFlags(&h->ev_done[r], hipEventDisableTiming) != hipSuccess) return 1;
    (void)hipGetDevice(&h->ev_dev[r]);
    h->have_ev[r] = true;
  }
  if (hipEventRecord(h->ev_post[r], st) != hipSuccess) return 1;  // my payload is complete at this point of my stream
  h->src[r] = send;
  h->stream[r] = st;
  if (!h->barrier()) return 1;
  for (int q = 0; q < P; q++) {
    if (h->stream[q] != st && hipStreamWaitEvent(st, h->ev_post[q], 0) != hipSuccess) return 1;
    const char* s = (const char*)h->src[q] + (mode ? (size_t)r * bytes : 0);
    const size_t nbytes = recv_bytes ? (size_t)recv_bytes[q] : (size_t)bytes;
    if (nbytes && hipMemcpyAsync((char*)recv + (size_t)q * bytes, s, nbytes, hipMemcpyDefault, st) != hipSuccess) return 1;
  }
  if (hipEventRecord(h->ev_done[r], st) != hipSuccess) return 1;  // I have read everybody's payload
  if (!h->barrier()) return 1;
  for (int q = 0; q < P; q++)  // nobody overwrites a payload a peer still reads
    if (h->stream[q] != st && hipStreamWaitEvent(st, h->ev_done[q], 0) != hipSuccess) return 1;
  return 0;
}
int hub_all_gather(void* u, void* recv, const void* send, int64_t bytes, void* st) {
  return hub_exchange(u, recv, send, bytes, st, 0);
}
int hub_all_to_all(void* u, void* recv, const void* send, int64_t bytes, void* st) {
  return hub_exchange(u, recv, send, bytes, st, 1);
}
int hub_all_to_all_v(void* u, void* recv, const void* send, int64_t slot, const int64_t* sb, const int64_t* rb, void* st) {
  (void)sb;  // (every reader copies what it receives)
  return hub_exchange(u, recv, send, slot, st, 1, rb);
}
void hub_release(void* user) { free(user); }

}  // namespace

// ------------------------------------------------------------------ rank
struct rank_ops {  // the engine's side of a step: bh_dd_* of a context, or a script
  void* user;
  int (*cube_pack)(void*, void*);
  int (*phase_migrate)(void*, const void*, void*, int);
  int (*migrate_pack)(void*, void*, int);
  int (*phase_tree)(void*, const void*, int, void*, int*, int*, int*);
  int (*phase_let)(void*, const void*, void*, int, int);
  int (*phase_force)(void*, const void*, int, int32_t*, int*);
  int (*phase_end)(void*, void*);
};

struct bh_rank {
  bh_comm comm;
  rank_ops ops;
  bool device_mem;  // buffers live on the device (false: scripted rank, host memory)
  bh_ctx* ctx;
  int device;
  hipStream_t stream;
  bool own_stream;
  bh_rank_plan plan;
  bh_rank_opts o;
  char* buf[8];  // x1s x1r x2s x2r x3s x3r x4s pool
  bool own_buf;
  int stride, mig_stride, mig_rounds, mig_last, let_retries, n_loc;
  int stride_last;  // the stride the last completed step's X4 used (the segments in the pool are laid out with it)
  long long x4_recv_bytes;  // what this rank received in the last X4
  // which form the force takes this step: fixed by bh_rank_opts.split 0 / 1, or — split 2, adaptive — decided from the
  // measured duration of X4 (events around the exchange on the rank's stream, read one step later): one pass while the
  // exchange is short, the first part of the bodies in two passes once it lasts long enough to pay (kSplitOnMs)
  int split_now;
  bool adaptive, x4_timed;
  hipEvent_t ev_xa, ev_xb;
  float x4_ema_ms;
  int x4_samples;
  bool x1_ready;
  int32_t let_counts[kMaxWorld];
  int left_rank, left_status;
  long long steps;
  // profile: one event per phase boundary and profiled step
  bool prof;
  std::vector<hipEvent_t>* prof_ev;  // 7 per step
  std::vector<int32_t>* log;         // (emigrants, boundary action) per step
};

namespace {

enum { X1S = 0, X1R, X2S, X2R, X3S, X3R, X4S, POOL };

int op_cube_pack(void* u, void* a) { return bh_dd_cube_pack((bh_ctx*)u, a); }
int op_phase_migrate(void* u, const void* a, void* b, int l) { return bh_dd_phase_migrate((bh_ctx*)u, a, b, l); }
int op_migrate_pack(void* u, void* a, int l) { return bh_dd_migrate_pack((bh_ctx*)u, a, l); }
int op_phase_tree(void* u, const void* a, int l, void* b, int* n, int* more, int* most) {
  return bh_dd_phase_tree((bh_ctx*)u, a, l, b, n, more, most);
}
int op_phase_let(void* u, const void* a, void* b, int s, int own) { return bh_dd_phase_let((bh_ctx*)u, a, b, s, own); }
int op_phase_force(void* u, const void* a, int s, int32_t* c, int* f) {
  return bh_dd_phase_force((bh_ctx*)u, a, s, c, f);
}
int op_phase_end(void* u, void* a) { return bh_dd_phase_end((bh_ctx*)u, a); }

int rk_zero(bh_rank* r, void* p, size_t n) {
  if (!r->device_mem) {
    memset(p, 0, n);
    return 0;
  }
  return hipMemsetAsync(p, 0, n, r->stream) == hipSuccess ? 0 : 1;
}

// header words (found, kept, sent, held) of every rank's X2 payload, on the host
int rk_x2_headers(bh_rank* r, size_t nb, int h[][4]) {
  const int P = r->comm.world;
  if (!r->device_mem) {
    for (int q = 0; q < P; q++) memcpy(h[q], r->buf[X2R] + (size_t)q * nb, 16);
    return 0;
  }
  if (hipMemcpy2DAsync(h, 16, r->buf[X2R], nb, 16, (size_t)P, hipMemcpyDeviceToHost, r->stream) != hipSuccess) return 1;
  return hipStreamSynchronize(r->stream) == hipSuccess ? 0 : 1;
}

// "this rank failed": header count -1 (record 0 of a digest pair: field `first` is dword 10, bh_internal.h) in every
// segment it sends, everything else zero
int rk_mark_left(bh_rank* r, int nseg, int stride) {
  if (rk_zero(r, r->buf[X4S], (size_t)nseg * stride * 32)) return 1;
  for (int s = 0; s < nseg; s++) {
    char* p = r->buf[X4S] + (size_t)s * stride * 32 + 40;
    if (!r->device_mem)
      memset(p, 0xFF, 4);
    else if (hipMemsetAsync(p, 0xFF, 4, r->stream) != hipSuccess)
      return 1;
  }
  return 0;
}

void rk_mark(bh_rank* r, int k) {
  if (!r->prof || !r->device_mem) return;
  hipEvent_t ev;
  if (hipEventCreate(&ev) != hipSuccess) return;
  (void)hipEventRecord(ev, r->stream);
  if (k == 0) r->prof_ev->resize(r->prof_ev->size() + 7, nullptr);
  if (r->prof_ev->size() < 7) {
    (void)hipEventDestroy(ev);
    return;
  }
  hipEvent_t& slot = (*r->prof_ev)[r->prof_ev->size() - 7 + k];
  if (slot) (void)hipEventDestroy(slot);  // a repeated phase (LET retry) keeps its last mark
  slot = ev;
}

void rk_prof_clear(bh_rank* r) {
  for (hipEvent_t e : *r->prof_ev)
    if (e) (void)hipEventDestroy(e);
  r->prof_ev->clear();
}

// Ranks that share one GPU (bh_rank_opts.serial: rehearsals) enqueue a PHASE at a time: the kernels of one rank's phase
// then run back to back, as on a GPU of its own, instead of interleaved with the other ranks' (whose working sets
// would evict what one kernel of the phase leaves in the caches for the next: local_sort_kernel measured 60 us that
// way against 40 on a dedicated GPU).
std::mutex g_phase_mutex;
struct phase_lock {
  bool on;
  explicit phase_lock(const bh_rank* r) : on(r->o.serial != 0) {
    if (on) g_phase_mutex.lock();
  }
  ~phase_lock() {
    if (on) g_phase_mutex.unlock();
  }
};

// Adaptive form (bh_rank_opts.split 2).  Measured on an MI355X, 8 x 1M bodies, every rank's force phase replayed on a GPU
// of its own (bh_rank_replay_force_phase, profiles/r05_dd/replay_8x1M.txt; mean of the ranks, X4 = 0 / 0.125 / 0.25 ms):
// one pass 1.240 / 1.371 / 1.499 ms; the first 20 % in two passes 1.308 / 1.349 / 1.448; 30 %: 1.360 / 1.358 / 1.391 —
// the own pass hides the exchange at the price of two more launches with a drain each, so the split pays from an X4
// of ~0.1 ms on and by 0.1 ms at 0.25; switched on at 0.15 ms (off again below 0.09) to stay clear of the break-even.
// The decision is the rank's own (the protocol does not change).
constexpr float kSplitOnMs = 0.15f, kSplitOffMs = 0.09f;
void rank_adapt(bh_rank* r) {
  if (!r->adaptive || !r->x4_timed || !r->ctx) return;
  r->x4_timed = false;
  float t = 0.0f;
  if (hipEventElapsedTime(&t, r->ev_xa, r->ev_xb) != hipSuccess) {
    (void)hipGetLastError();  // (not complete: cannot be — the step that recorded them has checked X4's headers)
    return;
  }
  r->x4_ema_ms = r->x4_samples == 0 ? t : 0.75f * r->x4_ema_ms + 0.25f * t;
  if (++r->x4_samples < 4) return;
  if (!r->split_now && r->x4_ema_ms > kSplitOnMs) {
    // as many bodies as it takes for their own pass to outlast the exchange: an own pass of p % lasts ~0.0107 p ms
    // at 1M bodies per rank and covers ~0.65 of that (its own drain is not filled while the main stream waits)
    int pct = (int)(r->x4_ema_ms / (0.0107f * 0.65f) * (1.0e6f / (float)(r->n_loc > 0 ? r->n_loc : 1)));
    pct = pct < 20 ? 20 : (pct > 60 ? 60 : pct);
    if (bh_dd_set_split_percent(r->ctx, pct) == BH_OK) r->split_now = 1;
  } else if (r->split_now && r->x4_ema_ms < kSplitOffMs) {
    if (bh_dd_set_one_pass(r->ctx) == BH_OK) r->split_now = 0;
  }
}

int rank_one_step(bh_rank* r) {
  const bh_comm& c = r->comm;
  rank_ops e = r->ops;
  rank_adapt(r);
  {  // every engine call of the step under the phase lock (serial ranks only)
    struct wrap {
      static int cube_pack(void* u, void* a) { bh_rank* r = (bh_rank*)u; phase_lock l(r); return r->ops.cube_pack(r->ops.user, a); }
      static int phase_migrate(void* u, const void* a, void* b, int n) { bh_rank* r = (bh_rank*)u; phase_lock l(r); return r->ops.phase_migrate(r->ops.user, a, b, n); }
      static int migrate_pack(void* u, void* a, int n) { bh_rank* r = (bh_rank*)u; phase_lock l(r); return r->ops.migrate_pack(r->ops.user, a, n); }
      static int phase_tree(void* u, const void* a, int n, void* b, int* x, int* y, int* z) { bh_rank* r = (bh_rank*)u; phase_lock l(r); return r->ops.phase_tree(r->ops.user, a, n, b, x, y, z); }
      static int phase_let(void* u, const void* a, void* b, int n, int o) { bh_rank* r = (bh_rank*)u; phase_lock l(r); return r->ops.phase_let(r->ops.user, a, b, n, o); }
      static int phase_force(void* u, const void* a, int n, int32_t* cts, int* f) { bh_rank* r = (bh_rank*)u; phase_lock l(r); return r->ops.phase_force(r->ops.user, a, n, cts, f); }
      static int phase_end(void* u, void* a) { bh_rank* r = (bh_rank*)u; phase_lock l(r); return r->ops.phase_end(r->ops.user, a); }
    };
    if (r->o.serial)
      e = rank_ops{r, wrap::cube_pack, wrap::phase_migrate, wrap::migrate_pack, wrap::phase_tree, wrap::phase_let,
                   wrap::phase_force, wrap::phase_end};
  }
  const bh_dd_sizes& sz = r->plan.sz;
  const int P = c.world;
  void* st = (void*)r->stream;
  rk_mark(r, 0);
  if (!r->x1_ready) {  // first step only: afterwards the previous step packed the X1 payload
    const int s = e.cube_pack(e.user, r->buf[X1S]);
    if (s) return s;  // nothing exchanged yet: a plain rank-local error
  }
  r->x1_ready = false;
  if (c.all_gather(c.user, r->buf[X1R], r->buf[X1S], sz.x1_bytes, st)) return BH_ERR_COMM;  // X1: cube + splitters
  rk_mark(r, 1);
  // X2: bodies that changed owner.  A rank-local failure must not strand the others inside a collective: the failing
  // rank keeps taking part with empty payloads and marks its X4 segments.
  int limit = r->mig_stride, first = -1, failed = 0, rounds = 0, more = 0, most = 0;
  for (;;) {
    const size_t nb = 32 + 32 * (size_t)limit;
    if (failed) {
      if (rk_zero(r, r->buf[X2S], 32)) return BH_ERR_HIP;
    } else if (rounds == 0) {
      failed = e.phase_migrate(e.user, r->buf[X1R], r->buf[X2S], limit);  // global cube + splitters, emigrants packed
      if (failed && rk_zero(r, r->buf[X2S], 32)) return BH_ERR_HIP;
    } else {
      failed = e.migrate_pack(e.user, r->buf[X2S], limit);
      if (failed && rk_zero(r, r->buf[X2S], 32)) return BH_ERR_HIP;
    }
    rounds++;
    if (c.all_gather(c.user, r->buf[X2R], r->buf[X2S], (int64_t)nb, st)) return BH_ERR_COMM;
    if (!failed) {  // immigrants absorbed; once no rank has emigrants left: local sort / build / COM, X3 descriptors
      int nl = 0;
      failed = e.phase_tree(e.user, r->buf[X2R], limit, r->buf[X3S], &nl, &more, &most);
      if (!failed) r->n_loc = nl;
    }
    if (failed) {  // what the engine would have told: from the gathered headers
      int h[kMaxWorld][4];
      if (rk_x2_headers(r, nb, h)) return BH_ERR_HIP;
      more = 0;
      most = 0;
      for (int q = 0; q < P; q++) {
        if (h[q][0] - h[q][2] > 0) more = 1;
        if (h[q][0] > most) most = h[q][0];
      }
    }
    if (first < 0) first = most;
    if (!more) break;
    r->mig_rounds++;  // rare: a boundary moved a long way
    const long long want = round_up(most, 256);
    limit = (int)(want > limit ? want : limit);
    if (limit > r->plan.mig_cap) limit = r->plan.mig_cap;
  }
  r->mig_last = first;
  if (r->log && r->ctx && !failed) {  // tests / tools: synchronises
    int32_t info[8];
    if (bh_dd_get_info(r->ctx, info) == BH_OK) {
      r->log->push_back(first);
      r->log->push_back(info[3]);
    }
  }
  {  // next step's slots: one and a half times what this step moved (boundaries that persist move a fraction of a
     // per cent of a rank per step; a rebalance moves more and goes in several rounds)
    long long ms = round_up((long long)(first * 1.5 + 512), 256);
    if (ms > r->plan.mig_cap) ms = r->plan.mig_cap;
    if (ms < 1024) ms = 1024;
    r->mig_stride = (int)ms;
  }
  rk_mark(r, 2);
  if (failed && rk_zero(r, r->buf[X3S], (size_t)sz.x3_bytes)) return BH_ERR_HIP;
  if (c.all_gather(c.user, r->buf[X3R], r->buf[X3S], sz.x3_bytes, st)) return BH_ERR_COMM;  // X3: piece descriptors
  rk_mark(r, 3);
  int tries = 0, need = 0;
  for (;;) {
    const int stride = r->stride;
    char* seg = r->buf[POOL] + (size_t)sz.seg_base * 32;
    const int nseg = r->o.let_mode == 1 ? P : 1;  // segments this rank sends
    if (!failed)  // own pieces on the side stream (first try only: it overlaps X4), LET marked / exported
      failed = e.phase_let(e.user, r->buf[X3R], r->buf[X4S], stride, (r->split_now && tries == 0) ? 1 : 0);
    if (failed) {
      if (rk_mark_left(r, nseg, stride)) return BH_ERR_HIP;
    }
    tries++;
    const bool timed = r->adaptive && r->device_mem && tries == 1;
    if (timed && hipEventRecord(r->ev_xa, r->stream) != hipSuccess) return BH_ERR_HIP;
    // X4: LET records, in place.  First try of a step over a transport that can: a size per pair, from what every
    // pair needed in the last fitting exchange (bh_dd_x4_sizes: two thirds of the slots are padding, tools/dd_needs.py);
    // a pair that outgrew its size shows as "does not fit" and the repeat moves whole slots.
    int xs;
    int64_t sb[kMaxWorld], rb[kMaxWorld];
    bool sized = false;
    if (r->ctx && r->device_mem && r->o.let_mode == 1) {
      if (bh_dd_x4_sizes(r->ctx, stride, (tries == 1 && c.all_to_all_v) ? 1 : 0, sb, rb)) return BH_ERR_BAD_ARG;
      sized = true;
    }
    if (sized && c.all_to_all_v && tries == 1) {
      xs = c.all_to_all_v(c.user, seg, r->buf[X4S], (int64_t)stride * 32, sb, rb, st);
      r->x4_recv_bytes = 0;
      for (int q = 0; q < P; q++) r->x4_recv_bytes += rb[q];
    } else {
      xs = r->o.let_mode == 1 ? c.all_to_all(c.user, seg, r->buf[X4S], (int64_t)stride * 32, st)
                              : c.all_gather(c.user, seg, r->buf[X4S], (int64_t)stride * 32, st);
      r->x4_recv_bytes = (long long)P * stride * 32;
    }
    if (xs) return BH_ERR_COMM;
#ifdef BH_STUDY
    {  // an exchange that lasts longer, for measurements at world size 1 (tools/r5_fake_x4.sh)
      static const int fake_us = getenv("BH_DD_FAKE_X4_US") ? atoi(getenv("BH_DD_FAKE_X4_US")) : 0;
      if (fake_us > 0 && r->ctx && bh_dd_idle_wave(r->ctx, fake_us)) return BH_ERR_HIP;
    }
#endif
    if (timed) {
      if (hipEventRecord(r->ev_xb, r->stream) != hipSuccess) return BH_ERR_HIP;
      r->x4_timed = true;
    }
    if (failed) {
      r->left_rank = c.rank;
      r->left_status = failed;
      return BH_ERR_DOMAIN_LEFT;
    }
    rk_mark(r, 4);
    int fits = 0;
    const int s = e.phase_force(e.user, r->buf[X3R], stride, r->let_counts, &fits);  // top tree, force pass, X4 sizes
    if (s) return s;  // rank-local after the last exchange of the step: the caller ends the job (bh_group aborts the
                      // transport; a multi-process job ends through its collective time-out)
    need = 0;
    for (int q = 0; q < P; q++) {
      if (r->let_counts[q] < 0) {
        r->left_rank = q;
        r->left_status = 0;
        return BH_ERR_DOMAIN_LEFT;
      }
      if (r->let_counts[q] > need) need = r->let_counts[q];
    }
    if (fits) {
      r->stride_last = stride;
      break;
    }
    if (need > r->plan.let_cap) {  // every rank sees the same counts
      r->left_rank = -1;
      r->left_status = 0;
      return BH_ERR_DOMAIN_LEFT;
    }
    r->let_retries++;
    long long ns = round_up((long long)(need * 1.25), 256);
    r->stride = (int)(ns < r->plan.let_cap ? ns : r->plan.let_cap);
  }
  {  // every rank sees the same counts, so every rank picks the same next stride
    long long ns = round_up((long long)(need * 1.15 + 1024), 256);
    if (ns > r->plan.let_cap) ns = r->plan.let_cap;
    if (ns < sz.let_min) ns = sz.let_min;
    r->stride = (int)ns;
  }
  rk_mark(r, 5);
  const int s = e.phase_end(e.user, r->buf[X1S]);  // integrate + the next step's X1 payload
  if (s) return s;
  r->x1_ready = true;
  rk_mark(r, 6);
  r->steps++;
  return BH_OK;
}

int rank_alloc(bh_rank** out) {
  bh_rank* r = (bh_rank*)calloc(1, sizeof(bh_rank));
  if (!r) return BH_ERR_OOM;
  r->prof_ev = new (std::nothrow) std::vector<hipEvent_t>();
  if (!r->prof_ev) {
    free(r);
    return BH_ERR_OOM;
  }
  r->left_rank = -1;
  *out = r;
  return BH_OK;
}

bool comm_ok(const bh_comm* c) {
  return c && c->world >= 1 && c->world <= kMaxWorld && c->rank >= 0 && c->rank < c->world && c->all_gather &&
         c->all_to_all;
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------ transports
int bh_comm_rccl_from(bh_comm* out, void* nccl_comm, int world, int rank) {
  if (!out || !nccl_comm || world < 1 || rank < 0 || rank >= world) return BH_ERR_BAD_ARG;
  if (!rccl()) return BH_ERR_COMM;
  return rccl_fill(out, (nccl_comm_t)nccl_comm, false, world, rank);
}

int bh_comm_rccl_unique_id(void* id128) {
  if (!id128) return BH_ERR_BAD_ARG;
  rccl_api* a = rccl();
  if (!a) return BH_ERR_COMM;
  nccl_uid id;
  if (a->GetUniqueId(&id)) return BH_ERR_COMM;
  memcpy(id128, &id, sizeof(id));
  return BH_OK;
}

int bh_comm_rccl_init_rank(bh_comm* out, const void* id128, int world, int rank, int device) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return BH_ERR_BAD_ARG;
  rccl_api* a = rccl();
  if (!a) return BH_ERR_COMM;
  if (hipSetDevice(device) != hipSuccess) return BH_ERR_NO_DEVICE;
  nccl_uid id;
  memcpy(&id, id128, sizeof(id));
  nccl_comm_t comm = nullptr;
  if (a->CommInitRank(&comm, world, id, rank) || !comm) return BH_ERR_COMM;
  const int s = rccl_fill(out, comm, true, world, rank);
  if (s) (void)a->CommDestroy(comm);
  return s;
}

// One all-gather and one all-to-all of known words through `comm` on the current device, checked on the host: a
// transport that delivers wrong chunks (or fails) is found before a step depends on it.  Collective: every rank
// of the comm calls it; every rank sees the same verdict only if the transport works, so callers that want to
// fall back together reduce the result over their own channel.
int bh_comm_check(const bh_comm* comm) {
  if (!comm || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world || !comm->all_gather || !comm->all_to_all)
    return BH_ERR_BAD_ARG;
  const int P = comm->world, r = comm->rank, W = 64;  // 64 words per chunk
  const size_t words = (size_t)P * W;
  std::vector<uint32_t> h(words), back(2 * words);
  uint32_t* d = nullptr;  // [send all-to-all: P chunks][recv all-gather: P chunks][recv all-to-all: P chunks]
  hipStream_t st = nullptr;
  if (hipMalloc(&d, 3 * words * sizeof(uint32_t)) != hipSuccess) return BH_ERR_OOM;
  int ret = BH_ERR_COMM;
  do {
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
    for (int q = 0; q < P; q++)  // chunk q of my send: (me, q, word)
      for (int w = 0; w < W; w++) h[(size_t)q * W + w] = 0x5a000000u | (uint32_t)r << 16 | (uint32_t)q << 8 | (uint32_t)w;
    if (hipMemcpyAsync(d, h.data(), words * 4, hipMemcpyHostToDevice, st) != hipSuccess) break;
    if (hipMemsetAsync(d + words, 0, 2 * words * 4, st) != hipSuccess) break;
    // all-gather: everybody's chunk 0
    if (comm->all_gather(comm->user, d + words, d, (int64_t)W * 4, st)) break;
    if (comm->all_to_all(comm->user, d + 2 * words, d, (int64_t)W * 4, st)) break;
    if (hipMemcpyAsync(back.data(), d + words, 2 * words * 4, hipMemcpyDeviceToHost, st) != hipSuccess) break;
    if (hipStreamSynchronize(st) != hipSuccess) break;
    bool ok = true;
    for (int q = 0; q < P && ok; q++)
      for (int w = 0; w < W; w++) {
        const uint32_t ag = 0x5a000000u | (uint32_t)q << 16 | 0u << 8 | (uint32_t)w;           // rank q's chunk 0
        const uint32_t aa = 0x5a000000u | (uint32_t)q << 16 | (uint32_t)r << 8 | (uint32_t)w;  // rank q's chunk `me`
        if (back[(size_t)q * W + w] != ag || back[words + (size_t)q * W + w] != aa) {
          ok = false;
          break;
        }
      }
    // the exchange with a size per pair, if the transport has one: rank q gets the first (q + me) % W + 1 words of my
    // chunk q, the rest of the slot stays as it was (zero)
    if (ok && comm->all_to_all_v) {
      int64_t sb[kMaxWorld], rbv[kMaxWorld];
      for (int q = 0; q < P; q++) sb[q] = rbv[q] = 4 * (int64_t)((q + r) % W + 1);
      if (hipMemsetAsync(d + 2 * words, 0, words * 4, st) != hipSuccess) break;
      if (comm->all_to_all_v(comm->user, d + 2 * words, d, (int64_t)W * 4, sb, rbv, st)) break;
      if (hipMemcpyAsync(back.data(), d + 2 * words, words * 4, hipMemcpyDeviceToHost, st) != hipSuccess) break;
      if (hipStreamSynchronize(st) != hipSuccess) break;
      for (int q = 0; q < P && ok; q++)
        for (int w = 0; w < W; w++) {
          const uint32_t aa = w < (q + r) % W + 1 ? (0x5a000000u | (uint32_t)q << 16 | (uint32_t)r << 8 | (uint32_t)w) : 0u;
          if (back[(size_t)q * W + w] != aa) {
            ok = false;
            break;
          }
        }
    }
    ret = ok ? BH_OK : BH_ERR_COMM;
  } while (false);
  if (st) (void)hipStreamDestroy(st);
  (void)hipFree(d);
  return ret;
}

int bh_hub_create(bh_hub** out, int world) {
  if (!out || world < 1 || world > kMaxWorld) return BH_ERR_BAD_ARG;
  bh_hub* h = new (std::nothrow) bh_hub();
  if (!h) return BH_ERR_OOM;
  h->world = world;
  for (int q = 0; q < kMaxWorld; q++) {
    h->src[q] = nullptr;
    h->stream[q] = nullptr;
    h->have_ev[q] = false;
  }
  *out = h;
  return BH_OK;
}

int bh_comm_hub(bh_comm* out, bh_hub* hub, int rank) {
  if (!out || !hub || rank < 0 || rank >= hub->world) return BH_ERR_BAD_ARG;
  hub_user* u = (hub_user*)calloc(1, sizeof(hub_user));
  if (!u) return BH_ERR_OOM;
  u->hub = hub;
  u->rank = rank;
  out->world = hub->world;
  out->rank = rank;
  out->user = u;
  out->all_gather = hub_all_gather;
  out->all_to_all = hub_all_to_all;
  out->release = hub_release;
  out->all_to_all_v = hub_all_to_all_v;
  return BH_OK;
}

void bh_hub_abort(bh_hub* h) {
  if (!h) return;
  std::lock_guard<std::mutex> lk(h->mu);
  h->aborted = true;
  h->cv.notify_all();
}

void bh_hub_destroy(bh_hub* h) {
  if (!h) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (int q = 0; q < h->world; q++)
    if (h->have_ev[q]) {
      (void)hipSetDevice(h->ev_dev[q]);
      (void)hipEventDestroy(h->ev_post[q]);
      (void)hipEventDestroy(h->ev_done[q]);
    }
  (void)hipSetDevice(cur);
  delete h;
}

// ------------------------------------------------------------------ rank
int bh_rank_default_opts(bh_rank_opts* o) {
  if (!o) return BH_ERR_BAD_ARG;
  memset(o, 0, sizeof(*o));
  o->let_mode = 1;
  o->split = -1;
  return BH_OK;
}

int bh_rank_query(int64_t n_total, int world, const bh_rank_opts* o_in, bh_rank_plan* out) {
  if (!out || n_total < 1 || world < 1) return BH_ERR_BAD_ARG;
  bh_rank_opts o;
  if (o_in)
    o = *o_in;
  else
    bh_rank_default_opts(&o);
  if (o.let_mode != 0 && o.let_mode != 1) return BH_ERR_BAD_ARG;
  const long long fair = (n_total + world - 1) / world;
  long long n_cap = o.n_cap > 0 ? o.n_cap : (long long)(fair * 1.3) + 4096;
  if (n_cap < 1024) n_cap = 1024;
  long long mig_cap = o.mig_cap;
  if (mig_cap <= 0) {
    mig_cap = n_cap / 2 > 4096 ? n_cap / 2 : 4096;
    if (mig_cap > 4 * n_cap / world) mig_cap = 4 * n_cap / world;
  }
  long long let_cap = o.let_cap > 0 ? o.let_cap : kLetMin + n_cap;
  let_cap += let_cap & 1;  // segments hold whole 64-byte digest pairs
  if (n_cap > 0x7fffffffLL || let_cap > 0x7fffffffLL) return BH_ERR_BAD_ARG;
  memset(out, 0, sizeof(*out));
  const int s = bh_dd_query((int)n_cap, world, (int)mig_cap, (int)let_cap, &out->sz);
  if (s) return s;
  out->n_cap = (int)n_cap;
  out->mig_cap = (int)mig_cap;
  out->let_cap = (int)let_cap;
  long long s0 = round_up(kLetMin + n_cap / 8, 256);
  out->stride0 = (int)(s0 < let_cap ? s0 : let_cap);
  const bh_dd_sizes& z = out->sz;
  out->bytes[X1S] = z.x1_bytes;
  out->bytes[X1R] = z.x1_bytes * world;
  out->bytes[X2S] = z.x2_bytes;
  out->bytes[X2R] = z.x2_bytes * world;
  out->bytes[X3S] = z.x3_bytes;
  out->bytes[X3R] = z.x3_bytes * world;
  out->bytes[X4S] = (int64_t)(o.let_mode == 1 ? world : 1) * let_cap * 32;
  out->bytes[POOL] = z.pool_records * 32;
  return BH_OK;
}

static void rank_common_init(bh_rank* r, const bh_comm* comm, const bh_rank_plan* plan, const bh_rank_opts* o) {
  r->comm = *comm;
  r->plan = *plan;
  r->o = *o;
  // one pass unless told otherwise: measured (rank_adapt) the split pays only for an exchange of ~0.18 ms and more
  if (r->o.split < 0) r->o.split = 0;
  // (the adaptive form only where two passes can pay at all: ranks whose launches fill the GPU, more than one rank)
  r->adaptive = r->o.split == 2 && comm->world > 1 && plan->n_cap >= 400000;
  r->split_now = r->o.split == 1 ? 1 : 0;
  r->stride = plan->stride0;
  r->mig_stride = plan->mig_cap < 4096 ? plan->mig_cap : 4096;
  if (o->log) r->log = new (std::nothrow) std::vector<int32_t>();
}

int bh_rank_create(bh_rank** out, const bh_comm* comm, int64_t n_total, const bh_params* p, const bh_rank_opts* o_in,
                   int device, void* hip_stream, const bh_rank_buffers* bufs) {
  if (!out || !comm_ok(comm)) return BH_ERR_BAD_ARG;
  *out = nullptr;
  bh_rank_opts o;
  if (o_in)
    o = *o_in;
  else
    bh_rank_default_opts(&o);
  bh_rank_plan plan;
  int s = bh_rank_query(n_total, comm->world, &o, &plan);
  if (s) return s;
  if (hipSetDevice(device) != hipSuccess) return BH_ERR_NO_DEVICE;
  bh_rank* r = nullptr;
  s = rank_alloc(&r);
  if (s) return s;
  rank_common_init(r, comm, &plan, &o);
  r->device_mem = true;
  r->device = device;
  r->stream = (hipStream_t)hip_stream;
  if (!r->stream) {
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) {
      r->comm.release = nullptr;  // the caller still owns the comm on failure
      bh_rank_destroy(r);
      return BH_ERR_HIP;
    }
    r->own_stream = true;
  }
  if (bufs) {
    void* const given[8] = {bufs->x1s, bufs->x1r, bufs->x2s, bufs->x2r, bufs->x3s, bufs->x3r, bufs->x4s, bufs->pool};
    for (int k = 0; k < 8; k++) {
      if (!given[k]) s = BH_ERR_BAD_ARG;
      r->buf[k] = (char*)given[k];
    }
  } else {
    r->own_buf = true;
    for (int k = 0; k < 8 && !s; k++) {
      if (hipMalloc((void**)&r->buf[k], (size_t)plan.bytes[k]) != hipSuccess) {
        r->buf[k] = nullptr;
        s = BH_ERR_OOM;
      } else if (k != POOL && hipMemsetAsync(r->buf[k], 0, (size_t)plan.bytes[k], r->stream) != hipSuccess) {
        s = BH_ERR_HIP;  // (bh_dd_init clears the pool)
      }
    }
  }
  if (!s) s = bh_create_on_stream(&r->ctx, plan.n_cap, p, device, (void*)r->stream);
  if (!s) s = bh_dd_init(r->ctx, comm->world, comm->rank, n_total, plan.mig_cap, plan.let_cap, r->buf[POOL],
                         plan.sz.pool_records);
  if (!s) s = bh_dd_set_let_mode(r->ctx, o.let_mode);
  if (!s && o.serial) s = bh_dd_set_serial(r->ctx, 1);
  if (!s && r->adaptive && (hipEventCreate(&r->ev_xa) != hipSuccess || hipEventCreate(&r->ev_xb) != hipSuccess)) s = BH_ERR_HIP;
  // two-pass steps (split 1) split the walk of the first 30 % of a rank's bodies unless told otherwise: their own pass —
  // launched behind the LET export, beside X4 — lasts ~0.3 ms at 1M bodies per rank and covers an X4 of ~0.2 ms
  if (!s) s = bh_dd_set_split_percent(r->ctx, o.split_pct <= 0 ? 30 : (o.split_pct > 100 ? 100 : o.split_pct));
  if (s) {
    r->comm.release = nullptr;
    bh_rank_destroy(r);
    return s;
  }
  r->ops = rank_ops{r->ctx,        op_cube_pack,   op_phase_migrate, op_migrate_pack,
                    op_phase_tree, op_phase_let,   op_phase_force,   op_phase_end};
  *out = r;
  return BH_OK;
}

int bh_rank_create_scripted(bh_rank** out, const bh_comm* comm, const bh_rank_script* sc, const bh_rank_plan* plan,
                            const bh_rank_opts* o_in) {
  if (!out || !comm_ok(comm) || !sc || !plan || !sc->cube_pack || !sc->phase_migrate || !sc->migrate_pack ||
      !sc->phase_tree || !sc->phase_let || !sc->phase_force || !sc->phase_end)
    return BH_ERR_BAD_ARG;
  *out = nullptr;
  bh_rank_opts o;
  if (o_in)
    o = *o_in;
  else
    bh_rank_default_opts(&o);
  bh_rank* r = nullptr;
  const int s = rank_alloc(&r);
  if (s) return s;
  rank_common_init(r, comm, plan, &o);
  r->device_mem = false;
  r->own_buf = true;
  for (int k = 0; k < 8; k++) {
    r->buf[k] = (char*)calloc(1, (size_t)plan->bytes[k] > 0 ? (size_t)plan->bytes[k] : 1);
    if (!r->buf[k]) {
      r->comm.release = nullptr;
      bh_rank_destroy(r);
      return BH_ERR_OOM;
    }
  }
  r->ops = rank_ops{sc->user,       sc->cube_pack, sc->phase_migrate, sc->migrate_pack,
                    sc->phase_tree, sc->phase_let, sc->phase_force,   sc->phase_end};
  *out = r;
  return BH_OK;
}

int bh_rank_upload(bh_rank* r, int n_loc, const float* x, const float* y, const float* z, const float* vx,
                   const float* vy, const float* vz, const float* m, const int32_t* ids) {
  if (!r || !r->ctx) return BH_ERR_BAD_ARG;
  const int s = bh_dd_upload(r->ctx, n_loc, x, y, z, vx, vy, vz, m, ids);
  if (s) return s;
  r->n_loc = n_loc;
  r->x1_ready = false;  // a payload packed from the previous bodies is void
  return BH_OK;
}

int bh_rank_step(bh_rank* r, int steps) {
  if (!r || steps < 0) return BH_ERR_BAD_ARG;
  if (r->device_mem && hipSetDevice(r->device) != hipSuccess) return BH_ERR_NO_DEVICE;
  for (int k = 0; k < steps; k++) {
    const int s = rank_one_step(r);
    if (s) return s;
  }
  return BH_OK;
}

int bh_rank_get_info(bh_rank* r, bh_rank_info* o) {
  if (!r || !o) return BH_ERR_BAD_ARG;
  memset(o, 0, sizeof(*o));
  o->n_loc = r->n_loc;
  o->stride = r->stride;
  o->mig_stride = r->mig_stride;
  o->mig_last = r->mig_last;
  o->mig_rounds = r->mig_rounds;
  o->let_retries = r->let_retries;
  o->left_rank = r->left_rank;
  o->left_status = r->left_status;
  o->steps = r->steps;
  o->split_now = r->split_now;
  o->x4_us = r->x4_samples > 0 ? (int32_t)(r->x4_ema_ms * 1000.0f + 0.5f) : -1;
  o->x4_recv_kb = (int32_t)(r->x4_recv_bytes >> 10);
  memcpy(o->let_counts, r->let_counts, sizeof(o->let_counts));
  return BH_OK;
}

bh_ctx* bh_rank_ctx(bh_rank* r) { return r ? r->ctx : nullptr; }

int bh_rank_buffers_of(bh_rank* r, bh_rank_buffers* out, bh_rank_plan* plan) {
  if (!r) return BH_ERR_BAD_ARG;
  if (out) {
    out->x1s = r->buf[X1S]; out->x1r = r->buf[X1R]; out->x2s = r->buf[X2S]; out->x2r = r->buf[X2R];
    out->x3s = r->buf[X3S]; out->x3r = r->buf[X3R]; out->x4s = r->buf[X4S]; out->pool = r->buf[POOL];
  }
  if (plan) *plan = r->plan;
  return BH_OK;
}

int bh_rank_set_profile(bh_rank* r, int on) {
  if (!r) return BH_ERR_BAD_ARG;
  if (on) rk_prof_clear(r);
  r->prof = on != 0;
  return BH_OK;
}

int bh_rank_phase_ms(bh_rank* r, double mean_ms[BH_RANK_PHASES], int* steps) {
  if (!r || !mean_ms) return BH_ERR_BAD_ARG;
  for (int k = 0; k < BH_RANK_PHASES; k++) mean_ms[k] = 0.0;
  if (steps) *steps = 0;
  if (!r->device_mem) return BH_OK;
  if (hipSetDevice(r->device) != hipSuccess) return BH_ERR_NO_DEVICE;
  if (hipStreamSynchronize(r->stream) != hipSuccess) return BH_ERR_HIP;
  int cnt = 0;
  const size_t ns = r->prof_ev->size() / 7;
  for (size_t i = 0; i < ns; i++) {
    hipEvent_t* ev = r->prof_ev->data() + 7 * i;
    bool whole = true;
    for (int k = 0; k < 7; k++) whole = whole && ev[k];
    if (!whole) continue;  // a step that left early
    float ms[BH_RANK_PHASES];
    bool ok = true;
    for (int k = 0; k < BH_RANK_PHASES; k++) ok = ok && hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]) == hipSuccess;
    if (!ok) continue;
    for (int k = 0; k < BH_RANK_PHASES; k++) mean_ms[k] += ms[k];
    cnt++;
  }
  for (int k = 0; k < BH_RANK_PHASES && cnt; k++) mean_ms[k] /= cnt;
  if (steps) *steps = cnt;
  return BH_OK;
}

int bh_rank_read_log(bh_rank* r, int32_t* pairs, int capacity_pairs, int* n_pairs) {
  if (!r || !n_pairs) return BH_ERR_BAD_ARG;
  const int have = r->log ? (int)(r->log->size() / 2) : 0;
  *n_pairs = have;
  if (!pairs) return BH_OK;
  if (capacity_pairs < have) return BH_ERR_SMALL_BUFFER;
  if (have) memcpy(pairs, r->log->data(), (size_t)have * 8);
  return BH_OK;
}

// Measurement only: see include/bh.h.  The rank's buffers still hold the last step's gathered descriptors, the pool its
// tree and the imported segments; the exchange itself is not repeated.
int bh_rank_replay_force_phase(bh_rank* r, int split, int split_pct, int x4_us, int reps, float* ms) {
  if (!r || !ms || reps < 1 || x4_us < 0 || (split && (split_pct < 1 || split_pct > 100))) return BH_ERR_BAD_ARG;
  if (!r->ctx || !r->device_mem || r->steps < 1 || r->stride_last <= 0) return BH_ERR_ORDER;
  if (hipSetDevice(r->device) != hipSuccess) return BH_ERR_NO_DEVICE;
  int saved[4];
  int s = bh_dd_replay_begin(r->ctx, split, split_pct, saved);
  if (s) return s;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) s = BH_ERR_HIP;
  double sum = 0.0;
  int32_t counts[kMaxWorld];
  for (int rep = -1; rep < reps && !s; rep++) {  // one untimed round first
    if (hipEventRecord(e0, r->stream) != hipSuccess) s = BH_ERR_HIP;
    if (!s) s = bh_dd_phase_let(r->ctx, r->buf[X3R], r->buf[X4S], r->stride_last, split ? 1 : 0);
    if (!s) s = bh_dd_idle_wave(r->ctx, x4_us);
    int fits = 0;
    if (!s) s = bh_dd_phase_force(r->ctx, r->buf[X3R], r->stride_last, counts, &fits);
    if (!s && !fits) s = BH_ERR_SMALL_BUFFER;
    if (!s && (hipEventRecord(e1, r->stream) != hipSuccess || hipStreamSynchronize(r->stream) != hipSuccess)) s = BH_ERR_HIP;
    float t = 0.0f;
    if (!s && hipEventElapsedTime(&t, e0, e1) != hipSuccess) s = BH_ERR_HIP;
    if (rep >= 0) sum += t;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  const int s2 = bh_dd_replay_end(r->ctx, saved);
  *ms = (float)(sum / reps);
  return s ? s : s2;
}

void bh_rank_destroy(bh_rank* r) {
  if (!r) return;
  if (r->device_mem) (void)hipSetDevice(r->device);
  if (r->ctx) {
    (void)bh_sync(r->ctx);
    bh_destroy(r->ctx);
  }
  if (r->prof_ev) {
    rk_prof_clear(r);
    delete r->prof_ev;
  }
  if (r->ev_xa) (void)hipEventDestroy(r->ev_xa);
  if (r->ev_xb) (void)hipEventDestroy(r->ev_xb);
  delete r->log;
  if (r->own_buf)
    for (int k = 0; k < 8; k++)
      if (r->buf[k]) {
        if (r->device_mem)
          (void)hipFree(r->buf[k]);
        else
          free(r->buf[k]);
      }
  if (r->own_stream && r->stream) (void)hipStreamDestroy(r->stream);
  if (r->comm.release) r->comm.release(r->comm.user);
  free(r);
}

}  // extern "C"

// ------------------------------------------------------------------ group: the ranks of one process
namespace {

struct worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = false, quit = false;
  int result = 0;
};

void worker_main(worker* w, int device) {
  (void)hipSetDevice(device);
  std::unique_lock<std::mutex> lk(w->mu);
  for (;;) {
    w->cv.wait(lk, [&] { return w->has_job || w->quit; });
    if (w->quit) return;
    std::function<int()> job = std::move(w->job);
    w->has_job = false;
    lk.unlock();
    const int res = job();
    lk.lock();
    w->result = res;
    w->done = true;
    w->cv.notify_all();
  }
}

}  // namespace

struct bh_group {
  int P;
  std::vector<int> dev;
  int64_t n_total;
  bh_params p;
  bh_rank_opts o;
  bh_hub* hub = nullptr;
  std::vector<nccl_comm_t> nccl;  // ncclCommInitAll's communicators (owned by the group)
  std::vector<bh_rank*> ranks;
  std::vector<worker*> w;
  std::vector<hipStream_t> stream;  // ranks that share a device share ONE stream (owned here): their kernels run one
                                    // after the other, as on a GPU of their own — what a one-GPU rehearsal measures
  bool dead = false;  // a rank failed on its own: the transport was aborted

  // fn(rank) on every rank's own thread; -> first non-zero result in rank order, BH_ERR_DOMAIN_LEFT last
  int run_all(const std::function<int(int)>& fn) {
    for (int q = 0; q < P; q++) {
      worker* k = w[q];
      std::lock_guard<std::mutex> lk(k->mu);
      k->job = [fn, q] { return fn(q); };
      k->has_job = true;
      k->done = false;
      k->cv.notify_all();
    }
    int first = BH_OK;
    bool left = false;
    for (int q = 0; q < P; q++) {
      worker* k = w[q];
      std::unique_lock<std::mutex> lk(k->mu);
      k->cv.wait(lk, [&] { return k->done; });
      if (k->result == BH_ERR_DOMAIN_LEFT)
        left = true;
      else if (k->result && !first)
        first = k->result;
    }
    return first ? first : (left ? BH_ERR_DOMAIN_LEFT : BH_OK);
  }

  void abort_transport() {
    dead = true;
    if (hub) bh_hub_abort(hub);
    rccl_api* a = rccl();
    if (a && a->CommAbort)
      for (nccl_comm_t c : nccl)
        if (c) (void)a->CommAbort(c);
    nccl.clear();
  }
};

extern "C" {

void bh_destroy_group(bh_group* g) {
  if (!g) return;
  if (!g->w.empty() && !g->ranks.empty())
    g->run_all([g](int q) {
      if (g->ranks[q]) bh_rank_destroy(g->ranks[q]);
      g->ranks[q] = nullptr;
      return 0;
    });
  for (worker* k : g->w) {
    {
      std::lock_guard<std::mutex> lk(k->mu);
      k->quit = true;
      k->cv.notify_all();
    }
    if (k->th.joinable()) k->th.join();
    delete k;
  }
  rccl_api* a = rccl();
  if (a)
    for (nccl_comm_t c : g->nccl)
      if (c) (void)a->CommDestroy(c);
  if (g->hub) bh_hub_destroy(g->hub);
  for (size_t q = 0; q < g->stream.size(); q++) {
    bool first = g->stream[q] != nullptr;
    for (size_t k = 0; k < q; k++) first = first && g->stream[k] != g->stream[q];
    if (first && hipSetDevice(g->dev[q]) == hipSuccess) (void)hipStreamDestroy(g->stream[q]);
  }
  delete g;
}

int bh_create_group(bh_group** out, int ngpus, const int* devices, int64_t n_total, const bh_params* p,
                    const bh_rank_opts* o, int transport) {
  if (!out || ngpus < 1 || ngpus > kMaxWorld || !devices || n_total < 1 || transport < 0 || transport > 2)
    return BH_ERR_BAD_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return BH_ERR_NO_DEVICE;
  bool distinct = true;
  for (int q = 0; q < ngpus; q++) {
    if (devices[q] < 0 || devices[q] >= ndev) return BH_ERR_NO_DEVICE;
    for (int k = 0; k < q; k++) distinct = distinct && devices[k] != devices[q];
  }
  const bool chosen_here = transport == 0;
  if (transport == 0) transport = distinct ? 1 : 2;
  if (transport == 1 && !distinct) return BH_ERR_BAD_ARG;  // RCCL refuses two ranks on one device
  bh_group* g = new (std::nothrow) bh_group();
  if (!g) return BH_ERR_OOM;
  g->P = ngpus;
  g->dev.assign(devices, devices + ngpus);
  g->n_total = n_total;
  if (p)
    g->p = *p;
  else
    bh_default_params(&g->p);
  if (o)
    g->o = *o;
  else
    bh_rank_default_opts(&g->o);
  g->ranks.assign(ngpus, nullptr);
  int s = BH_OK;
  auto make_hub = [&]() {
    const int s1 = bh_hub_create(&g->hub, ngpus);
    if (s1) return s1;
    if (distinct)  // copies between devices: peer access where the platform offers it
      for (int q = 0; q < ngpus; q++)
        for (int k = 0; k < ngpus; k++)
          if (k != q && hipSetDevice(devices[q]) == hipSuccess) (void)hipDeviceEnablePeerAccess(devices[k], 0);
    return (int)BH_OK;
  };
  if (transport == 1) {
    rccl_api* a = rccl();
    g->nccl.assign(ngpus, nullptr);
    if (!a || a->CommInitAll(g->nccl.data(), ngpus, devices)) {
      g->nccl.clear();
      if (!chosen_here) {  // the caller asked for RCCL
        delete g;
        return BH_ERR_COMM;
      }
      transport = 2;
    }
  }
  if (transport == 2 && (s = make_hub()) != BH_OK) {
    delete g;
    return s;
  }
  g->stream.assign(ngpus, nullptr);
  if (!distinct)
    for (int q = 0; q < ngpus; q++) {
      for (int k = 0; k < q && !g->stream[q]; k++)
        if (devices[k] == devices[q]) g->stream[q] = g->stream[k];
      if (!g->stream[q] && (hipSetDevice(devices[q]) != hipSuccess ||
                            hipStreamCreateWithFlags(&g->stream[q], hipStreamNonBlocking) != hipSuccess)) {
        g->stream[q] = nullptr;
        bh_destroy_group(g);
        return BH_ERR_HIP;
      }
    }
  for (int q = 0; q < ngpus; q++) {
    worker* k = new (std::nothrow) worker();
    if (!k) {
      bh_destroy_group(g);
      return BH_ERR_OOM;
    }
    g->w.push_back(k);
    k->th = std::thread(worker_main, k, devices[q]);
  }
  if (!distinct) g->o.serial = 1;  // ranks that share a GPU: one stream each, own pass included (bh_dd_set_serial)
  if (!g->hub) {  // RCCL: move known words through it once before any step depends on it
    s = g->run_all([g](int q) {
      bh_comm c;
      memset(&c, 0, sizeof(c));
      int s1 = rccl_fill(&c, g->nccl[q], false, g->P, q);
      if (s1) return s1;
      s1 = bh_comm_check(&c);
      c.release(c.user);
      return s1;
    });
    if (s && chosen_here) {  // nobody asked for RCCL by name: device copies between the ranks' threads instead
      rccl_api* a = rccl();
      for (nccl_comm_t& c : g->nccl) {
        if (c && a) (void)a->CommDestroy(c);
        c = nullptr;
      }
      g->nccl.clear();
      s = make_hub();
    }
    if (s) {
      bh_destroy_group(g);
      return s;
    }
  }
  s = g->run_all([g](int q) {
    bh_comm c;
    memset(&c, 0, sizeof(c));
    int s1 = g->hub ? bh_comm_hub(&c, g->hub, q) : rccl_fill(&c, g->nccl[q], false, g->P, q);
    if (s1) return s1;
    s1 = bh_rank_create(&g->ranks[q], &c, g->n_total, &g->p, &g->o, g->dev[q], (void*)g->stream[q], nullptr);
    if (s1 && c.release) c.release(c.user);
    return s1;
  });
  if (s) {
    bh_destroy_group(g);
    return s;
  }
  *out = g;
  return BH_OK;
}

int bh_group_upload(bh_group* g, const float* x, const float* y, const float* z, const float* vx, const float* vy,
                    const float* vz, const float* m) {
  if (!g || !x || !y || !z || !vx || !vy || !vz || !m) return BH_ERR_BAD_ARG;
  if (g->dead) return BH_ERR_COMM;
  const int64_t n = g->n_total;
  if (n > 0x7fffffffLL) return BH_ERR_BAD_ARG;
  // the bodies in the key order of the global cube: one throw-away full-size context on the first device (the slabs
  // a rank starts with decide only how much the first step migrates)
  std::vector<int32_t> order((size_t)n);
  {
    if (hipSetDevice(g->dev[0]) != hipSuccess) return BH_ERR_NO_DEVICE;
    bh_ctx* t = nullptr;
    int s = bh_create(&t, (int)n, &g->p, g->dev[0]);
    if (!s) s = bh_upload(t, x, y, z, vx, vy, vz, m);
    if (!s) s = bh_bbox(t);
    if (!s) s = bh_morton(t);
    if (!s) s = bh_sort(t);
    if (!s) s = bh_download_order(t, order.data());
    if (t) bh_destroy(t);
    if (s) return s;
  }
  const float* src[7] = {x, y, z, vx, vy, vz, m};
  return g->run_all([g, n, &order, &src](int q) {
    const int64_t lo = q * n / g->P, hi = (q + 1) * n / g->P;
    const size_t k = (size_t)(hi - lo);
    std::vector<float> a[7];
    for (int f = 0; f < 7; f++) {
      a[f].resize(k);
      for (size_t i = 0; i < k; i++) a[f][i] = src[f][order[(size_t)lo + i]];
    }
    return bh_rank_upload(g->ranks[q], (int)k, a[0].data(), a[1].data(), a[2].data(), a[3].data(), a[4].data(),
                          a[5].data(), a[6].data(), order.data() + lo);
  });
}

int bh_step_group(bh_group* g, int steps) {
  if (!g || steps < 0) return BH_ERR_BAD_ARG;
  if (g->dead) return BH_ERR_COMM;
  return g->run_all([g, steps](int q) {
    const int s = bh_rank_step(g->ranks[q], steps);
    if (s && s != BH_ERR_DOMAIN_LEFT) g->abort_transport();  // the other ranks may be waiting for this one
    return s;
  });
}

int bh_group_sync(bh_group* g) {
  if (!g) return BH_ERR_BAD_ARG;
  return g->run_all([g](int q) { return bh_sync(g->ranks[q]->ctx); });
}

static int group_download(bh_group* g, float* const dst[6], bool acc) {
  return g->run_all([g, dst, acc](int q) {
    bh_rank* r = g->ranks[q];
    const size_t k = (size_t)bh_n(r->ctx);
    std::vector<float> posm(4 * k), velid(4 * k), a(acc ? 4 * k : 0);
    const int s = bh_dd_download(r->ctx, acc ? nullptr : posm.data(), velid.data(), acc ? a.data() : nullptr);
    if (s) return s;
    for (size_t i = 0; i < k; i++) {
      int32_t id;
      memcpy(&id, &velid[4 * i + 3], 4);
      if (id < 0 || id >= g->n_total) return (int)BH_ERR_DEVICE_FLAG;
      if (acc) {
        for (int f = 0; f < 3; f++)
          if (dst[f]) dst[f][id] = a[4 * i + f];
      } else {
        for (int f = 0; f < 3; f++) {
          if (dst[f]) dst[f][id] = posm[4 * i + f];
          if (dst[3 + f]) dst[3 + f][id] = velid[4 * i + f];
        }
      }
    }
    return (int)BH_OK;
  });
}

int bh_group_download(bh_group* g, float* x, float* y, float* z, float* vx, float* vy, float* vz) {
  if (!g) return BH_ERR_BAD_ARG;
  float* const dst[6] = {x, y, z, vx, vy, vz};
  return group_download(g, dst, false);
}

int bh_group_download_acc(bh_group* g, float* ax, float* ay, float* az) {
  if (!g) return BH_ERR_BAD_ARG;
  float* const dst[6] = {ax, ay, az, nullptr, nullptr, nullptr};
  return group_download(g, dst, true);
}

int bh_group_size(const bh_group* g) { return g ? g->P : 0; }
bh_rank* bh_group_rank(bh_group* g, int rank) { return (g && rank >= 0 && rank < g->P) ? g->ranks[rank] : nullptr; }

}  // extern "C"
