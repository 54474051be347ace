// bh_bench — headless benchmark driver over the C-ABI (include/bh.h).
//
// Command-line counterpart of the reference's `main()` (nbody_v5_bench.cu:285-390): generate
// initial conditions, allocate, upload, run a frame loop timing every step, print the same
// `Frame | Trajanje (ms) | FPS` table (:351,:366), free.  With no arguments it reproduces the
// reference's run shape: disc IC, N = 500000 (:31), 1000 frames (:353).
//   bh_bench [--n N] [--steps K] [--warmup W] [--ic disc|msvc|plummer] [--seed S] [--theta T]
//            (--ic msvc: the disc exactly as the MSVC-built reference binary draws it, srand(seed) + rand())
//            [--leaf-cap C] [--strict] [--quiet] [--device D]
//            [--literal-force]   the root-monopole force the CUDA binary literally computes
//            [--dump FILE]       final state in the older generation's text format (output_bh.txt:1-4)
//            [--snapshot FILE]   lossless binary snapshot for restart
//            [--gpus G]          the whole node: G ranks, one per GPU, domain-decomposed step over RCCL
//                                (bh_create_group / bh_step_group; --n stays the TOTAL body count)
//            [--dist]            with --gpus 1: the multi-GPU step at world size 1 (RCCL path on one GPU)
//            [--devices a,b,..]  explicit device per rank; a device listed twice selects the in-process
//                                transport (one-GPU rehearsal of G ranks)
//            [--split | --one-pass | --adaptive]  the first part of a rank's bodies in two force passes (own pieces
//                                beside X4, then the remote pass), one pass after X4 (the default), or whichever the
//                                measured X4 makes cheaper (bh_rank_opts.split 2)
//            [--split-pct P]     per cent of a rank's bodies whose walk is split (default 30)
//            [--replay | --replay-rank Q]  after the run (Q: that rank only): every rank's force phase of the last step run again on its own
//                                streams with nothing else on the GPU, one pass / 20 / 30 / 100 % split, an idle
//                                wave of 0 / 125 / 250 us in the place of X4 (one-GPU rehearsals: what a rank-step
//                                spends between X3 and its end on a GPU of its own)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bh.h"

static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

#define CK(call)                                                         \
  do {                                                                   \
    int _s = (call);                                                     \
    if (_s != BH_OK) {                                                   \
      fprintf(stderr, "%s failed: %s (%d)\n", #call, bh_strerror(_s), _s); \
      return 1;                                                          \
    }                                                                    \
  } while (0)

// the frame loop of main() (ref:353-367) over bh_step_group: every rank steps, then the whole node is synchronised
static int run_group(int N, int frames, int warmup, bool quiet, const bh_params& p, const std::vector<int>& devs,
                     int split, int split_pct, const char* ic_name, std::vector<float>* a, const char* dump_path, const char* snap_path,
                     bool replay, int replay_rank) {
  bh_rank_opts o;
  bh_rank_default_opts(&o);
  o.split = split;
  o.split_pct = split_pct;
  bh_group* g = nullptr;
  CK(bh_create_group(&g, (int)devs.size(), devs.data(), N, &p, &o, 0));
  CK(bh_group_upload(g, a[0].data(), a[1].data(), a[2].data(), a[3].data(), a[4].data(), a[5].data(), a[6].data()));
  CK(bh_step_group(g, warmup));
  CK(bh_group_sync(g));
  printf("------------------------------------------\n");
  printf("\n%-10s | %-15s | %-10s\n", "Frame", "Trajanje (ms)", "FPS");  // ref:351
  double sum = 0.0;
  for (int frame = 0; frame < frames; frame++) {
    const double t0 = now_ms();
    CK(bh_step_group(g, 1));
    CK(bh_group_sync(g));  // bh_sync of every rank: BH_ERR_DEVICE_FLAG if a sticky device flag is set anywhere
    const double ms = now_ms() - t0;
    sum += ms;
    if (!quiet) printf("%-10d | %-15.3f | %-10.1f\n", frame, ms, 1000.0 / ms);  // ref:366
  }
  // the same frames without a host synchronisation per frame (what a caller that only needs the final state pays)
  const double t0 = now_ms();
  CK(bh_step_group(g, frames));
  CK(bh_group_sync(g));
  const double free_run = (now_ms() - t0) / (frames > 0 ? frames : 1);
  const double avg = sum / (frames > 0 ? frames : 1);
  double ph[BH_RANK_PHASES];
  int ph_steps = 0;
  CK(bh_rank_set_profile(bh_group_rank(g, 0), 1));
  CK(bh_step_group(g, 5));
  CK(bh_rank_phase_ms(bh_group_rank(g, 0), ph, &ph_steps));
  CK(bh_rank_set_profile(bh_group_rank(g, 0), 0));
  bh_rank_info info;
  CK(bh_rank_get_info(bh_group_rank(g, 0), &info));
  printf("------------------------------------------\n");
  printf("N=%d gpus=%d ic=%s theta=%.2f steps=%d avg %.3f ms/step  %.3e particles/s/step | %d steps without a sync per "
         "frame: %.3f ms/step  %.3e particles/s/step\n",
         N, (int)devs.size(), ic_name, p.theta, frames, avg, (double)N / (avg * 1e-3), frames, free_run,
         (double)N / (free_run * 1e-3));
  printf("rank 0, mean of %d steps (ms): x1 %.3f | cube+migration+local tree %.3f | x3 %.3f | LET export + x4 %.3f | "
         "top tree + force %.3f | integrate + next x1 %.3f || bodies %d LET stride %d emigrants %d extra migration "
         "rounds %d LET retries %d\n",
         ph_steps, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], info.n_loc, info.stride, info.mig_last, info.mig_rounds,
         info.let_retries);
  if (replay) {
    // Between X3 and the end of the force on a GPU of its own, rank by rank (the others idle): the last step's force
    // phase run again in every form, an idle wave in the place of X4 (bh_rank_replay_force_phase)
    CK(bh_group_sync(g));
    const int x4[3] = {0, 125, 250};
    const int forms[4][2] = {{0, 0}, {1, 20}, {1, 30}, {1, 100}};
    printf("\nforce phase of the last step, replayed per rank on its own streams, ms (LET kernels -> end of the last force "
           "launch; X4 = an idle wave)\nrank | form            | X4 0 us | 125 us | 250 us\n");
    double mean[4][3] = {};
    for (int q = 0; q < (int)devs.size(); q++)
      for (int f = 0; f < 4 && (replay_rank < 0 || q == replay_rank); f++) {
        float ms[3];
        for (int k = 0; k < 3; k++) CK(bh_rank_replay_force_phase(bh_group_rank(g, q), forms[f][0], forms[f][1], x4[k], 5, &ms[k]));
        char name[32];
        if (forms[f][0]) snprintf(name, sizeof(name), "first %3d %% split", forms[f][1]); else snprintf(name, sizeof(name), "one pass");
        printf("%4d | %-15s | %7.3f | %6.3f | %6.3f\n", q, name, ms[0], ms[1], ms[2]);
        for (int k = 0; k < 3; k++) mean[f][k] += ms[k] / (double)devs.size();
      }
    for (int f = 0; f < 4; f++) {
      char name[32];
      if (forms[f][0]) snprintf(name, sizeof(name), "first %3d %% split", forms[f][1]); else snprintf(name, sizeof(name), "one pass");
      printf("mean | %-15s | %7.3f | %6.3f | %6.3f\n", name, mean[f][0], mean[f][1], mean[f][2]);
    }
  }
  if (dump_path || snap_path) {
    CK(bh_group_download(g, a[0].data(), a[1].data(), a[2].data(), a[3].data(), a[4].data(), a[5].data()));
    if (dump_path)
      CK(bh_write_text(dump_path, N, warmup + 2 * frames + 5, p.theta, p.dt, a[0].data(), a[1].data(), a[2].data(),
                       a[3].data(), a[4].data(), a[5].data()));
    if (snap_path)
      CK(bh_write_snapshot(snap_path, N, warmup + 2 * frames + 5, &p, a[0].data(), a[1].data(), a[2].data(),
                           a[3].data(), a[4].data(), a[5].data(), a[6].data()));
  }
  bh_destroy_group(g);
  return 0;
}

int main(int argc, char** argv) {
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);  // RCCL between devices wants dmabuf IPC on this driver; set before any HIP call
  int N = 500000;  // ref:31
  int gpus = 0;
  bool dist = false;
  int split = -1, split_pct = 0;
  bool replay = false;
  int replay_rank = -1;
  std::vector<int> devs;
  int frames = 1000;  // ref:353
  int warmup = 0;
  int device = 0;
  bool plummer = false, msvc = false, quiet = false;
  const char* dump_path = nullptr;  // final state, older generation's text format (output_bh.txt:1-4)
  const char* snap_path = nullptr;  // lossless binary snapshot (checkpoint)
  unsigned long long seed = 42;  // ref:294 srand(42)
  bh_params p;
  bh_default_params(&p);
  for (int i = 1; i < argc; i++) {
    auto arg = [&](const char* name) { return !strcmp(argv[i], name) && i + 1 < argc; };
    if (arg("--n")) N = atoi(argv[++i]);
    else if (arg("--steps")) frames = atoi(argv[++i]);
    else if (arg("--warmup")) warmup = atoi(argv[++i]);
    else if (arg("--seed")) seed = strtoull(argv[++i], nullptr, 10);
    else if (arg("--theta")) p.theta = (float)atof(argv[++i]);
    else if (arg("--leaf-cap")) p.leaf_cap = atoi(argv[++i]);
    else if (arg("--device")) device = atoi(argv[++i]);
    else if (arg("--ic")) {
      ++i;
      plummer = !strcmp(argv[i], "plummer");
      msvc = !strcmp(argv[i], "msvc");
    }
    else if (arg("--gpus")) gpus = atoi(argv[++i]);
    else if (arg("--devices")) {
      for (char* t = strtok(argv[++i], ","); t; t = strtok(nullptr, ",")) devs.push_back(atoi(t));
    }
    else if (!strcmp(argv[i], "--dist")) dist = true;
    else if (!strcmp(argv[i], "--split")) split = 1;
    else if (!strcmp(argv[i], "--one-pass")) split = 0;
    else if (!strcmp(argv[i], "--adaptive")) split = 2;
    else if (!strcmp(argv[i], "--replay")) replay = true;
    else if (arg("--replay-rank")) { replay = true; replay_rank = atoi(argv[++i]); }
    else if (arg("--split-pct")) split_pct = atoi(argv[++i]);
    else if (arg("--dump")) dump_path = argv[++i];
    else if (arg("--snapshot")) snap_path = argv[++i];
    else if (!strcmp(argv[i], "--literal-force")) p.literal_force = 1;  // what the CUDA binary computes (D1)
    else if (!strcmp(argv[i], "--strict")) p.strict_fp = 1;
    else if (!strcmp(argv[i], "--quiet")) quiet = true;
    else {
      fprintf(stderr, "unknown argument %s\n", argv[i]);
      return 2;
    }
  }
  printf("Pokretanje Benchmarka za N = %d...\n", N);  // ref:287

  std::vector<float> x(N), y(N), z(N), vx(N), vy(N), vz(N), m(N);
  if (plummer)
    CK(bh_ic_plummer(N, seed, 400.0f, p.G, x.data(), y.data(), z.data(), vx.data(), vy.data(), vz.data(), m.data()));
  else if (msvc)  // ref:294-308 with the Microsoft C runtime's rand(): the binary's own initial conditions
    CK(bh_ic_disc_msvc(N, (uint32_t)seed, p.G, x.data(), y.data(), z.data(), vx.data(), vy.data(), vz.data(), m.data()));
  else
    CK(bh_ic_disc(N, seed, p.G, x.data(), y.data(), z.data(), vx.data(), vy.data(), vz.data(), m.data()));

  if (!devs.empty() || gpus > 1 || dist) {
    if (devs.empty())
      for (int q = 0; q < (gpus > 0 ? gpus : 1); q++) devs.push_back(device + q);
    std::vector<float> a[7] = {x, y, z, vx, vy, vz, m};
    return run_group(N, frames, warmup, quiet, p, devs, split, split_pct,
                     plummer ? "plummer" : (msvc ? "disc(msvc rand)" : "disc"), a, dump_path, snap_path, replay, replay_rank);
  }

  bh_ctx* c = nullptr;
  CK(bh_create(&c, N, &p, device));
  CK(bh_upload(c, x.data(), y.data(), z.data(), vx.data(), vy.data(), vz.data(), m.data()));
  CK(bh_set_timing(c, 3));  // frames: only the event pair around the force launch, on every 4th step (an event
                            // after every stage costs the stream ~40 us per step); the stage line below comes
                            // from one extra step
  for (int w = 0; w < warmup; w++) CK(bh_step(c));
  CK(bh_sync(c));

  printf("------------------------------------------\n");
  printf("\n%-10s | %-15s | %-10s\n", "Frame", "Trajanje (ms)", "FPS");  // ref:351

  double sum = 0.0, sum_force = 0.0;
  int force_samples = 0;
  bh_stats st;
  for (int frame = 0; frame < frames; frame++) {
    const double t0 = now_ms();
    CK(bh_step(c));
    CK(bh_sync(c));  // also the step loop's error check: BH_ERR_DEVICE_FLAG if a sticky device flag is set
    const double ms = now_ms() - t0;
    CK(bh_get_stats(c, &st));
    sum += ms;
    if (((st.steps - 1) & 3) == 0) {  // the step that just ran was a sampled one
      sum_force += st.ms_force;
      force_samples++;
    }
    if (!quiet) printf("%-10d | %-15.3f | %-10.1f\n", frame, ms, 1000.0 / ms);  // ref:366
  }
  const double avg = sum / (frames > 0 ? frames : 1);
  if (!dump_path && !snap_path) {  // per-stage times of one more step (not part of the frames; skipped when the
    CK(bh_set_timing(c, 1));       // final state is dumped, which must be the state after the last frame)
    CK(bh_step(c));
    CK(bh_sync(c));
    const double f = sum_force;
    CK(bh_get_stats(c, &st));
    sum_force = f;
  }
  printf("------------------------------------------\n");
  printf("N=%d ic=%s theta=%.2f steps=%d avg %.3f ms/step  %.3e particles/s/step\n", N,
         plummer ? "plummer" : (msvc ? "disc(msvc rand)" : "disc"), p.theta, frames, avg, (double)N / (avg * 1e-3));
  printf("stages of one further step (ms): bbox %.3f morton %.3f sort %.3f build %.3f com %.3f force %.3f integrate %.3f | "
         "cells %d entries %d depth %d flags %d | avg force %.3f ms\n",
         st.ms_bbox, st.ms_morton, st.ms_sort, st.ms_build, st.ms_com, st.ms_force, st.ms_integrate,
         st.n_internal, st.n_entries, st.max_level, st.status_flags, sum_force / (force_samples > 0 ? force_samples : 1));
  if (dump_path || snap_path) {
    CK(bh_download(c, x.data(), y.data(), z.data(), vx.data(), vy.data(), vz.data()));
    if (dump_path)
      CK(bh_write_text(dump_path, N, warmup + frames, p.theta, p.dt, x.data(), y.data(), z.data(), vx.data(),
                       vy.data(), vz.data()));
    if (snap_path) {
      CK(bh_download_mass(c, m.data()));
      CK(bh_write_snapshot(snap_path, N, warmup + frames, &p, x.data(), y.data(), z.data(), vx.data(),
                           vy.data(), vz.data(), m.data()));
    }
  }
  bh_destroy(c);  // ref:372-387
  return 0;
}
