// pb_ensemble_runner.cpp -- bin/particlebot_ensemble: an ensemble of independent simulations sharded
// over the GPUs of a node, ONE PROCESS PER GPU, C++ only (no Python, no torch), summary rows gathered
// with one ncclAllGather over RCCL/xGMI (include/particlebot_ensemble.h).
//
//   particlebot_ensemble <config.cfg> --members M [--seed0 S] [--set NAME VALUE]...
//                        [--sweep KEY V1 V2 ...] [--out FILE] [--rendezvous FILE]
//
// Launch with any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK (torchrun), OMPI_COMM_WORLD_* or
// SLURM_PROCID / SLURM_NTASKS / SLURM_LOCALID, or by hand:  RANK=r WORLD_SIZE=N LOCAL_RANK=r ...
// Member k (seed seed0 + k, sweep value V[k mod #V]) runs on rank k mod N.  Rank 0 creates the RCCL
// unique id and publishes it through the rendezvous file (default /tmp/particlebot_ensemble_<port>.id
// with <port> = $MASTER_PORT or 0; written to a temporary name and renamed, the other ranks poll).
// Rank 0 prints one JSON line and, with --out, writes the gathered rows as float32 [M][rows][4].
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "particlebot_ensemble.h"
#include "pb_config.hpp"

#define CHECK_HIP(x)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "rank %d: %s failed: %s\n", g_rank, #x, hipGetErrorString(e_));      \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)
#define CHECK_NCCL(x)                                                                      \
  do {                                                                                     \
    ncclResult_t r_ = (x);                                                                 \
    if (r_ != ncclSuccess) {                                                               \
      fprintf(stderr, "rank %d: %s failed: %s\n", g_rank, #x, ncclGetErrorString(r_));     \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

static int g_rank = 0;

static int envInt(const char *const *names, int fallback) {
  for (; *names; names++)
    if (const char *v = getenv(*names)) return atoi(v);
  return fallback;
}

static bool publishId(const std::string &path, const ncclUniqueId &id) {
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(&id, sizeof id, 1, f) == 1;
  if (fclose(f) != 0 || !ok) return false;
  return rename(tmp.c_str(), path.c_str()) == 0;
}

// A file left behind by an earlier, crashed launch with the same port must not be taken for this
// launch's id: only a file written no earlier than a minute before this process started counts.
static bool fetchId(const std::string &path, ncclUniqueId *id, double timeoutSeconds) {
  const auto t0 = std::chrono::steady_clock::now();
  const time_t started = time(nullptr);
  for (;;) {
    struct stat sb;
    if (stat(path.c_str(), &sb) == 0 && sb.st_mtime >= started - 60) {
      if (FILE *f = fopen(path.c_str(), "rb")) {
        const bool ok = fread(id, sizeof *id, 1, f) == 1;
        fclose(f);
        if (ok) return true;
      }
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutSeconds) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

int main(int argc, char **argv) {
  std::string cfgPath, outPath, rendezvous;
  std::vector<std::pair<std::string, std::string>> sets;
  std::string sweepKey;
  std::vector<std::string> sweepVals;
  int members = 32;
  long seed0 = 1000;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--members") && i + 1 < argc) members = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--seed0") && i + 1 < argc) seed0 = atol(argv[++i]);
    else if (!strcmp(argv[i], "--out") && i + 1 < argc) outPath = argv[++i];
    else if (!strcmp(argv[i], "--rendezvous") && i + 1 < argc) rendezvous = argv[++i];
    else if (!strcmp(argv[i], "--set") && i + 2 < argc) {
      sets.emplace_back(argv[i + 1], argv[i + 2]);
      i += 2;
    } else if (!strcmp(argv[i], "--sweep") && i + 2 < argc) {
      sweepKey = argv[++i];
      while (i + 1 < argc && strncmp(argv[i + 1], "--", 2) != 0) sweepVals.push_back(argv[++i]);
    } else if (argv[i][0] != '-' && cfgPath.empty()) cfgPath = argv[i];
    else {
      fprintf(stderr, "usage: %s <config.cfg> --members M [--seed0 S] [--set NAME VALUE]... "
                      "[--sweep KEY V1 V2 ...] [--out FILE] [--rendezvous FILE]\n", argv[0]);
      return 2;
    }
  }
  if (cfgPath.empty() || members < 1) {
    fprintf(stderr, "particlebot_ensemble: a configuration file and --members >= 1 are required\n");
    return 2;
  }
  const char *rankNames[] = {"RANK", "OMPI_COMM_WORLD_RANK", "SLURM_PROCID", nullptr};
  const char *worldNames[] = {"WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "SLURM_NTASKS", nullptr};
  const char *localNames[] = {"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", nullptr};
  const int rank = g_rank = envInt(rankNames, 0), world = envInt(worldNames, 1);
  const int local = envInt(localNames, rank);
  if (world < 1 || rank < 0 || rank >= world) {
    fprintf(stderr, "particlebot_ensemble: bad rank %d of %d\n", rank, world);
    return 2;
  }
  int ndev = 0;
  CHECK_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 1) {
    fprintf(stderr, "rank %d: no HIP device visible (there is no CPU fallback)\n", rank);
    return 1;
  }
  CHECK_HIP(hipSetDevice(local % ndev));

  // ---- RCCL communicator: one rank per GPU -----------------------------------------------------
  if (rendezvous.empty()) {
    const char *port = getenv("MASTER_PORT");
    rendezvous = std::string("/tmp/particlebot_ensemble_") + (port ? port : "0") + ".id";
  }
  ncclUniqueId id;
  if (rank == 0) {
    CHECK_NCCL(ncclGetUniqueId(&id));
    if (world > 1 && !publishId(rendezvous, id)) {
      fprintf(stderr, "rank 0: cannot write the rendezvous file %s\n", rendezvous.c_str());
      return 1;
    }
  } else if (!fetchId(rendezvous, &id, 120.0)) {
    fprintf(stderr, "rank %d: no RCCL id in %s after 120 s\n", rank, rendezvous.c_str());
    return 1;
  }
  ncclComm_t comm;
  CHECK_NCCL(ncclCommInitRank(&comm, world, id, rank));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));

  // ---- this rank's members -----------------------------------------------------------------------
  std::string common;
  for (auto &kv : sets) common += kv.first + "\n" + kv.second + "\n";
  std::vector<std::string> over;
  for (int k = rank; k < members; k += world) {
    std::string o = "seed\n" + std::to_string(seed0 + k);
    if (!sweepKey.empty() && !sweepVals.empty()) o += "\n" + sweepKey + "\n" + sweepVals[(size_t)k % sweepVals.size()];
    over.push_back(o);
  }
  std::vector<const char *> overPtr;
  for (auto &o : over) overPtr.push_back(o.c_str());
  const int mine = (int)over.size(), per = pbEnsembleShard(members, 0, world);
  const int maxRows = 4096;
  std::vector<float> rows((size_t)(mine ? mine : 1) * maxRows * 4, 0.0f);
  int nrows = 0;
  long steps = 0;
  unsigned nbots = 0;
  const auto t0 = std::chrono::steady_clock::now();
  if (mine > 0) {
    void *e = pbEnsembleCreate(cfgPath.c_str(), common.empty() ? nullptr : common.c_str(), overPtr.data(), mine);
    if (!e) {
      fprintf(stderr, "rank %d: pbEnsembleCreate failed\n", rank);
      return 1;
    }
    nbots = pbEnsembleNumBots(e);
    steps = pbEnsembleRun(e, rows.data(), maxRows, &nrows);
    pbEnsembleDestroy(e);
    if (steps < 0) {
      fprintf(stderr, "rank %d: pbEnsembleRun failed\n", rank);
      return 1;
    }
  }
  double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // ---- the one exchange: summary rows (plus three scalars), over RCCL ----------------------------
  // scalars first: max wall time, max row count, bots per member, steps (ranks without members send 0)
  double hs[4] = {wall, (double)nrows, (double)nbots, (double)steps}, *ds = nullptr;
  CHECK_HIP(hipMalloc((void **)&ds, sizeof hs));
  CHECK_HIP(hipMemcpyAsync(ds, hs, sizeof hs, hipMemcpyHostToDevice, stream));
  CHECK_NCCL(ncclAllReduce(ds, ds, 4, ncclDouble, ncclMax, comm, stream));
  CHECK_HIP(hipMemcpyAsync(hs, ds, sizeof hs, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  wall = hs[0];
  const int allRows = (int)hs[1];
  const size_t block = (size_t)per * allRows * 4;
  std::vector<float> send(block ? block : 1, std::numeric_limits<float>::quiet_NaN());
  for (int j = 0; j < mine; j++)
    for (int r = 0; r < nrows; r++)
      memcpy(&send[((size_t)j * allRows + r) * 4], &rows[((size_t)j * maxRows + r) * 4], 4 * sizeof(float));
  float *dSend = nullptr, *dRecv = nullptr;
  CHECK_HIP(hipMalloc((void **)&dSend, sizeof(float) * (block ? block : 1)));
  CHECK_HIP(hipMalloc((void **)&dRecv, sizeof(float) * (block ? block : 1) * world));
  CHECK_HIP(hipMemcpyAsync(dSend, send.data(), sizeof(float) * block, hipMemcpyHostToDevice, stream));
  if (block) CHECK_NCCL(ncclAllGather(dSend, dRecv, block, ncclFloat, comm, stream));
  std::vector<float> gathered((block ? block : 1) * world), all((size_t)members * allRows * 4 + 1);
  CHECK_HIP(hipMemcpyAsync(gathered.data(), dRecv, sizeof(float) * block * world, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  if (pbEnsembleAssemble(members, world, allRows, gathered.data(), all.data()) != 0) return 1;

  int rc = 0;
  if (rank == 0) {
    // progress of each member's centre of mass toward the light = decrease of the distance column
    double sum = 0, sum2 = 0;
    for (int k = 0; k < members; k++) {
      const float *first = &all[((size_t)k * allRows) * 4], *last = &all[((size_t)k * allRows + allRows - 1) * 4];
      const double d = (double)first[3] - (double)last[3];
      sum += d;
      sum2 += d * d;
    }
    const double mean = sum / members, var = sum2 / members - mean * mean;
    printf("{\"cfg\": \"%s\", \"members\": %d, \"n_gpus\": %d, \"bots_per_member\": %d, \"steps_per_member\": %ld, "
           "\"rows_per_member\": %d, \"wall_s\": %.6f, \"sims_per_s\": %.6g, \"particle_steps_per_s\": %.6g, "
           "\"progress_toward_light_mean\": %.9g, \"progress_toward_light_std\": %.9g, "
           "\"collective\": \"ncclAllGather of %zu floats per rank (RCCL)\"}\n",
           cfgPath.c_str(), members, world, (int)hs[2], (long)hs[3], allRows, wall, members / wall,
           (double)members * hs[2] * hs[3] / wall, mean, sqrt(var > 0 ? var : 0), block);
    if (!outPath.empty()) {
      FILE *f = fopen(outPath.c_str(), "wb");
      if (!f || fwrite(all.data(), sizeof(float), (size_t)members * allRows * 4, f) != (size_t)members * allRows * 4) {
        fprintf(stderr, "cannot write %s\n", outPath.c_str());
        rc = 1;
      }
      if (f) fclose(f);
    }
    if (world > 1) (void)remove(rendezvous.c_str());
  }
  (void)hipFree(ds);
  (void)hipFree(dSend);
  (void)hipFree(dRecv);
  (void)hipStreamDestroy(stream);
  ncclCommDestroy(comm);
  return rc;
}
