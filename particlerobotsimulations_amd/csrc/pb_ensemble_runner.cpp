// pb_ensemble_runner.cpp -- bin/particlebot_ensemble: an ensemble of independent simulations sharded
// over the GPUs of a node, ONE PROCESS PER GPU, C++ only (no Python, no torch), summary rows gathered
// with one ncclAllGather over RCCL/xGMI (include/particlebot_ensemble.h).
//
//   particlebot_ensemble <config.cfg> --members M [--seed0 S] [--set NAME VALUE]...
//                        [--sweep KEY V1 V2 ...]... [--out FILE] [--csv-dir DIR] [--sub-batch B] [--host-threads T]
//                        [--checkpoint DIR | --resume DIR] [--rendezvous FILE]
//
// Launch with any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK (torchrun), OMPI_COMM_WORLD_* or
// SLURM_PROCID / SLURM_NTASKS / SLURM_LOCALID, or by hand:  RANK=r WORLD_SIZE=N LOCAL_RANK=r ...
// Member k (seed seed0 + k, sweep value V[k mod #V] of every --sweep, `--sweep seed ...` replacing the seed) runs on
// rank k mod N.  A rank's members run through the
// placement/stepping pipeline (pbEnsemblePipeline*): sub-batches of B members (0: all at once, the default; -1: whole
// placement rounds of the producer pool that bring a sub-batch to ~3 x 10^6 bots -- for members of ~10^5 bots and
// more), the host placing the next ones
// while the device steps the current one.  --checkpoint DIR saves every member exactly at each summary row
// (DIR/rank<r>/...); --resume DIR continues a killed sweep from there (same M, N and B), bit-identically.
// --csv-dir DIR: member k also writes DIR/member_<k>.csv, byte for byte the CSV the reference writes for that member
// run on its own with testing 0 (seed line, header, one row per dump interval with its fp32 centroid).
//
// Rendezvous: rank 0 creates the RCCL unique id and serves it over TCP on MASTER_ADDR:(MASTER_PORT + 17) -- not
// MASTER_PORT itself, which a launcher's own store may hold; $PB_RENDEZVOUS_PORT overrides; default 127.0.0.1:29417 --
// to the other ranks, which retry until it listens: nothing left behind on disk, nothing a crashed earlier launch
// could have left either.  --rendezvous FILE exchanges it through that file instead
// (written to a temporary name and renamed; taken only if it carries this launch's token and rank 0's process, whose id
// it also carries, is alive: SINGLE HOST ONLY).  MASTER_ADDR may be a host name (getaddrinfo); rank 0 listens on every
// interface, serves until every rank has acknowledged the id, and gives up after one overall deadline.
// A rank that fails still takes part in the collectives: an error flag is reduced first, then all ranks leave.
#include <arpa/inet.h>
#include <hip/hip_runtime.h>
#include <netdb.h>
#include <netinet/in.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// ---- rendezvous over TCP ---------------------------------------------------------------------------------------
static bool sendAll(int fd, const void *p, size_t n) {
  const char *c = (const char *)p;
  while (n) {
    const ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) return false;
    c += k, n -= (size_t)k;
  }
  return true;
}
static bool recvAll(int fd, void *p, size_t n) {
  char *c = (char *)p;
  while (n) {
    const ssize_t k = recv(fd, c, n, 0);
    if (k <= 0) return false;
    c += k, n -= (size_t)k;
  }
  return true;
}

// MASTER_ADDR may be a dotted quad or a host name (torchrun and SLURM commonly pass names): resolved with
// getaddrinfo; a name that does not resolve is an error, not a silent fall-back to the loopback.
static bool resolveV4(const char *addr, int port, sockaddr_in *out) {
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  char portText[16];
  snprintf(portText, sizeof portText, "%d", port);
  const int rc = getaddrinfo(addr, portText, &hints, &res);
  if (rc != 0 || !res) {
    fprintf(stderr, "rank %d: cannot resolve MASTER_ADDR '%s': %s\n", g_rank, addr, gai_strerror(rc));
    return false;
  }
  memcpy(out, res->ai_addr, sizeof *out);
  freeaddrinfo(res);
  return true;
}

// The launch token: MASTER_PORT and (when the launcher provides one) its run id, so that a process that is not part
// of THIS launch cannot mark a rank as served, and a stale rendezvous file of another launch is not taken.
static uint64_t launchToken() {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const char *t) {
    for (; t && *t; t++) h = (h ^ (unsigned char)*t) * 1099511628211ull;
    h = (h ^ 0xffu) * 1099511628211ull;
  };
  mix(getenv("MASTER_PORT"));
  mix(getenv("TORCHELASTIC_RUN_ID"));
  mix(getenv("SLURM_JOB_ID"));
  mix(getenv("PB_LAUNCH_TOKEN"));
  return h;
}

// rank 0: listen on every interface, hand the id to world - 1 peers (each says "PBID" + its rank + the launch token
// first) and leave only when each of them HAS it; one overall deadline, however many stray connections come by.
static bool serveId(int port, int world, const ncclUniqueId &id, double timeoutSeconds) {
  const int ls = socket(AF_INET, SOCK_STREAM, 0);
  if (ls < 0) return false;
  int one = 1;
  setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  sockaddr_in sa{};
  sa.sin_family = AF_INET;
  sa.sin_port = htons((uint16_t)port);
  sa.sin_addr.s_addr = htonl(INADDR_ANY);
  if (bind(ls, (sockaddr *)&sa, sizeof sa) != 0 || listen(ls, world) != 0) {
    close(ls);
    return false;
  }
  const auto t0 = std::chrono::steady_clock::now();
  const uint64_t token = launchToken();
  std::vector<char> served(world, 0);
  int left = world - 1;
  while (left > 0) {
    const double remaining = timeoutSeconds - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (remaining <= 0) break;
    timeval tv{(time_t)remaining, (suseconds_t)((remaining - (double)(time_t)remaining) * 1e6)};
    if (tv.tv_sec == 0 && tv.tv_usec < 1000) tv.tv_usec = 1000;
    setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    const int fd = accept(ls, nullptr, nullptr);
    if (fd < 0) continue;  // timed out: the deadline test above ends the loop
    timeval peer{2, 0};    // a connection that says nothing does not hold the others up for long
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &peer, sizeof peer);
    char hello[4], ack = 0;
    int32_t r = -1;
    uint64_t theirs = 0;
    if (recvAll(fd, hello, 4) && memcmp(hello, "PBID", 4) == 0 && recvAll(fd, &r, 4) && recvAll(fd, &theirs, 8) &&
        theirs == token && r > 0 && r < world && sendAll(fd, &id, sizeof id) && recvAll(fd, &ack, 1) && ack == 'K' &&
        !served[r]) {
      served[r] = 1;
      left--;
    }
    close(fd);
  }
  close(ls);
  return left == 0;
}

static bool fetchIdTcp(const char *addr, int port, int rank, ncclUniqueId *id, double timeoutSeconds) {
  sockaddr_in sa{};
  if (!resolveV4(addr, port, &sa)) return false;
  const auto t0 = std::chrono::steady_clock::now();
  const uint64_t token = launchToken();
  for (;;) {
    const int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return false;
    if (connect(fd, (sockaddr *)&sa, sizeof sa) == 0) {
      const int32_t r = rank;
      const bool ok = sendAll(fd, "PBID", 4) && sendAll(fd, &r, 4) && sendAll(fd, &token, 8) &&
                      recvAll(fd, id, sizeof *id) && sendAll(fd, "K", 1);
      close(fd);
      if (ok) return true;
    } else {
      close(fd);
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutSeconds) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

// ---- rendezvous through a file (only on request) ------------------------------------------------------------------
// Single host only (rank 0's pid is tested with kill(pid, 0), which means nothing in another pid namespace); the
// record also carries the launch token, so a live process with a recycled pid does not make a stale file look current.
struct IdFile {
  char magic[8];
  int32_t pid;  // rank 0's process: a file whose writer is gone is stale
  uint64_t token;
  ncclUniqueId id;
};
static bool publishId(const std::string &path, const ncclUniqueId &id) {
  (void)remove(path.c_str());  // whatever an earlier launch left
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  IdFile rec;
  memcpy(rec.magic, "PBIDF2\0", 8);
  rec.pid = (int32_t)getpid();
  rec.token = launchToken();
  rec.id = id;
  const bool ok = fwrite(&rec, sizeof rec, 1, f) == 1;
  if (fclose(f) != 0 || !ok) return false;
  return rename(tmp.c_str(), path.c_str()) == 0;
}
static bool fetchIdFile(const std::string &path, ncclUniqueId *id, double timeoutSeconds) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    if (FILE *f = fopen(path.c_str(), "rb")) {
      IdFile rec;
      const bool ok = fread(&rec, sizeof rec, 1, f) == 1 && memcmp(rec.magic, "PBIDF2\0", 8) == 0;
      fclose(f);
      if (ok && rec.token == launchToken() && rec.pid > 0 &&
          (kill(rec.pid, 0) == 0 || errno == EPERM)) {  // this launch's, and its writer is still running
        *id = rec.id;
        return true;
      }
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutSeconds) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

int main(int argc, char **argv) {
  std::string cfgPath, outPath, rendezvous, ckptDir, csvDir;
  std::vector<std::pair<std::string, std::string>> sets;
  std::vector<std::pair<std::string, std::vector<std::string>>> sweeps;  // --sweep may be given several times
  int members = 32, subBatch = 0, hostThreads = 0, maxRows = 4096;
  bool resume = false, rendezvousTest = false, cartesian = false;
  long seed0 = 1000;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--members") && i + 1 < argc) members = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--seed0") && i + 1 < argc) seed0 = atol(argv[++i]);
    else if (!strcmp(argv[i], "--out") && i + 1 < argc) outPath = argv[++i];
    else if (!strcmp(argv[i], "--csv-dir") && i + 1 < argc) csvDir = argv[++i];
    else if (!strcmp(argv[i], "--rendezvous") && i + 1 < argc) rendezvous = argv[++i];
    else if (!strcmp(argv[i], "--rendezvous-test")) rendezvousTest = true;
    else if (!strcmp(argv[i], "--sub-batch") && i + 1 < argc) subBatch = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--host-threads") && i + 1 < argc) hostThreads = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--max-rows") && i + 1 < argc) maxRows = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--cartesian")) cartesian = true;
    else if (!strcmp(argv[i], "--checkpoint") && i + 1 < argc) ckptDir = argv[++i];
    else if (!strcmp(argv[i], "--resume") && i + 1 < argc) {
      ckptDir = argv[++i];
      resume = true;
    } else if (!strcmp(argv[i], "--set") && i + 2 < argc) {
      sets.emplace_back(argv[i + 1], argv[i + 2]);
      i += 2;
    } else if (!strcmp(argv[i], "--sweep") && i + 2 < argc) {
      sweeps.emplace_back(argv[++i], std::vector<std::string>());
      while (i + 1 < argc && strncmp(argv[i + 1], "--", 2) != 0) sweeps.back().second.push_back(argv[++i]);
    } else if (argv[i][0] != '-' && cfgPath.empty()) cfgPath = argv[i];
    else {
      fprintf(stderr, "usage: %s <config.cfg> --members M [--seed0 S] [--set NAME VALUE]... [--sweep KEY V1 V2 ...]... "
                      "[--cartesian] [--max-rows R] [--out FILE] [--csv-dir DIR] [--sub-batch B] [--host-threads T] "
                      "[--checkpoint DIR | --resume DIR] "
                      "[--rendezvous FILE (single host)]\n", argv[0]);
      return 2;
    }
  }
  if (!csvDir.empty() && !ckptDir.empty()) {
    fprintf(stderr, "particlebot_ensemble: --csv-dir cannot be combined with --checkpoint / --resume\n");
    return 2;
  }
  if (!rendezvousTest && (cfgPath.empty() || members < 1 || maxRows < 1)) {
    fprintf(stderr, "particlebot_ensemble: a configuration file, --members >= 1 and --max-rows >= 1 are required\n");
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
  if (rendezvousTest) {
    // (CPU tests of the id exchange, no GPU involved: rank 0 serves a known 128-byte pattern, the others print what
    //  they received)
    const char *addr0 = getenv("MASTER_ADDR");
    const char *pt = getenv("MASTER_PORT"), *op = getenv("PB_RENDEZVOUS_PORT");
    const int port0 = op ? atoi(op) : (pt ? atoi(pt) : 29400) + 17;
    ncclUniqueId fake;
    unsigned char *fb = (unsigned char *)&fake;
    for (size_t i = 0; i < sizeof fake; i++) fb[i] = (unsigned char)(i * 7 + 3);
    bool ok;
    if (rank == 0) {
      ok = rendezvous.empty() ? serveId(port0, world, fake, 30.0) : publishId(rendezvous, fake);
      if (ok && !rendezvous.empty()) std::this_thread::sleep_for(std::chrono::milliseconds(1500));  // stay alive for the readers
    } else {
      ncclUniqueId got;
      memset(&got, 0, sizeof got);
      ok = rendezvous.empty() ? fetchIdTcp(addr0 ? addr0 : "127.0.0.1", port0, rank, &got, 30.0)
                              : fetchIdFile(rendezvous, &got, 30.0);
      ok = ok && memcmp(&got, &fake, sizeof fake) == 0;
    }
    printf("rendezvous-test rank %d of %d: %s\n", rank, world, ok ? "ok" : "FAILED");
    return ok ? 0 : 1;
  }
  int ndev = 0;
  CHECK_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 1) {
    fprintf(stderr, "rank %d: no HIP device visible (there is no CPU fallback)\n", rank);
    return 1;
  }
  CHECK_HIP(hipSetDevice(local % ndev));

  // ---- RCCL communicator: one rank per GPU -----------------------------------------------------
  ncclUniqueId id;
  const char *addr = getenv("MASTER_ADDR");
  const char *portText = getenv("MASTER_PORT");
  const char *ownPort = getenv("PB_RENDEZVOUS_PORT");
  const int port = ownPort ? atoi(ownPort) : (portText ? atoi(portText) : 29400) + 17;
  if (rank == 0) {
    CHECK_NCCL(ncclGetUniqueId(&id));
    if (world > 1) {
      const bool ok = rendezvous.empty() ? serveId(port, world, id, 120.0)
                                         : publishId(rendezvous, id);
      if (!ok) {
        fprintf(stderr, "rank 0: rendezvous failed (%s)\n",
                rendezvous.empty() ? "could not serve the RCCL id to every rank over TCP" : rendezvous.c_str());
        return 1;
      }
    }
  } else {
    const bool ok = rendezvous.empty() ? fetchIdTcp(addr ? addr : "127.0.0.1", port, rank, &id, 120.0)
                                       : fetchIdFile(rendezvous, &id, 120.0);
    if (!ok) {
      fprintf(stderr, "rank %d: no RCCL id from rank 0 after 120 s\n", rank);
      return 1;
    }
  }
  // RCCL prints a version banner to stdout when the communicator is created; stdout carries ONE line, rank 0's JSON:
  // file descriptor 1 points at stderr from here on, the JSON goes to the saved descriptor
  fflush(stdout);
  const int jsonFd = dup(1);
  if (jsonFd >= 0) (void)dup2(2, 1);
  FILE *const jsonOut = jsonFd >= 0 ? fdopen(jsonFd, "w") : stdout;
  ncclComm_t comm;
  CHECK_NCCL(ncclCommInitRank(&comm, world, id, rank));
  // what the communicator itself says (the line reports it: "RCCL saw N ranks" is checkable from the output)
  int commRanks = 0, commDevice = -1;
  CHECK_NCCL(ncclCommCount(comm, &commRanks));
  CHECK_NCCL(ncclCommCuDevice(comm, &commDevice));
  if (commRanks != world) {
    fprintf(stderr, "rank %d: the communicator has %d ranks, the launcher said %d\n", rank, commRanks, world);
    return 1;
  }
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));

  // ---- this rank's members -----------------------------------------------------------------------
  std::string common;
  for (auto &kv : sets) common += kv.first + "\n" + kv.second + "\n";
  std::vector<std::string> over;
  // --cartesian: the swept keys form a grid (first --sweep fastest) and every grid point runs under each seed --
  // member k is grid point k mod G under seed seed0 + k / G, G = the product of the sweeps' lengths -- so that
  // `--members 1024 --sweep nDead <64 values> --cartesian` is BASELINE configs[4]'s "64 points x 16 seeds".  Members of
  // one seed then share their placement (the pipeline places each distinct blob once).
  long grid = 1;
  for (auto &sw : sweeps)
    if (cartesian && !sw.second.empty()) grid *= (long)sw.second.size();
  for (int k = rank; k < members; k += world) {
    std::string o = "seed\n" + std::to_string(seed0 + (cartesian ? k / grid : k));
    // (in the order given; a later key wins, so `--sweep seed ...` replaces seed0 + k)
    long stride = 1;
    for (auto &sw : sweeps) {
      if (sw.first.empty() || sw.second.empty()) continue;
      const size_t len = sw.second.size();
      o += "\n" + sw.first + "\n" + sw.second[cartesian ? (size_t)((k % grid) / stride) % len : (size_t)k % len];
      stride *= (long)len;
    }
    over.push_back(o);
  }
  std::vector<const char *> overPtr;
  for (auto &o : over) overPtr.push_back(o.c_str());
  const int mine = (int)over.size(), per = pbEnsembleShard(members, 0, world);
  std::vector<float> rows((size_t)(mine ? mine : 1) * maxRows * 4, 0.0f);
  int nrows = 0, failed = 0;
  long steps = 0;
  unsigned nbots = 0;
  pbEnsembleTimings tm{};
  std::string hostRule;
  {
    pbHostResources res;
    if (pbHostGetResources(&res) == 0) hostRule = res.rule;  // how this rank's producer pool was sized (quota, ranks, NUMA)
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (mine > 0) {
    std::string myCkpt;
    if (!ckptDir.empty()) {
      (void)mkdir(ckptDir.c_str(), 0777);
      myCkpt = ckptDir + "/rank" + std::to_string(rank) + "of" + std::to_string(world);
    }
    void *e = pbEnsemblePipelineCreateCheckpointed(cfgPath.c_str(), common.empty() ? nullptr : common.c_str(),
                                                   overPtr.data(), mine, subBatch, hostThreads, 0,
                                                   myCkpt.empty() ? nullptr : myCkpt.c_str(), resume ? 1 : 0);
    if (e && !csvDir.empty()) {
      // member k of the whole ensemble writes DIR/member_<k>.csv: the reference's own CSV of that member (testing 0)
      std::vector<int> ids;
      for (int k = rank; k < members; k += world) ids.push_back(k);
      if (pbEnsemblePipelineSetCsvDir(e, csvDir.c_str(), ids.data()) != 0) {
        fprintf(stderr, "rank %d: cannot write under %s\n", rank, csvDir.c_str());
        pbEnsemblePipelineDestroy(e);
        e = nullptr;
      }
    }
    if (!e) {
      fprintf(stderr, "rank %d: pbEnsemblePipelineCreate failed\n", rank);
      failed = 1;
    } else {
      steps = pbEnsemblePipelineRun(e, (long)1 << 62, rows.data(), maxRows, &nrows, &tm);
      nbots = pbEnsemblePipelineNumBots(e);
      pbEnsemblePipelineDestroy(e);
      if (steps < 0) {
        fprintf(stderr, "rank %d: pbEnsemblePipelineRun failed\n", rank);
        failed = 1;
        steps = 0;
        nrows = 0;
      }
    }
  }
  double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // ---- the one exchange: summary rows (plus a few scalars), over RCCL ----------------------------
  // scalars first, by max: wall time, row count, bots per member, steps, ERROR FLAG (every rank gets here,
  // whatever happened to its members, so that no rank waits in a collective for one that has left)
  double hs[5] = {wall, (double)nrows, (double)nbots, (double)steps, (double)failed}, *ds = nullptr;
  CHECK_HIP(hipMalloc((void **)&ds, sizeof hs));
  CHECK_HIP(hipMemcpyAsync(ds, hs, sizeof hs, hipMemcpyHostToDevice, stream));
  CHECK_NCCL(ncclAllReduce(ds, ds, 5, ncclDouble, ncclMax, comm, stream));
  CHECK_HIP(hipMemcpyAsync(hs, ds, sizeof hs, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  if (hs[4] != 0.0) {
    if (rank == 0) fprintf(stderr, "particlebot_ensemble: a rank failed; no result\n");
    (void)hipFree(ds);
    ncclCommDestroy(comm);
    return 1;
  }
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
    for (int k = 0; k < members && allRows > 0; k++) {
      const float *first = &all[((size_t)k * allRows) * 4], *last = &all[((size_t)k * allRows + allRows - 1) * 4];
      const double d = (double)first[3] - (double)last[3];
      sum += d;
      sum2 += d * d;
    }
    const double mean = sum / members, var = sum2 / members - mean * mean;
    fprintf(jsonOut, "{\"cfg\": \"%s\", \"members\": %d, \"n_gpus\": %d, \"bots_per_member\": %d, \"steps_per_member\": %ld, "
           "\"rows_per_member\": %d, \"wall_s\": %.6f, \"sims_per_s\": %.6g, \"particle_steps_per_s\": %.6g, "
           "\"progress_toward_light_mean\": %.9g, \"progress_toward_light_std\": %.9g, "
           "\"pipeline_rank0\": {\"sub_batch\": %d, \"sub_batches\": %d, \"lanes\": %d, \"host_threads\": %d, \"placement_cpu_s\": %.4f, "
           "\"placement_wait_s\": %.4f, \"upload_s\": %.4f, \"device_s\": %.4f, \"pinned_to_gpu_numa_node\": %s, "
           "\"numa_node\": %d, \"bound\": \"%s\"}, \"host\": \"%s\", \"resumed\": %s, "
           "\"collective\": {\"backend\": \"rccl\", \"ranks\": %d, \"rank0_device\": %d, \"placements_run_rank0\": %d, "
           "\"what\": \"ncclAllGather of %zu floats per rank\"}}\n",
           cfgPath.c_str(), members, world, (int)hs[2], (long)hs[3], allRows, wall, members / wall,
           (double)members * hs[2] * hs[3] / wall, mean, sqrt(var > 0 ? var : 0), tm.sub_batch, tm.sub_batches,
           tm.lanes, tm.host_threads, tm.placement_cpu_s, tm.placement_wait_s, tm.upload_s, tm.device_s, tm.pinned ? "true" : "false",
           tm.numa_node,
           // host-bound: this rank's placement CPU-seconds over its producer threads exceed the device's time
           tm.placement_cpu_s / (tm.host_threads > 0 ? tm.host_threads : 1) > tm.device_s + tm.upload_s ? "host" : "device",
           hostRule.c_str(), resume ? "true" : "false", commRanks, commDevice, tm.placements_run, block);
    fflush(jsonOut);
    if (!outPath.empty()) {
      FILE *f = fopen(outPath.c_str(), "wb");
      if (!f || fwrite(all.data(), sizeof(float), (size_t)members * allRows * 4, f) != (size_t)members * allRows * 4) {
        fprintf(stderr, "cannot write %s\n", outPath.c_str());
        rc = 1;
      }
      if (f) fclose(f);
    }
    if (world > 1 && !rendezvous.empty()) (void)remove(rendezvous.c_str());
  }
  (void)hipFree(ds);
  (void)hipFree(dSend);
  (void)hipFree(dRecv);
  (void)hipStreamDestroy(stream);
  ncclCommDestroy(comm);
  return rc;
}
