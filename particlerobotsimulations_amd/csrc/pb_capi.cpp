// pb_capi.cpp -- C wrappers around the C++ host side (class Particlebot + .cfg loader) so that
// scripts and tests can drive it through ctypes.  Exported from libparticlebot_host.so.
#include <gnu/libc-version.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "particlebot.h"
#include "particlebot_ensemble.h"
#include "pb_config.hpp"
#include "pb_xorwow.hpp"

extern "C" {

// flat, pointer-free view of a resolved configuration (for inspection from scripts)
struct pbFlatConfig {
  uint32_t gridSizeX, gridSizeY, numCells;
  float worldOriginX, worldOriginY, cellSizeX, cellSizeY;
  uint32_t nCells;
  int32_t nDead;
  float gravity, spring, damping, shear, attraction, boundaryDamping, friction;
  float massFactor, frictionFactor, radFactor, attractionFactor;
  float constraint, constraint_contraction;
  int32_t centroid_steps;
  float centroid_int, centroid_radius;
  float light_x, light_y, phase_update_interval;
  int32_t control, config;
  float min_radius, max_radius, rise_period, freq;
  int32_t nobstacles;
  float x1obs[PB_MAX_OBSTACLES], x2obs[PB_MAX_OBSTACLES], y1obs[PB_MAX_OBSTACLES], y2obs[PB_MAX_OBSTACLES];
  int32_t n_cir_obstacles;
  float x_cir_obs[PB_MAX_OBSTACLES], y_cir_obs[PB_MAX_OBSTACLES], r_cir_obs[PB_MAX_OBSTACLES];
  int32_t Nx;
  float phase_std;
  uint32_t seed;
  uint32_t light_shadow, testing, constrained_contraction, display_shadow;
  float time_to_dead, max_time;
  float timestep, sort_interval, dump_interval;
  float camera_x, camera_y, light_radius;
  int32_t display_interval, video_interval;
  char csv_filename[300];
  char video_filename[300];
  float wallHalf;
  int rngKind;  // pb_rng (PB_RNG_*)
};

}  // extern "C"

namespace {

// overrides: "name\nvalue\nname\nvalue..." applied after the file, through the same setParam
void applyOverrides(PbRunConfig &cfg, const char *overrides) {
  if (!overrides) return;
  std::string s(overrides);
  size_t pos = 0;
  while (pos < s.size()) {
    size_t e1 = s.find('\n', pos);
    if (e1 == std::string::npos) break;
    size_t e2 = s.find('\n', e1 + 1);
    if (e2 == std::string::npos) e2 = s.size();
    cfg.setParam(s.substr(pos, e1 - pos), s.substr(e1 + 1, e2 - e1 - 1));
    pos = e2 + 1;
  }
}

void flatten(const PbRunConfig &cfg, pbFlatConfig *o) {
  memset(o, 0, sizeof(*o));
  const SimParams &p = cfg.params;
  o->gridSizeX = p.gridSize.x;
  o->gridSizeY = p.gridSize.y;
  o->numCells = p.numCells;
  o->worldOriginX = p.worldOrigin.x;
  o->worldOriginY = p.worldOrigin.y;
  o->cellSizeX = p.cellSize.x;
  o->cellSizeY = p.cellSize.y;
  o->nCells = p.nCells;
  o->nDead = p.nDead;
  o->gravity = p.gravity;
  o->spring = p.spring;
  o->damping = p.damping;
  o->shear = p.shear;
  o->attraction = p.attraction;
  o->boundaryDamping = p.boundaryDamping;
  o->friction = p.friction;
  o->massFactor = p.massFactor;
  o->frictionFactor = p.frictionFactor;
  o->radFactor = p.radFactor;
  o->attractionFactor = p.attractionFactor;
  o->constraint = p.constraint;
  o->constraint_contraction = p.constraint_contraction;
  o->centroid_steps = p.centroid_steps;
  o->centroid_int = p.centroid_int;
  o->centroid_radius = p.centroid_radius;
  o->light_x = p.light_x;
  o->light_y = p.light_y;
  o->phase_update_interval = p.phase_update_interval;
  o->control = (int32_t)p.control;
  o->config = (int32_t)p.config;
  o->min_radius = p.min_radius;
  o->max_radius = p.max_radius;
  o->rise_period = p.rise_period;
  o->freq = p.freq;
  o->nobstacles = p.nobstacles;
  o->n_cir_obstacles = p.n_cir_obstacles;
  for (int i = 0; i < PB_MAX_OBSTACLES; i++) {
    if (i < p.nobstacles && i < (int)cfg.x1obs.size()) {
      o->x1obs[i] = cfg.x1obs[i];
      o->x2obs[i] = cfg.x2obs[i];
      o->y1obs[i] = cfg.y1obs[i];
      o->y2obs[i] = cfg.y2obs[i];
    }
    if (i < p.n_cir_obstacles && i < (int)cfg.x_cir_obs.size()) {
      o->x_cir_obs[i] = cfg.x_cir_obs[i];
      o->y_cir_obs[i] = cfg.y_cir_obs[i];
      o->r_cir_obs[i] = cfg.r_cir_obs[i];
    }
  }
  o->Nx = p.Nx;
  o->phase_std = p.phase_std;
  o->seed = p.seed;
  o->light_shadow = p.light_shadow;
  o->testing = p.testing;
  o->constrained_contraction = p.constrained_contraction;
  o->display_shadow = p.display_shadow;
  o->time_to_dead = p.time_to_dead;
  o->max_time = p.max_time;
  o->timestep = cfg.timestep;
  o->sort_interval = cfg.sort_interval;
  o->dump_interval = cfg.dump_interval;
  o->camera_x = cfg.camera_x;
  o->camera_y = cfg.camera_y;
  o->light_radius = cfg.light_radius;
  o->display_interval = cfg.display_interval;
  o->video_interval = cfg.video_interval;
  snprintf(o->csv_filename, sizeof(o->csv_filename), "%s", cfg.csv_filename.c_str());
  snprintf(o->video_filename, sizeof(o->video_filename), "%s", cfg.video_filename.c_str());
  o->wallHalf = cfg.wallHalf();
  o->rngKind = cfg.rng_kind;
}

struct HostSim {
  PbRunConfig cfg;
  Particlebot *bot = nullptr;
};

}  // namespace

extern "C" {

// Resolve a configuration exactly as main() does (defaults, file, overrides, derived values).
// cfg_path may be NULL (defaults only).  Returns 0, or -1 if the file cannot be opened.
int pbHostLoadConfig(const char *cfg_path, const char *overrides, pbFlatConfig *out) {
  PbRunConfig cfg;
  cfg.params.seed = 0;
  if (cfg_path && !cfg.loadFile(cfg_path)) return -1;
  applyOverrides(cfg, overrides);
  cfg.derive();
  flatten(cfg, out);
  return 0;
}

// main.cpp:913-952 without GL: load, srand(seed), construct.  engine: 0 fused, 1 legacy.
void *pbHostCreate(const char *cfg_path, const char *overrides, int engine) {
  HostSim *h = new HostSim();
  h->cfg.params.seed = 0;
  if (cfg_path && !h->cfg.loadFile(cfg_path)) {
    delete h;
    return nullptr;
  }
  applyOverrides(h->cfg, overrides);
  h->cfg.derive();
  srand(h->cfg.params.seed);  // main.cpp:929
  // engine: 0 fused, 1 legacy, 2 host only (placement and draws without any device: CPU tests)
  h->bot = new Particlebot(h->cfg.params,
                           engine == 2   ? Particlebot::Engine::HostOnly
                           : engine == 1 ? Particlebot::Engine::Legacy
                                         : Particlebot::Engine::Fused,
                           h->cfg.wallHalf());
  h->bot->setExitOnMaxTime(false);
  h->bot->setHexSpacing(h->cfg.hex_spacing);
  h->bot->setSquareLattice(h->cfg.square_lattice);
  h->bot->setFastBlob(h->cfg.fast_blob);
  h->bot->setRng(h->cfg.rng_kind);
  return h;
}

void pbHostDestroy(void *hv) {
  HostSim *h = (HostSim *)hv;
  if (!h) return;
  delete h->bot;
  delete h;
}

void pbHostReset(void *hv) { ((HostSim *)hv)->bot->reset(); }

void pbHostUpdate(void *hv) {
  HostSim *h = (HostSim *)hv;
  h->bot->update(h->cfg.timestep, h->cfg.sort_interval);
}

int pbHostAdvance(void *hv, int nsteps) {
  HostSim *h = (HostSim *)hv;
  return h->bot->advance(h->cfg.timestep, h->cfg.sort_interval, nsteps);
}

// steps that can run before the next dump row (or the end of the run) is due; >= 1
int pbHostStepsUntilDump(void *hv, int maxSteps) {
  HostSim *h = (HostSim *)hv;
  return h->bot->stepsUntilHostEvent(h->cfg.timestep, h->cfg.dump_interval, maxSteps);
}

float pbHostTime(void *hv) { return ((HostSim *)hv)->bot->getTime(); }
int pbHostFinished(void *hv) { return ((HostSim *)hv)->bot->finished() ? 1 : 0; }

// display()'s `dumpParticlebot(0, nCells, fp, dump_interval, testing, light)` (main.cpp:360)
int pbHostDump(void *hv, const char *path, const char *mode) {
  HostSim *h = (HostSim *)hv;
  FILE *fp = fopen(path, mode);
  if (!fp) return -1;
  const SimParams &p = h->bot->getParams();
  h->bot->dumpParticlebot(0, p.nCells, fp, h->cfg.dump_interval, p.testing, p.light_x, p.light_y);
  fclose(fp);
  return 0;
}

int pbHostLoadFromFile(void *hv, const char *path) {
  HostSim *h = (HostSim *)hv;
  FILE *fp = fopen(path, "r");
  if (!fp) return -1;
  h->bot->loadFromFile(0, h->bot->getParams().nCells, fp, h->cfg.dump_interval);
  fclose(fp);
  return 0;
}

// half <= 0 selects the reference's camera: centred on (camera_x, 0), half extent camera_y * tan(30 deg)
int pbHostWriteFrame(void *hv, const char *path, int width, int height, float cx, float cy, float half) {
  HostSim *h = (HostSim *)hv;
  if (!(half > 0)) {
    cx = h->cfg.camera_x;
    cy = 0.0f;
    half = h->cfg.camera_y * 0.57735027f;
  }
  return h->bot->writeFramePPM(path, width, height, cx, cy, half, h->cfg.light_radius) ? 0 : -1;
}

int pbHostSaveCheckpoint(void *hv, const char *path) {
  FILE *fp = fopen(path, "wb");
  if (!fp) return -1;
  const bool ok = ((HostSim *)hv)->bot->saveCheckpoint(fp);
  fclose(fp);
  return ok ? 0 : -2;
}

int pbHostLoadCheckpoint(void *hv, const char *path) {
  FILE *fp = fopen(path, "rb");
  if (!fp) return -1;
  const bool ok = ((HostSim *)hv)->bot->loadCheckpoint(fp);
  fclose(fp);
  return ok ? 0 : -2;
}

// draws the dead set now (what update() does at time_to_dead) and returns it; host mirrors only
int pbHostDrawDead(void *hv, int *out) {
  HostSim *h = (HostSim *)hv;
  const int *d = h->bot->drawDeadBotsNow();
  memcpy(out, d, sizeof(int) * h->bot->getParams().nCells);
  return 0;
}

// which: 0 POSITION (2n floats) 1 VELOCITY (2n) 2 RADII (n) 3 PHASE (n) 5 DEAD (n ints)
int pbHostGetArray(void *hv, int which, void *out) {
  HostSim *h = (HostSim *)hv;
  const size_t n = h->bot->getParams().nCells;
  switch (which) {
    case 0: memcpy(out, h->bot->getArray(POSITION), 8 * n); return 0;
    case 1: memcpy(out, h->bot->getArray(VELOCITY), 8 * n); return 0;
    case 2: memcpy(out, h->bot->getArray(RADII), 4 * n); return 0;
    case 3: memcpy(out, h->bot->getArray(PHASE), 4 * n); return 0;
    case 5: memcpy(out, h->bot->getDeadArray(), 4 * n); return 0;
    default: return -1;
  }
}

int pbHostSetArray(void *hv, int which, const float *data, int start, int count) {
  HostSim *h = (HostSim *)hv;
  if (which < 0 || which > 4) return -1;
  h->bot->setArray((ParticlebotArray)which, data, start, count);
  return 0;
}

// n draws of the class's glibc-compatible generator after seeding (for the CPU test against rand())
void pbHostLibcRandDraws(unsigned seed, int n, int *out) {
  PbLibcRand g(seed);
  for (int i = 0; i < n; i++) out[i] = g.next();
}

// ---- ensembles: many independent simulations in one batched pbSim -----------------------------
// Member k = the base .cfg + common overrides + its own overrides (typically "seed\n<k>").  The
// host work of every member (random placement, dead-bot draw) runs in its own HostOnly Particlebot
// with its own private libc-compatible stream; the device work of all members runs in ONE batched
// pbSim, one launch per timestep.  Summaries (time, COMx, COMy, distance of COM to the light) are
// taken whenever a dump row would be due.
struct Ensemble {
  std::vector<PbRunConfig *> cfgs;
  std::vector<Particlebot *> bots;
  pbSim *sim = nullptr;
  bool haveRow = false;  // pbEnsembleRunSteps: a summary row has been written at time rowTime
  float rowTime = 0.0f;
  ~Ensemble() {
    if (sim) pbSimDestroy(sim);
    for (auto *b : bots) delete b;
    for (auto *c : cfgs) delete c;
  }
};

void *pbEnsembleCreate(const char *cfg_path, const char *common_overrides, const char **member_overrides,
                       int nmembers) {
  if (nmembers < 1) return nullptr;
  Ensemble *e = new Ensemble();
  e->cfgs.assign(nmembers, nullptr);
  e->bots.assign(nmembers, nullptr);
  // Members are independent (own configuration, own private random stream, own placement grid):
  // build them on all host cores.  The reference's random placement is O(N^1.5) (2.7 s for 10^5
  // bots), so a sweep of large members would otherwise spend minutes here.
  std::atomic<int> next{0};
  std::atomic<bool> failed{false};
  auto worker = [&]() {
    for (int k = next++; k < nmembers && !failed; k = next++) {
      PbRunConfig *cfg = new PbRunConfig();
      cfg->params.seed = 0;
      e->cfgs[k] = cfg;
      if (cfg_path && !cfg->loadFile(cfg_path)) {
        failed = true;
        return;
      }
      applyOverrides(*cfg, common_overrides);
      if (member_overrides) applyOverrides(*cfg, member_overrides[k]);
      cfg->derive();
      Particlebot *bot = new Particlebot(cfg->params, Particlebot::Engine::HostOnly, cfg->wallHalf());
      bot->setHexSpacing(cfg->hex_spacing);
      bot->setSquareLattice(cfg->square_lattice);
      bot->setFastBlob(cfg->fast_blob);
      bot->setRng(cfg->rng_kind);
      bot->reset();
      e->bots[k] = bot;
    }
  };
  unsigned nthreads = std::thread::hardware_concurrency();
  // one process per GPU: share the host's cores between the ranks of this node
  for (const char *name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE"})
    if (const char *v = getenv(name)) {
      if (atoi(v) > 1) nthreads = std::max(1u, nthreads / (unsigned)atoi(v));
      break;
    }
  if (const char *v = getenv("PB_HOST_THREADS")) nthreads = (unsigned)atoi(v);
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) nthreads = std::min<unsigned>(nthreads, (unsigned)CPU_COUNT(&set));
  nthreads = std::max(1u, std::min<unsigned>(std::min(nthreads, 128u), (unsigned)nmembers));
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nthreads; t++) pool.emplace_back(worker);
  worker();
  for (auto &th : pool) th.join();
  if (failed) {
    delete e;
    return nullptr;
  }
  std::vector<SimParams> params;
  for (int k = 0; k < nmembers; k++) params.push_back(e->bots[k]->getParams());
  if (pbSimCreateBatch(&e->sim, params.data(), nmembers, e->cfgs[0]->wallHalf()) != PB_OK) {
    fprintf(stderr, "pbEnsembleCreate: %s\n", pbGetLastErrorString());
    delete e;
    return nullptr;
  }
  if (e->cfgs[0]->rng_kind != 0 && pbSimSetRng(e->sim, e->cfgs[0]->rng_kind) != PB_OK) {
    fprintf(stderr, "pbEnsembleCreate: %s\n", pbGetLastErrorString());
    delete e;
    return nullptr;
  }
  for (int k = 0; k < nmembers; k++) {
    Particlebot *b = e->bots[k];
    if (pbSimSetStateOf(e->sim, (unsigned)k, b->hostPositions(), b->hostVelocities(), b->hostRadii(),
                        b->hostPhases(), b->hostDead()) != PB_OK) {
      fprintf(stderr, "pbEnsembleCreate: %s\n", pbGetLastErrorString());
      delete e;
      return nullptr;
    }
  }
  return e;
}

void pbEnsembleDestroy(void *ev) { delete (Ensemble *)ev; }

// Runs every member for up to max_steps timesteps (or to max_time, whichever comes first) and can be
// called again to continue.  out: [nmembers][max_rows][4] floats (time, COMx, COMy, distance of the
// COM to the light), one row whenever a dump row would be due (particlebot.cpp:309); *rows counts
// the rows written per member so far (the same for all members) and is carried between calls.
// Returns the number of timesteps executed by this call, or -1 on error.
long pbEnsembleRunSteps(void *ev, long max_steps, float *out, int max_rows, int *rows) {
  Ensemble *e = (Ensemble *)ev;
  const int m = (int)e->bots.size();
  const PbRunConfig &c0 = *e->cfgs[0];
  const float dt = c0.timestep, di = c0.dump_interval;
  std::vector<double> com(2 * (size_t)m);
  long steps = 0;
  int nrows = rows ? *rows : 0;
  float t = 0.0f;
  if (pbSimGetTime(e->sim, &t) != PB_OK) return -1;
  for (;;) {
    // a row is due at time t; e->rowTime remembers the last one written so that a call which stopped
    // exactly at a dump time does not write it twice when the run is continued
    if (out && !(t - di * floorf(t / di) > 0.01f) && nrows < max_rows && !(e->haveRow && e->rowTime == t)) {
      if (pbSimCentroids(e->sim, com.data()) != PB_OK) return -1;
      for (int k = 0; k < m; k++) {
        const SimParams &p = e->bots[k]->getParams();
        float *row = out + ((size_t)k * max_rows + nrows) * 4;
        const double dx = com[2 * k] - p.light_x, dy = com[2 * k + 1] - p.light_y;
        row[0] = t;
        row[1] = (float)com[2 * k];
        row[2] = (float)com[2 * k + 1];
        row[3] = (float)sqrt(dx * dx + dy * dy);
      }
      nrows++;
      e->haveRow = true;
      e->rowTime = t;
    }
    if (t > c0.params.max_time || steps >= max_steps) break;
    // host events at this step: dead-bot draws
    for (int k = 0; k < m; k++) {
      Particlebot *b = e->bots[k];
      b->setHostTime(t);
      if (b->deadDrawDue(dt)) {
        if (pbSimSetStateOf(e->sim, (unsigned)k, nullptr, nullptr, nullptr, nullptr, b->drawDeadBotsNow()) != PB_OK)
          return -1;
      }
    }
    // run up to (not past) the next dump row or dead-bot draw of any member
    long run = 1;
    float tt = t + dt;
    for (;;) {
      bool stop = !(tt - di * floorf(tt / di) > 0.01f) || tt > c0.params.max_time || run >= (1 << 20) ||
                  steps + run >= max_steps;
      for (int k = 0; k < m && !stop; k++) {
        e->bots[k]->setHostTime(tt);
        stop = e->bots[k]->deadDrawDue(dt);
      }
      if (stop) break;
      tt = tt + dt;
      run++;
    }
    int done = 0;
    if (pbSimStep(e->sim, dt, c0.sort_interval, (int)run, &done) != PB_OK) return -1;
    steps += done;
    if (pbSimGetTime(e->sim, &t) != PB_OK) return -1;
    if (done == 0) break;
  }
  if (rows) *rows = nrows;
  return steps;
}

// Runs every member to max_time (the whole run in one call).
long pbEnsembleRun(void *ev, float *out, int max_rows, int *rows) {
  int nrows = 0;
  const long steps = pbEnsembleRunSteps(ev, LONG_MAX, out, max_rows, &nrows);
  if (rows) *rows = nrows;
  return steps;
}

int pbEnsembleSynchronize(void *ev) { return pbSimSynchronize(((Ensemble *)ev)->sim); }

int pbEnsembleShard(int nmembers, int rank, int world) {
  if (nmembers < 0 || world < 1 || rank < 0 || rank >= world) return 0;
  return (nmembers - rank + world - 1) / world;  // members rank, rank + world, ... below nmembers
}

int pbEnsembleAssemble(int nmembers, int world, int rows, const float *gathered, float *out) {
  if (nmembers < 0 || world < 1 || rows < 0 || !gathered || !out) return 1;
  const int per = pbEnsembleShard(nmembers, 0, world);
  const size_t rowFloats = (size_t)rows * 4;
  for (int r = 0; r < world; r++) {
    const int mine = pbEnsembleShard(nmembers, r, world);
    for (int j = 0; j < mine; j++)
      memcpy(out + (size_t)(r + j * world) * rowFloats, gathered + ((size_t)r * per + j) * rowFloats,
             sizeof(float) * rowFloats);
  }
  return 0;
}

int pbEnsembleGetState(void *ev, int member, float *pos, float *vel, float *rad) {
  Ensemble *e = (Ensemble *)ev;
  return pbSimGetStateOf(e->sim, (unsigned)member, pos, vel, rad, nullptr, nullptr, nullptr, nullptr);
}

unsigned pbEnsembleNumBots(void *ev) { return ((Ensemble *)ev)->bots[0]->getParams().nCells; }

// ---- the libm properties the phase update rests on (pbSimSetMinDistanceMode 0) ---------------------
// The engine returns min_i (dx*dx + dy*dy) from the device and takes powf(., 0.5f) on the host, where the
// reference takes min_i powf(powf(dx,2) + powf(dy,2), 0.5f) (particlebot.cpp:215-228).  The two agree for every
// input iff, on THIS host's libm, powf(x, 2.0f) == x*x bit for bit for every float and powf(., 0.5f) is
// non-decreasing over the non-negative floats.  Checked exhaustively: every non-negative float (the square also
// for every 16th negative one; 2^31 values, ~15 s on 8 cores).  stride > 1 samples every stride-th float.
// Returns 0 when both hold.
int pbHostLibmCheck(int threads, unsigned stride, unsigned long long *checked, unsigned long long *square_mismatches,
                    unsigned long long *root_inversions) {
  if (threads < 1) threads = (int)std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) threads = std::min(threads, CPU_COUNT(&set));
  threads = std::max(1, std::min(threads, 256));
  if (stride < 1) stride = 1;
  const uint64_t last = 0x7F800000ull;  // +inf included
  std::vector<unsigned long long> bad(threads, 0), inv(threads, 0), seen(threads, 0);
  auto work = [&](int t) {
    const uint64_t lo = last * (uint64_t)t / (uint64_t)threads, hi = last * (uint64_t)(t + 1) / (uint64_t)threads;
    // (each chunk starts one float early so that the monotonicity test spans the chunk boundaries)
    uint32_t b0 = (uint32_t)(lo == 0 ? 0 : lo - 1);
    float x0;
    memcpy(&x0, &b0, 4);
    float prev = powf(x0, 0.5f);
    const uint64_t end = (t == threads - 1) ? hi + 1 : hi;  // (+inf belongs to the last chunk)
    for (uint64_t b = lo; b < end; b += stride) {
      const uint32_t bits = (uint32_t)b;
      float x;
      memcpy(&x, &bits, 4);
      const float sq = powf(x, 2.0f), mul = x * x;
      if (memcmp(&sq, &mul, 4) != 0) bad[t]++;
      if ((bits & 15u) == 0u) {  // the negative argument: every 16th float
        const float sqn = powf(-x, 2.0f);
        if (memcmp(&sqn, &mul, 4) != 0) bad[t]++;
      }
      const float r = powf(x, 0.5f);
      if (r < prev) inv[t]++;
      prev = r;
      seen[t]++;
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  unsigned long long b = 0, i = 0, n = 0;
  for (int t = 0; t < threads; t++) b += bad[t], i += inv[t], n += seen[t];
  if (checked) *checked = n;
  if (square_mismatches) *square_mismatches = b;
  if (root_inversions) *root_inversions = i;
  return (b == 0 && i == 0) ? 0 : 1;
}

const char *pbHostLibcVersion(void) { return gnu_get_libc_version(); }

// ---- csrc/pb_xorwow.hpp on the host (CPU tests: the same code the kernels run) -------------------
static const uint32_t *hostJumpTable() {
  static std::vector<uint32_t> table;
  static std::once_flag once;
  std::call_once(once, [] {
    table.resize(PB_XW_TABLE_WORDS);
    pbXorwowBuildJumpTable(table.data());
  });
  return table.data();
}

// the first `count` raw outputs of curand_init(seed, subsequence, 0)
void pbHostXorwowOutputs(int kind, unsigned long long seed, unsigned subsequence, unsigned count, unsigned *out) {
  pbRngState s;
  pbXorwowSeed(s, seed, kind);
  pbXorwowSkipSubsequences(s, subsequence, hostJumpTable());
  for (unsigned i = 0; i < count; i++) out[i] = pbXorwowNext(s);
}

// `draws` normals of each of the bots 0..nbots-1 (out[draw][bot]), as add_normal_noise consumes them
void pbHostXorwowNormals(int kind, unsigned seed, unsigned nbots, unsigned draws, float *out) {
  for (unsigned i = 0; i < nbots; i++) {
    pbRngState s;
    pbXorwowSeed(s, (uint64_t)seed, kind);
    pbXorwowSkipSubsequences(s, i, hostJumpTable());
    for (unsigned k = 0; k < draws; k++) out[(size_t)k * nbots + i] = pbXorwowNormal(s);
  }
}

// table[k] (160 x 5 words), k = 0..31
void pbHostXorwowJumpMatrix(unsigned k, unsigned *rows) {
  memcpy(rows, hostJumpTable() + (size_t)(k & 31u) * PB_XW_MAT_WORDS, sizeof(uint32_t) * PB_XW_MAT_WORDS);
}

// the fused engine behind a HostSim (NULL for the other engines): lets a script reach pbSim* calls the
// class does not wrap
void *pbHostEngineHandle(void *hv) { return ((HostSim *)hv)->bot->engineHandle(); }

unsigned pbHostNumBots(void *hv) { return ((HostSim *)hv)->bot->getParams().nCells; }

}  // extern "C"
