// pb_capi.cpp -- C wrappers around the C++ host side (class Particlebot + .cfg loader) so that
// scripts and tests can drive it through ctypes.  Exported from libparticlebot_host.so.
#include <gnu/libc-version.h>
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <map>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <ctime>
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
  int forceVariant;  // pb_force_variant (-1: the engine's default)
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
  o->forceVariant = cfg.force_variant;
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
  h->bot->setForceVariant(h->cfg.force_variant);
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
// pbSim, one launch per timestep.  Summaries (time, COMx, COMy, distance of the COM to the light) are
// taken whenever a dump row would be due.
}  // extern "C"

namespace {

// one member: its resolved configuration and the HostOnly object that places it and draws its dead set
// what an ensemble checkpoint holds of one member beyond the host mirrors (pbEnsemblePipelineSetCheckpoint)
struct MemberSaved {
  float time = 0.0f;
  unsigned draws = 0;
  int sorted = 0, finished = 0, nrows = 0;
  long steps = 0;
  std::vector<float> rows, absA, absR;
  std::vector<unsigned> orig, keys;
};

struct Member {
  PbRunConfig *cfg = nullptr;
  Particlebot *bot = nullptr;
  bool deadDrawn = false;  // the dead set was drawn with the placement (a draw due at time 0)
  MemberSaved *saved = nullptr;  // restored from a checkpoint instead of placed
  ~Member() {
    delete saved;
    delete bot;
    delete cfg;
  }
};

// Host side of one member: configuration, placement (Particlebot::reset) and -- when the draw is due at the very
// first step -- the dead set, all from the member's PRIVATE random stream, so that it does not matter which thread
// builds which member, or when.  Returns false if the .cfg cannot be read.
bool configureMember(Member &m, const char *cfg_path, const char *common_overrides, const char *own_overrides) {
  m.cfg = new PbRunConfig();
  m.cfg->params.seed = 0;
  if (cfg_path && !m.cfg->loadFile(cfg_path)) return false;
  applyOverrides(*m.cfg, common_overrides);
  applyOverrides(*m.cfg, own_overrides);
  m.cfg->derive();
  Particlebot *bot = new Particlebot(m.cfg->params, Particlebot::Engine::HostOnly, m.cfg->wallHalf());
  bot->setHexSpacing(m.cfg->hex_spacing);
  bot->setSquareLattice(m.cfg->square_lattice);
  bot->setFastBlob(m.cfg->fast_blob);
  bot->setRng(m.cfg->rng_kind);
  m.bot = bot;
  return true;
}

// `shared`: a placement another member with the same Particlebot::placementKey() produced (installed instead of
// reset(): same positions, same generator state after the placement draws); `out`: capture this member's own.
bool buildMember(Member &m, const char *cfg_path, const char *common_overrides, const char *own_overrides,
                 const Particlebot::Placement *shared = nullptr, Particlebot::Placement *out = nullptr) {
  if (!configureMember(m, cfg_path, common_overrides, own_overrides)) return false;
  Particlebot *bot = m.bot;
  if (shared) {
    if (!bot->importPlacement(*shared)) return false;
  } else {
    bot->reset();
    if (out) bot->exportPlacement(*out);
  }
  bot->setHostTime(0.0f);
  if (bot->deadDrawDue(m.cfg->timestep)) {  // particlebot.cpp:178: drawn at the top of the first update()
    (void)bot->drawDeadBotsNow();
    m.deadDrawn = true;
  }
  m.bot = bot;
  return true;
}

// ---- host resources of a rank (include/particlebot_ensemble.h "host resources") ---------------------------------
std::string envOr(const char *name, const char *fallback) {
  const char *v = getenv(name);
  return v && v[0] ? v : fallback;
}

bool readLine(const std::string &path, std::string &out) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[4096];
  const bool ok = fgets(buf, sizeof buf, f) != nullptr;
  fclose(f);
  if (!ok) return false;
  out = buf;
  while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
  return true;
}

// CPUs the cgroup CPU controller grants: the tightest quota / period on the way from the process's own cgroup up to
// the mount point (v2: cpu.max "quota period" or "max period"; v1: cpu.cfs_quota_us, -1 = unlimited).  <= 0: unlimited.
double cgroupCpus() {
  const std::string root = envOr("PB_CGROUP_ROOT", "/sys/fs/cgroup");
  double best = 0.0;
  auto take = [&](double cpus) {
    if (cpus > 0.0 && (best <= 0.0 || cpus < best)) best = cpus;
  };
  // the process's cgroup path: "0::/a/b" (v2) -- inside a container's cgroup namespace this is "/"
  std::string rel = "/";
  if (FILE *f = fopen(envOr("PB_PROC_SELF_CGROUP", "/proc/self/cgroup").c_str(), "r")) {
    char buf[4096];
    while (fgets(buf, sizeof buf, f)) {
      if (strncmp(buf, "0::", 3) == 0) {
        rel = buf + 3;
        while (!rel.empty() && (rel.back() == '\n' || rel.back() == ' ')) rel.pop_back();
        break;
      }
    }
    fclose(f);
  }
  if (rel.empty() || rel[0] != '/' || rel.find("..") != std::string::npos) rel = "/";
  for (std::string dir = rel;;) {
    std::string line;
    if (readLine(root + dir + (dir.back() == '/' ? "" : "/") + "cpu.max", line)) {
      char q[64] = {0};
      double period = 0.0;
      if (sscanf(line.c_str(), "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0.0) take(atof(q) / period);
    }
    if (dir == "/" || dir.empty()) break;
    const size_t cut = dir.find_last_of('/');
    dir = cut == 0 ? "/" : dir.substr(0, cut);
  }
  std::string q, per;  // cgroup v1
  if (readLine(root + "/cpu/cpu.cfs_quota_us", q) && readLine(root + "/cpu/cpu.cfs_period_us", per) && atof(q.c_str()) > 0 &&
      atof(per.c_str()) > 0)
    take(atof(q.c_str()) / atof(per.c_str()));
  return best;
}

int affinityCpus(cpu_set_t *setOut) {
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) != 0) return 0;
  if (setOut) *setOut = set;
  return CPU_COUNT(&set);
}

int localWorldSize() {
  for (const char *name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE"})
    if (const char *v = getenv(name)) return std::max(1, atoi(v));
  return 1;
}

// "0-3,8,10-11" -> the listed cores that are also in the affinity mask
std::vector<int> parseCpuList(const char *text) {
  std::vector<int> cpus;
  cpu_set_t aff;
  const bool haveAff = affinityCpus(&aff) > 0;
  for (const char *p = text; p && *p;) {
    while (*p == ',' || *p == ' ' || *p == '\n') p++;
    if (!*p) break;
    char *end = nullptr;
    const long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    p = end;
    if (*p == '-') {
      b = strtol(p + 1, &end, 10);
      if (end == p + 1) break;
      p = end;
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++)
      if (c >= 0 && (!haveAff || CPU_ISSET((int)c, &aff))) cpus.push_back((int)c);
  }
  return cpus;
}

// the cores next to a device: /sys/bus/pci/devices/<bus id>/numa_node (>= 0 on a NUMA machine) + local_cpulist
int numaOfDevice(int device, std::string &busId, std::vector<int> &cpus) {
  cpus.clear();
  busId.clear();
  if (device < 0) return -1;
  char id[32] = {0};
  if (const char *fake = getenv("PB_FAKE_PCI_BUS_ID")) {  // CPU tests: no device to ask
    snprintf(id, sizeof id, "%s", fake);
  } else if (pbDevicePciBusId(device, id, (int)sizeof id) != PB_OK) {
    return -1;
  }
  for (char *c = id; *c; c++) *c = (char)tolower(*c);
  busId = id;
  const std::string dir = envOr("PB_SYSFS_ROOT", "/sys") + "/bus/pci/devices/" + busId;
  std::string node, list;
  if (!readLine(dir + "/numa_node", node)) return -1;
  const int n = atoi(node.c_str());
  if (n < 0) return -1;
  if (readLine(dir + "/local_cpulist", list)) cpus = parseCpuList(list.c_str());
  return n;
}

void describeResources(pbHostResources &r, int wanted) {
  memset(&r, 0, sizeof r);
  r.hardware_threads = (int)std::thread::hardware_concurrency();
  r.affinity_cpus = affinityCpus(nullptr);
  r.cgroup_cpus = cgroupCpus();
  int usable = r.hardware_threads > 0 ? r.hardware_threads : 1;
  if (r.affinity_cpus > 0) usable = std::min(usable, r.affinity_cpus);
  if (r.cgroup_cpus > 0.0) usable = std::min(usable, std::max(1, (int)std::floor(r.cgroup_cpus + 1e-9)));
  r.usable_cpus = std::max(1, usable);
  r.local_world_size = localWorldSize();
  int share = std::max(1, r.usable_cpus / r.local_world_size);
  const char *why = "usable cores / ranks of the node";
  bool automatic = true;  // the share is the rule's, not a number somebody asked for
  if (const char *v = getenv("PB_HOST_THREADS")) {
    if (atoi(v) > 0) {
      share = atoi(v);
      why = "PB_HOST_THREADS";
      automatic = false;
    }
  }
  if (wanted > 0) {
    share = wanted;
    why = "host_threads argument";
    automatic = false;
  }
  r.host_threads = std::max(1, std::min(share, 128));
  r.device = -1;
  r.numa_node = -1;
  int dev = -1;
  if (getenv("PB_FAKE_PCI_BUS_ID")) dev = 0;
  else if (pbGetDevice(&dev) != PB_OK) dev = -1;
  r.device = dev;
  std::string bus;
  std::vector<int> cpus;
  r.numa_node = numaOfDevice(dev, bus, cpus);
  snprintf(r.pci_bus_id, sizeof r.pci_bus_id, "%s", bus.c_str());
  r.numa_cpus = (int)cpus.size();
  const char *pin = getenv("PB_PIN_PRODUCERS");
  r.pin_producers = (r.numa_node >= 0 && r.numa_cpus > 0 && !(pin && pin[0] == '0')) ? 1 : 0;
  // A pinned pool must FIT the node's cores: pinning 127 producers of a lone rank to the 64 (NPS4: 16) cores next to
  // its GPU would oversubscribe them 2-8 x while the rest of the machine idles.  Several ranks per node: the cores
  // beyond the node belong to the other ranks' pools, so the automatic share shrinks to the node; a lone rank, or an
  // explicit thread count, keeps its threads and is not pinned.
  const char *pinNote = r.pin_producers ? "pinned to the GPU's NUMA node" : "not pinned (no NUMA node reported for the device)";
  if (pin && pin[0] == '0' && r.numa_node >= 0) pinNote = "not pinned (PB_PIN_PRODUCERS=0)";
  if (r.pin_producers && r.host_threads > r.numa_cpus) {
    if (automatic && r.local_world_size > 1) {
      r.host_threads = r.numa_cpus;
      pinNote = "pinned to the GPU's NUMA node, share clamped to its cores";
    } else {
      r.pin_producers = 0;
      pinNote = "not pinned (the pool is larger than the GPU's NUMA node)";
    }
  }
  char quota[48];
  if (r.cgroup_cpus > 0.0) snprintf(quota, sizeof quota, "%.2f", r.cgroup_cpus);
  else snprintf(quota, sizeof quota, "none");
  snprintf(r.rule, sizeof r.rule,
           "%d producer threads (%s): min(hardware %d, affinity %d, cgroup quota %s) = %d usable / %d rank(s) per node; "
           "%s",
           r.host_threads, why, r.hardware_threads, r.affinity_cpus, quota, r.usable_cpus, r.local_world_size,
           pinNote);
}

// host threads for placement: this rank's share of the cores the process may really use
unsigned hostThreads(int wanted) {
  pbHostResources r;
  describeResources(r, wanted);
  return (unsigned)r.host_threads;
}

double threadCpuSeconds() {
  timespec ts;
  if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) != 0) return 0.0;
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// ---- ensemble checkpoints (pbEnsemblePipelineSetCheckpoint) ------------------------------------------------------
// DIR/sub_<b>.manifest  "generation rows finished steps" of sub-batch b, written (tmp + rename) AFTER the member files
//                       of that generation are complete: the members of one sub-batch share a clock, so a
//                       checkpoint is only usable when all of them are from the same row
// DIR/member_<k>.<generation>  header, the member's summary rows so far and -- unless it has finished -- every state
//                       array, the stale slot layout and both generators (the exact checkpoint of class Particlebot,
//                       per member of a batch).  Generations alternate 0/1 so that the previous complete one survives
//                       a kill in the middle of writing the next.
struct MemberFileHeader {
  char magic[8];
  uint32_t nbots;
  float time;
  uint32_t draws;
  int32_t rngKind, sorted, deadDrawn, nrows, finished;
  int32_t rs[36];
};
const char kMemberMagic[8] = {'P', 'B', 'E', 'N', 'S', 'M', '1', 0};

std::string memberPath(const std::string &dir, int k, int gen) {
  char name[64];
  snprintf(name, sizeof name, "/member_%06d.%d", k, gen);
  return dir + name;
}
std::string manifestPath(const std::string &dir, int sub) {
  char name[64];
  snprintf(name, sizeof name, "/sub_%06d.manifest", sub);
  return dir + name;
}

template <class T>
bool putv(FILE *fp, const T *p, size_t count) { return fwrite(p, sizeof(T), count, fp) == count; }
template <class T>
bool getv(FILE *fp, T *p, size_t count) { return fread(p, sizeof(T), count, fp) == count; }

bool readManifest(const std::string &dir, int sub, int &gen, int &nrows, int &finished, long &steps) {
  FILE *f = fopen(manifestPath(dir, sub).c_str(), "r");
  if (!f) return false;
  const bool ok = fscanf(f, "%d %d %d %ld", &gen, &nrows, &finished, &steps) == 4 && (gen == 0 || gen == 1) &&
                  nrows >= 0 && nrows <= (1 << 24) && steps >= 0;
  fclose(f);
  return ok;
}

// the member's file of generation gen -> m.saved (+ host mirrors, generator, dead-draw flag); false if unusable
bool loadMemberFile(Member &m, const std::string &dir, int k, int gen, int wantRows, long steps) {
  FILE *f = fopen(memberPath(dir, k, gen).c_str(), "rb");
  if (!f) return false;
  MemberFileHeader h;
  const size_t n = m.bot->getParams().nCells;
  MemberSaved *sv = new MemberSaved();
  bool ok = getv(f, &h, 1) && memcmp(h.magic, kMemberMagic, 8) == 0 && h.nbots == n && h.nrows == wantRows &&
            h.rngKind == m.cfg->rng_kind;
  if (ok) {
    sv->rows.resize((size_t)h.nrows * 4);
    ok = getv(f, sv->rows.data(), sv->rows.size());
  }
  if (ok && !h.finished) {
    std::vector<float> pos(2 * n), vel(2 * n), rad(n), phase(n);
    std::vector<int> dead(n);
    sv->absA.resize(n), sv->absR.resize(n), sv->orig.resize(n), sv->keys.resize(n);
    ok = getv(f, pos.data(), 2 * n) && getv(f, vel.data(), 2 * n) && getv(f, rad.data(), n) &&
         getv(f, phase.data(), n) && getv(f, dead.data(), n) && getv(f, sv->absA.data(), n) &&
         getv(f, sv->absR.data(), n) && getv(f, sv->orig.data(), n) && getv(f, sv->keys.data(), n);
    if (ok) m.bot->restoreHostMirrors(pos.data(), vel.data(), rad.data(), phase.data(), dead.data());
  }
  fclose(f);
  if (!ok) {
    delete sv;
    return false;
  }
  sv->time = h.time, sv->draws = h.draws, sv->sorted = h.sorted, sv->finished = h.finished, sv->nrows = h.nrows;
  sv->steps = steps;
  m.bot->setHostRngState(h.rs);
  m.bot->setHostTime(h.time);
  m.deadDrawn = h.deadDrawn != 0;
  m.saved = sv;
  return true;
}

// A batch of members on the device: ONE pbSim, one launch per timestep.
struct Ensemble {
  std::vector<Member *> members;  // owned
  pbSim *sim = nullptr;
  bool haveRow = false;  // runSteps: a summary row has been written at time rowTime
  float rowTime = 0.0f;
  // checkpointing (pipeline): directory, this sub-batch's number and first member, steps done before this call
  std::string ckptDir;
  int ckptSub = 0, ckptFirst = 0, ckptGen = 0;
  long stepsBefore = 0;
  // per-member CSV files in the reference's own format (pbEnsemblePipelineSetCsvDir): directory, the members' numbers
  // in the whole ensemble, the open files
  std::string csvDir;
  std::vector<int> csvIds;
  std::vector<FILE *> csvFiles;
  ~Ensemble() {
    for (FILE *f : csvFiles)
      if (f) fclose(f);
    if (sim) pbSimDestroy(sim);
    for (auto *m : members) delete m;
  }
};

// One row of the reference's CSV with testing = 0 (particlebot.cpp:303-367: "Seed", the header and the time-0 row come
// together), from the reference's own fp32 centroid sums -- the text particlebot_run writes for the member run alone.
void writeCsvRow(FILE *fp, float time, unsigned seed, float sumX, float sumY, unsigned count, float light_x, float light_y) {
  if (time == 0) {
    fprintf(fp, "Seed, %u\n", seed);
    fprintf(fp, "Time,");
    fprintf(fp, "Centroid X, Centroid Y, Distance");
    fprintf(fp, "\n");
  }
  fprintf(fp, "%f,", time);
  fprintf(fp, "%f, %f, %f,", sumX / (float)count, sumY / (float)count,
          powf(powf(sumX / (float)count - light_x, 2.0) + powf(sumY / (float)count - light_y, 2.0), 0.5));
  fprintf(fp, "\n");
}

// device side: create the batched pbSim of already built members and upload their initial state
bool uploadEnsemble(Ensemble *e) {
  const int nmembers = (int)e->members.size();
  std::vector<SimParams> params;
  for (int k = 0; k < nmembers; k++) params.push_back(e->members[k]->bot->getParams());
  const PbRunConfig &c0 = *e->members[0]->cfg;
  // the force kernel and the phase-noise generator are chosen per BATCH: a member that asks for another one than
  // member 0 (a per-member override, `--sweep pb_force_variant 2 3`) would silently get member 0's -- refuse
  for (int k = 1; k < nmembers; k++) {
    const PbRunConfig &ck = *e->members[k]->cfg;
    if (ck.force_variant != c0.force_variant || ck.rng_kind != c0.rng_kind) {
      char msg[256];
      snprintf(msg, sizeof msg,
               "members of one batch must agree on pb_force_variant and pb_rng: member 0 has %d / %d, member %d has "
               "%d / %d (run them as separate ensembles)",
               c0.force_variant, c0.rng_kind, k, ck.force_variant, ck.rng_kind);
      fprintf(stderr, "pbEnsemble: %s\n", msg);
      return false;
    }
  }
  if (pbSimCreateBatch(&e->sim, params.data(), nmembers, c0.wallHalf()) != PB_OK) return false;
  if (c0.rng_kind != 0 && pbSimSetRng(e->sim, c0.rng_kind) != PB_OK) return false;
  if (c0.force_variant >= 0 && pbSimSetForceVariant(e->sim, c0.force_variant) != PB_OK) return false;  // pb_force_variant
  for (int k = 0; k < nmembers; k++) {
    Particlebot *b = e->members[k]->bot;
    if (pbSimSetStateOf(e->sim, (unsigned)k, b->hostPositions(), b->hostVelocities(), b->hostRadii(), b->hostPhases(),
                        b->hostDead()) != PB_OK)
      return false;
  }
  return true;
}

// writes generation (gen ^ 1) of every member of the batch (state as of now, `nrows` rows each), then the manifest
bool saveSubBatch(Ensemble *e, const float *out, int max_rows, int nrows, long steps, bool finished) {
  const int m = (int)e->members.size();
  const int gen = e->ckptGen ^ 1;
  const size_t n = e->members[0]->bot->getParams().nCells;
  float t = 0.0f;
  unsigned draws = 0;
  if (pbSimGetTime(e->sim, &t) != PB_OK || pbSimGetPhaseDraws(e->sim, &draws) != PB_OK) return false;
  std::vector<float> pos(2 * n), vel(2 * n), rad(n), phase(n), absA(n), absR(n);
  std::vector<int> dead(n);
  std::vector<unsigned> orig(n), keys(n);
  pbSimConfig conf;
  if (pbSimGetConfig(e->sim, &conf) != PB_OK) return false;
  for (int k = 0; k < m; k++) {
    MemberFileHeader h;
    memcpy(h.magic, kMemberMagic, 8);
    h.nbots = (uint32_t)n, h.time = t, h.draws = draws, h.rngKind = e->members[k]->cfg->rng_kind;
    h.deadDrawn = e->members[k]->deadDrawn ? 1 : 0, h.nrows = nrows, h.finished = finished ? 1 : 0;
    int sorted = 0;
    if (!finished) {
      if (pbSimGetStateOf(e->sim, (unsigned)k, pos.data(), vel.data(), rad.data(), phase.data(), dead.data(), absA.data(),
                          absR.data()) != PB_OK ||
          pbSimGetLayoutOf(e->sim, (unsigned)k, orig.data(), keys.data(), &sorted) != PB_OK)
        return false;
      if (!conf.attraction_sums) std::fill(absA.begin(), absA.end(), 0.0f);  // (not maintained: never NaN on disk)
    }
    h.sorted = sorted;
    e->members[k]->bot->getHostRngState(h.rs);
    const std::string dest = memberPath(e->ckptDir, e->ckptFirst + k, gen), tmp = dest + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    bool ok = f && putv(f, &h, 1) && putv(f, out + (size_t)k * max_rows * 4, (size_t)nrows * 4);
    if (ok && !finished)
      ok = putv(f, pos.data(), 2 * n) && putv(f, vel.data(), 2 * n) && putv(f, rad.data(), n) && putv(f, phase.data(), n) &&
           putv(f, dead.data(), n) && putv(f, absA.data(), n) && putv(f, absR.data(), n) && putv(f, orig.data(), n) &&
           putv(f, keys.data(), n);
    if (f) ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), dest.c_str()) != 0) return false;
  }
  const std::string dest = manifestPath(e->ckptDir, e->ckptSub), tmp = dest + ".tmp";
  FILE *f = fopen(tmp.c_str(), "w");
  bool ok = f && fprintf(f, "%d %d %d %ld\n", gen, nrows, finished ? 1 : 0, steps) > 0;
  if (f) ok = (fclose(f) == 0) && ok;
  if (!ok || rename(tmp.c_str(), dest.c_str()) != 0) return false;
  e->ckptGen = gen;
  return true;
}

// device side of a sub-batch whose members were restored from a checkpoint (all from the same row)
bool uploadRestored(Ensemble *e, float *out, int max_rows) {
  if (!uploadEnsemble(e)) return false;  // (the host mirrors hold the restored state)
  const int m = (int)e->members.size();
  const MemberSaved &s0 = *e->members[0]->saved;
  for (int k = 0; k < m; k++) {
    const MemberSaved &sv = *e->members[k]->saved;
    if (sv.time != s0.time || sv.draws != s0.draws || sv.sorted != s0.sorted || sv.nrows != s0.nrows) return false;
    if (sv.sorted && pbSimSetLayoutOf(e->sim, (unsigned)k, sv.orig.data(), sv.keys.data()) != PB_OK) return false;
  }
  for (int k = 0; k < m; k++) {
    // (the layout is installed once every member has provided one: the state goes in afterwards, in that order)
    const Member &mk = *e->members[k];
    const Particlebot *b = mk.bot;
    if (pbSimSetStateOf(e->sim, (unsigned)k, b->hostPositions(), b->hostVelocities(), b->hostRadii(), b->hostPhases(),
                        b->hostDead()) != PB_OK ||
        pbSimSetForcesOf(e->sim, (unsigned)k, mk.saved->absA.data(), mk.saved->absR.data()) != PB_OK)
      return false;
    if (out) memcpy(out + (size_t)k * max_rows * 4, mk.saved->rows.data(), sizeof(float) * mk.saved->rows.size());
  }
  if (pbSimSetTime(e->sim, s0.time) != PB_OK || pbSimSetPhaseDraws(e->sim, s0.draws) != PB_OK) return false;
  e->haveRow = true;  // the row at the checkpoint's time is among the restored ones
  e->rowTime = s0.time;
  e->stepsBefore = s0.steps;
  return true;
}

// How many summary rows a run from t = 0 writes: the clock and the gate of runSteps below (fp32 t = t + dt; a row
// whenever !(t - di * floorf(t / di) > 0.01f), the last one at the first t > max_time), for at most max_steps steps.
// Stops counting at `limit` + 1: callers only ask "does it fit".
long rowsNeeded(float dt, float di, float max_time, long max_steps, long limit) {
  long rows = 0, steps = 0;
  for (float t = 0.0f;; t = t + dt, steps++) {
    if (!(t - di * floorf(t / di) > 0.01f) && ++rows > limit) break;
    if (t > max_time || steps >= max_steps) break;
    if (t + dt == t) return limit + 1;  // the fp32 clock has stopped short of max_time: rows without end
  }
  return rows;
}

// Runs every member of the batch for up to max_steps timesteps (or to max_time, whichever comes first); can be
// called again to continue.  Row r of member k goes to out[(k * max_rows + r) * 4 ..]: (time, COMx, COMy, distance
// of the COM to the light), one row whenever a dump row would be due (particlebot.cpp:309).
long runSteps(Ensemble *e, long max_steps, float *out, int max_rows, int *rows) {
  const int m = (int)e->members.size();
  const PbRunConfig &c0 = *e->members[0]->cfg;
  const float dt = c0.timestep, di = c0.dump_interval;
  std::vector<double> com(2 * (size_t)m);
  long steps = 0;
  int nrows = rows ? *rows : 0;
  float t = 0.0f;
  if (pbSimGetTime(e->sim, &t) != PB_OK) return -1;
  for (;;) {
    // a row is due at time t; e->rowTime remembers the last one written so that a call which stopped
    // exactly at a dump time does not write it twice when the run is continued
    const bool rowDue = !(t - di * floorf(t / di) > 0.01f) && !(e->haveRow && e->rowTime == t);
    if (rowDue && !e->csvDir.empty() && (!out || nrows >= max_rows)) {
      // the member CSVs are documented as byte for byte the reference's: never a silently shortened file
      // (pbEnsemblePipelineRun refuses such a run before its first step; this is the stepwise API's guard)
      fprintf(stderr, "pbEnsemble: a CSV row is due at t = %g but the row buffer holds %d rows (max_rows %d): "
              "%s/member_*.csv would stop here; raise max_rows or the dump interval\n",
              (double)t, out ? nrows : 0, out ? max_rows : 0, e->csvDir.c_str());
      return -1;
    }
    if (rowDue && out && nrows >= max_rows) {
      // (the same for the summary rows themselves: a caller who passes a buffer gets every row or an error)
      fprintf(stderr, "pbEnsemble: a summary row is due at t = %g but the row buffer is full (max_rows %d); raise max_rows "
              "or the dump interval, or pass no buffer\n", (double)t, max_rows);
      return -1;
    }
    if (out && rowDue && nrows < max_rows) {
      if (pbSimCentroids(e->sim, com.data()) != PB_OK) return -1;
      for (int k = 0; k < m; k++) {
        const SimParams &p = e->members[k]->bot->getParams();
        float *row = out + ((size_t)k * max_rows + nrows) * 4;
        const double dx = com[2 * k] - p.light_x, dy = com[2 * k + 1] - p.light_y;
        row[0] = t;
        row[1] = (float)com[2 * k];
        row[2] = (float)com[2 * k + 1];
        row[3] = (float)sqrt(dx * dx + dy * dy);
      }
      if (!e->csvDir.empty()) {
        std::vector<float> sums(2 * (size_t)m);
        if (pbSimCentroidSums(e->sim, sums.data()) != PB_OK) return -1;
        if (e->csvFiles.empty()) e->csvFiles.assign(m, nullptr);
        for (int k = 0; k < m; k++) {
          if (!e->csvFiles[k]) {
            char name[64];
            snprintf(name, sizeof name, "/member_%06d.csv", e->csvIds[k]);
            e->csvFiles[k] = fopen((e->csvDir + name).c_str(), "w");
            if (!e->csvFiles[k]) {
              fprintf(stderr, "pbEnsemble: cannot write %s%s\n", e->csvDir.c_str(), name);
              return -1;
            }
          }
          const SimParams &p = e->members[k]->bot->getParams();
          writeCsvRow(e->csvFiles[k], t, p.seed, sums[2 * k], sums[2 * k + 1], p.nCells, p.light_x, p.light_y);
          if (ferror(e->csvFiles[k])) {
            fprintf(stderr, "pbEnsemble: write error on %s/member_%06d.csv\n", e->csvDir.c_str(), e->csvIds[k]);
            return -1;
          }
        }
      }
      nrows++;
      e->haveRow = true;
      e->rowTime = t;
      if (!e->ckptDir.empty() &&
          !saveSubBatch(e, out, max_rows, nrows, e->stepsBefore + steps, t > c0.params.max_time)) {
        fprintf(stderr, "pbEnsemble: cannot write the checkpoint of sub-batch %d under %s\n", e->ckptSub, e->ckptDir.c_str());
        return -1;
      }
    }
    if (t > c0.params.max_time) {
      // the run is over: the member CSVs are complete only if every buffered byte reached the disk
      for (size_t k = 0; k < e->csvFiles.size(); k++) {
        FILE *f = e->csvFiles[k];
        e->csvFiles[k] = nullptr;
        if (f && fclose(f) != 0) {
          fprintf(stderr, "pbEnsemble: cannot finish %s/member_%06d.csv\n", e->csvDir.c_str(), e->csvIds[k]);
          return -1;
        }
      }
      break;
    }
    if (steps >= max_steps) break;
    // host events at this step: dead-bot draws (those due at time 0 came with the placement)
    for (int k = 0; k < m; k++) {
      Member *mk = e->members[k];
      Particlebot *b = mk->bot;
      b->setHostTime(t);
      if (b->deadDrawDue(dt) && !mk->deadDrawn) {
        mk->deadDrawn = true;
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
        e->members[k]->bot->setHostTime(tt);
        stop = e->members[k]->bot->deadDrawDue(dt);
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

double nowSeconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- the pipelined form: host placement of sub-batch k+1 overlapped with device stepping of sub-batch k --------
// A rank's members are cut into sub-batches of `sub` members (in member order).  A pool of producer threads builds
// members in order (configuration, placement, early dead draw), never more than `ahead` sub-batches beyond the one
// on the device; the calling thread takes the sub-batches in order: waits until its members are built, creates the
// batched pbSim, uploads, steps it to the end, keeps what was asked for, frees it.  A member's trajectory does not
// depend on which members share its batch (every kernel form is bit-identical to the member's own oracle run; the
// centroid is a fixed-order per-member reduction), so the rows are independent of `sub`
// (tests/test_gpu_ensemble_pipeline.py).
struct Pipeline {
  std::string cfgPath, common;
  bool haveCfg = false;
  std::vector<std::string> over;
  int nmembers = 0, sub = 0, threads = 1, ahead = 2;
  int lanes = 1;  // sub-batches stepped at the same time (pbEnsemblePipelineRun)
  std::string csvDir;       // pbEnsemblePipelineSetCsvDir
  std::vector<int> csvIds;  // [nmembers]
  bool keepStates = false;
  std::vector<Member *> built;  // [nmembers], filled by the producers, taken by the consumer
  std::vector<char> ready;
  std::mutex mu;
  std::condition_variable cvReady, cvRoom;
  int nextToBuild = 0, consumedUpTo = 0;  // members < consumedUpTo have been taken by the consumer
  bool failed = false, stop = false;
  std::vector<std::thread> pool;
  std::vector<std::vector<float>> finalPos, finalVel, finalRad;
  unsigned nbots = 0;
  pbEnsembleTimings tm{};
  std::vector<double> cpuSeconds, wallSeconds;  // per producer thread: CPU time and wall time spent building members
  std::vector<int> pinCpus;                     // cores of the GPU's NUMA node (empty: producers are not pinned)
  int numaNode = -1;
  int device = -1;      // the creating thread's HIP device: Run may be called from another thread (which starts on 0)
  std::string ckptDir;  // checkpoints (pbEnsemblePipelineSetCheckpoint); empty: none
  bool resume = false;
  bool started = false;
  // One placement per distinct blob: members whose Particlebot::placementKey() agree -- a sweep of
  // nDead, light position, ... under one seed -- are placed once; the others take a copy of the placed state and of
  // the private generator's state after the placement, so every member is bit-identical to its stand-alone run.
  // keyOf[k] indexes `shared` (-1: this member's key is unique, or sharing is off: PB_SHARE_PLACEMENTS=0).
  struct SharedPlacement {
    int state = 0;      // 0 nobody has started it, 1 being placed, 2 ready, -1 failed
    int usesLeft = 0;   // members that still have to take it (freed at 0)
    Particlebot::Placement placed;
  };
  std::vector<int> keyOf;
  std::vector<SharedPlacement> shared;
  std::condition_variable cvPlaced;
  int placementsRun = 0, placementsShared = 0;

  // (calling thread, before the pool starts) group the members by placement key; a member whose configuration does
  // not load keeps -1 and fails in its producer as before
  void groupPlacements() {
    keyOf.assign(nmembers, -1);
    const char *env = getenv("PB_SHARE_PLACEMENTS");
    if ((env && env[0] == '0') || resume) return;
    std::map<std::string, std::vector<int>> groups;
    const char *cp = haveCfg ? cfgPath.c_str() : nullptr, *co = common.empty() ? nullptr : common.c_str();
    for (int k = 0; k < nmembers; k++) {
      // (the key needs the configuration only, not a Particlebot with its host arrays)
      PbRunConfig c;
      c.params.seed = 0;
      if (cp && !c.loadFile(cp)) continue;
      applyOverrides(c, co);
      applyOverrides(c, over[k].c_str());
      c.derive();
      groups[Particlebot::placementKeyOf(c.params, c.hex_spacing, c.square_lattice, c.fast_blob)].push_back(k);
    }
    for (auto &g : groups) {
      if (g.second.size() < 2) continue;
      SharedPlacement sp;
      sp.usesLeft = (int)g.second.size();
      for (int k : g.second) keyOf[k] = (int)shared.size();
      shared.push_back(std::move(sp));
    }
  }

  // Place shared blob `g` on this thread from member `k`'s configuration (its placement inputs are the group's) and
  // publish it.  Returns the member built along the way (placed, dead draw not yet done) or nullptr on failure.
  Member *placeShared(int g, int k) {
    Member *m = new Member();
    const char *cp = haveCfg ? cfgPath.c_str() : nullptr, *co = common.empty() ? nullptr : common.c_str();
    Particlebot::Placement placed;
    const bool ok = buildMember(*m, cp, co, over[k].c_str(), nullptr, &placed);
    std::lock_guard<std::mutex> lock(mu);
    SharedPlacement &sp = shared[g];
    if (ok) {
      sp.placed = std::move(placed);
      sp.state = 2;
      placementsRun++;
    } else {
      sp.state = -1;
    }
    cvPlaced.notify_all();
    if (!ok) {
      delete m;
      return nullptr;
    }
    return m;
  }

  // Member k of a shared group: take the group's placement, placing it first if nobody has.  While another thread is
  // placing it, this thread does not idle: it places the next group ahead that nobody has started (its members will
  // find it ready), then looks again.
  Member *buildShared(int k) {
    const int g = keyOf[k];
    for (;;) {
      int other = -1, otherMember = -1;
      {
        std::unique_lock<std::mutex> lock(mu);
        SharedPlacement &sp = shared[g];
        if (sp.state == 0) {
          sp.state = 1;
          lock.unlock();
          Member *m = placeShared(g, k);   // this member IS the one built along the way
          if (m) releaseShared(g);
          return m;
        }
        if (sp.state == -1) return nullptr;
        if (sp.state == 2) break;
        // being placed elsewhere: look ahead for work
        for (int j = k + 1; j < nmembers && other < 0; j++)
          if (keyOf[j] >= 0 && shared[keyOf[j]].state == 0) other = keyOf[j], otherMember = j;
        if (other >= 0) {
          shared[other].state = 1;
        } else {
          cvPlaced.wait(lock, [&] { return stop || failed || shared[g].state != 1; });
          if (stop || failed) return nullptr;
          continue;
        }
      }
      delete placeShared(other, otherMember);   // (only the placement is kept; member `otherMember` is built in its turn)
    }
    Member *m = new Member();
    const char *cp = haveCfg ? cfgPath.c_str() : nullptr, *co = common.empty() ? nullptr : common.c_str();
    // (state 2 entries are immutable until their last user has released them: read without the lock)
    if (!buildMember(*m, cp, co, over[k].c_str(), &shared[g].placed)) {
      delete m;
      return nullptr;
    }
    {
      std::lock_guard<std::mutex> lock(mu);
      placementsShared++;
    }
    releaseShared(g);
    return m;
  }
  void releaseShared(int g) {
    std::lock_guard<std::mutex> lock(mu);
    if (--shared[g].usesLeft == 0) shared[g].placed = Particlebot::Placement();
  }

  void producer(int tid) {
    for (;;) {
      int k;
      {
        std::unique_lock<std::mutex> lock(mu);
        // room: at most `ahead` sub-batches beyond the one the consumer is on
        cvRoom.wait(lock, [&] { return stop || failed || nextToBuild >= nmembers || nextToBuild < consumedUpTo + (ahead + 1) * sub; });
        if (stop || failed || nextToBuild >= nmembers) return;
        k = nextToBuild++;
      }
      const double t0 = nowSeconds(), c0 = threadCpuSeconds();
      Member *m = new Member();
      const char *cp = haveCfg ? cfgPath.c_str() : nullptr, *co = common.empty() ? nullptr : common.c_str();
      bool ok = true, restored = false;
      if (resume) {
        // a member whose sub-batch has a complete checkpoint is restored, not placed
        int gen = 0, nrows = 0, finished = 0;
        long steps = 0;
        if (readManifest(ckptDir, k / sub, gen, nrows, finished, steps)) {
          ok = configureMember(*m, cp, co, over[k].c_str());
          restored = ok && loadMemberFile(*m, ckptDir, k, gen, nrows, steps);
          if (ok && !restored) {
            fprintf(stderr, "pbEnsemblePipeline: checkpoint of member %d under %s is unusable\n", k, ckptDir.c_str());
            ok = false;
          }
        }
      }
      if (ok && !restored) {
        delete m;
        if (keyOf[k] >= 0) {
          m = buildShared(k);
          ok = m != nullptr;
        } else {
          m = new Member();
          ok = buildMember(*m, cp, co, over[k].c_str());
          if (ok) {
            std::lock_guard<std::mutex> lock(mu);
            placementsRun++;
          }
        }
      }
      cpuSeconds[tid] += threadCpuSeconds() - c0;
      wallSeconds[tid] += nowSeconds() - t0;
      std::lock_guard<std::mutex> lock(mu);
      if (!ok) {
        delete m;
        failed = true;
      } else {
        built[k] = m;
        ready[k] = 1;
      }
      cvReady.notify_all();
      cvRoom.notify_all();
      cvPlaced.notify_all();
    }
  }
  void start() {
    cpuSeconds.assign(threads, 0.0);
    wallSeconds.assign(threads, 0.0);
    for (int t = 0; t < threads; t++) pool.emplace_back(&Pipeline::producer, this, t);
    if (!pinCpus.empty()) {
      // the whole pool on the cores next to the GPU (the scheduler spreads the threads inside the set)
      cpu_set_t set;
      CPU_ZERO(&set);
      for (int c : pinCpus) CPU_SET(c, &set);
      bool ok = true;
      for (auto &th : pool) ok = pthread_setaffinity_np(th.native_handle(), sizeof set, &set) == 0 && ok;
      tm.pinned = ok ? 1 : 0;
    }
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lock(mu);
      stop = true;
    }
    cvRoom.notify_all();
    cvReady.notify_all();
    cvPlaced.notify_all();
    for (auto &th : pool) th.join();
    pool.clear();
    for (auto *&m : built) {
      delete m;
      m = nullptr;
    }
  }
  ~Pipeline() { shutdown(); }
};

}  // namespace

extern "C" {

void *pbEnsembleCreate(const char *cfg_path, const char *common_overrides, const char **member_overrides,
                       int nmembers) {
  if (nmembers < 1) return nullptr;
  Ensemble *e = new Ensemble();
  e->members.assign(nmembers, nullptr);
  // Members are independent (own configuration, own private random stream, own placement grid):
  // build them on all host cores.  The reference's random placement is O(N^1.5) (1.4 s for 10^5
  // bots), so a sweep of large members would otherwise spend minutes here.  (pbEnsemblePipeline* overlaps
  // this with the device work of the members built before.)
  // Members whose placement inputs agree (Particlebot::placementKey: a sweep under one seed) are placed ONCE, by the
  // thread that takes their group, and share the placed state + the generator state after it (as the pipeline does).
  std::vector<std::vector<int>> groups;
  {
    std::map<std::string, int> index;
    const char *env = getenv("PB_SHARE_PLACEMENTS");
    const bool share = !(env && env[0] == '0');
    for (int k = 0; k < nmembers; k++) {
      PbRunConfig c;
      c.params.seed = 0;
      std::string key;
      if (share && (!cfg_path || c.loadFile(cfg_path))) {
        applyOverrides(c, common_overrides);
        applyOverrides(c, member_overrides ? member_overrides[k] : nullptr);
        c.derive();
        key = Particlebot::placementKeyOf(c.params, c.hex_spacing, c.square_lattice, c.fast_blob);
      }
      auto it = key.empty() ? index.end() : index.find(key);
      if (it == index.end()) {
        if (!key.empty()) index[key] = (int)groups.size();
        groups.push_back({k});
      } else {
        groups[it->second].push_back(k);
      }
    }
  }
  std::atomic<int> next{0};
  std::atomic<bool> failed{false};
  auto worker = [&]() {
    for (int g = next++; g < (int)groups.size() && !failed; g = next++) {
      Particlebot::Placement placed;
      for (size_t j = 0; j < groups[g].size() && !failed; j++) {
        const int k = groups[g][j];
        Member *m = new Member();
        e->members[k] = m;
        const bool first = j == 0, more = groups[g].size() > 1;
        if (!buildMember(*m, cfg_path, common_overrides, member_overrides ? member_overrides[k] : nullptr,
                         first ? nullptr : &placed, first && more ? &placed : nullptr))
          failed = true;
      }
    }
  };
  const unsigned nthreads = std::min<unsigned>(hostThreads(0), (unsigned)groups.size());
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nthreads; t++) pool.emplace_back(worker);
  worker();
  for (auto &th : pool) th.join();
  if (failed) {
    delete e;
    return nullptr;
  }
  if (!uploadEnsemble(e)) {
    fprintf(stderr, "pbEnsembleCreate: %s\n", pbGetLastErrorString());
    delete e;
    return nullptr;
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
  return runSteps((Ensemble *)ev, max_steps, out, max_rows, rows);
}

// Runs every member to max_time (the whole run in one call).
long pbEnsembleRun(void *ev, float *out, int max_rows, int *rows) {
  int nrows = 0;
  const long steps = pbEnsembleRunSteps(ev, LONG_MAX, out, max_rows, &nrows);
  if (rows) *rows = nrows;
  return steps;
}

// ---- pipelined ensembles (include/particlebot_ensemble.h) ------------------------------------------------------
// The automatic sub-batch size (sub_batch -1) for members of `bots_per_member` bots and `producers` threads: whole
// placement rounds of the pool (every producer places one member per round; 1 ... 8 rounds) that bring a sub-batch
// to ~3 x 10^6 bots.  Measured on BASELINE configs[4] whole (1 024 members of 10^5 bots, 12 000 steps, 15 producers,
// one MI355X, end to end): sub-batches of 15 / 30 / 45 / 60 / 120 members = 96 / 90.1 / 93.3 / 97.1 / 96.0 s.  Small
// sub-batches pay a launch's ramp and drain on every step; large ones (> ~200 MB of state) leave the Infinity Cache,
// which the force kernel's neighbour reads of an evolving blob live on (and the first sub-batch is waited for).
int pbEnsemblePipelineAutoSubBatch(unsigned bots_per_member, int producers) {
  const int threads = std::max(producers, 1);
  const double want = 3.0e6 / (double)std::max(bots_per_member, 1u);
  const int rounds = std::max(1, std::min(8, (int)(want / threads + 0.5)));
  // ... but never more bots than the cache target, however many producers the host has (127 producers x 10^5 bots
  // would be 12.7 x 10^6 bots in flight over the two lanes, 4 x the target; with 10^6-bot members (ahead + 1 + lanes)
  // sub-batches of placed members are alive on the host): a pool larger than the sub-batch simply works further ahead
  const int cap = std::max(1, (int)(want + 0.5));
  return std::min(threads * rounds, std::max(cap, 1));
}

void *pbEnsemblePipelineCreateCheckpointed(const char *cfg_path, const char *common_overrides,
                                           const char **member_overrides, int nmembers, int sub_batch, int host_threads,
                                           int keep_final_states, const char *checkpoint_dir, int resume) {
  if (nmembers < 1) return nullptr;
  Pipeline *p = new Pipeline();
  p->haveCfg = cfg_path != nullptr;
  if (cfg_path) p->cfgPath = cfg_path;
  if (common_overrides) p->common = common_overrides;
  for (int k = 0; k < nmembers; k++) p->over.emplace_back(member_overrides && member_overrides[k] ? member_overrides[k] : "");
  p->nmembers = nmembers;
  // the calling thread drives the device: leave it a core when there are several
  pbHostResources res;
  describeResources(res, host_threads);
  const unsigned avail = (unsigned)res.host_threads;
  if (res.pin_producers) {
    std::string bus;
    p->numaNode = numaOfDevice(res.device, bus, p->pinCpus);
  }
  p->tm.numa_node = res.numa_node;
  p->threads = (int)std::max(1u, std::min<unsigned>(host_threads > 0 ? avail : (avail > 1 ? avail - 1 : 1), (unsigned)nmembers));
  // sub_batch -1: pbEnsemblePipelineAutoSubBatch for members of this size
  unsigned botsPerMember = 0;
  {
    PbRunConfig c0;
    c0.params.seed = 0;
    if (!cfg_path || c0.loadFile(cfg_path)) {
      applyOverrides(c0, common_overrides);
      applyOverrides(c0, p->over[0].c_str());
      c0.derive();
      botsPerMember = c0.params.nCells;
    }
  }
  int autoSub = pbEnsemblePipelineAutoSubBatch(botsPerMember, p->threads);
  // Members that share their placement cost the pool one placement per group, not one per member: when most members
  // are copies (a parameter sweep under a few seeds), the "whole placement rounds of the pool" bound above says
  // nothing -- a one-producer rank would step sub-batches of 4 members of 10^5 bots, paying a launch's ramp and drain
  // on a tenth of the chip -- and the cache target alone sizes the sub-batch.
  p->resume = resume != 0 && checkpoint_dir && checkpoint_dir[0];
  p->groupPlacements();
  int distinct = (int)p->shared.size();
  for (int k : p->keyOf) distinct += k < 0 ? 1 : 0;
  if (distinct * 4 <= nmembers)
    autoSub = std::max(autoSub, pbEnsemblePipelineAutoSubBatch(botsPerMember, 1 << 20));
  // The automatic decomposition steps TWO sub-batches of half that size at the same time (a launch's ramp and drain
  // overlap the other sub-batch's steady state, and together they still fit the Infinity Cache: configs[4] slice of
  // 240 members, one pipeline of 30-member sub-batches 22.0 s, two lanes of 15 21.2 s, three of 10 21.1 s;
  // tools/experiments/two_pipelines.py).  PB_PIPELINE_LANES overrides (1: one at a time, as an explicit sub_batch).
  if (sub_batch == -1) {
    p->lanes = 2;
    if (const char *e = getenv("PB_PIPELINE_LANES")) p->lanes = std::max(1, std::min(4, atoi(e)));
    if (p->lanes > 1) autoSub = std::max(1, autoSub / p->lanes);
  }
  p->sub = sub_batch == -1 ? std::min(autoSub, nmembers) : (sub_batch < 1 || sub_batch > nmembers) ? nmembers : sub_batch;
  if (sub_batch == -1 && resume && checkpoint_dir && checkpoint_dir[0]) {
    // a sweep resumed with the automatic size continues with the size it was started with, whatever the number of
    // producer threads is this time (the directory belongs to one decomposition)
    int m0 = 0, s0 = 0;
    FILE *f = fopen((std::string(checkpoint_dir) + "/run.info").c_str(), "r");
    if (f && fscanf(f, "members %d sub_batch %d", &m0, &s0) == 2 && m0 == nmembers && s0 >= 1 && s0 <= nmembers) p->sub = s0;
    if (f) fclose(f);
  }
  p->keepStates = keep_final_states != 0;
  p->built.assign(nmembers, nullptr);
  p->ready.assign(nmembers, 0);
  if (p->keepStates) {
    p->finalPos.resize(nmembers);
    p->finalVel.resize(nmembers);
    p->finalRad.resize(nmembers);
  }
  p->tm.host_threads = p->threads;
  p->tm.sub_batch = p->sub;
  if (pbGetDevice(&p->device) != PB_OK) p->device = -1;  // (no device: the dry-run consumer of the CPU tests)
  if (checkpoint_dir && checkpoint_dir[0]) {
    p->ckptDir = checkpoint_dir;
    p->resume = resume != 0;
    (void)mkdir(checkpoint_dir, 0777);
    // a checkpoint directory belongs to ONE decomposition of the ensemble
    const std::string info = p->ckptDir + "/run.info";
    char want[128];
    snprintf(want, sizeof want, "members %d sub_batch %d\n", nmembers, p->sub);
    if (p->resume) {
      char got[128] = {0};
      FILE *f = fopen(info.c_str(), "r");
      const bool same = f && fgets(got, sizeof got, f) && strcmp(got, want) == 0;
      if (f) fclose(f);
      if (!same) {
        fprintf(stderr, "pbEnsemblePipeline: %s was written for another decomposition (%s) than this one (%s)\n",
                checkpoint_dir, got, want);
        delete p;
        return nullptr;
      }
    } else {
      FILE *f = fopen(info.c_str(), "w");
      if (!f || fputs(want, f) < 0 || fclose(f) != 0) {
        fprintf(stderr, "pbEnsemblePipeline: cannot write under %s\n", checkpoint_dir);
        delete p;
        return nullptr;
      }
      // (manifests of an earlier run in the same directory must not be mistaken for this run's)
      for (int b = 0; b * p->sub < nmembers; b++) (void)remove(manifestPath(p->ckptDir, b).c_str());
    }
  }
  p->start();  // placement starts now, before the caller asks for the first step
  return p;
}

void *pbEnsemblePipelineCreate(const char *cfg_path, const char *common_overrides, const char **member_overrides,
                               int nmembers, int sub_batch, int host_threads, int keep_final_states) {
  return pbEnsemblePipelineCreateCheckpointed(cfg_path, common_overrides, member_overrides, nmembers, sub_batch,
                                              host_threads, keep_final_states, nullptr, 0);
}

void pbEnsemblePipelineDestroy(void *pv) { delete (Pipeline *)pv; }

int pbEnsemblePipelineHostThreads(void *pv) { return pv ? ((Pipeline *)pv)->threads : 0; }

// Every member also writes DIR/member_<id>.csv in the reference's own format (testing = 0: seed, header, one row per
// dump interval with the reference's fp32 centroid), ids[k] = the number of this pipeline's member k in the whole
// ensemble (NULL: k).  Before Run; not together with checkpoints (a resumed run could not rewrite the rows it skips).
int pbEnsemblePipelineSetCsvDir(void *pv, const char *dir, const int *ids) {
  Pipeline *p = (Pipeline *)pv;
  if (!p || p->consumedUpTo != 0 || !dir || !dir[0] || !p->ckptDir.empty()) return -1;
  (void)mkdir(dir, 0777);
  p->csvDir = dir;
  p->csvIds.resize(p->nmembers);
  for (int k = 0; k < p->nmembers; k++) p->csvIds[k] = ids ? ids[k] : k;
  return 0;
}

// Sub-batches stepped at the same time by Run (1 ... 4; before Run).  Rows and states do not depend on it.
int pbEnsemblePipelineSetLanes(void *pv, int lanes) {
  Pipeline *p = (Pipeline *)pv;
  if (!p || p->consumedUpTo != 0 || lanes < 1 || lanes > 4) return -1;
  p->lanes = lanes;
  return 0;
}

long pbEnsemblePipelineRun(void *pv, long max_steps, float *out, int max_rows, int *rows, pbEnsembleTimings *timings) {
  Pipeline *p = (Pipeline *)pv;
  if (!p || p->consumedUpTo != 0) return -1;  // one run per pipeline
  if (p->device >= 0 && pbSetDevice(p->device) != PB_OK) return -1;  // this thread may be new: batches go where Create was
  const double t0 = nowSeconds();
  long steps = -1;
  int nrowsAll = 0;
  const int nsub = (p->nmembers + p->sub - 1) / p->sub;
  const int lanes = std::max(1, std::min(p->lanes, nsub));
  std::mutex resMu;            // steps / nrowsAll / the timing sums
  double waitS = 0.0, uploadS = 0.0, deviceS = 0.0;
  // One sub-batch from "its members are placed" to "its rows are in `out`".  Runs on the calling thread, or -- with
  // two lanes -- on two threads that take the sub-batches alternately, each on its batch's own stream: a launch's
  // ramp and drain (~13 us of every step) then overlap the other sub-batch's steady state.
  auto runSub = [&](int b) -> int {
    const int first = b * p->sub;
    const int count = std::min(p->sub, p->nmembers - first);
    double myWait = 0.0, myUpload = 0.0, myDevice = 0.0;
    Ensemble e;
    {
      const double w0 = nowSeconds();
      std::unique_lock<std::mutex> lock(p->mu);
      p->cvReady.wait(lock, [&] {
        if (p->failed) return true;
        for (int k = first; k < first + count; k++)
          if (!p->ready[k]) return false;
        return true;
      });
      if (p->failed) return -1;
      // refuse a run whose rows cannot fit BEFORE its first step, not hours into it
      const PbRunConfig &c0 = *p->built[first]->cfg;
      const long need = out ? rowsNeeded(c0.timestep, c0.dump_interval, c0.params.max_time, max_steps, max_rows) : 0;
      if (need > max_rows) {
        fprintf(stderr, "pbEnsemblePipelineRun: the run writes more than %d summary rows per member (dump_interval %g, "
                "max_time %g) and the row buffer holds %d: raise max_rows (particlebot_ensemble --max-rows) or the dump "
                "interval; nothing was stepped\n", max_rows, (double)c0.dump_interval, (double)c0.params.max_time, max_rows);
        return -1;
      }
      for (int k = first; k < first + count; k++) {
        e.members.push_back(p->built[k]);
        p->built[k] = nullptr;
      }
      p->consumedUpTo = std::max(p->consumedUpTo, first + count);
      myWait = nowSeconds() - w0;
    }
    p->cvRoom.notify_all();
    const double u0 = nowSeconds();
    float *const outSub = out ? out + (size_t)first * max_rows * 4 : nullptr;
    const unsigned nbHere = e.members[0]->bot->getParams().nCells;
    // restored from a checkpoint?  (all members of a sub-batch or none: they share one manifest)
    int nRestored = 0;
    for (Member *mk : e.members) nRestored += mk->saved ? 1 : 0;
    if (nRestored != 0 && nRestored != count) {
      fprintf(stderr, "pbEnsemblePipelineRun: sub-batch %d is only partly in the checkpoint\n", first / p->sub);
      return -1;
    }
    if (!p->csvDir.empty()) {
      e.csvDir = p->csvDir;
      e.csvIds.assign(p->csvIds.begin() + first, p->csvIds.begin() + first + count);
    }
    e.ckptDir = p->ckptDir;
    e.ckptSub = first / p->sub;
    e.ckptFirst = first;
    long done = 0;
    int nrows = 0;
    if (nRestored && e.members[0]->saved->finished) {
      // nothing left to run: the rows are the result
      const MemberSaved &s0 = *e.members[0]->saved;
      nrows = s0.nrows;
      done = s0.steps;
      if (nrows > max_rows) return -1;
      for (int k = 0; k < count && outSub; k++)
        memcpy(outSub + (size_t)k * max_rows * 4, e.members[k]->saved->rows.data(), sizeof(float) * 4 * (size_t)nrows);
      myUpload = nowSeconds() - u0;
    } else {
      if (nRestored) {
        int gen = 0, r0 = 0, fin = 0;
        long st = 0;
        (void)readManifest(p->ckptDir, e.ckptSub, gen, r0, fin, st);
        e.ckptGen = gen;
        nrows = e.members[0]->saved->nrows;
        if (nrows > max_rows || !outSub) return -1;
      }
      if (nRestored ? !uploadRestored(&e, outSub, max_rows) : !uploadEnsemble(&e)) {
        fprintf(stderr, "pbEnsemblePipelineRun: %s\n", pbGetLastErrorString());
        return -1;
      }
      const double d0 = nowSeconds();
      myUpload = d0 - u0;
      const long more = runSteps(&e, max_steps - e.stepsBefore, outSub, max_rows, &nrows);
      if (more < 0 || pbSimSynchronize(e.sim) != PB_OK) return -1;
      done = e.stepsBefore + more;
      if (!p->ckptDir.empty()) {
        // the end of this sub-batch's run: finished (past max_time) or stopped by max_steps (resumable)
        float tEnd = 0.0f;
        if (pbSimGetTime(e.sim, &tEnd) != PB_OK) return -1;
        const bool finished = tEnd > e.members[0]->cfg->params.max_time;
        if (!outSub || !saveSubBatch(&e, outSub, max_rows, nrows, done, finished)) {
          fprintf(stderr, "pbEnsemblePipelineRun: cannot write the checkpoint of sub-batch %d\n", e.ckptSub);
          return -1;
        }
      }
      myDevice = nowSeconds() - d0;
    }
    if (p->keepStates && e.sim)
      for (int k = 0; k < count; k++) {
        const size_t n = nbHere;
        p->finalPos[first + k].resize(2 * n);
        p->finalVel[first + k].resize(2 * n);
        p->finalRad[first + k].resize(n);
        if (pbSimGetStateOf(e.sim, (unsigned)k, p->finalPos[first + k].data(), p->finalVel[first + k].data(),
                            p->finalRad[first + k].data(), nullptr, nullptr, nullptr, nullptr) != PB_OK)
          return -1;
      }
    std::lock_guard<std::mutex> lock(resMu);
    p->nbots = nbHere;
    if (steps < 0) {
      steps = done;
      nrowsAll = nrows;
    } else if (done != steps || nrows != nrowsAll) {
      fprintf(stderr, "pbEnsemblePipelineRun: sub-batches disagree on the step or row count (%ld/%d vs %ld/%d)\n", done,
              nrows, steps, nrowsAll);
      return -1;
    }
    waitS += myWait, uploadS += myUpload, deviceS += myDevice;
    p->tm.sub_batches++;
    return 0;
  };  // (~Ensemble frees the sub-batch's device memory and its members)
  if (lanes == 1) {
    for (int b = 0; b < nsub; b++)
      if (runSub(b) != 0) return -1;
  } else {
    std::atomic<int> next{0};
    std::atomic<bool> bad{false};
    auto lane = [&](bool setDevice) {
      if (setDevice && p->device >= 0 && pbSetDevice(p->device) != PB_OK) bad = true;
      for (int b; !bad && (b = next++) < nsub;)
        if (runSub(b) != 0) {
          bad = true;
          std::lock_guard<std::mutex> lock(p->mu);  // (wake a lane that waits for members)
          p->failed = true;
          p->cvReady.notify_all();
          p->cvRoom.notify_all();
        }
    };
    std::vector<std::thread> extra;
    for (int l = 1; l < lanes; l++) extra.emplace_back(lane, true);
    lane(false);
    for (std::thread &t : extra) t.join();
    if (bad) return -1;
  }
  // (per lane: the lanes wait, upload and step at the same time)
  p->tm.placement_wait_s = waitS / lanes, p->tm.upload_s = uploadS / lanes, p->tm.device_s = deviceS / lanes;
  p->tm.lanes = lanes;
  p->tm.wall_s = nowSeconds() - t0;
  p->tm.placement_cpu_s = p->tm.placement_thread_wall_s = 0.0;
  for (double c : p->cpuSeconds) p->tm.placement_cpu_s += c;
  for (double c : p->wallSeconds) p->tm.placement_thread_wall_s += c;
  {
    std::lock_guard<std::mutex> lock(p->mu);
    p->tm.placements_run = p->placementsRun, p->tm.placements_shared = p->placementsShared;
  }
  if (rows) *rows = nrowsAll;
  if (timings) *timings = p->tm;
  return steps;
}

// The consumer side WITHOUT a device (CPU tests of the pipeline's ordering): takes the sub-batches in order exactly
// as Run does, and instead of stepping them records, per member, a checksum of the placed state (positions, radii,
// dead flags) and dwells `dwell_ms` per sub-batch so that the producers run into the look-ahead bound.  *max_ahead
// receives the largest number of members that were ever claimed by producers beyond the consumed ones.
int pbEnsemblePipelineDryRun(void *pv, int dwell_ms, unsigned long long *checksums, int *max_ahead) {
  Pipeline *p = (Pipeline *)pv;
  if (!p || p->consumedUpTo != 0 || !checksums) return -1;
  int worst = 0;
  for (int first = 0; first < p->nmembers; first += p->sub) {
    const int count = std::min(p->sub, p->nmembers - first);
    std::vector<Member *> mine;
    {
      std::unique_lock<std::mutex> lock(p->mu);
      p->cvReady.wait(lock, [&] {
        if (p->failed) return true;
        for (int k = first; k < first + count; k++)
          if (!p->ready[k]) return false;
        return true;
      });
      if (p->failed) return -1;
      worst = std::max(worst, p->nextToBuild - p->consumedUpTo);
      for (int k = first; k < first + count; k++) {
        mine.push_back(p->built[k]);
        p->built[k] = nullptr;
      }
      p->consumedUpTo = first + count;
    }
    p->cvRoom.notify_all();
    for (int k = 0; k < count; k++) {
      const Particlebot *b = mine[k]->bot;
      const size_t n = b->getParams().nCells;
      p->nbots = (unsigned)n;
      unsigned long long h = 1469598103934665603ull;
      auto mix = [&](const void *data, size_t bytes) {
        const unsigned char *c = (const unsigned char *)data;
        for (size_t i = 0; i < bytes; i++) h = (h ^ c[i]) * 1099511628211ull;
      };
      mix(b->hostPositions(), 8 * n);
      mix(b->hostRadii(), 4 * n);
      mix(b->hostDead(), 4 * n);
      checksums[first + k] = h;
      delete mine[k];
    }
    if (dwell_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(dwell_ms));
    {
      std::lock_guard<std::mutex> lock(p->mu);
      worst = std::max(worst, p->nextToBuild - p->consumedUpTo);
    }
  }
  if (max_ahead) *max_ahead = worst;
  return 0;
}

unsigned pbEnsemblePipelineNumBots(void *pv) { return ((Pipeline *)pv)->nbots; }

void pbEnsemblePipelinePlacementCounts(void *pv, int *run, int *shared) {
  Pipeline *p = (Pipeline *)pv;
  if (!p) return;
  std::lock_guard<std::mutex> lock(p->mu);
  if (run) *run = p->placementsRun;
  if (shared) *shared = p->placementsShared;
}

int pbEnsemblePipelineGetState(void *pv, int member, float *pos, float *vel, float *rad) {
  Pipeline *p = (Pipeline *)pv;
  if (!p || !p->keepStates || member < 0 || member >= p->nmembers || p->finalRad[member].empty()) return 1;
  const size_t n = p->nbots;
  if (pos) memcpy(pos, p->finalPos[member].data(), 8 * n);
  if (vel) memcpy(vel, p->finalVel[member].data(), 8 * n);
  if (rad) memcpy(rad, p->finalRad[member].data(), 4 * n);
  return 0;
}

int pbEnsembleSynchronize(void *ev) { return pbSimSynchronize(((Ensemble *)ev)->sim); }

int pbHostGetResources(pbHostResources *out) {
  if (!out) return 1;
  describeResources(*out, 0);
  return 0;
}

int pbHostParseCpuList(const char *text, int *cpus, int cap) {
  const std::vector<int> v = parseCpuList(text);
  for (int i = 0; cpus && i < cap && i < (int)v.size(); i++) cpus[i] = v[i];
  return (int)v.size();
}

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

unsigned pbEnsembleNumBots(void *ev) { return ((Ensemble *)ev)->members[0]->bot->getParams().nCells; }

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
