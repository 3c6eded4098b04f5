/*
 * particlebot.h -- class Particlebot, headless, on MI355X.
 *
 * Keeps the public surface of the reference's class (particlebot.h:14-67): constructor from
 * SimParams, update, reset, getArray/setArray, the buffer getters, dumpParticlebot, loadFromFile,
 * getWorldOrigin/getCellSize.  What changed underneath:
 *   - no OpenGL: the VBO getters return 0 / nullptr unless the legacy engine is selected, in which
 *     case they return the headless buffer ids / device pointers;
 *   - the per-step work runs in the resident fused engine (pbSim*, include/particlebot_hip.h) by
 *     default; PB_ENGINE=legacy (or Engine::Legacy) replays the reference's own call sequence
 *     through the `extern "C"` device boundary instead, kernel by kernel;
 *   - absForce_a / absForce_r start at zero (the reference reads them uninitialised at step 0);
 *   - phase noise uses this build's counter RNG, not cuRAND (parity unpinned).
 */
#ifndef PARTICLEBOT_H
#define PARTICLEBOT_H

#include <cstdio>
#include <string>
#include <vector>

#include "particlebot_hip.h"
#include "particlebot_kernel.h"

/* glibc's rand()/srand() (the TYPE_3 additive-feedback generator of random_r) as an object.  The
 * reference draws the placement and the dead-bot set from the process-global libc stream seeded
 * by srand(params.seed) in main (main.cpp:929).  A private stream with the same arithmetic gives
 * the same numbers, is not disturbed by other users of rand() in the process (the HIP runtime
 * draws from it while initialising) and lets many simulations live in one process. */
class PbLibcRand {
 public:
  explicit PbLibcRand(unsigned seed = 1) { reseed(seed); }
  void reseed(unsigned seed);
  int next(); /* == rand(): 31 bits */
  static constexpr int kMax = 2147483647; /* RAND_MAX */
  /* full generator state (for checkpoints): 34 table words + the two cursors */
  void getState(int out[36]) const;
  void setState(const int in[36]);

 private:
  int r[34];
  int f, b; /* front / rear indices into r[3..33] */
};

class Particlebot {
 public:
  /* HostOnly: no device objects at all -- placement, dead-bot draw and host mirrors only; used by
   * the ensemble driver, which keeps every member's device state in one batched pbSim */
  enum class Engine { Fused, Legacy, HostOnly };

  /* particlebot.h:17 -- engine taken from $PB_ENGINE ("legacy" or "fused", default fused);
   * wall half-extent 64 as in the reference */
  Particlebot(SimParams params);
  Particlebot(SimParams params, Engine engine, float wallHalf);
  ~Particlebot();

  /* particlebot.h:21.  One timestep (particlebot.cpp:170-300).  Like the reference, exits the
   * process with status 0 once time > max_time -- unless setExitOnMaxTime(false), after which it
   * just returns and finished() reports it. */
  void update(float deltaTime, float sort_interval);
  /* Extension: up to nsteps timesteps in one call (lets the fused engine keep one kernel per
   * step); never exits the process; returns the number of steps run. */
  int advance(float deltaTime, float sort_interval, int nsteps);
  void reset(); /* particlebot.h:22 */

  /* particlebot.h:24-25.  getArray returns the host mirror after copying n*width floats (the
   * reference copies nCells*4 floats from 1- and 2-float buffers, particlebot.cpp:830: not
   * replicated).  PHASE and DEAD are supported too. */
  float *getArray(ParticlebotArray array);
  void setArray(ParticlebotArray array, const float *data, int start, int count);

  unsigned int getCurrentReadBuffer() const { return posVbo; }
  unsigned int getColorBuffer() const { return colorVBO; }
  unsigned int getRadBuffer() const { return radVbo; }
  void *getCudaPosVBO() const { return (void *)cudaPosVBO; }
  void *getCudaColorVBO() const { return (void *)cudaColorVBO; }
  void *getCudaRadVBO() const { return (void *)cudaRadVBO; }

  /* particlebot.h:54-59 */
  void dumpParticlebot(uint start, uint count, FILE *fp, float dump_interval, uint testing, float light_x,
                       float light_y);
  void loadFromFile(uint start, uint count, FILE *fp, float dump_interval);

  float2 getWorldOrigin() { return params.worldOrigin; }
  float2 getCellSize() { return params.cellSize; }

  /* ---- extensions ---- */
  void setExitOnMaxTime(bool on) { exitOnMaxTime = on; }
  bool finished() const { return time > params.max_time; }
  float getTime() const { return time; }
  void setTime(float t);
  /* true if a dump row is due at the current time (the test at particlebot.cpp:309) */
  bool dumpDue(float dump_interval) const;
  /* steps until the next host-side event (dump row, dead-bot draw), for batching; >= 1 */
  int stepsUntilHostEvent(float deltaTime, float dump_interval, int maxSteps) const;
  Engine engine() const { return engineKind; }
  const SimParams &getParams() const { return params; }
  int *getDeadArray();
  /* quiet the reference's per-bot "Placing %d th disc" chatter (particlebot.cpp:645) */
  static void setVerbosePlacement(bool on);
  /* lattice pitch used by CONFIG_HEX placement; <= 0 selects the reference's 2*min_radius */
  void setHexSpacing(float pitch) { hexSpacing = pitch; }
  /* Extension: place the bots on a centred square lattice (pitch as setHexSpacing) instead of what
   * params.config says.  Unlike any hexagonal packing, four contacts per bot are numerically stable
   * under the reference's parameters: the O(N) placement for very large arenas. */
  void setSquareLattice(bool on) { squareLattice = on; }
  /* Extension (`pb_placement fastblob`): a random blob grown by the reference's rule (random anchor,
   * random angle, pivot to contact; particlebot.cpp:612-748) with the anchors drawn from the discs
   * that still have room: O(N), for blobs the reference's O(N^1.5) loop cannot reach (10^6 bots in
   * seconds).  Same kind of blob, different random stream: NOT the reference's placement. */
  void setFastBlob(bool on) { fastBlob = on; }
  /* Extension: phase-noise generator, PB_RNG_* of particlebot_hip.h (default PB_RNG_COUNTER; the
   * `pb_rng` key of a .cfg: "pbrng", "curand" or "rocrand").  Re-initialises the per-bot generator
   * states, as curand_setup does at construction (particlebot.cpp:165); call it before the first step. */
  void setRng(int kind);
  int rngKind() const { return rngKindV; }
  /* Extension (`pb_force_variant` key): the force kernel of the fused engine -- 0/1/2 the exact forms (2 the default),
   * 3 the opt-in tolerance kernel (not bit-identical; as close to the reference as an FMA-contracted build of its own
   * arithmetic).  No effect on the Legacy engine. */
  void setForceVariant(int variant);
  pbSim *engineHandle() { return sim; }
  /* host mirrors in original bot order (valid after reset(); refreshed by getArray/dump) */
  const float *hostPositions() const { return hPos; }
  const float *hostVelocities() const { return hVel; }
  const float *hostRadii() const { return hRad; }
  const float *hostPhases() const { return hphase; }
  const int *hostDead() const { return hDead; }
  /* true when the step starting at the current time is the one that draws the dead bots
   * (particlebot.cpp:178) */
  bool deadDrawDue(float deltaTime) const {
    return params.nDead > 0 && time >= params.time_to_dead && time < params.time_to_dead + deltaTime;
  }
  /* draws the dead set into the host mirror (and the engine, if any); returns the mirror */
  const int *drawDeadBotsNow() {
    drawDeadBots();
    return hDead;
  }
  /* Exact checkpoint (extension; fused engine): time, phase-noise draw counter, the private
   * generator's state, every state array INCLUDING phase / dead / absForce_a / absForce_r, and the
   * stale slot layout (which the reference's CSV resume loses): a run resumed from it continues
   * bit-identically.  Both return false on I/O or format errors. */
  bool saveCheckpoint(FILE *fp);
  bool loadCheckpoint(FILE *fp);
  /* Headless stand-in for the reference's display()/video path (main.cpp:354-470, colours from
   * updateCol_k, particlebot_kernel_impl.cuh:401-443): writes a binary PPM (P6) of width x height
   * pixels showing the square |x - centerX|, |y - centerY| <= halfExtent seen from above, x
   * mirrored as the reference draws it.  Bots are discs of their current radius: dead bots black,
   * live ones R = 30, G = 20 + 180 ((max_r - r)/(max_r - min_r))^2, B = 30 + 180
   * sqrt((r - min_r)/(max_r - min_r)); obstacles grey, the light a yellow disc of lightRadius.
   * (The display_shadow tint is not drawn.)  Returns false on I/O errors. */
  bool writeFramePPM(const char *path, int width, int height, float centerX, float centerY, float halfExtent,
                     float lightRadius = 0.25f);
  /* HostOnly engines follow an external clock */
  void setHostTime(float t) { time = t; }
  /* The private placement / dead-draw generator's state and the host mirrors of a HostOnly instance: what an
   * ensemble checkpoint saves and restores for a member whose device state lives in a batched pbSim. */
  void getHostRngState(int out[36]) const { rng.getState(out); }
  void setHostRngState(const int in[36]) { rng.setState(in); }
  void restoreHostMirrors(const float *pos, const float *vel, const float *rad, const float *phase, const int *dead);
  /* Extension (ensemble pipeline: members of a sweep that share seed and geometry are placed ONCE).  placementKey():
   * every input reset() reads -- the seed of the private generator, nCells, min/max_radius, the payload flag
   * (nDead == -1) and radFactor with it, config and the pb_placement extensions, the lattice pitch, Nx as given, and
   * the grid the placement bins into (gridSize, worldOrigin, cellSize).  nDead >= 0 is NOT part of it: the reference's
   * placement reads nDead only for the payload (particlebot.cpp:731, 786).  Two instances with equal keys place
   * identically, so one may take the other's result: exportPlacement() right after reset() captures positions, radii,
   * phases, dead flags, Nx and the generator's state AFTER the placement draws (the dead draw continues that stream,
   * particlebot.cpp:178-194); importPlacement() installs it in place of reset().  HostOnly engines. */
  struct Placement {
    std::vector<float> pos, rad, phase;
    std::vector<int> dead;
    int rng[36];
    unsigned configX, configY, Nx;
  };
  std::string placementKey() const { return placementKeyOf(params, hexSpacing, squareLattice, fastBlob); }
  static std::string placementKeyOf(const SimParams &params, float hexSpacing, bool squareLattice, bool fastBlob);
  void exportPlacement(Placement &out) const;
  bool importPlacement(const Placement &in);

 protected:
  void _initialize();
  void _finalize();
  void initGrid(uint2 size, float spacing, float jitter, uint numParticles);
  void initHexGrid(uint numParticles, float spacing);
  void placeRandom();
  void placeFastBlob();
  void drawDeadBots();
  void pullState(bool pos, bool vel, bool rad);
  void legacyUpdate(float deltaTime, float sort_interval);

  /* host mirrors (original bot order) */
  std::vector<float> hPosV, hVelV, hRadV, hPhaseV, hFreqV;
  std::vector<int> hDeadV;
  float *hPos = nullptr, *hVel = nullptr, *hRad = nullptr, *hphase = nullptr;
  int *hDead = nullptr;

  /* fused engine */
  pbSim *sim = nullptr;

  /* legacy engine: the reference's device buffers (particlebot.h:93-123) */
  float *dVel = nullptr, *dAbsForce_a = nullptr, *dAbsForce_r = nullptr, *dphase = nullptr;
  int *dDead = nullptr;
  pbRngState *dState = nullptr;
  float *dSortedPos = nullptr, *dSortedVel = nullptr, *dSortedRad = nullptr;
  uint *dGridParticleHash = nullptr, *dGridParticleIndex = nullptr, *dCellStart = nullptr, *dCellEnd = nullptr;
  struct pbGraphicsResource *posRes = nullptr, *radRes = nullptr;

  uint posVbo = 0, colorVBO = 0, radVbo = 0;
  float *cudaPosVBO = nullptr, *cudaColorVBO = nullptr, *cudaRadVBO = nullptr;

  float time = 0.0f;
  SimParams params;
  std::vector<float> obsStore; /* owns copies of the caller's obstacle arrays */
  uint2 particlebotConfigSize;
  Engine engineKind = Engine::Fused;
  float wallHalf = 64.0f;
  bool exitOnMaxTime = true;
  float hexSpacing = 0.0f;
  bool squareLattice = false;
  bool fastBlob = false;
  int rngKindV = 0; /* PB_RNG_COUNTER */
  PbLibcRand rng; /* seeded with params.seed at construction */
};

#endif /* PARTICLEBOT_H */
