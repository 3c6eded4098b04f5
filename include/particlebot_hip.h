/*
 * particlebot_hip.h -- C-ABI of libparticlebot_hip.so (MI355X / gfx950).
 *
 * Two seams:
 *
 *  (1) The reference's own `extern "C"` device boundary, particlebot.cuh:15-121 (bodies in
 *      particlebot_cuda.cu:26-384).  Same names, same argument order and meaning, same
 *      "void + print + exit(EXIT_FAILURE) on error" convention (include/helper_cuda.h:1000-1029).
 *      Each declaration cites the reference line it replaces.  One global parameter block, default
 *      stream, caller owns every buffer: exactly the reference's contract.
 *
 *  (2) `pbSim*`: the resident, fused engine (new).  Per-instance parameters (many simulations per
 *      process), status-returning, state kept cell-sorted in HBM between re-sorts, one fused kernel
 *      per timestep.  This is what Particlebot::update drives by default and what bench.py times.
 *
 * Plain pointers and sizes only; no C++ or framework types cross this boundary.
 */
#ifndef PARTICLEBOT_HIP_H
#define PARTICLEBOT_HIP_H

#include <stddef.h>

#include "particlebot_kernel.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Opaque handle standing in for `struct cudaGraphicsResource` (particlebot.cuh:25-30).  In the
 * headless build a "GL buffer object" is a plain device buffer created with pbCreateBuffer(). */
struct pbGraphicsResource;

/* Phase-noise generators.  PB_RNG_COUNTER (default) is this project's own counter-based generator
 * (seed, bot, draw) -> N(0,1), bit-identical between the HIP kernels and the CPU oracle.
 * PB_RNG_XORWOW_CURAND restates cuRAND's published XORWOW algorithm (curand_init(seed, bot, 0) +
 * curand_normal; particlebot_kernel_impl.cuh:36-51): integer stream cuRAND-compatible by construction,
 * unverifiable in an image without CUDA.  PB_RNG_XORWOW_ROCRAND is the same generator with rocRAND's
 * seeding constants, bit-identical to rocrand_device::xorwow_engine (csrc/pb_xorwow.hpp). */
enum { PB_RNG_COUNTER = 0, PB_RNG_XORWOW_CURAND = 1, PB_RNG_XORWOW_ROCRAND = 2 };

/* Per-bot RNG state crossing the boundary where the reference passes `curandState*`
 * (particlebot.cuh:73-78, particlebot.h:98): 48 bytes, the size of cuRAND's curandStateXORWOW, so a
 * caller that allocates sizeof(curandState) * nCells gets the same bytes. */
typedef struct pbRngState {
  unsigned d;           /* XORWOW: Weyl sequence value          | counter generator: the seed */
  unsigned v[5];        /* XORWOW: the five xorshift words      | counter generator: v[0] = draws so far */
  int boxmuller_flag;   /* XORWOW: nonzero while boxmuller_extra holds the second normal of a pair */
  int kind;             /* PB_RNG_* */
  float boxmuller_extra;
  float reserved[3];
} pbRngState;

/* Generator that the legacy boundary's curand_setup() initialises (process-wide, like the reference's
 * one global parameter block); PB_RNG_COUNTER until changed.  Returns 0, or nonzero for a bad kind. */
int pbSetRngKind(int kind);
int pbGetRngKind(void);

/* ------------------------------------------------------------------------------------------ */
/* (1) reference device boundary                                                               */
/* ------------------------------------------------------------------------------------------ */

void cudaInit(int argc, char **argv);   /* particlebot.cuh:18  ; particlebot_cuda.cu:29-41  */
void cudaGLInit(int argc, char **argv); /* main.cpp:97         ; particlebot_cuda.cu:43-47  */

/* particlebot.cuh:20 declares `int size` but particlebot_cuda.cu:49 defines `size_t size`
 * (latent ABI mismatch in the reference); the definition's size_t is what is exported here. */
void allocateArray(void **devPtr, size_t size);
void freeArray(void *devPtr); /* particlebot.cuh:21 */
void threadSync(void);        /* particlebot.cuh:23 */

/* particlebot.cuh:25-26.  If `res` is non-NULL the device pointer is taken from the mapped buffer
 * object instead of `device` (particlebot_cuda.cu:97-113). */
void copyArrayFromDevice(void *host, const void *device, struct pbGraphicsResource **res, int size);
void copyArrayToDevice(void *device, const void *host, int offset, int size);

/* particlebot.cuh:27-30 (GL interop in the reference; plain device buffers here) */
void registerGLBufferObject(uint vbo, struct pbGraphicsResource **res);
void unregisterGLBufferObject(struct pbGraphicsResource *res);
void *mapGLBufferObject(struct pbGraphicsResource **res);
void unmapGLBufferObject(struct pbGraphicsResource *res);

/* Headless replacements for the GL calls Particlebot makes on its buffer objects
 * (glGenBuffers+glBufferData particlebot.cpp:871-880; glBufferSubData :843,:862;
 * glDeleteBuffers :909-912).  Buffers are zero-filled on creation. */
uint pbCreateBuffer(size_t size);
void pbBufferSubData(uint vbo, size_t offset, size_t size, const void *data);
void pbDeleteBuffer(uint vbo);

void setParameters(SimParams *hostParams); /* particlebot.cuh:33 ; particlebot_cuda.cu:111-123 */
/* Extension: half-extent of the integrator's wall clamp.  The reference hard-codes 64
 * (particlebot_kernel_impl.cuh:75-97); that is the default. */
void pbSetWallHalfExtent(float half);

/* particlebot.cuh:35-40 */
void integrateSystem(float *pos, float *vel, float *rad, float deltaTime, uint nCells, float time);
/* particlebot.cuh:42-45 */
void calcHash(uint *gridParticlebotHash, uint *gridParticlebotIndex, float *pos, int nCells);
/* particlebot.cuh:47-58 */
void reorderDataAndFindCellStart(uint *cellStart, uint *cellEnd, float *sortedPos, float *sortedVel,
                                 float *sortedRad, uint *gridParticlebotHash,
                                 uint *gridParticlebotIndex, float *oldPos, float *oldVel,
                                 float *oldRad, uint nCells, uint numCells);
/* particlebot.cuh:63-71 */
void updateRad_light_wave(float *pos, float *absForce_a, float *absForce_r, float *rad, float *phase,
                          float time, float deltaTime, int *dead, int nCells);
/* particlebot.cuh:73 */
void curand_setup(pbRngState *state, int N);
/* particlebot.cuh:75-78 */
void add_normal_noise(pbRngState *state, float *val, float std, int N);
/* particlebot.cuh:80-85 */
void updatePhase(float *pos, float *phase, float spacing, float max_d, float min_d, int nCells);
/* particlebot.cuh:89-94 (display only: feeds the renderer; exported, does nothing) */
void updateCol(float *rad, float *col, int nCells, float *pos, float *phase, int *dead);
/* particlebot.cuh:96-107 */
void collide(float *newVel, float *absForce_a, float *absForce_r, float *sortedPos, float *sortedVel,
             float *sortedRad, uint *gridParticlebotIndex, uint *cellStart, uint *cellEnd, uint nCells,
             uint numCells, float deltaTime);
/* particlebot.cuh:109-115 (display only: centroid trail; exported, does nothing) */
void calcCOG(float *pos, float *temppos, float *temppos1, int nCells, float time, int hist_steps,
             float hist_int);
/* particlebot.cuh:117-119 */
void sortParticlebots(uint *dGridParticlebotHash, uint *dGridParticlebotIndex, uint nCells);

/* ------------------------------------------------------------------------------------------ */
/* (2) resident fused engine                                                                   */
/* ------------------------------------------------------------------------------------------ */

typedef struct pbSim pbSim;

enum {
  PB_OK = 0,
  PB_ERR_HIP = 1,      /* a HIP runtime call failed; see pbGetLastErrorString() */
  PB_ERR_ARG = 2,      /* bad argument (null handle, grid not a power of two >= 8, ...) */
  PB_ERR_NO_DEVICE = 3 /* no gfx950 device visible */
};

typedef struct pbSimStats {
  unsigned long long steps;          /* timesteps executed */
  unsigned long long fused_launches; /* collide + next radius/integrate in one kernel */
  unsigned long long plain_launches; /* collide-only kernel */
  unsigned long long state_launches; /* stand-alone radius+integrate kernel */
  unsigned long long resorts;
  unsigned long long phase_updates;
  unsigned long long resident_launches; /* multi-step one-workgroup-per-simulation kernel */
} pbSimStats;

const char *pbGetLastErrorString(void);

/* The calling THREAD's current HIP device.  A batch lives on the device that was current when it was created
 * (pbSimCreate*), and HIP's current device is per host thread -- a new thread starts on device 0 -- so a program that
 * creates batches from worker threads (one process per GPU, LOCAL_RANK > 0) sets the device in each of them first.
 * (Every other pbSim* call switches to its batch's device by itself.)  The ensemble pipeline does this for the thread
 * that calls pbEnsemblePipelineRun. */
int pbGetDevice(int *device);
int pbSetDevice(int device);
/* "0000:c1:00.0"-style PCI address of a device (cap >= 13): the key under /sys/bus/pci/devices whose numa_node /
 * local_cpulist say which host cores sit next to the GPU (pbHostGetResources, include/particlebot_ensemble.h). */
int pbDevicePciBusId(int device, char *out, int cap);

/* Creates a simulation of params->nCells bots on the current device.  The parameter block is
 * copied (obstacle arrays included, at most PB_MAX_OBSTACLES each).  wallHalf <= 0 selects the
 * reference's 64.  gridSize must be a power of two >= 8 in each dimension.  State starts zeroed
 * with time = 0; absForce_a/r are zero (the reference reads them uninitialised at step 0). */
int pbSimCreate(pbSim **out, const SimParams *params, float wallHalf);
void pbSimDestroy(pbSim *sim);

/* A BATCH of nsims independent simulations (an ensemble: seeds, sweep points) stepped together by
 * the same kernel launches: params is an array of nsims blocks (at most 2^32 - 32 bots in total: slots are 32-bit;
 * from 2^28 bots on the throughput sweep switches from 32-bit to 64-bit byte offsets).  All
 * members must share nCells,
 * the grid, max_time, phase_update_interval, control and payload mode (nDead == -1 or not);
 * everything else (seed, light, obstacles, physics constants) may differ.  pbSimStep & co. advance
 * every member; the *Of functions address one member; pbSimCreate is the nsims == 1 case and the
 * un-suffixed state functions address member 0. */
int pbSimCreateBatch(pbSim **out, const SimParams *params, int nsims, float wallHalf);
int pbSimBatchSize(pbSim *sim, unsigned *nsims, unsigned *nbots);
int pbSimSetStateOf(pbSim *sim, unsigned member, const float *pos, const float *vel, const float *rad,
                    const float *phase, const int *dead);
/* The same for the bots [start, start + count) of the ORIGINAL order only (the arrays hold count
 * entries); every other bot keeps its device state.  This is what Particlebot::setArray(array, data,
 * start, count) needs once the simulation has stepped (particlebot.cpp:834-867 touches only that
 * range). */
int pbSimSetStateRangeOf(pbSim *sim, unsigned member, unsigned start, unsigned count, const float *pos,
                         const float *vel, const float *rad, const float *phase, const int *dead);
int pbSimGetStateOf(pbSim *sim, unsigned member, float *pos, float *vel, float *rad, float *phase, int *dead,
                    float *absForce_a, float *absForce_r);
/* Exact checkpoints.  Between re-sorts the cell lists are STALE by design (the reference re-hashes
 * only every sort_interval), so a bit-identical resume needs, besides the state arrays, the layout:
 * orig[i] = original index of the bot in slot i, keys[i] = the cell hash slot i was filed under at
 * the last sort (ascending).  *sorted is 0 if the simulation has not been sorted yet.  After
 * pbSimSetLayoutOf for every member, restore the state with pbSimSetStateOf + pbSimSetForcesOf,
 * pbSimSetTime and pbSimSetPhaseDraws. */
int pbSimGetLayoutOf(pbSim *sim, unsigned member, unsigned *orig, unsigned *keys, int *sorted);
int pbSimSetLayoutOf(pbSim *sim, unsigned member, const unsigned *orig, const unsigned *keys);
int pbSimSetForcesOf(pbSim *sim, unsigned member, const float *absForce_a, const float *absForce_r);
/* centre of mass of every member: cxcy[2*k], cxcy[2*k+1]; reduced on the device in a fixed order */
int pbSimCentroids(pbSim *sim, double *cxcy);
/* The reference's own centroid SUMS of every simulation of the batch (sumxy[2k], sumxy[2k+1]): fp32, the bots added
 * serially in original order as dumpParticlebot does (particlebot.cpp:335-338) -- bit for bit what its CSV's
 * "Centroid X, Centroid Y" columns are divided from (pbSimCentroids above is the accurately rounded mean). */
int pbSimCentroidSums(pbSim *sim, float *sumxy);

/* Host arrays in ORIGINAL bot order; NULL pointers leave that array unchanged.
 * pos, vel: 2*n floats; rad, phase: n floats; dead: n ints. */
int pbSimSetState(pbSim *sim, const float *pos, const float *vel, const float *rad, const float *phase,
                  const int *dead);
/* Same layout; also absForce_a / absForce_r (n floats each).  NULL pointers are skipped. */
int pbSimGetState(pbSim *sim, float *pos, float *vel, float *rad, float *phase, int *dead,
                  float *absForce_a, float *absForce_r);
int pbSimSetTime(pbSim *sim, float time);
int pbSimGetTime(pbSim *sim, float *time);
/* Number of phase-noise draws made so far (the k of the counter RNG); settable for resume. */
int pbSimGetPhaseDraws(pbSim *sim, unsigned *draws);
int pbSimSetPhaseDraws(pbSim *sim, unsigned draws);

/* Runs up to nsteps iterations of Particlebot::update's schedule (particlebot.cpp:170-300) minus
 * the host-side dead-bot draw (:178-194, done by the caller through pbSimSetState): phase update
 * every phase_update_interval, radius actuation, integration, re-hash + stable sort every
 * sort_interval, neighbour forces, fp32 `time += dt`.  Stops before a step whose start time
 * exceeds max_time (the reference calls exit(0) there).  *steps_done receives the count.
 * Asynchronous with respect to the host except at phase updates (4-byte read-back). */
int pbSimStep(pbSim *sim, float deltaTime, float sort_interval, int nsteps, int *steps_done);
/* Same, bracketed by HIP events on the simulation's stream; *elapsed_ms is device time. */
int pbSimStepTimed(pbSim *sim, float deltaTime, float sort_interval, int nsteps, int *steps_done,
                   float *elapsed_ms);
/* Same, and *wall_ms (if not NULL) = the HOST's clock over the region: from entry -- the caller has synchronised, the
 * stream is idle -- through every launch to the drained stream (the end event has completed; hipStreamQuery confirms).  The host polls for the
 * end of the region instead of sleeping on an interrupt, so for a region of a few milliseconds wall and device time
 * differ only by the dispatch and completion latencies (~10 us; PB_TIMED_TRACE=1 prints where the host's time went).
 * bench.py's `value` is over this clock, its `roofline` over *elapsed_ms, of the same launches. */
int pbSimStepTimedWall(pbSim *sim, float deltaTime, float sort_interval, int nsteps, int *steps_done,
                       float *elapsed_ms, double *wall_ms);
int pbSimSynchronize(pbSim *sim);

/* Centre of mass, reduced on the device in a fixed order (double accumulation). */
int pbSimCentroid(pbSim *sim, double *cx, double *cy);
int pbSimGetStats(pbSim *sim, pbSimStats *stats);
/* 0: stale cell lists re-sorted every sort_interval (reference behaviour, default);
 * 1: re-sort every step (never the default: it changes trajectories). */
int pbSimSetResortEveryStep(pbSim *sim, int on);

/* How the phase update finds the distance from the light to the nearest bot (particlebot.cpp:214-228: a host
 * loop of powf(powf(dx,2) + powf(dy,2), 0.5f) over every position).  0 (default): the device reduces
 * min(dx*dx + dy*dy), 4 bytes per simulation come back and the host takes powf(., 0.5f) -- the reference's value
 * provided the host libm has powf(x,2) == x*x for every float and a non-decreasing powf(., 0.5f), which
 * pbHostLibmCheck (libparticlebot_host.so; tests/test_libm_pin.py) verifies exhaustively.  1: the reference's own
 * loop on the host over all positions (8 n bytes per simulation): no assumption.  pbSetMinDistanceMode sets the
 * default of batches created afterwards (process-wide); the environment variable PB_MIN_DISTANCE_MODE (0 or 1) seeds
 * that default for a whole process tree, read once at the first batch.  Precedence: a batch's own
 * pbSimSetMinDistanceMode > pbSetMinDistanceMode > the environment > 0.  pbGetMinDistanceMode returns the default a
 * batch created now would get. */
int pbSimSetMinDistanceMode(pbSim *sim, int mode);
int pbSetMinDistanceMode(int mode);
int pbGetMinDistanceMode(void);

/* The smallest non-negative float x with sqrtf(x) >= c (0 when c <= 0 or NaN): `length(v) < c` of the static-friction
 * hold (particlebot_impl.cuh:809-811) is decided as `dot(v,v) < pbHostSqrtThreshold(c)`, the same decision for every
 * input because sqrtf is correctly rounded and therefore monotone.  Host-only (no GPU needed); exported for
 * tests/test_sqrt_threshold.py. */
float pbHostSqrtThreshold(float c);

/* Force-kernel variant of a simulation: 0 reference-shaped branches, 1 branch-free, 2 branch-free
 * with the fast exact sqrt/division forms (default; falls back to 1 when the simulation's
 * constants are outside their proven domain).  Variants 0-2 give bit-identical results.
 * 3 = STREAMLINED arithmetic (opt-in, NOT bit-identical): distance and unit vector from one
 * reciprocal square root, 1/gap^2 from one reciprocal, |F_attr| from its coefficient, FMA
 * contraction, a bot's contact terms added after its attraction terms.  About twice as fast;
 * agrees with variants 0-2 to ~1e-7 relative per 10 steps except where a bot lands on the other
 * side of one of the reference's force-law discontinuities.  It only
 * replaces the throughput form (batches above 131072 bots, or lanes-per-bot forced to 1); smaller
 * batches keep running the exact kernels. */
int pbSimSetForceVariant(pbSim *sim, int variant);
/* The two per-bot magnitude sums of the force kernel, absForce_a (Sum|F_attr|) and absForce_r
 * (Sum|F_rep|), are private scratch arrays of the reference (no getArray case, not in the dump);
 * their only reader is the next step's radius actuation, which reads absForce_a solely under
 * `if (params.constrained_contraction)` (particlebot_kernel_impl.cuh:167-169; 0 by default and in all
 * shipped examples).  mode 0 (default): absForce_a is maintained only when some member of the batch
 * has constrained_contraction set -- otherwise it is a dead value, the branch-free force kernels
 * (variants 1 and 2, every lanes-per-bot form, the resident kernel) do not compute it (one exact square
 * root per candidate pair less) and pbSimGetState returns NaN for it.  mode 1: always maintained (valid from the next step on).  Positions,
 * velocities, radii, phases and absForce_r do not depend on the mode. */
int pbSimSetForceSums(pbSim *sim, int mode);
/* Lanes per bot in the per-step force kernel: 1 = throughput form (one bot per lane); 2, 4, 8, 16, 32,
 * 64 = that many adjacent lanes share a bot's neighbour list and add the terms in list order (batches
 * too small to fill the chip: the serial neighbour loop is the limit); 0 = automatic (default:
 * 64 up to 1280 bots in the batch, 32 up to 2560, 16 up to 8192, 8 up to 40960, 4 up to 131072, else 1).
 * Results do not depend on it. */
int pbSimSetLanesPerBot(pbSim *sim, int lanes);
/* Resident form for simulations of at most 1024 bots: one workgroup per simulation keeps the state
 * in registers/LDS and runs every timestep up to the next re-sort, phase update or end of the
 * pbSimStep call in ONE launch.  0 = automatic (default: a cost model fitted to MI355X measurements
 * picks it for a lone simulation of ~100 bots and for ensembles of many small ones),
 * 1 = never, 2 = whenever the simulation fits.  Results do not depend on it. */
int pbSimSetResident(pbSim *sim, int mode);

/* What the next pbSimStep will launch for this batch (after the automatic choices): the force variant
 * as set and the kind of kernel that really runs (0-2 exact, 3 streamlined; 2 falls back to 1 when the
 * constants are outside the fast forms' domain), the effective lanes per bot of the per-step kernel,
 * whether the resident multi-step kernel is used, and the phase-noise generator.  bench.py echoes it
 * into its JSON line so a reader can see which kernel a number belongs to. */
/* Phase-noise generator of a batch (PB_RNG_*; default PB_RNG_COUNTER).  Selecting an XORWOW kind builds
 * one 48-byte state per bot -- curand_init(member's seed, bot, 0): the 2^67-step subsequence skip is a
 * 160x160 GF(2) jump per set bit of the bot index -- and restarts the draw counter.
 * pbSimSetPhaseDraws(k) afterwards replays k draws (exact checkpoint resume). */
int pbSimSetRng(pbSim *sim, int kind);
/* The generator states of one member in ORIGINAL bot order (n x pbRngState); PB_ERR_ARG with the
 * counter generator, which has none. */
int pbSimGetRngStatesOf(pbSim *sim, unsigned member, pbRngState *states);

/* The FORMS of the exact per-step force kernel (csrc/pb_force.hip): one table shared by the dispatch and by
 * the parity tests, so that every shape that can run is enumerable.  flat 0 = reference-shaped branches
 * (force variant 0), 1 = branch-free (variants 1 and 2; whether the fast exact forms are used is decided at
 * run time per wave); lanes_per_bot as pbSimSetLanesPerBot; attraction_sums 0 = the dead-sum form (no
 * Sum|F_attr|, pbSimSetForceSums); offsets64 1 = 64-bit byte offsets in the throughput sweep (what batches of
 * 2^28 bots and more run).  Every row exists for both payload modes.  All rows give bit-identical results.
 * pbSimSelectForceForm pins a batch to row `index` (and to per-step launches: the resident form is switched
 * off); -1 returns to the automatic choices.  A dead-sum row is refused (PB_ERR_ARG) for a batch in which a
 * member reads absForce_a (constrained_contraction). */
typedef struct pbForceForm {
  int flat;
  int lanes_per_bot;
  int attraction_sums;
  int offsets64;
} pbForceForm;
int pbForceFormCount(void);
int pbForceFormGet(int index, pbForceForm *form);
int pbSimSelectForceForm(pbSim *sim, int index);
/* The per-step force kernel this batch launches next, named as rocprofv3 names it (template arguments and
 * argument types, no "void", no namespace): "k_force<false, true, 1, 1, false, true>(PbDevParams const*, ...)".
 * bench.py matches it against the signature recorded in profiles/latest_traffic*.json before quoting that
 * profile's counters.  PB_ERR_ARG when `cap` is too small. */
int pbSimForceKernelName(pbSim *sim, char *buf, size_t cap);
/* The same for row `index` of the forms table (no device needed). */
int pbForceFormKernelName(int index, int payload, char *buf, size_t cap);

/* How the streamlined kernel (force variant 3) visits a bot's 25-cell stencil: 0 row by row (the wave runs the longest
 * row of its 64 lanes, five times), 1 flattened (every lane walks its own five ranges back to back; the wave runs its
 * longest list), -1 (default) chosen for the batch at every re-sort from the trip counts of both (flattened when it
 * saves >= 7 % of the trips: random blobs -- BASELINE configs[4] steps 7-9 % faster --, not the bench lattice).  Same
 * candidates in the same order: the results do not depend on it, bit for bit.  No effect on the exact kernels.
 * pbSimGetStreamWalkTrips: the two trip counts of the last choice (0, 0 before one was made). */
int pbSimSetStreamWalk(pbSim *sim, int mode);
int pbSimGetStreamWalkTrips(pbSim *sim, unsigned long long *row_by_row, unsigned long long *flattened);

typedef struct pbSimConfig {
  int force_variant;
  int force_kind;
  int lanes_per_bot;
  int resident;
  int fast_math_ok;
  int payload;
  int rng; /* 0 PB-RNG v1 (default), 1 cuRAND-compatible XORWOW */
  int offsets64; /* 1: the throughput sweep runs with 64-bit byte offsets (batches of 2^28 bots and more) */
  int attraction_sums; /* 1: absForce_a is maintained (pbSimSetForceSums) */
  int dead_sum_form;   /* 1: the force kernel that runs is a form without Sum|F_attr| */
  int stream_walk;     /* 1: the streamlined kernel runs and walks its stencil flattened (pbSimSetStreamWalk) */
} pbSimConfig;
int pbSimGetConfig(pbSim *sim, pbSimConfig *cfg);

/* Shader clock under load (diagnostic for the roofline report).  Begin launches ONE sleeping wave on
 * a stream of its own that spans `seconds` of the 100 MHz real-time counter; End waits for it and
 * returns shader cycles / real time in MHz over that span.  Run the workload in between: the figure
 * is the clock the chip holds under THAT load (it lowers its clock under dense VALU work). */
typedef struct pbClockSample pbClockSample;
int pbClockSampleBegin(pbClockSample **out, double seconds);
int pbClockSampleEnd(pbClockSample *sample, double *mhz, double *seconds_sampled);

/* On-device check that the fast exact forms equal the compiler's IEEE sqrtf and division: every
 * float in the sqrt domain (pbSqrtFast, and the root of the one-transcendental pair geometry
 * pbDistUnitFast), and div_samples sampled (numerator, numerator, denominator) triples inside the
 * division domain (pbDiv2Fast; pbDistUnitFast's unit vector on sampled (d2, dx, dy)).  Reports how many
 * values were checked and how many differed.  (tools/rsq_form_test.hip is the exhaustive check of
 * pbDistUnitFast's quotients: all 2^47 mantissa pairs.) */
int pbSelfTest(unsigned long long div_samples, unsigned long long *sqrt_checked,
               unsigned long long *sqrt_mismatches, unsigned long long *div_checked,
               unsigned long long *div_mismatches);

/* The EXHAUSTIVE form of the pair-geometry check: pbDistUnitFast (as the kernels call it) for every d2 of
 * `slices` consecutive slices (of 64) of [1, 4) -- together all 2^24 mantissa x exponent-parity cases --
 * against every numerator mantissa (2^23): the root against sqrtf and both quotients against IEEE division.
 * `checked` counts (d2, numerator) pairs (2^41 per slice, ~1.5 s of one MI355X); all 64 slices are the proof
 * DESIGN.md section 4 quotes (2^47 pairs, 0 mismatches). */
/* The static-friction hold's squared-length threshold for the constant c (pbHostSqrtThreshold) against the device's
 * IEEE sqrtf: `sqrtf(x) < c` and `x < T(c)` compared for EVERY non-negative float bit pattern x (2^31, NaNs included). */
int pbSelfTestHoldThreshold(float c, unsigned long long *checked, unsigned long long *mismatches);
int pbSelfTestPairGeometry(unsigned first_slice, unsigned slices, unsigned long long *checked,
                           unsigned long long *mismatches);
/* The same for pbDiv2Fast (the division of the attraction term by gap^2): every denominator mantissa of
 * `slices` of the 64 slices of [1, 2) against every numerator mantissa: 2^40 quotients per slice, 2^46 in all. */
int pbSelfTestDivision(unsigned first_slice, unsigned slices, unsigned long long *checked,
                       unsigned long long *mismatches);

#ifdef __cplusplus
}
#endif
#endif /* PARTICLEBOT_HIP_H */
