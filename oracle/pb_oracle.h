/*
 * pb_oracle.h -- CPU ORACLE for the particle-robot update loop.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm for the hot path
 * (richa-batra/ParticleRobotSimulations; file:line citations below are relative to the
 * reference checkout).  It exists to CHECK the HIP implementation.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing under
 * particlerobotsimulations_amd/ links, imports or executes it.
 *
 * Pinning status: PARITY UNPINNED in the strict sense.  The reference cannot be built in this image
 * under the build rules (it needs the CUDA runtime + cuRAND headers, Thrust, GLEW, freeglut and OpenCV,
 * none of which exist here, and stand-in headers are not allowed), and it ships no tests or golden
 * vectors.  What the oracle IS checked against, bit for bit: the position snapshots in
 * tests/golden/ref_probe/ -- outputs of the unmodified reference sources run on the CPU by the survey
 * session against its own stand-in CUDA/GL headers: initial placement and
 * steps 1..20000 of the example.cfg-like run; and, for the XORWOW generator (generator 2 below),
 * rocRAND's own host-callable xorwow_engine and precomputed jump matrices (tests/test_xorwow.py).
 * Unpinned in any sense (no CUDA toolchain to check against): cuRAND's four seeding constants and
 * the device logf/__sincosf inside curand_normal (the default noise generator is this project's
 * counter RNG, orc_normal()), the CUDA fast intrinsic __powf(x,2) (restated as x*x), and nvcc's FMA
 * contraction choices (restated with no contraction; build with -ffp-contract=off).
 *
 * Arithmetic rules of the restatement (identical in the HIP kernels so that the two are
 * bit-identical by construction):
 *   device powf(x,2), __powf(x,2)  -> x*x          device powf(x,0.5f) -> sqrtf(x)
 *   host   powf(...) calls (placement, min-distance, CSV distance) stay glibc powf.
 */
#ifndef PB_ORACLE_H
#define PB_ORACLE_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_OBS 10

/* Flat restatement of SimParams (particlebot_kernel.cuh:58-120) plus the host globals that
 * main.cpp keeps beside it (main.cpp:79-87) and one extension (wallHalf). */
typedef struct OrcParams {
  uint32_t gridSizeX, gridSizeY, numCells;
  float worldOriginX, worldOriginY;
  float cellSizeX, cellSizeY;
  uint32_t nCells;
  int32_t nDead;
  float gravity, spring, damping, shear, attraction, boundaryDamping, friction;
  float massFactor, frictionFactor, radFactor, attractionFactor;
  float constraint, constraint_contraction;
  int32_t centroid_steps;
  float centroid_int, centroid_radius;
  float light_x, light_y;
  float phase_update_interval;
  int32_t control, config;
  float min_radius, max_radius, rise_period, freq;
  int32_t nobstacles;
  float x1obs[ORC_MAX_OBS], x2obs[ORC_MAX_OBS], y1obs[ORC_MAX_OBS], y2obs[ORC_MAX_OBS];
  int32_t n_cir_obstacles;
  float x_cir_obs[ORC_MAX_OBS], y_cir_obs[ORC_MAX_OBS], r_cir_obs[ORC_MAX_OBS];
  int32_t Nx;
  float phase_std;
  uint32_t seed;
  uint32_t light_shadow, testing, constrained_contraction, display_shadow;
  float time_to_dead, max_time;
  /* main.cpp globals */
  float timestep, sort_interval, dump_interval;
  float camera_x, camera_y, light_radius;
  int32_t display_interval, video_interval;
  char csv_filename[300];
  char video_filename[300];
  /* extension: wall half-extent of the integrator clamp (reference hard-codes 64,
   * particlebot_kernel_impl.cuh:75-97) */
  float wallHalf;
  /* extension: phase-noise generator.  0 = PB-RNG v1 (counter based), 1 = cuRAND-compatible XORWOW
   * (particlebot_kernel_impl.cuh:36-51), 2 = the same generator with rocRAND's seeding constants */
  int32_t rngKind;
} OrcParams;

/* main.cpp:832-911 defaults (seed is NOT time(NULL) here: it is set to 0 and must come from the
 * file or the caller). */
void orc_params_defaults(OrcParams *p);
/* main.cpp:594-816 one name/value pair, with the reference's prefix-matching quirks */
void orc_set_param(OrcParams *p, const char *name, const char *value);
/* main.cpp:918-928 parse loop; returns 0 on success, -1 if the file cannot be opened.
 * Does NOT derive cellSize/grid (call orc_params_derive). */
int orc_load_cfg(OrcParams *p, const char *path);
/* main.cpp:932-939 derived cellSize, gridSize 512^2, numCells, worldOrigin (-64,-64).
 * grid_override > 0 / arena_half > 0 apply the generalised-arena extension instead. */
void orc_params_derive(OrcParams *p, uint32_t grid_override, float arena_half);

/* ---- kernel-level restatements: same argument meaning as particlebot.cuh:33-119 ---- */
void orc_integrateSystem(const OrcParams *P, float *pos, float *vel, const float *rad, float dt,
                         uint32_t n);
void orc_calcHash(const OrcParams *P, uint32_t *hash, uint32_t *index, const float *pos, uint32_t n);
void orc_sortParticlebots(uint32_t *hash, uint32_t *index, uint32_t n);
void orc_reorderDataAndFindCellStart(const OrcParams *P, uint32_t *cellStart, uint32_t *cellEnd,
                                     float *sortedPos, float *sortedVel, float *sortedRad,
                                     const uint32_t *hash, const uint32_t *index, const float *oldPos,
                                     const float *oldVel, const float *oldRad, uint32_t n,
                                     uint32_t numCells);
void orc_updateRad_light_wave(const OrcParams *P, const float *absForce_a, const float *absForce_r,
                              float *rad, const float *phase, float time, float dt, const int32_t *dead,
                              uint32_t n);
void orc_updatePhase(const OrcParams *P, const float *pos, float *phase, float spacing, float max_d,
                     float min_d, uint32_t n);
void orc_collide(const OrcParams *P, float *newVel, float *absForce_a, float *absForce_r,
                 const float *sortedPos, const float *sortedVel, const float *sortedRad,
                 const uint32_t *index, const uint32_t *cellStart, const uint32_t *cellEnd, uint32_t n,
                 float dt);
/* pair force, exposed for known-answer tests (particlebot_kernel_impl.cuh:541-594).
 * force[2], forcea, forcer are accumulated into. */
void orc_collideSpheres(const OrcParams *P, const float posA[2], const float posB[2],
                        const float velA[2], const float velB[2], float radA, float radB,
                        float attraction, float force[2], float *forcea, float *forcer);
/* host min/max distance loop (particlebot.cpp:215-228) */
void orc_minmax_light_distance(const OrcParams *P, const float *pos, uint32_t n, float *min_d,
                               float *max_d);

/* ---- phase noise: this build's own counter RNG (NOT cuRAND; parity unpinned) ---- */
/* standard normal for (seed, bot i, draw k) */
float orc_normal(uint32_t seed, uint32_t i, uint32_t k);
void orc_add_normal_noise(uint32_t seed, uint32_t draw, float *val, float std, uint32_t n);

/* ---- whole simulation object: restates class Particlebot (particlebot.cpp) ---- */
typedef struct OrcSim OrcSim;
/* Calls srand(P->seed) exactly as main.cpp:929 does, then allocates (zeroed). */
OrcSim *orc_sim_create(const OrcParams *P);
void orc_sim_destroy(OrcSim *s);
/* particlebot.cpp:485-801; only CONFIG_RANDOM and CONFIG_HEX are restated (config key never
 * takes effect in the reference, main.cpp:794-809).  use_hex!=0 selects initHexGrid. */
void orc_sim_reset(OrcSim *s, int use_hex);
/* particlebot.cpp:170-300.  Returns 1 instead of exit(0) when time > max_time, else 0. */
int orc_sim_update(OrcSim *s, float dt, float sort_interval);
/* particlebot.cpp:303-367; fp may be NULL (then only the centroid is computed when due).
 * Returns 1 if a row was due. */
int orc_sim_dump(OrcSim *s, FILE *fp, float dump_interval, uint32_t testing, int echo);
/* particlebot.cpp:369-411 */
int orc_sim_load_from_file(OrcSim *s, FILE *fp);
/* test helper, NOT reference behaviour: force the re-hash + sort on the next update */
void orc_sim_force_sort_once(OrcSim *s);
float orc_sim_time(const OrcSim *s);
void orc_sim_set_time(OrcSim *s, float t);
uint32_t orc_sim_phase_draws(const OrcSim *s);
/* raw array access (original index order).  which: 0 pos(2n) 1 vel(2n) 2 rad(n) 3 phase(n)
 * 4 absForce_a(n) 5 absForce_r(n) 6 dead(n, int32) 7 hash(n,u32) 8 index(n,u32) */
void *orc_sim_array(OrcSim *s, int which);
/* number of worker threads used by the OpenMP loops (1 when built without OpenMP) */
int orc_num_threads(void);
void orc_set_num_threads(int n);
/* "exact" for the oracle; "fma" / "fma+powf" for the bracket builds (pb_oracle.c header, oracle/Makefile) */
const char *orc_build_variant(void);

/* XORWOW test hooks (the generator-2 section of pb_oracle.c): raw outputs of
 * curand_init(seed, subsequence, 0); normals of bots 0..nbots-1 over `draws` phase updates
 * (out[draw][bot]); the 2^67-step jump matrix in the product's row layout (160 x 5 words). */
void orc_xorwow_outputs(int kind, uint64_t seed, uint32_t subsequence, uint32_t count, uint32_t *out);
void orc_xorwow_normals(int kind, uint32_t seed, uint32_t nbots, uint32_t draws, float *out);
void orc_xorwow_jump_rows(uint32_t *rows);

#ifdef __cplusplus
}
#endif
#endif
