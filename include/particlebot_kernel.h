/*
 * particlebot_kernel.h -- parameter block and enums of the particle-robot update loop.
 *
 * Drop-in for the reference's particlebot_kernel.cuh: `SimParams` keeps the field order, types and
 * therefore the memory layout of particlebot_kernel.cuh:58-120, so a caller built against the
 * reference header can hand the same struct to setParameters()/Particlebot(SimParams).
 * The enums follow particlebot_kernel.cuh:25-55.
 *
 * float2/uint2 are the HIP vector types (same layout as CUDA's vector_types.h).
 */
#ifndef PARTICLEBOT_KERNEL_H
#define PARTICLEBOT_KERNEL_H

#include <hip/hip_vector_types.h>

typedef unsigned int uint;

/* particlebot_kernel.cuh:25-35: initial-placement presets (values 0..6, then the count) */
enum ParticlebotConfig {
  CONFIG_RANDOM, CONFIG_GRID, CONFIG_BLOB, CONFIG_BLOB_UPLEFT, CONFIG_HEX, CONFIG_LINE, CONFIG_LIGHTTEST_7,
  _NUM_CONFIGS
};

/* particlebot_kernel.cuh:37-45: selector of getArray / setArray */
enum ParticlebotArray { POSITION, VELOCITY, RADII, PHASE, FREQUENCY, DEAD };

/* particlebot_kernel.cuh:47-50: the only control law */
enum ParticlebotControl { LIGHT_WAVE };

/* Obstacle lists hold at most this many entries on the device (the reference keeps
 * `__constant__ float x1obs[10]` etc., particlebot_kernel_impl.cuh:28-34). */
#define PB_MAX_OBSTACLES 10

/* particlebot_kernel.cuh:58-120.  Field ORDER and TYPES are the reference's (that is the ABI: 256
 * bytes, float2/uint2 8-byte aligned as in CUDA; static_assert'ed in csrc/pb_device.hpp); fields
 * of one type that are adjacent there are declared together here. */
struct SimParams {
  /* uniform grid */
  uint2 gridSize;
  uint numCells;
  float2 worldOrigin, cellSize;
  /* population: nDead == -1 selects object-transport mode (the last bot is a passive payload) */
  uint nCells;
  int nDead;
  uint maxParticlebotsPerCell;
  /* forces */
  float gravity, spring, damping, shear, attraction, boundaryDamping, friction;
  /* payload multipliers (object transport) */
  float massFactor, frictionFactor, radFactor, attractionFactor;
  /* actuation limits */
  float constraint, constraint_contraction;
  /* centroid trail (display) */
  int centroid_steps;
  float centroid_int, centroid_radius;
  /* light and control */
  float light_x, light_y, phase_update_interval;
  ParticlebotControl control;
  ParticlebotConfig config;
  float min_radius, max_radius, rise_period, freq;
  /* rectangular obstacles: nobstacles entries in each host array */
  int nobstacles;
  float *x1obs, *x2obs, *y1obs, *y2obs;
  /* circular obstacles: n_cir_obstacles entries in each host array */
  int n_cir_obstacles;
  float *x_cir_obs, *y_cir_obs, *r_cir_obs;
  int Nx;
  float phase_std;
  unsigned seed;
  uint light_shadow, testing, constrained_contraction, display_shadow;
  float time_to_dead, max_time;
};

#endif /* PARTICLEBOT_KERNEL_H */
