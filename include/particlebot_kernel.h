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

/* particlebot_kernel.cuh:25-35 */
enum ParticlebotConfig {
  CONFIG_RANDOM,
  CONFIG_GRID,
  CONFIG_BLOB,
  CONFIG_BLOB_UPLEFT,
  CONFIG_HEX,
  CONFIG_LINE,
  CONFIG_LIGHTTEST_7,
  _NUM_CONFIGS
};

/* particlebot_kernel.cuh:37-45 */
enum ParticlebotArray { POSITION, VELOCITY, RADII, PHASE, FREQUENCY, DEAD };

/* particlebot_kernel.cuh:47-50 */
enum ParticlebotControl { LIGHT_WAVE };

/* Obstacle lists hold at most this many entries on the device (the reference keeps
 * `__constant__ float x1obs[10]` etc., particlebot_kernel_impl.cuh:28-34). */
#define PB_MAX_OBSTACLES 10

/* particlebot_kernel.cuh:58-120 -- same fields, same order */
struct SimParams {
  uint2 gridSize;
  uint numCells;

  float2 worldOrigin;
  float2 cellSize;

  uint nCells;
  int nDead;
  uint maxParticlebotsPerCell;

  float gravity;
  float spring;
  float damping;
  float shear;
  float attraction;
  float boundaryDamping;
  float friction;

  float massFactor;
  float frictionFactor;
  float radFactor;
  float attractionFactor;

  float constraint;
  float constraint_contraction;
  int centroid_steps;
  float centroid_int;
  float centroid_radius;
  float light_x;
  float light_y;
  float phase_update_interval;
  ParticlebotControl control;
  ParticlebotConfig config;
  float min_radius;
  float max_radius;
  float rise_period;
  float freq;

  int nobstacles;
  float *x1obs;
  float *x2obs;
  float *y1obs;
  float *y2obs;

  int n_cir_obstacles;
  float *x_cir_obs;
  float *y_cir_obs;
  float *r_cir_obs;

  int Nx;

  float phase_std;
  unsigned seed;

  uint light_shadow;
  uint testing;
  uint constrained_contraction;
  uint display_shadow;
  float time_to_dead;
  float max_time;
};

#endif /* PARTICLEBOT_KERNEL_H */
