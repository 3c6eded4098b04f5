// pb_config.hpp -- the reference's .cfg loader (main.cpp:594-816, 832-939) as a reusable object.
#pragma once

#include <string>
#include <vector>

#include "particlebot_kernel.h"

// Everything main.cpp keeps in globals next to `SimParams params` (main.cpp:57-87).
struct PbRunConfig {
  SimParams params;
  float timestep, sort_interval, dump_interval;
  float camera_x, camera_y, light_radius;
  int display_interval, video_interval;
  std::string csv_filename, video_filename;
  // obstacle storage the SimParams pointers refer to (the reference mallocs these, :607-610)
  std::vector<float> x1obs, x2obs, y1obs, y2obs, x_cir_obs, y_cir_obs, r_cir_obs;
  // extensions (not in the reference): generalised arena
  unsigned grid_size;  // 0 -> 512 (main.cpp:937)
  float arena_half;    // 0 -> walls and world origin at +-64 (main.cpp:939, impl.cuh:75-97)
  float hex_spacing;   // 0 -> 2*min_radius (particlebot.cpp:760); lattice pitch of hex / square placement
  bool square_lattice; // pb_placement square
  bool fast_blob;      // pb_placement fastblob: O(N) random blob (Particlebot::placeFastBlob)
  int force_variant;   // pb_force_variant: -1 (default) the engine's choice = 2, the exact kernels; 0/1/2 exact forms; 3 the opt-in
                       // tolerance kernel (pbSimSetForceVariant; held to the FMA bracket of the reference's arithmetic)
  int rng_kind;        // pb_rng: PB_RNG_COUNTER ("pbrng", default), PB_RNG_XORWOW_CURAND ("curand"), PB_RNG_XORWOW_ROCRAND ("rocrand")

  PbRunConfig();
  // main.cpp:594-816: one name/value pair, prefix matching in source order, quirks included
  void setParam(const std::string &name, const std::string &value);
  // main.cpp:918-928; false if the file cannot be opened
  bool loadFile(const std::string &path);
  // main.cpp:932-939 (+ the extensions); call after all setParam/loadFile
  void derive();
  float wallHalf() const { return arena_half > 0.0f ? arena_half : 64.0f; }

 private:
  void repoint();
};
