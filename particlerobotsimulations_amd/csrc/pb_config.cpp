// pb_config.cpp -- .cfg loader with the reference's exact matching rules.
#include "pb_config.hpp"

#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>

PbRunConfig::PbRunConfig() {
  // main.cpp:832-911
  memset(&params, 0, sizeof(params));
  params.nobstacles = 0;
  params.n_cir_obstacles = 0;
  x1obs.assign(1, 0.0f);
  x2obs.assign(1, 0.0f);
  y1obs.assign(1, 0.0f);
  y2obs.assign(1, 0.0f);
  x_cir_obs.assign(1, 0.0f);
  y_cir_obs.assign(1, 0.0f);
  r_cir_obs.assign(1, 0.0f);
  params.min_radius = 0.0775;
  params.max_radius = 0.1175;
  params.centroid_int = 10;
  params.centroid_radius = 0.05f;
  params.centroid_steps = 24000;
  sort_interval = 180.0f;
  dump_interval = 60.0f;
  params.testing = 0;
  params.friction = 0.4;
  params.spring = 1000.0f;
  params.damping = 10.0f;
  params.shear = 40.0f;
  params.constraint = 0.5f;
  params.constrained_contraction = 0;
  params.constraint_contraction = 10.0f;
  params.attraction = 3.0f * 0.000015884f;
  params.boundaryDamping = -1.0f;
  params.gravity = 9.81 * 0.566f;
  camera_y = 10;
  camera_x = 0;
  light_radius = 0.25f;
  timestep = 0.01f;
  params.nCells = 501;
  params.nDead = -1;
  params.radFactor = 2.0;
  params.massFactor = 1.0;
  params.frictionFactor = 1.0;
  params.attractionFactor = 0.0f;
  params.time_to_dead = 0;
  params.max_time = 6400.0;
  params.seed = (unsigned)time(NULL);
  params.light_x = -5.0;
  params.light_y = 0;
  params.light_shadow = 0;
  params.rise_period = 2;
  params.phase_std = 0.3f * params.rise_period;
  params.config = CONFIG_RANDOM;
  params.display_shadow = 0;
  params.phase_update_interval = 12;
  params.control = LIGHT_WAVE;
  params.Nx = 5;
  params.freq = 0.5f / 25;
  display_interval = 100;
  video_interval = 100;
  csv_filename = "particle_bot_output_data.csv";
  video_filename = "particle_bot_output_video.avi";
  grid_size = 0;
  arena_half = 0.0f;
  hex_spacing = 0.0f;
  square_lattice = false;
  fast_blob = false;
  rng_kind = 0;
  force_variant = -1;
  repoint();
}

void PbRunConfig::repoint() {
  params.x1obs = x1obs.data();
  params.x2obs = x2obs.data();
  params.y1obs = y1obs.data();
  params.y2obs = y2obs.data();
  params.x_cir_obs = x_cir_obs.data();
  params.y_cir_obs = y_cir_obs.data();
  params.r_cir_obs = r_cir_obs.data();
}

namespace {

// whitespace-separated floats; how many is fixed by the count key seen earlier (main.cpp:612-676)
void readList(const std::string &value, std::vector<float> &dst, int count) {
  const char *s = value.c_str();
  for (int i = 0; i < count && i < (int)dst.size(); i++) {
    char *end = nullptr;
    const float v = strtof(s, &end);
    if (end == s) return;  // std::stof would throw in the reference
    dst[i] = v;
    s = end;
  }
}

}  // namespace

void PbRunConfig::setParam(const std::string &name, const std::string &value) {
  const char *p = name.c_str();
  const char *v = value.c_str();
  // strncmp(param, key, n) in the reference's order; the first match wins
  auto is = [&](const char *key, size_t n) { return strncmp(p, key, n) == 0; };
  auto f = [&]() { return strtof(v, NULL); };
  auto l = [&]() { return strtol(v, NULL, 10); };

  if (is("camera_y", 8)) camera_y = f();
  else if (is("camera_x", 8)) camera_x = f();
  else if (is("nobstacles", 11)) {
    params.nobstacles = l();
    const size_t m = params.nobstacles > 0 ? params.nobstacles : 1;
    x1obs.assign(m, 0.0f);
    x2obs.assign(m, 0.0f);
    y1obs.assign(m, 0.0f);
    y2obs.assign(m, 0.0f);
    repoint();
  } else if (is("x1obs", 5)) readList(value, x1obs, params.nobstacles);
  else if (is("x2obs", 5)) readList(value, x2obs, params.nobstacles);
  else if (is("y1obs", 5)) readList(value, y1obs, params.nobstacles);
  else if (is("y2obs", 5)) readList(value, y2obs, params.nobstacles);
  else if (is("n_cir_obstacles", 15)) {
    params.n_cir_obstacles = l();
    const size_t m = params.n_cir_obstacles > 0 ? params.n_cir_obstacles : 1;
    x_cir_obs.assign(m, 0.0f);
    y_cir_obs.assign(m, 0.0f);
    r_cir_obs.assign(m, 0.0f);
    repoint();
  } else if (is("x_cir_obs", 5)) readList(value, x_cir_obs, params.n_cir_obstacles);  // 5 chars compared
  else if (is("y_cir_obs", 5)) readList(value, y_cir_obs, params.n_cir_obstacles);
  else if (is("r_cir_obs", 5)) readList(value, r_cir_obs, params.n_cir_obstacles);
  else if (is("min_radius", 10)) params.min_radius = f();
  else if (is("max_radius", 10)) params.max_radius = f();
  else if (is("centroid_int", 12)) params.centroid_int = l();  // integer parse into a float field
  else if (is("centroid_radius", 15)) params.centroid_radius = f();
  else if (is("centroid_steps", 14)) params.centroid_steps = l();
  else if (is("radFactor", 9)) params.radFactor = f();
  else if (is("massFactor", 10)) params.massFactor = f();
  else if (is("frictionFactor", 14)) params.frictionFactor = f();
  else if (is("attractionFactor", 16)) params.attractionFactor = f();
  else if (is("dump_interval", 13)) dump_interval = f();
  else if (is("sort_interval", 13)) sort_interval = f();
  else if (is("testing", 7)) params.testing = l();
  else if (is("friction", 8)) params.friction = f();
  else if (is("spring", 6)) params.spring = f();
  else if (is("damping", 7)) params.damping = f();
  else if (is("shear", 5)) params.shear = f();
  else if (is("constraint", 10)) params.constraint = f();  // also swallows constraint_contraction
  else if (is("constrained_contraction", 23)) params.constrained_contraction = l();
  else if (is("constraint_contraction", 22)) params.constraint_contraction = f();  // unreachable
  else if (is("attraction", 10)) params.attraction = f();
  else if (is("boundaryDamping", 15)) params.boundaryDamping = f();
  else if (is("gravity", 7)) params.gravity = f();
  else if (is("nCells", 6)) params.nCells = l();
  else if (is("nDead", 5)) params.nDead = l();
  else if (is("time_to_dead", 14)) params.time_to_dead = f();  // n > strlen: exact match only
  else if (is("max_time", 8)) params.max_time = f();
  else if (is("seed", 4)) params.seed = l();
  else if (is("light_radius", 12)) light_radius = f();
  else if (is("light_x", 7)) params.light_x = f();
  else if (is("light_y", 7)) params.light_y = f();
  else if (is("timestep", 8)) timestep = f();
  else if (is("light_shadow", 12)) params.light_shadow = l();
  else if (is("csv_filename", 12)) csv_filename = value;
  else if (is("video_filename", 14)) video_filename = value;
  else if (is("rise_period", 11)) params.rise_period = f();
  else if (is("phase_std", 9)) params.phase_std = f();
  else if (is("display_shadow", 14)) params.display_shadow = l();
  else if (is("phase_update_interval", 21)) params.phase_update_interval = l();
  else if (is("Nx", 2)) params.Nx = l();  // unreachable from a file: names shorter than 4 are skipped
  else if (is("config", 6)) {
    // The reference compares the NAME ("config") with "CONFIG_*" (main.cpp:794-809): never true,
    // placement stays CONFIG_RANDOM.  Kept.
  } else if (is("DISPLAY_INTERVAL", 16)) display_interval = l();
  else if (is("VIDEO_INTERVAL", 14)) video_interval = l();
  // ---- extensions, tried only after every reference key ----
  else if (is("pb_grid_size", 12)) grid_size = (unsigned)l();
  else if (is("pb_arena_half", 13)) arena_half = f();
  else if (is("pb_hex_spacing", 14)) hex_spacing = f();
  else if (is("pb_placement", 12)) {
    square_lattice = value.rfind("square", 0) == 0;
    fast_blob = value.rfind("fastblob", 0) == 0;
    if (value.rfind("hex", 0) == 0) params.config = CONFIG_HEX;
    else if (value.rfind("grid", 0) == 0) params.config = CONFIG_GRID;
    else if (value.rfind("line", 0) == 0) params.config = CONFIG_LINE;
    else if (value.rfind("blob_upleft", 0) == 0) params.config = CONFIG_BLOB_UPLEFT;
    else if (value.rfind("blob", 0) == 0) params.config = CONFIG_BLOB;
    else if (value.rfind("lighttest7", 0) == 0) params.config = CONFIG_LIGHTTEST_7;
    else params.config = CONFIG_RANDOM;
  }
  else if (is("pb_force_variant", 16)) {
    // which force kernel the fused engine runs (include/particlebot_hip.h pbSimSetForceVariant): 0-2 exact, 3 the
    // opt-in tolerance kernel; anything else keeps the default
    const long variant = l();
    force_variant = (variant >= 0 && variant <= 3) ? (int)variant : -1;
  }
  else if (is("pb_rng", 6)) {
    // phase-noise generator (include/particlebot_hip.h PB_RNG_*): "curand" = cuRAND-compatible XORWOW
    // (what the reference's curand_init/curand_normal draw), "rocrand" = the same generator with
    // rocRAND's seeding constants, anything else = this project's counter generator
    rng_kind = value.rfind("curand", 0) == 0 ? 1 : value.rfind("rocrand", 0) == 0 ? 2 : 0;
  }
  // anything else: the value line is consumed and ignored, as in the reference
}

bool PbRunConfig::loadFile(const std::string &path) {
  std::ifstream in(path);
  if (!in) return false;
  std::string name, value;
  while (std::getline(in, name)) {
    if (name.length() < 4 || name[0] == '#') continue;  // main.cpp:924
    if (std::getline(in, value)) setParam(name, value);
  }
  return true;
}

void PbRunConfig::derive() {
  // main.cpp:932-939
  if (params.nDead == -1 && params.max_radius * 0.5 * params.radFactor > 2 * params.max_radius)
    params.cellSize.x = params.cellSize.y = params.max_radius * 0.5 * params.radFactor + 4 * params.max_radius;
  else
    params.cellSize.x = params.cellSize.y = params.max_radius * 2;
  params.gridSize.x = params.gridSize.y = grid_size ? grid_size : 512;
  params.numCells = params.gridSize.x * params.gridSize.y;
  const float half = wallHalf();
  params.worldOrigin = make_float2(-half, -half);
  repoint();
}
