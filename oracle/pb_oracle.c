/*
 * pb_oracle.c -- CPU ORACLE (test infrastructure only; see pb_oracle.h for scope and pinning).
 *
 * Plain-C restatement of the reference's per-timestep update loop.  Every function cites the
 * reference file:line it follows.  Build with -O2 -ffp-contract=off (no FMA contraction, no
 * -ffast-math, no -march=native): the placement accept/reject loop and the dynamics are
 * bit-fragile.
 */
#define _GNU_SOURCE
#include "pb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/*
 * BRACKET BUILDS (oracle/Makefile: libpb_oracle_fma.so, libpb_oracle_fma_powf.so).  The reference is
 * compiled by nvcc -O3 with its default -fmad=true (/root/reference/Makefile:79-89 sets no
 * -fmad=false) and calls the fast intrinsic __powf at particlebot_kernel_impl.cuh:586,589.  Neither
 * nvcc's contraction choices nor __powf's bits can be pinned in this image, so two more builds of
 * THIS file bracket them:
 *   -DORC_BRACKET_FMA   the kernel functions (everything between the two "device code" markers
 *                       below) are compiled with fp-contract=fast + FMA instructions; config,
 *                       placement, the host loop and the dump stay contract-off like the exact build
 *                       (nvcc hands host code to g++ unchanged)
 *   -DORC_BRACKET_POWF  additionally x*x -> exp2f(2*log2f(x)) at the two __powf sites (the published
 *                       definition of __powf(x,y) = exp2f(y * __log2f(x)); glibc's exp2f/log2f stand
 *                       in for the hardware approximations, so this is a LOWER bound on its error)
 *   -DORC_BRACKET_ORDER the oracle's own terms, bit for bit, added in the order of the product's two-pass tolerance
 *                       kernel (a bot's contact terms after its last candidate): isolates what the ORDER of fp32
 *                       additions alone does (tests/test_fma_bracket.py)
 * The exact build (no macro) is the oracle; the bracket builds only measure how far a
 * legitimately different build of the same source drifts from it (tests/test_fma_bracket.py).
 */
#if defined(ORC_BRACKET_POWF)
#define ORC_POW2(x) exp2f(2.0f * log2f(x))
#else
#define ORC_POW2(x) ((x) * (x))
#endif

/*
 * The DEVICE powf sites (round 5).  Besides the two __powf calls the reference's kernels call the device
 * library's powf(x, 2) and powf(x, 0.5f) in every obstacle and shadow test
 * (particlebot_kernel_impl.cuh:214-216, 224, 226, 228-229 checkIntersectionCircle; :704-705, :719 circular
 * obstacles; :757-779 rectangle corners).  CUDA's device powf is NOT correctly rounded (the CUDA C Programming
 * Guide's accuracy table allows it several ulp over the full range), so the oracle's x*x / sqrtf at these sites
 * is one member of a family, like its x*x at the __powf sites.  Two more bracket members:
 *   -DORC_BRACKET_DEVPOWF=1  powf(x,2) -> exp2f(2*log2f(|x|)), powf(x,0.5f) -> exp2f(0.5f*log2f(x)): the textbook
 *                            evaluation of a power through base-2 logarithms in fp32 (errors of a few ulp)
 *   -DORC_BRACKET_DEVPOWF=2  the correctly rounded result moved by -1 / 0 / +1 ulp, chosen by a hash of the
 *                            argument's bits (a quarter of the calls each way): what a faithfully-but-not-correctly
 *                            rounded powf does, without committing to one implementation
 * Only obstacle configurations (example_obstacle.cfg: circles; example_gap.cfg: rectangles; light_shadow) reach
 * these sites.
 */
#if !defined(ORC_BRACKET_DEVPOWF)
#define ORC_DPOW2(x) ((x) * (x))
#define ORC_DSQRT(x) sqrtf(x)
#elif ORC_BRACKET_DEVPOWF == 1
#define ORC_DPOW2(x) exp2f(2.0f * log2f(fabsf(x)))
#define ORC_DSQRT(x) exp2f(0.5f * log2f(x))
#else
static inline float orc_ulp_nudge(float exact, float arg) {
  uint32_t b, k;
  memcpy(&k, &arg, 4);
  k = (k * 2654435761u) >> 30; /* 0: one ulp down, 1: one ulp up, 2, 3: as it is */
  memcpy(&b, &exact, 4);
  if (k > 1u || exact == 0.0f || (b & 0x7f800000u) == 0x7f800000u) return exact;
  b = (k == 0u) ? b - 1u : b + 1u; /* (exact > 0 at every site) */
  memcpy(&exact, &b, 4);
  return exact;
}
#define ORC_DPOW2(x) orc_ulp_nudge((x) * (x), (x))
#define ORC_DSQRT(x) orc_ulp_nudge(sqrtf(x), (x))
#endif

const char *orc_build_variant(void) {
#if defined(ORC_BRACKET_ORDER)
  return "order";
#elif defined(ORC_BRACKET_FMA) && defined(ORC_BRACKET_POWF) && defined(ORC_BRACKET_DEVPOWF)
  return "fma+powf+devpowf";
#elif defined(ORC_BRACKET_DEVPOWF)
  return ORC_BRACKET_DEVPOWF == 1 ? "devpowf" : "devpowf_ulp";
#elif defined(ORC_BRACKET_FMA) && defined(ORC_BRACKET_POWF)
  return "fma+powf";
#elif defined(ORC_BRACKET_FMA)
  return "fma";
#elif defined(ORC_BRACKET_POWF)
  return "powf";
#else
  return "exact";
#endif
}

/* below this many bots every loop runs on the calling thread (fork/join costs more than the work) */
#define ORC_OMP_MIN_N 20000u

#define ORC_PI_F 3.141592654f /* particlebot.cpp:21-23 CUDART_PI_F fallback */

/* ------------------------------------------------------------------------------------------ */
/* float2 helpers with the exact operation order of include/helper_math.h                       */
/* ------------------------------------------------------------------------------------------ */


/* particlebot.cpp:32-34 host-side length(): glibc powf on purpose */
static inline float host_length(float x, float y) { return powf(powf(x, 2.0f) + powf(y, 2.0f), 0.5f); }

/* ------------------------------------------------------------------------------------------ */
/* config: defaults, setParam, loader, derived parameters                                       */
/* ------------------------------------------------------------------------------------------ */

void orc_params_defaults(OrcParams *p) {
  /* main.cpp:832-911 */
  memset(p, 0, sizeof(*p));
  p->nobstacles = 0;
  p->n_cir_obstacles = 0;
  p->min_radius = 0.0775;
  p->max_radius = 0.1175;
  p->centroid_int = 10;
  p->centroid_radius = 0.05f;
  p->centroid_steps = 24000;
  p->sort_interval = 180.0f;
  p->dump_interval = 60.0f;
  p->testing = 0;
  p->friction = 0.4;
  p->spring = 1000.0f;
  p->damping = 10.0f;
  p->shear = 40.0f;
  p->constraint = 0.5f;
  p->constrained_contraction = 0;
  p->constraint_contraction = 10.0f;
  p->attraction = 3.0f * 0.000015884f;
  p->boundaryDamping = -1.0f;
  p->gravity = 9.81 * 0.566f;
  p->camera_y = 10;
  p->camera_x = 0;
  p->light_radius = 0.25f;
  p->timestep = 0.01f;
  p->nCells = 501;
  p->nDead = -1;
  p->radFactor = 2.0;
  p->massFactor = 1.0;
  p->frictionFactor = 1.0;
  p->attractionFactor = 0.0f;
  p->time_to_dead = 0;
  p->max_time = 6400.0;
  p->seed = 0; /* reference: time(NULL); callers must set it */
  p->light_x = -5.0;
  p->light_y = 0;
  p->light_shadow = 0;
  p->rise_period = 2;
  p->phase_std = 0.3f * p->rise_period;
  p->config = 0; /* CONFIG_RANDOM */
  p->display_shadow = 0;
  p->phase_update_interval = 12;
  p->control = 0; /* LIGHT_WAVE */
  p->Nx = 5;
  p->freq = 0.5f / 25;
  p->display_interval = 100;
  p->video_interval = 100;
  snprintf(p->csv_filename, sizeof(p->csv_filename), "particle_bot_output_data.csv");
  snprintf(p->video_filename, sizeof(p->video_filename), "particle_bot_output_video.avi");
  p->wallHalf = 64.0f;
}

/* list value: whitespace-separated floats, count taken from the earlier count key
 * (main.cpp:612-676, std::stof + substr) */
static void parse_float_list(const char *value, float *dst, int count) {
  const char *s = value;
  if (count > ORC_MAX_OBS) count = ORC_MAX_OBS; /* device arrays are float[10], impl.cuh:28-34 */
  for (int i = 0; i < count; i++) {
    char *end = NULL;
    float v = strtof(s, &end);
    if (end == s) return; /* std::stof would throw here */
    dst[i] = v;
    s = end;
  }
}

enum { K_F, K_FL /* float field parsed with strtol */, K_I, K_U, K_STR, K_LIST, K_NOBS, K_NCIR, K_CONFIG };

void orc_set_param(OrcParams *p, const char *name, const char *value) {
  /* main.cpp:594-816: strncmp(param, key, n) down an if/else chain: prefix match in SOURCE ORDER.
   * The table below keeps that order and those n values; first hit wins. */
  struct Row {
    const char *key;
    size_t n;
    int kind;
    void *dst;
  } rows[] = {
      {"camera_y", 8, K_F, &p->camera_y},
      {"camera_x", 8, K_F, &p->camera_x},
      {"nobstacles", 11, K_NOBS, NULL},
      {"x1obs", 5, K_LIST, p->x1obs},
      {"x2obs", 5, K_LIST, p->x2obs},
      {"y1obs", 5, K_LIST, p->y1obs},
      {"y2obs", 5, K_LIST, p->y2obs},
      {"n_cir_obstacles", 15, K_NCIR, NULL},
      {"x_cir", 5, K_LIST, p->x_cir_obs}, /* only 5 chars are compared, main.cpp:653 */
      {"y_cir", 5, K_LIST, p->y_cir_obs},
      {"r_cir", 5, K_LIST, p->r_cir_obs},
      {"min_radius", 10, K_F, &p->min_radius},
      {"max_radius", 10, K_F, &p->max_radius},
      {"centroid_int", 12, K_FL, &p->centroid_int},
      {"centroid_radius", 15, K_F, &p->centroid_radius},
      {"centroid_steps", 14, K_I, &p->centroid_steps},
      {"radFactor", 9, K_F, &p->radFactor},
      {"massFactor", 10, K_F, &p->massFactor},
      {"frictionFactor", 14, K_F, &p->frictionFactor},
      {"attractionFactor", 16, K_F, &p->attractionFactor},
      {"dump_interval", 13, K_F, &p->dump_interval},
      {"sort_interval", 13, K_F, &p->sort_interval},
      {"testing", 7, K_U, &p->testing},
      {"friction", 8, K_F, &p->friction},
      {"spring", 6, K_F, &p->spring},
      {"damping", 7, K_F, &p->damping},
      {"shear", 5, K_F, &p->shear},
      {"constraint", 10, K_F, &p->constraint}, /* shadows constraint_contraction, main.cpp:725 */
      {"constrained_contraction", 23, K_U, &p->constrained_contraction},
      {"constraint_contraction", 22, K_F, &p->constraint_contraction}, /* unreachable */
      {"attraction", 10, K_F, &p->attraction},
      {"boundaryDamping", 15, K_F, &p->boundaryDamping},
      {"gravity", 7, K_F, &p->gravity},
      {"nCells", 6, K_U, &p->nCells},
      {"nDead", 5, K_I, &p->nDead},
      {"time_to_dead", 14, K_F, &p->time_to_dead}, /* n=14 > strlen: exact match */
      {"max_time", 8, K_F, &p->max_time},
      {"seed", 4, K_U, &p->seed},
      {"light_radius", 12, K_F, &p->light_radius},
      {"light_x", 7, K_F, &p->light_x},
      {"light_y", 7, K_F, &p->light_y},
      {"timestep", 8, K_F, &p->timestep},
      {"light_shadow", 12, K_U, &p->light_shadow},
      {"csv_filename", 12, K_STR, p->csv_filename},
      {"video_filename", 14, K_STR, p->video_filename},
      {"rise_period", 11, K_F, &p->rise_period},
      {"phase_std", 9, K_F, &p->phase_std},
      {"display_shadow", 14, K_U, &p->display_shadow},
      {"phase_update_interval", 21, K_FL, &p->phase_update_interval},
      {"Nx", 2, K_I, &p->Nx}, /* unreachable from a file: key length 2 < 4 */
      {"config", 6, K_CONFIG, NULL},
      {"DISPLAY_INTERVAL", 16, K_I, &p->display_interval},
      {"VIDEO_INTERVAL", 14, K_I, &p->video_interval},
  };
  const size_t nrows = sizeof(rows) / sizeof(rows[0]);
  for (size_t r = 0; r < nrows; r++) {
    if (strncmp(name, rows[r].key, rows[r].n) != 0) continue;
    switch (rows[r].kind) {
      case K_F: *(float *)rows[r].dst = strtof(value, NULL); break;
      case K_FL: *(float *)rows[r].dst = (float)strtol(value, NULL, 10); break;
      case K_I: *(int32_t *)rows[r].dst = (int32_t)strtol(value, NULL, 10); break;
      case K_U: *(uint32_t *)rows[r].dst = (uint32_t)strtol(value, NULL, 10); break;
      case K_STR: snprintf((char *)rows[r].dst, 300, "%s", value); break;
      case K_NOBS: p->nobstacles = (int32_t)strtol(value, NULL, 10); break;
      case K_NCIR: p->n_cir_obstacles = (int32_t)strtol(value, NULL, 10); break;
      case K_LIST: {
        int cnt = (rows[r].dst == p->x_cir_obs || rows[r].dst == p->y_cir_obs ||
                   rows[r].dst == p->r_cir_obs)
                      ? p->n_cir_obstacles
                      : p->nobstacles;
        parse_float_list(value, (float *)rows[r].dst, cnt);
      } break;
      case K_CONFIG: /* main.cpp:794-809 compares PARAM (== "config") with "CONFIG_*": never true */
        break;
    }
    return;
  }
  /* extension key of the product's loader (csrc/pb_config.cpp), tried after every reference key */
  if (strncmp(name, "pb_rng", 6) == 0) {
    p->rngKind = strncmp(value, "curand", 6) == 0 ? 1 : strncmp(value, "rocrand", 7) == 0 ? 2 : 0;
    return;
  }
  /* unknown keys silently consume their value line */
}

int orc_load_cfg(OrcParams *p, const char *path) {
  /* main.cpp:918-928 */
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  char *line = NULL, *val = NULL;
  size_t cap = 0, vcap = 0;
  ssize_t len;
  while ((len = getline(&line, &cap, f)) >= 0) {
    if (len > 0 && line[len - 1] == '\n') line[--len] = 0;
    if (len < 4 || line[0] == '#') continue;
    ssize_t vlen = getline(&val, &vcap, f);
    if (vlen < 0) break;
    if (vlen > 0 && val[vlen - 1] == '\n') val[--vlen] = 0;
    orc_set_param(p, line, val);
  }
  free(line);
  free(val);
  fclose(f);
  return 0;
}

void orc_params_derive(OrcParams *p, uint32_t grid_override, float arena_half) {
  /* main.cpp:932-939; products are evaluated in double exactly as written there */
  if (p->nDead == -1 && p->max_radius * 0.5 * p->radFactor > 2 * p->max_radius)
    p->cellSizeX = p->cellSizeY = p->max_radius * 0.5 * p->radFactor + 4 * p->max_radius;
  else
    p->cellSizeX = p->cellSizeY = p->max_radius * 2;
  p->gridSizeX = p->gridSizeY = grid_override ? grid_override : 512;
  p->numCells = p->gridSizeX * p->gridSizeY;
  if (arena_half > 0.0f) {
    p->worldOriginX = p->worldOriginY = -arena_half;
    p->wallHalf = arena_half;
  } else {
    p->worldOriginX = p->worldOriginY = -64.0f;
    p->wallHalf = 64.0f;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* kernels                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* >>> device code: what nvcc compiles for the GPU (contracted in the ORC_BRACKET_FMA build) >>> */
#if defined(ORC_BRACKET_FMA)
#pragma GCC push_options
#pragma GCC optimize("fp-contract=fast")
#pragma GCC target("fma")
#endif

/* helper_math.h:1244 dot = a.x*b.x + a.y*b.y ; :1287 length = sqrtf(dot(v,v)) -- device-side
 * helpers (the placement uses host_length above) */
static inline float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }
static inline float len2(float x, float y) { return sqrtf(dot2(x, y, x, y)); }

/* particlebot_kernel_impl.cuh:53-103 integrate_functor, launched by particlebot_cuda.cu:145-160 */
void orc_integrateSystem(const OrcParams *P, float *pos, float *vel, const float *rad, float dt,
                         uint32_t n) {
  const float W = P->wallHalf;
#pragma omp parallel for schedule(static) if (n >= ORC_OMP_MIN_N)
  for (int64_t i = 0; i < (int64_t)n; i++) {
    float px = pos[2 * i], py = pos[2 * i + 1];
    float vx = vel[2 * i], vy = vel[2 * i + 1];
    float r = rad[i];
    px = px + vx * dt;
    py = py + vy * dt;
    if (px > W - r) {
      px = W - r;
      vx *= P->boundaryDamping;
    }
    if (px < -W + r) {
      px = -W + r;
      vx *= P->boundaryDamping;
    }
    if (py > W - r) {
      py = W - r;
      vy *= P->boundaryDamping;
    }
    if (py < -W + r) {
      py = -W + r;
      vy *= P->boundaryDamping;
    }
    pos[2 * i] = px;
    pos[2 * i + 1] = py;
    vel[2 * i] = vx;
    vel[2 * i + 1] = vy;
  }
}

/* impl.cuh:106-112 calcGridPos */
static inline void grid_pos(const OrcParams *P, float x, float y, int *gx, int *gy) {
  *gx = (int)floorf((x - P->worldOriginX) / P->cellSizeX);
  *gy = (int)floorf((y - P->worldOriginY) / P->cellSizeY);
}
/* impl.cuh:115-120 calcGridHash (power-of-two wrap) */
static inline uint32_t grid_hash(const OrcParams *P, int gx, int gy) {
  uint32_t mx = (uint32_t)gx & (P->gridSizeX - 1);
  uint32_t my = (uint32_t)gy & (P->gridSizeY - 1);
  return my * P->gridSizeX + mx;
}

/* impl.cuh:446-465 calcHashD */
void orc_calcHash(const OrcParams *P, uint32_t *hash, uint32_t *index, const float *pos, uint32_t n) {
#pragma omp parallel for schedule(static) if (n >= ORC_OMP_MIN_N)
  for (int64_t i = 0; i < (int64_t)n; i++) {
    int gx, gy;
    grid_pos(P, pos[2 * i], pos[2 * i + 1], &gx, &gy);
    hash[i] = grid_hash(P, gx, gy);
    index[i] = (uint32_t)i;
  }
}

/* particlebot_cuda.cu:377-382 thrust::sort_by_key: a STABLE sort of (hash,index) by hash.
 * Here: LSD radix sort, 4 passes of 8 bits. */
void orc_sortParticlebots(uint32_t *hash, uint32_t *index, uint32_t n) {
  uint32_t *h2 = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
  uint32_t *i2 = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
  uint32_t *hs = hash, *is = index, *hd = h2, *id = i2;
  for (int pass = 0; pass < 4; pass++) {
    uint32_t cnt[257];
    memset(cnt, 0, sizeof(cnt));
    const int sh = pass * 8;
    for (uint32_t k = 0; k < n; k++) cnt[((hs[k] >> sh) & 255u) + 1]++;
    for (int b = 0; b < 256; b++) cnt[b + 1] += cnt[b];
    for (uint32_t k = 0; k < n; k++) {
      uint32_t d = cnt[(hs[k] >> sh) & 255u]++;
      hd[d] = hs[k];
      id[d] = is[k];
    }
    uint32_t *t = hs;
    hs = hd;
    hd = t;
    t = is;
    is = id;
    id = t;
  }
  /* 4 passes: result is back in the caller's arrays */
  free(h2);
  free(i2);
}

/* impl.cuh:469-538 + the cudaMemset(cellStart,0xff) of particlebot_cuda.cu:301.
 * cellEnd is NOT cleared (reference behaviour). */
void orc_reorderDataAndFindCellStart(const OrcParams *P, uint32_t *cellStart, uint32_t *cellEnd,
                                     float *sortedPos, float *sortedVel, float *sortedRad,
                                     const uint32_t *hash, const uint32_t *index, const float *oldPos,
                                     const float *oldVel, const float *oldRad, uint32_t n,
                                     uint32_t numCells) {
  (void)P;
  memset(cellStart, 0xff, (size_t)numCells * sizeof(uint32_t));
#pragma omp parallel for schedule(static) if (n >= ORC_OMP_MIN_N)
  for (int64_t i = 0; i < (int64_t)n; i++) {
    uint32_t h = hash[i];
    if (i == 0 || h != hash[i - 1]) {
      cellStart[h] = (uint32_t)i;
      if (i > 0) cellEnd[hash[i - 1]] = (uint32_t)i;
    }
    if (i == (int64_t)n - 1) cellEnd[h] = (uint32_t)i + 1;
    uint32_t src = index[i];
    sortedPos[2 * i] = oldPos[2 * src];
    sortedPos[2 * i + 1] = oldPos[2 * src + 1];
    sortedVel[2 * i] = oldVel[2 * src];
    sortedVel[2 * i + 1] = oldVel[2 * src + 1];
    sortedRad[i] = oldRad[src];
  }
}

/* impl.cuh:124-181 updateRad_light_wave */
void orc_updateRad_light_wave(const OrcParams *P, const float *absForce_a, const float *absForce_r,
                              float *rad, const float *phase, float time, float dt, const int32_t *dead,
                              uint32_t n) {
#pragma omp parallel for schedule(static) if (n >= ORC_OMP_MIN_N)
  for (int64_t i = 0; i < (int64_t)n; i++) {
    if (dead[i]) continue;
    if (phase[i] > 10000000.0f) continue;
    float time1 = time + phase[i];
    const float period = (P->Nx + 1) * P->rise_period;
    if (time1 < 0) time1 = time1 + 100 * (P->Nx + 1) * P->rise_period;
    if (time1 >= period) time1 = time1 - period * floorf(time1 / period);
    if (time1 >= 2 * P->rise_period) continue;
    float target_r;
    if (time1 <= P->rise_period)
      target_r = P->min_radius + (P->max_radius - P->min_radius) / P->rise_period * time1;
    else
      target_r = P->max_radius + (P->min_radius - P->max_radius) / P->rise_period * (time1 - P->rise_period);
    const float r = rad[i];
    float dr1 = target_r - r;
    float dr = 0;
    const float max_speed = 0.1;
    float torque = dr1 * P->constraint * r / max_speed / P->max_radius / dt;
    torque = fminf(torque, P->constraint);
    if (dr1 > 0) {
      if (torque / r > absForce_r[i])
        dr = max_speed * P->max_radius / P->constraint * (torque / r - absForce_r[i]) * dt;
    } else {
      if (P->constrained_contraction) {
        if (-P->constraint_contraction * dr1 > absForce_a[i] * r)
          dr = (P->constraint_contraction * dr1 + absForce_a[i] * r) / (P->constraint_contraction);
        dr = fmaxf(dr, -P->max_radius * dt);
      } else {
        dr = dr1;
      }
    }
    dr = r + dr;
    if (dr > P->max_radius) dr = P->max_radius;
    if (dr < P->min_radius) dr = P->min_radius;
    rad[i] = dr;
  }
}

/* impl.cuh:184-209 checkIntersectionLine */
static int intersects_segment(float x0, float y0, float x1, float y1, float x3, float y3, float x4,
                              float y4) {
  if (fabsf((x4 - x3) / (x1 - x0)) == fabsf((y4 - y3) / (y1 - y0))) return 0;
  float t, t1;
  if (fabsf(y4 - y3) > 0) {
    t = (x3 - x0 - (y3 - y0) * (x3 - x4) / (y3 - y4)) *
        ((y3 - y4) / ((x1 - x0) * (y3 - y4) - (y1 - y0) * (x3 - x4)));
    if (t <= 0 || t >= 1) return 0;
    t1 = (y3 - y0 - t * (y1 - y0)) / (y3 - y4);
    if (t1 <= 0 || t1 >= 1) return 0;
  } else if (fabsf(x4 - x3) > 0) {
    t = (y3 - y0 - (x3 - x0) * (y3 - y4) / (x3 - x4)) *
        ((x3 - x4) / ((y1 - y0) * (x3 - x4) - (x1 - x0) * (y3 - y4)));
    if (t <= 0 || t >= 1) return 0;
    t1 = (x3 - x0 - t * (x1 - x0)) / (x3 - x4);
    if (t1 <= 0 || t1 >= 1) return 0;
  } else {
    return 0;
  }
  return 1;
}

/* impl.cuh:211-236 checkIntersectionCircle (powf(x,2) -> x*x, powf(D,0.5f) -> sqrtf(D); ORC_DPOW2 / ORC_DSQRT are
 * exactly that in every build but the device-powf bracket members) */
static int intersects_circle(float lx, float ly, float px, float py, float ox, float oy, float orad) {
  float C1 = ORC_DPOW2(lx) + ORC_DPOW2(ly);
  float C2 = ORC_DPOW2(px) + ORC_DPOW2(py);
  float C3 = ORC_DPOW2(ox) + ORC_DPOW2(oy);
  float C4 = lx * px + ly * py;
  float C5 = lx * ox + ly * oy;
  float C6 = px * ox + py * oy;
  float A = C1 + C2 - 2 * C4;
  float B = -2 * C1 + 2 * C4 + 2 * C5 - 2 * C6;
  float C = C1 + C3 - 2 * C5 - ORC_DPOW2(orad);
  float D = ORC_DPOW2(B) - 4 * A * C;
  if (D >= 0) {
    float R1 = (-B + ORC_DSQRT(D)) / 2 / A;
    float R2 = (-B - ORC_DSQRT(D)) / 2 / A;
    if (R1 > 0 && R1 < 1) return 1;
    if (R2 > 0 && R2 < 1) return 1;
  }
  return 0;
}

/* impl.cuh:238-262 checkIntersection */
static int in_shadow(const OrcParams *P, float px, float py) {
  for (int i = 0; i < P->n_cir_obstacles; i++)
    if (intersects_circle(P->light_x, P->light_y, px, py, P->x_cir_obs[i], P->y_cir_obs[i], P->r_cir_obs[i]))
      return 1;
  for (int i = 0; i < P->nobstacles; i++) {
    const float x1 = P->x1obs[i], x2 = P->x2obs[i], y1 = P->y1obs[i], y2 = P->y2obs[i];
    if (intersects_segment(P->light_x, P->light_y, px, py, x1, y1, x1, y2)) return 1; /* left */
    if (intersects_segment(P->light_x, P->light_y, px, py, x1, y2, x2, y2)) return 1; /* top */
    if (intersects_segment(P->light_x, P->light_y, px, py, x2, y2, x2, y1)) return 1; /* right */
    if (intersects_segment(P->light_x, P->light_y, px, py, x2, y1, x1, y1)) return 1; /* bottom */
  }
  return 0;
}

/* impl.cuh:264-290 updatePhase */
void orc_updatePhase(const OrcParams *P, const float *pos, float *phase, float spacing, float max_d,
                     float min_d, uint32_t n) {
  (void)max_d;
#pragma omp parallel for schedule(static) if (n >= ORC_OMP_MIN_N)
  for (int64_t i = 0; i < (int64_t)n; i++) {
    const float px = pos[2 * i], py = pos[2 * i + 1];
    int visible = 1;
    float dist = len2(px - P->light_x, py - P->light_y);
    if (P->light_shadow) {
      if (in_shadow(P, px, py)) visible = 0;
    }
    if (!visible) {
      if (P->light_shadow == 1) phase[i] = -(P->Nx - 1) * P->rise_period;
      if (P->light_shadow == 2) phase[i] = 9999999999.0f;
    } else {
      phase[i] = (min_d - dist) / (spacing)*P->rise_period;
    }
  }
}

#if defined(ORC_BRACKET_FMA)
#pragma GCC pop_options /* host code */
#endif
/* particlebot.cpp:215-228 host loop */
void orc_minmax_light_distance(const OrcParams *P, const float *pos, uint32_t n, float *min_d,
                               float *max_d) {
  float mn = 0, mx = 0;
  for (uint32_t i = 0; i < n; i++) {
    float dist = powf(powf(P->light_x - pos[i * 2], 2) + powf(P->light_y - pos[i * 2 + 1], 2), 0.5f);
    if (i == 0) {
      mx = dist;
      mn = dist;
    } else {
      mn = (mn < dist ? mn : dist);
      mx = (mx > dist ? mx : dist);
    }
  }
  *min_d = mn;
  *max_d = mx;
}
#if defined(ORC_BRACKET_FMA)
#pragma GCC push_options /* device code again */
#pragma GCC optimize("fp-contract=fast")
#pragma GCC target("fma")
#endif

/* impl.cuh:541-594 collideSpheres */
static inline void pair_force(const OrcParams *P, float ax, float ay, float bx, float by, float avx,
                              float avy, float bvx, float bvy, float ra, float rb, float attraction,
                              float *fx, float *fy, float *fa, float *fr) {
  const float rx = bx - ax, ry = by - ay;
  const float dist = len2(rx, ry);
  const float collideDist = ra + rb;
  float tx = 0.0f, ty = 0.0f;
  if (dist < collideDist) {
    const float nx = rx / dist, ny = ry / dist;
    const float rvx = bvx - avx, rvy = bvy - avy;
    const float vn = dot2(rvx, rvy, nx, ny);
    const float tvx = rvx - vn * nx, tvy = rvy - vn * ny;
    const float ks = -P->spring * (collideDist - dist);
    tx += ks * nx;
    ty += ks * ny;
    tx += P->damping * rvx;
    ty += P->damping * rvy;
    tx += P->shear * tvx;
    ty += P->shear * tvy;
    *fx += tx;
    *fy += ty;
    *fr += len2(tx, ty);
  } else {
    const float int1 = 0.0009;
    const float int2 = 0.0019;
    const float min_attr = 2.5f;
    const float gap = dist - collideDist;
    if (gap < int1) {
      tx += min_attr * (rx / dist);
      ty += min_attr * (ry / dist);
    } else if (gap < int2) {
      const float c = min_attr + (attraction / ORC_POW2(int2) - min_attr) / (int2 - int1) * (gap - int1); /* :586 __powf */
      tx += c * (rx / dist);
      ty += c * (ry / dist);
    } else {
      const float g2 = ORC_POW2(gap); /* :589 __powf(dist - collideDist, 2.0f) */
      tx += attraction * (rx / dist) / g2;
      ty += attraction * (ry / dist) / g2;
    }
    *fx += tx;
    *fy += ty;
    *fa += len2(tx, ty);
  }
}

void orc_collideSpheres(const OrcParams *P, const float posA[2], const float posB[2],
                        const float velA[2], const float velB[2], float radA, float radB,
                        float attraction, float force[2], float *forcea, float *forcer) {
  pair_force(P, posA[0], posA[1], posB[0], posB[1], velA[0], velA[1], velB[0], velB[1], radA, radB,
             attraction, &force[0], &force[1], forcea, forcer);
}

/* contact with a wall/obstacle along unit vector (dx,dy), spring magnitude ks (already signed):
 * the common tail of impl.cuh:711-726 and :781-797 */
static inline void obstacle_tail(const OrcParams *P, float vx, float vy, float dx, float dy, float sx,
                                 float sy, float *fx, float *fy, float *fr) {
  const float rvx = -vx, rvy = -vy;
  const float vn = dot2(rvx, rvy, dx, dy);
  const float tvx = rvx - vn * dx, tvy = rvy - vn * dy;
  float tx = 0.0f, ty = 0.0f;
  tx += sx;
  ty += sy;
  tx += P->damping * rvx;
  ty += P->damping * rvy;
  tx += P->shear * tvx;
  ty += P->shear * tvy;
  *fx += tx;
  *fy += ty;
  *fr += len2(tx, ty);
}

/* impl.cuh:657-831 collideD, with collideCell (:597-653) inlined */
void orc_collide(const OrcParams *P, float *newVel, float *absForce_a, float *absForce_r,
                 const float *sortedPos, const float *sortedVel, const float *sortedRad,
                 const uint32_t *index, const uint32_t *cellStart, const uint32_t *cellEnd, uint32_t n,
                 float dt) {
  const int payloadMode = (P->nDead == -1);
  const uint32_t payloadIdx = P->nCells - 1;
#pragma omp parallel for schedule(dynamic, 256) if (n >= ORC_OMP_MIN_N)
  for (int64_t ii = 0; ii < (int64_t)n; ii++) {
    const uint32_t i = (uint32_t)ii;
    const float px = sortedPos[2 * i], py = sortedPos[2 * i + 1];
    float vx = sortedVel[2 * i], vy = sortedVel[2 * i + 1];
    const float rad = sortedRad[i];
    int gx, gy;
    grid_pos(P, px, py, &gx, &gy);
    float fx = 0.0f, fy = 0.0f;
    const uint32_t orig = index[i];
    float fa = 0.0f;
    float fr = 0.0f * absForce_r[orig]; /* impl.cuh:688 (NaN-propagating on purpose) */
#if defined(ORC_BRACKET_ORDER)
    float cfx[64], cfy[64], cfr[64];
    int ncontact = 0;
#endif
    const int selfPayload = payloadMode && orig == payloadIdx;

    for (int y = -2; y <= 2; y++) {
      for (int x = -2; x <= 2; x++) {
        const uint32_t h = grid_hash(P, gx + x, gy + y);
        const uint32_t start = cellStart[h];
        if (start == 0xffffffffu) continue;
        const uint32_t end = cellEnd[h];
        const float att1 = selfPayload ? P->attractionFactor : 1.0f;
        for (uint32_t j = start; j < end; j++) {
          if (j == i) continue;
          float att2 = 1.0f;
          if (payloadMode && index[j] == payloadIdx) att2 = P->attractionFactor;
#if defined(ORC_BRACKET_ORDER)
          /* the order of additions of the product's two-pass tolerance kernel (k_force_stream): a contact's term is
           * not added where the reference adds it but after the bot's last candidate, contacts among themselves in
           * the reference's order; every term is the oracle's own, bit for bit */
          {
            float tfx = 0.0f, tfy = 0.0f, tfa = 0.0f, tfr = 0.0f;
            pair_force(P, px, py, sortedPos[2 * j], sortedPos[2 * j + 1], vx, vy, sortedVel[2 * j],
                       sortedVel[2 * j + 1], rad, sortedRad[j], P->attraction * att2 * att1, &tfx, &tfy, &tfa, &tfr);
            if (tfr != 0.0f) { /* a contact (a contact whose term is exactly zero can be added anywhere) */
              if (ncontact < 64) {
                cfx[ncontact] = tfx;
                cfy[ncontact] = tfy;
                cfr[ncontact] = tfr;
                ncontact++;
              } else { /* (more than 64 contacts: in place) */
                fx += tfx;
                fy += tfy;
                fr += tfr;
              }
            } else {
              fx += tfx;
              fy += tfy;
              fa += tfa;
            }
          }
#else
          pair_force(P, px, py, sortedPos[2 * j], sortedPos[2 * j + 1], vx, vy, sortedVel[2 * j],
                     sortedVel[2 * j + 1], rad, sortedRad[j], P->attraction * att2 * att1, &fx, &fy, &fa,
                     &fr);
#endif
        }
      }
    }
#if defined(ORC_BRACKET_ORDER)
    for (int c = 0; c < ncontact; c++) {
      fx += cfx[c];
      fy += cfy[c];
      fr += cfr[c];
    }
#endif

    /* circular obstacles, impl.cuh:703-728 */
    for (int k = 0; k < P->n_cir_obstacles; k++) {
      const float ox = P->x_cir_obs[k], oy = P->y_cir_obs[k], orad = P->r_cir_obs[k];
      const float ddx = px - ox, ddy = py - oy;
      const float dist_2 = ORC_DPOW2(ddx) + ORC_DPOW2(ddy); /* :704 powf */
      const float reach = rad + orad;
      if (dist_2 < ORC_DPOW2(reach)) { /* :705 powf */
        float dx = -px + ox, dy = -py + oy;
        const float l = len2(dx, dy);
        dx = dx / l;
        dy = dy / l;
        const float ks = 2.0f * P->spring * (rad + orad - ORC_DSQRT(dist_2)); /* :719 powf(dist_2, 0.5f) */
        obstacle_tail(P, vx, vy, dx, dy, ks * (-dx), ks * (-dy), &fx, &fy, &fr);
      }
    }

    /* rectangular obstacles, impl.cuh:729-798 */
    {
      float dx = 0.0f, dy = 0.0f, overlap = 0.0f;
      for (int k = 0; k < P->nobstacles; k++) {
        const float x1 = P->x1obs[k], x2 = P->x2obs[k], y1 = P->y1obs[k], y2 = P->y2obs[k];
        int hit = 0;
        if (py > y1 && py < y2) {
          if (px > x1 - rad && px < x2 - rad) {
            hit = 1;
            dx = 1.0f;
            dy = 0.0f;
            overlap = px - x1 + rad;
          }
          if (px < x2 + rad && px > x1 + rad) {
            hit = 1;
            dx = -1.0f;
            dy = 0.0f;
            overlap = -px + x2 + rad;
          }
        } else if (px > x1 && px < x2) {
          if (py > y1 - rad && py < y2 - rad) {
            hit = 1;
            dx = 0.0f;
            dy = 1.0f;
            overlap = py - y1 + rad;
          }
          if (py < y2 + rad && py > y1 + rad) {
            hit = 1;
            dx = 0.0f;
            dy = -1.0f;
            overlap = -py + y2 + rad;
          }
        } else {
          /* corners in the reference's order: (x2,y2) (x1,y2) (x1,y1) (x2,y1) */
          const float cxs[4] = {x2, x1, x1, x2};
          const float cys[4] = {y2, y2, y1, y1};
          for (int c = 0; c < 4; c++) {
            const float ex = px - cxs[c], ey = py - cys[c];
            const float d2 = ORC_DPOW2(ex) + ORC_DPOW2(ey); /* :757-779 powf */
            if (d2 < ORC_DPOW2(rad)) {
              const float l = len2(ex, ey);
              dx = -ex / l;
              dy = -ey / l;
              hit = 1;
              overlap = rad - ORC_DSQRT(d2);
              break;
            }
          }
        }
        if (hit) {
          const float ks = -2.0f * P->spring * overlap;
          obstacle_tail(P, vx, vy, dx, dy, ks * dx, ks * dy, &fx, &fy, &fr);
        }
      }
    }

    /* friction + velocity update, impl.cuh:801-825 */
    float friction = P->friction;
    float gravity = P->gravity;
    if (selfPayload) {
      friction *= P->frictionFactor;
      gravity *= P->massFactor;
    }
    if (len2(vx, vy) < 0.000001f && len2(fx, fy) < (2.0f * friction * gravity)) {
      fx = 0.0f;
      fy = 0.0f;
    }
    if (selfPayload) {
      vx = vx + fx / P->massFactor * dt;
      vy = vy + fy / P->massFactor * dt;
    } else {
      vx = vx + fx * dt;
      vy = vy + fy * dt;
    }
    const float fric = friction * gravity * dt;
    const float speed = len2(vx, vy);
    if (speed < fric) {
      vx = 0.0f;
      vy = 0.0f;
    } else {
      vx -= fric * (vx / speed);
      vy -= fric * (vy / speed);
    }
    newVel[2 * orig] = vx;
    newVel[2 * orig + 1] = vy;
    absForce_a[orig] = fa;
    absForce_r[orig] = fr;
  }
}

#if defined(ORC_BRACKET_FMA)
#pragma GCC pop_options
#endif
/* <<< end of device code <<< (the phase-noise generators below are this repository's own or a
 * restatement of cuRAND's: not part of what the bracket measures; bracket runs use phase_std 0) */

/* ------------------------------------------------------------------------------------------ */
/* phase noise: PB-RNG v1 (own counter RNG + polynomial Box-Muller; deterministic in fp32)      */
/* ------------------------------------------------------------------------------------------ */

static inline uint64_t mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}

float orc_normal(uint32_t seed, uint32_t i, uint32_t k) {
  uint64_t x = (((uint64_t)seed << 32) | (uint64_t)i) + 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1u);
  x = mix64(x);
  x = mix64(x ^ 0xD1342543DE82EF95ull);
  const uint32_t k1 = (uint32_t)(x >> 40) & 0xFFFFFFu;
  const uint32_t k2 = (uint32_t)(x >> 8) & 0xFFFFFFu;
  /* ln(u1), u1 = (k1+1) * 2^-24 in (0,1] */
  const uint32_t v = k1 + 1u;
  int e = 31 - __builtin_clz(v);
  float m = (float)v * ldexpf(1.0f, -e); /* exact, in [1,2) */
  if (m > 1.41421356f) {
    m = m * 0.5f;
    e += 1;
  }
  const float t = (m - 1.0f) / (m + 1.0f);
  const float t2 = t * t;
  float p = 0.111111111f;
  p = p * t2 + 0.142857143f;
  p = p * t2 + 0.2f;
  p = p * t2 + 0.333333333f;
  p = p * t2 + 1.0f;
  const float lnu = 2.0f * t * p + (float)(e - 24) * 0.693147181f;
  const float r = sqrtf(-2.0f * lnu);
  /* cos(2*pi*u2), u2 = k2 * 2^-24: quadrant from the top two bits, polynomial on [0, pi/2) */
  const uint32_t q = k2 >> 22;
  const float a = (float)(k2 & 0x3FFFFFu) * (1.0f / 4194304.0f) * 1.57079633f;
  const float a2 = a * a;
  float s = -2.50521084e-8f;
  s = s * a2 + 2.75573192e-6f;
  s = s * a2 - 1.98412698e-4f;
  s = s * a2 + 8.33333333e-3f;
  s = s * a2 - 1.66666667e-1f;
  s = s * a2 + 1.0f;
  s = s * a;
  float c = 2.08767570e-9f;
  c = c * a2 - 2.75573192e-7f;
  c = c * a2 + 2.48015873e-5f;
  c = c * a2 - 1.38888889e-3f;
  c = c * a2 + 4.16666667e-2f;
  c = c * a2 - 0.5f;
  c = c * a2 + 1.0f;
  float cv;
  switch (q) {
    case 0: cv = c; break;
    case 1: cv = -s; break;
    case 2: cv = -c; break;
    default: cv = s; break;
  }
  return r * cv;
}

/* replaces impl.cuh:43-51 add_normal_noise_kernel: val[i] += std * N(0,1) */
void orc_add_normal_noise(uint32_t seed, uint32_t draw, float *val, float std, uint32_t n) {
#pragma omp parallel for schedule(static) if (n >= ORC_OMP_MIN_N)
  for (int64_t i = 0; i < (int64_t)n; i++) {
    float noise = std * orc_normal(seed, (uint32_t)i, draw);
    val[i] += noise;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* phase noise, generator 2: cuRAND-compatible XORWOW (particlebot_kernel_impl.cuh:36-51:       */
/* curand_init(seed, i, 0) per bot, curand_normal per phase update).  cuRAND is a third-party    */
/* dependency absent from the reference tree and from this image; this restates its PUBLISHED    */
/* algorithm (curand_kernel.h: curandStateXORWOW, _curand_init_scratch, curand(), curand_normal) */
/* independently of the product's csrc/pb_xorwow.hpp: the 2^67-step subsequence jump is built    */
/* here as a matrix of OUTPUT-bit masks and applied by AND + parity, bots are initialised         */
/* SEQUENTIALLY (state of bot i+1 = one jump of bot i), where the product uses input-bit images  */
/* and a binary decomposition of the bot index.  kind 1: cuRAND's seeding constants; kind 2:      */
/* rocRAND's (rocrand_xorwow.h:113-116), which rocRAND's own host engine pins (tests/test_xorwow).*/
/* The uniform -> normal transform uses fixed-order fp32 polynomials (log, sin, cos) so that CPU   */
/* and GPU agree bit for bit; against CUDA's logf/__sincosf it agrees to float rounding only.     */
/* ------------------------------------------------------------------------------------------ */

typedef struct OrcXw {
  uint32_t v[5];
  uint32_t d;
  int32_t have;
  float extra;
} OrcXw;

static void xw_shift(uint32_t *v) {
  uint32_t t = v[0] ^ (v[0] >> 2);
  v[0] = v[1];
  v[1] = v[2];
  v[2] = v[3];
  v[3] = v[4];
  v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

static uint32_t xw_next(OrcXw *s) {
  xw_shift(s->v);
  s->d += 362437u;
  return s->v[4] + s->d;
}

static void xw_seed(OrcXw *s, uint64_t seed, int kind) {
  const uint32_t x0 = kind == 2 ? 0x2c7f967fu : 0xaad26b49u, x1 = kind == 2 ? 0xa03697cbu : 0xf7dcefddu;
  const uint32_t m0 = kind == 2 ? 1228688033u : 1099087573u, m1 = kind == 2 ? 2073658381u : 2591861531u;
  const uint32_t s0 = (uint32_t)seed ^ x0, s1 = (uint32_t)(seed >> 32) ^ x1;
  const uint32_t t0 = m0 * s0, t1 = m1 * s1;
  s->d = 6615241u + t1 + t0;
  s->v[0] = 123456789u + t0;
  s->v[1] = 362436069u ^ t0;
  s->v[2] = 521288629u + t1;
  s->v[3] = 88675123u ^ t1;
  s->v[4] = 5783321u + t0;
  s->have = 0;
  s->extra = 0.0f;
}

/* J[o] = 160-bit mask (5 words) of the INPUT bits whose parity is output bit o after 2^67 steps */
static uint32_t g_xwJump[160][5];
static int g_xwJumpReady = 0;

static void xw_build_jump(void) {
  if (g_xwJumpReady) return;
  static uint32_t A[160][5], B[160][5];
  memset(A, 0, sizeof A);
  for (int i = 0; i < 160; i++) { /* one step applied to e_i tells which outputs depend on input i */
    uint32_t v[5] = {0, 0, 0, 0, 0};
    v[i >> 5] = 1u << (i & 31);
    xw_shift(v);
    for (int o = 0; o < 160; o++)
      if ((v[o >> 5] >> (o & 31)) & 1u) A[o][i >> 5] |= 1u << (i & 31);
  }
  for (int sq = 0; sq < 67; sq++) { /* (A o A)[o] = XOR of A[i] over the inputs i of A[o] */
    for (int o = 0; o < 160; o++) {
      uint32_t r[5] = {0, 0, 0, 0, 0};
      for (int i = 0; i < 160; i++)
        if ((A[o][i >> 5] >> (i & 31)) & 1u)
          for (int w = 0; w < 5; w++) r[w] ^= A[i][w];
      memcpy(B[o], r, sizeof r);
    }
    memcpy(A, B, sizeof A);
  }
  memcpy(g_xwJump, A, sizeof A);
  g_xwJumpReady = 1;
}

static void xw_jump(uint32_t *v) { /* v <- state 2^67 steps later */
  uint32_t r[5] = {0, 0, 0, 0, 0};
  for (int o = 0; o < 160; o++) {
    uint32_t acc = 0;
    for (int w = 0; w < 5; w++) acc ^= g_xwJump[o][w] & v[w];
    r[o >> 5] |= (uint32_t)(__builtin_popcount(acc) & 1) << (o & 31);
  }
  memcpy(v, r, sizeof r);
}

static float xw_log(float u) {
  uint32_t bits;
  memcpy(&bits, &u, 4);
  int e = (int)(bits >> 23) - 127;
  uint32_t mb = (bits & 0x007FFFFFu) | 0x3F800000u;
  float m;
  memcpy(&m, &mb, 4);
  if (m > 1.41421356f) {
    m = m * 0.5f;
    e += 1;
  }
  const float t = (m - 1.0f) / (m + 1.0f);
  const float t2 = t * t;
  float p = 0.111111111f;
  p = p * t2 + 0.142857143f;
  p = p * t2 + 0.2f;
  p = p * t2 + 0.333333333f;
  p = p * t2 + 1.0f;
  return 2.0f * t * p + (float)e * 0.693147181f;
}

static void xw_sincos(float v, float *sn, float *cs) {
  int q = (int)(v * 0.636619772f);
  float a = v - (float)q * 1.57079633f;
  if (a < 0.0f) {
    q -= 1;
    a = a + 1.57079633f;
  }
  const float a2 = a * a;
  float s = -2.50521084e-8f;
  s = s * a2 + 2.75573192e-6f;
  s = s * a2 - 1.98412698e-4f;
  s = s * a2 + 8.33333333e-3f;
  s = s * a2 - 1.66666667e-1f;
  s = s * a2 + 1.0f;
  s = s * a;
  float c = 2.08767570e-9f;
  c = c * a2 - 2.75573192e-7f;
  c = c * a2 + 2.48015873e-5f;
  c = c * a2 - 1.38888889e-3f;
  c = c * a2 + 4.16666667e-2f;
  c = c * a2 - 0.5f;
  c = c * a2 + 1.0f;
  switch (q & 3) {
    case 0: *sn = s, *cs = c; break;
    case 1: *sn = c, *cs = -s; break;
    case 2: *sn = -s, *cs = -c; break;
    default: *sn = -c, *cs = s; break;
  }
}

/* curand_normal(curandStateXORWOW_t*): Box-Muller on two consecutive outputs, second value kept */
static float xw_normal(OrcXw *s, int kind) {
  if (s->have) {
    s->have = 0;
    return s->extra;
  }
  const uint32_t x = xw_next(s);
  const uint32_t y = xw_next(s);
  float u, v;
  if (kind == 2) { /* rocrand_normal.h:56-57 */
    u = 2.3283064e-10f + ((float)x * 2.3283064e-10f);
    v = 1.46291807e-09f + ((float)y * 1.46291807e-09f);
  } else { /* CURAND_2POW32_INV, CURAND_2POW32_INV_2PI and their halves */
    const float c = 2.3283064e-10f, c2pi = 2.3283064e-10f * 6.2831855f;
    u = (float)x * c + (c / 2.0f);
    v = (float)y * c2pi + (c2pi / 2.0f);
  }
  const float r = sqrtf(-2.0f * xw_log(u));
  float sn, cs;
  xw_sincos(v, &sn, &cs);
  s->extra = cs * r;
  s->have = 1;
  return sn * r;
}

/* curand_setup_kernel (impl.cuh:36-41): st[i] = curand_init(seed, i, 0), i = 0..n-1 */
static void xw_setup(OrcXw *st, uint32_t seed, uint32_t n, int kind) {
  xw_build_jump();
  if (!n) return;
  xw_seed(&st[0], (uint64_t)seed, kind);
  for (uint32_t i = 1; i < n; i++) {
    st[i] = st[i - 1];
    xw_jump(st[i].v); /* d is unchanged: 362437 * 2^67 = 0 mod 2^32 */
  }
}

/* test hooks: the first `count` raw outputs of curand_init(seed, subsequence, 0), and the states */
void orc_xorwow_outputs(int kind, uint64_t seed, uint32_t subsequence, uint32_t count, uint32_t *out) {
  OrcXw s;
  xw_build_jump();
  xw_seed(&s, seed, kind);
  for (uint32_t i = 0; i < subsequence; i++) xw_jump(s.v);
  for (uint32_t i = 0; i < count; i++) out[i] = xw_next(&s);
}

void orc_xorwow_normals(int kind, uint32_t seed, uint32_t nbots, uint32_t draws, float *out /* draws x nbots */) {
  OrcXw *st = (OrcXw *)malloc(sizeof(OrcXw) * (nbots ? nbots : 1));
  xw_setup(st, seed, nbots, kind);
  for (uint32_t k = 0; k < draws; k++)
    for (uint32_t i = 0; i < nbots; i++) out[(size_t)k * nbots + i] = xw_normal(&st[i], kind);
  free(st);
}

/* the 2^67-step jump as the product stores it (row r = image of input bit r), for cross-checks */
void orc_xorwow_jump_rows(uint32_t *rows /* 160 x 5 */) {
  xw_build_jump();
  for (int r = 0; r < 160; r++) {
    uint32_t v[5] = {0, 0, 0, 0, 0};
    v[r >> 5] = 1u << (r & 31);
    xw_jump(v);
    memcpy(rows + (size_t)r * 5, v, sizeof v);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* simulation object                                                                            */
/* ------------------------------------------------------------------------------------------ */

struct OrcSim {
  OrcParams P;
  uint32_t n;
  float time;
  uint32_t phaseDraws;
  int forceSort; /* test helper, see orc_sim_force_sort_once */
  float *pos, *vel, *rad, *phase, *absA, *absR;
  int32_t *dead;
  uint32_t *hash, *index, *cellStart, *cellEnd;
  float *sPos, *sVel, *sRad;
  OrcXw *xw; /* per-bot XORWOW states when P.rngKind != 0 (the reference's dState) */
};

static void *zalloc(size_t bytes) {
  void *p = calloc(1, bytes ? bytes : 1);
  if (!p) {
    fprintf(stderr, "pb_oracle: out of memory\n");
    exit(EXIT_FAILURE);
  }
  return p;
}

OrcSim *orc_sim_create(const OrcParams *P) {
  /* main.cpp:929 srand(params.seed); particlebot.cpp:77-166 _initialize.  absForce_* are
   * zero-initialised (the reference reads uninitialised memory at step 0). */
  OrcSim *s = (OrcSim *)zalloc(sizeof(OrcSim));
  s->P = *P;
  s->n = P->nCells;
  const size_t n = s->n;
  srand(P->seed);
  s->pos = (float *)zalloc(8 * n);
  s->vel = (float *)zalloc(8 * n);
  s->rad = (float *)zalloc(4 * n);
  s->phase = (float *)zalloc(4 * n);
  s->absA = (float *)zalloc(4 * n);
  s->absR = (float *)zalloc(4 * n);
  s->dead = (int32_t *)zalloc(4 * n);
  s->hash = (uint32_t *)zalloc(4 * n);
  s->index = (uint32_t *)zalloc(4 * n);
  s->cellStart = (uint32_t *)zalloc(4 * (size_t)P->numCells);
  s->cellEnd = (uint32_t *)zalloc(4 * (size_t)P->numCells);
  s->sPos = (float *)zalloc(8 * n);
  s->sVel = (float *)zalloc(8 * n);
  s->sRad = (float *)zalloc(4 * n);
  if (P->rngKind != 0) { /* particlebot.cpp:161-165 curand_setup(dState, nCells) */
    s->xw = (OrcXw *)zalloc(sizeof(OrcXw) * (n ? n : 1));
    xw_setup(s->xw, P->seed, s->n, P->rngKind);
  }
  return s;
}

void orc_sim_destroy(OrcSim *s) {
  if (!s) return;
  free(s->pos);
  free(s->vel);
  free(s->rad);
  free(s->phase);
  free(s->absA);
  free(s->absR);
  free(s->dead);
  free(s->hash);
  free(s->index);
  free(s->cellStart);
  free(s->cellEnd);
  free(s->sPos);
  free(s->sVel);
  free(s->sRad);
  free(s->xw);
  free(s);
}

static inline float frand01(void) { return rand() / (float)RAND_MAX; } /* particlebot.cpp:27-30 */

/* placement cell of a point, particlebot.cpp:635-636 and friends */
static inline int place_cell(float v, float origin, float cell, uint32_t gsz) {
  return ((int)floorf((v - origin) / cell)) & (int)(gsz - 1);
}

/* particlebot.cpp:612-748 CONFIG_RANDOM.  The reference keeps a 512x512 vector<vector<int>>;
 * here a head/next chain per cell (iteration order is irrelevant: the scans only ask "any
 * overlap?").  Cells outside [0,grid) (the reference indexes them out of bounds) are empty. */
static void place_random(OrcSim *s) {
  const OrcParams *P = &s->P;
  const uint32_t n = s->n;
  const uint32_t GX = P->gridSizeX, GY = P->gridSizeY;
  if (n == 0) return;
  int32_t *head = (int32_t *)malloc(sizeof(int32_t) * (size_t)GX * GY);
  int32_t *next = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
  for (size_t c = 0; c < (size_t)GX * GY; c++) head[c] = -1;
#define CELL_PUSH(XG, YG, I)                       \
  do {                                             \
    size_t c_ = (size_t)(XG) * GY + (size_t)(YG);  \
    next[(I)] = head[c_];                          \
    head[c_] = (int32_t)(I);                       \
  } while (0)
  float *hPos = s->pos;
  uint32_t p = 0;
  int xg, yg, xgs, ygs;
  uint32_t placed = 0, start_ind = 0;
  const uint32_t max_unsuccessful_placements = 200;
  uint32_t unsuccessful_placements = 0;
  hPos[p++] = 5.0;
  hPos[p++] = 0.0;
  xg = place_cell(0.0f, P->worldOriginX, P->cellSizeX, GX);
  yg = place_cell(0.0f, P->worldOriginY, P->cellSizeY, GY);
  CELL_PUSH(xg, yg, 0);
  float x = 0, y = 0, theta = 0, r = 0, old_theta = 0;
  float min_x = 9999999.0;
  const float increment_theta = 2 * ORC_PI_F / 360.0 * 10.0;
  const double two_rmin = 2 * 1.0 * P->min_radius;
  for (uint32_t i = 1; i < n; i++) {
    if (i == 2) {
      int j = rand() % 2;
      float dx = hPos[2] - hPos[0], dy = hPos[3] - hPos[1];
      float l = host_length(dx, dy);
      dy = dy / l;
      dx = dx / l;
      float ex, ey;
      if (j) {
        ex = dy;
        ey = -dx;
      } else {
        ex = -dy;
        ey = dx;
      }
      x = (hPos[2] + hPos[0]) / 2.0f + ex * P->min_radius;
      y = (hPos[3] + hPos[1]) / 2.0f + ey * P->min_radius;
      if (x < min_x) min_x = x;
      hPos[p++] = x;
      hPos[p++] = y;
      xg = place_cell(hPos[2 * i], P->worldOriginX, P->cellSizeX, GX);
      yg = place_cell(hPos[2 * i + 1], P->worldOriginY, P->cellSizeY, GY);
      CELL_PUSH(xg, yg, i);
      continue;
    }
    placed = 0;
    r = P->min_radius;
    while (!placed) {
      start_ind = (uint32_t)rand() % i;
      placed = 1;
      if (unsuccessful_placements == max_unsuccessful_placements) {
        unsuccessful_placements = 0;
        r += P->min_radius;
      }
      theta = 2 * frand01() * ORC_PI_F;
      x = hPos[2 * start_ind] + 2 * r * cosf(theta);
      y = hPos[2 * start_ind + 1] + 2 * r * sinf(theta);
      xgs = place_cell(x, P->worldOriginX, P->cellSizeX, GX);
      ygs = place_cell(y, P->worldOriginY, P->cellSizeY, GY);
      for (xg = xgs - 1; (xg <= xgs + 1) & placed; xg++) {
        for (yg = ygs - 1; (yg <= ygs + 1) & placed; yg++) {
          if (xg < 0 || yg < 0 || xg >= (int)GX || yg >= (int)GY) continue;
          for (int32_t it = head[(size_t)xg * GY + yg]; it >= 0; it = next[it]) {
            if (host_length(x - hPos[2 * it], y - hPos[2 * it + 1]) < two_rmin) {
              placed = 0;
              unsuccessful_placements++;
              break;
            }
          }
        }
      }
      if (!placed) continue;
      old_theta = theta;
      int flag = 0;
      while (theta - old_theta < 2 * ORC_PI_F) {
        theta += increment_theta;
        x = hPos[2 * start_ind] + 2 * r * cosf(theta);
        y = hPos[2 * start_ind + 1] + 2 * r * sinf(theta);
        xgs = place_cell(x, P->worldOriginX, P->cellSizeX, GX);
        ygs = place_cell(y, P->worldOriginY, P->cellSizeY, GY);
        for (xg = xgs - 1; xg <= xgs + 1; xg++) {
          for (yg = ygs - 1; yg <= ygs + 1; yg++) {
            if (xg < 0 || yg < 0 || xg >= (int)GX || yg >= (int)GY) continue;
            for (int32_t it = head[(size_t)xg * GY + yg]; it >= 0; it = next[it]) {
              if (host_length(x - hPos[2 * it], y - hPos[2 * it + 1]) < two_rmin) {
                flag = 1;
                break;
              }
            }
          }
        }
        if (flag) {
          theta -= increment_theta;
          break;
        }
      }
      x = hPos[2 * start_ind] + 2 * r * cosf(theta);
      y = hPos[2 * start_ind + 1] + 2 * r * sinf(theta);
    }
    if (x < min_x) min_x = x;
    if (P->nDead == -1 && i == n - 1) {
      x = min_x - 1 * P->min_radius * P->radFactor - 2 * P->min_radius;
      y = 0;
    }
    hPos[p++] = x;
    hPos[p++] = y;
    xg = place_cell(x, P->worldOriginX, P->cellSizeX, GX);
    yg = place_cell(y, P->worldOriginY, P->cellSizeY, GY);
    CELL_PUSH(xg, yg, i);
  }
#undef CELL_PUSH
  free(head);
  free(next);
}

/* particlebot.cpp:438-481 initHexGrid (unreachable from a .cfg in the reference; used here for
 * the synthetic large-arena workload) */
static int place_hex(OrcSim *s, float spacing) { /* returns particlebotConfigSize.x (:479) */
  const uint32_t n = s->n;
  float *hPos = s->pos;
  const float h = powf(3, 0.5f) * 0.5f;
  const float dirs[7][3] = {{1.0, 0.0, 0.0}, {0.5, 0.0, h},   {-0.5, 0.0, h}, {-1.0, 0.0, 0.0},
                            {-0.5, 0.0, -h}, {0.5, 0.0, -h}, {1.0, 0.0, 0.0}};
  if (n == 0) return 2;
  uint32_t i = 0;
  hPos[0] = 0.0f;
  hPos[1] = 0.0f;
  i++;
  int n_ring = 1;
  while (i < n) {
    for (int k = 0; k < 6; k++) {
      for (int j = 0; j < n_ring; j++) {
        hPos[i * 2] = dirs[k][0] * (n_ring - j) * spacing + dirs[k + 1][0] * spacing * j;
        hPos[i * 2 + 1] = dirs[k][2] * (n_ring - j) * spacing + dirs[k + 1][2] * spacing * j;
        i++;
        if (i == n) break;
      }
      if (i == n) break;
    }
    n_ring++;
  }
  return n_ring * 2;
}

void orc_sim_reset(OrcSim *s, int use_hex) {
  /* particlebot.cpp:485-801 */
  const OrcParams *P = &s->P;
  const uint32_t n = s->n;
  s->time = 0;
  s->phaseDraws = 0;
  memset(s->vel, 0, 8 * (size_t)n);
  int configSizeX; /* particlebotConfigSize.x */
  if (use_hex) {
    configSizeX = place_hex(s, P->min_radius * 2.0f); /* :759-760, then :479 inside initHexGrid */
  } else {
    place_random(s);
    configSizeX = (int)ceilf(powf((float)n, 1.0f / 2.0f)); /* :624 */
  }
  /* particlebot.cpp:772-773: a zero Nx (unreachable from a .cfg, reachable through SimParams) falls
   * back to particlebotConfigSize.x */
  if (!s->P.Nx) s->P.Nx = configSizeX;
  for (uint32_t i = 0; i < n; i++) {
    s->rad[i] = P->min_radius;
    if (P->nDead == -1 && i == n - 1) {
      s->rad[i] = P->min_radius * P->radFactor;
      s->dead[i] = 1;
    }
    s->phase[i] = 0;
  }
}

int orc_sim_update(OrcSim *s, float dt, float sort_interval) {
  /* particlebot.cpp:170-300 */
  OrcParams *P = &s->P;
  const uint32_t n = s->n;
  if (s->time > P->max_time) return 1;

  if (s->time >= P->time_to_dead && s->time < P->time_to_dead + dt) {
    /* :178-194 draw nDead distinct bots with rand() % remaining + erase */
    int count = 0;
    uint32_t remaining = n;
    int32_t *inds = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) inds[i] = (int32_t)i;
    while (count < P->nDead) {
      uint32_t i = (uint32_t)((unsigned long)rand() % (unsigned long)remaining);
      s->dead[inds[i]] = 1;
      memmove(&inds[i], &inds[i + 1], sizeof(int32_t) * (remaining - i - 1));
      remaining--;
      count++;
    }
    free(inds);
  }

  if (P->control == 0 /* LIGHT_WAVE */) {
    if (s->time - P->phase_update_interval * floorf(s->time / P->phase_update_interval) < dt) {
      float min_d, max_d;
      orc_minmax_light_distance(P, s->pos, n, &min_d, &max_d);
      float spacing = 2.0f * P->min_radius;
      orc_updatePhase(P, s->pos, s->phase, spacing, max_d, min_d, n);
      if (P->phase_std) {
        if (s->xw) { /* impl.cuh:43-51 with the XORWOW generator */
          for (uint32_t i = 0; i < n; i++) {
            float noise = P->phase_std * xw_normal(&s->xw[i], P->rngKind);
            s->phase[i] += noise;
          }
        } else {
          orc_add_normal_noise(P->seed, s->phaseDraws, s->phase, P->phase_std, n);
        }
        s->phaseDraws++;
      }
    }
    if (s->time >= 0)
      orc_updateRad_light_wave(P, s->absA, s->absR, s->rad, s->phase, s->time, dt, s->dead, n);
  }

  orc_integrateSystem(P, s->pos, s->vel, s->rad, dt, n);

  if (s->forceSort || s->time - sort_interval * floorf(s->time / sort_interval) < dt) {
    s->forceSort = 0;
    orc_calcHash(P, s->hash, s->index, s->pos, n);
    orc_sortParticlebots(s->hash, s->index, n);
  }

  orc_reorderDataAndFindCellStart(P, s->cellStart, s->cellEnd, s->sPos, s->sVel, s->sRad, s->hash,
                                  s->index, s->pos, s->vel, s->rad, n, P->numCells);
  orc_collide(P, s->vel, s->absA, s->absR, s->sPos, s->sVel, s->sRad, s->index, s->cellStart, s->cellEnd,
              n, dt);
  s->time = s->time + dt;
  return 0;
}

int orc_sim_dump(OrcSim *s, FILE *fp, float dump_interval, uint32_t testing, int echo) {
  /* particlebot.cpp:303-367 with start = 0, count = nCells */
  const OrcParams *P = &s->P;
  const uint32_t count = s->n;
  float sumX = 0.0f, sumY = 0.0f;
  if (s->time - dump_interval * floorf(s->time / dump_interval) > 0.01f) return 0;
  if (fp) {
    if (s->time == 0) {
      fprintf(fp, "Seed, %u\n", P->seed);
      fprintf(fp, "Time,");
      if (testing) {
        for (uint32_t i = 0; i < count; i++) fprintf(fp, "Particlebot_%d_xpos, Particlebot_%d_ypos,", i, i);
        for (uint32_t i = 0; i < count; i++) fprintf(fp, "Particlebot_%d_xvel, Particlebot_%d_yvel,", i, i);
        for (uint32_t i = 0; i < count; i++) fprintf(fp, "Particlebot_%d_rad,", i);
      }
      fprintf(fp, "Centroid X, Centroid Y, Distance");
      fprintf(fp, "\n");
    }
    fprintf(fp, "%f,", s->time);
    if (testing) {
      for (uint32_t i = 0; i < count; i++) fprintf(fp, "%f, %f,", s->pos[i * 2 + 0], s->pos[i * 2 + 1]);
      for (uint32_t i = 0; i < count; i++) fprintf(fp, "%f, %f,", s->vel[i * 2 + 0], s->vel[i * 2 + 1]);
      for (uint32_t i = 0; i < count; i++) fprintf(fp, "%f,", s->rad[i]);
    }
  }
  for (uint32_t i = 0; i < count; i++) {
    sumX += s->pos[i * 2 + 0];
    sumY += s->pos[i * 2 + 1];
  }
  if (fp) {
    fprintf(fp, "%f, %f, %f,", sumX / (float)count, sumY / (float)count,
            powf(powf(sumX / (float)count - P->light_x, 2.0) + powf(sumY / (float)count - P->light_y, 2.0), 0.5));
    fprintf(fp, "\n");
  }
  if (echo) printf("%f %f %f \n", s->time, sumX / (float)count, sumY / (float)count);
  return 1;
}

int orc_sim_load_from_file(OrcSim *s, FILE *fp) {
  /* particlebot.cpp:369-411: seek to the last complete line of a testing=1 CSV */
  const uint32_t count = s->n;
  fseek(fp, 0, SEEK_SET);
  int c = fgetc(fp);
  long bytes = 1, line_start_1 = 0, line_start_2 = 0;
  while (c != EOF) {
    if (c == '\n') {
      line_start_1 = line_start_2;
      line_start_2 = bytes;
    }
    c = fgetc(fp);
    bytes += 1;
  }
  fseek(fp, line_start_1 - bytes, SEEK_END);
  if (fscanf(fp, "%f,", &s->time) != 1) return -1;
  for (uint32_t i = 0; i < count; i++)
    if (fscanf(fp, "%f, %f,", &s->pos[i * 2 + 0], &s->pos[i * 2 + 1]) != 2) return -1;
  for (uint32_t i = 0; i < count; i++)
    if (fscanf(fp, "%f, %f,", &s->vel[i * 2 + 0], &s->vel[i * 2 + 1]) != 2) return -1;
  for (uint32_t i = 0; i < count; i++)
    if (fscanf(fp, "%f,", &s->rad[i]) != 1) return -1;
  return 0;
}

/* NOT in the reference: makes the next update re-hash and sort regardless of the clock.  The
 * reference resumes from a CSV with all-zero hash/index arrays and computes garbage until the next
 * multiple of sort_interval (a latent bug behind its hard-coded `cont = 0`, main.cpp:886); the
 * product sorts on the first step instead, and tests use this switch to compare like with like. */
void orc_sim_force_sort_once(OrcSim *s) { s->forceSort = 1; }

float orc_sim_time(const OrcSim *s) { return s->time; }
void orc_sim_set_time(OrcSim *s, float t) { s->time = t; }
uint32_t orc_sim_phase_draws(const OrcSim *s) { return s->phaseDraws; }

void *orc_sim_array(OrcSim *s, int which) {
  switch (which) {
    case 0: return s->pos;
    case 1: return s->vel;
    case 2: return s->rad;
    case 3: return s->phase;
    case 4: return s->absA;
    case 5: return s->absR;
    case 6: return s->dead;
    case 7: return s->hash;
    case 8: return s->index;
    case 9: return s->xw;
    default: return NULL;
  }
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
