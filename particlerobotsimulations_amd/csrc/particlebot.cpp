// particlebot.cpp -- host side of class Particlebot (headless, MI355X).
//
// Mirrors the behaviour of the reference's particlebot.cpp (cited per method) without its OpenGL
// storage: host mirrors + either the resident fused engine (pbSim*) or, for Engine::Legacy, the
// reference's own kernel-by-kernel call sequence through the `extern "C"` device boundary.
//
// Host arithmetic that decides results (placement accept/reject, the min-distance square root,
// the CSV distance column, the dead-bot draw) uses libc rand()/powf/cosf/sinf exactly where the
// reference does; compile with -ffp-contract=off.
#include "particlebot.h"

#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>
#include <string>

namespace {

bool g_verbosePlacement = false;
constexpr float kPi = 3.141592654f;  // particlebot.cpp:21-23

// particlebot.cpp:27-30 frand(): rand() / (float)RAND_MAX
inline float frand(PbLibcRand &g) { return g.next() / (float)PbLibcRand::kMax; }
// particlebot.cpp:32-34: the host-side length() is powf-based, unlike the device one
inline float hostLength(float x, float y) { return powf(powf(x, 2.0f) + powf(y, 2.0f), 0.5f); }

inline bool everyGate(float t, float interval, float dt) { return t - interval * floorf(t / interval) < dt; }

void die(const char *what) {
  fprintf(stderr, "Particlebot: %s: %s\n", what, pbGetLastErrorString());
  exit(EXIT_FAILURE);
}

// Occupancy lists of the placement grid.  The reference keeps vector<vector<vector<int>>> indexed
// [x][y] (particlebot.cpp:614-623); the scans only ever ask "is any listed bot closer than 2 r_min",
// so any container of a cell's discs is equivalent.  Cells outside the grid (which the reference indexes
// out of bounds, :689-691) are treated as empty.
//
// On top of the lists, placeRandom keeps for every placed disc ("anchor") which directions of its first
// rings are certainly crowded (RimMask), updated when a disc is added: of i placed discs only ~sqrt(i) have
// room, and those mostly on one side, so ~99 % of the reference's draws fail -- and are now decided from
// the mask, without trigonometry or scan, and without changing a single decision.
struct PlacementGrid {
  uint gx, gy;
  float ox, oy, cx, cy;
  // One cache line per cell: positions and numbers of up to five discs filed under it (cells are 2 r_max wide
  // and discs 2 r_min apart: five is the most a cell of non-overlapping discs holds) and, for anything beyond
  // that, a chain of overflow nodes.  The scans walk 9 to 25 neighbouring cells per placed disc and per slow
  // draw: inline entries make those independent loads of adjacent lines instead of a pointer chase per disc.
  static constexpr int kInline = 5;
  struct Cell {
    float x[kInline], y[kInline];
    int id[kInline];  // | kNoCover: seen by the 3x3 scan only (see addTracked)
    int meta;         // bits 0-2: inline entries in use; bits 3...: 1 + index of the first overflow node, 0 = none
  };
  static constexpr int kNoCover = 1 << 30;
  static_assert(sizeof(Cell) == 64, "one line per cell");
  struct Node {
    float x, y;
    int id, next;  // next: 1 + index, 0 = end
  };
  Cell *cell;  // calloc'ed: pages of cells nothing is filed under are never touched
  std::vector<Node> spill;

  // ---- what is crowded around an anchor ----
  // A ring (the circle of candidate positions of radius `ring` around an anchor) is cut into 64 sectors, equal
  // steps of a pseudo-angle (monotone in the angle, no trigonometry).  A disc at distance D from the anchor
  // blocks the arc |angle - phi| < acos((ring^2 + D^2 - (limit - margin)^2) / (2 ring D)); bit k of the mask says
  // that sector k lies entirely inside one disc's arc.  Sufficient, never necessary: the margin (0.1 % of the
  // limit, and never less than a few float ulps of the coordinates involved -- the candidate is computed in float,
  // x = ax + 2 r cosf(theta) -- so that an arena thousands of units wide cannot make "crowded" optimistic) and the
  // 1e-6 added to the cosine shrink every arc, and discs only ever get added, so a set bit stays true.
  // Rings 1 and 2 always (the rejection counter reaches 200, and the ring widens, for every other disc of a large
  // blob); ring 3 when the blob is large enough for its draws (6 % of all at 10^5 discs) to outweigh the wider
  // neighbourhood every added disc then has to mark.
  static constexpr int kSectors = 64, kRings = 3;
  struct RimMask {
    unsigned long long m[kRings];
  };
  std::vector<RimMask> rim;  // per disc (placeRandom only)
  int rings = 0;             // rings tracked, <= kRings
  double ringOf[kRings] = {0, 0, 0}, limit = 0;

  PlacementGrid(const SimParams &p, uint)
      : gx(p.gridSize.x), gy(p.gridSize.y), ox(p.worldOrigin.x), oy(p.worldOrigin.y), cx(p.cellSize.x),
        cy(p.cellSize.y), cell((Cell *)calloc((size_t)p.gridSize.x * p.gridSize.y, sizeof(Cell))) {
    if (!cell) throw std::bad_alloc();
  }
  ~PlacementGrid() { free(cell); }
  PlacementGrid(const PlacementGrid &) = delete;
  PlacementGrid &operator=(const PlacementGrid &) = delete;
  int rawCol(double x) const { return (int)floor((x - ox) / cx); }
  int rawRow(double y) const { return (int)floor((y - oy) / cy); }
  int col(float x) const { return ((int)floorf((x - ox) / cx)) & (int)(gx - 1); }
  int row(float y) const { return ((int)floorf((y - oy) / cy)) & (int)(gy - 1); }
  // file disc `bot` under the cell of (x, y); (px, py) is where it really is (they differ only for the
  // reference's seed disc, particlebot.cpp:635-637)
  void add(int bot, float x, float y, float px, float py) {
    if (col(x) != col(px) || row(y) != row(py)) bot |= kNoCover;
    Cell &c = cell[(size_t)col(x) * gy + (size_t)row(y)];
    const int k = c.meta & 7;
    if (k < kInline) {
      c.x[k] = px, c.y[k] = py, c.id[k] = bot;
      c.meta++;
    } else {
      spill.push_back(Node{px, py, bot, c.meta >> 3});
      c.meta = (int)(spill.size() << 3) | k;
    }
  }
  void add(int bot, float x, float y) { add(bot, x, y, x, y); }
  // f(x, y, id) for every disc filed under cell (xg, yg) until it returns true; true if it did
  template <class F>
  bool anyIn(int xg, int yg, F f) const {
    const Cell &c = cell[(size_t)xg * gy + yg];
    for (int k = 0, m = c.meta & 7; k < m; k++)
      if (f(c.x[k], c.y[k], c.id[k])) return true;
    for (int o = c.meta >> 3; o; o = spill[o - 1].next)
      if (f(spill[o - 1].x, spill[o - 1].y, spill[o - 1].id)) return true;
    return false;
  }
  // any listed bot within `within` of (x,y) in the 3x3 cells around it?
  // The decision is the reference's `length(...) < within` with its three powf calls; a double
  // precision squared distance settles every pair that is not within 1e-5 (relative) of the limit
  // -- hostLength is good to a few 1e-7 -- so the powf form only runs for pairs that (nearly) touch.
  bool crowded(float x, float y, double within) const {
    const int xc = col(x), yc = row(y);
    const double far2 = within * within * (1.0 + 2e-5), near2 = within * within * (1.0 - 2e-5);
    auto close = [&](float bx, float by, int) {
      const float dx = x - bx, dy = y - by;
      const double d2 = (double)dx * dx + (double)dy * dy;
      if (d2 > far2) return false;
      return d2 < near2 || hostLength(dx, dy) < within;
    };
    // (the candidate's own cell first: that is where a blocker most often is; the answer is an "any")
    if (anyIn(xc, yc, close)) return true;
    for (int xg = xc - 1; xg <= xc + 1; xg++)
      for (int yg = yc - 1; yg <= yc + 1; yg++) {
        if (xg < 0 || yg < 0 || xg >= (int)gx || yg >= (int)gy || (xg == xc && yg == yc)) continue;
        if (anyIn(xg, yg, close)) return true;
      }
    return false;
  }

  static double pseudoAngle(double x, double y) {  // [0, 4), increasing with atan2(y, x) taken in [0, 2 pi)
    const double s = fabs(x) + fabs(y);
    if (y >= 0.0) return x >= 0.0 ? y / s : 1.0 - x / s;
    return x < 0.0 ? 2.0 - y / s : 3.0 + x / s;
  }
  // the bits of sectors from ... from + count - 1 (modulo kSectors; 0 < count <= kSectors)
  static unsigned long long sectors(int from, int count) {
    const unsigned long long run = count >= 64 ? ~0ull : (1ull << count) - 1;
    const int sh = from & 63;
    return sh ? run << sh | run >> (64 - sh) : run;
  }
  // the arc a disc in direction (ux, uy) (unit vector) at squared distance D2 (iD = 1 / D) blocks on the ring of
  // radius `ring`: first sector and number of sectors entirely inside it (0: none), -1: the whole ring
  static int arcSectors(double ux, double uy, double D2, double iD, double ring, double lim, int *from) {
    // cosine of the arc's half width, raised by 1e-6: the arc shrinks by at least 1e-6 rad at each end
    const double c = (ring * ring - lim * lim + D2) * (0.5 / ring) * iD + 1e-6;
    if (c >= 1.0) return 0;
    if (c <= -1.0 + 1e-9) return -1;
    const double sn = sqrt(1.0 - c * c);
    // the direction turned by -w and by +w, as pseudo-angles
    const double lo = pseudoAngle(ux * c + uy * sn, uy * c - ux * sn);
    double hi = pseudoAngle(ux * c - uy * sn, uy * c + ux * sn);
    if (hi <= lo) hi += 4.0;
    const int first = (int)ceil(lo * (kSectors / 4.0) + 1e-9), end = (int)floor(hi * (kSectors / 4.0) - 1e-9);
    *from = first;
    return end > first ? end - first : 0;
  }
  double marginAt(double x, double y) const {
    const double ulps = 8.0 * 1.1920929e-7 * (fabs(x) + fabs(y) + ringOf[rings - 1]);
    return std::max(1e-3 * limit, ulps);
  }
  // placeRandom: radius[k] = radius of ring k + 1, `touch` = the crowding limit, n discs to come
  void trackRims(uint n, int nrings, const double radius[kRings], double touch) {
    limit = touch;
    // the masks speak for the 3x3 scan only if that scan sees every disc within `touch` of a candidate
    if (cx < touch || cy < touch) return;
    rim.assign(n, RimMask{});
    rings = nrings;
    for (int k = 0; k < nrings; k++) ringOf[k] = radius[k];
  }
  // add() for a disc filed where it is, which also marks what it blocks on its neighbours' rings and what they
  // block on its own.  A disc filed under a cell other than its position's (the reference bins bot 0, at (5,0),
  // under the ORIGIN's cell, :635-637; a position outside the grid wraps) is only seen by the 3x3 crowded test of
  // candidates near the cell it is filed under, not near where it is: it neither marks nor is marked
  // (conservative: draws around it are simply tested the slow way).
  void addTracked(int bot, float x, float y) {
    if (rings == 0) return add(bot, x, y);
    const double mine = marginAt(x, y), reachMax = ringOf[rings - 1] + limit;
    const int x0 = rawCol(x - reachMax), x1 = rawCol(x + reachMax), y0 = rawRow(y - reachMax), y1 = rawRow(y + reachMax);
    // (a disc this close to the edge of the grid -- the reference's blobs never are -- could "block" candidates
    //  that lie outside it, whose 3x3 scan wraps to the other side: it is left out like a misfiled one)
    if (x0 < 0 || y0 < 0 || x1 >= (int)gx || y1 >= (int)gy || rawCol(x) != col(x) || rawRow(y) != row(y) ||
        mine > 0.25 * limit)
      return add(bot | kNoCover, x, y);
    add(bot, x, y);
    RimMask &me = rim[bot];
    for (int xg = x0; xg <= x1; xg++)
      for (int yg = y0; yg <= y1; yg++)
        anyIn(xg, yg, [&](float bx, float by, int id) {
          const double dx = (double)bx - x, dy = (double)by - y, D2 = dx * dx + dy * dy;
          if (id == bot || id >= kNoCover || D2 < 1e-18 || D2 >= reachMax * reachMax) return false;
          const double lim = limit - std::max(mine, marginAt(bx, by));
          if (lim < 0.75 * limit) return false;
          const double iD = 1.0 / sqrt(D2), ux = dx * iD, uy = dy * iD;  // from the new disc to the neighbour
          RimMask &other = rim[id];
          for (int k = 0; k < rings; k++) {
            const double reach = ringOf[k] + lim;
            if (D2 >= reach * reach) continue;
            int from = 0;
            const int cnt = arcSectors(ux, uy, D2, iD, ringOf[k], lim, &from);
            if (cnt < 0) {
              me.m[k] = other.m[k] = ~0ull;
            } else if (cnt > 0) {
              me.m[k] |= sectors(from, cnt);
              other.m[k] |= sectors(from + kSectors / 2, cnt);  // seen from the neighbour: the opposite direction
            }
          }
          return false;
        });
  }
  // the sectors a draw's angle may fall into: theta in 1/65536 of a turn, in steps of 16, two units of slack at
  // either end; {first sector, count}
  struct Span {
    unsigned char first, count;
  };
  static const Span *sectorsOfStep() {
    static Span table[4096];
    static const bool once = [] {
      for (int s = 0; s < 4096; s++) {
        const double a0 = (16.0 * s - 2.0) * (6.283185307179586 / 65536.0), a1 = (16.0 * s + 18.0) * (6.283185307179586 / 65536.0);
        const int k0 = (int)floor(pseudoAngle(cos(a0), sin(a0)) * (kSectors / 4.0) - 1e-9);
        const int k1 = (int)floor(pseudoAngle(cos(a1), sin(a1)) * (kSectors / 4.0) + 1e-9);
        const int cnt = ((k1 - k0) & (kSectors - 1)) + 1;
        table[s] = Span{(unsigned char)(k0 & (kSectors - 1)), (unsigned char)cnt};
      }
      return true;
    }();
    (void)once;
    return table;
  }
  // is a draw at angle theta (as placeRandom computes it: 2 frand pi, in [0, 2 pi + an ulp]) from anchor `bot` on
  // ring k + 1 certainly crowded?
  bool certainlyCrowded(int bot, int k, float theta) const {
    if (k >= rings) return false;
    const unsigned long long m = rim[bot].m[k];
    if (m == ~0ull) return true;  // buried: every direction is
    if (m == 0) return false;
    const int s = (int)(theta * (65536.0f / 6.2831855f) * (1.0f / 16.0f));
    if (s < 1 || s >= 4094) return false;  // (the ends of the turn are left to the slow test)
    const Span sp = sectorsOfStep()[s];
    const unsigned long long want = sectors(sp.first, sp.count);
    return (m & want) == want;
  }

  // Lazy form for the wider rings (rare): is EVERY point of the circle of radius `ring` around (ax, ay)
  // crowded?  One scan of the discs in reach, same arcs and margins.
  bool ringCovered(float ax, float ay, double ring, double touch) const {
    if (cx < touch || cy < touch) return false;
    const double ulps = 8.0 * 1.1920929e-7 * (fabs((double)ax) + fabs((double)ay) + ring);
    const double margin = std::max(1e-3 * touch, ulps);
    if (margin > 0.25 * touch) return false;
    const double lim = touch - margin, reach = ring + lim;
    const int x0 = rawCol(ax - reach), x1 = rawCol(ax + reach), y0 = rawRow(ay - reach), y1 = rawRow(ay + reach);
    if (x0 < 0 || y0 < 0 || x1 >= (int)gx || y1 >= (int)gy) return false;  // (candidates may leave the grid)
    unsigned long long m = 0;
    for (int xg = x0; xg <= x1; xg++)
      for (int yg = y0; yg <= y1; yg++)
        if (anyIn(xg, yg, [&](float bx, float by, int id) {
              const double dx = (double)bx - ax, dy = (double)by - ay, D2 = dx * dx + dy * dy;
              if (D2 < 1e-18 || D2 >= reach * reach) return false;  // the centre itself / too far to reach the ring
              if (id >= kNoCover || col(bx) != xg || row(by) != yg) return false;  // no cover (see addTracked)
              const double iD = 1.0 / sqrt(D2);
              int from = 0;
              const int cnt = arcSectors(dx * iD, dy * iD, D2, iD, ring, lim, &from);
              if (cnt < 0) return true;
              if (cnt > 0) m |= sectors(from, cnt);
              return false;
            }))
          return true;
    return m == ~0ull;
  }
};

}  // namespace

// glibc stdlib/random_r.c, TYPE_3 (x^31 + x^3 + 1): srandom_r fills r[0..30] with the Lehmer
// sequence 16807*x mod (2^31-1), then discards 310 outputs; random_r returns (r[f] += r[b]) >> 1.
void PbLibcRand::reseed(unsigned seed) {
  if (seed == 0) seed = 1;
  int *state = r + 3;
  state[0] = (int)seed;
  long word = (int)seed;
  for (int i = 1; i < 31; i++) {
    const long hi = word / 127773, lo = word % 127773;
    word = 16807 * lo - 2836 * hi;
    if (word < 0) word += 2147483647;
    state[i] = (int)word;
  }
  f = 3;
  b = 0;
  for (int i = 0; i < 310; i++) (void)next();
}

int PbLibcRand::next() {
  int *state = r + 3;
  const unsigned v = (unsigned)state[f] + (unsigned)state[b];
  state[f] = (int)v;
  const int out = (int)(v >> 1);
  if (++f >= 31) f = 0;
  if (++b >= 31) b = 0;
  return out;
}

void PbLibcRand::getState(int out[36]) const {
  for (int i = 0; i < 34; i++) out[i] = r[i];
  out[34] = f;
  out[35] = b;
}

void PbLibcRand::setState(const int in[36]) {
  for (int i = 0; i < 34; i++) r[i] = in[i];
  f = in[34];
  b = in[35];
}

void Particlebot::setVerbosePlacement(bool on) { g_verbosePlacement = on; }

static Particlebot::Engine engineFromEnv() {
  if (const char *e = getenv("PB_ENGINE"))
    if (!strcmp(e, "legacy")) return Particlebot::Engine::Legacy;
  return Particlebot::Engine::Fused;
}

Particlebot::Particlebot(SimParams simparams) : Particlebot(simparams, engineFromEnv(), 64.0f) {}

Particlebot::Particlebot(SimParams simparams, Engine engine, float wall) : time(0), rng(simparams.seed) {
  params = simparams;
  engineKind = engine;
  wallHalf = wall > 0.0f ? wall : 64.0f;
  particlebotConfigSize.x = particlebotConfigSize.y = 0;
  // The reference shallow-copies the obstacle pointers (particlebot.cpp:43); own them instead.
  const int nr = params.nobstacles > 0 ? params.nobstacles : 0, nc = params.n_cir_obstacles > 0 ? params.n_cir_obstacles : 0;
  obsStore.assign((size_t)4 * nr + (size_t)3 * nc + 1, 0.0f);
  float *w = obsStore.data();
  auto own = [&](float *&field, int cnt) {
    for (int i = 0; i < cnt; i++) w[i] = field ? field[i] : 0.0f;
    field = w;
    w += cnt;
  };
  own(params.x1obs, nr);
  own(params.x2obs, nr);
  own(params.y1obs, nr);
  own(params.y2obs, nr);
  own(params.x_cir_obs, nc);
  own(params.y_cir_obs, nc);
  own(params.r_cir_obs, nc);
  _initialize();
}

Particlebot::~Particlebot() { _finalize(); }

void Particlebot::_initialize() {
  // particlebot.cpp:77-166 minus GL.  Host mirrors are zeroed; device state starts zeroed too.
  const size_t n = params.nCells;
  hPosV.assign(2 * n, 0.0f);
  hVelV.assign(2 * n, 0.0f);
  hRadV.assign(n, 0.0f);
  hPhaseV.assign(n, 0.0f);
  hFreqV.assign(n, 0.0f);
  hDeadV.assign(n, 0);
  hPos = hPosV.data();
  hVel = hVelV.data();
  hRad = hRadV.data();
  hphase = hPhaseV.data();
  hDead = hDeadV.data();

  if (engineKind == Engine::HostOnly) return;
  if (engineKind == Engine::Fused) {
    if (pbSimCreate(&sim, &params, wallHalf) != PB_OK) die("pbSimCreate");
    return;
  }
  // Engine::Legacy: the reference's buffers (particlebot.cpp:101-165)
  cudaInit(0, nullptr);
  const size_t memSize = sizeof(float) * 2 * n;
  posVbo = pbCreateBuffer(memSize);
  registerGLBufferObject(posVbo, &posRes);
  radVbo = pbCreateBuffer(sizeof(float) * n);
  registerGLBufferObject(radVbo, &radRes);
  cudaPosVBO = (float *)mapGLBufferObject(&posRes);
  cudaRadVBO = (float *)mapGLBufferObject(&radRes);
  allocateArray((void **)&dVel, memSize);
  allocateArray((void **)&dSortedPos, memSize);
  allocateArray((void **)&dSortedVel, memSize);
  allocateArray((void **)&dSortedRad, sizeof(float) * n);
  allocateArray((void **)&dphase, sizeof(float) * n);
  allocateArray((void **)&dAbsForce_a, sizeof(float) * n);
  allocateArray((void **)&dAbsForce_r, sizeof(float) * n);
  allocateArray((void **)&dGridParticleHash, n * sizeof(uint));
  allocateArray((void **)&dGridParticleIndex, n * sizeof(uint));
  allocateArray((void **)&dCellStart, params.numCells * sizeof(uint));
  allocateArray((void **)&dCellEnd, params.numCells * sizeof(uint));
  allocateArray((void **)&dDead, sizeof(int) * n);
  allocateArray((void **)&dState, sizeof(pbRngState) * n);
  // zero what the reference leaves uninitialised
  std::vector<float> zeros(2 * n, 0.0f);
  copyArrayToDevice(dVel, zeros.data(), 0, (int)memSize);
  copyArrayToDevice(dAbsForce_a, zeros.data(), 0, (int)(sizeof(float) * n));
  copyArrayToDevice(dAbsForce_r, zeros.data(), 0, (int)(sizeof(float) * n));
  copyArrayToDevice(dphase, zeros.data(), 0, (int)(sizeof(float) * n));
  copyArrayToDevice(dDead, hDead, 0, (int)(sizeof(int) * n));
  copyArrayToDevice(dGridParticleHash, zeros.data(), 0, (int)(sizeof(uint) * n));
  copyArrayToDevice(dGridParticleIndex, zeros.data(), 0, (int)(sizeof(uint) * n));
  pbSetWallHalfExtent(wallHalf);
  setParameters(&params);
  curand_setup(dState, (int)n);
}

void Particlebot::setForceVariant(int variant) {
  if (variant < 0) return;  // (-1: the engine's default)
  if (variant > 3) die("setForceVariant: 0..3");
  if (engineKind == Engine::Fused && pbSimSetForceVariant(sim, variant) != PB_OK) die("pbSimSetForceVariant");
}

void Particlebot::setRng(int kind) {
  if (kind != PB_RNG_COUNTER && kind != PB_RNG_XORWOW_CURAND && kind != PB_RNG_XORWOW_ROCRAND) die("setRng: bad kind");
  rngKindV = kind;
  if (engineKind == Engine::Fused) {
    if (pbSimSetRng(sim, kind) != PB_OK) die("pbSimSetRng");
  } else if (engineKind == Engine::Legacy) {
    pbSetRngKind(kind);
    curand_setup(dState, (int)params.nCells);
  }
}

void Particlebot::_finalize() {
  if (sim) {
    pbSimDestroy(sim);
    sim = nullptr;
  }
  if (engineKind == Engine::Legacy && dVel) {
    freeArray(dVel);
    freeArray(dSortedPos);
    freeArray(dSortedVel);
    freeArray(dSortedRad);
    freeArray(dphase);
    freeArray(dAbsForce_a);
    freeArray(dAbsForce_r);
    freeArray(dGridParticleHash);
    freeArray(dGridParticleIndex);
    freeArray(dCellStart);
    freeArray(dCellEnd);
    freeArray(dDead);
    freeArray(dState);
    unregisterGLBufferObject(posRes);
    unregisterGLBufferObject(radRes);
    pbDeleteBuffer(posVbo);
    pbDeleteBuffer(radVbo);
    dVel = nullptr;
  }
}

// ---- stepping ---------------------------------------------------------------------------------

void Particlebot::drawDeadBots() {
  // particlebot.cpp:178-194: nDead distinct bots, `i = rand() % inds.size(); dead[inds[i]] = 1;
  // inds.erase(inds.begin() + i)` over inds = 0 .. nCells-1.  The list stays in ascending order, so
  // inds[i] is the i-th bot still alive: the same bots in the same order come out of a Fenwick tree of
  // alive flags ("position of the (i+1)-th one", O(log N)) without the erase's O(N) shuffle -- 0.1 s per
  // member at 10^5 bots, times every member of a dead-fraction sweep, all on one host thread.
  const uint n = params.nCells;
  uint top = 1;
  while (top * 2 <= n) top *= 2;
  std::vector<uint> tree(n + 1, 0u);
  for (uint i = 1; i <= n; i++) {  // all alive: node i covers i & -i elements
    tree[i] += 1u;
    const uint up = i + (i & (0u - i));
    if (up <= n) tree[up] += tree[i];
  }
  uint remaining = n;
  int count = 0;
  while (count < params.nDead && remaining > 0) {
    uint k = (uint)(rng.next() % remaining) + 1u;  // the k-th alive bot, 1-based
    uint pos = 0;
    for (uint step = top; step > 0; step >>= 1) {
      const uint nxt = pos + step;
      if (nxt <= n && tree[nxt] < k) {
        pos = nxt;
        k -= tree[nxt];
      }
    }
    // pos alive bots precede the one we want: it is bot `pos` (0-based)
    hDead[pos] = 1;
    for (uint i = pos + 1; i <= n; i += i & (0u - i)) tree[i] -= 1u;
    remaining--;
    count++;
  }
  if (engineKind == Engine::Fused) {
    if (pbSimSetState(sim, nullptr, nullptr, nullptr, nullptr, hDead) != PB_OK) die("pbSimSetState(dead)");
  } else if (engineKind == Engine::Legacy) {
    copyArrayToDevice(dDead, hDead, 0, (int)(params.nCells * sizeof(int)));
  }
}

void Particlebot::legacyUpdate(float deltaTime, float sort_interval) {
  // particlebot.cpp:196-299, call for call (minus calcCOG/updateCol, which only feed the renderer)
  float *dPos = (float *)mapGLBufferObject(&posRes);
  float *dRad = (float *)mapGLBufferObject(&radRes);
  unmapGLBufferObject(posRes);
  unmapGLBufferObject(radRes);
  const uint n = params.nCells;
  if (params.control == LIGHT_WAVE) {
    if (everyGate(time, params.phase_update_interval, deltaTime)) {
      copyArrayFromDevice(hPos, dPos, 0, (int)(sizeof(float) * 2 * n));
      float min_d = 0, max_d = 0, dist = 0;
      for (uint i = 0; i < n; i++) {
        dist = powf(powf(params.light_x - hPos[i * 2], 2) + powf(params.light_y - hPos[i * 2 + 1], 2), 0.5f);
        if (i == 0) {
          max_d = dist;
          min_d = dist;
        } else {
          min_d = (min_d < dist ? min_d : dist);
          max_d = (max_d > dist ? max_d : dist);
        }
      }
      const float spacing = 2.0f * params.min_radius;
      updatePhase(dPos, dphase, spacing, max_d, min_d, (int)n);
      if (params.phase_std) add_normal_noise(dState, dphase, params.phase_std, (int)n);
    }
    if (time >= 0) updateRad_light_wave(dPos, dAbsForce_a, dAbsForce_r, dRad, dphase, time, deltaTime, dDead, (int)n);
  }
  integrateSystem(dPos, dVel, dRad, deltaTime, n, time);
  if (everyGate(time, sort_interval, deltaTime)) {
    calcHash(dGridParticleHash, dGridParticleIndex, dPos, (int)n);
    sortParticlebots(dGridParticleHash, dGridParticleIndex, n);
  }
  reorderDataAndFindCellStart(dCellStart, dCellEnd, dSortedPos, dSortedVel, dSortedRad, dGridParticleHash,
                              dGridParticleIndex, dPos, dVel, dRad, n, params.numCells);
  collide(dVel, dAbsForce_a, dAbsForce_r, dSortedPos, dSortedVel, dSortedRad, dGridParticleIndex, dCellStart,
          dCellEnd, n, params.numCells, deltaTime);
  time = time + deltaTime;
}

void Particlebot::update(float deltaTime, float sort_interval) {
  if (time > params.max_time) {
    if (exitOnMaxTime) exit(0);  // particlebot.cpp:174-176
    return;
  }
  advance(deltaTime, sort_interval, 1);
}

int Particlebot::advance(float deltaTime, float sort_interval, int nsteps) {
  if (engineKind == Engine::HostOnly) {
    fprintf(stderr, "Particlebot: a HostOnly instance cannot step; its state lives in an ensemble batch\n");
    exit(EXIT_FAILURE);
  }
  int total = 0;
  const bool draws = params.nDead > 0;
  auto deadGate = [&](float t) { return t >= params.time_to_dead && t < params.time_to_dead + deltaTime; };
  while (total < nsteps) {
    if (time > params.max_time) break;
    if (draws && deadGate(time)) drawDeadBots();
    int run = nsteps - total;
    if (draws) {
      // stop the batch in front of the next step whose start time opens the dead-bot window
      run = 1;
      float t = time + deltaTime;
      while (total + run < nsteps && !deadGate(t)) {
        t = t + deltaTime;
        run++;
      }
    }
    if (engineKind == Engine::Fused) {
      int done = 0;
      if (pbSimStep(sim, deltaTime, sort_interval, run, &done) != PB_OK) die("pbSimStep");
      if (pbSimGetTime(sim, &time) != PB_OK) die("pbSimGetTime");
      total += done;
      if (done < run) break;
    } else {
      int done = 0;
      for (; done < run && !(time > params.max_time); done++) legacyUpdate(deltaTime, sort_interval);
      total += done;
      if (done < run) break;
    }
  }
  return total;
}

void Particlebot::setTime(float t) {
  time = t;
  if (sim && pbSimSetTime(sim, t) != PB_OK) die("pbSimSetTime");
}

bool Particlebot::dumpDue(float dump_interval) const {
  return !(time - dump_interval * floorf(time / dump_interval) > 0.01f);  // particlebot.cpp:309
}

int Particlebot::stepsUntilHostEvent(float deltaTime, float dump_interval, int maxSteps) const {
  // replay the fp32 clock to find the next step whose start time makes a dump row due
  float t = time + deltaTime;
  int k = 1;
  while (k < maxSteps && (t - dump_interval * floorf(t / dump_interval) > 0.01f) && !(t > params.max_time)) {
    t = t + deltaTime;
    k++;
  }
  return k;
}

// ---- state access -----------------------------------------------------------------------------

void Particlebot::pullState(bool pos, bool vel, bool rad) {
  const uint n = params.nCells;
  if (engineKind == Engine::HostOnly) return;
  if (engineKind == Engine::Fused) {
    if (pbSimGetState(sim, pos ? hPos : nullptr, vel ? hVel : nullptr, rad ? hRad : nullptr, nullptr, nullptr,
                      nullptr, nullptr) != PB_OK)
      die("pbSimGetState");
  } else {
    if (pos) copyArrayFromDevice(hPos, 0, &posRes, (int)(sizeof(float) * 2 * n));
    if (vel) copyArrayFromDevice(hVel, dVel, 0, (int)(sizeof(float) * 2 * n));
    if (rad) copyArrayFromDevice(hRad, 0, &radRes, (int)(sizeof(float) * n));
  }
}

float *Particlebot::getArray(ParticlebotArray array) {
  const uint n = params.nCells;
  switch (array) {
    default:
    case POSITION: pullState(true, false, false); return hPos;
    case VELOCITY: pullState(false, true, false); return hVel;
    case RADII: pullState(false, false, true); return hRad;
    case PHASE:
      if (engineKind == Engine::HostOnly) return hphase;
      if (engineKind == Engine::Fused) {
        if (pbSimGetState(sim, nullptr, nullptr, nullptr, hphase, nullptr, nullptr, nullptr) != PB_OK)
          die("pbSimGetState");
      } else {
        copyArrayFromDevice(hphase, dphase, 0, (int)(sizeof(float) * n));
      }
      return hphase;
  }
}

int *Particlebot::getDeadArray() {
  if (engineKind == Engine::HostOnly) return hDead;
  if (engineKind == Engine::Fused) {
    if (pbSimGetState(sim, nullptr, nullptr, nullptr, nullptr, hDead, nullptr, nullptr) != PB_OK) die("pbSimGetState");
  } else {
    copyArrayFromDevice(hDead, dDead, 0, (int)(sizeof(int) * params.nCells));
  }
  return hDead;
}

void Particlebot::setArray(ParticlebotArray array, const float *data, int start, int count) {
  // particlebot.cpp:834-867.  The reference's POSITION/RADII buffers carry centroid_steps+1 extra
  // display entries; only the nCells bots exist here, so longer ranges are clipped.
  const int n = (int)params.nCells;
  if (start < 0 || start >= n || count <= 0) return;
  if (start + count > n) count = n - start;
  const bool fused = engineKind == Engine::Fused;
  switch (array) {
    default:
    case POSITION:
      if (data != hPos + 2 * start) memcpy(hPos + 2 * start, data, sizeof(float) * 2 * count);
      if (fused) {
        if (pbSimSetStateRangeOf(sim, 0, (unsigned)start, (unsigned)count, data, nullptr, nullptr, nullptr, nullptr) != PB_OK)
          die("pbSimSetStateRangeOf");
      } else {
        pbBufferSubData(posVbo, sizeof(float) * 2 * start, sizeof(float) * 2 * count, data);
      }
      break;
    case VELOCITY:
      if (data != hVel + 2 * start) memcpy(hVel + 2 * start, data, sizeof(float) * 2 * count);
      if (fused) {
        if (pbSimSetStateRangeOf(sim, 0, (unsigned)start, (unsigned)count, nullptr, data, nullptr, nullptr, nullptr) != PB_OK)
          die("pbSimSetStateRangeOf");
      } else {
        copyArrayToDevice(dVel, data, (int)(start * 2 * sizeof(float)), (int)(count * 2 * sizeof(float)));
      }
      break;
    case PHASE:
      if (data != hphase + start) memcpy(hphase + start, data, sizeof(float) * count);
      if (fused) {
        if (pbSimSetStateRangeOf(sim, 0, (unsigned)start, (unsigned)count, nullptr, nullptr, nullptr, data, nullptr) != PB_OK)
          die("pbSimSetStateRangeOf");
      } else {
        copyArrayToDevice(dphase, data, (int)(start * sizeof(float)), (int)(count * sizeof(float)));
      }
      break;
    case FREQUENCY:  // stored, never read by any kernel (as in the reference)
      memcpy(hFreqV.data() + start, data, sizeof(float) * count);
      break;
    case RADII:
      if (data != hRad + start) memcpy(hRad + start, data, sizeof(float) * count);
      if (fused) {
        if (pbSimSetStateRangeOf(sim, 0, (unsigned)start, (unsigned)count, nullptr, nullptr, data, nullptr, nullptr) != PB_OK)
          die("pbSimSetStateRangeOf");
      } else {
        pbBufferSubData(radVbo, sizeof(float) * start, sizeof(float) * count, data);
      }
      break;
  }
}

// ---- CSV dump / reload ------------------------------------------------------------------------

void Particlebot::dumpParticlebot(uint start, uint count, FILE *fp, float dump_interval, uint testing,
                                  float light_x, float light_y) {
  // particlebot.cpp:303-367, same text byte for byte
  float sumX = 0.0f;
  float sumY = 0.0f;
  if (time - dump_interval * floorf(time / dump_interval) > 0.01f) return;
  pullState(true, true, true);
  if (time == 0) {
    fprintf(fp, "Seed, %u\n", params.seed);
    fprintf(fp, "Time,");
    if (testing) {
      for (uint i = start; i < start + count; i++) fprintf(fp, "Particlebot_%d_xpos, Particlebot_%d_ypos,", i, i);
      for (uint i = start; i < start + count; i++) fprintf(fp, "Particlebot_%d_xvel, Particlebot_%d_yvel,", i, i);
      for (uint i = start; i < start + count; i++) fprintf(fp, "Particlebot_%d_rad,", i);
    }
    fprintf(fp, "Centroid X, Centroid Y, Distance");
    fprintf(fp, "\n");
  }
  fprintf(fp, "%f,", time);
  if (testing) {
    for (uint i = start; i < start + count; i++) fprintf(fp, "%f, %f,", hPos[i * 2 + 0], hPos[i * 2 + 1]);
    for (uint i = start; i < start + count; i++) fprintf(fp, "%f, %f,", hVel[i * 2 + 0], hVel[i * 2 + 1]);
    for (uint i = start; i < start + count; i++) fprintf(fp, "%f,", hRad[i]);
  }
  for (uint i = start; i < start + count; i++) {
    sumX += hPos[i * 2 + 0];
    sumY += hPos[i * 2 + 1];
  }
  fprintf(fp, "%f, %f, %f,", sumX / (float)count, sumY / (float)count,
          powf(powf(sumX / (float)count - light_x, 2.0) + powf(sumY / (float)count - light_y, 2.0), 0.5));
  fprintf(fp, "\n");
  printf("%f %f %f \n", time, sumX / (float)count, sumY / (float)count);
}

void Particlebot::loadFromFile(uint start, uint count, FILE *fp, float) {
  // particlebot.cpp:369-411: last complete row of a testing=1 CSV -> time, pos, vel, rad
  fseek(fp, 0, SEEK_SET);
  int c = fgetc(fp);
  long bytes = 1, prevLine = 0, lastLine = 0;
  while (c != EOF) {
    if (c == '\n') {
      prevLine = lastLine;
      lastLine = bytes;
    }
    c = fgetc(fp);
    bytes += 1;
  }
  fseek(fp, prevLine - bytes, SEEK_END);
  float t = 0;
  if (fscanf(fp, "%f,", &t) != 1) return;
  for (uint i = start; i < start + count; i++)
    if (fscanf(fp, "%f, %f,", &hPos[i * 2 + 0], &hPos[i * 2 + 1]) != 2) return;
  for (uint i = start; i < start + count; i++)
    if (fscanf(fp, "%f, %f,", &hVel[i * 2 + 0], &hVel[i * 2 + 1]) != 2) return;
  for (uint i = start; i < start + count; i++)
    if (fscanf(fp, "%f,", &hRad[i]) != 1) return;
  setTime(t);
  setArray(RADII, hRad, 0, params.nCells);
  setArray(POSITION, hPos, 0, params.nCells);
  setArray(VELOCITY, hVel, 0, params.nCells);
  printf("Time = %f\n", time);
}

// ---- initial placement --------------------------------------------------------------------------

void Particlebot::initGrid(uint2 size, float spacing, float jitter, uint nCells) {
  // particlebot.cpp:413-436 (y is forced to 0 there: a line of bots)
  const float xs = size.x * spacing / 2.0f;
  for (uint y = 0; y < size.y; y++)
    for (uint x = 0; x < size.x; x++) {
      const uint i = (y * size.x) + x;
      if (i < nCells) {
        hPos[i * 2] = (spacing * x) + params.min_radius - xs + (frand(rng) * 2.0f - 1.0f) * jitter;
        hPos[i * 2 + 1] = 0;
        hVel[i * 2] = 0.0f;
        hVel[i * 2 + 1] = 0.0f;
      }
    }
}

void Particlebot::initHexGrid(uint nCells, float spacing) {
  // particlebot.cpp:438-481: concentric hexagonal rings around the origin
  const float h = powf(3, 0.5f) * 0.5f;
  const float ux[7] = {1.0f, 0.5f, -0.5f, -1.0f, -0.5f, 0.5f, 1.0f};
  const float uy[7] = {0.0f, h, h, 0.0f, -h, -h, 0.0f};
  if (nCells == 0) return;
  uint i = 0;
  hPos[0] = 0.0f;
  hPos[1] = 0.0f;
  hVel[0] = hVel[1] = 0.0f;
  i++;
  int ring = 1;
  while (i < nCells) {
    for (int k = 0; k < 6 && i < nCells; k++)
      for (int j = 0; j < ring && i < nCells; j++) {
        hPos[i * 2] = ux[k] * (ring - j) * spacing + ux[k + 1] * spacing * j;
        hPos[i * 2 + 1] = uy[k] * (ring - j) * spacing + uy[k + 1] * spacing * j;
        hVel[i * 2] = 0.0f;
        hVel[i * 2 + 1] = 0.0f;
        i++;
      }
    ring++;
  }
  particlebotConfigSize.x = particlebotConfigSize.y = ring * 2;
}

void Particlebot::placeRandom() {
  // particlebot.cpp:612-748.  Seed bot at (5,0); bot 2 perpendicular to the first pair; every
  // other bot: random anchor, random direction, reject if crowded, then pivot in 10-degree steps
  // until the next position would touch something.  After 200 rejections the ring widens.
  const uint n = params.nCells;
  if (n == 0) return;
  PlacementGrid grid(params, n);
  particlebotConfigSize.x = (int)ceilf(powf((float)n, 1.0f / 2.0f));
  const double touch = 2 * 1.0 * params.min_radius;
  const float pivot = 2 * kPi / 360.0 * 10.0;
  const uint maxRejections = 200;
  uint rejections = 0;
  float lowestX = 9999999.0;
  hPos[0] = 5.0;
  hPos[1] = 0.0;
  grid.add(0, 0.0f, 0.0f, hPos[0], hPos[1]);  // sic: the reference bins bot 0 at the origin's cell (:635-637)
  float x = 0, y = 0;
  // What is known per anchor (discs only get added, so it stays true):
  //   rings 1, 2 (3): PlacementGrid::RimMask, kept up to date by addTracked -- a draw into a crowded sector fails
  //                without trigonometry or scan
  //   rings ...-8: buried bit k-1 = the whole ring of radius 2 k min_radius is crowded (PlacementGrid::ringCovered,
  //                run after `retest` failed draws that had to be tested the slow way, then after 2, 4, ... times
  //                as many).  (The ring widens every 200 rejections; rings beyond the second are rare.)
  const int kRings = n >= 40000 ? 3 : 2;
  {
    double radius[PlacementGrid::kRings];
    float rr = params.min_radius;  // as the loop below accumulates it
    for (int k = 0; k < kRings; k++, rr += params.min_radius) radius[k] = 2.0 * (double)rr;
    grid.trackRims(n, kRings, radius, touch);
  }
  const int tracked = grid.rings;  // (0: a grid finer than the discs, nothing is tracked)
  std::vector<unsigned char> buried(n, 0), fails(n, 0), retest(n, 4);
  // PB_PLACEMENT_SELFCHECK=1: every draw decided by a mask is also tested the reference's way (tests/test_host_placement.py)
  const char *const selfCheckEnv = getenv("PB_PLACEMENT_SELFCHECK");
  const bool selfCheck = selfCheckEnv && selfCheckEnv[0] == '1';
  unsigned long long checked = 0;
  for (uint i = 1; i < n; i++) {
    if (g_verbosePlacement) printf("Placing %d th disc\n", i);
    if (i == 2) {
      const int side = rng.next() % 2;
      float dx = hPos[2] - hPos[0], dy = hPos[3] - hPos[1];
      const float l = hostLength(dx, dy);
      dy = dy / l;
      dx = dx / l;
      const float px = side ? dy : -dy, py = side ? -dx : dx;
      x = (hPos[2] + hPos[0]) / 2.0f + px * params.min_radius;
      y = (hPos[3] + hPos[1]) / 2.0f + py * params.min_radius;
      if (x < lowestX) lowestX = x;
      hPos[2 * i] = x;
      hPos[2 * i + 1] = y;
      grid.addTracked((int)i, x, y);
      continue;
    }
    float r = params.min_radius;
    int level = 1;  // r == level * min_radius (as accumulated)
    for (;;) {
      const uint anchor = (uint)rng.next() % i;
      if (rejections == maxRejections) {
        rejections = 0;
        r += params.min_radius;
        level++;
      }
      float theta = 2 * frand(rng) * kPi;
      if (level <= tracked ? grid.certainlyCrowded((int)anchor, level - 1, theta)
                          : (level <= 8 && (buried[anchor] >> (level - 1) & 1))) {
        // the draw of the angle is consumed, the outcome known
        if (selfCheck) {
          checked++;
          if (!grid.crowded(hPos[2 * anchor] + 2 * r * cosf(theta), hPos[2 * anchor + 1] + 2 * r * sinf(theta), touch)) {
            fprintf(stderr, "placeRandom self-check: disc %u, anchor %u, ring %d, theta %.9g is NOT crowded\n", i, anchor,
                    level, (double)theta);
            abort();
          }
        }
        rejections++;
        continue;
      }
      x = hPos[2 * anchor] + 2 * r * cosf(theta);
      y = hPos[2 * anchor + 1] + 2 * r * sinf(theta);
      if (grid.crowded(x, y, touch)) {
        rejections++;
        if (level > tracked && level <= 8 && ++fails[anchor] >= retest[anchor]) {
          fails[anchor] = 0;
          if (grid.ringCovered(hPos[2 * anchor], hPos[2 * anchor + 1], 2.0 * (double)r, touch))
            buried[anchor] |= (unsigned char)(1u << (level - 1));
          else if (retest[anchor] < 128)
            retest[anchor] = (unsigned char)(retest[anchor] * 2);
        }
        continue;
      }
      const float theta0 = theta;
      while (theta - theta0 < 2 * kPi) {
        theta += pivot;
        x = hPos[2 * anchor] + 2 * r * cosf(theta);
        y = hPos[2 * anchor + 1] + 2 * r * sinf(theta);
        if (grid.crowded(x, y, touch)) {
          theta -= pivot;
          break;
        }
      }
      x = hPos[2 * anchor] + 2 * r * cosf(theta);
      y = hPos[2 * anchor + 1] + 2 * r * sinf(theta);
      break;
    }
    if (x < lowestX) lowestX = x;
    if (params.nDead == -1 && i == n - 1) {  // the payload sits left of the blob (:731-735)
      x = lowestX - 1 * params.min_radius * params.radFactor - 2 * params.min_radius;
      y = 0;
    }
    hPos[2 * i] = x;
    hPos[2 * i + 1] = y;
    grid.addTracked((int)i, x, y);
  }
  if (selfCheck) fprintf(stderr, "placeRandom self-check: %llu mask decisions verified\n", checked);
}

void Particlebot::placeFastBlob() {
  // Extension (`pb_placement fastblob`): the same growth process as placeRandom -- a new disc is hung on
  // a random already-placed anchor at a random angle, rejected if crowded, then pivoted in 10-degree
  // steps until it is about to touch something (particlebot.cpp:676-726) -- in O(N) instead of
  // O(N^1.5).  The reference draws the anchor uniformly from ALL placed discs, and in a blob of i discs
  // only the ~sqrt(i) on its rim have room, so nearly every draw is wasted on an interior disc.  Here
  // anchors come from a list of discs that may still have room (a disc leaves it after kMaxFails
  // crowded attempts in a row), so the accepted anchors are distributed as the reference's accepted
  // anchors are; the wasted interior draws are only COUNTED, because the reference's rejection counter
  // widens the ring (r += min_radius every 200 rejections, :626-629) and that shapes large blobs.
  // Different random stream => a different blob of the same kind, not the reference's blob: the
  // default stays placeRandom (parity); tests hold this one to placeRandom's statistics.
  const uint n = params.nCells;
  if (n == 0) return;
  PlacementGrid grid(params, n);
  particlebotConfigSize.x = (int)ceilf(powf((float)n, 1.0f / 2.0f));
  const double touch = 2 * 1.0 * params.min_radius;
  const float pivot = 2 * kPi / 360.0 * 10.0;
  const int kMaxFails = 16;
  double rejections = 0.0;  // the reference's cumulative counter, interior draws included (expected value)
  float lowestX = 9999999.0;
  hPos[0] = 5.0;
  hPos[1] = 0.0;
  grid.add(0, hPos[0], hPos[1]);
  std::vector<uint> open(1, 0u);
  std::vector<unsigned char> fails(n, 0);
  float x = 0, y = 0;
  for (uint i = 1; i < n; i++) {
    float r = params.min_radius;
    for (;;) {
      if (open.empty()) {  // every disc looked full: widen the ring and look at all of them again
        r += params.min_radius;
        for (uint k = 0; k < i; k++) open.push_back(k);
        std::fill(fails.begin(), fails.begin() + i, 0);
      }
      const size_t slot = (size_t)((uint)rng.next() % (uint)open.size());
      const uint anchor = open[slot];
      rejections += (double)i / (double)open.size() - 1.0;
      if (rejections >= 200.0) {
        rejections -= 200.0;
        r += params.min_radius;
      }
      float theta = 2 * frand(rng) * kPi;
      x = hPos[2 * anchor] + 2 * r * cosf(theta);
      y = hPos[2 * anchor + 1] + 2 * r * sinf(theta);
      if (grid.crowded(x, y, touch)) {
        rejections += 1.0;
        if (r == params.min_radius && ++fails[anchor] >= kMaxFails) {
          open[slot] = open.back();
          open.pop_back();
        }
        continue;
      }
      fails[anchor] = 0;
      const float theta0 = theta;
      while (theta - theta0 < 2 * kPi) {
        theta += pivot;
        x = hPos[2 * anchor] + 2 * r * cosf(theta);
        y = hPos[2 * anchor + 1] + 2 * r * sinf(theta);
        if (grid.crowded(x, y, touch)) {
          theta -= pivot;
          break;
        }
      }
      x = hPos[2 * anchor] + 2 * r * cosf(theta);
      y = hPos[2 * anchor + 1] + 2 * r * sinf(theta);
      break;
    }
    if (x < lowestX) lowestX = x;
    if (params.nDead == -1 && i == n - 1) {  // the payload sits left of the blob (:731-735)
      x = lowestX - 1 * params.min_radius * params.radFactor - 2 * params.min_radius;
      y = 0;
    }
    hPos[2 * i] = x;
    hPos[2 * i + 1] = y;
    grid.add((int)i, x, y);
    open.push_back(i);
  }
}

void Particlebot::reset() {
  // particlebot.cpp:485-801: every ParticlebotConfig value (from a .cfg only CONFIG_RANDOM is
  // reachable in the reference -- the config key never takes effect, main.cpp:794-809 -- the
  // extension key pb_placement selects the others).
  time = 0;
  if (sim) pbSimSetTime(sim, 0.0f);
  const uint n = params.nCells;
  std::fill(hVelV.begin(), hVelV.end(), 0.0f);
  if (squareLattice) {
    // extension: side x side bots, row-major, centred on the origin
    const uint side = (uint)ceilf(sqrtf((float)n));
    const float pitch = hexSpacing > 0.0f ? hexSpacing : params.min_radius * 2.0f;
    const float half = (float)(side - 1) * 0.5f;
    for (uint i = 0; i < n; i++) {
      hPos[2 * i] = ((float)(i % side) - half) * pitch;
      hPos[2 * i + 1] = ((float)(i / side) - half) * pitch;
    }
    particlebotConfigSize.x = particlebotConfigSize.y = side;
  } else if (fastBlob) {
    placeFastBlob();
  } else
  switch (params.config) {
    case CONFIG_HEX:
      particlebotConfigSize.x = (int)ceilf(powf((float)n, 1.0f / 2.0f));
      initHexGrid(n, hexSpacing > 0.0f ? hexSpacing : params.min_radius * 2.0f);
      break;
    case CONFIG_GRID: {
      const float jitter = params.max_radius * 0.01f;
      const uint s = (int)ceilf(powf((float)n, 1.0f / 2.0f));
      particlebotConfigSize.x = particlebotConfigSize.y = s;
      initGrid(particlebotConfigSize, params.min_radius * 2.0f, jitter, n);
    } break;
    case CONFIG_LINE:
      particlebotConfigSize.x = n;
      particlebotConfigSize.y = 1;
      initGrid(particlebotConfigSize, params.min_radius * 2.0f, params.max_radius * 0.00f, n);
      break;
    case CONFIG_BLOB:
    case CONFIG_BLOB_UPLEFT:
    case CONFIG_LIGHTTEST_7: {
      // particlebot.cpp:492-611: three hand-laid 10-bot clusters on a triangular lattice of pitch
      // 2*min_radius (the reference asserts nCells == 10; extra bots here stay at the origin).
      // Entries are (x, y) in units of r: plain numbers, or k = 1 + sqrt3 / s = sqrt3 multiples.
      const float r = params.min_radius, s3 = powf(3.0f, 0.5f);
      const float k = -(1.0f + s3) * r, K = (1.0f + s3) * r, s = s3 * r, S = -s3 * r, s2 = s3 * 2.0f * r;
      const float blob[10][2] = {{r, -r}, {r, r}, {-r, -r}, {-r, r}, {k, 0.0f}, {0.0f, k}, {0.0f, K},
                                 {2.0f * r, k}, {2.0f * r, K}, {K, 0.0f}};
      const float upleft[10][2] = {{-r, r}, {r, r}, {-r, -r}, {r, -r}, {0.0f, k}, {k, 0.0f}, {K, 0.0f},
                                   {k, 2.0f * r}, {K, 2.0f * r}, {0.0f, K}};
      const float light7[10][2] = {{0.0f, 0.0f}, {S, r}, {s, -r}, {s, r}, {0.0f, 2.0f * r}, {S, -r}, {0.0f, -2.0f * r},
                                   {s, 3.0f * r}, {0.0f, 4.0f * r}, {s2, 2.0f * r}};
      const float(*t)[2] = params.config == CONFIG_BLOB ? blob : params.config == CONFIG_BLOB_UPLEFT ? upleft : light7;
      for (uint i = 0; i < n && i < 10; i++) {
        hPos[2 * i] = t[i][0];
        hPos[2 * i + 1] = t[i][1];
      }
      particlebotConfigSize.x = particlebotConfigSize.y = 4;
    } break;
    case CONFIG_RANDOM:
    default:
      placeRandom();
      break;
  }
  if (!params.Nx) params.Nx = particlebotConfigSize.x;
  for (uint i = 0; i < n; i++) {
    hRad[i] = params.min_radius;
    if (params.nDead == -1 && i == n - 1) {
      hRad[i] = params.min_radius * params.radFactor;
      hDead[i] = 1;
    }
    hphase[i] = 0;
  }
  if (engineKind == Engine::HostOnly) return;
  if (engineKind == Engine::Fused) {
    if (pbSimSetState(sim, hPos, hVel, hRad, hphase, hDead) != PB_OK) die("pbSimSetState");
  } else {
    copyArrayToDevice(dDead, hDead, 0, (int)(n * sizeof(int)));
    setArray(RADII, hRad, 0, n);
    setArray(PHASE, hphase, 0, n);
    setArray(POSITION, hPos, 0, n);
    setArray(VELOCITY, hVel, 0, n);
  }
}

// ---- one placement for several members (extension; include/particlebot.h) --------------------------
std::string Particlebot::placementKeyOf(const SimParams &params, float hexSpacing, bool squareLattice, bool fastBlob) {
  // everything reset() and the placement routines above read (grep `params.` / `rng` / the setters in this file)
  struct Key {
    unsigned seed, nCells, config, gridX, gridY, Nx;
    int payload, square, fast;
    float minRadius, maxRadius, radFactor, originX, originY, cellX, cellY, pitch;
  } k;
  memset(&k, 0, sizeof k);
  k.seed = params.seed, k.nCells = params.nCells, k.config = (unsigned)params.config;
  k.gridX = params.gridSize.x, k.gridY = params.gridSize.y, k.Nx = params.Nx;
  k.payload = params.nDead == -1 ? 1 : 0, k.square = squareLattice ? 1 : 0, k.fast = fastBlob ? 1 : 0;
  k.minRadius = params.min_radius, k.maxRadius = params.max_radius;
  k.radFactor = k.payload ? params.radFactor : 0.0f;
  k.originX = params.worldOrigin.x, k.originY = params.worldOrigin.y;
  k.cellX = params.cellSize.x, k.cellY = params.cellSize.y, k.pitch = hexSpacing;
  return std::string((const char *)&k, sizeof k);
}

void Particlebot::exportPlacement(Placement &out) const {
  out.pos = hPosV, out.rad = hRadV, out.phase = hPhaseV, out.dead = hDeadV;
  rng.getState(out.rng);
  out.configX = particlebotConfigSize.x, out.configY = particlebotConfigSize.y, out.Nx = params.Nx;
}

bool Particlebot::importPlacement(const Placement &in) {
  if (engineKind != Engine::HostOnly || in.pos.size() != hPosV.size() || in.rad.size() != hRadV.size() ||
      in.phase.size() != hPhaseV.size() || in.dead.size() != hDeadV.size())
    return false;
  time = 0;
  std::fill(hVelV.begin(), hVelV.end(), 0.0f);
  std::copy(in.pos.begin(), in.pos.end(), hPosV.begin());
  std::copy(in.rad.begin(), in.rad.end(), hRadV.begin());
  std::copy(in.phase.begin(), in.phase.end(), hPhaseV.begin());
  std::copy(in.dead.begin(), in.dead.end(), hDeadV.begin());
  rng.setState(in.rng);
  particlebotConfigSize.x = in.configX, particlebotConfigSize.y = in.configY;
  params.Nx = in.Nx;
  return true;
}

// ---- exact checkpoints (extension) -----------------------------------------

namespace {
// '2': the phase-noise generator kind follows the draw counter ('1' files had the counter generator only)
const char kCkptMagic[8] = {'P', 'B', 'C', 'K', 'P', 'T', '2', 0};
const char kCkptMagicV1[8] = {'P', 'B', 'C', 'K', 'P', 'T', '1', 0};
template <class T>
bool putv(FILE *fp, const T *p, size_t count) { return fwrite(p, sizeof(T), count, fp) == count; }
template <class T>
bool getv(FILE *fp, T *p, size_t count) { return fread(p, sizeof(T), count, fp) == count; }
}  // namespace

// ---- headless frame (stands in for display() + the OpenCV video writer) -------------------------

bool Particlebot::writeFramePPM(const char *path, int width, int height, float centerX, float centerY,
                                float halfExtent, float lightRadius) {
  if (!path || width <= 0 || height <= 0 || !(halfExtent > 0)) return false;
  pullState(true, false, true);
  const int *deadNow = getDeadArray();
  std::vector<unsigned char> img((size_t)width * height * 3, 245);
  const float scale = 0.5f * (float)height / halfExtent;  // pixels per world unit
  // world -> pixel: x mirrored (the reference translates by -x), y up
  auto px = [&](float x) { return 0.5f * (float)width - (x - centerX) * scale; };
  auto py = [&](float y) { return 0.5f * (float)height - (y - centerY) * scale; };
  auto disc = [&](float x, float y, float r, unsigned char R, unsigned char G, unsigned char B) {
    const float cx = px(x), cy = py(y), pr = r * scale;
    const int x0 = std::max(0, (int)floorf(cx - pr)), x1 = std::min(width - 1, (int)ceilf(cx + pr));
    const int y0 = std::max(0, (int)floorf(cy - pr)), y1 = std::min(height - 1, (int)ceilf(cy + pr));
    for (int yy = y0; yy <= y1; yy++)
      for (int xx = x0; xx <= x1; xx++) {
        const float dx = (float)xx + 0.5f - cx, dy = (float)yy + 0.5f - cy;
        if (dx * dx + dy * dy <= pr * pr) {
          unsigned char *p = &img[((size_t)yy * width + xx) * 3];
          p[0] = R, p[1] = G, p[2] = B;
        }
      }
  };
  for (int k = 0; k < params.nobstacles; k++) {  // rectangles
    const float xa = px(params.x2obs[k]), xb = px(params.x1obs[k]);  // mirrored: x2 is left of x1
    const float ya = py(params.y2obs[k]), yb = py(params.y1obs[k]);
    for (int yy = std::max(0, (int)floorf(ya)); yy <= std::min(height - 1, (int)ceilf(yb)); yy++)
      for (int xx = std::max(0, (int)floorf(xa)); xx <= std::min(width - 1, (int)ceilf(xb)); xx++) {
        unsigned char *p = &img[((size_t)yy * width + xx) * 3];
        p[0] = p[1] = p[2] = 110;
      }
  }
  for (int k = 0; k < params.n_cir_obstacles; k++)
    disc(params.x_cir_obs[k], params.y_cir_obs[k], params.r_cir_obs[k], 110, 110, 110);
  disc(params.light_x, params.light_y, lightRadius, 250, 210, 40);
  const float span = params.max_radius - params.min_radius;
  for (uint i = 0; i < params.nCells; i++) {
    const float r = hRad[i];
    unsigned char R = 0, G = 0, B = 0;
    if (!deadNow[i]) {  // updateCol_k, impl.cuh:413-417
      const float g = span > 0 ? (params.max_radius - r) / span : 0.0f;
      const float b = span > 0 ? (r - params.min_radius) / span : 0.0f;
      R = 30;
      G = (unsigned char)std::min(255.0f, std::max(0.0f, 20.0f + 180.0f * g * g));
      B = (unsigned char)std::min(255.0f, std::max(0.0f, 30.0f + 180.0f * sqrtf(std::max(0.0f, b))));
    }
    disc(hPos[2 * i], hPos[2 * i + 1], r, R, G, B);
  }
  FILE *fp = fopen(path, "wb");
  if (!fp) return false;
  fprintf(fp, "P6\n%d %d\n255\n", width, height);
  const bool ok = fwrite(img.data(), 1, img.size(), fp) == img.size();
  return fclose(fp) == 0 && ok;
}

void Particlebot::restoreHostMirrors(const float *pos, const float *vel, const float *rad, const float *phase,
                                     const int *dead) {
  const size_t n = params.nCells;
  if (pos) memcpy(hPos, pos, 8 * n);
  if (vel) memcpy(hVel, vel, 8 * n);
  if (rad) memcpy(hRad, rad, 4 * n);
  if (phase) memcpy(hphase, phase, 4 * n);
  if (dead) memcpy(hDead, dead, 4 * n);
}

bool Particlebot::saveCheckpoint(FILE *fp) {
  if (engineKind != Engine::Fused || !fp) return false;
  const uint n = params.nCells;
  std::vector<float> absA(n), absR(n);
  std::vector<unsigned> orig(n), keys(n);
  int sorted = 0;
  unsigned draws = 0;
  if (pbSimGetState(sim, hPos, hVel, hRad, hphase, hDead, absA.data(), absR.data()) != PB_OK) return false;
  if (pbSimGetLayoutOf(sim, 0, orig.data(), keys.data(), &sorted) != PB_OK) return false;
  if (pbSimGetPhaseDraws(sim, &draws) != PB_OK) return false;
  // Sum|F_attr| is not maintained when nothing reads it (pbSimSetForceSums): pbSimGetState hands out NaN for
  // it then.  Store zeros, so that a resumed run that does keep the sums never starts from NaN.
  pbSimConfig conf;
  if (pbSimGetConfig(sim, &conf) != PB_OK) return false;
  if (!conf.attraction_sums) std::fill(absA.begin(), absA.end(), 0.0f);
  int rs[36];
  rng.getState(rs);
  const int kind = rngKindV;
  return putv(fp, kCkptMagic, 8) && putv(fp, &n, 1) && putv(fp, &time, 1) && putv(fp, &draws, 1) && putv(fp, &kind, 1) &&
         putv(fp, &sorted, 1) && putv(fp, rs, 36) && putv(fp, hPos, 2 * (size_t)n) && putv(fp, hVel, 2 * (size_t)n) &&
         putv(fp, hRad, n) && putv(fp, hphase, n) && putv(fp, hDead, n) && putv(fp, absA.data(), n) &&
         putv(fp, absR.data(), n) && putv(fp, orig.data(), n) && putv(fp, keys.data(), n);
}

bool Particlebot::loadCheckpoint(FILE *fp) {
  if (engineKind != Engine::Fused || !fp) return false;
  const uint n = params.nCells;
  char magic[8];
  uint fileN = 0;
  float t = 0;
  unsigned draws = 0;
  int sorted = 0, rs[36];
  if (!getv(fp, magic, 8)) return false;
  const bool v1 = memcmp(magic, kCkptMagicV1, 8) == 0;
  if ((!v1 && memcmp(magic, kCkptMagic, 8) != 0) || !getv(fp, &fileN, 1) || fileN != n) return false;
  int kind = PB_RNG_COUNTER;
  std::vector<float> absA(n), absR(n);
  std::vector<unsigned> orig(n), keys(n);
  if (!(getv(fp, &t, 1) && getv(fp, &draws, 1) && (v1 || getv(fp, &kind, 1)) && getv(fp, &sorted, 1) && getv(fp, rs, 36) &&
        getv(fp, hPos, 2 * (size_t)n) && getv(fp, hVel, 2 * (size_t)n) && getv(fp, hRad, n) && getv(fp, hphase, n) &&
        getv(fp, hDead, n) && getv(fp, absA.data(), n) && getv(fp, absR.data(), n) && getv(fp, orig.data(), n) &&
        getv(fp, keys.data(), n)))
    return false;
  if (sorted && pbSimSetLayoutOf(sim, 0, orig.data(), keys.data()) != PB_OK) return false;
  // (checkpoints written before round 3 hold NaN for a Sum|F_attr| that was not maintained)
  if (std::all_of(absA.begin(), absA.end(), [](float x) { return x != x; })) std::fill(absA.begin(), absA.end(), 0.0f);
  if (pbSimSetState(sim, hPos, hVel, hRad, hphase, hDead) != PB_OK) return false;
  if (pbSimSetForcesOf(sim, 0, absA.data(), absR.data()) != PB_OK) return false;
  // the generator the run was using (its states are a function of seed, bot and draws made)
  if (kind != PB_RNG_COUNTER && kind != PB_RNG_XORWOW_CURAND && kind != PB_RNG_XORWOW_ROCRAND) return false;
  if (kind != rngKindV) setRng(kind);
  if (pbSimSetPhaseDraws(sim, draws) != PB_OK) return false;
  rng.setState(rs);
  setTime(t);
  return true;
}
