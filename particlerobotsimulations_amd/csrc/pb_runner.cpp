// pb_runner.cpp -- headless replacement for the reference's GLUT main loop.
//
//   particlebot_run [config.cfg] [--set NAME VALUE]... [--engine fused|legacy] [--quiet]
//                   [--frames DIR [--frame-size PIXELS]]
//
// main.cpp:823-967 minus the window: defaults, the .cfg file (default "example.cfg"), srand(seed),
// derived grid parameters, open the CSV, construct + reset, then `for(;;){ dump(); update(); }`
// (display(), main.cpp:354-361).  Between dump rows the steps are handed to the engine in one
// batch so that it can keep one fused kernel per timestep.  --frames DIR writes a PPM of the arena
// every VIDEO_INTERVAL timesteps (the reference's video cadence, main.cpp:455-470), viewed like the
// reference's camera: centred on (camera_x, 0), half extent camera_y * tan(30 deg).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "particlebot.h"
#include "pb_config.hpp"

int main(int argc, char **argv) {
  PbRunConfig cfg;
  std::string path = "example.cfg";
  std::vector<std::pair<std::string, std::string>> sets;
  Particlebot::Engine engine = Particlebot::Engine::Fused;
  bool quiet = false;
  std::string framesDir;
  int frameSize = 800;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--set") && i + 2 < argc) {
      sets.emplace_back(argv[i + 1], argv[i + 2]);
      i += 2;
    } else if (!strcmp(argv[i], "--engine") && i + 1 < argc) {
      engine = !strcmp(argv[++i], "legacy") ? Particlebot::Engine::Legacy : Particlebot::Engine::Fused;
    } else if (!strcmp(argv[i], "--quiet")) {
      quiet = true;
    } else if (!strcmp(argv[i], "--frames") && i + 1 < argc) {
      framesDir = argv[++i];
    } else if (!strcmp(argv[i], "--frame-size") && i + 1 < argc) {
      frameSize = atoi(argv[++i]);
    } else if (argv[i][0] != '-') {
      path = argv[i];
    } else {
      fprintf(stderr,
              "usage: %s [config.cfg] [--set NAME VALUE]... [--engine fused|legacy] [--quiet] "
              "[--frames DIR [--frame-size PIXELS]]\n",
              argv[0]);
      return 2;
    }
  }
  if (!cfg.loadFile(path)) fprintf(stderr, "warning: cannot open %s, running on defaults\n", path.c_str());
  for (auto &kv : sets) cfg.setParam(kv.first, kv.second);
  srand(cfg.params.seed);  // main.cpp:929 (the class itself draws from a private, identical stream)
  cfg.derive();

  FILE *fp = fopen(cfg.csv_filename.c_str(), "w+");
  if (!fp) {
    fprintf(stderr, "cannot open %s\n", cfg.csv_filename.c_str());
    return 1;
  }
  if (quiet) {
    // dumpParticlebot echoes "time cx cy" per row to stdout, as the reference does
    if (!freopen("/dev/null", "w", stdout)) return 1;
  }
  Particlebot sim(cfg.params, engine, cfg.wallHalf());
  sim.setExitOnMaxTime(false);
  sim.setHexSpacing(cfg.hex_spacing);
  sim.setSquareLattice(cfg.square_lattice);
  sim.setFastBlob(cfg.fast_blob);
  sim.setRng(cfg.rng_kind);
  sim.reset();
  const SimParams &p = sim.getParams();
  const int frameEvery = cfg.video_interval > 0 ? cfg.video_interval : 100;
  long stepsDone = 0, frames = 0;
  for (;;) {
    sim.dumpParticlebot(0, p.nCells, fp, cfg.dump_interval, p.testing, p.light_x, p.light_y);
    if (!framesDir.empty() && stepsDone % frameEvery == 0) {
      char name[64];
      snprintf(name, sizeof name, "/frame_%06ld.ppm", frames++);
      if (!sim.writeFramePPM((framesDir + name).c_str(), frameSize, frameSize, cfg.camera_x, 0.0f,
                             cfg.camera_y * 0.57735027f, cfg.light_radius)) {
        fprintf(stderr, "cannot write %s%s\n", framesDir.c_str(), name);
        return 1;
      }
    }
    if (sim.finished()) break;  // the reference exits from inside update() here
    int batch = sim.stepsUntilHostEvent(cfg.timestep, cfg.dump_interval, 1 << 20);
    if (!framesDir.empty()) batch = (int)std::min<long>(batch, frameEvery - stepsDone % frameEvery);
    const int ran = sim.advance(cfg.timestep, cfg.sort_interval, batch);
    if (ran == 0) break;
    stepsDone += ran;
  }
  fclose(fp);
  return 0;
}
