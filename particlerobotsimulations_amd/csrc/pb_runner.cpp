// pb_runner.cpp -- headless replacement for the reference's GLUT main loop.
//
//   particlebot_run [config.cfg] [--set NAME VALUE]... [--engine fused|legacy] [--quiet]
//
// main.cpp:823-967 minus the window: defaults, the .cfg file (default "example.cfg"), srand(seed),
// derived grid parameters, open the CSV, construct + reset, then `for(;;){ dump(); update(); }`
// (display(), main.cpp:354-361).  Between dump rows the steps are handed to the engine in one
// batch so that it can keep one fused kernel per timestep.
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
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--set") && i + 2 < argc) {
      sets.emplace_back(argv[i + 1], argv[i + 2]);
      i += 2;
    } else if (!strcmp(argv[i], "--engine") && i + 1 < argc) {
      engine = !strcmp(argv[++i], "legacy") ? Particlebot::Engine::Legacy : Particlebot::Engine::Fused;
    } else if (!strcmp(argv[i], "--quiet")) {
      quiet = true;
    } else if (argv[i][0] != '-') {
      path = argv[i];
    } else {
      fprintf(stderr, "usage: %s [config.cfg] [--set NAME VALUE]... [--engine fused|legacy] [--quiet]\n", argv[0]);
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
  sim.reset();
  const SimParams &p = sim.getParams();
  for (;;) {
    sim.dumpParticlebot(0, p.nCells, fp, cfg.dump_interval, p.testing, p.light_x, p.light_y);
    if (sim.finished()) break;  // the reference exits from inside update() here
    const int batch = sim.stepsUntilHostEvent(cfg.timestep, cfg.dump_interval, 1 << 20);
    if (sim.advance(cfg.timestep, cfg.sort_interval, batch) == 0) break;
  }
  fclose(fp);
  return 0;
}
