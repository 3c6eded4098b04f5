// pb_runner.cpp -- headless replacement for the reference's GLUT main loop.
//
//   particlebot_run [config.cfg] [--set NAME VALUE]... [--engine fused|legacy] [--quiet]
//                   [--frames DIR [--frame-size PIXELS]]
//                   [--resume FILE [--overwrite-csv]] [--checkpoint FILE [--checkpoint-every SECONDS] [--checkpoint-steps N]]
//                   [--final-checkpoint FILE]
//
// --resume FILE continues a run (main.cpp:954-957 -> particlebot.cpp:369-411): FILE is either an exact checkpoint
// written by --checkpoint (time, every state array incl. phase / dead / force sums, the stale slot layout, both
// random generators, and the length the CSV had: the resumed run is bit-identical to the uninterrupted one and the
// CSV ends up byte-identical) or, as in the reference, a testing=1 CSV whose last complete row supplies time,
// positions, velocities and radii (phases, dead set and force sums start afresh: the reference's lossy resume).
// When that CSV is csv_filename itself -- the same file by device and inode, however the two paths are spelled -- the
// run appends to it as the reference does (main.cpp:940-956); when it is another file, an existing non-empty
// csv_filename is only truncated with --overwrite-csv.
// --checkpoint FILE is rewritten (write to FILE.tmp, rename) every --checkpoint-every SECONDS of wall time
// and/or every --checkpoint-steps timesteps, always at a point where the main loop is about to dump.
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

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>

#include "particlebot.h"
#include "pb_config.hpp"

namespace {

// what the runner appends to the class's exact checkpoint: where its own loop was
struct RunTrailer {
  char magic[8] = {'P', 'B', 'R', 'U', 'N', 'v', '1', 0};
  long long csvBytes = 0;  // length of the CSV when the checkpoint was taken (rows written since are cut off)
  long long stepsDone = 0, frames = 0;
};

bool isExactCheckpoint(const std::string &path) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char m[8] = {0};
  const bool ok = fread(m, 1, 8, f) == 8 && memcmp(m, "PBCKPT", 6) == 0;
  fclose(f);
  return ok;
}

bool readTrailer(const std::string &path, RunTrailer &t) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  RunTrailer got;
  const bool ok = fseek(f, -(long)sizeof got, SEEK_END) == 0 && fread(&got, sizeof got, 1, f) == 1 &&
                  memcmp(got.magic, t.magic, 8) == 0;
  fclose(f);
  if (ok) t = got;
  return ok;
}

}  // namespace

int main(int argc, char **argv) {
  PbRunConfig cfg;
  std::string path = "example.cfg";
  std::vector<std::pair<std::string, std::string>> sets;
  Particlebot::Engine engine = Particlebot::Engine::Fused;
  bool quiet = false, overwriteCsv = false;
  std::string framesDir, resumePath, ckptPath, finalCkptPath;
  int frameSize = 800;
  double ckptEverySeconds = 0.0;
  long ckptEverySteps = 0, stopAfterSteps = -1;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--set") && i + 2 < argc) {
      sets.emplace_back(argv[i + 1], argv[i + 2]);
      i += 2;
    } else if (!strcmp(argv[i], "--engine") && i + 1 < argc) {
      engine = !strcmp(argv[++i], "legacy") ? Particlebot::Engine::Legacy : Particlebot::Engine::Fused;
    } else if (!strcmp(argv[i], "--quiet")) {
      quiet = true;
    } else if (!strcmp(argv[i], "--overwrite-csv")) {
      overwriteCsv = true;
    } else if (!strcmp(argv[i], "--frames") && i + 1 < argc) {
      framesDir = argv[++i];
    } else if (!strcmp(argv[i], "--frame-size") && i + 1 < argc) {
      frameSize = atoi(argv[++i]);
    } else if (!strcmp(argv[i], "--resume") && i + 1 < argc) {
      resumePath = argv[++i];
    } else if (!strcmp(argv[i], "--checkpoint") && i + 1 < argc) {
      ckptPath = argv[++i];
    } else if (!strcmp(argv[i], "--final-checkpoint") && i + 1 < argc) {
      finalCkptPath = argv[++i];
    } else if (!strcmp(argv[i], "--checkpoint-every") && i + 1 < argc) {
      ckptEverySeconds = atof(argv[++i]);
    } else if (!strcmp(argv[i], "--checkpoint-steps") && i + 1 < argc) {
      ckptEverySteps = atol(argv[++i]);
    } else if (!strcmp(argv[i], "--stop-after-steps") && i + 1 < argc) {
      stopAfterSteps = atol(argv[++i]);  // (tests: die like a killed process once this many steps have run)
    } else if (argv[i][0] != '-') {
      path = argv[i];
    } else {
      fprintf(stderr,
              "usage: %s [config.cfg] [--set NAME VALUE]... [--engine fused|legacy] [--quiet] "
              "[--frames DIR [--frame-size PIXELS]] [--resume FILE [--overwrite-csv]] [--checkpoint FILE "
              "[--checkpoint-every SECONDS] [--checkpoint-steps N]] [--final-checkpoint FILE]\n",
              argv[0]);
      return 2;
    }
  }
  if (!cfg.loadFile(path)) fprintf(stderr, "warning: cannot open %s, running on defaults\n", path.c_str());
  for (auto &kv : sets) cfg.setParam(kv.first, kv.second);
  srand(cfg.params.seed);  // main.cpp:929 (the class itself draws from a private, identical stream)
  cfg.derive();

  // resuming from an exact checkpoint keeps the CSV written so far (cut back to the length it had when the
  // checkpoint was taken); every other start truncates it, as the reference's fopen("w+") does (main.cpp:944)
  RunTrailer trailer;
  const bool exactResume = !resumePath.empty() && isExactCheckpoint(resumePath);
  if (exactResume && !readTrailer(resumePath, trailer)) {
    fprintf(stderr, "%s: not a checkpoint written by particlebot_run --checkpoint\n", resumePath.c_str());
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
  sim.setForceVariant(cfg.force_variant);
  sim.reset();
  const SimParams &p = sim.getParams();
  const int frameEvery = cfg.video_interval > 0 ? cfg.video_interval : 100;
  long stepsDone = 0, frames = 0;
  if (!resumePath.empty()) {
    FILE *rf = fopen(resumePath.c_str(), exactResume ? "rb" : "r");
    if (!rf) {
      fprintf(stderr, "cannot open %s\n", resumePath.c_str());
      return 1;
    }
    if (exactResume) {
      if (!sim.loadCheckpoint(rf)) {
        fprintf(stderr, "%s does not match this configuration (bot count?) or is damaged\n", resumePath.c_str());
        return 1;
      }
      stepsDone = trailer.stepsDone;
      frames = trailer.frames;
    } else {
      sim.loadFromFile(0, p.nCells, rf, cfg.dump_interval);  // main.cpp:954-957
    }
    fclose(rf);
  }
  // the CSV: truncated on a fresh start (main.cpp:944); appended to when the run continues from its own last row
  // (main.cpp:940-956); cut back to the length the checkpoint recorded on an exact resume
  // "Its own": the SAME FILE, however it is spelled (./run.csv, an absolute path, a symlink) -- decided by device and
  // inode, not by comparing strings; a resume from ANOTHER CSV never truncates an existing csv_filename silently
  // (`--resume ./run.csv` with csv_filename run.csv destroyed every earlier row of the run).
  bool csvResumeInPlace = false;
  if (!resumePath.empty() && !exactResume) {
    struct stat a, b;
    const bool haveA = stat(resumePath.c_str(), &a) == 0, haveB = stat(cfg.csv_filename.c_str(), &b) == 0;
    csvResumeInPlace = haveA && haveB && a.st_dev == b.st_dev && a.st_ino == b.st_ino;
    if (!csvResumeInPlace && haveB && b.st_size > 0 && !overwriteCsv) {
      fprintf(stderr,
              "%s exists and is not the file being resumed (%s): refusing to truncate it; pass --overwrite-csv or "
              "choose another csv_filename\n",
              cfg.csv_filename.c_str(), resumePath.c_str());
      return 1;
    }
  }
  FILE *fp = fopen(cfg.csv_filename.c_str(), exactResume ? "r+" : csvResumeInPlace ? "a" : "w+");
  if (!fp) {
    fprintf(stderr, "cannot open %s\n", cfg.csv_filename.c_str());
    return 1;
  }
  if (exactResume) {
    if (ftruncate(fileno(fp), (off_t)trailer.csvBytes) != 0 || fseek(fp, 0, SEEK_END) != 0) {
      fprintf(stderr, "cannot cut %s back to %lld bytes\n", cfg.csv_filename.c_str(), trailer.csvBytes);
      return 1;
    }
  }
  auto writeCheckpoint = [&](const std::string &dest) {
    // the CSV first: the checkpoint records how long it is
    fflush(fp);
    RunTrailer t;
    t.csvBytes = (long long)ftell(fp);
    t.stepsDone = stepsDone;
    t.frames = frames;
    const std::string tmp = dest + ".tmp";
    FILE *cf = fopen(tmp.c_str(), "wb");
    bool ok = cf && sim.saveCheckpoint(cf) && fwrite(&t, sizeof t, 1, cf) == 1;
    if (cf) ok = (fclose(cf) == 0) && ok;
    if (!ok || rename(tmp.c_str(), dest.c_str()) != 0) {
      fprintf(stderr, "cannot write checkpoint %s\n", dest.c_str());
      return false;
    }
    return true;
  };
  auto lastCkptWall = std::chrono::steady_clock::now();
  long lastCkptStep = stepsDone;
  for (;;) {
    // a checkpoint is taken here, where the loop is about to dump: resuming re-enters at this very point
    if (!ckptPath.empty() && stepsDone > lastCkptStep) {
      const double since = std::chrono::duration<double>(std::chrono::steady_clock::now() - lastCkptWall).count();
      if ((ckptEverySteps > 0 && stepsDone - lastCkptStep >= ckptEverySteps) ||
          (ckptEverySeconds > 0 && since >= ckptEverySeconds)) {
        if (!writeCheckpoint(ckptPath)) return 1;
        lastCkptWall = std::chrono::steady_clock::now();
        lastCkptStep = stepsDone;
      }
    }
    if (stopAfterSteps >= 0 && stepsDone >= stopAfterSteps) _exit(9);  // (no flush, no destructors: a kill)
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
    if (ckptEverySteps > 0) batch = (int)std::min<long>(batch, ckptEverySteps - (stepsDone - lastCkptStep) > 0
                                                                   ? ckptEverySteps - (stepsDone - lastCkptStep)
                                                                   : 1);
    else if (ckptEverySeconds > 0) batch = std::min(batch, 20000);  // look at the clock now and then
    const int ran = sim.advance(cfg.timestep, cfg.sort_interval, batch);
    if (ran == 0) break;
    stepsDone += ran;
  }
  if (!finalCkptPath.empty() && !writeCheckpoint(finalCkptPath)) return 1;
  fclose(fp);
  return 0;
}
