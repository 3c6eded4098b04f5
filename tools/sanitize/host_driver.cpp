// ASan/UBSan driver for the host-side code paths that need no GPU (HostOnly engine).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
extern "C" {
void *pbHostCreate(const char *cfg, const char *over, int engine);
void pbHostDestroy(void *);
void pbHostReset(void *);
int pbHostDrawDead(void *, int *out);
int pbHostDump(void *, const char *path, const char *mode);
int pbHostWriteFrame(void *, const char *path, int w, int h, float cx, float cy, float half);
unsigned pbHostNumBots(void *);
int pbHostGetArray(void *, int which, void *out);
void *pbEnsembleCreate(const char *cfg, const char *common, const char **members, int n);
}
int main(int argc, char **argv) {
  const char *root = argv[1];
  const char *names[] = {"example.cfg", "example_dead_cells.cfg", "example_gap.cfg", "example_object_transport.cfg",
                         "example_obstacle.cfg"};
  for (const char *nm : names) {
    char path[512];
    snprintf(path, sizeof path, "%s/examples/%s", root, nm);
    for (const char *place : {"random", "hex", "grid", "line", "blob", "blob_upleft", "lighttest7", "square"}) {
      char over[128];
      snprintf(over, sizeof over, "pb_placement\n%s\ntime_to_dead\n0", place);
      void *h = pbHostCreate(path, over, 2);
      if (!h) { printf("create failed %s %s\n", nm, place); return 1; }
      pbHostReset(h);
      unsigned n = pbHostNumBots(h);
      std::vector<int> dead(n);
      pbHostDrawDead(h, dead.data());
      pbHostDump(h, "d.csv", "w");
      if (pbHostWriteFrame(h, "f.ppm", 300, 200, 0, 0, 0) != 0) { printf("frame failed\n"); return 1; }
      std::vector<float> pos(2 * (size_t)n);
      pbHostGetArray(h, 0, pos.data());
      pbHostDestroy(h);
    }
  }
  // tiny and large placements
  for (const char *nc : {"1", "2", "3", "5000"}) {
    char path[512], over[64];
    snprintf(path, sizeof path, "%s/examples/example.cfg", root);
    snprintf(over, sizeof over, "nCells\n%s", nc);
    void *h = pbHostCreate(path, over, 2);
    pbHostReset(h);
    pbHostDestroy(h);
  }
  // ensemble creation builds members on threads, then fails cleanly without a GPU
  char path[512];
  snprintf(path, sizeof path, "%s/examples/example_obstacle.cfg", root);
  const char *members[] = {"seed\n1", "seed\n2", "seed\n3", "seed\n4", "seed\n5", "seed\n6"};
  void *e = pbEnsembleCreate(path, "max_time\n1", members, 6);
  printf("ensemble without GPU: %s\n", e ? "created?!" : "null (expected)");
  printf("asan driver done\n");
  return 0;
}
