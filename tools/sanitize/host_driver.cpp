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
void pbHostXorwowOutputs(int kind, unsigned long long seed, unsigned sub, unsigned count, unsigned *out);
void pbHostXorwowNormals(int kind, unsigned seed, unsigned nbots, unsigned draws, float *out);
int pbEnsembleShard(int nmembers, int rank, int world);
int pbEnsembleAssemble(int nmembers, int world, int rows, const float *gathered, float *out);
// round 4: host resources, the pipeline's producer pool and its dry-run consumer
struct pbHostResourcesBlob { char bytes[512]; };
int pbHostGetResources(pbHostResourcesBlob *out);
int pbHostParseCpuList(const char *text, int *cpus, int cap);
void *pbEnsemblePipelineCreate(const char *cfg, const char *common, const char **members, int n, int sub_batch, int host_threads,
                               int keep);
int pbEnsemblePipelineDryRun(void *p, int dwell_ms, unsigned long long *checksums, int *max_ahead);
int pbEnsemblePipelineHostThreads(void *p);
void pbEnsemblePipelineDestroy(void *p);
void pbEnsemblePipelinePlacementCounts(void *p, int *run, int *shared);
}
int main(int argc, char **argv) {
  const char *root = argv[1];
  const char *names[] = {"example.cfg", "example_dead_cells.cfg", "example_gap.cfg", "example_object_transport.cfg",
                         "example_obstacle.cfg"};
  for (const char *nm : names) {
    char path[512];
    snprintf(path, sizeof path, "%s/examples/%s", root, nm);
    for (const char *place : {"random", "hex", "grid", "line", "blob", "blob_upleft", "lighttest7", "square", "fastblob"}) {
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
  // the reference's placement rule at a size where the ring has widened many times (buried-anchor shortcut,
  // PlacementGrid::ringCovered) and a large dead-bot draw (order-statistics tree)
  {
    char path[512];
    snprintf(path, sizeof path, "%s/examples/example_dead_cells.cfg", root);
    void *h = pbHostCreate(path, "nCells\n30000\nnDead\n29999", 2);
    pbHostReset(h);
    std::vector<int> dead(30000);
    pbHostDrawDead(h, dead.data());
    long nd = 0;
    for (int d : dead) nd += d;
    if (nd != 29999) { printf("dead draw count %ld\n", nd); return 1; }
    pbHostDestroy(h);
  }
  // O(N) blob at a size where the open list churns, payload mode included
  for (const char *cfgname : {"example.cfg", "example_object_transport.cfg"}) {
    char path[512];
    snprintf(path, sizeof path, "%s/examples/%s", root, cfgname);
    void *h = pbHostCreate(path, "nCells\n20000\npb_placement\nfastblob", 2);
    pbHostReset(h);
    pbHostDestroy(h);
  }
  // XORWOW on the host (jump table build, subsequence skips, Box-Muller) and the gather layout helpers
  {
    std::vector<unsigned> u(64);
    pbHostXorwowOutputs(1, 5555ull, 12345u, 64, u.data());
    pbHostXorwowOutputs(2, 0xFFFFFFFFFFFFull, 0xFFFFFFFFu, 64, u.data());
    std::vector<float> z(3 * 257);
    pbHostXorwowNormals(1, 7u, 257, 3, z.data());
    const int members = 11, world = 4, rows = 3, per = pbEnsembleShard(members, 0, world);
    std::vector<float> gathered((size_t)world * per * rows * 4, 1.0f), out((size_t)members * rows * 4);
    if (pbEnsembleAssemble(members, world, rows, gathered.data(), out.data()) != 0) return 1;
  }
  // ensemble creation builds members on threads, then fails cleanly without a GPU
  char path[512];
  snprintf(path, sizeof path, "%s/examples/example_obstacle.cfg", root);
  const char *members[] = {"seed\n1", "seed\n2", "seed\n3", "seed\n4", "seed\n5", "seed\n6"};
  void *e = pbEnsembleCreate(path, "max_time\n1", members, 6);
  printf("ensemble without GPU: %s\n", e ? "created?!" : "null (expected)");
  // host resources (cgroup files, sysfs, affinity) and the producer pool: sized, pinned if sysfs says so, drained
  {
    pbHostResourcesBlob res;
    memset(&res, 0xee, sizeof res);
    if (pbHostGetResources(&res) != 0) return 1;
    int cpus[8];
    (void)pbHostParseCpuList("0-3,8,10-11,4096-5000,,x", cpus, 8);
    (void)pbHostParseCpuList("", nullptr, 0);
    for (int sub : {-1, 0, 2}) {
      void *p = pbEnsemblePipelineCreate(path, "nCells\n40\nmax_time\n1", members, 6, sub, sub == 2 ? 3 : 0, 0);
      if (!p) { printf("pipeline create failed\n"); return 1; }
      unsigned long long sums[6];
      int ahead = 0;
      if (pbEnsemblePipelineDryRun(p, 1, sums, &ahead) != 0) { printf("dry run failed\n"); return 1; }
      printf("pipeline sub %d: %d producers, look-ahead %d\n", sub, pbEnsemblePipelineHostThreads(p), ahead);
      pbEnsemblePipelineDestroy(p);
    }
    // shared placements (round 6): 3 seeds x 4 dead fractions -- three placements, nine copies -- with more producers
    // than groups (look-ahead placement, waits on the group's condition variable) and with one; a pipeline destroyed
    // before anything was consumed (producers stopped in mid-wait)
    const char *sweep[12] = {"seed\n7\nnDead\n0", "seed\n7\nnDead\n3", "seed\n7\nnDead\n9", "seed\n7\nnDead\n12",
                             "seed\n8\nnDead\n0", "seed\n8\nnDead\n3", "seed\n8\nnDead\n9", "seed\n8\nnDead\n12",
                             "seed\n9\nnDead\n0", "seed\n9\nnDead\n3", "seed\n9\nnDead\n9", "seed\n9\nnDead\n12"};
    snprintf(path, sizeof path, "%s/examples/example_dead_cells.cfg", root);
    unsigned long long first[12] = {0};
    for (int threads : {6, 1, 3}) {
      void *p = pbEnsemblePipelineCreate(path, "nCells\n300\nmax_time\n1", sweep, 12, 5, threads, 0);
      if (!p) { printf("sweep pipeline create failed\n"); return 1; }
      unsigned long long sums[12];
      int ahead = 0, run = 0, shared = 0;
      if (pbEnsemblePipelineDryRun(p, 0, sums, &ahead) != 0) { printf("sweep dry run failed\n"); return 1; }
      pbEnsemblePipelinePlacementCounts(p, &run, &shared);
      if (threads == 6) memcpy(first, sums, sizeof sums);
      printf("sweep with %d producers: %d placements, %d copies, checksums %s\n", threads, run, shared,
             memcmp(first, sums, sizeof sums) == 0 ? "equal" : "DIFFER");
      pbEnsemblePipelineDestroy(p);
    }
    void *p = pbEnsemblePipelineCreate(path, "nCells\n2000\nmax_time\n1", sweep, 12, 2, 4, 0);
    pbEnsemblePipelineDestroy(p);   // (nothing consumed: the producers are stopped where they are)
  }
  printf("asan driver done\n");
  return 0;
}
