/* particlebot_ensemble.h -- C-ABI of the ensemble layer (libparticlebot_host.so).
 *
 * New capability, no reference counterpart: the reference runs ONE simulation per process with one
 * global `__constant__ SimParams` (particlebot_kernel_impl.cuh:27) and has no multi-GPU code at all
 * (SURVEY.md section 2).  BASELINE.json's north star asks for ensembles of independent simulations
 * (Monte-Carlo seeds, parameter sweeps) sharded over the GPUs of a node with one gather of summary
 * rows.  The layer:
 *
 *   member k of M  ->  rank k mod N                  (pbEnsembleShard)
 *   per rank       ->  ONE batched pbSim holding the rank's members: placement and dead-bot draws on
 *                      the host from each member's private libc-compatible stream, one kernel launch
 *                      per timestep for the whole batch (pbEnsembleCreate / pbEnsembleRun[Steps])
 *   exchange       ->  the members' summary rows (time, COMx, COMy, distance of the COM to the light),
 *                      one row per dump time: ONE ncclAllGather of equal-sized, NaN-padded blocks
 *                      (bin/particlebot_ensemble over RCCL; ensemble.py over torch.distributed), put
 *                      back into member order by pbEnsembleAssemble.
 */
#ifndef PARTICLEBOT_ENSEMBLE_H
#define PARTICLEBOT_ENSEMBLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Member = base .cfg + common overrides + its own overrides; an override string is
 * "name\nvalue\nname\nvalue..." in the .cfg's own line-pair format.  Returns NULL on failure. */
void *pbEnsembleCreate(const char *cfg_path, const char *common_overrides, const char **member_overrides,
                       int nmembers);
void pbEnsembleDestroy(void *ensemble);
/* Runs every member to max_time.  out: [nmembers][max_rows][4] floats; *rows = rows written per
 * member.  Returns the number of timesteps executed, -1 on error. */
long pbEnsembleRun(void *ensemble, float *out, int max_rows, int *rows);
/* Up to max_steps timesteps (stops at max_time); may be called again to continue; *rows carries the
 * row count between calls. */
long pbEnsembleRunSteps(void *ensemble, long max_steps, float *out, int max_rows, int *rows);
int pbEnsembleSynchronize(void *ensemble);
int pbEnsembleGetState(void *ensemble, int member, float *pos, float *vel, float *rad);
unsigned pbEnsembleNumBots(void *ensemble);

/* ---- pipelined form: host placement overlapped with device stepping --------------------------------------------
 * The placement of a member is host work (the reference's CONFIG_RANDOM rule draws O(N^1.5) candidates: 0.3 to 0.5
 * CPU-s for a 10^5-bot member; particlebot.cpp:612-748) and pbEnsembleCreate does all of it before the first launch.  The pipeline cuts
 * the rank's members into sub-batches of `sub_batch` members (0, or more than there are: one batch; -1: automatic,
 * pbEnsemblePipelineAutoSubBatch below: for
 * members too large to batch all at once, e.g. BASELINE config 5's 10^5-bot members) and builds the members of sub-batch
 * k+1 (k+2, ...) on `host_threads` producer threads (<= 0: the rank's share of the host cores minus one for the
 * thread that drives the device) WHILE the device steps sub-batch k: wall time = max(host, device) + one
 * sub-batch's placement instead of their sum.  Rows (and final states) do not depend on sub_batch or on the
 * number of threads: every member draws from its own private random stream and every kernel form is
 * bit-identical to the member's own CPU-oracle run (tests/test_gpu_ensemble_pipeline.py).
 * Placement starts inside pbEnsemblePipelineCreate (it returns at once); Run may be called once:
 * every member runs up to max_steps timesteps (or to max_time); out/max_rows/rows as pbEnsembleRun.
 * keep_final_states: copy every member's final pos/vel/rad to the host as its sub-batch ends
 * (pbEnsemblePipelineGetState).  Returns the timesteps per member, -1 on error. */
typedef struct pbEnsembleTimings {
  double wall_s;            /* the whole Run call: placement still outstanding + upload + steps + read-backs */
  double placement_cpu_s;   /* CPU-seconds (CLOCK_THREAD_CPUTIME_ID) the producer threads spent building members, summed */
  double placement_wait_s;  /* time the device-driving thread waited for members to be built (the device idles) */
  double upload_s;          /* pbSimCreateBatch + state upload of all sub-batches */
  double device_s;          /* stepping + summary rows (+ final-state read-back) of all sub-batches */
  int sub_batches;
  int sub_batch;            /* members per sub-batch as used */
  int host_threads;         /* producer threads as used */
  int pinned;               /* 1: the producers ran pinned to the cores of the GPU's NUMA node */
  int numa_node;            /* that node, -1 unknown */
  double placement_thread_wall_s; /* wall seconds the producers spent building members (sum over threads): exceeds
                               placement_cpu_s when the threads did not get a core each (oversubscription, quota) */
  int lanes;                /* sub-batches stepped at the same time; placement_wait_s, upload_s and device_s are per lane */
  int placements_run;       /* placements actually computed (Particlebot::reset): one per DISTINCT blob of this pipeline */
  int placements_shared;    /* members that took a copy of another member's placement (equal placement keys: same seed,
                               size, radii, grid -- e.g. the points of an nDead sweep under one seed) */
} pbEnsembleTimings;
void *pbEnsemblePipelineCreate(const char *cfg_path, const char *common_overrides, const char **member_overrides,
                               int nmembers, int sub_batch, int host_threads, int keep_final_states);
/* The same with CHECKPOINTS under checkpoint_dir (created if missing): whenever a summary row is written, and
 * when a sub-batch ends, every member of the sub-batch on the device is saved exactly (state arrays, stale slot
 * layout, both generators, its rows so far; csrc/pb_capi.cpp "ensemble checkpoints"), two generations alternating
 * so that a kill in mid-write loses nothing.  resume != 0: sub-batches with a complete checkpoint in the directory
 * continue from it (finished ones only hand back their rows; their members are not even placed); the others start
 * afresh.  The resumed run's rows (and final states of the sub-batches that still ran) equal the uninterrupted
 * run's bit for bit.  The directory belongs to one (nmembers, sub_batch) decomposition (a resume with sub_batch -1 adopts
 * the directory's, whatever the number of producer threads is this time); Run needs `out`. */
void *pbEnsemblePipelineCreateCheckpointed(const char *cfg_path, const char *common_overrides,
                                           const char **member_overrides, int nmembers, int sub_batch, int host_threads,
                                           int keep_final_states, const char *checkpoint_dir, int resume);
long pbEnsemblePipelineRun(void *pipeline, long max_steps, float *out, int max_rows, int *rows,
                           pbEnsembleTimings *timings);
void pbEnsemblePipelineDestroy(void *pipeline);
/* producer threads this pipeline started (pbHostResources.host_threads minus one for the device-driving thread when
 * host_threads <= 0 was asked for, at most one per member) */
int pbEnsemblePipelineHostThreads(void *pipeline);
/* Sub-batches Run steps at the same time, each on its batch's own stream (1 ... 4, before Run; 2 with sub_batch -1, else
 * 1; PB_PIPELINE_LANES overrides the automatic choice).  Rows and final states do not depend on it. */
int pbEnsemblePipelineSetLanes(void *pipeline, int lanes);
/* Every member also writes dir/member_<id>.csv -- the file the reference writes for that member run on its own with
 * testing 0 (particlebot.cpp:303-367: "Seed, s", the header, one row per dump interval), byte for byte: the centroid
 * columns come from the reference's own fp32 sums over the bots in order (pbSimCentroidSums), not from the rows'
 * accurately rounded mean.  ids[k]: the number of this pipeline's member k in the whole ensemble (NULL: k).  Before
 * Run; refused together with checkpoints.  0, or -1. */
int pbEnsemblePipelineSetCsvDir(void *pipeline, const char *dir, const int *ids);
/* The size sub_batch -1 stands for: whole placement rounds of the producer pool (1 ... 8) that bring a sub-batch to
 * ~3 x 10^6 bots (smaller: every step carries a launch's ramp and drain; larger: the state leaves the Infinity Cache),
 * never more than that many bots whatever the size of the pool. */
int pbEnsemblePipelineAutoSubBatch(unsigned bots_per_member, int producers);
unsigned pbEnsemblePipelineNumBots(void *pipeline);
/* ONE PLACEMENT PER DISTINCT BLOB.  Members of a pipeline (and of pbEnsembleCreate) whose placement inputs agree --
 * Particlebot::placementKey(): seed, nCells, radii, payload mode, placement kind, lattice pitch, grid; NOT nDead >= 0,
 * light position or anything else the placement does not read (reference particlebot.cpp:612-748) -- are placed once;
 * the others take a copy of the placed state and of the private libc-rand state after the placement, so their dead
 * draw continues the same stream (particlebot.cpp:178-194): every member is bit-identical to its stand-alone run.
 * PB_SHARE_PLACEMENTS=0 in the environment switches it off.  The getter: placements computed / members that took a
 * copy, so far (pbEnsembleTimings carries the same two after a Run; this also serves the device-less DryRun). */
void pbEnsemblePipelinePlacementCounts(void *pipeline, int *run, int *shared);
int pbEnsemblePipelineGetState(void *pipeline, int member, float *pos, float *vel, float *rad);
/* The consumer side without a device (CPU tests): takes the sub-batches in order as Run does, records a checksum
 * of every member's placed state instead of stepping it, dwells dwell_ms per sub-batch; *max_ahead = the most
 * members ever claimed by producers beyond the consumed ones (bounded by 3 sub-batches). */
int pbEnsemblePipelineDryRun(void *pipeline, int dwell_ms, unsigned long long *checksums, int *max_ahead);

/* ---- host resources of a rank ----------------------------------------------------------------------------------
 * Placement is host work and a node's ranks share its cores, so the producer pool of a rank is sized from what the
 * process may REALLY use: min(hardware threads, scheduler affinity, cgroup CPU quota -- cpu.max of the process's
 * cgroup and its ancestors (v2), cpu.cfs_quota_us / cpu.cfs_period_us (v1)) divided by the ranks of the node
 * (LOCAL_WORLD_SIZE / OMPI_COMM_WORLD_LOCAL_SIZE / SLURM_NTASKS_PER_NODE), at most 128.  PB_HOST_THREADS or the
 * host_threads argument override the share.  When the rank's GPU reports a NUMA node
 * (/sys/bus/pci/devices/<bus id>/numa_node >= 0) the producer threads are pinned to that node's cores
 * (local_cpulist, intersected with the affinity mask; PB_PIN_PRODUCERS=0 disables): members are placed in memory
 * next to the GPU that will receive them and the pools of different ranks do not migrate across sockets.
 * A pinned pool always fits the node: with several ranks per node an automatic share larger than numa_cpus is
 * clamped to it; a lone rank (or an explicit thread count) that is larger keeps its threads and is NOT pinned.
 * The reference has nothing of this (one device, one thread: main.cpp:350).
 * Test hooks: PB_CGROUP_ROOT (default /sys/fs/cgroup), PB_SYSFS_ROOT (default /sys), PB_PROC_SELF_CGROUP. */
typedef struct pbHostResources {
  int hardware_threads;  /* std::thread::hardware_concurrency() */
  int affinity_cpus;     /* CPU_COUNT(sched_getaffinity) */
  double cgroup_cpus;    /* quota / period of the tightest cgroup level; <= 0: unlimited */
  int usable_cpus;       /* min of the three, >= 1 */
  int local_world_size;  /* ranks sharing this node, 1 if no launcher variable is set */
  int host_threads;      /* this rank's share (what pbEnsemblePipelineCreate(host_threads <= 0) starts from) */
  int device;            /* the calling thread's HIP device, -1: none */
  int numa_node;         /* of that device, -1: unknown / not a NUMA machine */
  int numa_cpus;         /* cores of that node this process may run on, 0: unknown */
  int pin_producers;     /* 1: producer threads are pinned to those cores */
  char pci_bus_id[32];
  char rule[320];        /* the sentence bench.py prints: how host_threads came about */
} pbHostResources;
int pbHostGetResources(pbHostResources *out);
/* Parses a sysfs cpulist ("0-31,64-95") restricted to the affinity mask; returns the number of cores and, if
 * `cpus` is given, the first `cap` of them. */
int pbHostParseCpuList(const char *text, int *cpus, int cap);

/* Number of members rank `rank` of `world` runs (members rank, rank + world, ...); block size per
 * rank in the gather = pbEnsembleShard(nmembers, 0, world). */
int pbEnsembleShard(int nmembers, int rank, int world);
/* gathered: [world][per][rows][4] floats as ncclAllGather leaves them (per = pbEnsembleShard(nmembers,
 * 0, world); rank r's block holds its members in order, NaN padded).  out: [nmembers][rows][4] in
 * member order.  Returns 0, or nonzero for inconsistent sizes. */
int pbEnsembleAssemble(int nmembers, int world, int rows, const float *gathered, float *out);

#ifdef __cplusplus
}
#endif
#endif /* PARTICLEBOT_ENSEMBLE_H */
