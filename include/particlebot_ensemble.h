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
