#!/bin/bash
# bin/particlebot_ensemble, 64-member slice of configs[4] (two lanes of sub-batches + RCCL's streams): default hardware
# queues (4) against GPU_MAX_HW_QUEUES=8
cd $GRAFT_REPO_ROOT
DEAD="0 5714 11429 17143 22857 28571 34286 40000"
for rep in 1 2; do for q in default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2947$rep particlerobotsimulations_amd/bin/particlebot_ensemble examples/example_dead_cells.cfg --members 64 --seed0 1000 --sub-batch -1 --set nCells 100000 --set light_x -40 --set light_y 0 --set max_time 120 --set dump_interval 6 --sweep nDead $DEAD 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline_rank0']
print('queues $q: wall %.2f s device_s %.2f lanes %d sub_batches %d' % (d['wall_s'], p['device_s'], p['lanes'], p['sub_batches']))"
done; done
