cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_PORT=29461
B="particlerobotsimulations_amd/bin/particlebot_ensemble examples/example_dead_cells.cfg --members 60 --sub-batch -1 --set nCells 100000 --set max_time 30 --set light_x -40 --set light_y 0 --set dump_interval 1.5 --sweep nDead 0 10000 20000 40000"
$B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain  wall', d['wall_s'], d['pipeline_rank0'])"
$B --csv-dir /tmp/csvd 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('csvdir wall', d['wall_s'], d['pipeline_rank0'])"
ls /tmp/csvd | wc -l; head -3 /tmp/csvd/member_000007.csv; wc -l /tmp/csvd/member_000007.csv
