# Soak: BASELINE configs[3] at full length (256 seeds x 120 000 steps, both .cfg files) through bin/particlebot_ensemble
# with --csv-dir; three members of each compared byte for byte with bin/particlebot_run of the same member alone.
cd ${GRAFT_REPO_ROOT:-.}
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_PORT=29481
B=particlerobotsimulations_amd/bin
for cfg in example_obstacle example_object_transport; do
  rm -rf /tmp/c4_$cfg; t0=$(date +%s.%N)
  $B/particlebot_ensemble examples/$cfg.cfg --members 256 --seed0 1000 --set max_time 1200 --set dump_interval 60 --csv-dir /tmp/c4_$cfg > /tmp/c4_$cfg.json 2>/dev/null || { echo "$cfg: runner failed"; exit 1; }
  t1=$(date +%s.%N)
  for k in 0 100 255; do
    $B/particlebot_run examples/$cfg.cfg --quiet --set seed $((1000 + k)) --set max_time 1200 --set dump_interval 60 --set testing 0 --set csv_filename /tmp/c4_single.csv > /dev/null 2>&1
    if cmp -s /tmp/c4_single.csv /tmp/c4_$cfg/member_$(printf %06d $k).csv; then echo "$cfg member $k: CSV byte-identical to particlebot_run ($(wc -l < /tmp/c4_single.csv) lines)"; else echo "$cfg member $k: DIFFERS"; exit 1; fi
  done
  python3 -c "
import json; d = json.loads(open('/tmp/c4_$cfg.json').read().strip().splitlines()[-1])
print('$cfg: 256 members x', d['steps_per_member'], 'steps, runner wall', round(d['wall_s'], 2), 's, command', round($t1 - $t0, 2), 's, files', len(__import__('os').listdir('/tmp/c4_$cfg')))"
done
