# the two lines the driver records, on the final code
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final2; mkdir -p $O
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/err.txt; echo rc=$?
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2>> $O/err.txt; echo rc=$?
timeout -k 10 300 python3 bench.py --workload walk1m_angles --cpu-steps 0 > $O/bench_walk1m_angles.json 2>> $O/err.txt; echo rc=$?
LAMMPS_LE_BENCH_SHM=1 timeout -k 10 600 python3 bench.py --gpus 4 --steps 20 --warmup 5 > $O/bench_shm4_rehearsal.json 2> $O/err_shm4.txt; echo "shm4 rc=$?"
for f in bench_driver_args bench_default bench_walk1m_angles bench_shm4_rehearsal; do python3 -c "
import json; j=json.loads(open('$O/$f.json').read().strip().split('\n')[-1]); print('$f', j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['roofline']['traffic'], j['le_firing'] and j['le_firing']['ms_per_period'], j.get('halo'))"; done
