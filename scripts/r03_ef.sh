# thermo step as one launch of the step kernel's energy variant: parity, then the 20-step line with and without it, then k_step under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03_ef}; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_md.py tests/test_gpu_le.py tests/test_gpu_misc.py tests/test_gpu_angle.py -x -q -m gpu > $O/quick.log 2>&1; echo "quick rc=$?"; tail -3 $O/quick.log
for rep in 1 2 3; do
  for m in fused unfused; do
    if [ $m = unfused ]; then export LAMMPS_LE_NO_FUSED_THERMO=1; else unset LAMMPS_LE_NO_FUSED_THERMO; fi
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 > $O/b20_$m.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
    python3 -c "
import json; j=json.loads(open('$O/b20_$m.json').read().strip().split('\n')[-1]); print('rep $rep $m 20 steps', j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])"
  done
done
unset LAMMPS_LE_NO_FUSED_THERMO
bash scripts/r03_bench.sh $1 walk1m
