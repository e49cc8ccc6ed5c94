# thermo-step cost at the driver's arguments (20 steps, the last one a thermo step): two bench lines + the MD parity tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_thermo_cost; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_md.py tests/test_gpu_angle.py -q -m gpu -x > $O/md.log 2>&1; echo "md rc=$?"; tail -3 $O/md.log
for k in 1 2 3; do timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$k.json 2> $O/bench_$k.err; echo "bench rc=$?"; python3 - <<PY
import json
for ln in open("$O/bench_$k.json"):
    if ln.startswith("{"):
        d = json.loads(ln); print(d["value"], d["ms_per_step"], d["loop_sections_s"])
PY
done
