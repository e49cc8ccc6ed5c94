set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof1m -o f --output-format csv -- python3 bench.py --cpu-steps 0 > gpurun_out/prof1m.json 2> gpurun_out/prof1m.err
python3 profiles/summarize_stats.py gpurun_out/prof1m/f_kernel_stats.csv 14
python3 -c "import json; d=json.load(open('gpurun_out/prof1m.json')); print(d['value'], d['ms_per_step'])"
