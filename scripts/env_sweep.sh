# The default bench line under one environment switch at a time (measurement aid): bash scripts/env_sweep.sh VAR v1 v2 ...
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/envsweep
mkdir -p $out
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 300 python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/line_${var}_$v.json 2> $out/line_${var}_$v.err
  python3 -c "import json; d=json.loads(open('$out/line_${var}_$v.json').read().strip().splitlines()[-1]); print('$var=$v', round(d['ms_per_step'],3), 'ms  pcg', round(d['pcg']['seconds_per_step']*1e3,3), ' iterations', d['pcg']['iterations_per_step'], ' gather', round(d['roofline_other_kernels']['k_gather']['avg_launch_us']))"
done
