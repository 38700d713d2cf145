# As env_sweep.sh with bench arguments: bash scripts/env_sweep2.sh "<bench args>" VAR v1 v2 ...
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/envsweep
mkdir -p $out
args=$1; shift
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 500 python3 $root/bench.py --no-cpu-baseline $args > $out/line2_${var}_$v.json 2> $out/line2_${var}_$v.err
  python3 -c "import json; d=json.loads(open('$out/line2_${var}_$v.json').read().strip().splitlines()[-1]); print('$args $var=$v', round(d['ms_per_step'],3), 'ms  pcg', round(d['pcg']['seconds_per_step']*1e3,3), ' iterations', d['pcg']['iterations_per_step'])"
done
