# As lambda_sweep.sh for another size: bash scripts/lambda_sweep2.sh "<bench args>" lo...
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/lambda
mkdir -p $out
args=$1; shift
for lo in "$@"; do
  FLIP_MG_LAMBDA_LO=$lo timeout -k 10 400 python3 $root/bench.py --no-cpu-baseline $args > $out/line2_$lo.json 2> $out/line2_$lo.err
  python3 -c "import json; d=json.loads(open('$out/line2_$lo.json').read().strip().splitlines()[-1]); print('$args lambda_lo $lo', round(d['ms_per_step'],3), 'ms  pcg', round(d['pcg']['seconds_per_step']*1e3,3), ' iterations', d['pcg']['iterations_per_step'])"
done
