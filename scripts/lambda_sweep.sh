# Iterations and time per substep of the default bench against the lower end of the smoother's interval (measurement aid).
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/lambda
mkdir -p $out
for lo in "$@"; do
  FLIP_MG_LAMBDA_LO=$lo timeout -k 10 300 python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/line_$lo.json 2> $out/line_$lo.err
  python3 -c "import json; d=json.loads(open('$out/line_$lo.json').read().strip().splitlines()[-1]); print('lambda_lo $lo', round(d['ms_per_step'],3), 'ms  pcg', round(d['pcg']['seconds_per_step']*1e3,3), ' iterations', d['pcg']['iterations_per_step'])"
done
