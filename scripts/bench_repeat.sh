# The default bench line several times on one box (box-to-box differences are of the size of the changes being measured).
# usage (on the GPU box, from the repository root): bash scripts/bench_repeat.sh N [bench args...]
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/repeat
mkdir -p $out
n=$1; shift
for i in $(seq 1 $n); do
  timeout -k 10 300 python3 $root/bench.py --no-cpu-baseline "$@" > $out/line_$i.json 2> $out/line_$i.err
  python3 -c "import json,sys; d=json.loads(open('$out/line_$i.json').read().strip().splitlines()[-1]); print('run $i', round(d['ms_per_step'],3), 'ms  pcg', round(d['pcg']['seconds_per_step']*1e3,3), ' hot', round(d['hot_path_only']['ms_per_step'],3))"
done
