# Gather kernel time under two switches, each setting three times on one box (measurement aid).
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/gather_ab
mkdir -p $out
for rep in 1 2 3; do
  for cfg in "2 2" "3 2" "2 3" "3 3"; do
    set -- $cfg
    FLIP_GATHER_DEPTH=$1 FLIP_GATHER_BLOCKS_PER_CU=$2 timeout -k 10 300 python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/l.json 2> $out/l.err
    python3 -c "import json; d=json.loads(open('$out/l.json').read().strip().splitlines()[-1]); print('depth $1 blocks/CU $2 rep $rep:', round(d['ms_per_step'],3), 'ms  gather', round(d['roofline_other_kernels']['k_gather']['avg_launch_us'],1))"
  done
done
