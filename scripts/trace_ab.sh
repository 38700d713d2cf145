# Kernel trace of a short bench run, with and without the side stream, reduced to scripts/substep_trace.py's listing
# (measurement aid).  usage (on the GPU box, from the repository root): bash scripts/trace_ab.sh
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/trace_ab
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for side in 1 0; do
  export FLIP_SIDE_STREAM=$side
  timeout -k 10 300 python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/line_$side.json 2> $out/line_$side.err
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$side -- python3 $root/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/trace_$side.log 2>&1
  t=$(find $out/trace_$side -name "*kernel_trace.csv" | head -1)
  python3 $root/scripts/substep_trace.py $t > $out/substep_$side.txt
  rm -rf $out/trace_$side
done
