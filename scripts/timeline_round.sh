# Kernel trace of the default bench command, reduced on the box to the overlap classes of scripts/solver_timeline.py and a
# per-substep list of what ran between the solves (measurement aid).
# usage (on the GPU box, from the repository root): bash scripts/timeline_round.sh [bench args...]
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/timeline
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/bench.py --no-cpu-baseline "$@" > $out/trace.log 2>&1
t=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 $root/scripts/solver_timeline.py $t > $out/solver_timeline.txt
python3 $root/scripts/substep_timeline.py $t > $out/substep_timeline.txt
rm -rf $out/trace
cat $out/solver_timeline.txt
