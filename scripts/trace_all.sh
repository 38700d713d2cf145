# Kernel trace of a short bench run reduced to the full listing of one substep (scripts/substep_trace.py all).
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/trace_all
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/trace.log 2>&1
t=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 $root/scripts/substep_trace.py $t all > $out/substep_all.txt
rm -rf $out/trace
