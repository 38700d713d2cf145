# Collects the per-round evidence of the default bench command into gpurun_out/prof_<tag>/ (copy what is to be judged
# into profiles/): the bench line, the rocprofv3 kernel statistics, and FETCH_SIZE / WRITE_SIZE in passes of their own.
# Every pass runs the same command (same --steps / --warmup), so the counters see the scene state the line was timed on.
# usage (on the GPU box, from the repository root): bash scripts/profile_round.sh TAG [bench args...]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 python3 $root/bench.py "$@" > $out/line.json 2> $out/line.err
tail -c 300 $out/line.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --no-cpu-baseline "$@" > $out/stats.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --no-cpu-baseline "$@" > $out/pmc_fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --no-cpu-baseline "$@" > $out/pmc_write.log 2>&1
f=$(find $out/pmc_fetch -name "*counter_collection.csv" | head -1)
w=$(find $out/pmc_write -name "*counter_collection.csv" | head -1)
python3 $root/profiles/summarize_pmc.py $f FETCH_SIZE $out/pmc_fetch_size.csv
python3 $root/profiles/summarize_pmc.py $w WRITE_SIZE $out/pmc_write_size.csv
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
ls $out
# the raw traces and counter dumps are tens of MB each: only the summaries travel back
rm -rf $out/stats $out/pmc_fetch $out/pmc_write
ls $out
