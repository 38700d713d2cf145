# Iteration counts of the slab solver against ghost thickness and preconditioner overlap: all ranks share the one GPU
# (gloo), so only the iteration counts mean anything.  usage: rehearse_overlap.sh N "ranks ghost overlap" ...
set -e
mkdir -p gpurun_out
n=$1; shift
for cfg in "$@"; do
  set -- $cfg
  FLIP_BENCH_OVERLAP=$3 FLIP_BENCH_N=$n FLIP_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $1 --ghost $2 --steps 2 --warmup 1 > gpurun_out/reh_$1_$2_$3.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/reh_$1_$2_$3.log") if x.startswith("{")][-1]
d=json.loads(l)
print("n $n ranks $1 ghost $2 overlap $3:", round(d["ms_per_step"],1), "ms", d["pcg"]["iterations_per_step"], "iterations", d["pcg"]["converged"])
PY
done
