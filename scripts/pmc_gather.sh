# Hardware counters of the gather kernel (scripts/time_p2g.py), one rocprofv3 pass per counter group.
# usage (GPU box, repository root): bash scripts/pmc_gather.sh TAG
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_gather_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 $root/scripts/time_p2g.py > $out/$name.log 2>&1
    f=$(find $out/$name -name "*counter_collection.csv" | head -1)
    python3 $root/profiles/summarize_pmc.py $f $(echo "$@" | tr ' ' ',') $out/$name.csv "k_gather<"
    rm -rf $out/$name
    cat $out/$name.csv
}
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass fetch FETCH_SIZE
