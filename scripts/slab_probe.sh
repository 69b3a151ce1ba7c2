# NOTE: drives experiment knobs (CUDAMAT_PB_SLABS / _SLAB_ORDER / _PMASK_MB / _SEG) that exist only in commits d474d75 and
# 7505629 (reverted afterwards); the logs of those runs are profiles/r02_probes/{slab,pmask,pmask_seg}_probe.log
# run on the GPU box: what is Infinity-Cache residency of the product stream worth to the REAL phase 2?
# column slabs (CUDAMAT_PB_SLABS=S): order 0 = all of phase 1, then phase 2 slab by slab (products cold, from HBM);
# order 1 = phase 1 and phase 2 alternate per slab (products just written).  Per-kernel totals from rocprofv3.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/slab
rm -rf $O && mkdir -p $O
export CUDAMAT_SPMV_MODE=pb
# correctness first: the slab path must stay bit-exact
CUDAMAT_PB_SLABS=5 python3 -m pytest $R/tests/test_gpu_parity.py -x -q -k "blocked_spmv_is_bit_exact" > $O/parity.log 2>&1 || { tail -5 $O/parity.log; exit 1; }
tail -1 $O/parity.log
run() {
  name=$1; shift
  env "$@" python3 $R/bench.py --steps 10 --warmup 2 --cpu-baseline off > $O/$name.json 2> $O/$name.err
  python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'it/s', round(d['value'],1), 'ms/spmv', round(d['roofline']['avg_launch_ms'],3))" $O/$name.json $name
}
run base CUDAMAT_PB_SLABS=0
for S in 8 32; do
  run s${S}_cold CUDAMAT_PB_SLABS=$S CUDAMAT_PB_SLAB_ORDER=0
  run s${S}_warm CUDAMAT_PB_SLABS=$S CUDAMAT_PB_SLAB_ORDER=1
done
# per-kernel durations of the two orders at S = 32 (and the baseline)
prof() {
  name=$1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-baseline off > /dev/null 2> $O/prof_$name.err
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1)
  echo "== $name"; grep -E "k_pb_phase" $f | cut -d, -f1-5
}
CUDAMAT_PB_SLABS=0 prof base
export CUDAMAT_PB_SLABS=32
CUDAMAT_PB_SLAB_ORDER=0 prof s32_cold
CUDAMAT_PB_SLAB_ORDER=1 prof s32_warm
