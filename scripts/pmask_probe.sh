# NOTE: drives experiment knobs (CUDAMAT_PB_SLABS / _SLAB_ORDER / _PMASK_MB / _SEG) that exist only in commits d474d75 and
# 7505629 (reverted afterwards); the logs of those runs are profiles/r02_probes/{slab,pmask,pmask_seg}_probe.log
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmask
rm -rf $O && mkdir -p $O
for seg in 48 24 12 6; do
for mb in 0 64; do
  export CUDAMAT_PB_SEG=$seg
  if [ $mb = 0 ]; then unset CUDAMAT_PB_PMASK_MB; else export CUDAMAT_PB_PMASK_MB=$mb; fi
  echo "seg $seg window $mb MB"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p${seg}_$mb -- python3 $R/scripts/pmask_probe.py 2> $O/p${seg}_$mb.err | grep PMASK
  f=$(find $O/p${seg}_$mb -name "*kernel_stats.csv" | head -1)
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_pb_phase' in r['Name']: print('   ', r['Name'][:32], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1))
PY
done
done
