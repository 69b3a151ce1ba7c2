export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# A/B on ONE box: cuda_mat_amd/libbase.so.keep (baseline build) vs the current libcudamat_hip.so, alternating
cd /root/repo
cp cuda_mat_amd/libcudamat_hip.so /tmp/new.so
for i in 1 2; do
  for v in new base; do
    if [ $v = base ]; then cp cuda_mat_amd/libbase.so.keep cuda_mat_amd/libcudamat_hip.so; else cp /tmp/new.so cuda_mat_amd/libcudamat_hip.so; fi
    CUDAMAT_SPMV_MODE=pb python bench.py --steps 25 --warmup 3 --cpu-baseline off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], 'C4', round(d['value'],1), round(d['roofline']['avg_launch_ms'],3))" $v
    CUDAMAT_OVERLAP=0 python scripts/rank_probe.py 8 2>&1 | grep "standard  plain" | sed "s/^/$v /"
  done
done
cp /tmp/new.so cuda_mat_amd/libcudamat_hip.so
