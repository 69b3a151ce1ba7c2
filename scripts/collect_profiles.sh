#!/bin/bash
export CUDAMAT_BENCH_OTHER_CONFIGS=off   # the headline alone: no side sections (bench.py other_configs) under a profiler / in an A/B
# run on the GPU box: rocprofv3 passes for the three single-GPU configurations -> gpurun_out/prof_*
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
run() { # name, bench args...
  name=$1; shift
  rm -rf $O/prof_$name $O/pmcf_$name $O/pmcw_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 $R/bench.py "$@" --cpu-baseline off --drop-in off --other-configs off > $O/prof_$name.json 2> $O/prof_$name.err
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$name -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --cpu-baseline off --drop-in off --other-configs off > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$name -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --cpu-baseline off --drop-in off --other-configs off > /dev/null 2>&1
  echo "done $name"
}
run rand50 --steps 10 --warmup 2
run poisson5 --workload poisson5 --steps 20 --warmup 2
run ilu0 --precond ilu0 --steps 4 --warmup 1
run mat10000 --workload mat10000 --steps 2000 --warmup 200
