set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
rm -rf $O/prof_mat10000
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mat10000 -- python3 $R/bench.py --workload mat10000 --steps 2000 --warmup 200 --cpu-baseline off > $O/prof_mat10000.json 2> $O/prof_mat10000.err
f=$(find $O/prof_mat10000 -name "*kernel_stats.csv" | head -1)
head -8 $f | cut -c1-160
tail -1 $O/prof_mat10000.json | cut -c1-300
