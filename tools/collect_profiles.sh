# refresh the judged evidence: bench line, kernel trace stats + medians, PMC traffic.  usage (GPU box): bash tools/collect_profiles.sh <tag>
set -e
TAG=${1:-r01_x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/pmc_write.log 2>&1
python3 $R/tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_step_kernel.json
cp $O/pmc_step_kernel.json $R/profiles/pmc_step_kernel.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/trace_bench.json 2> $O/trace.log
cp $(ls $O/trace/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 $R/tools/kstats.py $O/trace > $O/kernel_medians.txt
cd $R && timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
cat $O/bench.json; cat $O/kernel_medians.txt
rm -rf $O/pmc_fetch $O/pmc_write $O/trace
