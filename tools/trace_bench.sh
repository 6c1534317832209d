# kernel-trace medians + idle gaps of the bench loop.  usage (GPU box): bash tools/trace_bench.sh <tag> [bench.py args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-1m "$@" > $O/trace_bench.json 2> $O/trace.log || echo "trace failed"
cp $(ls $O/trace/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 $R/tools/kstats.py $O/trace > $O/kernel_medians.txt
python3 $R/tools/gaps.py $O/trace > $O/gaps.txt
python3 $R/tools/timeline.py $O/trace > $O/timeline.txt
cat $O/kernel_medians.txt $O/gaps.txt; head -60 $O/timeline.txt; cut -c1-400 $O/trace_bench.json
rm -rf $O/trace
