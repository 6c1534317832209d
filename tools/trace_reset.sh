# kernel medians of full resets of one env config (GPU box): bash tools/trace_reset.sh <tag> <env_id> <B> <resets> '<json kwargs>'
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/tools/reset_loop.py "$@" > $O/trace.out 2> $O/trace.log || echo "trace failed"
python3 $R/tools/kstats.py $O/trace > $O/kernel_medians.txt
cat $O/kernel_medians.txt
rm -rf $O/trace
