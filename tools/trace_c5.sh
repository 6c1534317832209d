# kernel trace of config 5's loop + GPU busy fraction (GPU box): bash tools/trace_c5.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tc5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace --output-format csv -- python3 $R/bench.py --config c5 --no-cpu-baseline --steps 200 > $O/bench.json 2> $O/trace.log || echo "trace failed"
python3 $R/tools/busy_union.py $O/trace
cut -c1-200 $O/bench.json
rm -rf $O/trace
