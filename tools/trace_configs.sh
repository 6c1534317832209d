# kernel trace of the side configs.  usage (GPU box): bash tools/trace_configs.sh dc c5 ...  -> gpurun_out/r03_trace/<cfg>_{kernel_stats.csv,medians.txt}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/$cfg --output-format csv -- python3 $R/tools/bench_configs.py $cfg > $O/$cfg.log 2>&1 || echo "$cfg failed"
  f=$(find $O/$cfg -name '*kernel_stats.csv' | head -1); cp $f $O/${cfg}_kernel_stats.csv
  python3 $R/tools/kstats.py $O/$cfg > $O/${cfg}_medians.txt 2>&1 || true
  python3 $R/tools/steady_sums.py $O/$cfg > $O/${cfg}_steady.txt 2>&1 || true
  rm -rf $O/$cfg
  cat $O/${cfg}_steady.txt; tail -2 $O/$cfg.log | cut -c1-300
done
