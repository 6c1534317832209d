# refresh the judged evidence of a round.  usage (GPU box): bash tools/collect_profiles.sh <tag, e.g. r02>
# -> gpurun_out/<tag>_prof/: c2 SQ + TCC counters and kernel trace (bench loop), 1 M-slot step-kernel traffic, c3 / c4 traces and
#    bench lines, the config-5 line; tools/make_pmc_summary.py then writes profiles/<tag>_pmc_summary.json
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_prof; mkdir -p $O
# what was measured, and when: bench.py only quotes counter evidence whose source hash is that of the library it runs
python3 -c "import sys, datetime; sys.path.insert(0, '$R'); from graphenvs_amd import _lib; print(_lib.source_hash()); print(datetime.datetime.now(datetime.timezone.utc).strftime('%Y-%m-%dT%H:%MZ'))" > $O/source_hash.txt
export PMC_EXTRA=--serial-shards
bash $R/tools/pmc_sq_passes.sh ${TAG}_prof/c2 bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-1m --no-configs > $O/c2_passes.log 2>&1
cd /tmp && export TMPDIR=/tmp
export GE_B=1048576 GE_REPS=3
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_COUNT"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $O/m1_$n --output-format csv -- python3 $R/tools/step_loop.py > $O/m1_$n.log 2>&1 || echo "1m $n failed"
done
python3 $R/tools/pmc_summary.py $O/m1_FETCH_SIZE $O/m1_WRITE_SIZE --kernel step_path64 > $O/step_1m_traffic.txt 2>&1
rm -rf $O/m1_FETCH_SIZE $O/m1_WRITE_SIZE
for cfg in c3 c4; do
  bash $R/tools/pmc_sq_passes.sh ${TAG}_prof/$cfg bench.py --config $cfg --steps 130 --warmup 5 --no-cpu-baseline --no-configs > $O/${cfg}_passes.log 2>&1
done
cd $R
timeout -k 10 400 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 400 python3 tools/bench_configs.py mis ds/3 mc/2 dc/3 ppd > $O/other_configs.jsonl 2> $O/other_configs.err
cat $O/c2/kernel_medians.txt; cut -c1-200 $O/bench_c2.json; cat $O/other_configs.jsonl
