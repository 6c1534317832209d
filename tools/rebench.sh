# the three bench lines again, once profiles/<tag>_pmc_summary.json of THIS tree is in place (tools/collect_profiles.sh runs them before
# tools/make_pmc_summary.py has written it, so their roofline.traffic is empty).  usage (GPU box): bash tools/rebench.sh r03
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_prof; mkdir -p $O
cd $R
timeout -k 10 400 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 300 python3 bench.py --config c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
timeout -k 10 300 python3 bench.py --config c4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
cut -c1-400 $O/bench_c2.json $O/bench_c3.json $O/bench_c4.json
