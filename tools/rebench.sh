# the three bench lines again, once profiles/<tag>_pmc_summary.json of THIS tree is in place (tools/collect_profiles.sh runs them before
# tools/make_pmc_summary.py has written it, so their roofline.traffic is empty).  usage (GPU box): bash tools/rebench.sh r03
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_prof; mkdir -p $O
cd $R
timeout -k 10 400 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
cut -c1-400 $O/bench_c2.json
