# full-reset time of the generic feature path at the shapes of configs 4 / 5 (GPU box): bash tools/gen_sizes.sh [variant specs ...]
R=$GRAFT_REPO_ROOT
GE_ENV=SteinerTree-v0 GE_N=256 GE_M=1024 GE_B=2048 python3 $R/tools/variant_reset.py "$@"
GE_ENV=ShortestPath-v0 GE_N=130 GE_M=390 GE_B=4096 python3 $R/tools/variant_reset.py "$@"
GE_ENV=ShortestPath-v0 GE_N=400 GE_M=1200 GE_B=1024 python3 $R/tools/variant_reset.py "$@"
GE_ENV=DensestSubgraph-v0 GE_N=512 GE_M=1536 GE_B=512 GE_KW='{"parenting":1}' python3 $R/tools/variant_reset.py "$@"
