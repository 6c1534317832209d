# graph-kernel A/B at mid sizes (GPU box): bash tools/ab_sizes2.sh name=lib.so ...
R=$GRAFT_REPO_ROOT
GE_ENV=SteinerTree-v0 GE_N=256 GE_M=1024 GE_B=4096 python3 $R/tools/ab_reset.py "$@"
GE_ENV=ShortestPath-v0 GE_N=200 GE_M=600 GE_B=4096 python3 $R/tools/ab_reset.py "$@"
GE_ENV=ShortestPath-v0 GE_N=130 GE_M=390 GE_B=4096 python3 $R/tools/ab_reset.py "$@"
GE_ENV=ShortestPath-v0 GE_N=100 GE_M=300 GE_B=8192 python3 $R/tools/ab_reset.py "$@"
