# old-vs-new full reset per slot over the sizes of configs 4 / 5 (GPU box): bash tools/ab_sizes.sh name=lib.so ... ("cur" = the built library)
R=$GRAFT_REPO_ROOT
GE_ENV=SteinerTree-v0 GE_N=256 GE_M=1024 GE_B=4096 python3 $R/tools/ab_reset.py "$@"
GE_ENV=ShortestPath-v0 GE_N=130 GE_M=390 GE_B=4096 python3 $R/tools/ab_reset.py "$@"
GE_ENV=ShortestPath-v0 GE_N=200 GE_M=600 GE_B=4096 python3 $R/tools/ab_reset.py "$@"
GE_ENV=ShortestPath-v0 GE_N=320 GE_M=960 GE_B=1024 python3 $R/tools/ab_reset.py "$@"
GE_ENV=ShortestPath-v0 GE_N=400 GE_M=1200 GE_B=1024 python3 $R/tools/ab_reset.py "$@"
GE_ENV=ShortestPath-v0 GE_N=512 GE_M=1536 GE_B=512 python3 $R/tools/ab_reset.py "$@"
