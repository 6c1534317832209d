# SQ / TCC counter passes (separate rocprofv3 --pmc runs, kernel-trace only) + a kernel-trace --stats run over one command.
# PMC_EXTRA: arguments for the counter passes only (bench.py --serial-shards: a launch's counters are device-wide while it runs, so the
# shards of a sharded engine are run one after the other there; the kernel trace is of the command as given)
# usage (GPU box): bash tools/pmc_sq_passes.sh <tag> <python script and args ...>     -> gpurun_out/<tag>/{summary.json,kernel_medians.txt,kernel_stats.csv}
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace -d $O/p$i --output-format csv -- python3 $R/"$@" $PMC_EXTRA > $O/p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/"$@" > $O/trace.out 2> $O/trace.log || echo "trace failed"
cp $(ls $O/trace/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 $R/tools/kstats.py $O/trace > $O/kernel_medians.txt
python3 $R/tools/gaps.py $O/trace > $O/gaps.txt || true
python3 $R/tools/pmc_sq_summary.py $O > $O/summary.json
cat $O/kernel_medians.txt
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/trace
