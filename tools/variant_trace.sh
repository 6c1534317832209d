# kernel-trace medians of the bench loop for library variants.  usage (GPU box): bash tools/variant_trace.sh <tag> "<flags>" "<flags>" ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for flags in "$@"; do
  i=$((i+1))
  echo "=== variant $i: [$flags]" >> $O/medians.txt
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/t$i --output-format csv -- python3 $R/tools/variant_bench.py "$flags" 150 > $O/v$i.out 2> $O/v$i.log || echo "variant $i failed"
  tail -1 $O/v$i.out >> $O/medians.txt
  python3 $R/tools/kstats.py $O/t$i >> $O/medians.txt
  rm -rf $O/t$i
done
cat $O/medians.txt
