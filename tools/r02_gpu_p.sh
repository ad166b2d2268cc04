#!/bin/bash
# variants A/B + HBM traffic counters of one variant ($2 = library path or empty)
O=gpurun_out/${1:-r02x}; mkdir -p $O
export TMPDIR=/tmp
bash tools/r02_gpu_d.sh $1
for lib in "" $2; do
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && INFERCNV_HIP_LIB=${lib:+$OLDPWD/$lib} timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/$O/pmc_$grp -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/pmc_$grp.log 2>&1)
  f=$(find $O/pmc_$grp -name "*counter_collection.csv" | head -1)
  echo "lib=${lib:-default}" | tee -a $O/pmc.txt
  [ -n "$f" ] && python tools/summarize_pmc.py "$f" | grep -A3 "x16" | tee -a $O/pmc.txt
  rm -rf $O/pmc_$grp
done
done
exit 0
