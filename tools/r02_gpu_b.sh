#!/bin/bash
# round 2, GPU call B: x16 v2 (two-level histogram, adjacent blocks/windows, 2-stage median pipeline)
set -u
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"; }
for i in 1 2; do timeout 120 $B 2>&1 | tail -1 | one x16_default | tee -a $O/bench.txt; done
ICV_NO_X16=1 timeout 120 $B 2>&1 | tail -1 | one ws_prev | tee -a $O/bench.txt
for lib in variants/libicv_addr1.so variants/libicv_a1_nocoarse.so variants/libicv_a1_noatom.so; do
  INFERCNV_HIP_LIB=$PWD/$lib timeout 120 $B 2>&1 | tail -1 | one $lib | tee -a $O/bench.txt
done
INFERCNV_HIP_LIB=$PWD/variants/libicv_prof.so ICV_PHASE_PROFILE=1 timeout 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "x16 profile" | tee $O/phase.txt
# instruction counts of the smoothing kernel (counters in their own run, kernel-trace only)
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/$O/pmc_$name -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/pmc_$name.log 2>&1)
  f=$(find $O/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/summarize_pmc.py "$f" | tee -a $O/pmc.txt
done
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete 2>/dev/null
