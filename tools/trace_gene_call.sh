#!/bin/bash
O=$PWD/gpurun_out/dbg; mkdir -p $O; export TMPDIR=/tmp; REPO=$PWD
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/_t -o tr -- python $REPO/tools/trace_gene_call.py > $O/trace_gene.log 2>&1)
f=$(find $O/_t -name "*kernel_trace.csv" | head -1)
python $REPO/tools/trace_gene_call.py $f
rm -rf $O/_t
