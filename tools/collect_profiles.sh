#!/bin/bash
# Copy what is to be judged from the evidence run (tools/round_end.sh -> gpurun_out/${ROUND}final/) into profiles/.
#   ROUND=r06 tools/collect_profiles.sh
set -eu
ROUND=${ROUND:-r06}
O=gpurun_out/${ROUND}final; P=profiles
cp $O/bench_n1.json $P/${ROUND}_bench_n1.json
tail -1 $O/bench_n1.err > $P/${ROUND}_bench_n1_summary_line.txt
cp $O/bench_dense_w100_traced.json $P/${ROUND}_bench_traced_dense_w100.json
cp $O/bench_dry_run_2ranks_one_gpu.json $P/${ROUND}_bench_dry_run_2ranks_one_gpu.json
cp $O/bench_dry_run_8ranks_one_gpu.json $P/${ROUND}_bench_dry_run_8ranks_one_gpu.json
cp $O/kernel_stats_dense_w100.csv $P/${ROUND}_rocprofv3_kernel_stats.csv
cp $O/kernel_stats_csr_w250.csv $P/${ROUND}_rocprofv3_kernel_stats_csr_w250.csv
cp $O/kernel_stats_csr_w100.csv $P/${ROUND}_rocprofv3_kernel_stats_csr_w100.csv
cp $O/kernel_stats_config5.csv $P/${ROUND}_config5_rocprofv3_kernel_stats.csv
cp $O/kernel_stats_gene_values_scores.csv $P/${ROUND}_rocprofv3_kernel_stats_gene_values_scores.csv
cp $O/bench_gene_values_scores_traced.json $P/${ROUND}_bench_traced_gene_values_scores.json
cp $O/pmc_summary.txt $P/${ROUND}_rocprofv3_pmc_summary.txt
cp $O/csr_density_lines.txt $P/${ROUND}_csr_density_lines.txt
cp $O/csr_means_density.txt $P/${ROUND}_csr_means_density.txt
cp $O/chain_blocks_times.txt $P/${ROUND}_chain_blocks_times.txt
cp $O/hbm_write.txt $P/${ROUND}_hbm_write.txt
[ -f $O/gene_kernel.txt ] && cp $O/gene_kernel.txt $P/${ROUND}_gene_kernel.txt
[ -f $O/host_issue_after_e2e.txt ] && cp $O/host_issue_after_e2e.txt $P/${ROUND}_host_issue_after_e2e.txt
cp $O/bench_1m_leg.json $P/${ROUND}_bench_1m_leg.json
python - $O/clocks_1m_leg.txt > $P/${ROUND}_clocks_1m_leg.txt <<'PY'
import re, sys
lines = [l for l in open(sys.argv[1]) if l.strip()]
print(f"rocm-smi --showclocks --showpower every 0.25 s over bench.py --steps 300 --extra config3_cells_on_one_gpu ({len(lines)} samples)")
for l in lines[::max(1, len(lines) // 60)]:
    print(l.strip()[:200])
PY
cp $O/host_pack.txt $P/${ROUND}_host_pack.txt
cp $O/fuzz_gpu.txt $P/${ROUND}_fuzz_gpu.txt
cp $O/soak_public_api.txt $P/${ROUND}_soak_public_api.txt
(cat $O/box.txt; cat $O/pytest.txt; cat $O/smoke.txt; cat $O/bench_n1.time) > $P/${ROUND}_pytest_gpu.txt
ls -la $P | grep ${ROUND}_ | wc -l
