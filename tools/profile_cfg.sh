#!/bin/bash
# Developer tool (GPU box): counters of ONE configuration's dominant kernel(s) -> gpurun_out/<tag>_pmc_counters_<cfg>.json (hashed),
# plus the kernel-trace summary.  Two PMC passes, never combined with other trace domains.
#   usage: tools/profile_cfg.sh <tag> v1|v1mesh|v3mesh|c3
set -u
TAG=${1:-rX}; CFG=${2:-v1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$CFG" = "v1" ]; then
  CMD="python $REPO/tools/v1_bench.py --no-oracle --only-cone"
  export PMC_KERNELS="swept_scan_kernel,swept_descent_kernel,swept_search_kernel" PMC_HASH=v1
elif [ "$CFG" = "v1mesh" ]; then
  CMD="python $REPO/tools/v1_bench.py --no-oracle --mesh720"
  export PMC_KERNELS="swept_scan_kernel,swept_descent_kernel" PMC_HASH=v1
elif [ "$CFG" = "v3mesh" ]; then
  CMD="python $REPO/tools/mesh720_only.py"
  export PMC_KERNELS="mesh_exact_kernel" PMC_HASH=sweep
else
  CMD="python $REPO/tools/c3_only.py 12"
  export PMC_KERNELS="sweep_kernel" PMC_HASH=sweep
fi
finddb() { find "$1" -name "*results.db" | head -1; }
rm -rf /tmp/pc_kt /tmp/pc_a /tmp/pc_b
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pc_kt -- $CMD > $OUT/${TAG}_${CFG}_under_rocprof.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $CMD   (MI355X)"; python $REPO/tools/rocprof_summary.py "$(finddb /tmp/pc_kt)"; } > $OUT/${TAG}_${CFG}_kernel_stats.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d /tmp/pc_a -- $CMD > /dev/null 2>&1
python $REPO/tools/pmc_sq.py "$(finddb /tmp/pc_a)" $OUT/${TAG}_${CFG}_pmc_sq.txt "$CMD" > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES -d /tmp/pc_b -- $CMD > /dev/null 2>&1
python $REPO/tools/pmc_sq.py "$(finddb /tmp/pc_b)" $OUT/${TAG}_${CFG}_pmc_sq2.txt "$CMD" > /dev/null 2>&1
# lane utilisation of the VALU instructions = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) (both from the SAME pass)
rm -rf /tmp/pc_c
timeout 600 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU -d /tmp/pc_c -- $CMD > /dev/null 2>&1
python $REPO/tools/pmc_sq.py "$(finddb /tmp/pc_c)" $OUT/${TAG}_${CFG}_pmc_sq3.txt "$CMD" > /dev/null 2>&1
python $REPO/tools/pmc_collect.py $OUT/${TAG}_pmc_counters_${CFG}.json "$(finddb /tmp/pc_kt)" "$(finddb /tmp/pc_a)" "$(finddb /tmp/pc_b)" "$(finddb /tmp/pc_c)" > $OUT/${TAG}_${CFG}_pmc_collect.log 2>&1
cat $OUT/${TAG}_${CFG}_kernel_stats.txt | head -12
