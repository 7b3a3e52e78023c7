#!/bin/bash
# Developer tool (GPU box): rocprofv3 evidence for the V1 swept-volume sweep (tools/v1_bench.py) -> gpurun_out/<tag>_v1_*.txt
#   kernel trace + stats, then a separate SQ counter pass (never combined with other trace domains).  usage: tools/profile_v1.sh <tag>
set -u
TAG=${1:-rX}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/v1_bench.py --no-oracle"
finddb() { find "$1" -name "*results.db" | head -1; }
rm -rf /tmp/v1_kt /tmp/v1_sq /tmp/v1_sq2
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/v1_kt -- $CMD > $OUT/${TAG}_v1_bench_under_rocprof.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $CMD   (MI355X)"; python $REPO/tools/rocprof_summary.py "$(finddb /tmp/v1_kt)"; } > $OUT/${TAG}_v1_kernel_stats.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d /tmp/v1_sq -- $CMD > /dev/null 2>&1
python $REPO/tools/pmc_sq.py "$(finddb /tmp/v1_sq)" $OUT/${TAG}_v1_pmc_sq.txt "$CMD" > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d /tmp/v1_sq2 -- $CMD > /dev/null 2>&1
python $REPO/tools/pmc_sq.py "$(finddb /tmp/v1_sq2)" $OUT/${TAG}_v1_pmc_sq2.txt "$CMD" > /dev/null 2>&1
cd $REPO
python tools/v1_bench.py > $OUT/${TAG}_v1_bench.txt 2>&1
tail -3 $OUT/${TAG}_v1_bench.txt
