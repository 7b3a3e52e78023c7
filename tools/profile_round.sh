#!/bin/bash
# Developer tool (GPU box): the rocprofv3 evidence of one round -> gpurun_out/<tag>_*.txt|json (copy into profiles/).
#   kernel trace + stats of the default bench, then separate PMC passes (HBM bytes; SQ issue counters; instruction mix) -
#   never combined with other trace domains.   usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-rX}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline --lbfgs-iters 0 --no-extra-configs --no-host-api"
BENCHS="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --lbfgs-iters 0 --no-extra-configs --no-host-api"
finddb() { find "$1" -name "*results.db" | head -1; }
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w /tmp/prof_sq /tmp/prof_sq2 /tmp/prof_sq3
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- $BENCH > $OUT/${TAG}_bench_under_rocprof.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $BENCH   (MI355X)"; python $REPO/tools/rocprof_summary.py "$(finddb /tmp/prof_kt)"; } > $OUT/${TAG}_kernel_stats.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -- $BENCHS > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -- $BENCHS > /dev/null 2>&1
python $REPO/tools/pmc_traffic.py "$(finddb /tmp/prof_f)" "$(finddb /tmp/prof_w)" $OUT/${TAG}_hbm_traffic.json $OUT/${TAG}_pmc_hbm.txt > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d /tmp/prof_sq -- $BENCHS > /dev/null 2>&1
python $REPO/tools/pmc_sq.py "$(finddb /tmp/prof_sq)" $OUT/${TAG}_pmc_sq.txt "$BENCHS" > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 GRBM_GUI_ACTIVE -d /tmp/prof_sq2 -- $BENCHS > /dev/null 2>&1
python $REPO/tools/pmc_sq.py "$(finddb /tmp/prof_sq2)" $OUT/${TAG}_pmc_sq2.txt "$BENCHS" > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU -d /tmp/prof_sq3 -- $BENCHS > /dev/null 2>&1
python $REPO/tools/pmc_sq.py "$(finddb /tmp/prof_sq3)" $OUT/${TAG}_pmc_sq3.txt "$BENCHS" > /dev/null 2>&1
# one JSON with the dominant kernel's counters, stamped with the kernel sources' hash: bench.py reads profiles/pmc_counters.json
python $REPO/tools/pmc_collect.py $OUT/${TAG}_pmc_counters.json "$(finddb /tmp/prof_kt)" "$(finddb /tmp/prof_f)" "$(finddb /tmp/prof_w)" "$(finddb /tmp/prof_sq)" "$(finddb /tmp/prof_sq2)" "$(finddb /tmp/prof_sq3)" > $OUT/${TAG}_pmc_collect.log 2>&1
cd $REPO
cp $OUT/${TAG}_pmc_counters.json $REPO/profiles/pmc_counters.json     # (on the box: so that the bench line below carries them)
python bench.py > $OUT/${TAG}_bench.log 2>&1
tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench.json
python tools/configs_bench.py > $OUT/${TAG}_configs.txt 2>&1
ls -la $OUT | tail -12
