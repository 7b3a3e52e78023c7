cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/tools/c5_mesh_bench.py lthick"
rm -rf /tmp/p1 /tmp/p2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d /tmp/p1 -- $CMD > /dev/null 2>&1
python $R/tools/pmc_sq.py "$(find /tmp/p1 -name '*results.db' | head -1)" $R/gpurun_out/r6g_lthick_pmc_sq.txt "$CMD" > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d /tmp/p2 -- $CMD > /dev/null 2>&1
python $R/tools/pmc_sq.py "$(find /tmp/p2 -name '*results.db' | head -1)" $R/gpurun_out/r6g_lthick_pmc_sq2.txt "$CMD" > /dev/null 2>&1
cat $R/gpurun_out/r6g_lthick_pmc_sq.txt $R/gpurun_out/r6g_lthick_pmc_sq2.txt | grep -v "^#" | head -60
