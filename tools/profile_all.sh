#!/bin/bash
# Developer tool (GPU box): the whole evidence set of a round on the tree's sources -> gpurun_out/<tag>_*  (copy into profiles/).
#   usage: tools/profile_all.sh <tag>
TAG=${1:-rX}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1
for c in c3 v1 v1mesh v3mesh; do bash tools/profile_cfg.sh $TAG $c > gpurun_out/${TAG}_profile_cfg_$c.log 2>&1; done
cd $REPO
for n in 2 8; do
  ISDF_BENCH_SINGLE_DEVICE=1 ISDF_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) bench.py --gpus $n --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_${n}ranks_one_gpu.log 2>&1
  grep '^{' gpurun_out/${TAG}_bench_${n}ranks_one_gpu.log | tail -1 > gpurun_out/${TAG}_bench_${n}ranks_one_gpu.json
done
# the ESDF point sampler: measured HBM bytes (separate FETCH_SIZE / WRITE_SIZE passes) next to its algorithmic 88 B per point
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pe_f /tmp/pe_w
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pe_f -- python $REPO/tools/esdf_sample_bench.py 2e7 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pe_w -- python $REPO/tools/esdf_sample_bench.py 2e7 > /dev/null 2>&1
  python $REPO/tools/pmc_esdf.py "$(find /tmp/pe_f -name '*results.db' | head -1)" "$(find /tmp/pe_w -name '*results.db' | head -1)" > $REPO/gpurun_out/${TAG}_esdf_sample_pmc.txt 2>&1 )
python tools/esdf_sample_bench.py > gpurun_out/${TAG}_esdf_sample_bench.txt 2>&1
python tools/mesh_bench.py > gpurun_out/${TAG}_mesh_bench.txt 2>&1
python tools/c5_mesh_bench.py blob720 trefoil lthick lprism mybox icosa drone kuang box96 > gpurun_out/${TAG}_c5_mesh.txt 2>&1
python tools/multi_host_bench.py > gpurun_out/${TAG}_multi_host.txt 2>&1
python tools/batch_opt_bench.py 128 30 > gpurun_out/${TAG}_batch_opt_bench.txt 2>&1
python tools/shapes_bench.py > gpurun_out/${TAG}_shapes_bench.txt 2>&1
{ python tools/chunk_hist.py; python tools/chunk_hist_c3.py; } > gpurun_out/${TAG}_chunk_hist.txt 2>&1
python tools/v1_bench.py --mesh20 --no-oracle > gpurun_out/${TAG}_v1mesh20.txt 2>&1
ls -la gpurun_out | tail -40
