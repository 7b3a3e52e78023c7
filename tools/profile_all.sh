#!/bin/bash
# Developer tool (GPU box): the whole evidence set of a round on the tree's sources -> gpurun_out/<tag>_*  (copy into profiles/).
#   usage: tools/profile_all.sh <tag>
TAG=${1:-rX}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1
for c in c3 v1 v1mesh; do bash tools/profile_cfg.sh $TAG $c > gpurun_out/${TAG}_profile_cfg_$c.log 2>&1; done
cd $REPO
for n in 2 8; do
  ISDF_BENCH_SINGLE_DEVICE=1 ISDF_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) bench.py --gpus $n --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_${n}ranks_one_gpu.log 2>&1
  grep '^{' gpurun_out/${TAG}_bench_${n}ranks_one_gpu.log | tail -1 > gpurun_out/${TAG}_bench_${n}ranks_one_gpu.json
done
{ python tools/chunk_hist.py; python tools/chunk_hist_c3.py; } > gpurun_out/${TAG}_chunk_hist.txt 2>&1
python tools/v1_bench.py --mesh20 --no-oracle > gpurun_out/${TAG}_v1mesh20.txt 2>&1
ls -la gpurun_out | tail -40
