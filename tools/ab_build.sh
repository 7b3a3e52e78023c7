#!/bin/bash
# Developer tool (GPU box): A/B of two builds of the library on one box - the previous build saved as
# implicit-sdf-planner_amd/lib/libisdf_accel_prev.so (git-ignored) against the tree's: C3 step time with a hash of the outputs
# (bitwise comparison of the two builds), the bench line, the per-wave phase timing.
P=$PWD/implicit-sdf-planner_amd/lib/libisdf_accel_prev.so
for i in 1 2; do
ISDF_ACCEL_LIB=$P python tools/c3_only.py 30 | sed 's/^/prev /'
python tools/c3_only.py 30 | sed 's/^/new  /'
done
for i in 1 2; do
ISDF_ACCEL_LIB=$P python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('prev', d['value'], d['ms_per_step'], d.get('value_steady'), d['roofline']['achieved'])"
python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('new ', d['value'], d['ms_per_step'], d.get('value_steady'), d['roofline']['achieved'])"
done
python tools/timing.py 40 box 2>&1 | grep "drain\|exact rounds\|rows  "
