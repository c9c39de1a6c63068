#!/usr/bin/env bash
# Round-end measurement run on the GPU box (one gpurun call): full GPU tests, smoke, the default bench line (with other_configs and the forced
# world-1 RCCL group), rocprofv3 kernel stats + PMC passes of the inference bench and of the training step, SQ counters of the SDF kernel, the
# training lines of both policies, the scene parts.
set -u
O=gpurun_out/r06final; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash scripts/profile_bench.sh r06 > $O/profile_bench.log 2>&1
python scripts/summarize_profile.py r06 > /dev/null 2>&1      # profiles/r06_bench_pmc.csv of THESE sources, so that the line below carries roofline.traffic
python bench.py > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo; wc -l $O/bench.json
SURF_PREC=bf16x3 TSDF_ARGS=576 bash scripts/pmc_time_sdf.sh > $O/sdf_sq_bf16x3.txt 2>&1
bash scripts/profile_train.sh r06 pmc > $O/profile_train.log 2>&1
for i in 1 2 3; do
  python bench.py --workload train > $O/train$i.json 2> $O/train$i.err
  python bench.py --workload train --train-precision bf16 > $O/train_bf16_$i.json 2> $O/train_bf16_$i.err
done
python bench.py --workload train --force-group 0 --cpu-seconds 0 > $O/train_nogroup.json 2> $O/train_nogroup.err
python scripts/step_phases.py > $O/phases.txt 2>&1
SURF_SIDE_STREAM=0 python scripts/step_phases.py > $O/phases_inorder.txt 2>&1
python scripts/time_scene_parts.py > $O/scene_parts.log 2>&1
ls -la $O
