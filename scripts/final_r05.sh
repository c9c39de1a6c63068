#!/usr/bin/env bash
# Round-end measurement run on the GPU box (one gpurun call): full GPU tests, the default bench line, rocprofv3 kernel stats + PMC passes
# of the inference bench and of the training step, SQ counters of the SDF kernel, and the other workloads' lines.
set -u
O=gpurun_out/r05final; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
bash scripts/profile_bench.sh r05 > $O/profile_bench.log 2>&1
SURF_PREC=bf16x3 TSDF_ARGS=576 bash scripts/pmc_time_sdf.sh > $O/sdf_sq_bf16x3.txt 2>&1
SURF_PREC=f16x2 TSDF_ARGS=576 bash scripts/pmc_time_sdf.sh > $O/sdf_sq_f16x2.txt 2>&1
bash scripts/profile_train.sh r05 pmc > $O/profile_train.log 2>&1
python bench.py --workload train > $O/train.json 2> $O/train.err
python bench.py --workload train --train-precision bf16 > $O/train_bf16.json 2> $O/train_bf16.err
python bench.py --gpus 1 --scenes 15 --cpu-seconds 0 --build 0 --train-step 0 --mesh-grid 0 --also "" > $O/scenes15.json 2> $O/scenes15.err
python bench.py --workload tnt --cpu-seconds 0 --build 0 --train-step 0 --mesh-grid 0 > $O/tnt.json 2> $O/tnt.err
python bench.py --split rays --check-split --gpus 1 > $O/split1.json 2> $O/split1.err
python bench.py --split rays --check-split --gpus 2 --backend gloo --one-gpu --steps 3 > $O/split2_onegpu.json 2> $O/split2.err
python scripts/time_scene_parts.py > $O/scene_parts.log 2>&1
ls -la $O
