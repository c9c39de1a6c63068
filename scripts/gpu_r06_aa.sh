#!/usr/bin/env bash
set -u
O=gpurun_out/r06aa; mkdir -p $O
python scripts/step_phases.py > $O/phases.txt 2>&1; tail -8 $O/phases.txt
python scripts/step_phases.py fused > $O/phases_fused.txt 2>&1; tail -8 $O/phases_fused.txt
SURF_SIDE_STREAM=0 python scripts/step_phases.py > $O/phases_inorder.txt 2>&1; tail -8 $O/phases_inorder.txt
