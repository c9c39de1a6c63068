#!/usr/bin/env bash
# Round 6, first GPU run: the forced world-1 RCCL group (tests + bench lines), the convention hedges, the full GPU suite, the default line.
set -u
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests/test_rccl_world1.py -x -q > $O/rccl.log 2>&1; echo "rc=$?" >> $O/rccl.log; tail -15 $O/rccl.log
python -m pytest tests -m gpu -q -x --deselect tests/test_rccl_world1.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; echo; tail -5 $O/bench.err
python bench.py --workload train > $O/train.json 2> $O/train.err; tail -c 600 $O/train.json; echo; tail -5 $O/train.err
