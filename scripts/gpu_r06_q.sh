#!/usr/bin/env bash
# Re-entry check: the world-1 RCCL training test (rewritten on step-1 gradients) five times, then the whole GPU suite.
set -u
O=gpurun_out/r06q; mkdir -p $O
for i in 1 2 3 4 5; do
  python -m pytest tests/test_rccl_world1.py -m gpu -q -s -k training_step > $O/rccl$i.log 2>&1; echo "rc=$?" >> $O/rccl$i.log
  grep -h "gradient gaps\|passed\|failed\|rc=" $O/rccl$i.log | cut -c1-400
done
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
