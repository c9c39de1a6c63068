#!/usr/bin/env bash
set -u
O=gpurun_out/r06al; mkdir -p $O
python -m pytest tests/test_side_streams.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log | cut -c1-300
