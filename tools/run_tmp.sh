#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 $O/pytest.log
