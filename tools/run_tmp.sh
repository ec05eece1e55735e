#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2i
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --isolated-steps 0 > $O/under.jsonl 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
cd $R
python bench.py --no-cpu-baseline > $O/b_c2.jsonl 2> $O/b_c2.err && python - <<PY
import json
d=json.loads(open("$O/b_c2.jsonl").read().strip().splitlines()[-1])
print(round(d["value"],1), d["phase_ms"], d["host_step_ms"])
PY
