#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (one MI355X).  usage: tools/profile_round.sh <tag>
# Writes under gpurun_out/prof_<tag>/; copy the summaries into profiles/ afterwards (profiles/README.md).
set -eo pipefail
tag=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# 1. the default bench line (with the CPU baseline), outside the profiler, FIRST: the same state a fresh box is in when the
#    round-end driver runs it (sustained HBM load warms the board; later runs of one call are a few % slower)
cd "$R" && python3 bench.py > "$O/bench_default.jsonl" 2> "$O/bench_default.err"
cd /tmp
# 2. per-kernel time summary (kernel trace only) of the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o runc -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$O/under_rocprof.jsonl" 2> "$O/trace.err"
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one pass; no other trace domain)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -o runc -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --isolated-steps 0 > "$O/pmc_fetch.jsonl" 2> "$O/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -o runc -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --isolated-steps 0 > "$O/pmc_write.jsonl" 2> "$O/pmc_write.err"
