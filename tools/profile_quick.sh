#!/bin/bash
# The kernel-trace, HBM-traffic and two-rank steps of tools/profile_round.sh (about 4 GPU-minutes).  usage: tools/profile_quick.sh <tag>
set -eo pipefail
tag=${1:-quick}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/prof_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o runc -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --sustained-steps 0 > "$O/under_rocprof.jsonl" 2> "$O/trace.err"
echo trace done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -o runc -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --isolated-steps 0 --sustained-steps 0 > "$O/pmc_fetch.jsonl" 2> "$O/pmc_fetch.err"
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -o runc -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --isolated-steps 0 --sustained-steps 0 > "$O/pmc_write.jsonl" 2> "$O/pmc_write.err"
echo write done
cd $R
GEV_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > "$O/bench_n2_one_gpu.jsonl" 2> "$O/bench_n2_one_gpu.err"
GEV_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --migration-rate 0.01 > "$O/bench_n2_migration_one_gpu.jsonl" 2> "$O/bench_n2_migration_one_gpu.err"
echo n2 done
find $O -name "*.csv" | head -20
