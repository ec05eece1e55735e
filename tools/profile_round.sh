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
python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('bench_default:', round(d['value'],1), 'generations/s; cpu_baseline.kind =', d['cpu_baseline']['kind'], '(reference binary present:', d['cpu_baseline'].get('reference_binary_present'), '); sustained_300 =', round(d['sustained_300']['generations_per_s'],1))" "$O/bench_default.jsonl"
cd /tmp
# 2. per-kernel time summary (kernel trace only) of the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o runc -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --sustained-steps 0 > "$O/under_rocprof.jsonl" 2> "$O/trace.err"
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one pass; no other trace domain)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -o runc -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --isolated-steps 0 --sustained-steps 0 > "$O/pmc_fetch.jsonl" 2> "$O/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -o runc -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --isolated-steps 0 --sustained-steps 0 > "$O/pmc_write.jsonl" 2> "$O/pmc_write.err"
# 3b. instruction counts of the sampling kernels (bench.py: sampling_kernels), streams serialised, their own passes
GEV_OVERLAP=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d "$O/pmc_insts" -o runc -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --isolated-steps 0 --sustained-steps 0 > "$O/pmc_insts.jsonl" 2> "$O/pmc_insts.err"
GEV_OVERLAP=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$O/pmc_cycles" -o runc -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --isolated-steps 0 --sustained-steps 0 > "$O/pmc_cycles.jsonl" 2> "$O/pmc_cycles.err"
# 4. every kernel ALONE on the GPU (streams serialised): the sampling / sparse / A-D kernel times quoted in DESIGN.md
GEV_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_serial" -o runc -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --isolated-steps 0 --sustained-steps 0 > "$O/serial.jsonl" 2> "$O/trace_serial.err"
# 5. BASELINE config 4's per-GPU shard (125k individuals x 11 chromosomes x 227k SNPs = 156 GB resident): bench line + kernel stats, overlapped and serialised
cd "$R" && python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --nchr 11 --n-ind 125000 --n-loci 227000 > "$O/shard11_bench.jsonl" 2> "$O/shard11_bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_shard11" -o runc -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --isolated-steps 0 --nchr 11 --n-ind 125000 --n-loci 227000 > "$O/shard11_under_rocprof.jsonl" 2> "$O/trace_shard11.err"
GEV_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_shard11_serial" -o runc -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --isolated-steps 0 --nchr 11 --n-ind 125000 --n-loci 227000 > "$O/shard11_serial.jsonl" 2> "$O/trace_shard11_serial.err"
# 6. the drop-in PROGRAM (reference CLI bound to the library) next to the unmodified reference, 100k individuals
cd "$R" && python3 tools/cli_timing.py --exe both --gens 3 > "$O/cli_dropin.json" 2> "$O/cli_dropin.err"
python3 tools/cli_timing.py --exe gpu --gens 3 --assortative > "$O/cli_dropin_assortative.json" 2> "$O/cli_dropin_assortative.err" || true
# 7. the multi-rank launcher on this one GPU (gloo rehearsal of the N>1 bench: both ranks share device 0)
GEV_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 10 --warmup 3 > "$O/bench_n2_one_gpu.jsonl" 2> "$O/bench_n2_one_gpu.err"
GEV_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 10 --warmup 3 --migration-rate 0.01 > "$O/bench_n2_migration_one_gpu.jsonl" 2> "$O/bench_n2_migration_one_gpu.err"
ls "$O"
