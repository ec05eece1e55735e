set -x
mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -x -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -3 gpurun_out/r2b/pytest.log
GEV_OVERLAP=0 python bench.py --steps 10 --no-cpu-baseline > gpurun_out/r2b/b_serial.jsonl 2> gpurun_out/r2b/b_serial.err
GEV_OVERLAP=0 GEV_SAMPLE_BATCHED=0 python bench.py --steps 10 --no-cpu-baseline > gpurun_out/r2b/b_serial_old.jsonl 2> gpurun_out/r2b/b_serial_old.err
python bench.py --no-cpu-baseline > gpurun_out/r2b/b_def.jsonl 2> gpurun_out/r2b/b_def.err
GEV_SAMPLE_BATCHED=0 python bench.py --no-cpu-baseline > gpurun_out/r2b/b_def_old.jsonl 2> gpurun_out/r2b/b_def_old.err
cd /tmp && export TMPDIR=/tmp && GEV_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2b/prof_serial -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --isolated-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r2b/prof_serial.jsonl 2> $GRAFT_REPO_ROOT/gpurun_out/r2b/prof_serial.err
ls -R $GRAFT_REPO_ROOT/gpurun_out/r2b/prof_serial | head
