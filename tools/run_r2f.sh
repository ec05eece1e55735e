set -x
O=gpurun_out/r2f; mkdir -p $O
python tests/rccl_single_rank.py > $O/rccl.log 2>&1; tail -5 $O/rccl.log
python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
grep "gev_rank_f64\|text fields differ" $O/pytest.log
python bench.py --no-cpu-baseline > $O/b_def.jsonl 2> $O/b_def.err
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --nchr 11 --n-ind 125000 --n-loci 227000 > $O/b_shard11.jsonl 2> $O/b_shard11.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_shard11 -o p -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --isolated-steps 0 --nchr 11 --n-ind 125000 --n-loci 227000 > $R/$O/prof_shard11.jsonl 2> $R/$O/prof_shard11.err
