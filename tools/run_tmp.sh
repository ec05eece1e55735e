#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $O
timeout -k 10 300 ./tools/stitch_bench > $O/stitch_bench.txt 2>&1; cat $O/stitch_bench.txt
timeout -k 10 300 ./tools/stitch_bench 125000 227000 > $O/stitch_bench_s11.txt 2>&1; cat $O/stitch_bench_s11.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
p=d["phase_ms"]; r=d["roofline"]
print("$1", round(d["value"],1), "gen/s  samp %.2f sparse %.2f stitch %.2f wall %.2f mate %.2f | frac %.3f iso %s alg %.2f GB" % (p["sampling"],p["sparse_lists_and_cv_planes"],p["dense_stitch"],p["gev_reproduce_wall"],p["host_mating"], r["frac"], r["isolated_kernel_ms"], r["algorithmic_bytes_per_launch"]/1e9), flush=True)
PY
}
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/b_c2.jsonl 2> $O/b_c2.err && show $O/b_c2.jsonl
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 4 --nchr 11 --n-ind 125000 --n-loci 227000 > $O/b_s11.jsonl 2> $O/b_s11.err && show $O/b_s11.jsonl
