#!/bin/bash
# A/B runs of the bench under different scheduling knobs (one GPU box call); usage: tools/variants.sh <outfile> [bench args...]
out=$1; shift
cd ${GRAFT_REPO_ROOT:-.}
EXTRA="$*"
run() { echo "== $*" >> $out; args=""; envs=""; for a in "$@"; do case $a in --*|[0-9]*) args="$args $a";; *) envs="$envs $a";; esac; done
  env $envs python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --sustained-steps 0 $EXTRA $args 2>> $out.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(round(d['value'],1), 'gen/s', round(d['ms_per_step'],3), 'ms  stitch', round(d['phase_ms']['dense_stitch'],3), 'iso', round(r['isolated_kernel_ms'] or 0,3), 'segs', int(r['segments_written_per_launch']), 'frac', round(r['frac'],3), 'isofrac', round(r['isolated_frac'] or 0,3), 'sampling', round(d['phase_ms']['sampling'],3), 'sparse', round(d['phase_ms']['sparse_lists_and_cv_planes'],3), d['phase_ms']['host_ms_inside_calls'])" >> $out; }
if [ -n "$VARIANTS" ]; then
  IFS=';' read -ra V <<< "$VARIANTS"
  for v in "${V[@]}"; do run $v; done
else
run A=1
run A=1 --no-host-overlap
run GEV_STITCH_START=0
run A=1 --no-intervals
fi
cat $out
