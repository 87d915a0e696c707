#!/bin/bash
# Round-5 experiment t: shapes of the pipelined k_match_split (SF_SPLIT_MATCH_VARIANT: 0 = 4 tiles / 3 workgroups per CU by
# registers, 1 = 2 tiles / 4, 2 = 2 tiles / 3, 3 = 4 tiles / 2) on the 3D-3D split step, the PnP step and the cfg3 split step.
set -o pipefail
out=gpurun_out/r05t; mkdir -p $out
B="timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline"
for v in 0 1 2 3; do
  for what in "3d3d SF_FUSED=2" "pnp SF_NOP=1 --estimator pnp"; do
    set -- $what; label=$1; envs=$2; shift 2
    echo "== variant $v $label" | tee -a $out/summary.txt
    env SF_SPLIT_MATCH_VARIANT=$v $envs $B "$@" > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
    python - $out/b.json <<'PY' | tee -a $out/summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("value %.3f M  ms_per_step %.4f  kernel %s  launch_ms %s" % (d["value"] / 1e6, d["ms_per_step"], r.get("kernel"), r.get("avg_launch_ms")))
PY
  done
done
for v in 0 2 3; do
  echo "== variant $v cfg3 split" | tee -a $out/summary.txt
  env SF_SPLIT_MATCH_VARIANT=$v SF_FUSED=2 timeout -k 10 300 python bench.py --workload cfg3 --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
  python -c "
import json
d = json.loads(open('$out/b.json').read().strip().splitlines()[-1])
print('value %.3f M  ms_per_step %.4f' % (d['value'] / 1e6, d['ms_per_step']))" | tee -a $out/summary.txt
done
