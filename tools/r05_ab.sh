#!/bin/bash
# Round-5 A/B driver: runs bench.py once per environment setting, one JSON line each into gpurun_out/$1/.
# usage: tools/r05_ab.sh <outdir-name> <bench args...> -- "ENV1=a ENV2=b" "ENV1=c" ...
out=gpurun_out/$1; shift
mkdir -p "$out"
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
i=0
for envs in "$@"; do
  tag=$(echo "$envs" | tr ' =' '_-' | tr -cd 'A-Za-z0-9_-')
  [ -z "$tag" ] && tag=default
  echo "== $envs" | tee -a "$out/summary.txt"
  env $envs timeout -k 10 300 python bench.py "${args[@]}" > "$out/$tag.json" 2> "$out/$tag.err" || { echo "FAILED rc=$?" | tee -a "$out/summary.txt"; tail -5 "$out/$tag.err" | tee -a "$out/summary.txt"; break; }
  python - "$out/$tag.json" <<'PY' | tee -a "$out/summary.txt"
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("value %.3f M  ms_per_step %.4f  kernel %s  launch_ms %s  decisions_ok %s" % (d["value"] / 1e6, d["ms_per_step"], r.get("kernel"), r.get("avg_launch_ms"), d.get("checks", {}).get("decisions_equal_ground_truth", d.get("decisions_equal_ground_truth"))))
PY
  i=$((i+1))
done
