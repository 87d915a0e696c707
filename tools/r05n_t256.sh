#!/bin/bash
A="--steps 60 --warmup 10 --no-extras --no-cpu-baseline --strict"
mkdir -p gpurun_out/r05n
run() { tag=$1; shift; envs=$1; shift; echo "== $tag: $envs $*"; env $envs timeout -k 10 400 python bench.py $A "$@" > gpurun_out/r05n/$tag.json 2> gpurun_out/r05n/$tag.err || { echo FAILED; tail -5 gpurun_out/r05n/$tag.err; return 1; }
python - gpurun_out/r05n/$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.3f M  ms_per_step %.4f  ok %s/%s  filter: %.3f ms in flight = %.0f TF (%.3f)" % (d["value"] / 1e6, d["ms_per_step"], d["check"]["decisions_matching_ground_truth"], d["check"]["of"], r["avg_launch_ms"], r["achieved"], r["frac"]))
PY
}
run t128 "SF_NN_T256=0" || exit 1
run t256 "SF_NN_T256=1" || exit 1
run t128_lanes1 "SF_NN_T256=0 SF_STEP_LANES=1" || exit 1
run t256_lanes1 "SF_NN_T256=1 SF_STEP_LANES=1" || exit 1
