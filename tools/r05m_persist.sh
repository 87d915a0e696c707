#!/bin/bash
A="--steps 200 --warmup 20 --no-extras --no-cpu-baseline"
mkdir -p gpurun_out/r05m
run() { tag=$1; shift; envs=$1; shift; echo "== $tag: $envs $*"; env $envs timeout -k 10 400 python bench.py $A "$@" > gpurun_out/r05m/$tag.json 2> gpurun_out/r05m/$tag.err || { echo FAILED; tail -5 gpurun_out/r05m/$tag.err; return 1; }
python - gpurun_out/r05m/$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.3f M  ms_per_step %.4f  ok %s/%s" % (d["value"] / 1e6, d["ms_per_step"], d["check"]["decisions_matching_ground_truth"], d["check"]["of"]))
print("   kernels", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items() if v > 0})
PY
}
run fused "SF_X=0" || exit 1
run split "SF_STEP_SPLIT=1" || exit 1
run split_persist "SF_STEP_SPLIT=1 SF_MATCH_PERSIST=1" || exit 1
run pnp "SF_X=0" --estimator pnp || exit 1
run pnp_persist "SF_MATCH_PERSIST=1" --estimator pnp || exit 1
