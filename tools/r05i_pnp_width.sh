#!/bin/bash
A="--steps 100 --warmup 10 --no-extras --no-cpu-baseline"
mkdir -p gpurun_out/r05i
run() { tag=$1; shift; envs=$1; shift; echo "== $tag: $envs $*"; env $envs timeout -k 10 400 python bench.py $A "$@" > gpurun_out/r05i/$tag.json 2> gpurun_out/r05i/$tag.err || { echo FAILED; tail -5 gpurun_out/r05i/$tag.err; return 1; }
python - gpurun_out/r05i/$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.3f M  ms_per_step %.4f  ok %s/%s" % (d["value"] / 1e6, d["ms_per_step"], d["check"]["decisions_matching_ground_truth"], d["check"]["of"]))
print("   kernels", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items() if v > 0})
PY
}
for nw in 4 2 1; do
run pnp_nw$nw SF_CHAIN_PNP_NW=$nw --estimator pnp || exit 1
done
for nw in 4 2 1; do
run pnp_ba_nw$nw SF_CHAIN_PNP_NW=$nw --estimator pnp --bundle-adjustment || exit 1
done
