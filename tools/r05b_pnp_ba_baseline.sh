set -e
A="--steps 100 --warmup 10 --no-extras --no-cpu-baseline"
run() { tag=$1; shift; echo "== $tag: $*"; timeout -k 10 400 python bench.py $A "$@" > gpurun_out/r05b/$tag.json 2> gpurun_out/r05b/$tag.err || { echo FAILED; tail -5 gpurun_out/r05b/$tag.err; return 1; }
python - gpurun_out/r05b/$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.3f M  ms_per_step %.4f  form %s  check %s" % (d["value"] / 1e6, d["ms_per_step"], d.get("verification_form"), d.get("check")))
print("   kernels", {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items() if v > 0})
PY
}
mkdir -p gpurun_out/r05b
run pnp --estimator pnp
run pnp_ba --estimator pnp --bundle-adjustment
run 3d3d_ba --bundle-adjustment
run pnp_ba_bidir --estimator pnp --bundle-adjustment --forward-est-only 0
