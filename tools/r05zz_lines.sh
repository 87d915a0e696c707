#!/bin/bash
# The round's bench lines: driver command, default, PnP, PnP + BA (as shipped), 3D-3D + BA, strict, cfg3.
mkdir -p gpurun_out/r05zz
run() { tag=$1; shift; echo "== $tag: python bench.py $*"; timeout -k 10 600 python bench.py "$@" > gpurun_out/r05zz/$tag.json 2> gpurun_out/r05zz/$tag.err || { echo FAILED; tail -5 gpurun_out/r05zz/$tag.err; return 1; }
python - gpurun_out/r05zz/$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.3f M  ms_per_step %.4f  roofline %s frac %.4f  cpu %s  parity %s" % (d["value"] / 1e6, d["ms_per_step"], r["kernel"], r["frac"], d.get("cpu_baseline", {}).get("value"), d.get("parity_in_this_run")))
for k in ("value_fixed_iterations", "value_full_length_filter", "value_strict", "value_one_synchronisation_per_step"):
    if k in d: print("   ", k, "%.3f M" % (d[k] / 1e6))
if "roofline_strict" in d: print("    roofline_strict", d["roofline_strict"])
PY
}
run driver_command --gpus 1 --steps 20 --warmup 5 || exit 1
run default || exit 1
run pnp --estimator pnp || exit 1
run pnp_ba --estimator pnp --bundle-adjustment || exit 1
run 3d3d_ba --bundle-adjustment --no-extras || exit 1
run strict --strict --no-extras || exit 1
run cfg3 --workload cfg3 || exit 1
