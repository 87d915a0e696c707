#!/bin/bash
# Round-5 experiment s: k_match_split with the scan software-pipelined inside a wavefront (SF_MATCH_PIPE, k_match.hip)
# against the same sources built with -DSF_MATCH_PIPE=0 (libsepfinder_ab.so).  Parity tests first.
set -o pipefail
out=gpurun_out/r05s; mkdir -p $out
SF_FUSED=2 timeout -k 10 600 python -m pytest tests/test_gpu_verify.py tests/test_gpu_fuzz.py tests/test_gpu_branches.py -m gpu -x -q > $out/tests_split.log 2>&1 || { tail -20 $out/tests_split.log; exit 1; }
tail -1 $out/tests_split.log
timeout -k 10 600 python -m pytest tests/test_gpu_verify.py tests/test_gpu_pnp.py tests/test_gpu_step.py tests/test_gpu_ba.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -20 $out/tests.log; exit 1; }
tail -1 $out/tests.log
run() {  # lib, label, env..., -- args
  lib=$1; shift; label=$1; shift
  echo "== $lib $label" | tee -a $out/summary.txt
  env "SEPFINDER_LIB=$PWD/multi_robot_slam_separators_amd/$lib" "$@" > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
  python - $out/b.json <<'PY' | tee -a $out/summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("value %.3f M  ms_per_step %.4f  form %s  kernel %s  launch_ms %s" % (d["value"] / 1e6, d["ms_per_step"], d.get("verification_form"), r.get("kernel"), r.get("avg_launch_ms")))
PY
}
B="timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline"
for rep in 1 2; do
  for lib in libsepfinder.so libsepfinder_ab.so; do
    run $lib "pnp rep $rep" $B --estimator pnp
    run $lib "3d3d split rep $rep" SF_FUSED=2 $B
    run $lib "3d3d default rep $rep" $B
  done
done
for lib in libsepfinder.so libsepfinder_ab.so; do
  run $lib "cfg3 split" SF_FUSED=2 timeout -k 10 300 python bench.py --workload cfg3 --steps 40 --warmup 5 --no-extras --no-cpu-baseline
done
