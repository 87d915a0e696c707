#!/bin/bash
# Round-5 experiment p: staging of the "from" rows by LDS-DMA with the first group's "to" rows loaded beside it
# (SF_MATCH_PRELOAD, k_match.hip) against the register copy (libsepfinder_ab.so = the same sources with -DSF_MATCH_PRELOAD=0).
set -o pipefail
out=gpurun_out/r05p; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_verify.py tests/test_gpu_fuzz.py tests/test_gpu_branches.py tests/test_gpu_pnp.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -20 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for rep in 1 2 3; do
  for lib in libsepfinder.so libsepfinder_ab.so; do
    for est in "" "--estimator pnp"; do
      echo "== $lib $est rep $rep" | tee -a $out/summary.txt
      SEPFINDER_LIB=$PWD/multi_robot_slam_separators_amd/$lib timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline $est > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
      python - $out/b.json <<'PY' | tee -a $out/summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("value %.3f M  ms_per_step %.4f  kernel %s  launch_ms %s" % (d["value"] / 1e6, d["ms_per_step"], r.get("kernel"), r.get("avg_launch_ms")))
PY
    done
  done
done
