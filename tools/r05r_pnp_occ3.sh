#!/bin/bash
# Round-5 experiment r: the four-wavefront PnP chain compiled for 3 wavefronts per SIMD (the build that failed
# test_pnp_refinement_rounds) -- does the failure follow the SGPR spills to VGPR lanes?  libsepfinder_x3.so = that build,
# libsepfinder_x3m.so = the same with -mllvm -amdgpu-spill-sgpr-to-vgpr=0 (scalar spills go to scratch memory).
out=gpurun_out/r05r; mkdir -p $out
for v in "" _x3 _x3m "$@"; do
  echo "== libsepfinder$v.so" | tee -a $out/summary.txt
  SF_CHAIN_PNP_NW=4 SEPFINDER_LIB=$PWD/multi_robot_slam_separators_amd/libsepfinder$v.so timeout -k 10 300 python -m pytest tests/test_gpu_pnp.py -m gpu -q -k "refinement_rounds or parity" > $out/test$v.log 2>&1
  echo "rc=$?" | tee -a $out/summary.txt
  grep -E "passed|failed|AssertionError|assert " $out/test$v.log | head -8 | tee -a $out/summary.txt
done
