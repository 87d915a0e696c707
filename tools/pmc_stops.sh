#!/bin/bash
# Vector-instruction budget of the motion-estimation chains per phase: the chain kernel's counters with the estimator cut
# short behind phase N (SF_RANSAC_STOP, diagnostics only).  3D-3D (k_ransac.hip): 1 = gather + PCA, 2 = hypothesis rounds,
# 4 = selection, 5 = refinement, 0 = whole chain.  PnP (k_pnp.hip): 1 = gather, 2 = RANSAC, 3 = mask, 4 = solve + refinement
# rounds, 0 = whole chain.  A truncated first pass yields no estimate: guided matching and the second pass do not run, so
# stop N measures pass 1 up to phase N (+ the matcher, listed beside it).
# usage: tools/pmc_stops.sh [pnp]
est=${1:-3d3d}; kern=$([ "$est" = pnp ] && echo k_chain_pnp || echo "k_chain<")
out=gpurun_out/pmc_stops_$est; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in 1 2 3 4 5 0; do
  SF_RANSAC_STOP=$s timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/stop$s -- python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline --estimator $est > $out/stop$s.log 2>&1
  echo "== stop $s"; python tools/pmc_kernel.py $out/stop$s "$kern" | grep -E "SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_INSTS_LDS"
done
