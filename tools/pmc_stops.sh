mkdir -p gpurun_out/r03f; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in 0 1 2 4 5; do
  SF_RANSAC_STOP=$s timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/r03f/pmc_stop$s -- python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r03f/pmc_stop$s.log 2>&1
  echo "== stop $s"; python tools/pmc_kernel.py gpurun_out/r03f/pmc_stop$s k_verify_fused
done
