# all GPU parity tests, then the randomized sweep (tests/fuzz_parity.py SEED N)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 900 python tests/fuzz_parity.py ${FUZZ_SEED:-7} ${FUZZ_N:-24} > gpurun_out/fuzz.log 2>&1; echo fuzz rc=$?; tail -4 gpurun_out/fuzz.log; grep -c OK gpurun_out/fuzz.log
