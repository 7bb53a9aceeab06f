set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import eppm_amd; print(eppm_amd.lib().eppm_version())" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -40
