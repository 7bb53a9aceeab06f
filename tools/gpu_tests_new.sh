set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_configs_gpu.py -m gpu -x -q --deselect tests/test_configs_gpu.py::test_config5_uhd_patch_radius_17_bit_exact 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -5
python bench.py > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err; tail -3 gpurun_out/bench_r02a.err; cat gpurun_out/bench_r02a.json
