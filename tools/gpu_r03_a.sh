# round 3, first GPU call: the new host-boundary paths (registered caller memory, batch begin_into), the window-span boundary
# cases of the refine, and the host-boundary rates
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03_a
timeout 1500 python -m pytest tests -m gpu -x -q -k "host_boundary or pinned or window or c2f or single_scale or begin_end or abi" 2>&1 | tail -15 | tee gpurun_out/r03_a/tests_new.txt
timeout 600 python tools/host_boundary.py --json > gpurun_out/r03_a/host_boundary.json 2> gpurun_out/r03_a/host_boundary.err; cat gpurun_out/r03_a/host_boundary.json; tail -5 gpurun_out/r03_a/host_boundary.err
timeout 900 python bench.py --no-cpu-baseline --no-extras --steps 40 > gpurun_out/r03_a/bench_quick.json 2> gpurun_out/r03_a/bench_quick.err; cut -c1-300 gpurun_out/r03_a/bench_quick.json
