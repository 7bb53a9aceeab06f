# Vector-memory-pipeline counters (texture addresser TA, L1 TCP, data return TD) per kernel, batch 4, one context in flight:
# is a gather-heavy kernel bound by L1 tag look-ups rather than by VALU issue or latency?  Counters in their own passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B=${BATCH:-4}
P1="TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
P2="TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TD_TD_BUSY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE"

i=1
for P in "$P1" "$P2"; do
  rm -rf $O/pmc_tcp$i
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_tcp$i -- python3 $R/bench.py --steps 8 --warmup 4 --batch $B --inflight 1 --no-cpu-baseline --no-extras > $O/pmc_tcp$i.log 2>&1
  tail -2 $O/pmc_tcp$i.log | cut -c1-300
  python3 $R/tools/pmc_summary.py $O/pmc_tcp$i > $O/pmc_tcp${i}_b$B.csv
  i=$((i+1))
done
