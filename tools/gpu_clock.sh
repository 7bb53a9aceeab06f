# samples the shader clock (rocm-smi) while the bench keeps the GPU busy: is the VALU peak priced at the real clock?
cd $GRAFT_REPO_ROOT
python bench.py --steps 8000 --warmup 3 --inflight 3 --no-cpu-baseline > /tmp/b.json 2>/dev/null &
BP=$!
for i in $(seq 1 45); do
  echo "t=$i $(rocm-smi --showclocks 2>&1 | grep -i 'sclk' | head -1 | sed 's/.*(//;s/).*//') $(rocm-smi --showpower 2>&1 | grep -i 'power' | head -1 | sed 's/.*: //')"
  sleep 1
  kill -0 $BP 2>/dev/null || break
done
wait $BP
cut -c1-160 /tmp/b.json
