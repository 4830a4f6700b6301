cd /root/repo
OUT=gpurun_out/attn_long4.txt
rm -f $OUT
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "slice_tiers or long_slices or attention" 2>&1 | tail -3 >> $OUT
for pr in 128 600 1024 2048 4096 8100; do
echo -n "prompt=$pr pipelined tail rounds: " >> $OUT
python bench.py --no-parity --no-cpu-baseline --prompt $pr --steps 64 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'tok/s')" >> $OUT
done
cat $OUT
