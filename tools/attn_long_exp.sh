cd /root/repo
OUT=gpurun_out/attn_long3.txt
rm -f $OUT
for pr in 128 1024 2048 4096 8100; do
echo -n "prompt=$pr default tiers: " >> $OUT
python bench.py --no-parity --no-cpu-baseline --prompt $pr --steps 64 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'tok/s')" >> $OUT
done
cat $OUT
