cd /root/repo
OUT=gpurun_out/gemm_lds_8.txt
rm -f $OUT
echo "== tile kernel (scalar f32 scaling)" >> $OUT
JH_GEMM_LDS=0 GB_PREFILL_SHAPES=1 GB_KINDS=2 python tools/gemm_bench.py 2>&1 | grep "M=" >> $OUT
echo "== default dispatch" >> $OUT
GB_PREFILL_SHAPES=1 GB_KINDS=2 python tools/gemm_bench.py  2>&1 | grep "M=" >> $OUT
python bench.py --no-parity --no-cpu-baseline 2>/dev/null | cut -c1-2000 >> $OUT
python bench.py --no-parity --no-cpu-baseline --prompt 8100 --steps 32 2>/dev/null | cut -c1-2000 >> $OUT
cat $OUT
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -6
