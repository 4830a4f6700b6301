cd /root/repo
OUT=gpurun_out/bf16_lds_sweep2.txt
rm -f $OUT
for cfg in "2 8" "2 16" "4 8" "4 16" "8 16"; do set -- $cfg
echo "== LDS kernel JH_BF16_CWB=$1 JH_BF16_S=$2" >> $OUT
JH_BF16_CWB=$1 JH_BF16_S=$2 GB_PREFILL_SHAPES=1 GB_KINDS=3 python tools/gemm_bench.py 2>&1 | grep "M=" | grep -v "N=28672" >> $OUT
done
for cfg in "4 2" "4 1" "8 2"; do set -- $cfg
echo "== LDS kernel JH_BF16_CWB=$1 JH_BF16_S=$2" >> $OUT
JH_BF16_CWB=$1 JH_BF16_S=$2 GB_PREFILL_SHAPES=1 GB_KINDS=3 python tools/gemm_bench.py 2>&1 | grep "M=256" >> $OUT
done
cat $OUT
