set -x
cd /root/repo
export TMPDIR=/tmp
for mode in base combine direct128 direct64; do
  case $mode in
    base) export JH_X=1;;
    combine) export JH_ATTN_COMBINE_KERNEL=1;;
    direct128) unset JH_ATTN_COMBINE_KERNEL; export JH_ATTN_DIRECT=1; export JH_ATTN_DIRECT_CHUNK=128;;
    direct64) export JH_ATTN_DIRECT=1; export JH_ATTN_DIRECT_CHUNK=64;;
  esac
  rm -rf /tmp/prof_x
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_x -- python /root/repo/bench.py --steps 128 --warmup 8 --no-cpu-baseline --no-parity > /tmp/bx.log 2>&1)
  echo "== $mode" >> gpurun_out/attn_exp.txt
  tail -1 /tmp/bx.log | cut -c1-120 >> gpurun_out/attn_exp.txt
  python tools/rocpd_stats.py $(find /tmp/prof_x -name "*.db" | head -1) 2>&1 | grep -E "attn_|gemv_i8q4_kernel<5|gemv_i8q4_kernel<2, 1, 1, 2|gemv_i8q4_kernel<2, 1, 2" | cut -c1-120 >> gpurun_out/attn_exp.txt
done
cat gpurun_out/attn_exp.txt
