set -x
cd /root/repo
JH_BENCH_FORCE_PIPELINE=1 timeout 300 python bench.py --steps 64 --warmup 8 2>/dev/null | tail -1 | cut -c1-400 > gpurun_out/pipeline_w1.json
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $REPO/bench.py --steps 256 --warmup 16 --no-cpu-baseline > /tmp/bench_prof.log 2>&1
tail -1 /tmp/bench_prof.log | cut -c1-3000 > $REPO/gpurun_out/r01b_bench_under_rocprof.json
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $REPO/tools/rocpd_stats.py $DB > $REPO/gpurun_out/r01b_kernel_trace_stats.md 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -- python $REPO/bench.py --steps 32 --warmup 4 --no-cpu-baseline > /tmp/bench_f.log 2>&1
DBF=$(find /tmp/prof_f -name "*.db" | head -1)
python $REPO/tools/rocpd_pmc.py $DBF > $REPO/gpurun_out/r01b_pmc_fetch_size.md 2>&1
ls -la $REPO/gpurun_out/
