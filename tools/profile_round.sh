#!/bin/bash
# Round profile of one bench config on the GPU box:  tools/profile_round.sh <round tag, e.g. r02> <CONFIG> [extra bench args]
#   1. un-profiled bench line                         -> gpurun_out/<tag>_<CONFIG>_bench.json
#   2. rocprofv3 --kernel-trace --stats               -> gpurun_out/<tag>_<CONFIG>_kernel_trace_stats.md (+ the traced run's bench line)
#      and the begin/end timeline of 4 consecutive decode tokens -> gpurun_out/<tag>_<CONFIG>_decode_timeline_4_tokens.md
#   3. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE  -> gpurun_out/<tag>_<CONFIG>_pmc_{fetch,write}_size.md  (separate passes, counters only)
#   4. the dominant kernel's figures + the source hash of the profiled binary -> gpurun_out/<tag>_<CONFIG>_dominant_kernel.json
# Copy what should be judged into profiles/ (bench.py reports `traffic` only from a profile whose hash matches its build).
set -x
TAG=${1:-r03}; CFG=${2:-LLAMA3_8B}; shift; shift
PMC=${PMC:-1}     # PMC=0: bench line + kernel trace only (the secondary configs)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
EXTRA="$@"
if [ "${SKIP_BENCH:-0}" != "1" ]; then python $REPO/bench.py --config $CFG --steps 256 --warmup 16 $EXTRA > $OUT/${TAG}_${CFG}_bench.json 2> $OUT/${TAG}_${CFG}_bench.err; fi
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $REPO/bench.py --config $CFG --steps 128 --warmup 8 --no-cpu-baseline --no-parity --no-tier1-host > /tmp/bench_prof.log 2>&1
tail -c 1500 /tmp/bench_prof.log > $OUT/${TAG}_${CFG}_rocprof_log_tail.txt
grep '^{"metric"' /tmp/bench_prof.log | tail -1 | cut -c1-4000 > $OUT/${TAG}_${CFG}_bench_under_rocprof.json
python $REPO/tools/rocpd_stats.py $(find /tmp/prof_kt -name "*.db" | head -1) > $OUT/${TAG}_${CFG}_kernel_trace_stats.md 2>&1
python $REPO/tools/rocpd_timeline.py $(find /tmp/prof_kt -name "*.db" | head -1) 4 > $OUT/${TAG}_${CFG}_decode_timeline_4_tokens.md 2>&1
python $REPO/tools/rocpd_timeline.py $(find /tmp/prof_kt -name "*.db" | head -1) 4 order-free > $OUT/${TAG}_${CFG}_decode_timeline_4_tokens_order_free.md 2>&1
if [ "$PMC" = "0" ]; then ls -la $OUT; exit 0; fi
# (round 6: bench.py under --pmc dies with SIGSEGV inside the profiler a few seconds in, whatever legs are switched off; the same decode through
# tools/strict_bench.py -- same library, same graphs, same kernels -- collects fine, so the counter passes run that)
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -- python $REPO/tools/strict_bench.py $CFG 16 strict > /tmp/bench_f.log 2>&1
python $REPO/tools/rocpd_pmc.py $(find /tmp/prof_f -name "*.db" | head -1) > $OUT/${TAG}_${CFG}_pmc_fetch_size.md 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -- python $REPO/tools/strict_bench.py $CFG 16 strict > /tmp/bench_w.log 2>&1
python $REPO/tools/rocpd_pmc.py $(find /tmp/prof_w -name "*.db" | head -1) > $OUT/${TAG}_${CFG}_pmc_write_size.md 2>&1
python $REPO/tools/dominant_kernel_json.py $OUT/${TAG}_${CFG} $CFG > $OUT/${TAG}_${CFG}_dominant_kernel.json
ls -la $OUT
