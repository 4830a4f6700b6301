cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03n_gpu_tests_full.log 2>&1
tail -6 gpurun_out/r03n_gpu_tests_full.log
timeout 700 bash tools/profile_round.sh r03 LLAMA3_8B > gpurun_out/r03_profile_round_8b.log 2>&1
PMC=0 timeout 500 bash tools/profile_round.sh r03 LLAMA32_1B > gpurun_out/r03_profile_round_1b.log 2>&1
PMC=0 timeout 600 bash tools/profile_round.sh r03 MISTRAL_7B > gpurun_out/r03_profile_round_mistral.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_*_bench.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d.get('strict_tokens_per_s'), d['config'].get('prefill_ms'), (d.get('strict_order') or {}).get('prefill_ms'), d['roofline'].get('frac'), d['roofline'].get('traffic'))
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
