cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -k "one_process_bench_host" > gpurun_out/r03o_one_process_host_test.log 2>&1; tail -15 gpurun_out/r03o_one_process_host_test.log
timeout 400 python bench.py > gpurun_out/r03_LLAMA3_8B_bench.json 2> gpurun_out/r03_LLAMA3_8B_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_LLAMA3_8B_bench.json').read().strip().splitlines()[-1])
print(d['value'], d.get('strict_tokens_per_s'), d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['us_per_launch'], d['roofline']['us_per_launch_rocprof'])
PY
timeout 900 python bench.py --config MISTRAL_7B --cpu-steps 24 > gpurun_out/r03_MISTRAL_7B_bench.json 2> gpurun_out/r03_MISTRAL_7B_bench.err
tail -c 1800 gpurun_out/r03_MISTRAL_7B_bench.json
GPU_MAX_HW_QUEUES=8 timeout 300 python -m jlama_amd.distributed --one-process --config LLAMA32_1B --gpus 2 --devices 0,0 --steps 64 --warmup 4 --prompt 16 > gpurun_out/r03o_one_process_1b_2stages_loopback_with_tp_leg.json 2> gpurun_out/r03o_op.err; tail -c 900 gpurun_out/r03o_one_process_1b_2stages_loopback_with_tp_leg.json; tail -3 gpurun_out/r03o_op.err
