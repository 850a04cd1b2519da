mkdir -p gpurun_out/r5
(timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_ipc.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r5/pytest_dist.log 2>&1; echo rc=$? >> gpurun_out/r5/pytest_dist.log)
tail -5 gpurun_out/r5/pytest_dist.log | cut -c1-300
export AMG_DIST_ONE_GPU=1
(timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 3 --warmup 1 --transport ipc > gpurun_out/r5/dist4b.json 2> gpurun_out/r5/dist4b.err; echo rc=$? >> gpurun_out/r5/dist4b.err)
tail -2 gpurun_out/r5/dist4b.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/dist4b.json').read().strip().splitlines()[-1])
print("primary ms", d['ms_per_step'], d['parity']['rel_err'], d['config'].get('gs_pipelined_by_level'), "setup", d['setup_s'], "shard_s", d['shard_s'], d['setup_breakdown'])
for k,v in d.get('secondary',{}).items(): print(k, v.get('ms_per_step'), v.get('shard_s'), v.get('error'))
for r in d['preflight']: print(r)
PY
