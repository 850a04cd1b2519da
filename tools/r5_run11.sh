mkdir -p gpurun_out/r5
export AMG_DIST_ONE_GPU=1
(timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 3 --warmup 1 --transport ipc > gpurun_out/r5/dist4.json 2> gpurun_out/r5/dist4.err; echo rc=$? >> gpurun_out/r5/dist4.err)
tail -5 gpurun_out/r5/dist4.err | cut -c1-400
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5/dist4.json').read().strip().splitlines()[-1])
    print("primary ms", d['ms_per_step'], "parity", d['parity'], "piped", d['config'].get('gs_pipelined_by_level'), "sharded", d['config']['sharded_levels'], "setup", d['setup_s'], d['shard_s'])
    for k,v in d.get('secondary',{}).items(): print(k, v.get('ms_per_step'), v.get('parity',{}).get('rel_err') if isinstance(v.get('parity'),dict) else v.get('error'), v.get('gs_pipelined_by_level'), v.get('sharded_levels'))
except Exception as e: print("parse failed", e)
PY
