mkdir -p gpurun_out/r5
export AMG_DIST_ONE_GPU=1
(timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --steps 2 --warmup 1 --transport ipc-staged --no-secondary --no-cpu-baseline --size 128 > gpurun_out/r5/dist2s.json 2> gpurun_out/r5/dist2s.err; echo rc=$? >> gpurun_out/r5/dist2s.err)
tail -3 gpurun_out/r5/dist2s.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/dist2s.json').read().strip().splitlines()[-1])
print("primary ms", d['ms_per_step'], d['parity']['rel_err'], d['config'].get('gs_pipelined_by_level'), d['config']['sharded_levels'], d.get('ipc_staged'))
for r in d['preflight']: print(r)
PY
