mkdir -p gpurun_out/r5
for n in 4 2; do timeout 900 python tools/dist_local_bench.py 256 $n 2000000 2>&1 | grep -v "^\[amghip\]" | tee -a gpurun_out/r5/dist_local_final.log | tail -6; done
export AMG_DIST_ONE_GPU=1
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 3 --warmup 1 --transport ipc > gpurun_out/r5/dist_ipc4_final.json 2> gpurun_out/r5/dist_ipc4_final.err; tail -3 gpurun_out/r5/dist_ipc4_final.err; cut -c1-1200 gpurun_out/r5/dist_ipc4_final.json
