echo "== 1-rank RCCL forced DP"
RLIPV2_FORCE_DP=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/dp1.err | grep '^{' | cut -c100-330
echo "== 2 ranks on one GPU over gloo"
RLIPV2_SINGLE_DEVICE=1 RLIPV2_DIST_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/dp2.err | grep '^{' | cut -c100-400
tail -3 gpurun_out/dp2.err | cut -c1-300
