#!/bin/bash
set +e
export TMPDIR=/tmp
rocm-smi --showserial 2>/dev/null | grep Serial
echo "== N=2 flow over gloo, both ranks on GPU 0 (testing the multi-rank code path only)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 4 --backend gloo --share-gpu 2>&1 | tail -3 | cut -c1-900
echo "== N=2 stream mode"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --mode stream --backend gloo --share-gpu 2>&1 | tail -2 | cut -c1-900
echo "== N=1 via torchrun with nccl backend selected (no group is created at world 1)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu --no-host-path 2>&1 | tail -1 | cut -c1-300
echo "== nccl init smoke (1 rank group)"
timeout 300 python - <<'PY'
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29520", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.tensor([1.5], dtype=torch.float64).cuda()
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
print("nccl(RCCL) 1-rank group ok", t.item(), dist.get_backend())
dist.destroy_process_group()
PY
./tools/hbm_probe 10 32768 | head -8
