#!/usr/bin/env python3
"""Probe: does RCCL accept two ranks on ONE device?  (Only informational -- NCCL refuses duplicate GPUs.)"""
import os, sys, socket, subprocess
if "RANK" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = [subprocess.Popen([sys.executable, __file__], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                           MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(2)]
    print("exit codes", [p.wait(timeout=240) for p in ps]); sys.exit(0)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=2)
    x = torch.full((1024,), float(dist.get_rank() + 1), device="cuda")
    dist.all_reduce(x); torch.cuda.synchronize()
    print("rank", dist.get_rank(), "all_reduce ->", float(x[0]))
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print("rank", os.environ["RANK"], "FAILED:", str(e)[:300]); sys.exit(3)
