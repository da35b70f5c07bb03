"""One rank of a Go-ICP run whose bound batches are shared by the ranks on the device (rehearsal / debugging).
usage: torchrun ... tools/goicp_rank.py <repo root> <output prefix> <speculate> [cubes|launches]"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from icp_amd.host import GoIcp, normalize_goicp
from conftest import load_scan
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)                       # both ranks share the one card of the box
m, d, _ = normalize_goicp(load_scan("bunny000"), load_scan("bun270"))
d = np.ascontiguousarray(d[::8])
buf = torch.zeros(1 << 14, dtype=torch.float64, device="cuda")
calls = [0]
def allreduce_dev(t):
    calls[0] += 1
    # Two ranks on ONE card cannot use RCCL (one communicator rank per GPU), and gloo's own CUDA path hangs on this
    # stack, so this stand-in for the caller's collective stages through the host; what is under test is the library
    # side: shares evaluated on the device, a device buffer handed to the collective, one download afterwards.
    h = t.cpu(); dist.all_reduce(h); t.copy_(h)
go = GoIcp(m, d, mse=1e-3, dt_div=300, device=0, rank=rank, world=world, allreduce_dev=allreduce_dev, dev_buffer=buf)
go.options(batched=True, lookahead=4, speculate=int(sys.argv[3]), shard_cubes=len(sys.argv) < 5 or sys.argv[4] == "cubes")
r = go.registration()
out = dict(optError=float(r["optError"]), lb=float(r["lb"]), rot_pops=r["rot_pops"], optMat=r["optMat"].tolist(), launches=r["launches"],
           collectives=calls[0], initError=float(r["initError"]), icpError=float(r["icpError"]))
json.dump(out, open(sys.argv[2] + f".{rank}", "w"))
dist.barrier(); dist.destroy_process_group()
