import sys
sys.path.insert(0, ".")
import numpy as np
mode = sys.argv[1]
if mode == "torch_first":
    import torch
    t = torch.zeros(4, device="cuda")
import bohip
rng = np.random.default_rng(0)
X = rng.random((300, 3)); y = rng.random(300); Xs = rng.random((1000, 3))
m = bohip.ElasticGPE(3, kernel=bohip.SEArd(np.full(3, -0.5), 0.0), capacity=300); m.append_(X.T, y)
print(m.score("EI", [0.5], Xs.T)[1:])
if mode != "torch_first":
    import torch
    t = torch.zeros(4, device="cuda")
    print(t.sum().item())
if mode == "nccl":
    import torch.distributed as dist, os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29777", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    dist.all_reduce(t); print(t.sum().item()); dist.destroy_process_group()
