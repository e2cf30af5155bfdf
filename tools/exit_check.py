"""Does a process that used the multi-GPU entry points exit cleanly?  (tools only)  usage: exit_check.py <mode>"""
import sys

import numpy as np

sys.path.insert(0, ".")
import bohip

mode = sys.argv[1]
rng = np.random.default_rng(0)
X = rng.random((300, 3)); y = rng.random(300); Xs = rng.random((1000, 3))
ll = np.full(3, -0.5)
if mode == "gp":
    m = bohip.ElasticGPE(3, kernel=bohip.SEArd(ll, 0.0), capacity=300); m.append_(X.T, y)
    print(m.score("EI", [0.5], Xs.T)[1:])
elif mode == "mgp_keep":       # object alive at interpreter exit
    m = bohip.MultiGPE(3, devices=[0], shards_per_device=4, kernel=bohip.SEArd(ll, 0.0), capacity=300); m.append_(X.T, y)
    print(m.score("EI", [0.5], Xs.T)[1:])
elif mode == "mgp_close":
    m = bohip.MultiGPE(3, devices=[0], shards_per_device=4, kernel=bohip.SEArd(ll, 0.0), capacity=300); m.append_(X.T, y)
    print(m.score("EI", [0.5], Xs.T)[1:]); m.close()
elif mode == "mgp_thompson":
    m = bohip.MultiGPE(3, devices=[0], shards_per_device=4, kernel=bohip.SEArd(ll, 0.0), capacity=300); m.append_(X.T, y)
    print(m.thompson(Xs.T, 64, seed=1)[1][:4]); m.close()
elif mode == "comm":
    m = bohip.ElasticGPE(3, kernel=bohip.SEArd(ll, 0.0), capacity=300); m.append_(X.T, y)
    m.comm_init(bohip.comm_unique_id(), 0, 1)
    print(m.thompson_sharded(Xs.T, 8, 3, 0, 1000)[1][:4]); m.comm_destroy()
elif mode == "comm_keep":
    m = bohip.ElasticGPE(3, kernel=bohip.SEArd(ll, 0.0), capacity=300); m.append_(X.T, y)
    m.comm_init(bohip.comm_unique_id(), 0, 1)
    print(m.thompson_sharded(Xs.T, 8, 3, 0, 1000)[1][:4])
elif mode == "mgp_threads":
    import os
    os.environ["BOHIP_MGP_THREADS"] = "1"
    m = bohip.MultiGPE(3, devices=[0], shards_per_device=4, kernel=bohip.SEArd(ll, 0.0), capacity=300); m.append_(X.T, y)
    print(m.score("EI", [0.5], Xs.T)[1:])
elif mode == "torch_then_mgp":
    import torch
    t = torch.zeros(4, device="cuda")
    m = bohip.MultiGPE(3, devices=[0], shards_per_device=4, kernel=bohip.SEArd(ll, 0.0), capacity=300); m.append_(X.T, y)
    print(m.score("EI", [0.5], Xs.T)[1:])
