"""Which first touch is slow on a cold box?  Times, in ONE fresh process: import torch, torch's first GPU operation (its lazy HIP initialisation
and the code objects of libtorch_hip), pinned memory, then libbohip's first RCCL communicator.  (Round 6: the first in-process test that does
all of this took 11-13 s on most boxes and 530-630 s on two.)   usage: python tools/cold_start_probe.py"""
import os, sys, time
t0 = time.time()
def say(m): print(f"[{time.time() - t0:8.2f}s] {m}", flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
say("import bohip"); import bohip
rng = np.random.default_rng(0)
X = rng.random((200, 3)); y = np.sin(3 * X).sum(1)
m = bohip.ElasticGPE(3, kernel=bohip.SEArd(np.full(3, -0.5), 0.0), logNoise=-2.0, capacity=200)
m.append_(X.T, y); m.fit_()
say("model fitted; import torch"); import torch
say("torch imported; first torch GPU op"); d = torch.zeros(8, dtype=torch.float64).to("cuda:0"); torch.cuda.synchronize()
say("torch GPU op done; pin_memory"); h = torch.zeros(2, dtype=torch.int64).pin_memory()
say("pinned; comm_unique_id (dlopen librccl)"); uid = bohip.comm_unique_id()
say("comm_init (ncclCommInitRank, 1 rank)"); m.comm_init(uid, 0, 1)
say("comm_init done"); m.comm_destroy(); say("destroyed")
