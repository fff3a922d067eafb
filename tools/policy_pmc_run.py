#!/usr/bin/env python3
"""Dev tool (GPU box): a short fixed workload for rocprofv3 --pmc passes on the policy kernel: 20 forwards at N = 4096."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_policy_weights, synthetic_obs
from isaac_rover_orbit_amd.policy import RoverNet
ws, bs = random_policy_weights(seed=0, scale=3.0)
net = RoverNet(ws, bs)
obs = torch.from_numpy(synthetic_obs(4096)).cuda()
for _ in range(20):
    net(obs)
torch.cuda.synchronize()
