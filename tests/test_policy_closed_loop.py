"""Closed-loop sanity signal with the reference's shipped checkpoint (build container only: needs /root/reference).
The policy was trained in Isaac Sim / PhysX; that it reaches targets in this repository's model -- and stops doing so
when an observation convention is flipped -- is the only end-to-end evidence available for the unpinned layer."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CKPT = "/root/reference/rover_envs/envs/navigation/robots/aau_rover/policies/best_agent.pt"
pytestmark = pytest.mark.skipif(not os.path.exists(CKPT), reason="reference checkpoint not present")


def test_pretrained_policy_reaches_targets_in_this_model(oracle):
    import policy_closed_loop as pcl
    from helpers import oracle_terrain, small_procedural
    n, steps = 128, 220
    ter = small_procedural()
    t = oracle_terrain(oracle, ter, 2 * n)
    pol = pcl.load_policy()
    rng = np.random.RandomState(0)
    good = pcl.run(n, steps, lambda o: pol(o), t)
    rand = pcl.run(n, steps, lambda o: rng.uniform(-1, 1, (n, 2)).astype(np.float32), t)
    flipped = pcl.run(n, steps, lambda o: pol(pcl.variant("heading sign flipped", np.where(np.isfinite(o), o, 0))), t)
    transposed = pcl.run(n, steps, lambda o: pol(pcl.variant("scan transposed (y fastest)", np.where(np.isfinite(o), o, 0))), t)
    # [time_limit, success, far, collision]
    assert good[1] >= 20, good                       # the Isaac-Sim-trained policy drives this model to its targets
    assert rand[1] == 0 and flipped[1] == 0          # ... which neither random actions nor a flipped heading sign do
    assert good[1] > 1.3 * transposed[1] and good[3] < transposed[3]   # the ORBIT ray order (x fastest) is the right one
