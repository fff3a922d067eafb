"""The rover model tables (tools/derive_rover_model.py -> tests/golden/rover_model.json -> kernel constants) against the
reference's binary USD asset, read with the dev-only crate reader (build container only)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
USD = "/root/reference/rover_envs/assets/robots/aau_rover_simple/rover_instance.usd"
pytestmark = pytest.mark.skipif(not os.path.exists(USD), reason="reference asset not present")


@pytest.mark.timeout(120)
def test_transcribed_tables_match_the_usd_asset():
    import usd_crate_dump as u
    crate, joints, links = u.rover_tables(USD)
    assert len(joints) == 13 and len(links) == 14                       # 14 bodies, 13 revolute joints
    assert abs(sum(l["mass"] for l in links.values()) - 25.0) < 1e-6
    assert u.check_model_fixture() < 2e-4
    # articulation / scene settings quoted in SURVEY App. A
    assert crate.attr("/rover", "physxArticulation:enabledSelfCollisions") is False
    assert crate.attr("/rover", "physxArticulation:solverPositionIterationCount") == 32
    assert abs(crate.attr("/physicsScene", "physics:gravityMagnitude") - 9.81) < 1e-5
    fl = [j for n, j in joints.items() if n.startswith("FL_Boogie")][0]
    assert abs(fl["lowerLimit"] + 10.0) < 1e-4 and abs(fl["upperLimit"] - 10.0) < 1e-4     # +-10 deg bogie stops
    st = [j for n, j in joints.items() if n.startswith("FL_Steer")][0]
    assert abs(st["stiffness"] - 8000.0) < 1e-3 and abs(st["damping"] - 1000.0) < 1e-3


@pytest.mark.timeout(60)
def test_derived_model_json_is_current(tmp_path):
    """tests/golden/rover_model.json is what tools/derive_rover_model.py produces from its tables."""
    import json
    import subprocess
    golden = os.path.join(ROOT, "tests", "golden", "rover_model.json")
    before = json.load(open(golden))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "derive_rover_model.py")], stdout=subprocess.DEVNULL, timeout=50)
    after = json.load(open(golden))
    assert before == after
