"""Run one of the reference's scripts unmodified on the MI355X-native environment:

    python -m isaac_rover_orbit_amd.compat.run /path/to/isaac_rover_orbit/examples/02_train/train.py \
        --task AAURoverEnv-v0 --num_envs 4096 --headless

Installs the compatibility namespace (no Isaac Sim / ORBIT needed), puts the reference checkout on ``sys.path`` and
executes the script as ``__main__``.  The script's own requirements that are NOT part of the hot path (skrl,
gymnasium, h5py ...) must be installed for it to do real training.
"""
from __future__ import annotations

import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    from . import install
    install()
    # <repo>/examples/02_train/train.py -> <repo> on sys.path so that `import rover_envs` resolves
    root = os.path.dirname(script)
    for _ in range(4):
        if os.path.isdir(os.path.join(root, "rover_envs")):
            sys.path.insert(0, root)
            break
        root = os.path.dirname(root)
    os.environ.setdefault("EXP_PATH", os.getcwd())      # train.py:27-29 reads it
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
