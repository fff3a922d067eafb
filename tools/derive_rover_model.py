#!/usr/bin/env python3
"""Derive the constants of the reduced AAU-rover model from the articulation data of the reference asset.

Input  : the joint / mass table read out of ``rover_envs/assets/robots/aau_rover_simple/rover_instance.usd``
         (binary USD crate; values transcribed in SURVEY.md Appendix A / E -- joint ``localPos0/1``,
         ``localRot0/1`` (w,x,y,z), ``physics:axis = X`` for every joint, link masses and centres of mass).
Output : ``tests/golden/rover_model.json`` -- rest-pose link frames, wheel centres, bogie pivots / axes, the
         composite mass, centre of mass and inertia used by BOTH the oracle (oracle/rover_oracle.c) and the
         HIP kernels (isaac_rover_orbit_amd/csrc/rover_model.hpp).  tests/test_model_constants.py checks that
         the two hand-written constant tables agree with this file.

Frames : rover frame = ``Body`` link frame, X forward, Y left, Z up (stage is Z-up, metres).
"""
from __future__ import annotations

import json
import os

import numpy as np

S = 0.7071067811865476


def qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qrot(q, v):
    return qmul(qmul(q, np.array([0.0, *v])), qconj(q))[1:]


def unit(q):
    q = np.asarray(q, float)
    return q / np.linalg.norm(q)


# joint : (parent, localPos0, localPos1, localRot0, localRot1)     -- SURVEY App. A table + App. E table
JOINTS = {
    "FL_Boogie": ("Body", (0.1535, 0.2225, 0.03), (-0.07, 0, 0), (0.5, -0.5, -0.5, 0.5), (S, 0, -S, 0)),
    "FR_Boogie": ("Body", (0.1535, -0.2225, 0.03), (-0.07, 0, 0), (0.5, 0.5, -0.5, -0.5), (S, 0, -S, 0)),
    "R_Boogie": ("Body", (-0.325, 0, 0.03), (0, 0, 0), (S, -S, 0, 0), (S, 0, -S, 0)),
    "FL_Steer": ("FL_Boogie", (0.2165, 0.025, 0.17), (0, 0, 0), (0, S, -S, 0), (S, 0, 0, S)),
    "FR_Steer": ("FR_Boogie", (0.2165, -0.025, 0.17), (0, 0, 0), (0, S, S, 0), (S, 0, 0, S)),
    "RL_Steer": ("R_Boogie", (-0.3925, 0.025, -0.115), (0, 0, 0), (0.5, -0.5, 0.5, -0.5), (S, 0, 0, S)),
    "RR_Steer": ("R_Boogie", (0.3925, 0.025, -0.115), (0, 0, 0), (0.5, 0.5, -0.5, -0.5), (S, 0, 0, S)),
    "CL_Drive": ("FL_Boogie", (-0.2165, 0.19699, 0.166), (0, 0, 0), (0.20253, 0.67748, -0.20253, 0.67748), (1, 0, 0, 0)),
    "CR_Drive": ("FR_Boogie", (-0.2165, -0.19699, 0.166), (0, 0, 0), (0.51531, 0.48421, 0.51531, -0.48421), (0, 0, 0, 1)),
    "FL_Drive": ("FL_Steer", (0, -0.17199, 0), (0, 0, 0), (0.12037, 0.69679, 0.12037, -0.69679), (1, 0, 0, 0)),
    "FR_Drive": ("FR_Steer", (0, -0.17199, 0), (0, 0, 0), (0.70112, -0.09179, -0.70112, -0.09181), (0, 0, 0, 1)),
    "RL_Drive": ("RL_Steer", (0, -0.17199, 0), (0, 0, 0), (0.52117, 0.47789, 0.52117, -0.47789), (1, 0, 0, 0)),
    "RR_Drive": ("RR_Steer", (0, -0.17199, 0), (0, 0, 0), (0.52125, 0.47781, -0.52125, 0.47781), (0, 0, 0, 1)),
}
# link : (mass, COM in link frame, box extents used for the link's own inertia [link frame is rotated, so the
#         extents are given in the ROVER frame at q = 0])                      -- masses / COMs: SURVEY App. A
LINKS = {
    "Body": (2.0, (0, 0, 0.105), (0.65, 0.445, 0.21)),
    "FL_Boogie": (3.0, (-0.0354, 0.0111, 0.1489), (0.50, 0.06, 0.20)),
    "FR_Boogie": (3.0, (-0.0354, -0.0111, 0.1489), (0.50, 0.06, 0.20)),
    "R_Boogie": (3.0, (0, 0, 0), (0.06, 0.80, 0.15)),
    "FL_Steer": (2.0, (0, -0.1022, 0.0425), (0.06, 0.06, 0.20)),
    "FR_Steer": (2.0, (0, -0.1022, 0.0425), (0.06, 0.06, 0.20)),
    "RL_Steer": (2.0, (0, -0.1022, 0.0425), (0.06, 0.06, 0.20)),
    "RR_Steer": (2.0, (0, -0.1022, 0.0425), (0.06, 0.06, 0.20)),
    "FL_Drive": (1.0, (0.0072, 0, 0), (0.2, 0.125, 0.2)),
    "FR_Drive": (1.0, (0.0072, 0, 0), (0.2, 0.125, 0.2)),
    "RL_Drive": (1.0, (0.0072, 0, 0), (0.2, 0.125, 0.2)),
    "RR_Drive": (1.0, (0.0072, 0, 0), (0.2, 0.125, 0.2)),
    "CL_Drive": (1.0, (0.0742, 0, 0), (0.2, 0.125, 0.2)),
    "CR_Drive": (1.0, (0.0742, 0, 0), (0.2, 0.125, 0.2)),
}
BOGIE_OF = {"FL": "FL_Boogie", "CL": "FL_Boogie", "FR": "FR_Boogie", "CR": "FR_Boogie", "RL": "R_Boogie", "RR": "R_Boogie"}
SUBTREE = {
    "FL_Boogie": ["FL_Boogie", "FL_Steer", "FL_Drive", "CL_Drive"],
    "FR_Boogie": ["FR_Boogie", "FR_Steer", "FR_Drive", "CR_Drive"],
    "R_Boogie": ["R_Boogie", "RL_Steer", "RR_Steer", "RL_Drive", "RR_Drive"],
}


def box_inertia(m, ext):
    x, y, z = ext
    return m / 12.0 * np.diag([y * y + z * z, x * x + z * z, x * x + y * y])


def main():
    frames = {"Body": (np.zeros(3), np.array([1.0, 0, 0, 0]))}
    joint_axis = {}
    for name, (par, p0, p1, r0, r1) in JOINTS.items():
        pp, pq = frames[par]
        r0, r1 = unit(r0), unit(r1)
        jq = qmul(pq, r0)
        jp = pp + qrot(pq, np.array(p0, float))
        cq = qmul(jq, qconj(r1))
        cp = jp - qrot(cq, np.array(p1, float))
        frames[name] = (cp, cq)
        joint_axis[name] = (jp, qrot(jq, np.array([1.0, 0, 0])))

    # composite mass properties at q = 0
    m_tot = 0.0
    com = np.zeros(3)
    coms = {}
    for link, (m, c, _) in LINKS.items():
        p, q = frames[link]
        cw = p + qrot(q, np.array(c, float))
        coms[link] = cw
        m_tot += m
        com += m * cw
    com /= m_tot
    inertia = np.zeros((3, 3))
    for link, (m, _, ext) in LINKS.items():
        d = coms[link] - com
        inertia += box_inertia(m, ext) + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))

    bogies = {}
    for b, links in SUBTREE.items():
        piv, ax = joint_axis[b]
        ax = np.round(ax, 6)
        ib = 0.0
        for link in links:
            m, _, ext = LINKS[link]
            d = coms[link] - piv
            d_perp = d - np.dot(d, ax) * ax
            ib += m * np.dot(d_perp, d_perp) + float(ax @ box_inertia(m, ext) @ ax)
        bogies[b] = {"pivot": piv.round(6).tolist(), "axis": ax.tolist(), "inertia": round(ib, 6)}

    # the three bogie SUBTREES (beam + steer links + wheels) at q = 0: mass, centre of mass, box-model inertia about that centre --
    # what cfg.mass_model = 1 needs (the subtree's weight as a generalised force on its bogie coordinate) and the static
    # wheel-load check of tests/test_oracle_physics.py derives its expectation from
    subtrees = {}
    for b, links in SUBTREE.items():
        mb = sum(LINKS[l][0] for l in links)
        cb = sum(LINKS[l][0] * coms[l] for l in links) / mb
        ib = np.zeros((3, 3))
        for l in links:
            m, _, ext = LINKS[l]
            d = coms[l] - cb
            ib += box_inertia(m, ext) + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        subtrees[b] = {"mass": mb, "com": cb.round(5).tolist(), "inertia_com_diag": np.diag(ib).round(4).tolist(), "links": links}
    link_table = {l: {"mass": LINKS[l][0], "com": coms[l].round(6).tolist()} for l in LINKS}

    wheels = {k: frames[f"{k}_Drive"][0].round(6).tolist() for k in ["FL", "FR", "CL", "CR", "RL", "RR"]}
    out = {
        "source": "rover_envs/assets/robots/aau_rover_simple/rover_instance.usd via SURVEY.md App. A/E",
        "total_mass": m_tot,
        "com": com.round(6).tolist(),
        "inertia_diag": np.diag(inertia).round(6).tolist(),
        "inertia_offdiag_xy_xz_yz": [round(inertia[0, 1], 6), round(inertia[0, 2], 6), round(inertia[1, 2], 6)],
        "wheel_centres": wheels,
        "wheel_bogie": BOGIE_OF,
        "bogies": bogies,
        "subtrees": subtrees,
        "links": link_table,
        "steer_axis": {k: joint_axis[f"{k}_Steer"][1].round(6).tolist() for k in ["FL", "FR", "RL", "RR"]},
        "drive_axis": {k: joint_axis[f"{k}_Drive"][1].round(6).tolist() for k in wheels},
        "link_frames": {k: {"pos": v[0].round(6).tolist(), "quat_wxyz": v[1].round(6).tolist()} for k, v in frames.items()},
        # body origin height above the wheel-contact plane is hard-coded in the reference:
        # rover_envs/envs/navigation/mdp/observations.py:43-45
        "body_height_above_contact_plane": 0.26878,
        "wheel_contact_radius": round(0.26878 + wheels["FL"][2], 6),
    }
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rover_model.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ["total_mass", "com", "inertia_diag", "inertia_offdiag_xy_xz_yz",
                                          "wheel_centres", "bogies", "subtrees", "wheel_contact_radius"]}, indent=1))


if __name__ == "__main__":
    main()
