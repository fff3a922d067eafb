// rover_model.hpp -- constants of the reduced AAU-rover model used by the HIP kernels (gfx950).
//
// Values come from the reference asset rover_envs/assets/robots/aau_rover_simple/rover_instance.usd (joint frames,
// masses; SURVEY.md App. A/E -> tools/derive_rover_model.py -> tests/golden/rover_model.json) and from the actuator /
// rigid-body configuration rover_envs/assets/robots/aau_rover_simple.py:18-65.  tests/test_model_constants.py checks
// this table against the JSON fixture and against the CPU oracle's own copy.
#pragma once

#define RV_M_TOTAL 25.0f
#define RV_COM_B_INIT {0.006704f, 0.0f, -0.054646f}
#define RV_INERTIA_B_INIT {3.354645f, 3.344834f, 6.149923f}
// wheel order: FL, FR, CL, CR, RL, RR
#define RV_WHEEL_B_INIT                                                                                                \
    {{0.44f, 0.3925f, -0.16699f}, {0.44f, -0.3925f, -0.16699f}, {0.007f, 0.3885f, -0.16699f},                          \
     {0.007f, -0.3885f, -0.16699f}, {-0.44f, 0.3925f, -0.16699f}, {-0.44f, -0.3925f, -0.16699f}}
#define RV_WHEEL_BOGIE_INIT {0, 1, 0, 1, 2, 2}
#define RV_WHEEL_STEER_INIT {0, 1, -1, -1, 2, 3}   // steer order: FL, FR, RL, RR
#define RV_WHEEL_BODY_INIT {9, 10, 7, 8, 11, 12}   // row in the 13-body contact sensor (rover_env_cfg.py:72-75)
#define RV_SLOT_WHEEL_INIT {0, 2, 1, 3, 4, 5}      // solver slot -> wheel: FL, CL, FR, CR, RL, RR (slots 2j, 2j+1 share bogie j)
// bogie order: FL_Boogie, FR_Boogie, R_Boogie
#define RV_BOGIE_PIVOT_INIT {{0.1535f, 0.2225f, 0.03f}, {0.1535f, -0.2225f, 0.03f}, {-0.325f, 0.0f, 0.03f}}
#define RV_BOGIE_AXIS_INIT {{0.0f, 1.0f, 0.0f}, {0.0f, -1.0f, 0.0f}, {1.0f, 0.0f, 0.0f}}
#define RV_BOGIE_INERTIA_INIT {0.47474f, 0.47474f, 1.141279f}
// the three bogie subtrees (beam + steer links + wheels) at q = 0: mass and centre of mass (body frame) from the USD link table
// (tools/derive_rover_model.py -> tests/golden/rover_model.json "subtrees"): cfg.mass_model = 1
#define RV_SUBTREE_MASS_INIT {7.0f, 7.0f, 9.0f}
#define RV_SUBTREE_COM_INIT {{0.27019f, 0.38237f, -0.06738f}, {0.27019f, -0.38237f, -0.06738f}, {-0.40167f, 0.0f, -0.07031f}}
#define RV_WHEEL_CONTACT_RADIUS 0.10179f  // 0.26878 (observations.py:45) - 0.16699
// actuators: aau_rover_simple.py:42-64
#define RV_STEER_INERTIA 0.005f
#define RV_STEER_KP 8000.0f
#define RV_STEER_KD 1000.0f
#define RV_STEER_EFFORT 12.0f
#define RV_STEER_VLIM 6.0f
#define RV_STEER_QLIM 1.5707963267948966f
#define RV_WHEEL_INERTIA 0.005f
#define RV_WHEEL_KP 100.0f
#define RV_WHEEL_KD 4000.0f
#define RV_WHEEL_EFFORT 12.0f
#define RV_WHEEL_VLIM 6.0f
#define RV_BOGIE_QLIM 0.17453292519943295f
#define RV_BOGIE_DAMPING 2.0f
#define RV_BAUMGARTE 0.2f
#define RV_MAX_DEPENETRATION_VEL 1.0f  // aau_rover_simple.py:27
#define RV_MAX_LINEAR_VEL 1.5f         // aau_rover_simple.py:25
#define RV_GRAVITY 9.81f
#define RV_OBSTACLE_EPS 1.0e-3f
// Contact report of the seven LINK bodies of the 13-body sensor (rover_env_cfg.py:72-75 matches .*_(Drive|Steer|Boogie|Body):
// 3 bogies, 4 steer links, 6 drive wheels; the chassis prim "Body" has no underscore and is not a sensor body): twelve sample
// points on the undersides of the links, body frame at zero bogie angle, one per (solver slot, role) lane of the group
// mapping.  Slot order FL, CL, FR, CR, RL, RR; per slot {role A point, role B point}.  Derived from the link frames of
// tests/golden/rover_model.json: steer forks inboard of their wheel at hub height + 0.067 (role A of the steered slots),
// the front bogie arms (pivot -> steer joint, pivot -> centre hub) and the rear bogie beam (pivot -> rear steer joints).
#define RV_LINK_POINT_INIT                                                                                             \
    {{{0.44f, 0.3125f, -0.10f}, {0.29675f, 0.3075f, 0.0175f}},    /* FL: FL_Steer fork | FL_Boogie front arm        */ \
     {{0.08025f, 0.3055f, -0.0685f}, {0.007f, 0.3085f, -0.12f}},  /* CL: FL_Boogie rear arm | rear arm, lower end   */ \
     {{0.44f, -0.3125f, -0.10f}, {0.29675f, -0.3075f, 0.0175f}},  /* FR                                            */ \
     {{0.08025f, -0.3055f, -0.0685f}, {0.007f, -0.3085f, -0.12f}}, /* CR                                            */ \
     {{-0.44f, 0.3125f, -0.10f}, {-0.3825f, 0.19625f, 0.0175f}},   /* RL: RL_Steer fork | R_Boogie beam, left half   */ \
     {{-0.44f, -0.3125f, -0.10f}, {-0.3825f, -0.19625f, 0.0175f}}} /* RR: RR_Steer fork | R_Boogie beam, right half  */
// sensor body row of each sample point (bodies 0..2 = FL / FR / R bogie, 3..6 = FL, FR, RL, RR steer)
#define RV_LINK_BODY_INIT {{3, 0}, {0, 0}, {4, 1}, {1, 1}, {5, 2}, {6, 2}}
#define RV_LINK_STIFFNESS 2.0e4f   // N per metre of penetration into the obstacle layer (report only: the episode ends on contact)
#define RV_WARM_START 0.85f

#define RV_PI_F 3.14159265358979323846f
#define RV_TWO_PI_F 6.28318530717958647692f
