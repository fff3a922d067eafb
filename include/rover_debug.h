/*
 * rover_debug.h -- TEST-ONLY / MEASUREMENT-ONLY hooks exported by librover_hip.so.  NOT part of the drop-in boundary.
 *
 * The product never calls these: every one of them only SELECTS between kernel forms that produce the same bits (the parity tests
 * use them to pin exactly that: tests/test_gpu_parity.py, tests/test_gpu_lift.py; the measurement tools to time one form against
 * another: tools/n_sweep.py, tools/lift_time.py).  No reference counterpart -- the reference has one code path
 * (rover_envs/envs/navigation/entrypoints/rover_env.py:42-102) and no notion of kernel forms.  A binding that only wants the
 * reference's behaviour includes rover_hip.h / rover_lift.h and ignores this header; tests/test_abi.py checks that the library
 * exports nothing beyond what the headers under include/ declare.
 */
#ifndef ROVER_DEBUG_H
#define ROVER_DEBUG_H

#include "rover_hip.h"
#include "rover_lift.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How rover_step launches: -1 = automatic (default), 0 = two launches (step kernel + scan kernel), 1 = one launch with copy waves
 * (rover_step_scan_kernel) wherever it can run, 2 = one launch, single tile per wave (rover_step_scan1_kernel).  Same results. */
int rover_debug_set_fused(rover_sim *sim, int fused);

/* Scan kernel of the two-launch path: 0 = automatic, 1 = the generic kernel on the step path as well, 2 = the step form with one
 * env per synchronisation round, 7 = the wave-private scan (the scan phase of the one-launch kernels) as a kernel of its own. */
int rover_debug_set_scan_form(rover_sim *sim, int form);

/* FrankaCubeLift-v0 step kernel: lanes per env (8 = default, 16) and the two-wave arm / cube pipeline (1 = on, 0 = off,
 * -1 = automatic by batch size).  Same results. */
int rover_lift_debug_set_lanes(rover_lift_sim *sim, int lanes);
int rover_lift_debug_set_pipeline(rover_lift_sim *sim, int on);

#ifdef __cplusplus
}
#endif
#endif /* ROVER_DEBUG_H */
