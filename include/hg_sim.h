/*
 * hg_sim.h -- C-ABI of libhg_sim.so, the MI355X (gfx950) batched replacement
 * for the PyBullet calls on humanoid-gym's env-step path.
 *
 * The reference has no FFI of its own on this path: it calls the `pybullet`
 * CPython extension once per query, per env.  Each entry
 * point below therefore cites the PyBullet/qibullet call site(s) in the
 * reference that it replaces; INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns 0 on success, a negative HG_E* code on failure;
 *    hg_last_error() gives the thread-local message of the last failure;
 *  - all array arguments named *_dev are DEVICE pointers owned by the caller
 *    (e.g. torch tensors' data_ptr()), float32 / uint8 / int32, env-major and
 *    contiguous; nullable where stated;
 *  - work is enqueued on `stream` (a hipStream_t passed as void*, NULL = the
 *    default stream) and is asynchronous with respect to the host;
 *  - one hg_sim lives on one device; a handle is not thread-safe, distinct
 *    handles are independent;
 *  - there is no CPU fallback: every entry point fails with HG_ENODEVICE when
 *    no HIP device is usable.
 */
#ifndef HG_SIM_H
#define HG_SIM_H

#include <stdint.h>
#include "hg_model.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HG_OK          0
#define HG_EINVAL     -1   /* bad argument (null, size, index out of range)   */
#define HG_ENODEVICE  -2   /* no usable HIP device / HIP runtime error        */
#define HG_ENOMEM     -3   /* device allocation failed                        */
#define HG_EUNSUPPORTED -4 /* model exceeds what the kernels were built for   */

typedef struct hg_sim hg_sim;

/* Message of the last error raised on the calling thread ("" if none). */
const char *hg_last_error(void);

/* Library/ABI version, bumped on any signature or struct-layout change (3: hg_model.joint_limited,
 * hg_config.max_coord_velocity / enable_limits / limit_max_impulse, hg_stats_flush, hg_set_base_velocity;
 * 4: hg_get_joint_torques; 5: HG_MAX_LINKS 48 -> 62 (hg_model layout), hg_stats_accumulate_noalloc;
 * 6: hg_config.residual_threshold / store_torques; 7: hg_model.sphere_half, HG_MAX_SPHERES 1024;
 * 8: hg_config.store_contacts in the place of the reserved word).  A binding must compare it with the version its struct mirrors
 * were written for before it hands a struct to the library (humanoid_gym_amd/_capi.py does). */
int hg_abi_version(void);

/*
 * Build n_envs copies of the robot on device `device_id`, all at the reset
 * pose.  Replaces p.connect + p.loadURDF + p.loadMJCF(ground) + setGravity +
 * setPhysicsEngineParameter + setTimeStep  (dancer_env.py:13,58-65) and
 * SimulationManager.launchSimulation + spawnNao/spawnPepper/spawnRomeo
 * (nao_env.py:17-20, pepper_env.py:15-18, romeo_env.py:14-17).
 * The model and config are checked on the host before a device is looked for: HG_EINVAL for sizes / indices out of
 * range, links not ordered parents-first, masses or principal inertias that are not positive and finite, a joint axis
 * or reset quaternion that is not of unit length, lower > upper, negative or NaN gains / bounds, a self-collision pair
 * inside one body; HG_EUNSUPPORTED for trees no kernel instantiation fits (depth > 12, forks nested deeper than 2,
 * rows beyond HG_MAX_ROWS).  hg_last_error() names the offending field and index.
 */
int hg_create(const hg_model *model, const hg_config *config, int n_envs, int device_id, hg_sim **out);

/* Frees device memory.  Replaces p.disconnect() (dancer_env.py:116, nao_env.py:92). */
int hg_destroy(hg_sim *sim);

/*
 * Restore the reset pose (zero velocities, targets = q_init) for the envs
 * whose byte in env_mask_dev is non-zero; NULL resets every env.  Replaces
 * p.resetSimulation + reload (dancer_env.py:56-65) and removeNao/spawnNao
 * (nao_env.py:84-85, pepper_env.py:79-80, romeo_env.py:55-56).
 */
int hg_reset(hg_sim *sim, const uint8_t *env_mask_dev, void *stream);

/*
 * One env step for every env: write position targets, advance n_substeps
 * fixed physics steps, read observations and flags.  Replaces
 *   p.setJointMotorControlArray(..., POSITION_CONTROL, action)  dancer_env.py:19
 *   robot.setAngles(names, actions, 1.0)            nao_env.py:70, pepper_env.py:61, romeo_env.py:37
 *   p.stepSimulation()                              dancer_env.py:20
 *   p.getJointState x n / getAnglesPosition / getAnglesVelocity
 *                                                   dancer_env.py:24, nao_env.py:73, pepper_env.py:66-67
 *   robot.getPosition()                             pepper_env.py:65, romeo_env.py:41
 *   the getContactPoints scan                       dancer_env.py:32-35
 *
 *   actions_dev  [n_envs][n_sel] target angles (rad); NULL keeps the targets
 *   joint_sel    HOST array [n_sel] of dof indices addressed by the action
 *                columns (NaoEnv.step(actions, joints=...), nao_env.py:48,70);
 *                NULL = the model's act_joint list (then n_sel is ignored)
 *   obs_q_dev, obs_qd_dev [n_envs][n_act]  joint positions / velocities in
 *                act_joint order (nullable)
 *   base_xyyaw_dev [n_envs][3]  x, y, yaw of the base link frame (nullable)
 *   reward_dev   [n_envs] written 0 (the reference's default reward), nullable
 *   done_dev     [n_envs] written 1 where the env's state went non-finite and
 *                was reset in place, else 0 (the reference's default done is
 *                False), nullable
 *   collision_dev [n_envs] 1 if any non-base link had a contact in the last
 *                physics step, ground included (dancer_env.py:32-35), nullable
 */
int hg_step(hg_sim *sim, const float *actions_dev, const int32_t *joint_sel, int n_sel, int n_substeps,
            float *obs_q_dev, float *obs_qd_dev, float *base_xyyaw_dev,
            float *reward_dev, uint8_t *done_dev, uint8_t *collision_dev, void *stream);

/* Positions and velocities of ALL dofs, [n_envs][n_links] each (nullable).
 * Replaces the p.getJointState loop, dancer_env.py:22-24. */
int hg_get_joint_states(hg_sim *sim, float *q_dev, float *qd_dev, void *stream);

/* Motor torque applied to every dof during the last physics step, [n_envs][n_links]: the motor row's
 * impulse / dt, i.e. getJointState()[3] (appliedJointMotorTorque), dancer_env.py:24.  The other entry of
 * that tuple, the joint reaction forces [6], is all zeros in the reference (it never enables the joint
 * force/torque sensors), so there is nothing to compute for it.  Zero after hg_reset. */
int hg_get_joint_torques(hg_sim *sim, float *tau_dev, void *stream);

/*
 * World poses of every body, [n_envs][n_links+1] records, body 0 = base:
 *   com_pos[3], com_quat[4]      inertial frame (getLinkState()[0], [1])
 *   frame_pos[3], frame_quat[4]  URDF link frame (getLinkState()[4], [5])
 * Replaces the p.getLinkState loop, dancer_env.py:25-27.  All nullable.
 */
int hg_get_link_states(hg_sim *sim, float *com_pos_dev, float *com_quat_dev,
                       float *frame_pos_dev, float *frame_quat_dev, void *stream);

/*
 * Contacts found in the last physics step.  Replaces
 * p.getContactPoints(bodyA=uid, linkIndexA=i) and the bodyB=uid variant whose
 * contact[3], contact[4] link indices the reference prints (dancer_env.py:34-39).
 *   pairs_dev  [n_envs][HG_MAX_CONTACTS][2] int32 (link index A, link index B);
 *              base = -1, ground plane = -2, unused slot = -3
 *   counts_dev [n_envs] int32 number of used slots
 *   link_mask_dev [n_envs] uint64, bit b set = body b-1 has a contact
 * All nullable.  HG_EINVAL on a handle created with hg_config.store_contacts = 0 (the step then keeps none of the three).
 */
int hg_get_contact_pairs(hg_sim *sim, int32_t *pairs_dev, int32_t *counts_dev,
                         uint64_t *link_mask_dev, void *stream);

/* Raw state records [n_envs][HG_STATE_SIZE(n_links)] float32 (hg_model.h). */
int hg_get_state(hg_sim *sim, float *state_dev, void *stream);
int hg_set_state(hg_sim *sim, const float *state_dev, void *stream);

/* Sizes a caller needs to allocate buffers: any pointer may be NULL. */
int hg_get_layout(const hg_sim *sim, int32_t *n_envs, int32_t *n_links, int32_t *n_act,
                  int32_t *state_size, int32_t *max_contacts);

/*
 * Episode statistics, the only cross-GPU quantity on the path:
 * stats_dev[8] += { n_envs, sum(reward), sum(done), sum(collision), 0,0,0,0 }
 * reduced over this sim's envs on the device.  The caller all-reduces the
 * 8-vector over ranks (RCCL) whenever it wants a global figure.
 */
int hg_stats_accumulate(hg_sim *sim, const float *reward_dev, const uint8_t *done_dev,
                        const uint8_t *collision_dev, float *stats_dev, void *stream);

/*
 * hg_stats_accumulate for callers that must not allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments
 * and result as hg_stats_accumulate, computed as a fixed-order two-stage sum (no atomics: bit-reproducible
 * from run to run) whose partial results live in a static device array of the library, 16 KB per GPU for
 * the life of the process.  It calls no hipMalloc / hipFree.  Limits: n_envs of the sim <= HG_NOALLOC_MAX_ENVS
 * (HG_EUNSUPPORTED beyond), device index < HG_NOALLOC_MAX_DEVICES.  All calls on one GPU share that GPU's
 * array: calls that may overlap on one GPU (different streams, different sims) race; order them.
 */
#define HG_NOALLOC_MAX_ENVS    (1 << 20)
#define HG_NOALLOC_MAX_DEVICES 64
int hg_stats_accumulate_noalloc(hg_sim *sim, const float *reward_dev, const uint8_t *done_dev,
                                const uint8_t *collision_dev, float *stats_dev, void *stream);

/*
 * Commanded planar base velocity, the three leading action slots that
 * PepperEnv declares and never consumes (pepper_env.py:53 vs :61; qibullet
 * exposes the same motion as PepperVirtual.move(x, y, theta)).
 * cmd_dev [n_envs][3] float32: (vx, vy) in the heading frame of the base link
 * [m/s], yaw rate [rad/s].  The values are copied; from the next hg_step on the
 * base velocity is overwritten before every physics substep with
 * linear (vx, vy, 0) and angular (0, 0, yaw rate), the way qibullet's base
 * controller calls resetBaseVelocity.  cmd_dev = NULL releases the base (the
 * reference's behaviour, and the default).
 */
int hg_set_base_velocity(hg_sim *sim, const float *cmd_dev, void *stream);

/*
 * The same statistics without a reduction launch per step: hg_step keeps, per
 * env, running counters {env steps, sum of the reward output, done count,
 * collision count} (fused into the step kernel's output stage; the reference
 * recomputes such figures in Python from every step's return values,
 * dancer_env.py:41-53).  hg_stats_flush adds their sums over this sim's envs to
 * stats_dev[0..4) and clears them, so a caller that reduces every K steps pays
 * one small launch per K steps.  Counters are not cleared by hg_reset.
 */
int hg_stats_flush(hg_sim *sim, float *stats_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HG_SIM_H */
