/*
 * hg_model.h -- flat (structure-of-arrays) robot description and world
 * parameters shared by the HIP library (humanoid_gym_amd/csrc) and by the CPU
 * oracle (oracle/).  Plain C, fixed-size arrays, so that a ctypes.Structure can
 * mirror it field by field.
 *
 * What it replaces in the reference: the in-memory multibody that
 * `p.loadURDF(...)` builds for `DancerEnv.reset`
 * (/root/reference/humanoid_gym/envs/dancer_env.py:58-65) and that
 * `spawnNao/spawnPepper/spawnRomeo` build for the qibullet envs
 * (nao_env.py:20, pepper_env.py:18, romeo_env.py:17).
 *
 * Conventions (all doubles; the HIP side converts to float once, at create):
 *  - every moving link has exactly one revolute degree of freedom; links are
 *    numbered in depth-first pre-order (parents before children), which is
 *    also PyBullet's joint index order for URDFs without fixed joints;
 *  - the INTERNAL frame of a body sits at its centre of mass with axes along
 *    the principal axes of inertia (what Bullet's URDF importer does), so the
 *    rigid-body inertia is diag(inertia[0..2]) + mass;
 *  - quaternions are xyzw (PyBullet order) and rotate body vectors to world.
 */
#ifndef HG_MODEL_H
#define HG_MODEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HG_MAX_LINKS    62   /* moving links == joint dofs (one lane each, plus the base; 64-bit link masks) */
#define HG_MAX_SPHERES  1024 /* collision primitives: points (radius 0), spheres, sphere-swept segments */
#define HG_MAX_PAIRS    1024 /* self-collision sphere pairs                        */
#define HG_MAX_CONTACTS 12   /* contact slots per env per physics step             */
#define HG_MAX_ROWS     128  /* n_links + 3*max_contacts: one row per lane up to 64, two per lane beyond   */
/* A constraint row whose denominator J M^-1 J^T is not above this is DISABLED (impulse 0), as
 * btMultiBodyConstraint::fillMultiBodyConstraint does with SIMD_EPSILON ("disable the constraint row to handle
 * singularity/redundant constraint") [UPSTREAM-RECALL].  Such rows exist: two collision spheres centred on the
 * joint axes of two root children overlap for ever and no coordinate of the robot moves one against the other
 * (J = 0 identically).  What the arithmetic then sees is rounding noise of J: ~1e-13 / kg in FP32, ~1e-30 in FP64.
 * The value is FLT_EPSILON -- Bullet's single-precision SIMD_EPSILON, 8e6 kg of effective mass -- for the FP32
 * kernel AND the FP64 oracle, so that both disable the same rows (a double-precision Bullet uses DBL_EPSILON; no
 * row of a real robot lies between the two). */
#ifndef HG_ROW_DEN_EPS
#define HG_ROW_DEN_EPS  1.1920929e-7
#endif

typedef struct hg_model {
    int32_t n_links;                       /* n: moving links / dofs                */
    int32_t n_spheres;                     /* collision spheres                     */
    int32_t n_pairs;                       /* self-collision candidate pairs        */
    int32_t n_act;                         /* controlled joints (env joint_names)   */

    int32_t parent[HG_MAX_LINKS];          /* -1 = base                             */
    int32_t act_joint[HG_MAX_LINKS];       /* action/obs slot -> dof index          */

    /* per-link kinematics, internal frames */
    double  R0[HG_MAX_LINKS][9];           /* parent frame -> link frame at q = 0   */
    double  axis[HG_MAX_LINKS][3];         /* joint axis, link frame (unit)         */
    double  d[HG_MAX_LINKS][3];            /* parent COM -> joint pivot, parent frm */
    double  e[HG_MAX_LINKS][3];            /* joint pivot -> link COM, link frame   */

    /* per-link inertia and joint parameters */
    double  mass[HG_MAX_LINKS];
    double  inertia[HG_MAX_LINKS][3];      /* principal moments about the COM       */
    double  joint_damping[HG_MAX_LINKS];   /* URDF <dynamics damping>               */
    double  lower[HG_MAX_LINKS];           /* position limits (targets are clamped  */
    double  upper[HG_MAX_LINKS];           /*   to them when clamp_targets != 0)    */
    double  max_force[HG_MAX_LINKS];       /* motor force bound (N m)               */
    double  max_vel[HG_MAX_LINKS];         /* motor target-velocity clamp (rad/s)   */
    double  q_init[HG_MAX_LINKS];          /* reset posture and initial targets     */
    int32_t joint_limited[HG_MAX_LINKS];   /* 1: URDF `revolute` with lower < upper -> Bullet adds a
                                              btMultiBodyJointLimitConstraint; 0: `continuous`     */

    /* base body */
    double  base_mass;
    double  base_inertia[3];
    double  base_pos0[3];                  /* reset pose of the base INTERNAL frame */
    double  base_quat0[4];

    /* URDF link frame expressed in the internal frame, index 0 = base, i+1 = link i */
    double  frame_pos[HG_MAX_LINKS + 1][3];
    double  frame_quat[HG_MAX_LINKS + 1][4];

    /* collision spheres, internal frame of their link */
    int32_t sphere_link[HG_MAX_SPHERES];   /* -1 = base                             */
    int32_t sphere_flags[HG_MAX_SPHERES];  /* bit0: tested against the ground plane,
                                              bit1: takes part in self-collision pairs */
    double  sphere_pos[HG_MAX_SPHERES][3];
    double  sphere_radius[HG_MAX_SPHERES];
    double  sphere_half[HG_MAX_SPHERES][3];/* half axis of a sphere-swept segment (capsule): the primitive is the set of points
                                              within `radius` of the segment pos -+ half; 0 = a sphere.  Segments take part
                                              in link-link pairs only (bit1); ground points / spheres have half = 0        */

    /* self-collision candidate pairs (indices into the sphere arrays) */
    int32_t pair_a[HG_MAX_PAIRS];
    int32_t pair_b[HG_MAX_PAIRS];
} hg_model;

typedef struct hg_config {
    double  dt;                /* physics step, 1/240 (dancer_env.py:65)              */
    double  gravity[3];        /* (0,0,-10) dancer_env.py:63; (0,0,-9.81) qibullet    */
    double  kp;                /* POSITION_CONTROL positionGain, PyBullet default 0.1 */
    double  kd;                /* velocityGain, PyBullet default 1.0                  */
    double  contact_erp;       /* 0.2                                                 */
    double  friction;          /* combined lateral friction coefficient               */
    double  contact_margin;    /* manifold distance threshold, 0.02                   */
    double  lin_damping;       /* Bullet multibody default 0.04                       */
    double  ang_damping;       /* Bullet multibody default 0.04                       */
    double  max_coord_velocity;/* btMultiBody m_maxCoordinateVelocity, 100: every velocity coordinate
                                  is clamped to +-this after each velocity update; <= 0: off */
    int32_t solver_iters;      /* 150 (dancer_env.py:64) / 50                         */
    int32_t max_contacts;      /* <= HG_MAX_CONTACTS                                  */
    int32_t enable_ground;     /* plane z = 0                                         */
    int32_t enable_self;       /* sphere-pair self collision                          */
    int32_t clamp_targets;     /* qibullet setAngles clamps to the limits             */
    int32_t enable_limits;     /* joint-limit constraint rows for joint_limited joints */
    double  limit_max_impulse; /* btMultiBodyConstraint m_maxAppliedImpulse, 100      */
    double  residual_threshold;/* > 0: the solver leaves its sweep loop after the first sweep whose largest squared
                                  velocity residual (impulse change x J M^-1 J^T of the row) is <= this, as Bullet's
                                  m_leastSquaresResidualThreshold does (PyBullet sets 1e-7 [UPSTREAM-RECALL]; the
                                  reference sets only the iteration count, dancer_env.py:64).  0: all solver_iters sweeps */
    int32_t store_torques;     /* 1: the applied motor torques of every physics step are kept for hg_get_joint_torques
                                  (getJointState()[3], dancer_env.py:24); 0: not stored, the getter fails             */
    int32_t store_contacts;    /* 1: the contact link pairs, their count and the per-env link mask of the last physics step
                                  are kept for hg_get_contact_pairs (getContactPoints, dancer_env.py:32-39); 0: not
                                  stored, the getter fails (the `collision` flag of hg_step is written either way)  */
} hg_config;

/* State record of one env: HG_STATE_BASE + 3*n values (double in the oracle,
 * float in the HIP library), env-major:
 *   [0:3) base COM position (world)   [3:7) base quaternion xyzw (body->world)
 *   [7:10) base COM linear velocity   [10:13) base angular velocity (world)
 *   [13:13+n) q   [13+n:13+2n) qd   [13+2n:13+3n) motor targets            */
#define HG_STATE_BASE 13
#define HG_STATE_SIZE(n) (HG_STATE_BASE + 3 * (n))

#ifdef __cplusplus
}
#endif
#endif /* HG_MODEL_H */
