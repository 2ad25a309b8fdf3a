/*
 * oracle/classic_control.c — CPU restatement of openai/gym 0.26.2 classic-control step()
 * and of SyncVectorEnv's TimeLimit + autoreset loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gym_amd/ may import, link or execute this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / CPU baseline, never as the thing shipped or measured as product.
 *
 * Parity pin: the reference's tests hold NO golden numbers for these dynamics
 * (SURVEY.md §8c), so this restatement is pinned against outputs of the reference itself,
 * generated in the build container by tests/golden/make_golden.py (reference imported
 * from /root/reference under NumPy 2.2.6 + glibc 2.35) and committed as
 * the .npz files under tests/golden/; tests/test_oracle_golden.py checks every vector.
 *
 * Each function cites the reference file:line it follows (paths relative to the
 * reference root).  Arithmetic is written operation-by-operation in the reference's
 * evaluation order; Python's `x**2` is libm pow(x, 2.0) (this file is compiled with
 * -fno-builtin so gcc cannot rewrite it as x*x) and math.cos/sin are libm cos/sin.
 * Compile with -O2 -ffp-contract=off -fno-builtin (see oracle/Makefile).
 *
 * State layout: struct-of-arrays, state[k*n + i] = component k of env i, fp64.
 * Parameter vectors P[] use the index tables below (the same tables as include/mxv.h,
 * restated here on purpose: the oracle shares no header with the product).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORC_CARTPOLE 0
#define ORC_PENDULUM 1
#define ORC_ACROBOT 2
#define ORC_MOUNTAINCAR 3
#define ORC_MOUNTAINCAR_CONT 4

#define ORC_MAX_PARAMS 12

static const int ORC_STATE_DIM[5] = {4, 2, 4, 2, 2};
static const int ORC_OBS_DIM[5] = {4, 3, 6, 2, 2};

int orc_state_dim(int env_id) { return (env_id >= 0 && env_id < 5) ? ORC_STATE_DIM[env_id] : -1; }
int orc_obs_dim(int env_id) { return (env_id >= 0 && env_id < 5) ? ORC_OBS_DIM[env_id] : -1; }

/* Default parameter vectors = the attribute values set in each env's __init__. */
void orc_default_params(int env_id, double *P) {
    memset(P, 0, sizeof(double) * ORC_MAX_PARAMS);
    switch (env_id) {
    case ORC_CARTPOLE: /* gym/envs/classic_control/cartpole.py:90-102 */
        P[0] = 9.8;                         /* gravity */
        P[1] = 1.0;                         /* masscart */
        P[2] = 0.1;                         /* masspole */
        P[3] = P[2] + P[1];                 /* total_mass = masspole + masscart */
        P[4] = 0.5;                         /* length */
        P[5] = P[2] * P[4];                 /* polemass_length */
        P[6] = 10.0;                        /* force_mag */
        P[7] = 0.02;                        /* tau */
        P[8] = 12 * 2 * M_PI / 360;         /* theta_threshold_radians */
        P[9] = 2.4;                         /* x_threshold */
        P[10] = 0.0;                        /* kinematics_integrator: 0 euler, 1 semi-implicit */
        break;
    case ORC_PENDULUM: /* gym/envs/classic_control/pendulum.py:95-101 */
        P[0] = 8.0;  /* max_speed */
        P[1] = 2.0;  /* max_torque */
        P[2] = 0.05; /* dt */
        P[3] = 10.0; /* g */
        P[4] = 1.0;  /* m */
        P[5] = 1.0;  /* l */
        break;
    case ORC_ACROBOT: /* gym/envs/classic_control/acrobot.py:143-165 */
        P[0] = 0.2;        /* dt */
        P[1] = 1.0;        /* LINK_LENGTH_1 */
        P[2] = 1.0;        /* LINK_LENGTH_2 */
        P[3] = 1.0;        /* LINK_MASS_1 */
        P[4] = 1.0;        /* LINK_MASS_2 */
        P[5] = 0.5;        /* LINK_COM_POS_1 */
        P[6] = 0.5;        /* LINK_COM_POS_2 */
        P[7] = 1.0;        /* LINK_MOI */
        P[8] = 4 * M_PI;   /* MAX_VEL_1 */
        P[9] = 9 * M_PI;   /* MAX_VEL_2 */
        P[10] = 0.0;       /* torque_noise_max */
        P[11] = 0.0;       /* book_or_nips: 0 book, 1 nips */
        break;
    case ORC_MOUNTAINCAR: /* gym/envs/classic_control/mountain_car.py:103-111 */
        P[0] = -1.2;   /* min_position */
        P[1] = 0.6;    /* max_position */
        P[2] = 0.07;   /* max_speed */
        P[3] = 0.5;    /* goal_position */
        P[4] = 0.0;    /* goal_velocity */
        P[5] = 0.001;  /* force */
        P[6] = 0.0025; /* gravity */
        break;
    case ORC_MOUNTAINCAR_CONT: /* gym/envs/classic_control/continuous_mountain_car.py:108-118 */
        P[0] = -1.0;   /* min_action */
        P[1] = 1.0;    /* max_action */
        P[2] = -1.2;   /* min_position */
        P[3] = 0.6;    /* max_position */
        P[4] = 0.07;   /* max_speed */
        P[5] = 0.45;   /* goal_position */
        P[6] = 0.0;    /* goal_velocity */
        P[7] = 0.0015; /* power */
        break;
    default:
        break;
    }
}

/* ------------------------------------------------------------------------------------
 * CartPoleEnv.step — gym/envs/classic_control/cartpole.py:130-188 (fp64, math.cos/sin)
 * ---------------------------------------------------------------------------------- */
static void cartpole_step(const double *P, double *s, int64_t action, float *obs, double *reward,
                          int *terminated) {
    const double gravity = P[0], masspole = P[2], total_mass = P[3], length = P[4];
    const double polemass_length = P[5], force_mag = P[6], tau = P[7];
    const double theta_threshold_radians = P[8], x_threshold = P[9];
    double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
    double force = (action == 1) ? force_mag : -force_mag; /* :135 */
    double costheta = cos(theta);                           /* :136 */
    double sintheta = sin(theta);                           /* :137 */
    double temp = (force + polemass_length * pow(theta_dot, 2.0) * sintheta) / total_mass; /* :141-143 */
    double thetaacc = (gravity * sintheta - costheta * temp) /
                      (length * (4.0 / 3.0 - masspole * pow(costheta, 2.0) / total_mass)); /* :144-146 */
    double xacc = temp - polemass_length * thetaacc * costheta / total_mass;               /* :147 */
    if (P[10] == 0.0) { /* "euler" :149-153 */
        x = x + tau * x_dot;
        x_dot = x_dot + tau * xacc;
        theta = theta + tau * theta_dot;
        theta_dot = theta_dot + tau * thetaacc;
    } else { /* semi-implicit euler :154-158 */
        x_dot = x_dot + tau * xacc;
        x = x + tau * x_dot;
        theta_dot = theta_dot + tau * thetaacc;
        theta = theta + tau * theta_dot;
    }
    s[0] = x; s[1] = x_dot; s[2] = theta; s[3] = theta_dot; /* :160 */
    *terminated = (x < -x_threshold || x > x_threshold || theta < -theta_threshold_radians ||
                   theta > theta_threshold_radians); /* :162-167 */
    *reward = 1.0; /* :169-184 — autoreset means steps_beyond_terminated is always None here */
    obs[0] = (float)x; obs[1] = (float)x_dot; obs[2] = (float)theta; obs[3] = (float)theta_dot; /* :188 */
}

static void cartpole_obs(const double *s, float *obs) { /* cartpole.py:207 */
    obs[0] = (float)s[0]; obs[1] = (float)s[1]; obs[2] = (float)s[2]; obs[3] = (float)s[3];
}

/* ------------------------------------------------------------------------------------
 * PendulumEnv.step — gym/envs/classic_control/pendulum.py:119-139,161-163,270-271
 * state fp64, action fp32; NumPy-2 (NEP 50) promotion: python-float (op) np.float32 → float32.
 * ---------------------------------------------------------------------------------- */
static double np_remainder(double a, double b) { /* numpy npy_remainder for float64 `%` */
    double mod = fmod(a, b);
    if (b == 0.0) return mod;
    if (mod != 0.0) {
        if ((b < 0) != (mod < 0)) mod += b;
    } else {
        mod = copysign(0.0, b);
    }
    return mod;
}

static double angle_normalize(double x) { /* pendulum.py:270-271 */
    return np_remainder(x + M_PI, 2 * M_PI) - M_PI;
}

static void pendulum_obs(const double *s, float *obs) { /* pendulum.py:161-163 */
    obs[0] = (float)cos(s[0]); obs[1] = (float)sin(s[0]); obs[2] = (float)s[1];
}

static void pendulum_step(const double *P, double *s, float a0, float *obs, double *reward,
                          int *terminated) {
    const double max_speed = P[0], max_torque = P[1], dt = P[2], g = P[3], m = P[4], l = P[5];
    double th = s[0], thdot = s[1];
    /* u = np.clip(u, -max_torque, max_torque)[0]  (:127) — float32 array, bounds cast to f32 */
    float lo = (float)(-max_torque), hi = (float)max_torque;
    float u = a0;
    if (u < lo) u = lo;
    if (u > hi) u = hi;
    /* costs (:129): f64 + f64 + float64(float32(0.001) * (u**2 in f32)) */
    float usq = powf(u, 2.0f);
    float uterm = (float)0.001 * usq;
    double an = angle_normalize(th);
    double costs = pow(an, 2.0) + 0.1 * pow(thdot, 2.0) + (double)uterm;
    /* newthdot (:131): 3*g/(2*l) python floats; 3.0/(m*l**2) python float times np.float32 → f32 */
    double A = 3 * g / (2 * l);
    float B = (float)(3.0 / (m * pow(l, 2.0)));
    float Bu = B * u;
    double newthdot = thdot + (A * sin(th) + (double)Bu) * dt;
    /* np.clip(newthdot, -max_speed, max_speed) (:132) = minimum(maximum(x, lo), hi) */
    if (newthdot < -max_speed) newthdot = -max_speed;
    if (newthdot > max_speed) newthdot = max_speed;
    double newth = th + newthdot * dt; /* :133 */
    s[0] = newth; s[1] = newthdot;     /* :135 */
    pendulum_obs(s, obs);
    *reward = -costs; /* :139 */
    *terminated = 0;
}

/* ------------------------------------------------------------------------------------
 * AcrobotEnv.step — gym/envs/classic_control/acrobot.py:196-223, _dsdt 237-277,
 * rk4 418-465, wrap 378-396, bound 399-415, _terminal 232-235, _get_ob 225-230
 * ---------------------------------------------------------------------------------- */
static void acrobot_dsdt(const double *P, const double *sa, double *out) {
    const double m1 = P[3], m2 = P[4], l1 = P[1], lc1 = P[5], lc2 = P[6], I1 = P[7], I2 = P[7];
    const double g = 9.8; /* :245 */
    double a = sa[4];
    double theta1 = sa[0], theta2 = sa[1], dtheta1 = sa[2], dtheta2 = sa[3];
    double d1 = m1 * pow(lc1, 2.0) +
                m2 * (pow(l1, 2.0) + pow(lc2, 2.0) + 2 * l1 * lc2 * cos(theta2)) + I1 + I2; /* :252-257 */
    double d2 = m2 * (pow(lc2, 2.0) + l1 * lc2 * cos(theta2)) + I2;                          /* :258 */
    double phi2 = m2 * lc2 * g * cos(theta1 + theta2 - M_PI / 2.0);                          /* :259 */
    double phi1 = -m2 * l1 * lc2 * pow(dtheta2, 2.0) * sin(theta2) -
                  2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * sin(theta2) +
                  (m1 * lc1 + m2 * l1) * g * cos(theta1 - M_PI / 2) + phi2; /* :260-265 */
    double ddtheta2;
    if (P[11] != 0.0) { /* "nips" :266-269 */
        ddtheta2 = (a + d2 / d1 * phi1 - phi2) / (m2 * pow(lc2, 2.0) + I2 - pow(d2, 2.0) / d1);
    } else { /* "book" :270-275 */
        ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * pow(dtheta1, 2.0) * sin(theta2) - phi2) /
                   (m2 * pow(lc2, 2.0) + I2 - pow(d2, 2.0) / d1);
    }
    double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1; /* :276 */
    out[0] = dtheta1; out[1] = dtheta2; out[2] = ddtheta1; out[3] = ddtheta2; out[4] = 0.0; /* :277 */
}

static double acrobot_wrap(double x, double m, double M) { /* :378-396 */
    double diff = M - m;
    while (x > M) x = x - diff;
    while (x < m) x = x + diff;
    return x;
}

static double acrobot_bound(double x, double m, double M) { /* :399-415 min(max(x, m), M) */
    double t = (m > x) ? m : x; /* python max(x, m): returns m only if m > x */
    return (M < t) ? M : t;     /* python min(t, M): returns M only if M < t */
}

static void acrobot_obs(const double *s, float *obs) { /* :225-230 */
    obs[0] = (float)cos(s[0]); obs[1] = (float)sin(s[0]);
    obs[2] = (float)cos(s[1]); obs[3] = (float)sin(s[1]);
    obs[4] = (float)s[2];      obs[5] = (float)s[3];
}

/* `noise` = the value np_random.uniform(-torque_noise_max, torque_noise_max) returned (:202-205), 0.0 when the attribute is 0 */
static void acrobot_step(const double *P, double *s, int64_t action, double noise, float *obs, double *reward,
                         int *terminated) {
    static const double AVAIL_TORQUE[3] = {-1.0, 0.0, +1.0}; /* :157 */
    double torque = AVAIL_TORQUE[action];                    /* :199 */
    if (P[10] > 0.0) torque += noise;                        /* :202-205 */
    double y0[5] = {s[0], s[1], s[2], s[3], torque};         /* :208 np.append → float64 */
    /* rk4(self._dsdt, s_augmented, [0, self.dt]) :210, body :447-463 */
    double dt = P[0] - 0;     /* t[i+1] - this */
    double dt2 = dt / 2.0;
    double k1[5], k2[5], k3[5], k4[5], y[5];
    int k;
    acrobot_dsdt(P, y0, k1);
    for (k = 0; k < 5; ++k) y[k] = y0[k] + dt2 * k1[k];
    acrobot_dsdt(P, y, k2);
    for (k = 0; k < 5; ++k) y[k] = y0[k] + dt2 * k2[k];
    acrobot_dsdt(P, y, k3);
    for (k = 0; k < 5; ++k) y[k] = y0[k] + dt * k3[k];
    acrobot_dsdt(P, y, k4);
    double dt6 = dt / 6.0;
    double ns[4];
    for (k = 0; k < 4; ++k) ns[k] = y0[k] + dt6 * (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]); /* :463 */
    ns[0] = acrobot_wrap(ns[0], -M_PI, M_PI);  /* :213 */
    ns[1] = acrobot_wrap(ns[1], -M_PI, M_PI);  /* :214 */
    ns[2] = acrobot_bound(ns[2], -P[8], P[8]); /* :215 */
    ns[3] = acrobot_bound(ns[3], -P[9], P[9]); /* :216 */
    s[0] = ns[0]; s[1] = ns[1]; s[2] = ns[2]; s[3] = ns[3];
    *terminated = (-cos(s[0]) - cos(s[1] + s[0]) > 1.0); /* :235 */
    *reward = (!*terminated) ? -1.0 : 0.0;               /* :219 */
    acrobot_obs(s, obs);
}

/* AcrobotEnv.step with the noise draw supplied by the caller (golden replay of np_random.uniform's own values) */
void orc_acrobot_step_noise(const double *P, double *s, int64_t action, double noise, float *obs, double *reward,
                            int *terminated) {
    acrobot_step(P, s, action, noise, obs, reward, terminated);
}

/* ------------------------------------------------------------------------------------
 * MountainCarEnv.step — gym/envs/classic_control/mountain_car.py:127-148 (fp64)
 * ---------------------------------------------------------------------------------- */
static void mountaincar_obs(const double *s, float *obs) { obs[0] = (float)s[0]; obs[1] = (float)s[1]; }

static void mountaincar_step(const double *P, double *s, int64_t action, float *obs, double *reward,
                             int *terminated) {
    const double min_position = P[0], max_position = P[1], max_speed = P[2];
    const double goal_position = P[3], goal_velocity = P[4], force = P[5], gravity = P[6];
    double position = s[0], velocity = s[1];
    velocity = velocity + ((double)(action - 1) * force + cos(3 * position) * (-gravity)); /* :133 */
    if (velocity < -max_speed) velocity = -max_speed; /* np.clip :134 */
    if (velocity > max_speed) velocity = max_speed;
    position = position + velocity; /* :135 */
    if (position < min_position) position = min_position; /* np.clip :136 */
    if (position > max_position) position = max_position;
    if (position == min_position && velocity < 0) velocity = 0; /* :137-138 */
    *terminated = (position >= goal_position && velocity >= goal_velocity); /* :140-142 */
    *reward = -1.0;                                                        /* :143 */
    s[0] = position; s[1] = velocity;                                      /* :145 */
    mountaincar_obs(s, obs);
}

/* ------------------------------------------------------------------------------------
 * Continuous_MountainCarEnv.step — gym/envs/classic_control/continuous_mountain_car.py:142-175
 * executed under NumPy 2 (NEP 50): after the first step the state is a float32 array and
 * the whole update runs in float32 with python-float constants cast to float32; on the
 * first step after reset() the state is a float64 array (:182) and the update runs in
 * float64, except the force term which is float32 arithmetic on the float32 action.
 * `fresh` selects that first-step flow.
 * ---------------------------------------------------------------------------------- */
static void mcc_step(const double *P, double *s, int fresh, float a0, float *obs, double *reward,
                     int *terminated) {
    const double min_action = P[0], max_action = P[1], min_position = P[2], max_position = P[3];
    const double max_speed = P[4], goal_position = P[5], goal_velocity = P[6], power = P[7];
    /* force = min(max(action[0], min_action), max_action)  (:146) — python max/min */
    int clipped_lo = (min_action > (double)a0);
    double tmp = clipped_lo ? min_action : (double)a0;
    int clipped_hi = (max_action < tmp);
    int clipped = clipped_lo || clipped_hi;
    double force_py = clipped_hi ? max_action : min_action; /* python float when clipped */
    int term;
    if (fresh) {
        double position = s[0], velocity = s[1];
        double g = 0.0025 * cos(3 * position); /* :148 */
        double inc;
        if (!clipped) {
            float fp = a0 * (float)power;      /* np.float32 * python float → f32 */
            inc = (double)(fp - (float)g);     /* np.float32 - python float → f32 */
        } else {
            inc = force_py * power - g;        /* python floats */
        }
        velocity = velocity + inc;             /* np.float64 += ... */
        if (velocity > max_speed) velocity = max_speed;    /* :149-150 */
        if (velocity < -max_speed) velocity = -max_speed;  /* :151-152 */
        position = position + velocity;                    /* :153 */
        if (position > max_position) position = max_position; /* :154-155 */
        if (position < min_position) position = min_position; /* :156-157 */
        if (position == min_position && velocity < 0) velocity = 0; /* :158-159 */
        term = (position >= goal_position && velocity >= goal_velocity); /* :162-164 */
        s[0] = (double)(float)position; /* :171 np.array(..., dtype=np.float32) */
        s[1] = (double)(float)velocity;
    } else {
        float position = (float)s[0], velocity = (float)s[1];
        float three_p = 3.0f * position;               /* int * np.float32 → f32 */
        double g = 0.0025 * cos((double)three_p);      /* math.cos → python float */
        float inc;
        if (!clipped) {
            float fp = a0 * (float)power;
            inc = fp - (float)g;
        } else {
            inc = (float)(force_py * power - g);       /* python float, cast when added to f32 */
        }
        velocity = velocity + inc;
        if (velocity > (float)max_speed) velocity = (float)max_speed;
        if (velocity < (float)(-max_speed)) velocity = (float)(-max_speed);
        position = position + velocity;
        if (position > (float)max_position) position = (float)max_position;
        if (position < (float)min_position) position = (float)min_position;
        if (position == (float)min_position && velocity < 0) velocity = 0;
        term = (position >= (float)goal_position && velocity >= (float)goal_velocity);
        s[0] = (double)position;
        s[1] = (double)velocity;
    }
    double reward_ = term ? 100.0 : 0.0;               /* :166-168 */
    reward_ = reward_ - pow((double)a0, 2.0) * 0.1;    /* :169 math.pow(action[0], 2) * 0.1 */
    *reward = reward_;
    *terminated = term;
    obs[0] = (float)s[0]; obs[1] = (float)s[1];        /* :175 returns self.state (float32) */
}

/* ------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
 * SC'11; Random123).  Not part of the reference (which uses PCG64, gym/utils/seeding.py:24-27):
 * north_star mandates Philox on device, so this is the CPU twin of the RNG contract stated in
 * DESIGN.md §RNG, pinned by the Random123 known-answer vectors (tests/test_philox.py).
 * ---------------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    int r;
    for (r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define ORC_STREAM_ACTION 1u
#define ORC_STREAM_RESET 2u
#define ORC_STREAM_ACTION_BITS 6u

/* u in (0,1): (w + 0.5) * 2^-32, exact in fp64 */
static double u01(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }

/* RNG contract, action stream (mirrors MultiDiscrete.sample gym/spaces/multi_discrete.py:123
 * `floor(random*nvec)` and Box.sample gym/spaces/box.py:216-222 `uniform(low, high).astype(f32)`), g = env >> 2,
 * word = out[env & 3], key = action_seed (lo, hi):
 *   Discrete(3) / Box: counter = (g_lo, g_hi, t_lo, (t_hi & 0x0fffffff) | 1<<28), one word per env and step;
 *                      Discrete(n): a = (word * n) >> 32 ;  Box(low, high): a = float32(low + (high-low) * u01(word));
 *   Discrete(2)      : one random bit per step: counter = (g_lo, g_hi, b_lo, (b_hi & 0x0fffffff) | 6<<28), b = t >> 5,
 *                      a = (word >> (t & 31)) & 1. */
static uint32_t action_word(uint64_t action_seed, uint64_t unit, uint32_t stream, uint64_t env) {
    uint32_t ctr[4], key[2], out[4];
    uint64_t g = env >> 2;
    ctr[0] = (uint32_t)g; ctr[1] = (uint32_t)(g >> 32);
    ctr[2] = (uint32_t)unit; ctr[3] = ((uint32_t)(unit >> 32) & 0x0fffffffu) | (stream << 28);
    key[0] = (uint32_t)action_seed; key[1] = (uint32_t)(action_seed >> 32);
    orc_philox4x32_10(ctr, key, out);
    return out[env & 3];
}

/* actions for vector step index t: out_i64[n] (discrete) or out_f32[n] (box). env0 = global index
 * of local env 0 (shards of one logical vector env draw the same numbers as the unsharded one). */
void orc_sample_actions(int env_id, int64_t n, uint64_t env0, uint64_t action_seed, uint64_t t,
                        const double *P, int64_t *out_i64, float *out_f32) {
    int64_t i;
    for (i = 0; i < n; ++i) {
        uint32_t w = (env_id == ORC_CARTPOLE) ? action_word(action_seed, t >> 5, ORC_STREAM_ACTION_BITS, env0 + (uint64_t)i)
                                              : action_word(action_seed, t, ORC_STREAM_ACTION, env0 + (uint64_t)i);
        switch (env_id) {
        case ORC_CARTPOLE: out_i64[i] = (int64_t)((w >> (uint32_t)(t & 31u)) & 1u); break;
        case ORC_ACROBOT:
        case ORC_MOUNTAINCAR: out_i64[i] = (int64_t)(((uint64_t)w * 3u) >> 32); break;
        case ORC_PENDULUM: {
            double lo = -P[1], hi = P[1]; /* Box(-max_torque, max_torque) pendulum.py:113-115 */
            out_f32[i] = (float)(lo + (hi - lo) * u01(w));
            break;
        }
        case ORC_MOUNTAINCAR_CONT: {
            double lo = P[0], hi = P[1]; /* Box(min_action, max_action) continuous_mountain_car.py:132-134 */
            out_f32[i] = (float)(lo + (hi - lo) * u01(w));
            break;
        }
        default: break;
        }
    }
}

/* RNG contract, reset stream: key = per-env seed (lo, hi);
 *   counter = (k, 0, 0, 2<<28): k = 0, 1, 2, ... = how many resets (explicit reset() or autoreset inside a vector step) this
 *   env has had since it was seeded — each env consumes its own stream in order, like its np_random in the reference.
 *   state_k = low_k + (high_k - low_k) * u01(word_k)  — np_random.uniform(low, high) restated
 *   (cartpole.py:202, pendulum.py:154, acrobot.py:188-190 (+ .astype(float32)),
 *    mountain_car.py:160 and continuous_mountain_car.py:182 (velocity = 0)). */
static void reset_state(int env_id, uint64_t seed, uint32_t k, const double *bounds, double *s) {
    uint32_t ctr[4], key[2], w[4];
    ctr[0] = k; ctr[1] = 0u; ctr[2] = 0u; ctr[3] = (ORC_STREAM_RESET << 28);
    key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
    orc_philox4x32_10(ctr, key, w);
    switch (env_id) {
    case ORC_CARTPOLE: {
        int k;
        for (k = 0; k < 4; ++k) s[k] = bounds[0] + (bounds[1] - bounds[0]) * u01(w[k]);
        break;
    }
    case ORC_PENDULUM: /* bounds = (x_init, y_init): low = -high (:152) */
        s[0] = -bounds[0] + (bounds[0] - (-bounds[0])) * u01(w[0]);
        s[1] = -bounds[1] + (bounds[1] - (-bounds[1])) * u01(w[1]);
        break;
    case ORC_ACROBOT: {
        int k;
        for (k = 0; k < 4; ++k)
            s[k] = (double)(float)(bounds[0] + (bounds[1] - bounds[0]) * u01(w[k]));
        break;
    }
    case ORC_MOUNTAINCAR:
    case ORC_MOUNTAINCAR_CONT:
        s[0] = bounds[0] + (bounds[1] - bounds[0]) * u01(w[0]);
        s[1] = 0.0;
        break;
    default: break;
    }
}

static void env_obs(int env_id, const double *s, float *obs) {
    switch (env_id) {
    case ORC_CARTPOLE: cartpole_obs(s, obs); break;
    case ORC_PENDULUM: pendulum_obs(s, obs); break;
    case ORC_ACROBOT: acrobot_obs(s, obs); break;
    default: mountaincar_obs(s, obs); break;
    }
}

void orc_default_reset_bounds(int env_id, double *bounds) {
    switch (env_id) {
    case ORC_CARTPOLE: bounds[0] = -0.05; bounds[1] = 0.05; break;       /* cartpole.py:199-201 */
    case ORC_PENDULUM: bounds[0] = M_PI; bounds[1] = 1.0; break;          /* pendulum.py:14-15 */
    case ORC_ACROBOT: bounds[0] = -0.1; bounds[1] = 0.1; break;           /* acrobot.py:185-187 */
    default: bounds[0] = -0.6; bounds[1] = -0.4; break;                   /* mountain_car.py:159 */
    }
}

/* Explicit reset of envs [0,n) (mask NULL = all): SyncVectorEnv.reset_wait
 * gym/vector/sync_vector_env.py:90-129 + TimeLimit.reset gym/wrappers/time_limit.py:58-68.
 * seeds: per-env 64-bit seeds (NULL → base_seed + env0 + i, sync_vector_env.py:106-107). */
void orc_vec_reset(int env_id, int64_t n, uint64_t env0, const uint64_t *seeds, uint64_t base_seed,
                   uint32_t *episodes, const double *bounds, const uint8_t *mask, double *state,
                   int32_t *elapsed, float *obs) {
    int S = ORC_STATE_DIM[env_id], O = ORC_OBS_DIM[env_id];
    int64_t i;
    int k;
    for (i = 0; i < n; ++i) {
        double s[4];
        if (mask && !mask[i]) continue;
        uint64_t seed = seeds ? seeds[i] : base_seed + env0 + (uint64_t)i;
        reset_state(env_id, seed, episodes[i], bounds, s);
        episodes[i] += 1;
        for (k = 0; k < S; ++k) state[(int64_t)k * n + i] = s[k];
        elapsed[i] = 0;
        if (obs) env_obs(env_id, s, obs + i * O);
    }
}

/* One env, one step of dynamics only (no TimeLimit / autoreset). actions: int64 or float32. */
static void env_step(int env_id, const double *P, double *s, int fresh, int64_t ai, float af, double noise,
                     float *obs, double *reward, int *term) {
    switch (env_id) {
    case ORC_CARTPOLE: cartpole_step(P, s, ai, obs, reward, term); break;
    case ORC_PENDULUM: pendulum_step(P, s, af, obs, reward, term); break;
    case ORC_ACROBOT: acrobot_step(P, s, ai, noise, obs, reward, term); break;
    case ORC_MOUNTAINCAR: mountaincar_step(P, s, ai, obs, reward, term); break;
    default: mcc_step(P, s, fresh, af, obs, reward, term); break;
    }
}

/* SyncVectorEnv.step_wait — gym/vector/sync_vector_env.py:135-169 with TimeLimit.step
 * gym/wrappers/time_limit.py:50-54 inlined per sub-env.
 *   autoreset != 0: on terminated|truncated the env is reset from its Philox reset stream
 *     (draw number episodes[i], then episodes[i] += 1), obs row = reset obs, final_obs row = terminal obs, final_mask = 1.
 *   autoreset == 0: dynamics + TimeLimit only (used by "external reset" trajectory tests).
 * Returns the number of envs whose discrete action was out of range (those envs are not
 * stepped): Discrete.contains assert, cartpole.py:131-132 / mountain_car.py:128-130. */
int64_t orc_vec_step_beyond(int env_id, int64_t n, uint64_t env0, const double *P, int max_episode_steps,
                            int autoreset, const uint64_t *seeds, uint64_t base_seed, uint64_t t, uint32_t *episodes,
                            const double *bounds, const int64_t *act_i64, const float *act_f32,
                            double *state, int32_t *elapsed, float *obs, double *reward,
                            uint8_t *terminated, uint8_t *truncated, float *final_obs,
                            uint8_t *final_mask, uint8_t *beyond);

int64_t orc_vec_step(int env_id, int64_t n, uint64_t env0, const double *P, int max_episode_steps,
                     int autoreset, const uint64_t *seeds, uint64_t base_seed, uint64_t t, uint32_t *episodes,
                     const double *bounds, const int64_t *act_i64, const float *act_f32,
                     double *state, int32_t *elapsed, float *obs, double *reward,
                     uint8_t *terminated, uint8_t *truncated, float *final_obs,
                     uint8_t *final_mask) {
    return orc_vec_step_beyond(env_id, n, env0, P, max_episode_steps, autoreset, seeds, base_seed, t, episodes, bounds, act_i64,
                               act_f32, state, elapsed, obs, reward, terminated, truncated, final_obs, final_mask, NULL);
}

/* The same with CartPole's `steps_beyond_terminated` bookkeeping (cartpole.py:169-184) for envs that are stepped on after they
 * terminated — only possible without autoreset: beyond[i] != 0 = env i has terminated before (steps_beyond_terminated is not None);
 * the step in which the pole falls still pays 1.0 and sets the mark, every later step that is (still) terminated pays 0.0.  The caller
 * clears beyond[i] when it resets env i (cartpole.py:205).  beyond == NULL: no bookkeeping (autoreset makes the branch unreachable). */
int64_t orc_vec_step_beyond(int env_id, int64_t n, uint64_t env0, const double *P, int max_episode_steps,
                            int autoreset, const uint64_t *seeds, uint64_t base_seed, uint64_t t, uint32_t *episodes,
                            const double *bounds, const int64_t *act_i64, const float *act_f32,
                            double *state, int32_t *elapsed, float *obs, double *reward,
                            uint8_t *terminated, uint8_t *truncated, float *final_obs,
                            uint8_t *final_mask, uint8_t *beyond) {
    int S = ORC_STATE_DIM[env_id], O = ORC_OBS_DIM[env_id];
    int nact = (env_id == ORC_CARTPOLE) ? 2 : 3;
    int discrete = (env_id == ORC_CARTPOLE || env_id == ORC_ACROBOT || env_id == ORC_MOUNTAINCAR);
    int64_t bad = 0, i;
    int k;
    for (i = 0; i < n; ++i) {
        double s[4] = {0, 0, 0, 0};
        float o[6];
        double rew = 0.0;
        int term = 0, trunc = 0;
        int64_t ai = 0;
        float af = 0.0f;
        for (k = 0; k < S; ++k) s[k] = state[(int64_t)k * n + i];
        if (discrete) {
            ai = act_i64[i];
            if (ai < 0 || ai >= nact) { ++bad; continue; }
        } else {
            af = act_f32[i];
        }
        double noise = 0.0;
        if (env_id == ORC_ACROBOT && P[10] > 0.0) { /* RNG contract, step-noise stream: key = env seed, ctr = (t, 0, 4<<28), word x */
            uint32_t ctr[4] = {(uint32_t)t, (uint32_t)(t >> 32), 0u, 4u << 28}, key[2], w[4];
            uint64_t seed = seeds ? seeds[i] : base_seed + env0 + (uint64_t)i;
            key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
            orc_philox4x32_10(ctr, key, w);
            noise = -P[10] + (P[10] - (-P[10])) * u01(w[0]);       /* np_random.uniform(low, high) = low + (high-low)*u */
        }
        env_step(env_id, P, s, elapsed[i] == 0, ai, af, noise, o, &rew, &term);
        if (env_id == ORC_CARTPOLE && beyond && term) {            /* cartpole.py:171-184 */
            if (beyond[i]) rew = 0.0;                              /* steps_beyond_terminated was not None: reward = 0.0 */
            else beyond[i] = 1;                                    /* "Pole just fell!": reward stays 1.0 */
        }
        elapsed[i] += 1;                                           /* time_limit.py:51 */
        if (max_episode_steps > 0 && elapsed[i] >= max_episode_steps) trunc = 1; /* :53-54 */
        reward[i] = rew;
        terminated[i] = (uint8_t)term;
        truncated[i] = (uint8_t)trunc;
        if (final_mask) final_mask[i] = 0;
        if (autoreset && (term || trunc)) {                        /* sync_vector_env.py:152-156 */
            if (final_obs) for (k = 0; k < O; ++k) final_obs[i * O + k] = o[k];
            if (final_mask) final_mask[i] = 1;
            uint64_t seed = seeds ? seeds[i] : base_seed + env0 + (uint64_t)i;
            reset_state(env_id, seed, episodes[i], bounds, s);
            episodes[i] += 1;
            elapsed[i] = 0;
            env_obs(env_id, s, o);
        }
        for (k = 0; k < O; ++k) obs[i * O + k] = o[k];
        for (k = 0; k < S; ++k) state[(int64_t)k * n + i] = s[k];
    }
    return bad;
}

/* Random-action rollout of K vector steps starting at step index t0; returns a checksum-friendly
 * summary (sum of rewards, number of dones) and leaves state/elapsed advanced.  Used as the CPU
 * baseline leg of bench.py and by the full-size property tests.  Scratch buffers are caller-owned. */
void orc_rollout(int env_id, int64_t n, uint64_t env0, const double *P, int max_episode_steps,
                 uint64_t base_seed, uint64_t action_seed, uint64_t t0, uint32_t *episodes, int K, const double *bounds,
                 double *state, int32_t *elapsed, int64_t *act_i64, float *act_f32, float *obs,
                 double *reward, uint8_t *terminated, uint8_t *truncated, double *sum_reward,
                 int64_t *num_done) {
    int step;
    int64_t i;
    double sr = 0.0;
    int64_t nd = 0;
    for (step = 0; step < K; ++step) {
        uint64_t t = t0 + (uint64_t)step;
        orc_sample_actions(env_id, n, env0, action_seed, t, P, act_i64, act_f32);
        orc_vec_step(env_id, n, env0, P, max_episode_steps, 1, NULL, base_seed, t, episodes, bounds, act_i64,
                     act_f32, state, elapsed, obs, reward, terminated, truncated, NULL, NULL);
        for (i = 0; i < n; ++i) {
            sr += reward[i];
            nd += (terminated[i] | truncated[i]) ? 1 : 0;
        }
    }
    *sum_reward = sr;
    *num_done = nd;
}
