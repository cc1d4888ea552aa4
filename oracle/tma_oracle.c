/*
 * tma_oracle.c -- CPU restatement (plain C) of the reference's vector-env hot path.
 * TEST INFRASTRUCTURE ONLY (see tma_oracle.h).  Build: make -C oracle  (gcc, -ffp-contract=off).
 *
 * Each function cites the reference file:line it restates (paths relative to /root/reference/).
 */
#include "tma_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* =========================================================================================
 * numpy legacy global RNG.  np.random.seed(int) / shuffle / choice / randint / uniform are
 * numpy 2.3.5 (backend/uv.lock:986-987) `RandomState` legacy paths over MT19937; call sites:
 * backend/mlagents/envs.py:117-119, backend/examples/gridworld.py:45,50,
 * backend/examples/push.py:41,46, backend/examples/ball3d.py:49-57.
 * ======================================================================================= */
void orc_mt_seed(orc_mt *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}

static void mt_twist(orc_mt *s) {
    uint32_t *mt = s->mt;
    for (int k = 0; k < 624; k++) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s->idx = 0;
}

uint32_t orc_mt_u32(orc_mt *s) {
    if (s->idx >= 624) mt_twist(s);
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* legacy rk_interval (masked rejection); used by shuffle, choice([0,1]) and randint(0,6) */
uint32_t orc_mt_interval(orc_mt *s, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    do {
        v = orc_mt_u32(s) & mask;
    } while (v > max);
    return v;
}

double orc_mt_double(orc_mt *s) {
    uint32_t a = orc_mt_u32(s) >> 5, b = orc_mt_u32(s) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

static double mt_uniform(orc_mt *s, double lo, double hi) {
    double scale = hi - lo;
    double d = orc_mt_double(s);
    double p = scale * d; /* two roundings, no fma (build uses -ffp-contract=off) */
    return lo + p;
}

void orc_mt_raw(uint32_t seed, uint32_t *out, int n) {
    orc_mt s;
    orc_mt_seed(&s, seed);
    for (int i = 0; i < n; i++) out[i] = orc_mt_u32(&s);
}

static void mt_shuffle_idx(orc_mt *s, int32_t *x, int n) {
    for (int i = n - 1; i >= 1; i--) {
        uint32_t j = orc_mt_interval(s, (uint32_t)i);
        int32_t t = x[i];
        x[i] = x[j];
        x[j] = t;
    }
}

void orc_mt_shuffle(uint32_t seed, int32_t *perm, int n, int32_t *next_interval_max_inout) {
    orc_mt s;
    orc_mt_seed(&s, seed);
    for (int i = 0; i < n; i++) perm[i] = i;
    mt_shuffle_idx(&s, perm, n);
    if (next_interval_max_inout) *next_interval_max_inout = (int32_t)orc_mt_interval(&s, (uint32_t)*next_interval_max_inout);
}

void orc_mt_uniform(uint32_t seed, double lo, double hi, double *out, int n) {
    orc_mt s;
    orc_mt_seed(&s, seed);
    for (int i = 0; i < n; i++) out[i] = mt_uniform(&s, lo, hi);
}

/* =========================================================================================
 * task metadata: backend/mlagents/envs.py:35,38-44,166-199 ; examples/*.py constants
 * ======================================================================================= */
#define CRAWLER_NJ 20
#define CRAWLER_OBS 172
#define CRAWLER_STATE (2 * CRAWLER_NJ + CRAWLER_NJ + 8 + 1) /* q, qd, prev action, root(8), steps */
/* ORC_ANT: the same articulated chain with 8 joints and observations in gymnasium Ant-v5's order (the SHAPES the reference's `ant` task
 * builds, backend/mlagents/envs.py:274-277: Box(105,) observations, Box(-1,1,(8,)) torques); dynamics build-defined, parity unpinned */
#define ANT_NJ 8
#define ANT_OBS 105
#define ANT_STATE (3 * ANT_NJ + 8 + 1)

int orc_obs_dim(int task) {
    static const int d[] = {21, 4, 6, 4, CRAWLER_OBS, 4, 7, 45, 16, ANT_OBS};
    return d[task];
}
int orc_num_actions(int task) {
    static const int d[] = {3, 5, 5, 5, 0, 4, 3, 3, 5, 0};
    return d[task];
}
int orc_act_dim(int task) { return task == ORC_CRAWLER ? CRAWLER_NJ : (task == ORC_ANT ? ANT_NJ : 1); }
int orc_state_dim(int task) {
    static const int d[] = {2, 8, 8, 6, CRAWLER_STATE, 4, 10, 46, 14, ANT_STATE};
    return d[task];
}
int orc_max_episode_steps(int task) {
    static const int d[] = {50, 100, 200, 120, 1000, 150, 2000, 2000, 4000, 1000};
    return d[task];
}

/* position of the step counter in the flat state vector */
static int orc_steps_index(int task) {
    static const int d[] = {1, 7, 6, 5, 3 * CRAWLER_NJ + 8, 3, 9, 5, 13, 3 * ANT_NJ + 8};
    return d[task];
}

uint32_t orc_episode_seed(uint32_t base, uint32_t env_index, uint32_t episode) {
    return base + env_index + episode * ORC_EP_STRIDE;
}

uint32_t orc_mix32(uint32_t seed, uint32_t i, uint32_t t) {
    uint32_t x = (seed * 0x9E3779B1u) ^ (i * 0x85EBCA77u) ^ (t * 0xC2B2AE3Du);
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

void orc_action_tape(uint32_t tape_seed, int n_envs, int env_offset, int T, int n_actions, int32_t *out) {
    for (int t = 0; t < T; t++)
        for (int i = 0; i < n_envs; i++)
            out[(size_t)t * n_envs + i] = (int32_t)(orc_mix32(tape_seed, (uint32_t)(i + env_offset), (uint32_t)t) % (uint32_t)n_actions);
}

/* =========================================================================================
 * Basic -- backend/mlagents/envs.py:17-27,48-84.   state = [position, steps]
 * ======================================================================================= */
static void basic_obs(const double *st, float *obs) {
    for (int k = 0; k < 21; k++) obs[k] = 0.0f;
    int p = (int)st[0];
    if (p < 0) p = 0;
    if (p > 20) p = 20;
    obs[p] = 1.0f;
}
static void basic_reset(double *st) { /* envs.py:54-57, no options -> START_POS */
    st[0] = 10;
    st[1] = 0;
}
/* envs.py:60-81; returns terminated in *done, truncated separately through basic rule in vec step */
static void basic_step(double *st, int a, float *obs, double *reward, int *terminated) {
    static const int delta[3] = {-1, 0, 1};
    int p = (int)st[0] + delta[a];
    if (p < 0) p = 0;
    if (p > 20) p = 20;
    st[0] = p;
    st[1] += 1;
    double r = -0.01;
    int term = 0;
    if (p == 7) {
        r += 0.1;
        term = 1;
    } else if (p == 17) {
        r += 1.0;
        term = 1;
    }
    *reward = r;
    *terminated = term;
    basic_obs(st, obs);
}

/* =========================================================================================
 * GridWorld -- backend/examples/gridworld.py:14-30,40-95.  state = [ax,ay,gx,gy,rx,ry,type,steps]
 * ======================================================================================= */
static void grid_obs(const double *st, float *obs) { /* gridworld.py:55-64 */
    int gt = (int)st[6];
    double gx = gt == 0 ? st[2] : st[4], gy = gt == 0 ? st[3] : st[5];
    obs[0] = (float)((gx - st[0]) / 4.0);
    obs[1] = (float)((gy - st[1]) / 4.0);
    obs[2] = gt == 0 ? 1.0f : 0.0f;
    obs[3] = gt == 0 ? 0.0f : 1.0f;
}
static void grid_reset(orc_mt *rng, double *st) { /* gridworld.py:40-52 */
    int32_t cells[25];
    for (int i = 0; i < 25; i++) cells[i] = i; /* (x,y) x-major: cell = x*5+y */
    mt_shuffle_idx(rng, cells, 25);
    st[0] = cells[0] / 5;
    st[1] = cells[0] % 5;
    st[2] = cells[1] / 5;
    st[3] = cells[1] % 5;
    st[4] = cells[2] / 5;
    st[5] = cells[2] % 5;
    st[6] = (double)orc_mt_interval(rng, 1); /* np.random.choice([0,1]) */
    st[7] = 0;
}
static int clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static const int GRID_DX[5] = {0, 0, 0, -1, 1};
static const int GRID_DY[5] = {0, 1, -1, 0, 0};
static void grid_step(double *st, int a, float *obs, double *reward, int *done) { /* gridworld.py:67-95 */
    int ax = clipi((int)st[0] + GRID_DX[a], 0, 4), ay = clipi((int)st[1] + GRID_DY[a], 0, 4);
    st[0] = ax;
    st[1] = ay;
    st[7] += 1;
    double r = -0.01;
    int d = 0, gt = (int)st[6];
    if (ax == (int)st[2] && ay == (int)st[3]) {
        r = gt == 0 ? 1.0 : -1.0;
        d = 1;
    } else if (ax == (int)st[4] && ay == (int)st[5]) {
        r = gt == 1 ? 1.0 : -1.0;
        d = 1;
    }
    if (st[7] >= 100) d = 1;
    *reward = r;
    *done = d;
    grid_obs(st, obs);
}

/* =========================================================================================
 * Push -- backend/examples/push.py:10-24,39-125.   state = [ax,ay,bx,by,gx,steps]
 * ======================================================================================= */
static void push_obs(const double *st, float *obs) { /* push.py:53-59 */
    obs[0] = (float)((st[2] - st[0]) / 5.0);
    obs[1] = (float)((st[3] - st[1]) / 5.0);
    obs[2] = (float)((st[4] - st[2]) / 5.0);
    obs[3] = (float)((5.0 - st[3]) / 5.0);
}
static void push_reset(orc_mt *rng, double *st) { /* push.py:39-50 */
    int32_t cells[36];
    for (int i = 0; i < 36; i++) cells[i] = i;
    mt_shuffle_idx(rng, cells, 36);
    st[0] = cells[0] / 6;
    st[1] = cells[0] % 6;
    st[2] = cells[1] / 6;
    st[3] = cells[1] % 6;
    st[4] = (double)orc_mt_interval(rng, 5); /* np.random.randint(0, 6) */
    st[5] = 0;
}
static void push_step(double *st, int a, float *obs, double *reward, int *done) { /* push.py:62-125 */
    int dx = GRID_DX[a], dy = GRID_DY[a];
    int ax = (int)st[0], ay = (int)st[1], bx = (int)st[2], by = (int)st[3], gx = (int)st[4];
    int nax = clipi(ax + dx, 0, 5), nay = clipi(ay + dy, 0, 5);
    int nbx = bx, nby = by;
    int prev_bg = abs(gx - bx) + abs(5 - by);
    int prev_ab = abs(bx - ax) + abs(by - ay);
    double r = -0.01;
    int d = 0, invalid = 0;
    if (nax == bx && nay == by) {
        int tx = bx + dx, ty = by + dy;
        if (0 <= tx && tx < 6 && 0 <= ty && ty < 6) {
            nbx = tx;
            nby = ty;
        } else {
            nax = ax;
            nay = ay;
            invalid = 1;
        }
    }
    st[0] = nax;
    st[1] = nay;
    st[2] = nbx;
    st[3] = nby;
    st[5] += 1;
    int bg = abs(gx - nbx) + abs(5 - nby);
    int ab = abs(nbx - nax) + abs(nby - nay);
    double t1 = 0.05 * (double)(prev_ab - ab);
    r = r + t1;
    double t2 = 0.3 * (double)(prev_bg - bg);
    r = r + t2;
    if (invalid) r = r - 0.05;
    if (nby == 5) {
        r = 1.0;
        d = 1;
    }
    if (st[5] >= 120) d = 1;
    *reward = r;
    *done = d;
    push_obs(st, obs);
}

/* =========================================================================================
 * Ball3D -- backend/examples/ball3d.py:10-38,47-113.
 * state = [rot0,rot1,pos0,pos1,vel0,vel1,steps,first]; `first`=1 while rot is still the float32
 * array produced by reset (NumPy NEP-50 promotion recipe).
 * ======================================================================================= */
#define B3_MAX_TILT 0.4363323129985824   /* np.deg2rad(25.0) */
#define B3_TILT_DELTA 0.05235987755982989 /* np.deg2rad(3.0)  */
static void ball_obs(const double *st, float *obs) { /* ball3d.py:61-72 */
    for (int k = 0; k < 6; k++) obs[k] = (float)st[k];
}
static void ball_reset(orc_mt *rng, double *st) { /* ball3d.py:47-59 */
    double half = B3_MAX_TILT * 0.5;
    for (int k = 0; k < 2; k++) st[k] = (double)(float)mt_uniform(rng, -half, half);
    for (int k = 0; k < 2; k++) st[2 + k] = (double)(float)mt_uniform(rng, -1.5, 1.5);
    for (int k = 0; k < 2; k++) st[4 + k] = (double)(float)mt_uniform(rng, -1.0, 1.0);
    st[6] = 0;
    st[7] = 1;
}
static void ball_step(double *st, int a, float *obs, double *reward, int *done) { /* ball3d.py:74-113 */
    static const double DEL[5][2] = {{B3_TILT_DELTA, 0.0}, {-B3_TILT_DELTA, 0.0}, {0.0, B3_TILT_DELTA}, {0.0, -B3_TILT_DELTA}, {0.0, 0.0}};
    int first = st[7] != 0.0;
    float pos[2], vel[2];
    for (int k = 0; k < 2; k++) {
        double r = st[k] + DEL[a][k]; /* :77  rot += delta  (float32 in-place on the first step) */
        if (first) r = (double)(float)r;
        if (r < -B3_MAX_TILT) r = -B3_MAX_TILT; /* :78 np.clip with float64 bounds -> float64 */
        if (r > B3_MAX_TILT) r = B3_MAX_TILT;
        st[k] = r;
        double acc = 9.81 * sin(r);                 /* :81-82 */
        double accdt = acc * 0.02;
        float v = (float)((double)(float)st[4 + k] + accdt); /* :83-84 */
        v = v * 0.98f;                              /* :87 float32 * weak python float */
        float step = v * 0.02f;                     /* :90 */
        float p = (float)st[2 + k] + step;
        vel[k] = v;
        pos[k] = p;
        st[4 + k] = (double)v;
        st[2 + k] = (double)p;
    }
    st[6] += 1;
    st[7] = 0;
    int off = (fabsf(pos[0]) > 3.0f) || (fabsf(pos[1]) > 3.0f); /* :96-98 */
    int timeout = st[6] >= 200;
    int d = off || timeout;
    float s0 = pos[0] * pos[0], s1 = pos[1] * pos[1];
    float norm = sqrtf(s0 + s1);                  /* np.linalg.norm on float32[2] */
    float q = norm / 3.0f;
    float rew = 1.0f - q;                         /* :104 */
    if (d) {
        rew = -1.0f;
        if (timeout && !off) rew = 1.0f;
    }
    float pen = -0.02f * norm;                    /* :110 */
    rew = rew + pen;
    *reward = (double)rew;
    *done = d;
    ball_obs(st, obs);
}


/* =========================================================================================
 * WallJump -- backend/examples/walljump.py:14-20,33-98.   state = [agent_x, in_air, wall_height, steps]
 * ======================================================================================= */
static void wj_obs(const double *st, float *obs) { /* walljump.py:48-53 */
    obs[0] = (float)((19.0 - st[0]) / 19.0);
    obs[1] = (float)((10.0 - st[0]) / 19.0);
    obs[2] = (float)st[2];
    obs[3] = st[1] == 0 ? 1.0f : 0.0f;
}
static void wj_reset(orc_mt *rng, double *st) { /* walljump.py:40-45 */
    st[0] = 0;
    st[1] = 0;
    st[2] = orc_mt_double(rng) < 0.7 ? 1 : 0; /* int(np.random.rand() < 0.7) */
    st[3] = 0;
}
static void wj_step(double *st, int a, float *obs, double *reward, int *done) { /* walljump.py:56-98 */
    static const int DX[4] = {0, 1, -1, 1};
    int x = (int)st[0], in_air = (int)st[1], wall = (int)st[2];
    double r = -0.01;
    int d = 0, just_jumped = 0;
    if (a == 3 && in_air == 0) {
        in_air = 3;
        just_jumped = 1;
    }
    int px = clipi(x + DX[a], 0, 19);
    int crossing = (x < 10 && 10 <= px) || (px < 10 && 10 <= x);
    if (crossing && wall == 1 && in_air == 0) {
        px = x;
        r = r - 0.02;
    }
    if (just_jumped && !crossing && abs(10 - x) > 1) r = r - 0.03;
    x = px;
    if (in_air > 0) in_air -= 1;
    if (x == 19) {
        r = 1.0;
        d = 1;
    }
    st[0] = x;
    st[1] = in_air;
    st[3] += 1;
    if (st[3] >= 150) d = 1;
    *reward = r;
    *done = d;
    wj_obs(st, obs);
}

/* =========================================================================================
 * Crawler-shape (BUILD-DEFINED, parity unpinned): 172-dim obs / 20-dim Box(-1,1) action chain
 * of damped, spring-coupled joints driving a planar root.  Stands in for the reference's
 * gym.make("Ant-v5") (backend/mlagents/envs.py:274-277), whose MuJoCo physics is not importable.
 * state = [q[20], qd[20], prev_a[20], root: z,vx,vy,pitch,roll,pitch_rate,roll_rate,x, steps]
 * All arithmetic float32, no contraction; sin/cos evaluated via a fixed odd/even polynomial so CPU
 * and GPU agree bit-for-bit.
 * ======================================================================================= */
static float cr_sin(float x) { /* |x| <= 1.3: degree-9 Taylor, Horner, float32 */
    float x2 = x * x;
    float p = 2.7557319e-06f;
    p = p * x2 + -1.9841270e-04f;
    p = p * x2 + 8.3333333e-03f;
    p = p * x2 + -1.6666667e-01f;
    p = p * x2;
    p = p * x;
    return x + p;
}
static float cr_cos(float x) { /* degree-10 Taylor */
    float x2 = x * x;
    float p = -2.7557319e-07f;
    p = p * x2 + 2.4801587e-05f;
    p = p * x2 + -1.3888889e-03f;
    p = p * x2 + 4.1666668e-02f;
    p = p * x2 + -0.5f;
    p = p * x2;
    return 1.0f + p;
}
static float cr_clip(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void chain_obs(const double *st, float *obs, const int nj) {
    const double *q = st, *qd = st + nj, *pa = st + 2 * nj, *root = st + 3 * nj;
    /* 12 root features */
    float pitch = (float)root[3], roll = (float)root[4];
    obs[0] = (float)root[0];
    obs[1] = (float)root[1];
    obs[2] = (float)root[2];
    obs[3] = pitch;
    obs[4] = roll;
    obs[5] = (float)root[5];
    obs[6] = (float)root[6];
    obs[7] = cr_sin(pitch);
    obs[8] = cr_cos(pitch);
    obs[9] = cr_sin(roll);
    obs[10] = cr_cos(roll);
    obs[11] = (float)root[0] - 0.55f;
    /* 8 features per joint */
    for (int j = 0; j < nj; j++) {
        float qj = (float)q[j], qdj = (float)qd[j];
        float qn = (float)q[(j + 1) % nj];
        float s = cr_sin(qj), c = cr_cos(qj);
        float side = (j & 1) ? -1.0f : 1.0f;
        float contact = -(s + side * pitch * 0.5f);
        float *o = obs + 12 + 8 * j;
        o[0] = qj;
        o[1] = qdj * 0.1f;
        o[2] = s;
        o[3] = c;
        o[4] = (float)pa[j];
        o[5] = qn - qj;
        o[6] = contact > 0.0f ? contact : 0.0f;
        o[7] = qj * qj;
    }
}
static void chain_reset_hash(uint32_t seed, double *st, const int nj) {
    /* counter-based init (no MT19937): small random joint angles / velocities, upright root */
    for (int j = 0; j < nj; j++) {
        uint32_t h0 = orc_mix32(seed, (uint32_t)j, 0x51u), h1 = orc_mix32(seed, (uint32_t)j, 0x52u);
        float u0 = (float)(h0 >> 8) * (1.0f / 16777216.0f), u1 = (float)(h1 >> 8) * (1.0f / 16777216.0f);
        st[j] = (double)((u0 - 0.5f) * 0.2f);
        st[nj + j] = (double)((u1 - 0.5f) * 0.2f);
        st[2 * nj + j] = 0.0;
    }
    double *root = st + 3 * nj;
    root[0] = (double)0.55f;
    for (int k = 1; k < 8; k++) root[k] = 0.0;
    st[3 * nj + 8] = 0;
}
static void ant_obs(const double *st, float *obs);
static void chain_step(double *st, const float *act, float *obs, double *reward, int *done, const int nj) {
    const float dt = 0.05f, gear = 8.0f, kq = 4.0f, cq = 1.5f, kc = 1.0f;
    float q[20], qd[20], a[20];
    double *root = st + 3 * nj;
    for (int j = 0; j < nj; j++) {
        q[j] = (float)st[j];
        qd[j] = (float)st[nj + j];
        a[j] = cr_clip(act[j], -1.0f, 1.0f);
    }
    float thrust_x = 0.0f, thrust_y = 0.0f, asym = 0.0f, ctrl = 0.0f, lift = 0.0f;
    float nq[20], nqd[20];
    for (int j = 0; j < nj; j++) {
        float ql = q[(j + nj - 1) % nj], qr = q[(j + 1) % nj];
        float lap = (ql + qr) - 2.0f * q[j];
        float acc = gear * a[j];
        acc = acc - kq * q[j];
        acc = acc - cq * qd[j];
        acc = acc + kc * lap;
        float v = qd[j] + dt * acc;
        float p = q[j] + dt * v;
        if (p > 1.2f) {
            p = 1.2f;
            v = 0.0f;
        }
        if (p < -1.2f) {
            p = -1.2f;
            v = 0.0f;
        }
        nq[j] = p;
        nqd[j] = v;
        float c = cr_cos(p), s = cr_sin(p);
        float side = (j & 1) ? -1.0f : 1.0f;
        float w = (j < nj / 2) ? 1.0f : -1.0f;
        thrust_x = thrust_x + (side * v) * c;
        thrust_y = thrust_y + (w * v) * c;
        asym = asym + side * s;
        lift = lift + c;
        ctrl = ctrl + a[j] * a[j];
    }
    float z = (float)root[0], vx = (float)root[1], vy = (float)root[2], pitch = (float)root[3], roll = (float)root[4];
    float pr = (float)root[5], rr = (float)root[6], x = (float)root[7];
    vx = vx + dt * (0.15f * thrust_x - 0.8f * vx);
    vy = vy + dt * (0.15f * thrust_y - 0.8f * vy);
    pr = pr + dt * (0.3f * asym - 6.0f * pitch - 1.2f * pr);
    rr = rr + dt * (0.05f * thrust_y - 6.0f * roll - 1.2f * rr);
    pitch = pitch + dt * pr;
    roll = roll + dt * rr;
    z = 0.25f + (nj == 20 ? 0.015f : 0.3f / (float)nj) * lift; /* mean cos(q) in [0.36,1] -> z in about [0.36,0.55] */
    x = x + dt * vx;
    for (int j = 0; j < nj; j++) {
        st[j] = (double)nq[j];
        st[nj + j] = (double)nqd[j];
        st[2 * nj + j] = (double)a[j];
    }
    root[0] = z;
    root[1] = vx;
    root[2] = vy;
    root[3] = pitch;
    root[4] = roll;
    root[5] = pr;
    root[6] = rr;
    root[7] = x;
    st[3 * nj + 8] += 1;
    int unhealthy = (z < 0.38f) || (fabsf(pitch) > 1.0f) || (fabsf(roll) > 1.0f);
    float rew = 1.0f + vx;
    rew = rew - 0.5f * ctrl * 0.05f;
    *reward = (double)rew;
    *done = unhealthy || st[3 * nj + 8] >= 1000;
    if (nj == ANT_NJ) ant_obs(st, obs);
    else chain_obs(st, obs, nj);
}

/* Ant-v5 observation order over the 8-joint chain state (csrc/tma_tasks.h ChainTask<8, 1>::obs) */
static void ant_obs(const double *st, float *obs) {
    const double *q = st, *qd = st + ANT_NJ, *root = st + 3 * ANT_NJ;
    float pitch = (float)root[3], roll = (float)root[4];
    float hp = 0.5f * pitch, hr = 0.5f * roll;
    float sp = cr_sin(hp), cp = cr_cos(hp), sr = cr_sin(hr), cr = cr_cos(hr);
    obs[0] = (float)root[0];
    obs[1] = cp * cr;
    obs[2] = sr * cp;
    obs[3] = sp * cr;
    obs[4] = -(sp * sr);
    for (int j = 0; j < ANT_NJ; j++) obs[5 + j] = (float)q[j];
    obs[13] = (float)root[1];
    obs[14] = (float)root[2];
    obs[15] = 0.0f;
    obs[16] = (float)root[6];
    obs[17] = (float)root[5];
    obs[18] = 0.0f;
    for (int j = 0; j < ANT_NJ; j++) obs[19 + j] = (float)qd[j];
    for (int b = 0; b < 13; b++) {
        float cz = 0.0f;
        if (b >= 1 && b <= ANT_NJ) {
            int j = b - 1;
            float side = (j & 1) ? -1.0f : 1.0f;
            float contact = -(cr_sin((float)q[j]) + side * pitch * 0.5f);
            cz = contact > 0.0f ? (contact < 1.0f ? contact : 1.0f) : 0.0f;
        }
        float *p = obs + 27 + 6 * b;
        p[0] = 0.0f, p[1] = 0.0f, p[2] = cz, p[3] = 0.0f, p[4] = 0.0f, p[5] = 0.0f;
    }
}
static void crawler_obs(const double *st, float *obs) { chain_obs(st, obs, 20); }
static void crawler_reset_hash(uint32_t seed, double *st) { chain_reset_hash(seed, st, 20); }
static void crawler_step(double *st, const float *act, float *obs, double *reward, int *done) { chain_step(st, act, obs, reward, done, 20); }
static void ant_reset_hash(uint32_t seed, double *st) { chain_reset_hash(seed, st, ANT_NJ); }
static void ant_step(double *st, const float *act, float *obs, double *reward, int *done) { chain_step(st, act, obs, reward, done, ANT_NJ); }

/* =========================================================================================
 * numpy / OpenBLAS summation orders the three float tasks below depend on (probed against numpy 2.2.6 + its bundled
 * OpenBLAS in the build container; the fixtures pin them): np.dot / np.linalg.norm / `@` of 2- and 3-element float64
 * operands are one FMA chain from the left, s = a0 b0; s = fma(a1, b1, s); s = fma(a2, b2, s); the 3x3 matrix-vector
 * product starts from the middle column, s = a1 b1; s = fma(a0, b0, s); s = fma(a2, b2, s).
 * ======================================================================================= */
static double dot2(double a0, double b0, double a1, double b1) { return fma(a1, b1, a0 * b0); }
static double dot3(const double *a, const double *b) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }
static double matvec3_row(const double *a, const double *v) { return fma(a[2], v[2], fma(a[0], v[0], a[1] * v[1])); }
static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* `x ** y` on a numpy float64 scalar is libm pow(x, y) (scalarmath): within 1 ulp of, not always equal to, x * x or sqrt(x).  Called through
 * a volatile pointer so that gcc does not rewrite pow(x, 2.0) as x * x. */
static double (*volatile libm_pow)(double, double) = pow;

/* =========================================================================================
 * Bicycle -- backend/examples/bicycle.py:14-37 (constants), :40-58 (reset), :60-125 (step), :127-141 (obs); adapter
 * backend/mlagents/envs.py:228-239 (Discrete(3), 2000-step limit).  SURVEY.md 8f rank N3.
 * state = [x, z, theta, phi, phi_dot, delta, goal_x, goal_z, dist_to_goal, steps]
 * ======================================================================================= */
#define BK_MAX_PHI 0.78539816339744828   /* np.pi / 4 */
#define BK_MAX_DELTA 0.52359877559829882 /* np.pi / 6 */
static void bike_obs(const double *st, float *obs) { /* bicycle.py:127-141 */
    double dx = st[6] - st[0], dz = st[7] - st[1];
    double dist = sqrt(dot2(dx, dx, dz, dz));
    double nx = 0.0, nz = 0.0;
    if (dist > 0) nx = dx / dist, nz = dz / dist;
    obs[0] = (float)st[3];
    obs[1] = (float)st[4];
    obs[2] = (float)st[5];
    obs[3] = (float)cos(st[2]);
    obs[4] = (float)sin(st[2]);
    obs[5] = (float)nx;
    obs[6] = (float)nz;
}
static void bike_reset(orc_mt *rng, double *st) { /* bicycle.py:40-58 */
    st[0] = st[1] = st[2] = 0.0;
    st[3] = mt_uniform(rng, -0.1, 0.1);
    st[4] = mt_uniform(rng, -0.1, 0.1);
    st[5] = 0.0;
    double radius = mt_uniform(rng, 15, 25);
    double angle = mt_uniform(rng, -BK_MAX_PHI, BK_MAX_PHI);
    st[6] = radius * cos(angle);
    st[7] = radius * sin(angle);
    st[8] = sqrt(dot2(st[6], st[6], st[7], st[7])); /* goal - [0, 0] */
    st[9] = 0;
}
static void bike_step(double *st, int a, float *obs, double *reward, int *done) { /* bicycle.py:60-125 */
    const double g = 9.8, h = 0.8, L = 1.0, v = 5.0, dt = 0.02;
    st[9] += 1;
    double delta = st[5] + (a == 0 ? -0.05 : (a == 2 ? 0.05 : 0.0));
    delta = clipd(delta, -BK_MAX_DELTA, BK_MAX_DELTA);
    double grav = (g / h) * sin(st[3]);
    double cen = (v * v / (L * h)) * tan(delta);
    cen = cen * cos(st[3]);
    double phi_ddot = grav - cen;
    st[4] = st[4] + phi_ddot * dt;
    st[3] = st[3] + st[4] * dt;
    delta = delta * 0.95;
    st[5] = delta;
    double th = (v / L) * tan(delta);
    st[2] = st[2] + th * dt;
    double cx = v * cos(st[2]);
    st[0] = st[0] + cx * dt;
    double cz = v * sin(st[2]);
    st[1] = st[1] + cz * dt;
    double dx = st[6] - st[0], dz = st[7] - st[1];
    double nd = sqrt(dot2(dx, dx, dz, dz));
    double progress = (st[8] - nd) * 10.0;
    st[8] = nd;
    double upright = (1.0 - libm_pow(fabs(st[3]) / BK_MAX_PHI, 0.5)) * 0.2;
    double den = nd > 0 ? nd : 1.0;
    double heading = dot2(cos(st[2]), dx / den, sin(st[2]), dz / den) * 0.3;
    double steer = -(fabs(delta) / BK_MAX_DELTA) * 0.1;
    double r = progress + upright;
    r = r + heading;
    r = r + steer;
    int d = 0;
    if (fabs(st[3]) > BK_MAX_PHI) r = -10.0, d = 1;
    if (st[9] > 2000) d = 1;
    if (nd < 2.0) r = 50.0, d = 1;
    *reward = r;
    *done = d;
    bike_obs(st, obs);
}

/* =========================================================================================
 * BrickBreak -- backend/examples/brick_break.py:14-37 (constants), :39-47 (reset), :49-121 (step), :123-131 (obs); adapter
 * backend/mlagents/envs.py:214-225 (Discrete(3), 2000-step limit).
 * state = [paddle_x, ball_x, ball_y, vel_x, vel_y, steps, bricks[5][8]]
 * ======================================================================================= */
#define BB_OBS 45
static void brick_obs(const double *st, float *obs) { /* brick_break.py:123-131 */
    obs[0] = (float)(st[1] / 40.0);
    obs[1] = (float)(st[2] / 40.0);
    obs[2] = (float)st[3];
    obs[3] = (float)st[4];
    obs[4] = (float)(st[0] / 40.0);
    for (int k = 0; k < 40; k++) obs[5 + k] = (float)st[6 + k];
}
static void brick_reset(orc_mt *rng, double *st) { /* brick_break.py:39-47 */
    st[0] = 20.0;
    st[1] = 20.0;
    st[2] = 10.0;
    double angle = mt_uniform(rng, 0.78539816339744828, 2.3561944901923448); /* np.pi / 4, 3 * np.pi / 4 */
    st[3] = cos(angle) * 1.5;
    st[4] = sin(angle) * 1.5;
    st[5] = 0;
    for (int k = 0; k < 40; k++) st[6 + k] = 1.0;
}
static void brick_step(double *st, int a, float *obs, double *reward, int *done) { /* brick_break.py:49-121 */
    st[5] += 1;
    if (a == 0) st[0] -= 3;
    else if (a == 2) st[0] += 3;
    st[0] = clipd(st[0], 4.0, 36.0);
    st[1] += st[3];
    st[2] += st[4];
    double r = 0.0;
    if (st[1] <= 1 || st[1] >= 39) st[3] *= -1;
    if (st[2] >= 39) st[4] *= -1;
    if (st[4] < 0 && st[2] - 1 <= 2 && st[1] >= st[0] - 4.0 && st[1] <= st[0] + 4.0) {
        st[4] *= -1;
        double offset = (st[1] - st[0]) / 4.0;
        st[3] += offset * 0.5;
        r = 0.1;
    }
    const double y0 = 40 - 5 * 2 - 10; /* brick_y_start */
    int hit = 0;
    for (int row = 0; row < 5 && !hit; row++)
        for (int c = 0; c < 8; c++)
            if (st[6 + row * 8 + c] == 1) {
                double bx = c * 5.0, by = y0 + row * 2;
                if (st[1] >= bx && st[1] <= bx + 5.0 && st[2] >= by && st[2] <= by + 2) {
                    st[6 + row * 8 + c] = 0;
                    st[4] *= -1;
                    r = 1.0;
                    hit = 1;
                    break;
                }
            }
    int d = 0;
    if (st[2] < 1) r = -1.0, d = 1;
    double left = 0;
    for (int k = 0; k < 40; k++) left += st[6 + k];
    if (left == 0) r = 10.0, d = 1;
    if (st[5] > 2000) d = 1;
    *reward = r;
    *done = d;
    brick_obs(st, obs);
}

/* =========================================================================================
 * Glider -- backend/examples/glider.py:14-53 (constants), :55-79 (wind), :81-88 (reset), :90-237 (step), :239-265 (obs);
 * adapter backend/mlagents/envs.py:242-253 (Discrete(5), 4000-step limit).
 * state = [pos(3), vel(3), rot(3), ang_vel(3), waypoint index, steps]
 * ======================================================================================= */
#define GL_OBS 16
static const double GL_WP[2][3] = {{-160.0, 0.0, 70.0}, {160.0, 0.0, 70.0}};
static void glider_obs(const double *st, float *obs) { /* glider.py:239-265 */
    const double *wp = GL_WP[(int)st[12]];
    double vec[3] = {wp[0] - st[0], wp[1] - st[1], wp[2] - st[2]};
    double dist = sqrt(dot3(vec, vec));
    obs[0] = (float)(st[5] / 10.0);
    obs[1] = (float)((st[2] - 50.0) / 50.0);
    obs[2] = (float)st[6];
    obs[3] = (float)st[7];
    obs[4] = (float)sin(st[8]);
    obs[5] = (float)cos(st[8]);
    for (int k = 0; k < 3; k++) obs[6 + k] = (float)st[9 + k];
    for (int k = 0; k < 3; k++) obs[9 + k] = (float)(st[3 + k] / 20.0);
    for (int k = 0; k < 3; k++) obs[12 + k] = (float)(vec[k] / (dist + 1e-8));
    obs[15] = (float)(dist / 100.0);
}
static void glider_reset(orc_mt *rng, double *st) { /* glider.py:81-88 */
    st[0] = 0.0, st[1] = 0.0, st[2] = 60.0;
    st[3] = 15.0, st[4] = 0.0, st[5] = -1.0;
    st[6] = st[7] = st[8] = 0.0;
    for (int k = 0; k < 3; k++) st[9 + k] = mt_uniform(rng, -0.1, 0.1);
    st[12] = (double)orc_mt_interval(rng, 1); /* np.random.randint(0, 2): one masked 32-bit draw */
    st[13] = 0;
}
static void matmul3(const double A[3][3], const double B[3][3], double C[3][3]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i][j] = fma(A[i][2], B[2][j], fma(A[i][1], B[1][j], A[i][0] * B[0][j]));
}
static void glider_step(double *st, int a, float *obs, double *reward, int *done) { /* glider.py:90-237 */
    const double dt = 0.02, two_pi = 6.283185307179586, pi = 3.141592653589793;
    double *pos = st, *vel = st + 3, *rot = st + 6, *av = st + 9;
    st[13] += 1;
    double tq[3] = {0.0, 0.0, 0.0}; /* roll, pitch, yaw */
    if (a == 1) tq[0] = -15.0, tq[2] = 4.0;
    else if (a == 2) tq[0] = 15.0, tq[2] = -4.0;
    else if (a == 3) tq[1] = 10.0;
    else if (a == 4) tq[1] = -10.0;
    for (int k = 0; k < 3; k++) av[k] = av[k] + tq[k] * dt;
    for (int k = 0; k < 3; k++) av[k] = av[k] * 0.95;
    for (int k = 0; k < 3; k++) rot[k] = rot[k] + av[k] * dt;
    rot[0] = clipd(rot[0], -pi / 2, pi / 2);
    rot[1] = clipd(rot[1], -pi / 4, pi / 4);
    /* wind (glider.py:55-79) */
    const double f1 = 1.0 / 250.0, f2 = 1.0 / 400.0;
    double u1 = sin(pos[0] * f1 * 2 * pi) * cos(pos[1] * f1 * 2 * pi) * 8.0 * 1.0;
    double u2 = sin(pos[0] * f2 * 2 * pi / 1.5) * cos(pos[1] * f1 * 2 * pi / 1.5) * 8.0 * 0.7;
    double wind[3] = {1.0, 0.5, u1 + u2};
    double va[3] = {vel[0] - wind[0], vel[1] - wind[1], vel[2] - wind[2]};
    double vam = sqrt(dot3(va, va));
    double aoa = va[0] != 0 ? atan2(-va[2], va[0]) : 0.0;
    double aero[3] = {0.0, 0.0, 0.0};
    if (vam > 0.1) {
        double CL = two_pi * aoa;
        double CD = 0.02 + 0.05 * libm_pow(CL, 2.0);
        double q = 0.5 * 1.225 * libm_pow(vam, 2.0) * 0.5;
        double lift = q * CL, drag = q * CD;
        double F[3] = {0 + -drag, 0.0, lift + 0};
        double cr = cos(rot[0]), sr = sin(rot[0]), cp = cos(rot[1]), sp = sin(rot[1]), cy = cos(rot[2]), sy = sin(rot[2]);
        const double Rr[3][3] = {{1, 0, 0}, {0, cr, -sr}, {0, sr, cr}};
        const double Rp[3][3] = {{cp, 0, sp}, {0, 1, 0}, {-sp, 0, cp}};
        const double Ry[3][3] = {{cy, -sy, 0}, {sy, cy, 0}, {0, 0, 1}};
        double T[3][3], R[3][3];
        matmul3(Ry, Rp, T);
        matmul3(T, Rr, R);
        for (int i = 0; i < 3; i++) aero[i] = matvec3_row(R[i], F);
    } else {
        aoa = 0;
    }
    double total[3] = {aero[0] + 0, aero[1] + 0, aero[2] + -(1.5 * 9.81)};
    for (int k = 0; k < 3; k++) vel[k] = vel[k] + (total[k] / 1.5) * dt;
    for (int k = 0; k < 3; k++) pos[k] = pos[k] + vel[k] * dt;
    int d = 0, wi = (int)st[12];
    double vec[3] = {GL_WP[wi][0] - pos[0], GL_WP[wi][1] - pos[1], GL_WP[wi][2] - pos[2]};
    double dist = sqrt(dot3(vec, vec));
    if (dist < 15.0) st[12] = (double)((wi + 1) % 2);
    double vn = sqrt(dot3(vel, vel));
    double vd[3], td[3];
    for (int k = 0; k < 3; k++) vd[k] = vel[k] / (vn + 1e-8), td[k] = vec[k] / (dist + 1e-8);
    double H = (dot3(vd, td) + 1) / 2;
    double E = clipd(vn / 30.0, 0, 2.0);
    double r = E * (H - E + 1);
    double lateral = fabs(pos[1]);
    if (lateral > 250.0) {
        double pr = (lateral - 250.0) / 100.0;
        r -= 2.0 * libm_pow(pr, 2.0);
    }
    if (pos[2] > 250.0) {
        double pr = (pos[2] - 250.0) / 50.0;
        r -= 2.0 * libm_pow(pr, 2.0);
    } else if (pos[2] < 25.0) {
        r -= 0.5;
    }
    if (pos[2] < 5.0) r = -50.0, d = 1;
    if (fabs(aoa) > 0.26179938779914941) r = -50.0, d = 1; /* np.deg2rad(15) */
    if (dist > 500) r = -50.0, d = 1;
    if (st[13] > 4000) d = 1;
    *reward = r;
    *done = d;
    glider_obs(st, obs);
}

/* =========================================================================================
 * adapter + vec env
 * ======================================================================================= */
static void task_obs(int task, const double *st, float *obs) {
    switch (task) {
    case ORC_BASIC: basic_obs(st, obs); break;
    case ORC_GRIDWORLD: grid_obs(st, obs); break;
    case ORC_BALL3D: ball_obs(st, obs); break;
    case ORC_PUSH: push_obs(st, obs); break;
    case ORC_CRAWLER: crawler_obs(st, obs); break;
    case ORC_ANT: ant_obs(st, obs); break;
    case ORC_WALLJUMP: wj_obs(st, obs); break;
    case ORC_BICYCLE: bike_obs(st, obs); break;
    case ORC_BRICKBREAK: brick_obs(st, obs); break;
    case ORC_GLIDER: glider_obs(st, obs); break;
    }
}

/* LegacySingleAgentGymAdapter.reset(seed=s): np.random.seed(s); env_ctor() [its __init__ resets];
 * env.reset() -- the SECOND reset is the visible one (backend/mlagents/envs.py:110-123). */
void orc_reset_from_seed(int task, uint32_t seed, double *st, float *obs) {
    orc_mt rng;
    switch (task) {
    case ORC_BASIC: basic_reset(st); break; /* envs.py:48-58: no RNG use */
    case ORC_GRIDWORLD:
        orc_mt_seed(&rng, seed);
        grid_reset(&rng, st);
        grid_reset(&rng, st);
        break;
    case ORC_BALL3D:
        orc_mt_seed(&rng, seed);
        ball_reset(&rng, st);
        ball_reset(&rng, st);
        break;
    case ORC_PUSH:
        orc_mt_seed(&rng, seed);
        push_reset(&rng, st);
        push_reset(&rng, st);
        break;
    case ORC_CRAWLER: crawler_reset_hash(seed, st); break;
    case ORC_ANT: ant_reset_hash(seed, st); break;
    case ORC_WALLJUMP:
        orc_mt_seed(&rng, seed);
        wj_reset(&rng, st);
        wj_reset(&rng, st);
        break;
    case ORC_BICYCLE:
        orc_mt_seed(&rng, seed);
        bike_reset(&rng, st);
        bike_reset(&rng, st);
        break;
    case ORC_BRICKBREAK:
        orc_mt_seed(&rng, seed);
        brick_reset(&rng, st);
        brick_reset(&rng, st);
        break;
    case ORC_GLIDER:
        orc_mt_seed(&rng, seed);
        glider_reset(&rng, st);
        glider_reset(&rng, st);
        break;
    }
    if (obs) task_obs(task, st, obs);
}

void orc_legacy_step(int task, double *st, const void *action, float *obs, double *reward, int *done) {
    float tmp[CRAWLER_OBS];
    if (!obs) obs = tmp;
    switch (task) {
    case ORC_BASIC: basic_step(st, *(const int32_t *)action, obs, reward, done); break;
    case ORC_GRIDWORLD: grid_step(st, *(const int32_t *)action, obs, reward, done); break;
    case ORC_BALL3D: ball_step(st, *(const int32_t *)action, obs, reward, done); break;
    case ORC_PUSH: push_step(st, *(const int32_t *)action, obs, reward, done); break;
    case ORC_CRAWLER: crawler_step(st, (const float *)action, obs, reward, done); break;
    case ORC_ANT: ant_step(st, (const float *)action, obs, reward, done); break;
    case ORC_WALLJUMP: wj_step(st, *(const int32_t *)action, obs, reward, done); break;
    case ORC_BICYCLE: bike_step(st, *(const int32_t *)action, obs, reward, done); break;
    case ORC_BRICKBREAK: brick_step(st, *(const int32_t *)action, obs, reward, done); break;
    case ORC_GLIDER: glider_step(st, *(const int32_t *)action, obs, reward, done); break;
    }
}

struct orc_vec {
    int task, n, D, S, A, n_threads;
    uint32_t seed_base, env_offset;
    double *state;     /* [n][S] */
    uint32_t *episode; /* [n] index of the running episode */
    double *ep_ret;    /* Monitor: running sum of float(reward) */
    int32_t *ep_len;
};

orc_vec *orc_vec_create(int task, int n_envs, uint32_t seed_base, uint32_t env_offset) {
    orc_vec *v = (orc_vec *)calloc(1, sizeof(orc_vec));
    v->task = task;
    v->n = n_envs;
    v->D = orc_obs_dim(task);
    v->S = orc_state_dim(task);
    v->A = orc_act_dim(task);
    v->n_threads = 1;
    v->seed_base = seed_base;
    v->env_offset = env_offset;
    v->state = (double *)calloc((size_t)n_envs * v->S, sizeof(double));
    v->episode = (uint32_t *)calloc(n_envs, sizeof(uint32_t));
    v->ep_ret = (double *)calloc(n_envs, sizeof(double));
    v->ep_len = (int32_t *)calloc(n_envs, sizeof(int32_t));
    return v;
}
void orc_vec_destroy(orc_vec *v) {
    if (!v) return;
    free(v->state);
    free(v->episode);
    free(v->ep_ret);
    free(v->ep_len);
    free(v);
}
void orc_vec_set_threads(orc_vec *v, int n) { v->n_threads = n < 1 ? 1 : n; }

/* DummyVecEnv.reset with seeds seed+i (backend/mlagents/training.py:80,84) */
void orc_vec_reset(orc_vec *v, float *obs_out) {
#pragma omp parallel for num_threads(v->n_threads) schedule(static)
    for (int i = 0; i < v->n; i++) {
        v->episode[i] = 0;
        v->ep_ret[i] = 0.0;
        v->ep_len[i] = 0;
        orc_reset_from_seed(v->task, orc_episode_seed(v->seed_base, v->env_offset + (uint32_t)i, 0),
                            v->state + (size_t)i * v->S, obs_out ? obs_out + (size_t)i * v->D : NULL);
    }
}

void orc_vec_step(orc_vec *v, const void *actions, float *obs_out, float *rew32_out, double *rew64_out,
                  uint8_t *term_out, uint8_t *trunc_out, float *term_obs_out, double *ep_ret_out,
                  int32_t *ep_len_out) {
    const int task = v->task, D = v->D, S = v->S;
    const int max_steps = orc_max_episode_steps(task);
#pragma omp parallel for num_threads(v->n_threads) schedule(static)
    for (int i = 0; i < v->n; i++) {
        double *st = v->state + (size_t)i * S;
        float obs[CRAWLER_OBS];
        double r;
        int done, terminated, truncated;
        const void *act = (task == ORC_CRAWLER || task == ORC_ANT) ? (const void *)((const float *)actions + (size_t)i * v->A)
                                              : (const void *)((const int32_t *)actions + i);
        orc_legacy_step(task, st, act, obs, &r, &done);
        int steps = (int)st[orc_steps_index(task)];
        if (task == ORC_BASIC) { /* envs.py:76 */
            terminated = done;
            truncated = (steps >= max_steps) && !terminated;
        } else { /* adapter rule, envs.py:139-145 */
            int hit = steps >= max_steps;
            terminated = done && !hit;
            truncated = hit;
        }
        v->ep_ret[i] += r; /* Monitor.step: rewards.append(float(reward)); sum() in order */
        v->ep_len[i] += 1;
        if (rew32_out) rew32_out[i] = (float)r; /* DummyVecEnv.buf_rews is float32 */
        if (rew64_out) rew64_out[i] = r;
        if (term_out) term_out[i] = (uint8_t)terminated;
        if (trunc_out) trunc_out[i] = (uint8_t)truncated;
        if (terminated || truncated) {
            if (term_obs_out) memcpy(term_obs_out + (size_t)i * D, obs, sizeof(float) * D);
            if (ep_ret_out) ep_ret_out[i] = v->ep_ret[i];
            if (ep_len_out) ep_len_out[i] = v->ep_len[i];
            v->ep_ret[i] = 0.0;
            v->ep_len[i] = 0;
            v->episode[i] += 1;
            orc_reset_from_seed(task, orc_episode_seed(v->seed_base, v->env_offset + (uint32_t)i, v->episode[i]), st, obs);
        } else {
            if (ep_ret_out) ep_ret_out[i] = 0.0;
            if (ep_len_out) ep_len_out[i] = 0;
        }
        if (obs_out) memcpy(obs_out + (size_t)i * D, obs, sizeof(float) * D);
    }
}

void orc_vec_get_state(const orc_vec *v, double *out) { memcpy(out, v->state, sizeof(double) * (size_t)v->n * v->S); }
void orc_vec_set_state(orc_vec *v, const double *in) { memcpy(v->state, in, sizeof(double) * (size_t)v->n * v->S); }
void orc_vec_episode_index(const orc_vec *v, uint32_t *out) { memcpy(out, v->episode, sizeof(uint32_t) * v->n); }

/* =========================================================================================
 * GAE -- stable-baselines3 2.9.0 RolloutBuffer.compute_returns_and_advantage (3P).
 * All float32, numpy op order: delta = r + gamma*next_v*nnt - v ; gae = delta + (gamma*lam)*nnt*gae
 * `gl` = float32(gamma*gae_lambda) with the product taken in float64 by the caller (python floats).
 * ======================================================================================= */
static void gae_envs(const float *rewards, const float *values, const float *episode_starts, const float *last_values, const uint8_t *dones,
                     float gamma, float gl, int T, int N, int i0, int i1, float *adv, float *ret);
void orc_gae(const float *rewards, const float *values, const float *episode_starts, const float *last_values,
             const uint8_t *dones, float gamma, float gl, int T, int N, float *adv, float *ret) {
    gae_envs(rewards, values, episode_starts, last_values, dones, gamma, gl, T, N, 0, N, adv, ret);
}
/* the same per-env chains, env blocks dealt to `threads` OpenMP threads (bench.py's all-cores leg): identical arithmetic per element */
void orc_gae_threads(const float *rewards, const float *values, const float *episode_starts, const float *last_values, const uint8_t *dones,
                     float gamma, float gl, int T, int N, float *adv, float *ret, int threads) {
    const int blk = 16, nb = (N + blk - 1) / blk;
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int b = 0; b < nb; b++) gae_envs(rewards, values, episode_starts, last_values, dones, gamma, gl, T, N, b * blk, (b + 1) * blk < N ? (b + 1) * blk : N, adv, ret);
}
static void gae_envs(const float *rewards, const float *values, const float *episode_starts, const float *last_values, const uint8_t *dones,
                     float gamma, float gl, int T, int N, int i0, int i1, float *adv, float *ret) {
    for (int i = i0; i < i1; i++) {
        float last = 0.0f;
        for (int t = T - 1; t >= 0; t--) {
            float nnt, nv;
            if (t == T - 1) {
                nnt = 1.0f - (dones[i] ? 1.0f : 0.0f);
                nv = last_values[i];
            } else {
                nnt = 1.0f - episode_starts[(size_t)(t + 1) * N + i];
                nv = values[(size_t)(t + 1) * N + i];
            }
            float a = gamma * nv;
            a = a * nnt;
            float delta = rewards[(size_t)t * N + i] + a;
            delta = delta - values[(size_t)t * N + i];
            float b = gl * nnt;
            b = b * last;
            last = delta + b;
            adv[(size_t)t * N + i] = last;
            ret[(size_t)t * N + i] = last + values[(size_t)t * N + i];
        }
    }
}
