// pmcx_device_math.h -- gfx950 device-side scalar building blocks.
//
// Counter-based RNG (Philox4x32-10) and exp/log/cbrt/inverse-normal built from
// IEEE-754 +,-,*,/,sqrt and integer ops only, so results are reproducible
// bit for bit across launch geometries and against a host restatement.
// Built with -ffp-contract=off: no fused multiply-add may be formed here.
//
// Replaces on the reference path: pmc_srand/pmc_random/pmc_rand_int/
// rand_poisson (interface/wrf_pmc_init.F90:131,
// interface/wrf_pmc_trans_aero.F90:2107-2143), which draw from one sequential
// GSL stream per MPI rank; here every draw is addressed by
// (seed, global cell, step, site) so cells are independent.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pmcx {

// ------------------------------------------------------------------ Philox

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32x32->64 multiply (v_mad_u64_u32) per product: integer multiplies
        // are quarter-rate, so hi and lo must not be separate instructions
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

enum : uint32_t {
    SITE_POISSON = 1,
    SITE_CAND = 2,
    SITE_COAG = 3,
    SITE_HALVE = 4,
    SITE_GEN = 5,
    SITE_GENCLS = 6,
    SITE_EMIT_N = 7,
    SITE_EMIT_P = 8,
    SITE_DILUTE = 9,
    SITE_TRANS = 10,
    SITE_DEPOSIT = 11,
    SITE_SCALE = 12,
    SITE_RESAMPLE = 13
};

// counter = (cell, step, a, site<<28 | b); key = seed
__device__ __forceinline__ void rng_block(uint64_t seed, uint32_t cell, uint32_t step,
                                          uint32_t a, uint32_t site, uint32_t b,
                                          uint32_t (&w)[4])
{
    w[0] = cell; w[1] = step; w[2] = a;
    w[3] = (site << 28) | (b & 0x0FFFFFFFu);
    philox4x32_10(w, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// The uniform that decides a slot's Poisson count: word gB & 3 of the block addressed by
// (gA, gB & ~3), 32 bits (the resolution of the reference's GSL uniform).  A sweep along a row
// of partner bins draws four slots from one Philox block.
__device__ __forceinline__ double slot_uniform(const uint32_t (&w)[4], uint32_t gB)
{
    const uint32_t k = gB & 3u;
    const uint32_t x = k == 0 ? w[0] : (k == 1 ? w[1] : (k == 2 ? w[2] : w[3]));
    return (double)x * 0x1.0p-32;
}
__device__ __forceinline__ double poisson_uniform(uint64_t seed, uint32_t cell, uint32_t step,
                                                  uint32_t gA, uint32_t gB)
{
    uint32_t w[4];
    rng_block(seed, cell, step, (gA << 16) | (gB & ~3u), SITE_POISSON, 0, w);
    return slot_uniform(w, gB);
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo)
{
    const uint64_t x = ((uint64_t)hi << 32) | lo;
    return (double)(x >> 11) * 0x1.0p-53;
}

__device__ __forceinline__ uint32_t index_of(uint32_t w, uint32_t n)
{
    return __umulhi(w, n);
}

// -------------------------------------------------------------- exp / log

// (exp and log also run on the host: the logarithms of a mode's parameters are formed there, and
// with these the result does not depend on the host's libm -- same operations, same bits)
__host__ __device__ __forceinline__ double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
__host__ __device__ __forceinline__ uint64_t d2u(double x) { return __builtin_bit_cast(uint64_t, x); }

__host__ __device__ inline double det_exp(double x)
{
    if (x < -745.0) return 0.0;
    if (x > 709.0) return __builtin_inf();
    const double inv_ln2 = 1.4426950408889634;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double t = x * inv_ln2;
    const int k = (int)(t + (t < 0.0 ? -0.5 : 0.5));
    const double kd = (double)k;
    const double r = (x - kd * ln2_hi) - kd * ln2_lo;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    if (k >= -1021) return p * u2d((uint64_t)(k + 1023) << 52);
    return (p * u2d((uint64_t)(k + 1000 + 1023) << 52)) * 0x1.0p-1000;
}

__host__ __device__ inline double det_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const uint64_t u = d2u(x);
    int e = (int)((u >> 52) & 0x7FF) - 1023;
    double m = u2d((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double q = 2.0 / 23.0;
    q = q * s2 + 2.0 / 21.0;
    q = q * s2 + 2.0 / 19.0;
    q = q * s2 + 2.0 / 17.0;
    q = q * s2 + 2.0 / 15.0;
    q = q * s2 + 2.0 / 13.0;
    q = q * s2 + 2.0 / 11.0;
    q = q * s2 + 2.0 / 9.0;
    q = q * s2 + 2.0 / 7.0;
    q = q * s2 + 2.0 / 5.0;
    q = q * s2 + 2.0 / 3.0;
    q = q * s2 + 2.0;
    const double ed = (double)e;
    return ed * ln2_hi + (s * q + ed * ln2_lo);
}

// x^(-1/3): bit-level first guess, three division-free third-order steps
__device__ __forceinline__ double det_rcbrt(double x)
{
    const uint64_t u = 0x553EEEAD3BAB5000ull - d2u(x) / 3;
    double y = u2d(u);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double h = 1.0 - x * (y * y * y);
        y = y + y * h * (1.0 / 3.0 + h * (2.0 / 9.0));
    }
    return y;
}

__device__ __forceinline__ double det_cbrt(double x)
{
    if (x <= 0.0) return 0.0;
    const double y = det_rcbrt(x);
    return x * y * y;
}

__device__ inline double det_norminv(double p)
{
    const double a0 = -3.969683028665376e+01, a1 = 2.209460984245205e+02,
                 a2 = -2.759285104469687e+02, a3 = 1.383577518672690e+02,
                 a4 = -3.066479806614716e+01, a5 = 2.506628277459239e+00;
    const double b0 = -5.447609879822406e+01, b1 = 1.615858368580409e+02,
                 b2 = -1.556989798598866e+02, b3 = 6.680131188771972e+01,
                 b4 = -1.328068155288572e+01;
    const double c0 = -7.784894002430293e-03, c1 = -3.223964580411365e-01,
                 c2 = -2.400758277161838e+00, c3 = -2.549732539343734e+00,
                 c4 = 4.374664141464968e+00, c5 = 2.938163982698783e+00;
    const double d0 = 7.784695709041462e-03, d1 = 3.224671290700398e-01,
                 d2 = 2.445134137142996e+00, d3 = 3.754408661907416e+00;
    const double p_low = 0.02425;
    if (p < 0x1.0p-53) p = 0x1.0p-53;
    if (p < p_low) {
        const double q = __builtin_sqrt(-2.0 * det_log(p));
        return (((((c0 * q + c1) * q + c2) * q + c3) * q + c4) * q + c5) /
               ((((d0 * q + d1) * q + d2) * q + d3) * q + 1.0);
    }
    if (p <= 1.0 - p_low) {
        const double q = p - 0.5;
        const double r = q * q;
        return (((((a0 * r + a1) * r + a2) * r + a3) * r + a4) * r + a5) * q /
               (((((b0 * r + b1) * r + b2) * r + b3) * r + b4) * r + 1.0);
    }
    {
        const double q = __builtin_sqrt(-2.0 * det_log(1.0 - p));
        return -(((((c0 * q + c1) * q + c2) * q + c3) * q + c4) * q + c5) /
               ((((d0 * q + d1) * q + d2) * q + d3) * q + 1.0);
    }
}

// ---------------------------------------------------------------- Poisson

constexpr double POISSON_WALK_MAX = 200.0;
constexpr int32_t NSAMP_MAX = 0x00FFFFFF; // per-slot cap: a row's total fits 32 bits

// 1/n for n = 1..POISSON_RCP_N, correctly rounded (the walk multiplies by these instead of
// dividing: a division is ~35 instructions on gfx950, and the walk is as long as the largest
// count drawn by any lane of the wavefront).  Kernels that draw often keep a copy in LDS.
constexpr int POISSON_RCP_N = 64;
__device__ const double g_poisson_rcp[POISSON_RCP_N + 1] = {
    0.0, 1.0 / 1, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6, 1.0 / 7, 1.0 / 8, 1.0 / 9, 1.0 / 10,
    1.0 / 11, 1.0 / 12, 1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16, 1.0 / 17, 1.0 / 18, 1.0 / 19,
    1.0 / 20, 1.0 / 21, 1.0 / 22, 1.0 / 23, 1.0 / 24, 1.0 / 25, 1.0 / 26, 1.0 / 27, 1.0 / 28,
    1.0 / 29, 1.0 / 30, 1.0 / 31, 1.0 / 32, 1.0 / 33, 1.0 / 34, 1.0 / 35, 1.0 / 36, 1.0 / 37,
    1.0 / 38, 1.0 / 39, 1.0 / 40, 1.0 / 41, 1.0 / 42, 1.0 / 43, 1.0 / 44, 1.0 / 45, 1.0 / 46,
    1.0 / 47, 1.0 / 48, 1.0 / 49, 1.0 / 50, 1.0 / 51, 1.0 / 52, 1.0 / 53, 1.0 / 54, 1.0 / 55,
    1.0 / 56, 1.0 / 57, 1.0 / 58, 1.0 / 59, 1.0 / 60, 1.0 / 61, 1.0 / 62, 1.0 / 63, 1.0 / 64};

template <typename RcpPtr>
__device__ inline int32_t poisson(double mean, double u, RcpPtr rcp)
{
    if (!(mean > 0.0)) return 0;
    if (mean < POISSON_WALK_MAX) {
        double p = det_exp(-mean);
        double F = p;
        int32_t n = 0;
        while (u > F && n < 1000) {
            n = n + 1;
            if (n <= POISSON_RCP_N) p = p * (mean * rcp[n]);
            else p = p * (mean / (double)n);
            F = F + p;
        }
        return n;
    }
    const double z = det_norminv(u);
    const double x = mean + __builtin_sqrt(mean) * z + 0.5;
    if (x < 0.0) return 0;
    if (x > (double)NSAMP_MAX) return NSAMP_MAX;
    return (int32_t)x;
}
__device__ inline int32_t poisson(double mean, double u) { return poisson(mean, u, g_poisson_rcp); }

// ------------------------------------------------------ coagulation kernels
//
// Brownian kernel: Fuchs form, Jacobson (2005) eq. 15.33 -- the form of
// PartMC's kernel_brown [upstream] and of the in-tree
// WRFV3/chem/module_mosaic_coag.F:898-936 (slip constants 1.257/0.42/1.1
// there, Jacobson's 1.249/0.42/0.87 here).  Selected by
// COAG_KERNEL_TYPE_BROWN at interface/wrf_pmc_driver.F90:192.

constexpr double PI = 3.14159265358979323846;
constexpr double BOLTZMANN = 1.3806505e-23;
constexpr double AVOGADRO = 6.02214179e23;
constexpr double UNIV_GAS = 8.314472;
constexpr double AIR_MW = 2.89644e-2;
constexpr double GRAV = 9.80665;

enum : int {
    KERNEL_SEDI = 1,
    KERNEL_ADDITIVE = 2,
    KERNEL_CONSTANT = 3,
    KERNEL_BROWN = 4,
    KERNEL_ZERO = 5,
    KERNEL_BROWN_SEDI = 101
};

struct EnvDerived {
    double kT, viscosd, gasfreepath, rhoair;
    // per-cell constants that keep divisions out of the per-particle code
    double inv_gfp;     // 1 / gasfreepath
    double c_diff;      // kT / (6 pi mu)
    double c_speed;     // 8 kT / pi
    double c_inv_speed; // pi / (8 kT)
    double c_fall;      // 2 g / (9 mu)
};

struct PartProps {
    double rad, diffus, speedsq, deltasq, wfall;
};

struct KernelParams {
    double const_k, additive_beta;
};

__device__ inline EnvDerived env_derive(double temp, double pressure)
{
    EnvDerived e;
    e.kT = BOLTZMANN * temp;
    e.rhoair = AIR_MW * pressure / (UNIV_GAS * temp);
    const double tr = temp / 296.16;
    e.viscosd = (1.8325e-5 * (296.16 + 120.0) / (temp + 120.0)) * (tr * __builtin_sqrt(tr));
    const double viscosk = e.viscosd / e.rhoair;
    const double gasspeed =
        __builtin_sqrt(8.0 * BOLTZMANN * temp * AVOGADRO / (PI * AIR_MW));
    e.gasfreepath = 2.0 * viscosk / gasspeed;
    e.inv_gfp = 1.0 / e.gasfreepath;
    e.c_diff = e.kT / (6.0 * PI * e.viscosd);
    e.c_speed = 8.0 * e.kT / PI;
    e.c_inv_speed = PI / (8.0 * e.kT);
    e.c_fall = 2.0 * GRAV / (9.0 * e.viscosd);
    return e;
}

// Per-particle quantities of the Fuchs form.  Same formulas as Jacobson
// eq. 15.33, arranged so that one particle costs two divisions: 1/rad comes
// from the reciprocal cube root, 1/speed from a square root.
template <bool SEDI>
__device__ inline PartProps particle_props(double v, double m, const EnvDerived &e)
{
    PartProps q;
    const double x = v * (0.75 / PI);
    const double inv_rad = det_rcbrt(x);
    const double rad = x * inv_rad * inv_rad;
    const double knud = e.gasfreepath * inv_rad;
    const double cunning = 1.0 + knud * (1.249 + 0.42 * det_exp(-0.87 * (rad * e.inv_gfp)));
    const double diffus = e.c_diff * cunning * inv_rad;
    const double speedsq = e.c_speed / m;
    const double inv_speed = __builtin_sqrt(m * e.c_inv_speed);
    const double freepath = (8.0 / PI) * diffus * inv_speed;
    const double t1 = 2.0 * rad + freepath;
    const double tmp1 = t1 * t1 * t1;
    const double t2 = 4.0 * rad * rad + freepath * freepath;
    const double tmp2 = t2 * __builtin_sqrt(t2);
    const double d = (tmp1 - tmp2) / (6.0 * rad * freepath) - 2.0 * rad;
    q.rad = rad;
    q.diffus = diffus;
    q.speedsq = speedsq;
    q.deltasq = d * d;
    q.wfall = 0.0;
    if (SEDI) q.wfall = e.c_fall * (rad * rad) * (m / v - e.rhoair) * cunning;
    return q;
}

// K = 4 pi R D / (R/(R+delta) + 4D/(R c)) with the two inner quotients
// cleared: one division.
__device__ inline double brown_pair(const PartProps &a, const PartProps &b)
{
    const double R = a.rad + b.rad;
    const double D = a.diffus + b.diffus;
    const double Rd = R + __builtin_sqrt(a.deltasq + b.deltasq);
    const double Rc = R * __builtin_sqrt(a.speedsq + b.speedsq);
    return (4.0 * PI) * (R * D) * (Rd * Rc) / (R * Rc + 4.0 * D * Rd);
}

// ---- single-precision estimate of the Brownian kernel (a SCREEN, never a result) ----
//
// The accept test of a candidate pair is  u k_max < K f  with K in fp64 (the value both this
// library and the oracle define).  Most tests are far from the boundary: an estimate of K good to
// 2.5e-4 decides them, and only a test within 1e-3 of the boundary (or a particle outside the
// range where the estimate is that good) needs the fp64 value.  The decision taken is always the
// fp64 one; the estimate only tells when it cannot differ.  The fp64 evaluation costs about 600
// issue slots per pair (two IEEE square roots and two divisions per particle), this one about 120
// (v_log_f32 / v_exp_f32 / v_rcp_f32 / v_sqrt_f32).
//
// Accuracy: tests/test_gpu_parity.py::test_brownian_screen_is_within_its_margin sweeps the size
// range below at several temperatures and pressures (op 9 of pmcx_test_math): the largest
// relative deviation from the fp64 kernel found is asserted below 2.5e-4, a quarter of the margin.
// (measured: 1e-5 up to 1 um of radius, 1e-4 at 100 um, growing with radius / mean free path of
// the particle, where the difference of cubes that gives Fuchs' delta cancels in single
// precision).  Outside  FP_MIN < particle mean free path / radius < FP_MAX  (radius above about
// 100 um) a pair reports `ok = false` and takes the fp64 path.
struct EnvF32 {
    float gasfreepath, inv_gfp, c_diff, c_speed, c_inv_speed;
};
constexpr float BROWN_SCREEN_MARGIN = 1.0e-3f;
constexpr float BROWN_SCREEN_FP_MIN = 1.0f / 256.0f, BROWN_SCREEN_FP_MAX = 16384.0f;

struct PartPropsF32 {
    float rad, diffus, speedsq, deltasq;
    bool ok;
};

__device__ __forceinline__ PartPropsF32 particle_props_f32(float v, float m, const EnvF32 &e)
{
    PartPropsF32 q;
    const float x = v * (float)(0.75 / PI);
    const float inv_rad = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (-1.0f / 3.0f));
    const float rad = x * inv_rad * inv_rad;
    const float knud = e.gasfreepath * inv_rad;
    const float cunning =
        1.0f + knud * (1.249f + 0.42f * __builtin_amdgcn_exp2f(-0.87f * 1.44269504f * (rad * e.inv_gfp)));
    const float diffus = e.c_diff * cunning * inv_rad;
    const float speedsq = e.c_speed * __builtin_amdgcn_rcpf(m);
    const float inv_speed = __builtin_amdgcn_sqrtf(m * e.c_inv_speed);
    const float freepath = (float)(8.0 / PI) * diffus * inv_speed;
    const float t1 = 2.0f * rad + freepath;
    const float tmp1 = t1 * t1 * t1;
    const float t2 = 4.0f * rad * rad + freepath * freepath;
    const float tmp2 = t2 * __builtin_amdgcn_sqrtf(t2);
    const float d = (tmp1 - tmp2) * __builtin_amdgcn_rcpf(6.0f * rad * freepath) - 2.0f * rad;
    const float ratio = freepath * inv_rad;
    q.rad = rad; q.diffus = diffus; q.speedsq = speedsq; q.deltasq = d * d;
    q.ok = ratio > BROWN_SCREEN_FP_MIN && ratio < BROWN_SCREEN_FP_MAX;
    return q;
}

// estimate of brown_pair() for particles (v1, m1), (v2, m2); ok = false: do not trust it
__device__ __forceinline__ float brown_screen_f32(double v1, double m1, double v2, double m2,
                                                  const EnvF32 &e, bool &ok)
{
    const PartPropsF32 a = particle_props_f32((float)v1, (float)m1, e);
    const PartPropsF32 b = particle_props_f32((float)v2, (float)m2, e);
    const float R = a.rad + b.rad;
    const float D = a.diffus + b.diffus;
    const float Rd = R + __builtin_amdgcn_sqrtf(a.deltasq + b.deltasq);
    const float Rc = R * __builtin_amdgcn_sqrtf(a.speedsq + b.speedsq);
    ok = a.ok && b.ok;
    // K = 4 pi R D / (R / Rd + 4 D / Rc)
    return (float)(4.0 * PI) * (R * D) *
           __builtin_amdgcn_rcpf(R * __builtin_amdgcn_rcpf(Rd) + 4.0f * D * __builtin_amdgcn_rcpf(Rc));
}

// E pi (r1+r2)^2 |w1-w2|, Long (1974) efficiency, Stokes-Cunningham fall speed
__device__ inline double sedi_pair(const PartProps &a, const PartProps &b)
{
    const double r_big = a.rad > b.rad ? a.rad : b.rad;
    const double r_small = a.rad > b.rad ? b.rad : a.rad;
    double eff;
    if (r_big <= 50.0e-6) {
        const double rb_cm = r_big * 100.0, rs_cm = r_small * 100.0;
        eff = 4.5e4 * rb_cm * rb_cm * (1.0 - 3.0e-4 / rs_cm);
        if (eff < 1.0e-3) eff = 1.0e-3;
    } else {
        eff = 1.0;
    }
    const double rs = a.rad + b.rad;
    double dw = a.wfall - b.wfall;
    if (dw < 0.0) dw = -dw;
    return eff * PI * rs * rs * dw;
}

__device__ __forceinline__ bool kernel_needs_props(int type)
{
    return type == KERNEL_BROWN || type == KERNEL_SEDI || type == KERNEL_BROWN_SEDI;
}
__device__ __forceinline__ bool kernel_needs_fall(int type)
{
    return type == KERNEL_SEDI || type == KERNEL_BROWN_SEDI;
}

// props of one particle for kernel `type` (zeros for the size-only kernels)
__device__ __forceinline__ PartProps props_for(int type, double v, double m, const EnvDerived &e)
{
    if (type == KERNEL_BROWN) return particle_props<false>(v, m, e);
    if (kernel_needs_fall(type)) return particle_props<true>(v, m, e);
    return PartProps{0, 0, 0, 0, 0};
}

__device__ inline double kernel_pair(int type, double v1, double v2, const PartProps &a,
                                     const PartProps &b, const KernelParams &kp)
{
    switch (type) {
    case KERNEL_BROWN: return brown_pair(a, b);
    case KERNEL_SEDI: return sedi_pair(a, b);
    case KERNEL_BROWN_SEDI: return brown_pair(a, b) + sedi_pair(a, b);
    case KERNEL_CONSTANT: return kp.const_k;
    case KERNEL_ADDITIVE: return kp.additive_beta * (v1 + v2);
    default: return 0.0;
    }
}

// bin b has v_edge[b] <= v < v_edge[b+1]; out-of-range clamps to the end bins
template <typename EdgePtr>
__device__ __forceinline__ int bin_of(EdgePtr v_edge, int n_bin, double v)
{
    int lo = 0, hi = n_bin;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (v >= v_edge[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------ condensation
//
// EXTENSION: the growth ODE has no call site on the reference's timestep
// (condense_equilib_particles is called only inside the optional optics
// diagnostics, interface/wrf_pmc_driver.F90:1181,1201).  Seinfeld & Pandis
// eq. 17.70 with kappa-Koehler equilibrium:
//   D dD/dt = 4 (RH - a_w exp(A/D)) / (F_d + F_k),
//   a_w = (D^3 - Dd^3)/(D^3 - Dd^3 (1 - kappa)).

constexpr double WATER_MW = 18e-3;
constexpr double WATER_DENSITY = 1000.0;
constexpr double WATER_SURF_ENG = 0.073;
constexpr double WATER_LATENT = 2.272e6;
constexpr double AIR_CP = 1005.0;

struct CondEnv {
    double rh, kelvin_a, fd_coef, fk, dv, ka, c_dv, c_ka;
};

__device__ inline CondEnv cond_env_derive(double temp, double pressure, double rh)
{
    CondEnv e;
    const double R = UNIV_GAS;
    const double p_sat =
        611.0 * det_exp(2.302585092994046 * (7.45 * (temp - 273.15) / (temp - 38.0)));
    const double dv = 0.211e-4 / (pressure / 101325.0) * det_exp(1.94 * det_log(temp / 273.0));
    const double ka = 1e-3 * (4.39 + 0.071 * temp);
    const double rho_air = AIR_MW * pressure / (R * temp);
    e.rh = rh;
    e.kelvin_a = 4.0 * WATER_SURF_ENG * WATER_MW / (R * temp * WATER_DENSITY);
    e.fd_coef = WATER_DENSITY * R * temp / (p_sat * WATER_MW);
    e.fk = WATER_LATENT * WATER_DENSITY / temp * (WATER_LATENT * WATER_MW / (R * temp) - 1.0);
    e.dv = dv;
    e.ka = ka;
    e.c_dv = 2.0 * dv * __builtin_sqrt(2.0 * PI * WATER_MW / (R * temp));
    e.c_ka = 2.0 * ka / (rho_air * AIR_CP) * __builtin_sqrt(2.0 * PI * AIR_MW / (R * temp));
    return e;
}

// D * dD/dt / D, i.e. dD/dt, at wet diameter d
__device__ inline double cond_rate(double d, double dd3, double kappa, const CondEnv &e)
{
    const double d3 = d * d * d;
    const double aw = (d3 - dd3) / (d3 - dd3 * (1.0 - kappa));
    const double seq = aw * det_exp(e.kelvin_a / d);
    const double dvp = e.dv / (1.0 + e.c_dv / d);
    const double kap = e.ka / (1.0 + e.c_ka / d);
    return 4.0 * (e.rh - seq) / (e.fd_coef / dvp + e.fk / kap) / d;
}

__device__ inline double cond_equilibrium(double dd, double dd3, double kappa, double d_min,
                                          const CondEnv &e)
{
    const double gf3 = 1.0 + kappa * e.rh / (1.0 - e.rh);
    double b = dd * det_cbrt(gf3);
    if (b < d_min) b = d_min;
    double a = d_min;
    double fa = cond_rate(a, dd3, kappa, e), fb = cond_rate(b, dd3, kappa, e);
    if (!(fa > 0.0) || !(fb < 0.0)) return fb >= 0.0 ? b : a;
    for (int it = 0; it < 12; it++) {
        const double x = (a * fb - b * fa) / (fb - fa);
        const double fx = cond_rate(x, dd3, kappa, e);
        if ((fx > 0.0) == (fb > 0.0)) { b = x; fb = fx; fa = fa * 0.5; }
        else { a = b; fa = fb; b = x; fb = fx; }
    }
    return b;
}

// ------------------------------------------------------------ dry deposition
//
// Deposition velocity of one particle, Zhang et al. (2001) as written in
// interface/wrf_pmc_dep_aero.F90 (updated_version = .true.): slip_correction_factor
// :377-386, calculate_vs :265-276, compute_brownian_diff :409-420, calculate_st :390-404,
// calculate_rs :280-357, calculate_vd :361-372.  Default-real literals of the reference
// (.2, 0.8, 0.4, 2.0/3.0, 1.257, .4, 1.1) are the doubles those single-precision values
// convert to; x**2.0d0 and x**.5 are x*x and sqrt(x), the other powers exp(y log x).

constexpr double AIR_DYN_VISC = 1.78e-5; // const%air_dyn_visc [upstream]

__device__ __forceinline__ double det_pow(double x, double y) { return det_exp(y * det_log(x)); }

__device__ inline double dep_slip(double d_p)
{
    const double lambda = .065e-6;
    return 1.0 + ((2.0 * lambda) / d_p) *
                     ((double)1.257f + (double).4f * det_exp(-((double)1.1f * d_p) / (2.0 * lambda)));
}

__device__ inline double dep_velocity(double diameter, double density, double temp, double rrho,
                                      double ustar, double ra, double alpha, double big_a)
{
    const double c_c = dep_slip(diameter);
    const double vs = ((diameter * diameter) * density * GRAV * c_c) / (18.0 * AIR_DYN_VISC);
    const double diff = (BOLTZMANN * temp * c_c) / (3.0 * PI * AIR_DYN_VISC * diameter);
    const double sc = 1.45e-5 / diff;
    const double gamma = (double)(2.0f / 3.0f), c_b = (double).2f, nu = (double)0.8f, c_im = (double)0.4f;
    double st;
    if (big_a > 1.0) st = (vs * (ustar * ustar)) / (AIR_DYN_VISC * rrho);
    else st = (vs * ustar) / (GRAV * big_a);
    const double eb = c_b * det_pow(sc, -gamma);
    const double r = st / (alpha + st);
    const double eim = c_im * (r * r);
    const double ein = .5 * det_pow(diameter / big_a, nu);
    const double r1 = det_exp(-__builtin_sqrt(st));
    const double rs = 1.0 / (3.0 * ustar * (eb + eim + ein) * r1);
    const double tmp = ra + rs + ra * rs * vs;
    return (1.0 / tmp) + vs;
}

} // namespace pmcx
