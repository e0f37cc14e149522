/*
 * pmco_math.h -- ORACLE (test infrastructure, not product code).
 *
 * Deterministic scalar building blocks of the CPU oracle: the Philox4x32-10
 * counter RNG and exp/log/cbrt/inverse-normal routines written with nothing
 * but IEEE-754 +,-,*,/,sqrt and integer bit operations, so that a second
 * implementation that performs the same operations in the same order (the
 * HIP kernels under wrf-partmc_amd/csrc/) produces bit-identical doubles.
 * Compile with -ffp-contract=off: a fused multiply-add would change results.
 *
 * Reference anchors (what these replace):
 *   pmc_srand / pmc_random / pmc_rand_int / rand_poisson
 *       /root/reference/interface/wrf_pmc_init.F90:131 (seed),
 *       /root/reference/interface/wrf_pmc_trans_aero.F90:2107-2143 (uses).
 *   The reference draws from ONE sequential GSL stream per MPI rank
 *   (-DPMC_USE_GSL, WRFV3/arch/postamble_new:71); that stream cannot be
 *   reproduced cell-parallel, so the spec here is a counter-based stream
 *   keyed by (seed, cell, step, site) -- see DESIGN.md "RNG".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * include, link or call anything under oracle/.
 */
#ifndef PMCO_MATH_H
#define PMCO_MATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

/* ---------------------------------------------------------------- Philox */

#define PMCO_PHILOX_M0 0xD2511F53u
#define PMCO_PHILOX_M1 0xCD9E8D57u
#define PMCO_PHILOX_W0 0x9E3779B9u
#define PMCO_PHILOX_W1 0xBB67AE85u

/* Philox4x32-10 (Salmon et al., SC'11).  ctr is replaced by the output. */
static inline void pmco_philox4x32_10(uint32_t ctr[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)PMCO_PHILOX_M0 * ctr[0];
        uint64_t p1 = (uint64_t)PMCO_PHILOX_M1 * ctr[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
        k0 += PMCO_PHILOX_W0;
        k1 += PMCO_PHILOX_W1;
    }
}

/* RNG "sites": which draw of the algorithm a counter block belongs to. */
enum {
    PMCO_SITE_POISSON = 1,  /* n_samp of one (group,group) slot          */
    PMCO_SITE_CAND    = 2,  /* one candidate pair: i1, i2, accept         */
    PMCO_SITE_COAG    = 3,  /* weighted removal draws of an accepted pair */
    PMCO_SITE_HALVE   = 4,  /* rebalance: keep/drop one particle          */
    PMCO_SITE_GEN     = 5,  /* synthetic generator: one particle          */
    PMCO_SITE_GENCLS  = 6,  /* synthetic generator: weight class draw     */
    PMCO_SITE_EMIT_N  = 7,  /* scenario: number of new particles of a mode */
    PMCO_SITE_EMIT_P  = 8,  /* scenario: one new particle                  */
    PMCO_SITE_DILUTE  = 9,  /* scenario: keep/drop one particle (dilution) */
    PMCO_SITE_TRANS   = 10, /* transport: destination and copies of one particle */
    PMCO_SITE_DEPOSIT = 11, /* dry deposition: keep/drop one particle          */
    PMCO_SITE_SCALE   = 12, /* weight scaling: keep/drop one particle (preweight) */
    PMCO_SITE_RESAMPLE = 13 /* boundary resampling: target count of a class; one particle's removal */
};

/* One 128-bit block of the stream: counter = (cell, step, a, site<<28 | b),
 * key = 64-bit seed. */
static inline void pmco_rng_block(uint64_t seed, uint32_t cell, uint32_t step,
                                  uint32_t a, uint32_t site, uint32_t b,
                                  uint32_t out[4])
{
    out[0] = cell; out[1] = step; out[2] = a;
    out[3] = (site << 28) | (b & 0x0FFFFFFFu);
    pmco_philox4x32_10(out, (uint32_t)seed, (uint32_t)(seed >> 32));
}

/* 53-bit uniform in [0,1) from two words. */
static inline double pmco_u53(uint32_t hi, uint32_t lo)
{
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return (double)(x >> 11) * 0x1.0p-53;
}

/* Index in [0,n) from one word (multiply-high; bias <= n / 2^32). */
static inline uint32_t pmco_index(uint32_t w, uint32_t n)
{
    return (uint32_t)(((uint64_t)w * n) >> 32);
}

/* ------------------------------------------------------ bit reinterpret */

static inline uint64_t pmco_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double pmco_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* --------------------------------------------------------------- exp(x) */

static inline double pmco_exp(double x)
{
    if (x < -745.0) return 0.0;
    if (x > 709.0) return INFINITY;
    const double inv_ln2 = 1.4426950408889634;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    double t = x * inv_ln2;
    int k = (int)(t + (t < 0.0 ? -0.5 : 0.5));
    double kd = (double)k;
    double r = (x - kd * ln2_hi) - kd * ln2_lo;
    /* Taylor/Horner to degree 13, |r| <= 0.347 */
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
    if (k >= -1021) {
        return p * pmco_u2d((uint64_t)(k + 1023) << 52);
    }
    /* subnormal result: scale in two steps */
    return (p * pmco_u2d((uint64_t)(k + 1000 + 1023) << 52)) * 0x1.0p-1000;
}

/* --------------------------------------------------------------- log(x) */

/* natural log of a positive, finite, normal double */
static inline double pmco_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    uint64_t u = pmco_d2u(x);
    int e = (int)((u >> 52) & 0x7FF) - 1023;
    double m = pmco_u2d((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
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
    double ed = (double)e;
    return ed * ln2_hi + (s * q + ed * ln2_lo);
}

/* -------------------------------------------------------------- cbrt(x) */

/* x^(-1/3) of a positive, finite, normal double: bit-level first guess
 * (3.5% error) and three division-free third-order corrections. */
static inline double pmco_rcbrt(double x)
{
    uint64_t u = 0x553EEEAD3BAB5000ull - pmco_d2u(x) / 3;
    double y = pmco_u2d(u);
    for (int i = 0; i < 3; i++) {
        double h = 1.0 - x * (y * y * y);
        y = y + y * h * (1.0 / 3.0 + h * (2.0 / 9.0));
    }
    return y;
}

/* cube root of a positive, finite, normal double */
static inline double pmco_cbrt(double x)
{
    if (x <= 0.0) return 0.0;
    double y = pmco_rcbrt(x);
    return x * y * y;
}

/* ------------------------------------------- inverse normal CDF (Acklam) */

/* z with Phi(z) = p, 0 < p < 1; relative error about 1e-9 */
static inline double pmco_norminv(double p)
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
        double q = sqrt(-2.0 * pmco_log(p));
        return (((((c0 * q + c1) * q + c2) * q + c3) * q + c4) * q + c5) /
               ((((d0 * q + d1) * q + d2) * q + d3) * q + 1.0);
    }
    if (p <= 1.0 - p_low) {
        double q = p - 0.5;
        double r = q * q;
        return (((((a0 * r + a1) * r + a2) * r + a3) * r + a4) * r + a5) * q /
               (((((b0 * r + b1) * r + b2) * r + b3) * r + b4) * r + 1.0);
    }
    {
        double q = sqrt(-2.0 * pmco_log(1.0 - p));
        return -(((((c0 * q + c1) * q + c2) * q + c3) * q + c4) * q + c5) /
               ((((d0 * q + d1) * q + d2) * q + d3) * q + 1.0);
    }
}

/* -------------------------------------------------------------- Poisson */

#define PMCO_POISSON_WALK_MAX 200.0
#define PMCO_NSAMP_MAX 0x00FFFFFF

/* Poisson variate from one 53-bit uniform: CDF walk below
 * PMCO_POISSON_WALK_MAX, rounded normal approximation above it (the
 * no-GSL branch of PartMC's rand_poisson uses the same normal form). */
static inline int32_t pmco_poisson(double mean, double u)
{
    if (!(mean > 0.0)) return 0;
    if (mean < PMCO_POISSON_WALK_MAX) {
        double p = pmco_exp(-mean);
        double F = p;
        int32_t n = 0;
        while (u > F && n < 1000) {
            n = n + 1;
            /* n <= 64: multiply by the correctly rounded 1/n (the device keeps these in a
             * table; a division costs it ~35 instructions per step of the walk) */
            if (n <= 64) p = p * (mean * (1.0 / (double)n));
            else p = p * (mean / (double)n);
            F = F + p;
        }
        return n;
    }
    double z = pmco_norminv(u);
    double x = mean + sqrt(mean) * z + 0.5;
    if (x < 0.0) return 0;
    if (x > (double)PMCO_NSAMP_MAX) return PMCO_NSAMP_MAX;
    return (int32_t)x;
}

/* x^y for x > 0 from the deterministic exp and log */
static inline double pmco_pow(double x, double y) { return pmco_exp(y * pmco_log(x)); }

/* ------------------------------------------------------------------------
 * Dry deposition velocity of one particle, Zhang et al. (2001) as written in
 * interface/wrf_pmc_dep_aero.F90 (updated_version = .true., :313-329):
 *   slip_correction_factor :377-386   calculate_vs :265-276
 *   compute_brownian_diff  :409-420   calculate_st :390-404
 *   calculate_rs           :280-357   calculate_vd :361-372
 * The reference writes several constants as default-real literals (.2, 0.8,
 * 0.4, 2.0/3.0, 1.257, .4, 1.1); they are reproduced as the doubles those
 * single-precision values convert to.  const%air_dyn_visc, const%grav,
 * const%boltzmann are PartMC's [upstream]: 1.78e-5, 9.80665, 1.3806505e-23.
 * x**2.0d0 and x**.5 are taken as x*x and sqrt(x); the other powers are
 * exp(y log x) with the deterministic exp/log.  alpha is the land-use impaction
 * parameter, big_a the collector radius A(season, land use) (:113-130); the
 * land-use gamma is overwritten by the reference (:325) and therefore unused.
 * ---------------------------------------------------------------------- */
#define PMCO_AIR_DYN_VISC 1.78e-5
#define PMCO_DEP_GRAV 9.80665
#define PMCO_DEP_BOLTZMANN 1.3806505e-23
#define PMCO_DEP_PI 3.14159265358979323846

static inline double pmco_dep_slip(double d_p)
{
    const double lambda = .065e-6;
    return 1.0 + ((2.0 * lambda) / d_p) *
                     ((double)1.257f + (double).4f * pmco_exp(-((double)1.1f * d_p) / (2.0 * lambda)));
}

static inline double pmco_dep_velocity(double diameter, double density, double temp, double rrho,
                                       double ustar, double ra, double alpha, double big_a)
{
    const double c_c = pmco_dep_slip(diameter);
    const double vs = ((diameter * diameter) * density * PMCO_DEP_GRAV * c_c) / (18.0 * PMCO_AIR_DYN_VISC);
    const double diff = (PMCO_DEP_BOLTZMANN * temp * c_c) / (3.0 * PMCO_DEP_PI * PMCO_AIR_DYN_VISC * diameter);
    const double sc = 1.45e-5 / diff;
    const double gamma = (double)(2.0f / 3.0f), c_b = (double).2f, nu = (double)0.8f, c_im = (double)0.4f;
    double st;
    if (big_a > 1.0) st = (vs * (ustar * ustar)) / (PMCO_AIR_DYN_VISC * rrho);
    else st = (vs * ustar) / (PMCO_DEP_GRAV * big_a);
    const double eb = c_b * pmco_pow(sc, -gamma);
    const double r = st / (alpha + st);
    const double eim = c_im * (r * r);
    const double ein = .5 * pmco_pow(diameter / big_a, nu);
    const double r1 = pmco_exp(-sqrt(st));
    const double rs = 1.0 / (3.0 * ustar * (eb + eim + ein) * r1);
    const double tmp = ra + rs + ra * rs * vs;
    return (1.0 / tmp) + vs;
}

#endif /* PMCO_MATH_H */
