/*
 * pmco.c -- ORACLE (test infrastructure, not product code).  See pmco.h.
 *
 * Serial restatement of mc_coag and its neighbours on the WRF-PartMC
 * timestep.  Straight loops over explicit per-(class,bin) lists; no attempt
 * at speed beyond what the cpu_baseline timing needs.
 */
#include "pmco.h"
#include "pmco_math.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------
 * Physical constants (PartMC constants.F90 values; upstream, uncited).
 * ---------------------------------------------------------------------- */
#define PMCO_PI 3.14159265358979323846
#define PMCO_BOLTZMANN 1.3806505e-23
#define PMCO_AVOGADRO 6.02214179e23
#define PMCO_UNIV_GAS 8.314472
#define PMCO_AIR_MW 2.89644e-2
#define PMCO_GRAV 9.80665

/* exported wrappers of the inline building blocks (for the tests) */
void pmco_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    out[0] = ctr[0]; out[1] = ctr[1]; out[2] = ctr[2]; out[3] = ctr[3];
    pmco_philox4x32_10(out, key[0], key[1]);
}
double pmco_exp_(double x) { return pmco_exp(x); }
double pmco_log_(double x) { return pmco_log(x); }
double pmco_cbrt_(double x) { return pmco_cbrt(x); }
double pmco_norminv_(double p) { return pmco_norminv(p); }
int32_t pmco_poisson_(double mean, double u) { return pmco_poisson(mean, u); }
double pmco_dep_velocity_(double diameter, double density, double temp, double rrho, double ustar,
                          double ra, double alpha, double big_a)
{
    return pmco_dep_velocity(diameter, density, temp, rrho, ustar, ra, alpha, big_a);
}

/* ------------------------------------------------------------------------
 * Coagulation kernels.
 *
 * Brownian: Fuchs interpolation between free-molecular and continuum
 * regimes in the form of Jacobson (2005) eq. 15.33, which is the form
 * PartMC's kernel_brown uses [upstream, uncited] and, up to the Cunningham
 * slip constants, the form of WRFV3/chem/module_mosaic_coag.F:898-936
 * (1.257/0.42/1.1 there; 1.249/0.42/0.87 here, Jacobson's values).
 * ---------------------------------------------------------------------- */


/* ---- particle%component(:) (see pmco_cell) ---- */
static void comp_copy(pmco_cell *d, int at, const pmco_cell *s, int from)
{
    if (!d->n_comp_max) return;
    const int K = d->n_comp_max;
    d->comp_n[at] = s->comp_n[from];
    for (int k = 0; k < K; k++) {
        d->comp_src[(size_t)at * K + k] = s->comp_src[(size_t)from * K + k];
        d->comp_time[(size_t)at * K + k] = s->comp_time[(size_t)from * K + k];
    }
}

static void comp_single(pmco_cell *c, int i, int source, double t)
{
    if (!c->n_comp_max) return;
    const int K = c->n_comp_max;
    c->comp_n[i] = 1;
    for (int k = 0; k < K; k++) { c->comp_src[(size_t)i * K + k] = -1; c->comp_time[(size_t)i * K + k] = 0.0; }
    c->comp_src[(size_t)i * K] = source;
    c->comp_time[(size_t)i * K] = t;
}

/* list(first) ++ list(second), cut at n_comp_max, becomes the list of slot pn (one of the two) */
static void comp_merge(pmco_cell *c, int pn, int first, int second)
{
    if (!c->n_comp_max) return;
    const int K = c->n_comp_max;
    int32_t src[64]; double tm[64];
    int n = 0;
    for (int k = 0; k < c->comp_n[first] && n < K; k++, n++) {
        src[n] = c->comp_src[(size_t)first * K + k]; tm[n] = c->comp_time[(size_t)first * K + k];
    }
    for (int k = 0; k < c->comp_n[second] && n < K; k++, n++) {
        src[n] = c->comp_src[(size_t)second * K + k]; tm[n] = c->comp_time[(size_t)second * K + k];
    }
    c->comp_n[pn] = n;
    for (int k = 0; k < K; k++) {
        c->comp_src[(size_t)pn * K + k] = k < n ? src[k] : -1;
        c->comp_time[(size_t)pn * K + k] = k < n ? tm[k] : 0.0;
    }
}

typedef struct {
    double kT;          /* k_B T                              */
    double viscosd;     /* dynamic viscosity of air, kg/m/s   */
    double gasfreepath; /* mean free path of air, m           */
    double rhoair;      /* dry-air density, kg/m^3            */
    /* per-cell constants that keep divisions out of the per-particle code */
    double inv_gfp, c_diff, c_speed, c_inv_speed, c_fall;
} env_derived;

typedef struct {
    double rad, diffus, speedsq, deltasq, wfall;
} part_props;

static void env_derive(double temp, double pressure, env_derived *e)
{
    e->kT = PMCO_BOLTZMANN * temp;
    e->rhoair = PMCO_AIR_MW * pressure / (PMCO_UNIV_GAS * temp);
    double tr = temp / 296.16;
    e->viscosd = (1.8325e-5 * (296.16 + 120.0) / (temp + 120.0)) * (tr * sqrt(tr));
    double viscosk = e->viscosd / e->rhoair;
    double gasspeed = sqrt(8.0 * PMCO_BOLTZMANN * temp * PMCO_AVOGADRO /
                           (PMCO_PI * PMCO_AIR_MW));
    e->gasfreepath = 2.0 * viscosk / gasspeed;
    e->inv_gfp = 1.0 / e->gasfreepath;
    e->c_diff = e->kT / (6.0 * PMCO_PI * e->viscosd);
    e->c_speed = 8.0 * e->kT / PMCO_PI;
    e->c_inv_speed = PMCO_PI / (8.0 * e->kT);
    e->c_fall = 2.0 * PMCO_GRAV / (9.0 * e->viscosd);
}

/* Per-particle quantities of the Fuchs form, from volume v and mass m.
 * Jacobson eq. 15.33 term by term:
 *   rad      = (3v/4pi)^(1/3)                 Kn = lambda/rad
 *   cunning  = 1 + Kn (1.249 + 0.42 exp(-0.87/Kn))
 *   diffus   = kT cunning / (6 pi rad mu)     speedsq = 8kT/(pi m)
 *   freepath = 8 diffus / (pi sqrt(speedsq))
 *   delta    = ((2rad+fp)^3 - (4rad^2+fp^2)^1.5)/(6 rad fp) - 2rad
 * written with 1/rad from the reciprocal cube root and 1/speed from a square
 * root, so one particle costs two divisions. */
static void particle_props(double v, double m, const env_derived *e, int sedi,
                           part_props *q)
{
    double x = v * (0.75 / PMCO_PI);
    double inv_rad = pmco_rcbrt(x);
    double rad = x * inv_rad * inv_rad;
    double knud = e->gasfreepath * inv_rad;
    double cunning = 1.0 + knud * (1.249 + 0.42 * pmco_exp(-0.87 * (rad * e->inv_gfp)));
    double diffus = e->c_diff * cunning * inv_rad;
    double speedsq = e->c_speed / m;
    double inv_speed = sqrt(m * e->c_inv_speed);
    double freepath = (8.0 / PMCO_PI) * diffus * inv_speed;
    double t1 = 2.0 * rad + freepath;
    double tmp1 = t1 * t1 * t1;
    double t2 = 4.0 * rad * rad + freepath * freepath;
    double tmp2 = t2 * sqrt(t2);
    double d = (tmp1 - tmp2) / (6.0 * rad * freepath) - 2.0 * rad;
    q->rad = rad;
    q->diffus = diffus;
    q->speedsq = speedsq;
    q->deltasq = d * d;
    q->wfall = 0.0;
    /* Stokes-Cunningham terminal velocity (sedimentation kernels only) */
    if (sedi) q->wfall = e->c_fall * (rad * rad) * (m / v - e->rhoair) * cunning;
}

/* K = 4 pi R D / (R/(R+delta) + 4D/(R c)), inner quotients cleared */
static double brown_pair(const part_props *a, const part_props *b)
{
    double R = a->rad + b->rad;
    double D = a->diffus + b->diffus;
    double Rd = R + sqrt(a->deltasq + b->deltasq);
    double Rc = R * sqrt(a->speedsq + b->speedsq);
    return (4.0 * PMCO_PI) * (R * D) * (Rd * Rc) / (R * Rc + 4.0 * D * Rd);
}

/* Gravitational collection: E * pi (r1+r2)^2 |w1-w2| with the Long (1974)
 * polynomial collection efficiency (as tabulated in Bott 1998, eq. 12).
 * The upstream coag_kernel_sedi uses a Hall (1980) efficiency table that is
 * not available here; this closed form is the documented substitute. */
static double sedi_pair(const part_props *a, const part_props *b)
{
    double r_big = a->rad > b->rad ? a->rad : b->rad;
    double r_small = a->rad > b->rad ? b->rad : a->rad;
    double eff;
    if (r_big <= 50.0e-6) {
        double rb_cm = r_big * 100.0, rs_cm = r_small * 100.0;
        eff = 4.5e4 * rb_cm * rb_cm * (1.0 - 3.0e-4 / rs_cm);
        if (eff < 1.0e-3) eff = 1.0e-3;
    } else {
        eff = 1.0;
    }
    double rs = a->rad + b->rad;
    double dw = a->wfall - b->wfall;
    if (dw < 0.0) dw = -dw;
    return eff * PMCO_PI * rs * rs * dw;
}

static int kernel_needs_props(int type)
{
    return type == PMCO_KERNEL_BROWN || type == PMCO_KERNEL_SEDI ||
           type == PMCO_KERNEL_BROWN_SEDI;
}
static int kernel_needs_fall(int type)
{
    return type == PMCO_KERNEL_SEDI || type == PMCO_KERNEL_BROWN_SEDI;
}

static double kernel_pair(int type, double v1, double v2, const part_props *a,
                          const part_props *b, const pmco_kernel_params *kp)
{
    switch (type) {
    case PMCO_KERNEL_BROWN: return brown_pair(a, b);
    case PMCO_KERNEL_SEDI: return sedi_pair(a, b);
    case PMCO_KERNEL_BROWN_SEDI: return brown_pair(a, b) + sedi_pair(a, b);
    case PMCO_KERNEL_CONSTANT: return kp->const_k;
    case PMCO_KERNEL_ADDITIVE: return kp->additive_beta * (v1 + v2);
    default: return 0.0;
    }
}

double pmco_kernel(int type, double v1, double m1, double v2, double m2,
                   double temp, double pressure, const pmco_kernel_params *kp)
{
    env_derived e;
    part_props a = {0}, b = {0};
    env_derive(temp, pressure, &e);
    if (kernel_needs_props(type)) {
        particle_props(v1, m1, &e, kernel_needs_fall(type), &a);
        particle_props(v2, m2, &e, kernel_needs_fall(type), &b);
    }
    return kernel_pair(type, v1, v2, &a, &b, kp);
}

/* ------------------------------------------------------------------------
 * Sort grid (aero_sorted's bin_grid) and kernel bounds
 * (est_k_minmax_binned_unweighted) [upstream, uncited].
 * ---------------------------------------------------------------------- */

void pmco_make_bin_tables(int n_bin, double r_min, double r_max, double *v_edge,
                          double *v_samp)
{
    double l0 = log(r_min), l1 = log(r_max);
    for (int k = 0; k <= n_bin; k++) {
        double r = exp(l0 + (l1 - l0) * (double)k / (double)n_bin);
        v_edge[k] = (4.0 / 3.0) * PMCO_PI * r * r * r;
    }
    for (int b = 0; b < n_bin; b++) {
        v_samp[3 * b + 0] = v_edge[b];
        v_samp[3 * b + 1] = sqrt(v_edge[b] * v_edge[b + 1]);
        v_samp[3 * b + 2] = v_edge[b + 1];
    }
}

/* bin b has v_edge[b] <= v < v_edge[b+1]; out-of-range volumes clamp */
int pmco_bin_of(const double *v_edge, int n_bin, double v)
{
    int lo = 0, hi = n_bin; /* invariant: v_edge[lo] <= v < v_edge[hi] (clamped) */
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (v >= v_edge[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

/* smallest float >= k (k >= 0, finite) */
float pmco_up_f32(double k)
{
    float f = (float)k;
    if ((double)f < k) {
        uint32_t u;
        memcpy(&u, &f, 4);
        u += 1u;
        memcpy(&f, &u, 4);
    }
    return f;
}

void pmco_kmax_table(int type, int n_bin, const double *v_samp, double rho_min,
                     double rho_max, double temp, double pressure,
                     const pmco_kernel_params *kp, double *kmax)
{
    env_derived e;
    env_derive(temp, pressure, &e);
    int np = n_bin * 6; /* (bin, sample, density) */
    part_props *pp = (part_props *)calloc((size_t)np, sizeof(part_props));
    if (kernel_needs_props(type)) {
        for (int b = 0; b < n_bin; b++)
            for (int s = 0; s < 3; s++)
                for (int d = 0; d < 2; d++) {
                    double v = v_samp[3 * b + s];
                    double rho = d ? rho_max : rho_min;
                    particle_props(v, v * rho, &e, kernel_needs_fall(type),
                                   &pp[(b * 3 + s) * 2 + d]);
                }
    }
    for (int b1 = 0; b1 < n_bin; b1++)
        for (int b2 = 0; b2 < n_bin; b2++) {
            double km = 0.0;
            for (int s1 = 0; s1 < 3; s1++)
                for (int s2 = 0; s2 < 3; s2++)
                    for (int d1 = 0; d1 < 2; d1++)
                        for (int d2 = 0; d2 < 2; d2++) {
                            double k = kernel_pair(
                                type, v_samp[3 * b1 + s1], v_samp[3 * b2 + s2],
                                &pp[(b1 * 3 + s1) * 2 + d1],
                                &pp[(b2 * 3 + s2) * 2 + d2], kp);
                            if (k > km) km = k;
                        }
            /* stored as the next float at or above the bound (the device keeps the table
             * in fp32: half the bytes, four bounds per 16-byte load) */
            kmax[b1 * n_bin + b2] = (double)pmco_up_f32(km);
        }
    free(pp);
}

/* ------------------------------------------------------------------------
 * Caches.
 * ---------------------------------------------------------------------- */

static void refresh_one(pmco_cell *c, int i)
{
    double vt = 0.0, ms = 0.0;
    for (int s = 0; s < c->n_spec; s++) {
        double v = c->vol[(size_t)s * c->cap + i];
        vt = vt + v;
        ms = ms + c->density[s] * v;
    }
    c->vtot[i] = vt;
    c->mass[i] = ms;
}

void pmco_refresh_caches(pmco_cell *c)
{
    for (int i = 0; i < *c->n_part; i++) refresh_one(c, i);
    *c->sort_valid = 0;
}

/* ------------------------------------------------------------------------
 * mc_coag.
 *
 * Spec (DESIGN.md "mc_coag semantics"): particles are sorted into groups
 * g = class*n_bin + bin, lists in ascending slot order.  Slots (pairs of
 * groups) are visited in the order c1, c2<=c1, b1, b2 (b2>=b1 when c1==c2)
 * [loop nest of PartMC mc_coag].  Per slot: n_samp ~ Poisson(k_max dt
 * n_pairs) [compute_n_samp]; per sample: uniform pair [find_rand_pair],
 * accept with K f / (k_max f_max) [maybe_coag_pair, num_conc_weighted_kernel],
 * weighted removal [coagulate_weighting]; flat weights only
 * (AERO_STATE_WEIGHT_FLAT_SPECIFIED, wrf_pmc_init.F90:252-253).
 * ---------------------------------------------------------------------- */

typedef struct { int32_t *e; int32_t n, cap; } ilist;

static void il_push(ilist *l, int32_t x)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 8;
        l->e = (int32_t *)realloc(l->e, sizeof(int32_t) * (size_t)l->cap);
    }
    l->e[l->n++] = x;
}

typedef struct { int32_t pn, po, rem_o, cc, pn_is_1; } coag_event;

static void compact_stable(pmco_cell *c, const unsigned char *dead, int n_old,
                              int n_dead);

/* Room left behind a group of n entries when the lists are laid out: 16 + n/8 behind every group,
 * the empty ones too, while perm has the entries for that (level 0: up to 512 groups and a cap that
 * leaves the device's offsets 16 bits).  With tens of weight classes (registry.partmc_trans:2-41: 40
 * classes x 80 bins, nearly all groups empty) or a cell near 65 280 slots perm holds less and the room
 * shrinks, first behind the empty groups, by the first level of a fixed ladder that fits:
 *   level 0: 16 + n/8 | 16   1: 16 + n/8 | 4   2: 16 + n/8 | 1   3: 16 + n/8 | 0
 *         4:  4 + n/8 | 0    5: n/8 | 0        6: 0 | 0        (populated | empty) */
int pmco_sort_slack(int n, int level)
{
    if (n == 0) return level == 0 ? 16 : level == 1 ? 4 : level == 2 ? 1 : 0;
    return (level <= 3 ? 16 : level == 4 ? 4 : 0) + (level <= 5 ? (n >> 3) : 0);
}
int pmco_sort_level(int N, int S3, int P, int G, int pcap)
{
    static const int base[7] = {16, 16, 16, 16, 4, 0, 0}, empty[7] = {16, 4, 1, 0, 0, 0, 0};
    for (int l = 0; l < 6; l++)
        if ((int64_t)N + S3 + (int64_t)base[l] * P + (int64_t)empty[l] * (G - P) <= pcap) return l;
    return 6;
}

/* aero_state_sort [upstream; wrf_pmc_init.F90:1279, trans_aero:801]: groups
 * g = class*n_bin + bin, every list in ascending slot order, laid out one
 * after the other in perm with pmco_sort_slack() free entries behind each (by the level of the
 * ladder that fits pcap). */
void pmco_sort(pmco_cell *c, int *status)
{
    const int B = c->n_bin, G = B * c->n_class, n = *c->n_part;
    int32_t *key = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int g = 0; g < G; g++) c->gcnt[g] = 0;
    for (int i = 0; i < n; i++) {
        const double v = c->vtot[i];
        if (v < c->v_edge[0] || v >= c->v_edge[B]) *status |= PMCO_ST_OUT_OF_GRID;
        key[i] = c->wclass[i] * B + pmco_bin_of(c->v_edge, B, v);
        c->gcnt[key[i]]++;
    }
    int32_t o = 0, S3 = 0, P = 0;
    for (int g = 0; g < G; g++) { S3 += c->gcnt[g] >> 3; P += c->gcnt[g] > 0; }
    const int level = pmco_sort_level(n, S3, P, G, c->pcap);
    for (int g = 0; g < G; g++) {
        c->goff[g] = o;
        o += c->gcnt[g] + pmco_sort_slack(c->gcnt[g], level);
        c->gcnt[g] = 0;
    }
    c->goff[G] = o; /* <= pcap */
    for (int i = 0; i < n; i++) c->perm[c->goff[key[i]] + c->gcnt[key[i]]++] = i;
    free(key);
    *c->sort_valid = 1;
}

/* One group's list during a call: the first nb entries live in perm, later
 * ones (appended after the group's room ran out) in ovf. */
typedef struct { int32_t *base; int32_t n, nb, room; ilist ovf; } glist;

static int32_t gl_get(const glist *l, int i) { return i < l->nb ? l->base[i] : l->ovf.e[i - l->nb]; }
static void gl_set(glist *l, int i, int32_t x)
{
    if (i < l->nb) l->base[i] = x; else l->ovf.e[i - l->nb] = x;
}
static void gl_remove(glist *l, int i) /* swap-with-last */
{
    const int last = l->n - 1;
    if (i != last) gl_set(l, i, gl_get(l, last));
    if (last >= l->nb) l->ovf.n--; else l->nb = last;
    l->n = last;
}
/* returns 1 when the entry went to the overflow list */
static int gl_append(glist *l, int32_t x)
{
    if (l->ovf.n == 0 && l->n < l->room) { l->base[l->n] = x; l->nb = ++l->n; return 0; }
    il_push(&l->ovf, x);
    l->n++;
    return 1;
}

int pmco_mc_coag(pmco_cell *c, int kernel_type, const pmco_kernel_params *kp,
                 double del_t, uint64_t seed, uint32_t cell_id, uint32_t step,
                 int32_t *n_samp_out, int32_t *n_coag_out)
{
    const int B = c->n_bin, C = c->n_class, G = B * C;
    const int n0 = *c->n_part;
    int status = 0;
    int64_t tot_samp = 0;
    int32_t tot_coag = 0;
    env_derived env;
    env_derive(c->temp, c->pressure, &env);
    const int use_props = kernel_needs_props(kernel_type);
    const int use_fall = kernel_needs_fall(kernel_type);

    coag_event *ev = (coag_event *)malloc(sizeof(coag_event) * (size_t)(c->max_events > 0 ? c->max_events : 1));
    int n_ev = 0, n_app = 0;

    /* the sort persists between calls while aero_state%valid_sort holds
     * (consulted by the reference at wrf_pmc_trans_aero.F90:2094,2237) */
    if (!*c->sort_valid) pmco_sort(c, &status);
    glist *L = (glist *)calloc((size_t)G, sizeof(glist));
    for (int g = 0; g < G; g++) {
        L[g].base = c->perm + c->goff[g];
        L[g].n = L[g].nb = c->gcnt[g];
        L[g].room = c->goff[g + 1] - c->goff[g];
    }

    int stop = 0;
    for (int c1 = 0; c1 < C && !stop; c1++)
    for (int c2 = 0; c2 <= c1 && !stop; c2++)
    for (int b1 = 0; b1 < B && !stop; b1++)
    for (int b2 = (c1 == c2 ? b1 : 0); b2 < B && !stop; b2++) {
        const int gA = c1 * B + b1, gB = c2 * B + b2;
        const int same = (gA == gB);
        glist *A = &L[gA], *Bl = &L[gB];
        if (A->n == 0 || Bl->n == 0 || (same && A->n < 2)) continue;
        const double n1 = c->mag[c1], n2 = c->mag[c2];
        const int cc = (n1 < n2) ? c1 : c2;       /* coag_dest_class, flat */
        const double nmin = (n1 < n2) ? n1 : n2;
        const double fmax = (n1 < n2) ? n2 : n1;  /* max_coag_num_conc_factor */
        const double k_max = c->kmax[b1 * B + b2] * fmax;
        const double n_possible = same ? (double)A->n * ((double)A->n - 1.0) / 2.0
                                       : (double)A->n * (double)Bl->n;
        const double mean = k_max * del_t * n_possible;
        const uint32_t slot = ((uint32_t)gA << 16) | (uint32_t)gB;
        uint32_t w[4];
        /* the slot's uniform: word gB & 3 of the block addressed by (gA, gB & ~3), 32 bits (the
         * resolution of the reference's GSL uniform): four partner bins of a row share a block */
        pmco_rng_block(seed, cell_id, step, slot & ~3u, PMCO_SITE_POISSON, 0, w);
        const int32_t ns = pmco_poisson(mean, (double)w[gB & 3] * 0x1.0p-32);
        tot_samp += ns;
        const double factor = n1 * n2 / nmin;     /* coag_num_conc_factor, flat */

        for (int32_t s = 0; s < ns; s++) {
            const int nA = A->n, nB = Bl->n;
            if (nA < 1 || nB < 1 || (same && nA < 2)) break;
            pmco_rng_block(seed, cell_id, step, slot, PMCO_SITE_CAND, (uint32_t)s, w);
            int i1 = (int)pmco_index(w[0], (uint32_t)nA), i2;
            if (same) {
                i2 = (int)pmco_index(w[1], (uint32_t)(nA - 1));
                if (i2 >= i1) i2++;
            } else {
                i2 = (int)pmco_index(w[1], (uint32_t)nB);
            }
            const int p1 = gl_get(A, i1), p2 = gl_get(Bl, i2);
            const double v1 = c->vtot[p1], v2 = c->vtot[p2];
            const double m1 = c->mass[p1], m2 = c->mass[p2];
            part_props q1 = {0}, q2 = {0};
            if (use_props) {
                particle_props(v1, m1, &env, use_fall, &q1);
                particle_props(v2, m2, &env, use_fall, &q2);
            }
            const double k = kernel_pair(kernel_type, v1, v2, &q1, &q2, kp);
            /* accept with probability K f / k_max, without forming the quotient */
            const double kf = k * factor;
            if (kf > k_max) status |= PMCO_ST_KMAX_EXCEEDED;
            if (!(pmco_u53(w[2], w[3]) * k_max < kf)) continue;

            /* ---- coagulate ---- */
            if (n_ev >= c->max_events) { status |= PMCO_ST_EVENT_OVERFLOW; stop = 1; break; }
            pmco_rng_block(seed, cell_id, step, slot, PMCO_SITE_COAG, (uint32_t)s, w);
            const int rem1 = pmco_u53(w[0], w[1]) < nmin / n1;
            const int rem2 = pmco_u53(w[2], w[3]) < nmin / n2;
            const double vn = v1 + v2, mn = m1 + m2;
            const int big1 = (v1 >= v2);
            int pn_is_1;
            if (rem1 && rem2) pn_is_1 = big1; else pn_is_1 = rem1;
            const int pn = pn_is_1 ? p1 : p2, po = pn_is_1 ? p2 : p1;
            const int rem_o = pn_is_1 ? rem2 : rem1;
            glist *Ln = pn_is_1 ? A : Bl, *Lo = pn_is_1 ? Bl : A;
            int in = pn_is_1 ? i1 : i2, io = pn_is_1 ? i2 : i1;
            const int g_old = pn_is_1 ? gA : gB;
            if (vn < c->v_edge[0] || vn >= c->v_edge[B]) status |= PMCO_ST_OUT_OF_GRID;
            const int g_new = cc * B + pmco_bin_of(c->v_edge, B, vn);
            /* n_app counts the appends that did not fit behind their group;
             * once that budget is spent any further move between groups stops
             * the step */
            if (g_new != g_old && n_app >= c->max_appends) {
                status |= PMCO_ST_APPEND_OVERFLOW; stop = 1; break;
            }
            ev[n_ev].pn = pn; ev[n_ev].po = po; ev[n_ev].rem_o = rem_o; ev[n_ev].cc = cc;
            ev[n_ev].pn_is_1 = pn_is_1;
            n_ev++;
            if (rem_o) {                       /* swap-with-last removal */
                const int last = Lo->n - 1;
                gl_remove(Lo, io);
                if (same && in == last) in = io;
            }
            if (g_new != g_old) {
                gl_remove(Ln, in);
                n_app += gl_append(&L[g_new], pn);
            }
            c->vtot[pn] = vn;
            c->mass[pn] = mn;
            tot_coag++;
        }
    }

    /* ---- apply the event log to the particle data, in order ---- */
    unsigned char *dead = (unsigned char *)calloc((size_t)(n0 > 0 ? n0 : 1), 1);
    int n_dead = 0;
    for (int e = 0; e < n_ev; e++) {
        const int pn = ev[e].pn, po = ev[e].po;
        for (int s = 0; s < c->n_spec; s++)
            c->vol[(size_t)s * c->cap + pn] += c->vol[(size_t)s * c->cap + po];
        for (int s = 0; s < c->n_source; s++)
            c->n_orig[(size_t)s * c->cap + pn] += c->n_orig[(size_t)s * c->cap + po];
        if (c->lct[po] < c->lct[pn]) c->lct[pn] = c->lct[po];
        if (c->gct[po] > c->gct[pn]) c->gct[pn] = c->gct[po];
        c->wclass[pn] = ev[e].cc;
        if (ev[e].pn_is_1) comp_merge(c, pn, pn, po); else comp_merge(c, pn, po, pn);
        if (ev[e].rem_o) { dead[po] = 1; n_dead++; }
    }
    /* vtot/mass of merged particles stay the sums of their parents' caches
     * (set at event time); they are recomputed from the species only by
     * pmco_refresh_caches (upload, condensation). */

    /* the lists stay valid for the next call unless some entry is held outside
     * perm; closing the holes renames the slots (monotone map), which the
     * entries follow */
    for (int g = 0; g < G; g++) {
        c->gcnt[g] = L[g].n;
        if (L[g].ovf.n > 0) *c->sort_valid = 0;
        free(L[g].ovf.e);
    }
    if (*c->sort_valid && n_dead > 0) {
        int32_t *below = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n0 + 1));
        below[0] = 0;
        for (int i = 0; i < n0; i++) below[i + 1] = below[i] + dead[i];
        for (int g = 0; g < G; g++)
            for (int j = 0; j < c->gcnt[g]; j++) {
                int32_t *e = &c->perm[c->goff[g] + j];
                *e -= below[*e];
            }
        free(below);
    }
    compact_stable(c, dead, n0, n_dead);

    free(L); free(ev); free(dead);
    *n_samp_out = tot_samp > 0x7FFFFFFF ? 0x7FFFFFFF : (int32_t)tot_samp;
    *n_coag_out = tot_coag;
    return status;
}

static void move_particle(pmco_cell *c, int dst, int src)
{
    for (int s = 0; s < c->n_spec; s++)
        c->vol[(size_t)s * c->cap + dst] = c->vol[(size_t)s * c->cap + src];
    for (int s = 0; s < c->n_source; s++)
        c->n_orig[(size_t)s * c->cap + dst] = c->n_orig[(size_t)s * c->cap + src];
    c->vtot[dst] = c->vtot[src];
    c->mass[dst] = c->mass[src];
    c->wclass[dst] = c->wclass[src];
    c->id[dst] = c->id[src];
    c->lct[dst] = c->lct[src];
    c->gct[dst] = c->gct[src];
    comp_copy(c, dst, c, src);
}

/* Close the holes left by removed particles, keeping the survivors' relative
 * order (stable).  [aero_particle_array removal moves the last particle into
 * the hole; the order of particles carries no physical meaning.  A stable
 * order is specified here so that WHEN holes are closed cannot change any
 * later result: the GPU side leaves tombstones and compacts lazily.] */
static void compact_stable(pmco_cell *c, const unsigned char *dead, int n_old,
                              int n_dead)
{
    int dst = 0;
    for (int i = 0; i < n_old; i++) {
        if (dead[i]) continue;
        if (dst != i) move_particle(c, dst, i);
        dst++;
    }
    (void)n_dead;
    *c->n_part = dst;
}

/* ------------------------------------------------------------------------
 * aero_state_rebalance (call: wrf_pmc_driver.F90:248-250).
 * [upstream, uncited: aero_state_double / aero_state_halve]
 * ---------------------------------------------------------------------- */
int pmco_rebalance(pmco_cell *c, const double *n_ideal, int allow_doubling,
                   int allow_halving, uint64_t seed, uint32_t cell_id,
                   uint32_t step, int64_t *next_id)
{
    int status = 0;
    for (int cl = 0; cl < c->n_class; cl++) {
        for (uint32_t iter = 0; iter < 32; iter++) {
            int n = *c->n_part, ncl = 0;
            for (int i = 0; i < n; i++) ncl += (c->wclass[i] == cl);
            if (allow_doubling && ncl > 0 && (double)ncl < n_ideal[cl] / 2.0) {
                if (n + ncl > c->cap) { status |= PMCO_ST_CAP_OVERFLOW; break; }
                int k = 0;
                for (int i = 0; i < n; i++) {
                    if (c->wclass[i] != cl) continue;
                    move_particle(c, n + k, i);
                    c->id[n + k] = *next_id + k;
                    k++;
                }
                *next_id += k;
                *c->n_part = n + k;
                c->mag[cl] = c->mag[cl] / 2.0;
                *c->sort_valid = 0;
            } else if (allow_halving && (double)ncl > n_ideal[cl] * 2.0) {
                unsigned char *dead = (unsigned char *)calloc((size_t)n, 1);
                int n_dead = 0;
                for (int i = 0; i < n; i++) {
                    if (c->wclass[i] != cl) continue;
                    uint32_t w[4];
                    pmco_rng_block(seed, cell_id, step, (uint32_t)i, PMCO_SITE_HALVE,
                                   (uint32_t)cl * 32u + iter, w);
                    if (pmco_u53(w[0], w[1]) < 0.5) { dead[i] = 1; n_dead++; }
                }
                compact_stable(c, dead, n, n_dead);
                free(dead);
                c->mag[cl] = c->mag[cl] * 2.0;
                *c->sort_valid = 0;
            } else {
                break;
            }
        }
    }
    return status;
}

/* ------------------------------------------------------------------------
 * partmc_process observables (wrf_pmc_driver.F90:783,838-857):
 *   dry_diameters = aero_state_dry_diameters   (water excluded)
 *   num_concs     = aero_state_num_concs       (= class magnitude, flat)
 *   masses        = aero_state_masses          (all species)
 *   tot_num_conc  = sum(num_concs); tot_mass_conc = sum(masses*num_concs)
 *   num_dist  = bin_grid_histogram_1d(diam_grid, dry_diameters, num_concs)
 *   mass_dist = bin_grid_histogram_1d(diam_grid, dry_diameters, num_concs*masses)
 * with diam_grid = 100 log bins on [1e-9,1e-3] m and histogram entries
 * divided by the bin's log-width (widths = log(e(i+1)) - log(e(i)),
 * same file :795-797).
 * ---------------------------------------------------------------------- */
void pmco_diagnostics(const pmco_cell *c, int i_water, int n_dbin,
                      const double *d_edge, const double *d_width,
                      double *tot_num, double *tot_mass, double *num_dist,
                      double *mass_dist)
{
    double tn = 0.0, tm = 0.0;
    for (int b = 0; b < n_dbin; b++) { num_dist[b] = 0.0; mass_dist[b] = 0.0; }
    for (int i = 0; i < *c->n_part; i++) {
        double vdry = 0.0, ms = 0.0;
        for (int s = 0; s < c->n_spec; s++) {
            double v = c->vol[(size_t)s * c->cap + i];
            if (s != i_water) vdry = vdry + v;
            ms = ms + c->density[s] * v;
        }
        double diam = 2.0 * pmco_cbrt(vdry * (0.75 / PMCO_PI));
        double nc = c->mag[c->wclass[i]];
        tn = tn + nc;
        tm = tm + ms * nc;
        if (diam >= d_edge[0] && diam < d_edge[n_dbin]) {
            int b = pmco_bin_of(d_edge, n_dbin, diam);
            num_dist[b] = num_dist[b] + nc / d_width[b];
            mass_dist[b] = mass_dist[b] + nc * ms / d_width[b];
        }
    }
    *tot_num = tn;
    *tot_mass = tm;
}

/* ------------------------------------------------------------------------
 * Synthetic population (shape of aero_state_add_aero_dist_sample for
 * log-normal modes, wrf_pmc_init.F90:951,1063,1764-1777).
 * ---------------------------------------------------------------------- */
void pmco_generate(pmco_cell *c, const pmco_gen_spec *g, int32_t n, uint64_t seed,
                   uint32_t cell_id, int64_t first_id, double create_time)
{
    for (int i = 0; i < n; i++) {
        uint32_t w[4];
        pmco_rng_block(seed, cell_id, 0u, (uint32_t)i, PMCO_SITE_GEN, 0, w);
        double um = (double)w[0] * 0x1.0p-32;
        int mode = 0;
        while (mode < g->n_mode - 1 && um >= g->mode_frac[mode]) mode++;
        double z = pmco_norminv(pmco_u53(w[2], w[3]));
        double r = pmco_exp(g->ln_gm_radius[mode] + g->ln_gsd[mode] * z);
        double v = (4.0 / 3.0) * PMCO_PI * r * r * r;
        for (int s = 0; s < c->n_spec; s++)
            c->vol[(size_t)s * c->cap + i] = g->vol_frac[mode * c->n_spec + s] * v;
        for (int s = 0; s < c->n_source; s++)
            c->n_orig[(size_t)s * c->cap + i] = (s == g->mode_source[mode]) ? 1 : 0;
        c->wclass[i] = g->mode_class[mode];
        c->id[i] = first_id + i;
        c->lct[i] = create_time;
        c->gct[i] = create_time;
        comp_single(c, i, g->mode_source[mode], create_time);
        refresh_one(c, i);
    }
    *c->n_part = n;
    *c->sort_valid = 0;
}

/* ------------------------------------------------------------------------
 * Condensation of water (EXTENSION, not on the reference's timestep).
 *
 * The reference links condense.o but calls only condense_equilib_particles,
 * and only inside the optional optics diagnostics
 * (interface/wrf_pmc_driver.F90:1181,1201); the growth ODE condense_particles
 * has no call site and no test: parity doubly unpinned.  The
 * BASELINE north_star nevertheless names "condensation growth as a
 * per-particle ODE sub-step", so this is a self-contained restatement of the
 * textbook growth law with kappa-Koehler equilibrium (Seinfeld & Pandis
 * eq. 17.70; Petters & Kreidenweis 2007), environment held fixed over the step:
 *
 *   D dD/dt = 4 (RH - a_w exp(A/D)) / (F_d + F_k)
 *   a_w = (D^3 - Dd^3) / (D^3 - Dd^3 (1 - kappa)),   A = 4 sigma M_w / (R T rho_w)
 *   F_d = rho_w R T / (p_sat D_v' M_w),  F_k = L rho_w / (k_a' T) (L M_w / (R T) - 1)
 *   D_v' = D_v / (1 + c_dv / D),  k_a' = k_a / (1 + c_ka / D)   (transition regime)
 *
 * integrated with n_sub forward-Euler sub-steps; a sub-step that would cross
 * the equilibrium diameter lands on it instead (8 Illinois regula-falsi
 * iterations), which keeps the stiff small-particle limit stable.
 * ---------------------------------------------------------------------- */

#define PMCO_WATER_MW 18e-3
#define PMCO_WATER_DENSITY 1000.0
#define PMCO_WATER_SURF_ENG 0.073
#define PMCO_WATER_LATENT 2.272e6
#define PMCO_AIR_CP 1005.0

typedef struct {
    double rh, kelvin_a, fd_coef, fk, dv, ka, c_dv, c_ka;
} cond_env;

static void cond_env_derive(double temp, double pressure, double rh, cond_env *e)
{
    const double R = PMCO_UNIV_GAS;
    double p_sat = 611.0 * pmco_exp(2.302585092994046 * (7.45 * (temp - 273.15) / (temp - 38.0)));
    double dv = 0.211e-4 / (pressure / 101325.0) * pmco_exp(1.94 * pmco_log(temp / 273.0));
    double ka = 1e-3 * (4.39 + 0.071 * temp);
    double rho_air = PMCO_AIR_MW * pressure / (R * temp);
    e->rh = rh;
    e->kelvin_a = 4.0 * PMCO_WATER_SURF_ENG * PMCO_WATER_MW / (R * temp * PMCO_WATER_DENSITY);
    e->fd_coef = PMCO_WATER_DENSITY * R * temp / (p_sat * PMCO_WATER_MW); /* F_d = fd_coef / D_v' */
    e->fk = PMCO_WATER_LATENT * PMCO_WATER_DENSITY / temp *
            (PMCO_WATER_LATENT * PMCO_WATER_MW / (R * temp) - 1.0);       /* F_k = fk / k_a'   */
    e->dv = dv;
    e->ka = ka;
    e->c_dv = 2.0 * dv * sqrt(2.0 * PMCO_PI * PMCO_WATER_MW / (R * temp));
    e->c_ka = 2.0 * ka / (rho_air * PMCO_AIR_CP) * sqrt(2.0 * PMCO_PI * PMCO_AIR_MW / (R * temp));
}

/* D * dD/dt at wet diameter d for a particle with dry diameter cubed dd3 */
static double cond_rate(double d, double dd3, double kappa, const cond_env *e)
{
    double d3 = d * d * d;
    double aw = (d3 - dd3) / (d3 - dd3 * (1.0 - kappa));
    double seq = aw * pmco_exp(e->kelvin_a / d);
    double dvp = e->dv / (1.0 + e->c_dv / d);
    double kap = e->ka / (1.0 + e->c_ka / d);
    return 4.0 * (e->rh - seq) / (e->fd_coef / dvp + e->fk / kap) / d;
}

/* equilibrium wet diameter: root of RH - a_w exp(A/D) on [d_min, d0], where d0 is the
 * kappa-Koehler diameter without the Kelvin term (the Kelvin term only lowers the root) */
static double cond_equilibrium(double dd, double dd3, double kappa, double d_min,
                               const cond_env *e)
{
    double gf3 = 1.0 + kappa * e->rh / (1.0 - e->rh);
    double b = dd * pmco_cbrt(gf3);
    if (b < d_min) b = d_min;
    double a = d_min;
    double fa = cond_rate(a, dd3, kappa, e), fb = cond_rate(b, dd3, kappa, e);
    if (!(fa > 0.0) || !(fb < 0.0)) return fb >= 0.0 ? b : a;
    for (int it = 0; it < 12; it++) {
        double x = (a * fb - b * fa) / (fb - fa);
        double fx = cond_rate(x, dd3, kappa, e);
        if ((fx > 0.0) == (fb > 0.0)) { b = x; fb = fx; fa = fa * 0.5; }
        else { a = b; fa = fb; b = x; fb = fx; }
    }
    return b;
}

void pmco_condense(pmco_cell *c, const double *kappa, int i_water, double rel_humid,
                   double del_t, int n_sub)
{
    cond_env e;
    if (rel_humid > 0.999) rel_humid = 0.999; /* the reference clamps to 0.95, driver:664 */
    cond_env_derive(c->temp, c->pressure, rel_humid, &e);
    const double h = del_t / (double)n_sub;
    for (int i = 0; i < *c->n_part; i++) {
        double vdry = 0.0, kv = 0.0;
        for (int s = 0; s < c->n_spec; s++) {
            if (s == i_water) continue;
            double v = c->vol[(size_t)s * c->cap + i];
            vdry = vdry + v;
            kv = kv + kappa[s] * v;
        }
        if (!(vdry > 0.0) || !(kv > 0.0)) continue;
        const double kap = kv / vdry;
        const double dd3 = vdry * (6.0 / PMCO_PI);
        const double dd = pmco_cbrt(dd3);
        double d3 = (vdry + c->vol[(size_t)i_water * c->cap + i]) * (6.0 / PMCO_PI);
        double d = pmco_cbrt(d3);
        const double d_min = dd * (1.0 + 1e-9);
        if (d < d_min) d = d_min;
        const double d_eq = cond_equilibrium(dd, dd3, kap, d_min, &e);
        for (int sub = 0; sub < n_sub; sub++) {
            const int below = d < d_eq;
            double dn = d + h * cond_rate(d, dd3, kap, &e);
            /* a sub-step that would pass the equilibrium diameter lands on it */
            if (below ? !(dn < d_eq) : !(dn > d_eq)) dn = d_eq;
            d = dn;
        }
        double vw = d * d * d * (PMCO_PI / 6.0) - vdry;
        if (vw < 0.0) vw = 0.0;
        c->vol[(size_t)i_water * c->cap + i] = vw;
        refresh_one(c, i);
    }
    *c->sort_valid = 0;
}

/* ------------------------------------------------------------------------
 * scenario_update_aero_state (interface/wrf_pmc_driver.F90:207-210)
 * [upstream, uncited: aero_state_add_aero_dist_sample for the emissions and
 * the background, aero_state_sample_particles for the dilution loss].
 *   emission   : per mode, Poisson(rate dt / magnitude(class)) new particles
 *   dilution   : every particle is lost with probability 1 - exp(-rate dt)
 *   background : per mode, Poisson((1 - exp(-rate dt)) N_bg / magnitude) new particles
 * New particles are appended in order; ids come from next_id.
 * ---------------------------------------------------------------------- */
/* aero_state_scale_weight(weight_ratio) of one class [upstream, uncited].
 *   ratio > 1 (and allow_halving, or the class is empty): prob_round(n (1 - 1 / ratio)) of the
 *     class's n particles leave, a uniformly random subset -- the count from one uniform (key
 *     0xFFFFFFFF), the subset by selection sampling in rank order: member t of n goes when
 *     u (n - t) < still to go;
 *   ratio < 1 (and allow_doubling, or the class is empty): every particle of the class stands for
 *     prob_round(1 / ratio) >= 1 particles -- its copies, under new ids, are appended in rank order;
 *   then the magnitude is multiplied by ratio.  The uniform of a particle is keyed by its rank
 *   among the cell's particles; bkey separates the callers (the transport's preweight: the
 *   class; an add: PMCO_SCALE_ADD | kind << 16 | mode << 8 | class).
 * Returns the number removed (ratio > 1) or added (ratio < 1). */
#define PMCO_SCALE_ADD 0x4000000u
static int scale_weight(pmco_cell *c, int cl, double ratio, int allow_doubling, int allow_halving,
                        uint64_t seed, uint32_t cell_id, uint32_t step, uint32_t bkey, int64_t *next_id,
                        int *status)
{
    const int n = *c->n_part;
    int n_cls = 0, changed = 0;
    for (int i = 0; i < n; i++) n_cls += (c->wclass[i] == cl);
    uint32_t w[4];
    if (ratio > 1.0 && (allow_halving || n_cls == 0)) {
        unsigned char *dead = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
        pmco_rng_block(seed, cell_id, step, 0xFFFFFFFFu, PMCO_SITE_SCALE, bkey, w);
        const double x_rm = (double)n_cls * (1.0 - 1.0 / ratio);
        const double x_fl = (double)(int64_t)x_rm;
        int m_rm = (int)x_fl + ((pmco_u53(w[0], w[1]) < x_rm - x_fl) ? 1 : 0);
        if (m_rm > n_cls) m_rm = n_cls;
        int t = 0;
        for (int i = 0; i < n && changed < m_rm; i++) {
            if (c->wclass[i] != cl) continue;
            pmco_rng_block(seed, cell_id, step, (uint32_t)i, PMCO_SITE_SCALE, bkey, w);
            if (pmco_u53(w[0], w[1]) * (double)(n_cls - t) < (double)(m_rm - changed)) { dead[i] = 1; changed++; }
            t++;
        }
        compact_stable(c, dead, n, changed);
        free(dead);
    } else if (ratio < 1.0 && (allow_doubling || n_cls == 0)) {
        /* (no particle stands for more than the cell has slots: the count stays an integer whatever the ratio) */
        const double x = 1.0 / ratio < (double)c->cap ? 1.0 / ratio : (double)c->cap, x_fl = (double)(int64_t)x;
        for (int i = 0; i < n; i++) {
            if (c->wclass[i] != cl) continue;
            pmco_rng_block(seed, cell_id, step, (uint32_t)i, PMCO_SITE_SCALE, bkey, w);
            const int copies = (int)x_fl + ((pmco_u53(w[0], w[1]) < x - x_fl) ? 1 : 0);
            for (int g = 1; g < copies; g++) {
                const int at = *c->n_part;
                const int64_t id = (*next_id)++;        /* (spent whether or not the copy finds room) */
                if (at >= c->cap) { *status |= PMCO_ST_CAP_OVERFLOW; continue; }
                move_particle(c, at, i);
                c->id[at] = id;
                *c->n_part = at + 1;
                changed++;
            }
        }
    } else {
        return 0;
    }
    c->mag[cl] = c->mag[cl] * ratio;
    *c->sort_valid = 0;
    return changed;
}

/* aero_state_prepare_weight_for_add [upstream, uncited]: about n_add particles are going to join
 * class cl; if that leaves the class outside [n_ideal / 2, 2 n_ideal] the weight is scaled so
 * that it comes to n_ideal.  Off unless the cell carries prep_n_ideal. */
static void prepare_weight_for_add(pmco_cell *c, int cl, double n_add, uint64_t seed, uint32_t cell_id,
                                   uint32_t step, uint32_t bkey, int64_t *next_id, int *status)
{
    if (!c->prep_n_ideal) return;
    int n_cls = 0;
    for (int i = 0; i < *c->n_part; i++) n_cls += (c->wclass[i] == cl);
    const double n_new = (double)n_cls + n_add, ideal = c->prep_n_ideal[cl];
    if (n_new == 0.0 || !(ideal > 0.0)) return;
    if (n_new < ideal / 2.0 || n_new > ideal * 2.0)
        scale_weight(c, cl, n_new / ideal, c->prep_doubling, c->prep_halving, seed, cell_id, step, bkey,
                     next_id, status);
}

/* means != NULL: the Poisson mean of every mode is given (negative: the mode is skipped) */
static int add_dist_sample_means(pmco_cell *c, const pmco_aero_dist *d, double factor, double char_factor, const double *means,
                                 int kind, double create_time, uint64_t seed, uint32_t cell_id,
                                 uint32_t step, int64_t *next_id, int *status)
{
    int added = 0;
    for (int m = 0; m < d->n_mode; m++) {
        uint32_t w[4];
        const int cl = d->mode_class[m];
        if (!means && d->num_conc[m] < 0.0) continue; /* the cell's distribution does not have this mode */
        if (!means) /* adjust the weight if necessary, then the mean at the weight that holds now */
            prepare_weight_for_add(c, cl, (d->num_conc[m] * factor / c->mag[cl]) * char_factor, seed, cell_id, step,
                                   PMCO_SCALE_ADD | ((uint32_t)kind << 16) | ((uint32_t)m << 8) | (uint32_t)cl,
                                   next_id, status);
        const double mean = means ? means[m] : d->num_conc[m] * factor / c->mag[cl];
        if (means && mean < 0.0) continue;
        pmco_rng_block(seed, cell_id, step, ((uint32_t)kind << 8) | (uint32_t)m, PMCO_SITE_EMIT_N, 0, w);
        int n = pmco_poisson(mean, pmco_u53(w[0], w[1]));
        const int n0 = *c->n_part;
        if (n0 + n > c->cap) { *status |= PMCO_ST_CAP_OVERFLOW; n = c->cap - n0; }
        for (int k = 0; k < n; k++) {
            const int i = n0 + k;
            pmco_rng_block(seed, cell_id, step, ((uint32_t)kind << 28) | ((uint32_t)m << 20) | (uint32_t)k,
                           PMCO_SITE_EMIT_P, 0, w);
            double z = pmco_norminv(pmco_u53(w[2], w[3]));
            double r = pmco_exp(d->ln_gm_radius[m] + d->ln_gsd[m] * z);
            double v = (4.0 / 3.0) * PMCO_PI * r * r * r;
            for (int s = 0; s < c->n_spec; s++)
                c->vol[(size_t)s * c->cap + i] = d->vol_frac[m * c->n_spec + s] * v;
            for (int s = 0; s < c->n_source; s++)
                c->n_orig[(size_t)s * c->cap + i] = (s == d->mode_source[m]) ? 1 : 0;
            c->wclass[i] = cl;
            c->id[i] = *next_id + k;
            c->lct[i] = create_time;
            c->gct[i] = create_time;
            comp_single(c, i, d->mode_source[m], create_time);
            refresh_one(c, i);
        }
        *next_id += n;
        *c->n_part = n0 + n;
        added += n;
    }
    return added;
}

/* char_factor: the characteristic_factor of the add -- the weight is prepared for char_factor times
 * the sample's mean, the particles expected over the sampling timescale rather than over this one
 * call (wrf_pmc_driver.F90:1794, 1916-1922: max(1, min(3600 s, elapsed_time)) / delta_t in the
 * sea-salt source, a copy of the library's emission code; 1 in the boundary refill,
 * wrf_pmc_trans_aero.F90:1016, 1031) */
static int add_dist_sample(pmco_cell *c, const pmco_aero_dist *d, double factor, double char_factor, int kind,
                           double create_time, uint64_t seed, uint32_t cell_id, uint32_t step,
                           int64_t *next_id, int *status)
{
    return add_dist_sample_means(c, d, factor, char_factor, NULL, kind, create_time, seed, cell_id, step, next_id,
                                 status);
}
static double characteristic_factor(double elapsed_time, double dt)
{
    double t = elapsed_time < 3600.0 ? elapsed_time : 3600.0;
    if (t < 1.0) t = 1.0;
    return t / dt;
}

int pmco_scenario_update(pmco_cell *c, double dt, double create_time, uint64_t seed,
                         uint32_t cell_id, uint32_t step, const pmco_aero_dist *emission,
                         double dilution_rate, const pmco_aero_dist *background,
                         int64_t *next_id, int32_t *n_emit, int32_t *n_dil_in,
                         int32_t *n_dil_out)
{
    return pmco_scenario_update_h(c, dt, create_time, seed, cell_id, step, emission, dilution_rate, background,
                                  next_id, n_emit, n_dil_in, n_dil_out, 0.0, 0.0);
}

/* the same with env_state%height and old_env_state%height (the reference hands both states to the
 * update, wrf_pmc_driver.F90:207-209): where the layer grew, p = exp(-rate dt) is multiplied by
 * old_height / height [upstream, uncited; as scenario_update_gas_state] */
int pmco_scenario_update_h(pmco_cell *c, double dt, double create_time, uint64_t seed,
                           uint32_t cell_id, uint32_t step, const pmco_aero_dist *emission,
                           double dilution_rate, const pmco_aero_dist *background,
                           int64_t *next_id, int32_t *n_emit, int32_t *n_dil_in,
                           int32_t *n_dil_out, double height, double old_height)
{
    int status = 0;
    *n_emit = *n_dil_in = *n_dil_out = 0;
    *c->sort_valid = 0;
    if (emission && emission->n_mode > 0)
        *n_emit = add_dist_sample(c, emission, dt, characteristic_factor(create_time, dt), 0, create_time, seed, cell_id, step, next_id,
                                  &status);
    const int grew = height > old_height && old_height > 0.0;
    if (dilution_rate > 0.0 || grew) {
        double p_keep = pmco_exp(-dilution_rate * dt);
        if (grew) p_keep = p_keep * old_height / height;
        const double p_loss = 1.0 - p_keep;
        const int n = *c->n_part;
        unsigned char *dead = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
        int nd = 0;
        for (int i = 0; i < n; i++) {
            uint32_t w[4];
            pmco_rng_block(seed, cell_id, step, (uint32_t)i, PMCO_SITE_DILUTE, 0, w);
            if (pmco_u53(w[0], w[1]) < p_loss) { dead[i] = 1; nd++; }
        }
        compact_stable(c, dead, n, nd);
        free(dead);
        *n_dil_out = nd;
        if (background && background->n_mode > 0)
            *n_dil_in = add_dist_sample(c, background, p_loss, characteristic_factor(create_time, dt), 1, create_time, seed, cell_id, step,
                                        next_id, &status);
    }
    return status;
}

/* ------------------------------------------------------------------------
 * Sea-salt source (interface/wrf_pmc_driver.F90:1767-1924).  The constants are the
 * reference's (in-tree): restated as values.
 * ---------------------------------------------------------------------- */
int pmco_seasalt_modes(int sea_salt_opt, double wind_speed, double *flux, double *gm_radius,
                       double *gsd_log10, double *rate_scale, double *vol_frac_na)
{
    *vol_frac_na = 22.990 / (22.990 + 35.450);                       /* driver:1830-1831 */
    if (sea_salt_opt == 1) {                                          /* Gong, MAM3 modes */
        /* default-real literals in the reference (driver:1819-1822, and the exponent 3.41 at
         * :1853): the doubles those single-precision values convert to */
        static const float num_flux[2] = {73.7095947f, 10.1706810f};
        static const float diam[2] = {2.03347408e-07f, 1.09840811e-06f};
        static const float gsd[2] = {2.0f, 2.2f};
        for (int m = 0; m < 2; m++) {
            flux[m] = (double)num_flux[m];
            gm_radius[m] = (double)diam[m] / 2;                       /* driver:1849 */
            gsd_log10[m] = log10((double)gsd[m]);                     /* driver:1850 */
        }
        *rate_scale = pow(wind_speed, (double)3.41f);                 /* driver:1853 */
        return 2;
    }
    if (sea_salt_opt == 2) {                                          /* Ovadnevaite 2014 */
        /* default-real literals in the reference (driver:1801-1805): the doubles those
         * single-precision values convert to */
        static const float sigma[5] = {1.37f, 1.5f, 1.42f, 1.53f, 1.85f};
        static const float cmd[5] = {.018f, .041f, .09f, .23f, .83f};
        static const float coef[5] = {104.5f, .0442f, 149.6f, 2.96f, .51f};
        static const double offset[5] = {1e5, 1e5, 1e5, 1e5, 2e5};
        static const float expon[5] = {.556f, 1.08f, .545f, .79f, .87f};
        const double Re = 3.1e6;                                      /* driver:1909 */
        for (int m = 0; m < 5; m++) {
            gm_radius[m] = .5 * (double)cmd[m] * 1e-6;                /* driver:1906 */
            gsd_log10[m] = log10((double)sigma[m]);
            flux[m] = (double)coef[m] * pow(Re - offset[m], (double)expon[m]);  /* driver:1910 */
        }
        *rate_scale = 1.0;
        return 5;
    }
    return 0;
}

int pmco_seasalt_emissions(pmco_cell *c, double dt, double create_time, uint64_t seed,
                           uint32_t cell_id, uint32_t step, int sea_salt_opt, double wind_speed,
                           double height, int i_na, int i_cl, const int32_t *mode_class,
                           const int32_t *mode_source, int64_t *next_id, int32_t *n_emit)
{
    double flux[5], gm[5], gsd[5], scale, f_na, rate[5], lg[5], ls[5];
    int32_t cls[5] = {0, 0, 0, 0, 0}, src[5] = {0, 0, 0, 0, 0};
    const int M = pmco_seasalt_modes(sea_salt_opt, wind_speed, flux, gm, gsd, &scale, &f_na);
    int status = 0;
    *n_emit = 0;
    if (M == 0) return 0;
    double *vf = (double *)calloc((size_t)M * c->n_spec, sizeof(double));
    for (int m = 0; m < M; m++) {
        rate[m] = flux[m] * scale / height;       /* p = emission_rate_scale dt / height, :1917 */
        lg[m] = pmco_log(gm[m]);
        ls[m] = gsd[m] * 2.302585092994046;   /* ln 10 */
        vf[m * c->n_spec + i_na] = f_na;
        vf[m * c->n_spec + i_cl] = 1.0 - f_na;
        if (mode_class) cls[m] = mode_class[m];
        if (mode_source) src[m] = mode_source[m];
    }
    pmco_aero_dist d = {M, rate, lg, ls, vf, cls, src};
    *c->sort_valid = 0;
    *n_emit = add_dist_sample(c, &d, dt, characteristic_factor(create_time, dt), 2, create_time, seed, cell_id, step, next_id, &status);
    free(vf);
    return status;
}

/* ------------------------------------------------------------------------
 * scenario_update_gas_state of one cell (pmco.h).  [UPSTREAM-UNCITED]
 * ---------------------------------------------------------------------- */
double pmco_interp_1d(int n, const double *x, const double *y, int y_stride, double xq)
{
    if (n <= 1 || xq <= x[0]) return y[0];
    if (xq >= x[n - 1]) return y[(size_t)(n - 1) * y_stride];
    int p = 0;
    while (p < n - 2 && xq >= x[p + 1]) p++;
    const double alpha = (xq - x[p]) / (x[p + 1] - x[p]);
    return (1.0 - alpha) * y[(size_t)p * y_stride] + alpha * y[(size_t)(p + 1) * y_stride];
}

void pmco_scenario_update_gas(int n_gas, double *mix_rat, double dt, double temp, double pressure,
                              double height, double old_height, const double *emission,
                              double emission_rate_scale, const double *background,
                              double dilution_rate)
{
    const double air_molar_den = pressure / (8.314472 * temp); /* env_state_air_molar_den */
    const double p_emit = emission_rate_scale * dt / height;
    /* the dilution half (exchange with the background, entrainment when the layer grows)
     * needs a background to exchange with: a scenario without one skips it whole */
    double p = 1.0;
    if (background) {
        p = pmco_exp(-dilution_rate * dt);
        if (height > old_height) p = p * old_height / height;
    }
    for (int g = 0; g < n_gas; g++) {
        double x = mix_rat[g];
        if (emission) x = x + p_emit * (emission[g] / air_molar_den * 1e9);
        if (background) {
            x = x * p;
            x = x + background[g] * (1.0 - p);
        }
        mix_rat[g] = x < 0.0 ? 0.0 : x;
    }
}

/* ------------------------------------------------------------------------
 * Dry deposition of one (surface) cell: dry_dep_aero_state
 * (interface/wrf_pmc_dep_aero.F90:154-192): deposition velocity of every
 * particle (compute_dep_vel :196-234, diameter and density of the whole
 * particle), removal probability (dt / dz) vd (compute_dep_prob :238-256,
 * option 1 as the reference has it), removal by one uniform per particle
 * (aero_particles_remove_by_dep :260-272).  The reference draws from its
 * sequential stream walking the particles backwards; here the draw of a
 * particle is addressed by its rank among the live particles.  Returns the
 * number removed; *unrealistic is set when a probability is outside (0, 1)
 * (the reference prints "Unrealistic deposition", :181-183).
 * ---------------------------------------------------------------------- */
int pmco_dry_deposition(pmco_cell *c, double dt, uint64_t seed, uint32_t cell_id, uint32_t step,
                        double aer_res_a, double ustar, double alpha, double big_a,
                        double height, double inv_density, int32_t *unrealistic)
{
    const int n = *c->n_part;
    unsigned char *dead = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int nd = 0;
    *unrealistic = 0;
    for (int i = 0; i < n; i++) {
        const double v = c->vtot[i], m = c->mass[i];
        const double diam = 2.0 * pmco_cbrt(v * (0.75 / PMCO_DEP_PI));
        const double dens = m / v;
        const double vd = pmco_dep_velocity(diam, dens, c->temp, inv_density, ustar, aer_res_a,
                                            alpha, big_a);
        const double prob = (dt / height) * vd;
        if (prob >= 1.0 || prob <= 0.0) *unrealistic = 1;
        uint32_t w[4];
        pmco_rng_block(seed, cell_id, step, (uint32_t)i, PMCO_SITE_DEPOSIT, 0, w);
        if (pmco_u53(w[0], w[1]) < prob) { dead[i] = 1; nd++; }
    }
    if (nd > 0) *c->sort_valid = 0;
    compact_stable(c, dead, n, nd);
    free(dead);
    return nd;
}

/* ------------------------------------------------------------------------
 * More partmc_process observables (interface/wrf_pmc_driver.F90):
 *   species_mass[s] = sum(aero_state_masses(include=species s) * num_concs)   :864-932
 *   source_num[k]   = aero_state_num_concs_by_source                          :858-861
 *   pm[0..2]        = PM1 / PM2.5 / PM10 dry mass concentration               :1153-1157
 * ---------------------------------------------------------------------- */
void pmco_diagnostics_ext(const pmco_cell *c, int i_water, double *species_mass,
                          double *source_num, double *pm)
{
    for (int s = 0; s < c->n_spec; s++) species_mass[s] = 0.0;
    for (int k = 0; k < c->n_source; k++) source_num[k] = 0.0;
    pm[0] = pm[1] = pm[2] = 0.0;
    for (int i = 0; i < *c->n_part; i++) {
        const double nc = c->mag[c->wclass[i]];
        double vdry = 0.0, mdry = 0.0;
        for (int s = 0; s < c->n_spec; s++) {
            const double v = c->vol[(size_t)s * c->cap + i];
            species_mass[s] = species_mass[s] + c->density[s] * v * nc;
            if (s != i_water) { vdry = vdry + v; mdry = mdry + c->density[s] * v; }
        }
        for (int k = 0; k < c->n_source; k++)
            if (c->n_orig[(size_t)k * c->cap + i] > 0) source_num[k] = source_num[k] + nc;
        const double diam = 2.0 * pmco_cbrt(vdry * (0.75 / PMCO_PI));
        if (diam < 1e-6) pm[0] = pm[0] + nc * mdry;
        if (diam < 2.5e-6) pm[1] = pm[1] + nc * mdry;
        if (diam < 10e-6) pm[2] = pm[2] + nc * mdry;
    }
}

/* ------------------------------------------------------------------------
 * Stochastic transport, per-particle form of
 *   aero_state_multisample                  wrf_pmc_trans_aero.F90:2351-2482
 *   aero_state_sample_particles_by_class    wrf_pmc_trans_aero.F90:2181-2322
 *
 * Reference, per source cell and weight class: destinations with prob > 0
 * (W, E, S, N with prob_advection + prob_diffusion, :2419-2445; every other
 * level of the column with prob_vert_diffusion, :2446-2460) are sorted by
 * probability (:2462-2465) and visited with adjusted probabilities
 * min(prob/remainder, 1) while remainder >= 1e-8 (:2466-2479); each visit
 * draws a Binomial/Poisson number of particles and moves them one by one,
 * adding/removing according to the ratio of the destination weight (scaled
 * by dz_source/dz_dest, :2214-2218) to the source weight (:2236-2262).
 *
 * Here the same law is applied particle by particle, which is what makes it
 * data-parallel: one uniform places a particle in [thr_lo, thr_hi) of one
 * destination (cumulative probabilities in the sorted order = the sequence of
 * adjusted Bernoulli trials) or leaves it where it is.  A selected particle
 * leaves its cell.  With r = weight_to' / weight_from:
 *   r == 1: one copy arrives                                   (:2254-2257)
 *   r  > 1: it arrives with probability 1/r                    (:2267-2273)
 *   r  < 1 (law 0; law 1, the default, is described at g_trans_law below):
 *           G >= 1 copies arrive, P(G = g) = r (1-r)^(g-1), capped at 4096 (a guard: 16 until round 3, which cost 18 % of the arrivals at r = 0.1);
 *           the last keeps the id, the others get new ids      (:2258-2266,2279-2283)
 *           [the reference reaches the same mean 1/r with repeated picks]
 * The placing uniform of the particle of rank i (among the cell's live particles) is 32 bits:
 * word i & 3 of the Philox block (seed, cell, step, i >> 2, site TRANS, 0) times 2^-32; a mover
 * with r != 1 draws the uniform of its copies from the block (…, i, site TRANS, 1).
 * Arrivals are appended to the destination in a fixed order: from the east,
 * west, north, south neighbour, then from levels 0..nz-1 of the column; inside
 * one source in the order of the particles' rank.
 * ---------------------------------------------------------------------- */

#define PMCO_TRANS_MAX_COPIES 4096

typedef struct { int dest; double lo, hi, ratio; } trans_dest;

/* Law 1 (the default; pmcx_set_transport_law): the r < 1 branch in the reference's own
 * law, by Poisson thinning of its picks (wrf_pmc_trans_aero.F90:2244-2300: Poisson(n p / r) picks
 * with replacement, each adds a copy and removes the picked particle with probability r).  The
 * removing picks (rate p) are the movers -- a mover towards r < 1 arrives ONCE, id kept; the
 * non-removing picks (rate mu = p (1 - r) / r) are copies from particles that STAY: behind the
 * movers' intervals the placing uniform has one more, of width w = 1 - exp(-sum mu) -- an emitter,
 * which draws a zero-truncated Poisson(sum mu) number of copies and deals them to the r < 1
 * destinations in proportion mu, every copy under a new id.  Where the two do not fit side by side
 * (sum p + w > 1) the emitters' interval is [1 - w, 1) and overlaps the movers': a particle in the
 * overlap sends its copies and moves -- the reference, too, can pick a particle without removal
 * before the pick that removes it -- so P(moves to d) = p_d and E[copies to d] = mu_d at every
 * ratio.  Mean n p / r as law 0, variance n p / r as the reference's. */
#define PMCO_TRANS_BOTH_BLOCK (1u << 20)   /* block of the copy count of a particle that moves and emits */
#define PMCO_TRANS_EMIT_WALK_MAX 32.0      /* above this sum of rates the zero-truncation is below 2^-46 */
static int emit_count(double sum_mu, double u2, int *clamped)
{
    int n;
    if (sum_mu > PMCO_TRANS_EMIT_WALK_MAX) {
        n = pmco_poisson(sum_mu, u2);
        if (n < 1) n = 1;
    } else {
        const double e = pmco_exp(-sum_mu);
        double term = e * sum_mu / (1.0 - e), cdf = term;   /* P(K = 1) of the zero-truncated Poisson */
        n = 1;
        while (u2 >= cdf && n < PMCO_TRANS_MAX_COPIES) {
            n++;
            term = term * sum_mu / (double)n;
            const double next = cdf + term;
            if (next == cdf) break;
            cdf = next;
        }
    }
    if (n > PMCO_TRANS_MAX_COPIES) { *clamped = 1; n = PMCO_TRANS_MAX_COPIES; }
    return n;
}
static int g_trans_law = 1;     /* the default, as the library's */
void pmco_set_transport_law(int law) { g_trans_law = law; }

/* destination table of (cell c, class cl): returns the number of entries */
static int g_trans_lateral = 0; /* pmco_transport_lateral: the reference's open lateral boundaries */

static int trans_table(const pmco_cell *cells, int c, int cl, int nx, int nz, int ny, int px, int py,
                       const double *prob_h, const double *prob_v, const double *vol, int n_class,
                       trans_dest *out)
{
    const int i = c % nx, k = (c / nx) % nz, j = c / (nx * nz);
    double p[4 + 128];
    int dcell[4 + 128], n = 0;
    const int di[4] = {-1, 1, 0, 0}, dj[4] = {0, 0, -1, 1};
    const int bx = (i == 0 || i == nx - 1), by = (j == 0 || j == ny - 1);
    if (g_trans_lateral && (bx || by)) {
        /* a lateral boundary cell only sends inwards, with min(prob, 1); a corner nothing
         * (wrf_pmc_trans_aero.F90:689-768) */
        if (bx && by) return 0;
        const int d = i == 0 ? 1 : i == nx - 1 ? 0 : j == 0 ? 3 : 2;
        double pr = prob_h[((size_t)c * n_class + cl) * 4 + d];
        if (pr > 1.0) pr = 1.0;
        if (!(pr > 0.0)) return 0;
        const int dc = ((j + dj[d]) * nz + k) * nx + (i + di[d]);
        out[0].dest = dc;
        out[0].lo = 0.0;
        out[0].hi = pr;
        out[0].ratio = (cells[dc].mag[cl] / (vol[c] / vol[dc])) / cells[c].mag[cl];
        return 1;
    }
    for (int d = 0; d < 4; d++) {
        int ii = i + di[d], jj = j + dj[d];
        if (ii < 0 || ii >= nx) { if (!px) continue; ii = (ii + nx) % nx; }
        if (jj < 0 || jj >= ny) { if (!py) continue; jj = (jj + ny) % ny; }
        const double pr = prob_h[((size_t)c * n_class + cl) * 4 + d];
        if (pr > 0.0) { p[n] = pr; dcell[n] = (jj * nz + k) * nx + ii; n++; }
    }
    for (int kk = 0; kk < nz; kk++) {
        if (kk == k) continue;
        const double pr = prob_v[((size_t)c * n_class + cl) * nz + kk];
        if (pr > 0.0) { p[n] = pr; dcell[n] = (j * nz + kk) * nx + i; n++; }
    }
    /* stable ascending sort by probability (:2462-2465) */
    int ord[4 + 128];
    for (int a = 0; a < n; a++) {
        int r = 0;
        for (int b = 0; b < n; b++) r += (p[b] < p[a]) || (p[b] == p[a] && b < a);
        ord[r] = a;
    }
    double rem = 1.0, cum = 0.0;
    for (int r = 0; r < n; r++) {
        const int a = ord[r];
        out[r].dest = dcell[a];
        out[r].lo = cum;
        if (rem >= 1e-8) {               /* remainder_eps, :2400 */
            cum = cum + p[a];
            if (cum > 1.0) cum = 1.0;
            rem = rem - p[a];
        }
        out[r].hi = cum;
        /* weight of the class in the destination scaled by dz_source/dz_dest (:2214-2218) */
        const double factor = vol[c] / vol[dcell[a]];
        out[r].ratio = (cells[dcell[a]].mag[cl] / factor) / cells[c].mag[cl];
    }
    return n;
}

int pmco_transport(pmco_cell *cells, int n_cells, int nx, int nz, int ny, int periodic_x,
                   int periodic_y, const double *prob_h, const double *prob_v,
                   const double *cell_volume, uint64_t seed, uint32_t cell_id_base,
                   uint32_t step, int64_t *next_id, int32_t *n_moved, int32_t *n_arrived)
{
    const int C = cells[0].n_class;
    int status = 0;
    /* a periodic direction one cell wide is closed: the cell would be its own neighbour */
    periodic_x = periodic_x && nx > 1;
    periodic_y = periodic_y && ny > 1;
    for (int k = 0; k < n_cells; k++) *cells[k].sort_valid = 0;
    /* phase 1: every particle decides; outbox entries (slot, dest, copies) in rank order */
    typedef struct { int32_t slot, dest, copies, stays; } entry;
    entry **box = (entry **)calloc((size_t)n_cells, sizeof(entry *));
    int *nbox = (int *)calloc((size_t)n_cells, sizeof(int));
    trans_dest *tab = (trans_dest *)malloc(sizeof(trans_dest) * (size_t)C * (4 + 128));
    int *ntab = (int *)malloc(sizeof(int) * (size_t)C);
    for (int c = 0; c < n_cells; c++) {
        pmco_cell *s = &cells[c];
        const int n = *s->n_part;
        size_t box_cap = (size_t)(n > 0 ? n : 1);
        box[c] = (entry *)malloc(sizeof(entry) * box_cap);
        int n_movers = 0, closed = 0;
        for (int cl = 0; cl < C; cl++)
            ntab[cl] = trans_table(cells, c, cl, nx, nz, ny, periodic_x, periodic_y, prob_h, prob_v,
                                   cell_volume, C, tab + (size_t)cl * (4 + 128));
        for (int i = 0; i < n; i++) {
            const int cl = s->wclass[i];
            const trans_dest *t = tab + (size_t)cl * (4 + 128);
            if (ntab[cl] == 0) continue;
            /* the destination uniform starts with an 8-bit prefix, sixteen particles to a Philox
             * block (byte i & 15 of the block addressed by i >> 4): prefix / 256 at or above the
             * class's total probability of leaving says the particle stays.  One the prefix cannot
             * tell for draws the rest from a block of its own -- u = (prefix + w0 2^-32) / 256, a
             * uniform of 40 bits -- and from the same block the uniform of its copies. */
            uint32_t w[4];
            pmco_rng_block(seed, cell_id_base + (uint32_t)c, step, (uint32_t)i >> 4, PMCO_SITE_TRANS, 0, w);
            const uint32_t p8 = (w[(i & 15) >> 2] >> (8 * (i & 3))) & 0xFFu;
            /* law 1: the emitters' interval behind the movers', or -- where the two do not fit side by
             * side -- at the top of the unit interval, overlapping them */
            const double s_hi = t[ntab[cl] - 1].hi;
            double sum_mu = 0.0, a_lo = s_hi, a_hi = s_hi;
            if (g_trans_law == 1) {
                for (int d = 0; d < ntab[cl]; d++)
                    if (t[d].ratio < 1.0) sum_mu = sum_mu + (t[d].hi - t[d].lo) * (1.0 - t[d].ratio) / t[d].ratio;
                const double w_emit = 1.0 - pmco_exp(-sum_mu);
                a_hi = a_lo + w_emit;
                if (a_hi > 1.0) { a_hi = 1.0; a_lo = 1.0 - w_emit; }
            }
            if (!((double)p8 * 0x1.0p-8 < a_hi)) continue;      /* stays */
            pmco_rng_block(seed, cell_id_base + (uint32_t)c, step, (uint32_t)i, PMCO_SITE_TRANS, 1, w);
            const double u = ((double)p8 + (double)w[0] * 0x1.0p-32) * 0x1.0p-8;
            const int moves = u < s_hi;
            const int emits = g_trans_law == 1 && !(u < a_lo) && u < a_hi;
            if (!moves && !emits) continue;      /* stays */
            /* the particle's entries: a mover's one, then an emitter's -- one per destination that
             * takes copies, in table order, with its count */
            entry ent[1 + 4 + 128];
            int n_ent = 0, n_emit = 0;
            if (emits) {                    /* it sends copies (and stays, unless it also moves) */
                double u2 = pmco_u53(w[2], w[3]);
                if (moves) {
                    uint32_t v[4];
                    pmco_rng_block(seed, cell_id_base + (uint32_t)c, step, (uint32_t)i, PMCO_SITE_TRANS, PMCO_TRANS_BOTH_BLOCK, v);
                    u2 = pmco_u53(v[0], v[1]);
                }
                int clamped = 0;
                n_emit = emit_count(sum_mu, u2, &clamped);
                if (clamped) status |= PMCO_ST_TRANS_COPIES;
            }
            if (moves) {
                int d = 0;
                while (!(u < t[d].hi)) d++;
                const double r = t[d].ratio;
                int copies = 1;
                if (r != 1.0) {
                    const double u2 = pmco_u53(w[2], w[3]);
                    if (r > 1.0) copies = (u2 < 1.0 / r) ? 1 : 0;
                    else if (g_trans_law == 0) {
                        copies = 1 + (int)(pmco_log(1.0 - u2) / pmco_log(1.0 - r));
                        if (copies > PMCO_TRANS_MAX_COPIES) copies = PMCO_TRANS_MAX_COPIES;
                    }   /* (law 1: a mover arrives once; the other 1 / r - 1 come from the emitters) */
                }
                ent[n_ent].slot = i; ent[n_ent].dest = t[d].dest; ent[n_ent].copies = copies; ent[n_ent].stays = 0;
                n_ent++;
            }
            if (n_emit > 0) {
                int deal[4 + 128];
                for (int y = 0; y < ntab[cl]; y++) deal[y] = 0;
                for (int q = 0; q < n_emit; q++) {
                    uint32_t v[4];
                    pmco_rng_block(seed, cell_id_base + (uint32_t)c, step, (uint32_t)i, PMCO_SITE_TRANS, 2u + (uint32_t)(q >> 1), v);
                    const double x = pmco_u53(v[2 * (q & 1)], v[2 * (q & 1) + 1]) * sum_mu;
                    double cm = 0.0;
                    int d = -1, last = -1;
                    for (int y = 0; y < ntab[cl]; y++) {
                        if (!(t[y].ratio < 1.0)) continue;
                        cm = cm + (t[y].hi - t[y].lo) * (1.0 - t[y].ratio) / t[y].ratio;
                        last = y;
                        if (x < cm) { d = y; break; }
                    }
                    if (d < 0) d = last;
                    deal[d]++;
                }
                for (int y = 0; y < ntab[cl]; y++)
                    if (deal[y] > 0) {
                        ent[n_ent].slot = i; ent[n_ent].dest = t[y].dest; ent[n_ent].copies = deal[y]; ent[n_ent].stays = 1;
                        n_ent++;
                    }
            }
            /* an outbox holds cap entries (law 1; law 0 has one per mover and cannot run out): the first
             * particle whose entries do not fit closes it -- they and every later particle's are
             * dropped, a mover among them stays */
            if (g_trans_law == 1 && (closed || nbox[c] + n_ent > s->cap)) {
                closed = 1;
                status |= PMCO_ST_OUTBOX_OVERFLOW;
                continue;
            }
            for (int e = 0; e < n_ent; e++) {
                if ((size_t)nbox[c] == box_cap) { box_cap *= 2; box[c] = (entry *)realloc(box[c], sizeof(entry) * box_cap); }
                box[c][nbox[c]++] = ent[e];
            }
            if (moves) n_movers++;
        }
        n_moved[c] = n_movers;
    }
    /* phase 2: arrivals appended in the fixed source order; sources still hold their data */
    int *n_old = (int *)malloc(sizeof(int) * (size_t)n_cells);
    for (int c = 0; c < n_cells; c++) n_old[c] = *cells[c].n_part;
    for (int c = 0; c < n_cells; c++) {
        pmco_cell *dst = &cells[c];
        const int i = c % nx, k = (c / nx) % nz, j = c / (nx * nz);
        if (g_trans_lateral && (i == 0 || i == nx - 1 || j == 0 || j == ny - 1)) {
            n_arrived[c] = 0; /* what is sent to a lateral boundary cell leaves the domain (:775-801) */
            continue;
        }
        int src[4 + 128], ns = 0;
        const int di[4] = {1, -1, 0, 0}, dj[4] = {0, 0, 1, -1}; /* east, west, north, south neighbour */
        for (int d = 0; d < 4; d++) {
            int ii = i + di[d], jj = j + dj[d];
            if (ii < 0 || ii >= nx) { if (!periodic_x) continue; ii = (ii + nx) % nx; }
            if (jj < 0 || jj >= ny) { if (!periodic_y) continue; jj = (jj + ny) % ny; }
            src[ns++] = (jj * nz + k) * nx + ii;
        }
        for (int kk = 0; kk < nz; kk++) if (kk != k) src[ns++] = (j * nz + kk) * nx + i;
        int arrived = 0;
        for (int q = 0; q < ns; q++) {
            const int sc = src[q];
            /* a neighbour can appear twice under periodic wrap with nx or ny <= 2: take it once */
            int dup = 0;
            for (int q2 = 0; q2 < q; q2++) dup |= (src[q2] == sc);
            if (dup || sc == c) continue;
            const pmco_cell *s = &cells[sc];
            for (int e = 0; e < nbox[sc]; e++) {
                if (box[sc][e].dest != c) continue;
                for (int g = 0; g < box[sc][e].copies; g++) {
                    const int at = *dst->n_part;
                    if (at >= dst->cap) { /* no room: flagged, and the copy's new id is spent all the same */
                        status |= PMCO_ST_CAP_OVERFLOW;
                        if (g != box[sc][e].copies - 1 || box[sc][e].stays) next_id[c]++;
                        continue;
                    }
                    const int from = box[sc][e].slot;
                    for (int sp = 0; sp < dst->n_spec; sp++)
                        dst->vol[(size_t)sp * dst->cap + at] = s->vol[(size_t)sp * s->cap + from];
                    for (int sp = 0; sp < dst->n_source; sp++)
                        dst->n_orig[(size_t)sp * dst->cap + at] = s->n_orig[(size_t)sp * s->cap + from];
                    dst->vtot[at] = s->vtot[from]; dst->mass[at] = s->mass[from];
                    dst->wclass[at] = s->wclass[from];
                    dst->lct[at] = s->lct[from]; dst->gct[at] = s->gct[from];
                    comp_copy(dst, at, s, from);
                    if (g == box[sc][e].copies - 1 && !box[sc][e].stays) dst->id[at] = s->id[from];
                    else { dst->id[at] = next_id[c]; next_id[c]++; }
                    *dst->n_part = at + 1;
                    arrived++;
                }
            }
        }
        n_arrived[c] = arrived;
    }
    /* phase 3: the movers leave (stable compaction keeps arrivals behind the survivors) */
    for (int c = 0; c < n_cells; c++) {
        const int n = *cells[c].n_part;
        unsigned char *dead = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
        int n_dead = 0;
        for (int e = 0; e < nbox[c]; e++)
            if (!box[c][e].stays) { dead[box[c][e].slot] = 1; n_dead++; }
        compact_stable(&cells[c], dead, n, n_dead);
        free(dead);
        free(box[c]);
    }
    (void)n_old;
    free(n_old); free(box); free(nbox); free(tab); free(ntab);
    return status;
}


/* ------------------------------------------------------------------------
 * Transport probabilities (interface/wrf_pmc_trans.F90:95-156 and the routines it calls).
 * ------------------------------------------------------------------------ */
int pmco_transport_probs(const pmco_trans_fields *f, const int32_t *n_parts, const double *mag,
                         double *prob_h, double *prob_v, int32_t *n_substeps, int32_t *bad)
{
    const int nx = f->nx, nz = f->nz, ny = f->ny, C = f->n_class;
    if (nx < 1 || ny < 1 || nz < 2 || C < 1) return -1;
    const size_t nc = (size_t)nx * nz * ny;
    const double dt = f->dt, dx = f->dx, dy = f->dy;
    /* prob_advection (1,0,0) (-1,0,0) (0,0,1) (0,0,-1) (0,1,0) (0,-1,0) per cell and class */
    double *adv = (double *)calloc(nc * (size_t)C * 6, sizeof(double));
    double *dif = (double *)calloc(nc * 4, sizeof(double));      /* x+, x-, y+, y- (all classes) */
    double *B = (double *)malloc(sizeof(double) * (size_t)nz * nz * 4);
    double *Bt = B + (size_t)nz * nz, *R = Bt + (size_t)nz * nz, *A = R + (size_t)nz * nz;
    double *zm = (double *)malloc(sizeof(double) * (size_t)(5 * nz + 5));
    double *rm = zm + nz, *zw = rm + nz, *kw = zw + nz + 1, *rw = kw + nz + 1;
#define HALO(a, i, k, j) ((double)(a)[((size_t)((j) + 1) * nz + (k)) * (size_t)(nx + 2) + (size_t)((i) + 1)])
#define HALOF(a, i, k, j) ((a)[((size_t)((j) + 1) * nz + (k)) * (size_t)(nx + 2) + (size_t)((i) + 1)])
    for (int j = 0; j < ny; j++)
        for (int k = 0; k < nz; k++)
            for (int i = 0; i < nx; i++) {
                const size_t c = ((size_t)j * nz + k) * nx + i;
                const double rho_mass = 1.0 / HALO(f->alt, i, k, j);
                /* compute_advect_probs_wrf (:205-286) */
                for (int cl = 0; cl < C; cl++) {
                    double *a = adv + (c * C + cl) * 6;
                    const double mix_rat = (double)n_parts[c * C + cl] * mag[c * C + cl] / rho_mass;
                    const float ue = f->u_flux[(((size_t)cl * ny + j) * nz + k) * (size_t)(nx + 1) + i + 1];
                    const float uw = f->u_flux[(((size_t)cl * ny + j) * nz + k) * (size_t)(nx + 1) + i];
                    const float vn = f->v_flux[(((size_t)cl * (ny + 1) + j + 1) * nz + k) * (size_t)nx + i];
                    const float vs = f->v_flux[(((size_t)cl * (ny + 1) + j) * nz + k) * (size_t)nx + i];
                    const float wu = f->w_flux[(((size_t)cl * ny + j) * (nz + 1) + k + 1) * (size_t)nx + i];
                    const float wd = f->w_flux[(((size_t)cl * ny + j) * (nz + 1) + k) * (size_t)nx + i];
                    if (mix_rat > 0.0) {
                        if (ue > 0.0f) a[0] = (double)ue / mix_rat;
                        if (uw < 0.0f) a[1] = -1.0 * (double)uw / mix_rat;
                        if (vn > 0.0f) a[2] = (double)vn / mix_rat;
                        if (vs < 0.0f) a[3] = -1.0 * (double)vs / mix_rat;
                        if (wu > 0.0f) { const double q = (double)wu / mix_rat; a[4] = q < .95 ? q : .95; }
                        if (wd < 0.0f) { const double q = -1.0 * (double)wd / mix_rat; a[5] = q < .95 ? q : .95; }
                    }
                    /* sum over (-1:1,-1:1,-1:1) in array element order: y-, z-, x-, x+, z+, y+ */
                    const double ps = ((((a[3] + a[5]) + a[1]) + a[0]) + a[4]) + a[2];
                    if (ps > 1.0)
                        for (int d = 0; d < 6; d++) a[d] = a[d] / (ps + 1.0e-8);
                }
                /* compute_diffusion_probs (:358-409) */
                const int di[4] = {1, -1, 0, 0}, dj[4] = {0, 0, 1, -1};
                for (int d = 0; d < 4; d++) {
                    const float ksum = HALOF(f->xkhh, i, k, j) + HALOF(f->xkhh, i + di[d], k, j + dj[d]);
                    double diff_coef = .5 * (double)ksum;
                    if (!(diff_coef > 0.0)) diff_coef = 0.0;
                    const double rho_edge = .5 * (rho_mass + (1.0 / HALO(f->alt, i + di[d], k, j + dj[d])));
                    const double h = d < 2 ? dx : dy;
                    dif[c * 4 + d] = (dt * diff_coef * rho_edge) / (h * h * rho_mass);
                }
            }
    memset(prob_v, 0, sizeof(double) * nc * (size_t)C * nz);
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            if (n_substeps) n_substeps[(size_t)j * nx + i] = 0;
            if (!f->all_columns) {
                const int gi = i + f->i_offset + 1, gj = j + f->j_offset + 1;
                if (gi < 2 || gi > f->global_nx - 1 || gj < 2 || gj > f->global_ny - 1) continue;
            }
#define COL(a, k, nzz) ((a)[((size_t)j * (nzz) + (k)) * (size_t)nx + i])
            /* stable_timestep (:629-651) */
            const double min_dz = (double)(COL(f->z, 1, nz) - COL(f->z, 0, nz));  /* single-precision difference */
            float kmaxf = COL(f->exch_h, 0, nz + 1);
            for (int k = 1; k <= nz; k++) if (COL(f->exch_h, k, nz + 1) > kmaxf) kmaxf = COL(f->exch_h, k, nz + 1);
            const double max_k = (double)kmaxf > 1.0e-6 ? (double)kmaxf : 1.0e-6;
            const double max_time = .1 * (min_dz * min_dz) / max_k;
            const int num_steps = (int)ceil(dt / max_time);
            const double small_dt = dt / num_steps;
            if (n_substeps) n_substeps[(size_t)j * nx + i] = num_steps;
            /* compute_vertical_probs (:414-562) */
            for (int k = 0; k < nz; k++) {
                zm[k] = (double)COL(f->z, k, nz);
                rm[k] = 1.0 / HALO(f->alt, i, k, j);
            }
            for (int k = 0; k <= nz; k++) { zw[k] = (double)COL(f->z_at_w, k, nz + 1); kw[k] = 0.0; }
            for (int k = 0; k < nz - 1; k++) {
                const double e = (double)COL(f->exch_h, k + 1, nz + 1);
                kw[k] = e > 1.0e-6 ? e : 1.0e-6;
                const double tmp = zm[k + 1] - zm[k];
                rw[k] = ((zm[k + 1] - zw[k + 1]) / tmp) * rm[k] + ((zw[k + 1] - zm[k]) / tmp) * rm[k + 1];
            }
            memset(B, 0, sizeof(double) * (size_t)nz * nz);
#define M(a, r, q) ((a)[(size_t)(r) * nz + (q)])
            for (int k = 1; k < nz - 1; k++) {
                M(B, k, k - 1) = (small_dt / ((zw[k + 1] - zw[k]) * (zm[k] - zm[k - 1]))) * (rw[k - 1] / rm[k]) * kw[k - 1];
                M(B, k, k + 1) = (small_dt / ((zw[k + 1] - zw[k]) * (zm[k + 1] - zm[k]))) * (rw[k] / rm[k]) * kw[k];
                M(B, k, k) = 1.0 - M(B, k, k - 1) - M(B, k, k + 1);
            }
            M(B, 0, 1) = (small_dt / ((zw[1] - zw[0]) * (zm[1] - zm[0]))) * (rw[0] / rm[0]) * kw[0];
            M(B, 0, 0) = 1.0 - M(B, 0, 1);
            M(B, nz - 1, nz - 2) = (small_dt / ((zw[nz] - zw[nz - 1]) * (zm[nz - 1] - zm[nz - 2])))
                                   * (rw[nz - 2] / rm[nz - 1]) * kw[nz - 2];
            M(B, nz - 1, nz - 1) = 1.0 - M(B, nz - 1, nz - 2);
            memcpy(Bt, B, sizeof(double) * (size_t)nz * nz);
            for (int t = 2; t <= num_steps; t++) {
                for (int r = 0; r < nz; r++)
                    for (int q = 0; q < nz; q++) {
                        double acc = 0.0;
                        for (int m = 0; m < nz; m++) acc = acc + M(Bt, r, m) * M(B, m, q);
                        M(R, r, q) = acc;
                    }
                for (int r = 0; r < nz; r++) {
                    double rs = 0.0;
                    for (int q = 0; q < nz; q++) rs = rs + M(R, r, q);
                    for (int q = 0; q < nz; q++) M(Bt, r, q) = M(R, r, q) / rs;
                }
            }
            for (int cl = 0; cl < C; cl++) {
#define ADV(k, d) adv[((((size_t)j * nz + (k)) * nx + i) * C + cl) * 6 + (d)]
                memset(A, 0, sizeof(double) * (size_t)nz * nz);
                for (int k = 1; k < nz - 1; k++) {
                    M(A, k, k) = 1.0 - (ADV(k, 4) + ADV(k, 5));
                    M(A, k, k - 1) = ADV(k, 5);
                    M(A, k, k + 1) = ADV(k, 4);
                }
                M(A, 0, 0) = 1.0 - ADV(0, 4);
                M(A, 0, 1) = ADV(0, 4);
                M(A, nz - 1, nz - 1) = 1.0 - ADV(nz - 1, 5);
                M(A, nz - 1, nz - 2) = ADV(nz - 1, 5);
                for (int r = 0; r < nz; r++)
                    for (int q = 0; q < nz; q++) {
                        double acc = 0.0;
                        if (f->do_vertmix > 0)
                            for (int m = 0; m < nz; m++) acc = acc + M(Bt, r, m) * M(A, m, q);
                        else
                            acc = M(A, r, q);
                        prob_v[((((size_t)j * nz + r) * nx + i) * C + cl) * nz + q] = acc;
                    }
            }
        }
    /* normalize_probs (:692-773) */
    for (size_t c = 0; c < nc; c++) {
        const int k_own = (int)((c / (size_t)nx) % (size_t)nz);
        int flag = 0;
        for (int cl = 0; cl < C; cl++) {
            double *a = adv + (c * C + cl) * 6, *pv = prob_v + (c * C + cl) * nz;
            double dl[4];
            for (int d = 0; d < 4; d++) dl[d] = dif[c * 4 + d];
            int neg = 0;
            for (int d = 0; d < 6; d++) neg |= !(a[d] >= 0.0);
            for (int d = 0; d < 4; d++) neg |= !(dl[d] >= 0.0);
            for (int k = 0; k < nz; k++) neg |= !(pv[k] >= 0.0);
            if (neg) flag |= 1;
            /* sums in array element order: y-, x-, x+, y+ */
            double ps = ((a[3] + a[1]) + a[0]) + a[2];
            ps = ps + (((dl[3] + dl[1]) + dl[0]) + dl[2]);
            for (int k = 0; k < nz; k++) if (k != k_own) ps = ps + pv[k];
            if (ps >= 1.0) {
                for (int k = 0; k < nz; k++) pv[k] = pv[k] / ps;
                for (int d = 0; d < 4; d++) { dl[d] = dl[d] / ps; a[d] = a[d] / ps; }
            }
            neg = 0;
            int big = 0;
            for (int d = 0; d < 6; d++) { neg |= !(a[d] >= 0.0); big |= !(a[d] <= 1.0); }
            for (int d = 0; d < 4; d++) { neg |= !(dl[d] >= 0.0); big |= !(dl[d] <= 1.0); }
            for (int k = 0; k < nz; k++) { neg |= !(pv[k] >= 0.0); big |= !(pv[k] <= 1.0); }
            if (neg) flag |= 2;
            if (big) flag |= 4;
            ps = ((a[3] + a[1]) + a[0]) + a[2];
            ps = ps + (((dl[3] + dl[1]) + dl[0]) + dl[2]);
            for (int k = 0; k < nz; k++) if (k != k_own) ps = ps + pv[k];
            if (!(ps < 1.0 + 1.0e-8)) flag |= 8;
            double *ph = prob_h + (c * C + cl) * 4;    /* i-1, i+1, j-1, j+1 */
            ph[0] = a[1] + dl[1];
            ph[1] = a[0] + dl[0];
            ph[2] = a[3] + dl[3];
            ph[3] = a[2] + dl[2];
        }
        if (bad) bad[c] = flag;
    }
#undef HALO
#undef HALOF
#undef COL
#undef M
#undef ADV
    free(adv); free(dif); free(B); free(zm);
    return 0;
}

/* ------------------------------------------------------------------------
 * trans_aero_preweight + compute_fluxes (interface/wrf_pmc_trans_aero.F90:1103-1423, 1428-1627)
 * ------------------------------------------------------------------------ */
int pmco_transport_preweight(pmco_cell *cells, int n_cells, int nx, int nz, int ny, int periodic_x,
                             int periodic_y, const double *prob_h, const double *prob_v,
                             const double *cell_volume, const double *n_ideal, uint64_t seed,
                             uint32_t cell_id_base, uint32_t step, double *expected,
                             int32_t *n_removed)
{
    const int C = cells[0].n_class;
    const int ideal = periodic_x && periodic_y;
    double *old = (double *)calloc((size_t)n_cells * C, sizeof(double));
    /* ghost cells beyond the four lateral edges (ideal case): [side][k][along][class] */
    double *gy = (double *)calloc((size_t)2 * nz * nx * C, sizeof(double));
    double *gx = (double *)calloc((size_t)2 * nz * ny * C, sizeof(double));
    for (int c = 0; c < n_cells; c++)
        for (int cl = 0; cl < C; cl++) {
            int n = 0;
            for (int i = 0; i < *cells[c].n_part; i++) n += (cells[c].wclass[i] == cl);
            old[(size_t)c * C + cl] = (double)n * cells[c].mag[cl];        /* :1211-1217 */
            expected[(size_t)c * C + cl] = old[(size_t)c * C + cl];
        }
#define CELL(i, k, j) (((size_t)(j) * nz + (k)) * nx + (i))
    for (int j = 0; j < ny; j++)
        for (int k = 0; k < nz; k++)
            for (int i = 0; i < nx; i++)
                for (int cl = 0; cl < C; cl++) {
                    const size_t c = CELL(i, k, j);
                    const double o = old[c * C + cl], sv = cell_volume[c];
                    double *self = &expected[c * C + cl];
                    const double *ph = prob_h + (c * C + cl) * 4;          /* i-1, i+1, j-1, j+1 */
                    int dirs[4] = {0, 0, 0, 0}, vert = 0;                  /* which moves the cell makes */
                    /* (a periodic direction one cell wide is closed: the cell would be its own neighbour) */
                    if (ideal) { dirs[0] = dirs[1] = nx > 1; dirs[2] = dirs[3] = ny > 1; vert = 1; }
                    else if (i == 0) { if (j != 0 && j != ny - 1) dirs[1] = 1; }        /* :1510-1524 */
                    else if (i == nx - 1) { if (j != 0 && j != ny - 1) dirs[0] = 1; }   /* :1525-1539 */
                    else if (j == 0) { dirs[3] = 1; }                                   /* :1540-1554 */
                    else if (j == ny - 1) { dirs[2] = 1; }                              /* :1555-1569 */
                    else { dirs[0] = dirs[1] = dirs[2] = dirs[3] = 1; vert = 1; }
                    /* the reference's order: j-1, j+1, i-1, i+1, then the levels */
                    const int order[4] = {2, 3, 0, 1};
                    for (int q = 0; q < 4; q++) {
                        const int d = order[q];
                        if (!dirs[d]) continue;
                        const int ii = i + (d == 0 ? -1 : d == 1 ? 1 : 0), jj = j + (d == 2 ? -1 : d == 3 ? 1 : 0);
                        const double flux = ph[d] * o;
                        double *dst;
                        double dv;
                        if (jj < 0) { dst = &gy[(((size_t)0 * nz + k) * nx + i) * C + cl]; dv = cell_volume[CELL(i, k, ny - 1)]; }
                        else if (jj >= ny) { dst = &gy[(((size_t)1 * nz + k) * nx + i) * C + cl]; dv = cell_volume[CELL(i, k, 0)]; }
                        else if (ii < 0) { dst = &gx[(((size_t)0 * nz + k) * ny + j) * C + cl]; dv = cell_volume[CELL(nx - 1, k, j)]; }
                        else if (ii >= nx) { dst = &gx[(((size_t)1 * nz + k) * ny + j) * C + cl]; dv = cell_volume[CELL(0, k, j)]; }
                        else { dst = &expected[CELL(ii, k, jj) * C + cl]; dv = cell_volume[CELL(ii, k, jj)]; }
                        *dst = *dst + flux * (dv / sv);
                        *self = *self - flux;
                    }
                    if (vert)
                        for (int kk = 0; kk < nz; kk++) {
                            if (kk == k) continue;
                            const double flux = prob_v[(c * C + cl) * nz + kk] * o;
                            double *dst = &expected[CELL(i, kk, j) * C + cl];
                            *dst = *dst + flux * (cell_volume[CELL(i, kk, j)] / sv);
                            *self = *self - flux;
                        }
                }
    if (ideal) {   /* what the ghost cells gathered arrives at the cells they stand for */
        for (int k = 0; k < nz; k++) for (int i = 0; i < nx; i++) for (int cl = 0; cl < C; cl++)
            expected[CELL(i, k, ny - 1) * C + cl] += gy[(((size_t)0 * nz + k) * nx + i) * C + cl];
        for (int k = 0; k < nz; k++) for (int i = 0; i < nx; i++) for (int cl = 0; cl < C; cl++)
            expected[CELL(i, k, 0) * C + cl] += gy[(((size_t)1 * nz + k) * nx + i) * C + cl];
        for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int cl = 0; cl < C; cl++)
            expected[CELL(nx - 1, k, j) * C + cl] += gx[(((size_t)0 * nz + k) * ny + j) * C + cl];
        for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int cl = 0; cl < C; cl++)
            expected[CELL(0, k, j) * C + cl] += gx[(((size_t)1 * nz + k) * ny + j) * C + cl];
    }
#undef CELL
    /* the comparison with n_part_ideal and aero_state_scale_weight (:1386-1398) */
    for (int c = 0; c < n_cells; c++) {
        pmco_cell *s = &cells[c];
        int removed = 0;
        for (int cl = 0; cl < C; cl++) {
            const double n_parts_new = expected[(size_t)c * C + cl] / s->mag[cl];
            const double ideal_n = n_ideal[(size_t)c * C + cl];
            if (!(n_parts_new > 2.0 * ideal_n)) continue;
            const double ratio = n_parts_new / ideal_n;
            /* aero_state_scale_weight(weight_ratio > 1, .true., .true.) (:1390-1393) */
            int st_ = 0;
            removed += scale_weight(s, cl, ratio, 1, 1, seed, cell_id_base + (uint32_t)c, step, (uint32_t)cl,
                                    NULL, &st_);
        }
        if (n_removed) n_removed[c] = removed;
    }
    free(old); free(gy); free(gx);
    return 0;
}

/* pmco_transport with the reference's open lateral boundaries (real case, neither direction
 * periodic; interface/wrf_pmc_trans_aero.F90:497-534, 689-801): cells 1 .. n-2 sample as always,
 * a lateral boundary cell only sends inwards with min(prob, 1), a corner nothing, and what is
 * sent to a boundary cell is dropped. */
int pmco_transport_lateral(pmco_cell *cells, int n_cells, int nx, int nz, int ny,
                           const double *prob_h, const double *prob_v, const double *cell_volume,
                           uint64_t seed, uint32_t cell_id_base, uint32_t step, int64_t *next_id,
                           int32_t *n_moved, int32_t *n_arrived)
{
    if (nx < 3 || ny < 3) return -1;
    g_trans_lateral = 1;
    const int st = pmco_transport(cells, n_cells, nx, nz, ny, 0, 0, prob_h, prob_v, cell_volume, seed,
                                  cell_id_base, step, next_id, n_moved, n_arrived);
    g_trans_lateral = 0;
    return st;
}

/* ------------------------------------------------------------------------
 * The refill of the lateral boundary cells after the transport
 * (interface/wrf_pmc_trans_aero.F90:803-955).
 * ------------------------------------------------------------------------ */

/* aero_state_dup_all_particles(from, to) [upstream, uncited], as the reference uses it on an
 * outflow boundary (:829-831): the boundary cell becomes a copy of its inner neighbour -- the
 * same particles in the same order under new ids, the same weights. */
void pmco_boundary_copy(pmco_cell *dst, const pmco_cell *src, int64_t *next_id_dst)
{
    const int n = *src->n_part;
    for (int i = 0; i < n; i++) {
        for (int sp = 0; sp < dst->n_spec; sp++)
            dst->vol[(size_t)sp * dst->cap + i] = src->vol[(size_t)sp * src->cap + i];
        for (int sp = 0; sp < dst->n_source; sp++)
            dst->n_orig[(size_t)sp * dst->cap + i] = src->n_orig[(size_t)sp * src->cap + i];
        dst->vtot[i] = src->vtot[i]; dst->mass[i] = src->mass[i];
        dst->wclass[i] = src->wclass[i];
        dst->lct[i] = src->lct[i]; dst->gct[i] = src->gct[i];
        comp_copy(dst, i, src, i);
        dst->id[i] = *next_id_dst + i;
    }
    *next_id_dst += n;
    *dst->n_part = n;
    for (int cl = 0; cl < dst->n_class; cl++) dst->mag[cl] = src->mag[cl];
    *dst->sort_valid = 0;
}

/* aero_state_resample (interface/wrf_pmc_trans_aero.F90:976-1098) on an inflow boundary
 * (:832-845): bg->num_conc are the background's per-mode number mixing ratios, density the air's
 * (1 / inverse_density), sample_prop = 1 and the create time 0 at the reference's call sites
 * (create_time here).
 *   throw_away: the cell is emptied and sampled anew, Poisson(num_conc density / magnitude) per
 *               mode (:1028-1032; aero_state_add_aero_dist_sample [upstream, uncited], as the
 *               emission of the scenario update)
 *   otherwise, per class: n_samp = Poisson(total_num_conc density / magnitude) is the count the
 *               class should have (:1037-1049).  Short of it by n_parts_to_add: per mode of the
 *               class Poisson(share n_parts_to_add) particles are sampled and added if the
 *               magnitude exceeds 0.1 density num_conc / (share n_parts_to_add) (:1050-1084; with
 *               flat weights: if the shortfall is more than a tenth of the target).  Over it:
 *               min(excess, particles) particles chosen at random leave (:1085-1095; here
 *               selection sampling in rank order: particle t of n goes when
 *               u (n - t) < still to go -- a uniformly random subset of exactly that size).
 * [aero_mode_sample_radius takes a size_factor in this fork (:1073-1075) whose meaning is in
 * the absent submodule: the radius is drawn from the plain log-normal mode.] */
int pmco_boundary_resample(pmco_cell *c, int throw_away, const pmco_aero_dist *bg, double density,
                           double create_time, uint64_t seed, uint32_t cell_id, uint32_t step,
                           int64_t *next_id, int32_t *n_added, int32_t *n_removed)
{
    int status = 0;
    *n_added = *n_removed = 0;
    *c->sort_valid = 0;
    if (throw_away) {
        *n_removed = *c->n_part;
        *c->n_part = 0;                                             /* aero_state_zero */
        *n_added = add_dist_sample(c, bg, density, 1.0, 3, create_time, seed, cell_id, step, next_id, &status);
        return status;
    }
    double means[64];
    for (int cl = 0; cl < c->n_class; cl++) {
        int n_before = 0;
        for (int i = 0; i < *c->n_part; i++) n_before += (c->wclass[i] == cl);
        double total = 0.0;
        for (int m = 0; m < bg->n_mode; m++)
            if (bg->mode_class[m] == cl && bg->num_conc[m] >= 0.0) /* (negative: no such mode in this cell) */
                total = total + bg->num_conc[m] * density;
        const double n_parts_new = total / c->mag[cl];
        uint32_t w[4];
        pmco_rng_block(seed, cell_id, step, (uint32_t)cl, PMCO_SITE_RESAMPLE, 0, w);
        const int n_samp = pmco_poisson(n_parts_new, pmco_u53(w[0], w[1]));
        const int n_to_add = n_samp - n_before;
        if (n_to_add > 0) {
            for (int m = 0; m < bg->n_mode; m++) {
                means[m] = -1.0;
                if (bg->mode_class[m] != cl || bg->num_conc[m] < 0.0) continue;
                const double share = (density * bg->num_conc[m]) / total;
                const double n_samp_avg = share * (double)n_to_add;
                const double threshold = (density * bg->num_conc[m] / n_samp_avg) * .1;
                if (c->mag[cl] > threshold) means[m] = n_samp_avg;
            }
            *n_added += add_dist_sample_means(c, bg, density, 1.0, means, 3, create_time, seed, cell_id, step,
                                              next_id, &status);
        } else {
            const int n = *c->n_part;
            int m_rm = -n_to_add < n_before ? -n_to_add : n_before;
            unsigned char *dead = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
            int t = 0, k = 0;
            for (int i = 0; i < n; i++) {
                if (c->wclass[i] != cl) continue;
                pmco_rng_block(seed, cell_id, step, (uint32_t)i, PMCO_SITE_RESAMPLE, 1u + (uint32_t)cl, w);
                if (pmco_u53(w[0], w[1]) * (double)(n_before - t) < (double)(m_rm - k)) { dead[i] = 1; k++; }
                t++;
            }
            compact_stable(c, dead, n, k);
            free(dead);
            *n_removed += k;
        }
    }
    return status;
}
