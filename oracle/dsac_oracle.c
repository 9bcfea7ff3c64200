/*
 * oracle/dsac_oracle.c — CPU restatement of CrossLoc's `dsacstar.forward_rgb`.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (crossloc_amd/, dsacstar.py)
 * may include, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED vs the reference binary: the reference extension delegates all of its
 * geometry to OpenCV 3.4.2 (cv::solvePnP P3P / ITERATIVE, cv::projectPoints,
 * cv::Rodrigues), which is neither vendored in /root/reference nor installed in the
 * build image, and the reference ships no tests or golden vectors for this path.  The
 * restatement is therefore pinned by analytic known answers (tests/test_oracle_dsac.py):
 * exact-coordinate scenes recover the ground-truth pose, P3P recovers a known pose,
 * closed-form soft-inlier scores, refinement monotonicity, hypothesis-order invariance.
 *
 * What is restated (reference file:line, relative to /root/reference/):
 *   orchestration, constants      dsacstar/dsacstar.cpp:47-48, 63-178
 *   createSampling                dsacstar/dsacstar_util.h:59-76
 *   sampleHypotheses/safeSolvePnP dsacstar/dsacstar_util.h:91-120, 135-221
 *   getReproErrs (calcJ=false)    dsacstar/dsacstar_util.h:356-446
 *   getHypScores                  dsacstar/dsacstar_util.h:316-343
 *   softMax / draw(argmax)        dsacstar/dsacstar_util.h:684-752
 *   refineHyp                     dsacstar/dsacstar_util.h:522-597
 *   pose2trans + write-back       dsacstar/dsacstar_util.h:759-770, dsacstar.cpp:172-177
 *   ThreadRand semantics          dsacstar/thread_rand.cpp:13-54, 68-71
 *
 * Third-party arithmetic (OpenCV 3.4.2, pinned by setup/environment.yml:11) restated
 * from its published algorithms, see DESIGN.md §"Solver arithmetic":
 *   - projectPoints: Xc = R*X + t in double, z = Z ? 1/Z : 1 (no cheirality test),
 *     u = x*f + cx, stored as float.
 *   - P3P + 4th point: law-of-cosines quartic (Grunert form of the Gao et al. system
 *     OpenCV solves), Ferrari resolution with a Newton root of the resolvent cubic,
 *     two Newton polish steps on the three distances, rigid alignment through the two
 *     triangle frames; the candidate with the smallest squared pixel error on the 4th
 *     point wins (OpenCV p3p::solve four-point overload behaviour).
 *   - ITERATIVE PnP with extrinsic guess: CvLevMarq state machine (max 20 outer
 *     iterations, eps FLT_EPSILON, lambda = 10^k, k0 = -3, k+1 on error increase up to
 *     16, k-1 on accept, damping JtJ.diag *= 1+lambda), 6x6 solve by Cholesky, on a
 *     left-multiplicative rotation increment instead of OpenCV's global Rodrigues vector.
 *
 * Deliberate, documented deviations that make "bit-exact hypothesis indices" definable
 * (SURVEY.md §7 hard parts): the RNG is counter-based (seed, image, hypothesis, try),
 * not std::mt19937 per OpenMP thread; every transcendental is a fixed polynomial built
 * from + - * / sqrt so gcc and hipcc produce identical bits (compile with
 * -ffp-contract=off); reductions over cells use a fixed order (lane-strided partials +
 * xor-butterfly) instead of the reference's x-major serial order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define XO_MAX_REF_STEPS 100          /* dsacstar.cpp:47 */
#define XO_LM_MAX_ITER 20             /* cv::solvePnP ITERATIVE term criteria */
#define XO_FLT_EPSILON 1.1920928955078125e-07

/* ------------------------------------------------------------------ deterministic math */

static double xo_pow2i(int k)
{
    /* 2^k for k in [-1022, 1023] built from the exponent field (exact) */
    uint64_t bits = (uint64_t)(k + 1023) << 52;
    double d;
    memcpy(&d, &bits, 8);
    return d;
}

/* exp(x): k = floor(x/ln2 + 1/2), r = x - k ln2 (two-constant Cody-Waite), Taylor degree 14 */
static double xo_exp(double x)
{
    if (x != x) return x;
    if (x > 709.0) return INFINITY;
    if (x < -708.0) return 0.0;
    const double INV_LN2 = 0x1.71547652b82fep+0;
    const double LN2_HI = 0x1.62e42f8000000p-1;
    const double LN2_LO = 0x1.be8e7bcd5e4f2p-27;
    double kf = floor(x * INV_LN2 + 0.5);
    double r = (x - kf * LN2_HI) - kf * LN2_LO;
    double p = 1.0 / 87178291200.0;              /* 1/14! */
    p = p * r + 1.0 / 6227020800.0;              /* 1/13! */
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
    return p * xo_pow2i((int)kf);
}

/* sin and cos: k = floor(x*2/pi + 1/2), r = x - k*pi/2, Taylor degree 17 / 18 on [-pi/4, pi/4] */
static void xo_sincos(double x, double *s, double *c)
{
    const double TWO_OVER_PI = 0x1.45f306dc9c883p-1;
    const double PIO2_HI = 0x1.921fb50000000p+0;
    const double PIO2_LO = 0x1.110b4611a6263p-26;
    double kf = floor(x * TWO_OVER_PI + 0.5);
    double r = (x - kf * PIO2_HI) - kf * PIO2_LO;
    double r2 = r * r;
    double ps = -1.0 / 355687428096000.0;        /* -1/17! */
    ps = ps * r2 + 1.0 / 1307674368000.0;        /* 1/15! */
    ps = ps * r2 - 1.0 / 6227020800.0;
    ps = ps * r2 + 1.0 / 39916800.0;
    ps = ps * r2 - 1.0 / 362880.0;
    ps = ps * r2 + 1.0 / 5040.0;
    ps = ps * r2 - 1.0 / 120.0;
    ps = ps * r2 + 1.0 / 6.0;
    double sr = r - r * r2 * ps;
    double pc = -1.0 / 6402373705728000.0;       /* -1/18! */
    pc = pc * r2 + 1.0 / 20922789888000.0;       /* 1/16! */
    pc = pc * r2 - 1.0 / 87178291200.0;
    pc = pc * r2 + 1.0 / 479001600.0;
    pc = pc * r2 - 1.0 / 3628800.0;
    pc = pc * r2 + 1.0 / 40320.0;
    pc = pc * r2 - 1.0 / 720.0;
    pc = pc * r2 + 1.0 / 24.0;
    double cr = 1.0 - r2 * (0.5 - r2 * pc);
    /* quadrant: k mod 4 (kf may be negative) */
    double q = kf - 4.0 * floor(kf * 0.25);
    int qi = (int)q;
    if (qi == 0) { *s = sr; *c = cr; }
    else if (qi == 1) { *s = cr; *c = -sr; }
    else if (qi == 2) { *s = -sr; *c = -cr; }
    else { *s = -cr; *c = sr; }
}

/* ------------------------------------------------------------------ counter-based RNG */

static uint64_t xo_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

/* state for (seed, image, hypothesis, try); draw j uses mix64(state + (j+1)*golden) */
static uint64_t xo_try_state(uint64_t seed, uint64_t image, uint32_t hyp, uint32_t t)
{
    uint64_t s = xo_mix64(seed + 0x9e3779b97f4a7c15ULL * (image + 1));
    s = xo_mix64(s ^ (((uint64_t)hyp << 32) | (uint64_t)t));
    return s;
}

/* irand(0, n) of thread_rand.cpp:68-71 semantics (uniform in [0, n)), multiply-shift */
static int xo_draw(uint64_t state, int j, int n)
{
    uint64_t r = xo_mix64(state + 0x9e3779b97f4a7c15ULL * (uint64_t)(j + 1));
    uint32_t hi = (uint32_t)(r >> 32);
    return (int)(((uint64_t)hi * (uint64_t)(uint32_t)n) >> 32);
}

/* ------------------------------------------------------------------ small helpers */

typedef struct { double R[9]; double t[3]; } xo_pose;   /* world -> camera */

static void xo_pose_identity(xo_pose *p)
{
    for (int i = 0; i < 9; ++i) p->R[i] = 0.0;
    p->R[0] = p->R[4] = p->R[8] = 1.0;
    p->t[0] = p->t[1] = p->t[2] = 0.0;
}

typedef struct {
    const float *base; int64_t sc, sy, sx;   /* strides in elements */
    int Ho, Wo;
} xo_coords;

static void xo_fetch(const xo_coords *c, int x, int y, double X[3])
{
    const float *p = c->base + (int64_t)y * c->sy + (int64_t)x * c->sx;
    X[0] = (double)p[0];
    X[1] = (double)p[c->sc];
    X[2] = (double)p[2 * c->sc];
}

/* cv::projectPoints restated: returns float pixel (dsacstar_util.h:199-205, 395-401) */
static void xo_project(const xo_pose *p, const double X[3], double f, double cx, double cy,
                       float *u, float *v)
{
    double xc = p->R[0] * X[0] + p->R[1] * X[1] + p->R[2] * X[2] + p->t[0];
    double yc = p->R[3] * X[0] + p->R[4] * X[1] + p->R[5] * X[2] + p->t[1];
    double zc = p->R[6] * X[0] + p->R[7] * X[1] + p->R[8] * X[2] + p->t[2];
    double z = (zc != 0.0) ? 1.0 / zc : 1.0;
    double x = xc * z, y = yc * z;
    *u = (float)(x * f + cx);
    *v = (float)(y * f + cy);
}

/* reprojection error of one cell, clamped (dsacstar_util.h:438-443) */
static float xo_cell_err(const xo_pose *p, const double X[3], float px, float py,
                         double f, double cx, double cy, float maxReproj)
{
    float u, v;
    xo_project(p, X, f, cx, cy, &u, &v);
    float dx = px - u, dy = py - v;
    double n = sqrt((double)dx * (double)dx + (double)dy * (double)dy);
    float a = (float)n;
    return (maxReproj < a) ? maxReproj : a;       /* std::min(a, maxReproj) */
}

/* ------------------------------------------------------------------ quartic (Ferrari) */

/* a positive real root of g(z) = z^3 + c2 z^2 + c1 z + c0 with g(0) <= 0: Newton safeguarded by
 * the bracket [lo, hi] (g(lo) <= 0 < g(hi), hi starts at the Cauchy bound); bisect when Newton leaves it */
static double xo_cubic_pos_root(double c2, double c1, double c0)
{
    double m = fabs(c2);
    if (fabs(c1) > m) m = fabs(c1);
    if (fabs(c0) > m) m = fabs(c0);
    double lo = 0.0, hi = 1.0 + m;
    double z = hi;
    for (int it = 0; it < 128; ++it) {
        double g = ((z + c2) * z + c1) * z + c0;
        double dg = (3.0 * z + 2.0 * c2) * z + c1;
        if (g > 0.0) hi = z; else lo = z;
        if (g == 0.0) break;
        double zn = z - g / dg;
        if (!(zn > lo && zn < hi)) zn = 0.5 * (lo + hi);
        if (zn == z || !(hi > lo)) break;
        z = zn;
    }
    return z;
}

/* real roots of y^2 + b y + c */
static int xo_quadratic(double b, double c, double r[2])
{
    double disc = b * b - 4.0 * c;
    if (!(disc >= 0.0)) return 0;
    double sq = sqrt(disc);
    /* numerically stable pair */
    double q = (b >= 0.0) ? -0.5 * (b + sq) : -0.5 * (b - sq);
    if (q != 0.0) { r[0] = q; r[1] = c / q; }
    else { r[0] = 0.0; r[1] = 0.0; }
    return 2;
}

/* real roots of A4 x^4 + A3 x^3 + A2 x^2 + A1 x + A0, A4 != 0 */
static int xo_quartic(double A4, double A3, double A2, double A1, double A0, double roots[4])
{
    double a = A3 / A4, b = A2 / A4, c = A1 / A4, d = A0 / A4;
    double a2 = a * a;
    double p = b - 0.375 * a2;
    double q = c - 0.5 * a * b + 0.125 * a2 * a;
    double r = d - 0.25 * a * c + 0.0625 * a2 * b - (3.0 / 256.0) * a2 * a2;
    double shift = -0.25 * a;
    int n = 0;
    double z0 = xo_cubic_pos_root(2.0 * p, p * p - 4.0 * r, -(q * q));
    if (z0 > 0.0) {
        double s = sqrt(z0);
        double h = 0.5 * (p + z0);
        double g = 0.5 * q / s;
        double r2[2];
        int k = xo_quadratic(s, h - g, r2);
        for (int i = 0; i < k; ++i) roots[n++] = r2[i] + shift;
        k = xo_quadratic(-s, h + g, r2);
        for (int i = 0; i < k; ++i) roots[n++] = r2[i] + shift;
    } else {
        /* biquadratic: y^4 + p y^2 + r */
        double w[2];
        int k = xo_quadratic(p, r, w);
        for (int i = 0; i < k; ++i) {
            if (w[i] >= 0.0) {
                double y = sqrt(w[i]);
                roots[n++] = y + shift;
                roots[n++] = -y + shift;
            }
        }
        if (n > 4) n = 4;
    }
    return n;
}

/* ------------------------------------------------------------------ P3P + 4th point */

static void xo_cross(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* orthonormal frame of triangle (A,B,C): e1 along AB, e3 normal, e2 = e3 x e1; 0 if degenerate */
static int xo_frame(const double A[3], const double B[3], const double C[3], double E[9])
{
    double ab[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] };
    double ac[3] = { C[0] - A[0], C[1] - A[1], C[2] - A[2] };
    double n1 = sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]);
    if (!(n1 > 0.0)) return 0;
    double e1[3] = { ab[0] / n1, ab[1] / n1, ab[2] / n1 };
    double nn[3];
    xo_cross(ab, ac, nn);
    double n3 = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
    if (!(n3 > 0.0)) return 0;
    double e3[3] = { nn[0] / n3, nn[1] / n3, nn[2] / n3 };
    double e2[3];
    xo_cross(e3, e1, e2);
    for (int i = 0; i < 3; ++i) { E[i] = e1[i]; E[3 + i] = e2[i]; E[6 + i] = e3[i]; }  /* rows = axes */
    return 1;
}

/*
 * P[4][3] object points, uv[4][2] pixels.  Returns 1 and the world->camera pose on success.
 * cv::solvePnP(..., SOLVEPNP_P3P) call-site contract (dsacstar_util.h:185-193).
 */
static int xo_p3p(const double P[4][3], const double uv[4][2], double f, double cx, double cy,
                  xo_pose *out)
{
    double fb[3][3];
    for (int i = 0; i < 3; ++i) {
        double mx = (uv[i][0] - cx) / f, my = (uv[i][1] - cy) / f;
        double nrm = sqrt(mx * mx + my * my + 1.0);
        fb[i][0] = mx / nrm; fb[i][1] = my / nrm; fb[i][2] = 1.0 / nrm;
    }
    double ca = fb[1][0] * fb[2][0] + fb[1][1] * fb[2][1] + fb[1][2] * fb[2][2];
    double cb = fb[0][0] * fb[2][0] + fb[0][1] * fb[2][1] + fb[0][2] * fb[2][2];
    double cg = fb[0][0] * fb[1][0] + fb[0][1] * fb[1][1] + fb[0][2] * fb[1][2];
    double d0, d1, d2;
    d0 = P[1][0] - P[2][0]; d1 = P[1][1] - P[2][1]; d2 = P[1][2] - P[2][2];
    double a2 = d0 * d0 + d1 * d1 + d2 * d2;
    d0 = P[0][0] - P[2][0]; d1 = P[0][1] - P[2][1]; d2 = P[0][2] - P[2][2];
    double b2 = d0 * d0 + d1 * d1 + d2 * d2;
    d0 = P[0][0] - P[1][0]; d1 = P[0][1] - P[1][1]; d2 = P[0][2] - P[1][2];
    double c2 = d0 * d0 + d1 * d1 + d2 * d2;
    if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return 0;

    double E[9];
    if (!xo_frame(P[0], P[1], P[2], E)) return 0;

    double pq = (a2 - c2) / b2, qq = (a2 + c2) / b2;
    double c2b = c2 / b2, a2b = a2 / b2;
    double A4 = (pq - 1.0) * (pq - 1.0) - 4.0 * c2b * ca * ca;
    double A3 = 4.0 * (pq * (1.0 - pq) * cb - (1.0 - qq) * ca * cg + 2.0 * c2b * ca * ca * cb);
    double A2 = 2.0 * (pq * pq - 1.0 + 2.0 * pq * pq * cb * cb + 2.0 * ((b2 - c2) / b2) * ca * ca
                       - 4.0 * qq * ca * cb * cg + 2.0 * ((b2 - a2) / b2) * cg * cg);
    double A1 = 4.0 * (-pq * (1.0 + pq) * cb + 2.0 * a2b * cg * cg * cb - (1.0 - qq) * ca * cg);
    double A0 = (1.0 + pq) * (1.0 + pq) - 4.0 * a2b * cg * cg;
    if (!(A4 != 0.0) || A4 != A4) return 0;

    double roots[4];
    int nr = xo_quartic(A4, A3, A2, A1, A0, roots);

    int found = 0;
    double best = 0.0;
    for (int ri = 0; ri < nr; ++ri) {
        double v = roots[ri];
        if (!(v > 0.0)) continue;
        double den = cg - v * ca;
        if (!(den != 0.0)) continue;
        double u = ((pq - 1.0) * v * v - 2.0 * pq * cb * v + 1.0 + pq) / (2.0 * den);
        if (!(u > 0.0)) continue;
        double w = 1.0 + v * v - 2.0 * v * cb;
        if (!(w > 0.0)) continue;
        double s1 = sqrt(b2 / w), s2 = u * s1, s3 = v * s1;
        /* two Newton steps on the law-of-cosines system */
        for (int it = 0; it < 2; ++it) {
            double F1 = s2 * s2 + s3 * s3 - 2.0 * s2 * s3 * ca - a2;
            double F2 = s1 * s1 + s3 * s3 - 2.0 * s1 * s3 * cb - b2;
            double F3 = s1 * s1 + s2 * s2 - 2.0 * s1 * s2 * cg - c2;
            double j12 = 2.0 * s2 - 2.0 * s3 * ca, j13 = 2.0 * s3 - 2.0 * s2 * ca;
            double j21 = 2.0 * s1 - 2.0 * s3 * cb, j23 = 2.0 * s3 - 2.0 * s1 * cb;
            double j31 = 2.0 * s1 - 2.0 * s2 * cg, j32 = 2.0 * s2 - 2.0 * s1 * cg;
            /* J = [[0,j12,j13],[j21,0,j23],[j31,j32,0]] */
            double det = j12 * j23 * j31 + j13 * j21 * j32;
            if (!(det != 0.0)) break;
            double dx1 = (F1 * (-(j23 * j32)) - j12 * (-(j23 * F3)) + j13 * (F2 * j32)) / det;
            double dx2 = (-(F1 * (-(j23 * j31))) + j13 * (j21 * F3 - F2 * j31)) / det;
            double dx3 = (-(j12 * (j21 * F3 - F2 * j31)) + F1 * (j21 * j32)) / det;
            s1 -= dx1; s2 -= dx2; s3 -= dx3;
        }
        if (!(s1 > 0.0) || !(s2 > 0.0) || !(s3 > 0.0)) continue;
        double C0[3] = { s1 * fb[0][0], s1 * fb[0][1], s1 * fb[0][2] };
        double C1[3] = { s2 * fb[1][0], s2 * fb[1][1], s2 * fb[1][2] };
        double C2[3] = { s3 * fb[2][0], s3 * fb[2][1], s3 * fb[2][2] };
        double D[9];
        if (!xo_frame(C0, C1, C2, D)) continue;
        xo_pose cand;
        /* R = D^T E  (columns of D^T are camera axes, rows of E are world axes) */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                cand.R[3 * i + j] = D[i] * E[j] + D[3 + i] * E[3 + j] + D[6 + i] * E[6 + j];
        /* t = centroid_c - R centroid_w */
        double pw[3], pc[3];
        for (int i = 0; i < 3; ++i) {
            pw[i] = (P[0][i] + P[1][i] + P[2][i]) / 3.0;
            pc[i] = (C0[i] + C1[i] + C2[i]) / 3.0;
        }
        for (int i = 0; i < 3; ++i)
            cand.t[i] = pc[i] - (cand.R[3 * i] * pw[0] + cand.R[3 * i + 1] * pw[1] + cand.R[3 * i + 2] * pw[2]);
        /* 4th point squared pixel error, double, no z test */
        double xc = cand.R[0] * P[3][0] + cand.R[1] * P[3][1] + cand.R[2] * P[3][2] + cand.t[0];
        double yc = cand.R[3] * P[3][0] + cand.R[4] * P[3][1] + cand.R[5] * P[3][2] + cand.t[1];
        double zc = cand.R[6] * P[3][0] + cand.R[7] * P[3][1] + cand.R[8] * P[3][2] + cand.t[2];
        double up = cx + f * xc / zc, vp = cy + f * yc / zc;
        double e = (up - uv[3][0]) * (up - uv[3][0]) + (vp - uv[3][1]) * (vp - uv[3][1]);
        if (!found || e < best) { best = e; *out = cand; found = 1; }
    }
    return found;
}

/* ------------------------------------------------------------------ one sampling try */

/* returns 1 if the try is accepted; pose holds the try's result (identity if P3P failed),
 * cells[4] the sampled (y*Wo + x) indices (dsacstar_util.h:159-219). */
static int xo_sample_try(const xo_coords *co, uint64_t seed, uint64_t image, uint32_t hyp, uint32_t t,
                         float thr, double f, double cx, double cy, int sub,
                         xo_pose *pose, int cells[4])
{
    uint64_t st = xo_try_state(seed, image, hyp, t);
    double P[4][3], uv[4][2];
    float px[4], py[4];
    for (int j = 0; j < 4; ++j) {
        int x = xo_draw(st, 2 * j, co->Wo);       /* x first, then y (dsacstar_util.h:171-172) */
        int y = xo_draw(st, 2 * j + 1, co->Ho);
        cells[j] = y * co->Wo + x;
        px[j] = (float)(x * sub + sub / 2);       /* createSampling, dsacstar_util.h:70-72 */
        py[j] = (float)(y * sub + sub / 2);
        uv[j][0] = (double)px[j]; uv[j][1] = (double)py[j];
        xo_fetch(co, x, y, P[j]);
    }
    if (!xo_p3p(P, uv, f, cx, cy, pose)) {
        xo_pose_identity(pose);                   /* safeSolvePnP zeroes rvec/tvec, :114-116 */
        return 0;
    }
    for (int j = 0; j < 4; ++j) {
        float u, v;
        xo_project(pose, P[j], f, cx, cy, &u, &v);
        float dx = px[j] - u, dy = py[j] - v;
        double n = sqrt((double)dx * (double)dx + (double)dy * (double)dy);
        if (!(n < (double)thr)) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ fixed-order reductions */

static double xo_butterfly64(double p[64])
{
    double q[64];
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
        memcpy(p, q, sizeof(q));
    }
    return p[0];
}

/* soft-inlier score of one pose (dsacstar_util.h:316-343 over the map of :356-446):
 * lane l accumulates cells l, l+64, ... in row-major cell order, then the butterfly. */
static double xo_score(const xo_coords *co, const xo_pose *pose, float thr, float alpha, float maxReproj,
                       double f, double cx, double cy, int sub)
{
    int N = co->Ho * co->Wo;
    float beta = 5.0f / thr;
    double part[64];
    for (int l = 0; l < 64; ++l) {
        double acc = 0.0;
        for (int i = l; i < N; i += 64) {
            int y = i / co->Wo, x = i - y * co->Wo;
            double X[3];
            xo_fetch(co, x, y, X);
            float e = xo_cell_err(pose, X, (float)(x * sub + sub / 2), (float)(y * sub + sub / 2),
                                  f, cx, cy, maxReproj);
            float stf = beta * (e - thr);
            double st = (double)stf;
            st = 1.0 / (1.0 + xo_exp(-st));
            acc += 1.0 - st;
        }
        part[l] = acc;
    }
    double total = xo_butterfly64(part);
    float fac = alpha / (float)co->Wo / (float)co->Ho;
    return total * (double)fac;
}

/* ------------------------------------------------------------------ LM refinement */

#define XO_T 256   /* threads of the canonical reduction: 4 waves of 64 */

/* canonical sum over cells of per-cell 28-vectors: thread-strided partials, butterfly per wave,
 * waves added in order 0..3. */
typedef struct { double v[28]; } xo_vec28;

static void xo_reduce28(const xo_vec28 *part /*[XO_T]*/, double out[28])
{
    for (int k = 0; k < 28; ++k) {
        double tot = 0.0;
        for (int w = 0; w < XO_T / 64; ++w) {
            double p[64];
            for (int l = 0; l < 64; ++l) p[l] = part[w * 64 + l].v[k];
            double ws = xo_butterfly64(p);
            tot = (w == 0) ? ws : tot + ws;
        }
        out[k] = tot;
    }
}

/* normal equations of the reprojection residuals over the inlier set at pose p:
 * out[0..20] upper triangle of JtJ (row-major), out[21..26] Jt r, out[27] sum r^2 */
static void xo_normal_eq(const xo_coords *co, const unsigned char *inl, const xo_pose *p,
                         double f, double cx, double cy, int sub, double out[28])
{
    int N = co->Ho * co->Wo;
    xo_vec28 *part = (xo_vec28 *)malloc(sizeof(xo_vec28) * XO_T);
    for (int tdx = 0; tdx < XO_T; ++tdx) {
        double a[28];
        for (int k = 0; k < 28; ++k) a[k] = 0.0;
        for (int i = tdx; i < N; i += XO_T) {
            if (!inl[i]) continue;
            int y = i / co->Wo, x = i - y * co->Wo;
            double X[3];
            xo_fetch(co, x, y, X);
            double qx = p->R[0] * X[0] + p->R[1] * X[1] + p->R[2] * X[2];
            double qy = p->R[3] * X[0] + p->R[4] * X[1] + p->R[5] * X[2];
            double qz = p->R[6] * X[0] + p->R[7] * X[1] + p->R[8] * X[2];
            double xc = qx + p->t[0], yc = qy + p->t[1], zc = qz + p->t[2];
            double z = (zc != 0.0) ? 1.0 / zc : 1.0;
            double xn = xc * z, yn = yc * z;
            double ru = (xn * f + cx) - (double)(float)(x * sub + sub / 2);
            double rv = (yn * f + cy) - (double)(float)(y * sub + sub / 2);
            double fa = f * z;               /* du/dXc */
            double fc = -(fa * xn);          /* du/dZc */
            double fd = -(fa * yn);          /* dv/dZc */
            double Ju[6], Jv[6];
            Ju[0] = fc * qy;            Ju[1] = fa * qz - fc * qx;  Ju[2] = -(fa * qy);
            Ju[3] = fa;                 Ju[4] = 0.0;                Ju[5] = fc;
            Jv[0] = fd * qy - fa * qz;  Jv[1] = -(fd * qx);         Jv[2] = fa * qx;
            Jv[3] = 0.0;                Jv[4] = fa;                 Jv[5] = fd;
            int k = 0;
            for (int r = 0; r < 6; ++r)
                for (int c = r; c < 6; ++c)
                    a[k++] += Ju[r] * Ju[c] + Jv[r] * Jv[c];
            for (int r = 0; r < 6; ++r) a[21 + r] += Ju[r] * ru + Jv[r] * rv;
            a[27] += ru * ru + rv * rv;
        }
        for (int k = 0; k < 28; ++k) part[tdx].v[k] = a[k];
    }
    xo_reduce28(part, out);
    free(part);
}

/* solve (JtJ with diag*(1+lambda)) d = Jtr by Cholesky; 0 on breakdown */
static int xo_solve6(const double ne[28], double lambda, double d[6])
{
    double A[6][6];
    int k = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) { A[r][c] = ne[k]; A[c][r] = ne[k]; ++k; }
    for (int r = 0; r < 6; ++r) A[r][r] = A[r][r] * (1.0 + lambda);
    double L[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) L[i][j] = 0.0;
    for (int j = 0; j < 6; ++j) {
        double s = A[j][j];
        for (int m = 0; m < j; ++m) s -= L[j][m] * L[j][m];
        if (!(s > 0.0)) return 0;
        double ljj = sqrt(s);
        L[j][j] = ljj;
        for (int i = j + 1; i < 6; ++i) {
            double v = A[i][j];
            for (int m = 0; m < j; ++m) v -= L[i][m] * L[j][m];
            L[i][j] = v / ljj;
        }
    }
    double yv[6];
    for (int i = 0; i < 6; ++i) {
        double v = ne[21 + i];
        for (int m = 0; m < i; ++m) v -= L[i][m] * yv[m];
        yv[i] = v / L[i][i];
    }
    for (int i = 5; i >= 0; --i) {
        double v = yv[i];
        for (int m = i + 1; m < 6; ++m) v -= L[m][i] * d[m];
        d[i] = v / L[i][i];
    }
    for (int i = 0; i < 6; ++i)
        if (!(d[i] == d[i]) || fabs(d[i]) > 1.0e300) return 0;
    return 1;
}

/* param = prev (-) d : R = Exp(-d_w) R_prev, t = t_prev - d_t */
static void xo_apply_step(const xo_pose *prev, const double d[6], xo_pose *out)
{
    double wx = -d[0], wy = -d[1], wz = -d[2];
    double th2 = wx * wx + wy * wy + wz * wz;
    double th = sqrt(th2);
    double E[9];
    if (!(th > 1.0e-300)) {
        E[0] = 1.0; E[1] = -wz; E[2] = wy;
        E[3] = wz;  E[4] = 1.0; E[5] = -wx;
        E[6] = -wy; E[7] = wx;  E[8] = 1.0;
    } else {
        double s, c;
        xo_sincos(th, &s, &c);
        double kx = wx / th, ky = wy / th, kz = wz / th;
        double c1 = 1.0 - c;
        E[0] = c + c1 * kx * kx;      E[1] = c1 * kx * ky - s * kz; E[2] = c1 * kx * kz + s * ky;
        E[3] = c1 * kx * ky + s * kz; E[4] = c + c1 * ky * ky;      E[5] = c1 * ky * kz - s * kx;
        E[6] = c1 * kx * kz - s * ky; E[7] = c1 * ky * kz + s * kx; E[8] = c + c1 * kz * kz;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            out->R[3 * i + j] = E[3 * i] * prev->R[j] + E[3 * i + 1] * prev->R[3 + j] + E[3 * i + 2] * prev->R[6 + j];
    out->t[0] = prev->t[0] - d[3];
    out->t[1] = prev->t[1] - d[4];
    out->t[2] = prev->t[2] - d[5];
}

static const double XO_LAMBDA[33] = {   /* 10^k, k = -16..16 */
    1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3,
    1e-2, 1e-1, 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14,
    1e15, 1e16 };

/* cv::solvePnP(ITERATIVE, useExtrinsicGuess=true) restated; returns 0 on failure (pose untouched) */
static int xo_lm_pnp(const xo_coords *co, const unsigned char *inl, double f, double cx, double cy,
                     int sub, xo_pose *pose, int *evals)
{
    xo_pose cur = *pose, prev;
    double ne[28], ne_new[28];
    int lg = -3, iters = 0;
    xo_normal_eq(co, inl, &cur, f, cx, cy, sub, ne);
    ++*evals;
    for (;;) {
        double d[6];
        prev = cur;
        if (!xo_solve6(ne, XO_LAMBDA[lg + 16], d)) return 0;
        xo_apply_step(&prev, d, &cur);
        double prevErr = ne[27];
        xo_normal_eq(co, inl, &cur, f, cx, cy, sub, ne_new);
        ++*evals;
        while (ne_new[27] > prevErr) {
            if (++lg <= 16) {
                if (!xo_solve6(ne, XO_LAMBDA[lg + 16], d)) return 0;
                xo_apply_step(&prev, d, &cur);
                xo_normal_eq(co, inl, &cur, f, cx, cy, sub, ne_new);
                ++*evals;
            } else break;
        }
        lg = (lg - 1 > -16) ? lg - 1 : -16;
        double dn = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
        double pn = (3.0 - (prev.R[0] + prev.R[4] + prev.R[8]))
                    + prev.t[0] * prev.t[0] + prev.t[1] * prev.t[1] + prev.t[2] * prev.t[2];
        ++iters;
        if (iters >= XO_LM_MAX_ITER || dn < XO_FLT_EPSILON * XO_FLT_EPSILON * pn) break;
        memcpy(ne, ne_new, sizeof(ne));
    }
    for (int i = 0; i < 9; ++i) if (!(cur.R[i] == cur.R[i])) return 0;
    for (int i = 0; i < 3; ++i) if (!(cur.t[i] == cur.t[i])) return 0;
    *pose = cur;
    return 1;
}

/* ------------------------------------------------------------------ public entry points */

/*
 * Debug record layout (dbg may be NULL), doubles:
 *   [0] winner index   [1] refinement rounds accepted   [2] final inlier count
 *   [3] LM normal-equation evaluations   [4..15] winner pose before refinement (R row-major, t)
 *   [16..27] refined pose (R, t)
 */
int xo_dsac_forward_rgb(const float *coords, int64_t sc, int64_t sy, int64_t sx, int Ho, int Wo,
                        float *out_pose16, int nHyp, float thr, float focal, float ppx, float ppy,
                        float alpha, float maxReproj, int sub, uint64_t seed, uint64_t image,
                        uint32_t maxTries,
                        int32_t *out_cells /*[nHyp*4] or NULL*/, int32_t *out_tries /*[nHyp] or NULL*/,
                        double *out_scores /*[nHyp] or NULL*/, double *dbg /*[28] or NULL*/)
{
    if (!coords || !out_pose16 || nHyp <= 0 || Ho <= 0 || Wo <= 0 || sub <= 0 || maxTries == 0) return -1;
    xo_coords co = { coords, sc, sy, sx, Ho, Wo };
    double f = (double)focal, cx = (double)ppx, cy = (double)ppy;
    int N = Ho * Wo;

    xo_pose *hyps = (xo_pose *)malloc(sizeof(xo_pose) * (size_t)nHyp);
    double *scores = (double *)malloc(sizeof(double) * (size_t)nHyp);
    int32_t *cells = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)nHyp);
    int32_t *tries = (int32_t *)malloc(sizeof(int32_t) * (size_t)nHyp);

    /* sampleHypotheses + getReproErrs + getHypScores, parallel over hypotheses like the reference */
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < nHyp; ++h) {
        xo_pose p;
        int c4[4];
        uint32_t t = 0;
        int ok = 0;
        for (t = 0; t < maxTries; ++t) {
            ok = xo_sample_try(&co, seed, image, (uint32_t)h, t, thr, f, cx, cy, sub, &p, c4);
            if (ok) break;
        }
        hyps[h] = p;                               /* last try's pose if never accepted */
        for (int j = 0; j < 4; ++j) cells[4 * h + j] = c4[j];
        tries[h] = ok ? (int32_t)(t + 1) : -(int32_t)maxTries;
        scores[h] = xo_score(&co, &p, thr, alpha, maxReproj, f, cx, cy, sub);
    }

    /* softMax + draw(argmax): first maximum wins; any NaN score makes every prob NaN -> index 0 */
    int win = 0, anyNan = 0;
    for (int h = 0; h < nHyp; ++h) if (scores[h] != scores[h]) anyNan = 1;
    if (!anyNan)
        for (int h = 1; h < nHyp; ++h) if (scores[h] > scores[win]) win = h;

    xo_pose pose = hyps[win];
    if (dbg) {
        dbg[0] = (double)win;
        for (int i = 0; i < 9; ++i) dbg[4 + i] = pose.R[i];
        for (int i = 0; i < 3; ++i) dbg[13 + i] = pose.t[i];
    }

    /* refineHyp */
    float *errs = (float *)malloc(sizeof(float) * (size_t)N);
    unsigned char *inl = (unsigned char *)malloc((size_t)N);
    for (int i = 0; i < N; ++i) {
        int y = i / Wo, x = i - y * Wo;
        double X[3];
        xo_fetch(&co, x, y, X);
        errs[i] = xo_cell_err(&pose, X, (float)(x * sub + sub / 2), (float)(y * sub + sub / 2), f, cx, cy, maxReproj);
    }
    unsigned best = 4;
    int rounds = 0, evals = 0;
    unsigned finalInl = 0;
    for (int step = 0; step < XO_MAX_REF_STEPS; ++step) {
        unsigned cnt = 0;
        for (int i = 0; i < N; ++i) { inl[i] = (errs[i] < thr) ? 1 : 0; cnt += inl[i]; }
        if (cnt <= best) break;
        best = cnt;
        xo_pose upd = pose;
        if (!xo_lm_pnp(&co, inl, f, cx, cy, sub, &upd, &evals)) break;
        pose = upd;
        finalInl = cnt;
        ++rounds;
        for (int i = 0; i < N; ++i) {
            int y = i / Wo, x = i - y * Wo;
            double X[3];
            xo_fetch(&co, x, y, X);
            errs[i] = xo_cell_err(&pose, X, (float)(x * sub + sub / 2), (float)(y * sub + sub / 2), f, cx, cy, maxReproj);
        }
    }

    /* pose2trans: inverse of [R t; 0 1], stored float row-major */
    double Ti[16];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Ti[4 * i + j] = pose.R[3 * j + i];
        Ti[4 * i + 3] = -(pose.R[i] * pose.t[0] + pose.R[3 + i] * pose.t[1] + pose.R[6 + i] * pose.t[2]);
    }
    Ti[12] = 0.0; Ti[13] = 0.0; Ti[14] = 0.0; Ti[15] = 1.0;
    for (int i = 0; i < 16; ++i) out_pose16[i] = (float)Ti[i];

    if (out_cells) memcpy(out_cells, cells, sizeof(int32_t) * 4 * (size_t)nHyp);
    if (out_tries) memcpy(out_tries, tries, sizeof(int32_t) * (size_t)nHyp);
    if (out_scores) memcpy(out_scores, scores, sizeof(double) * (size_t)nHyp);
    if (dbg) {
        dbg[1] = (double)rounds; dbg[2] = (double)finalInl; dbg[3] = (double)evals;
        for (int i = 0; i < 9; ++i) dbg[16 + i] = pose.R[i];
        for (int i = 0; i < 3; ++i) dbg[25 + i] = pose.t[i];
    }
    free(errs); free(inl); free(hyps); free(scores); free(cells); free(tries);
    return 0;
}

/* --- unit-test hooks (known-answer tests call these directly) --- */

int xo_test_p3p(const double *P12, const double *uv8, double f, double cx, double cy, double *Rt12)
{
    double P[4][3], uv[4][2];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 3; ++j) P[i][j] = P12[3 * i + j];
        uv[i][0] = uv8[2 * i]; uv[i][1] = uv8[2 * i + 1];
    }
    xo_pose p;
    int ok = xo_p3p(P, uv, f, cx, cy, &p);
    if (ok) {
        for (int i = 0; i < 9; ++i) Rt12[i] = p.R[i];
        for (int i = 0; i < 3; ++i) Rt12[9 + i] = p.t[i];
    }
    return ok;
}

double xo_test_score(const float *coords, int64_t sc, int64_t sy, int64_t sx, int Ho, int Wo,
                     const double *Rt12, float thr, float alpha, float maxReproj,
                     float focal, float ppx, float ppy, int sub)
{
    xo_coords co = { coords, sc, sy, sx, Ho, Wo };
    xo_pose p;
    for (int i = 0; i < 9; ++i) p.R[i] = Rt12[i];
    for (int i = 0; i < 3; ++i) p.t[i] = Rt12[9 + i];
    return xo_score(&co, &p, thr, alpha, maxReproj, (double)focal, (double)ppx, (double)ppy, sub);
}

double xo_test_exp(double x) { return xo_exp(x); }
void xo_test_sincos(double x, double *s, double *c) { xo_sincos(x, s, c); }
int xo_test_quartic(const double *A5, double *roots4) { return xo_quartic(A5[0], A5[1], A5[2], A5[3], A5[4], roots4); }
void xo_test_draws(uint64_t seed, uint64_t image, uint32_t hyp, uint32_t t, int Wo, int Ho, int32_t *xy8)
{
    uint64_t st = xo_try_state(seed, image, hyp, t);
    for (int j = 0; j < 4; ++j) { xy8[2 * j] = xo_draw(st, 2 * j, Wo); xy8[2 * j + 1] = xo_draw(st, 2 * j + 1, Ho); }
}
void xo_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int xo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* backward_rgb restatement (same translation unit: it uses the static helpers above) */
#include "dsac_bwd_oracle.c"
