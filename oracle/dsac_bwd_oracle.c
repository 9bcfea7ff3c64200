/*
 * oracle/dsac_bwd_oracle.c — CPU restatement of CrossLoc's `dsacstar.backward_rgb` (DSAC* expected pose loss and its
 * gradient with respect to the scene coordinates).  Included at the end of dsac_oracle.c (one translation unit).
 *
 * TEST INFRASTRUCTURE ONLY — same rules as dsac_oracle.c.  PARITY UNPINNED vs the reference binary (OpenCV absent);
 * pinned by numeric-derivative known answers in tests/test_oracle_dsac_bwd.py.
 *
 * What is restated (reference file:line, relative to /root/reference/dsacstar/):
 *   orchestration                    dsacstar.cpp:200-483
 *   getReproErrs(calcJ=true)         dsacstar_util.h:403-434   (rows d err / d (rvec, tvec), zero above maxReproj)
 *   softMax                          dsacstar_util.h:684-704
 *   loss / calcAngularDistance       dsacstar_loss.h:47-88
 *   dLoss                            dsacstar_loss.h:99-212
 *   dProjectdObj                     dsacstar_derivative.h:51-106
 *   dPNP (central differences)       dsacstar_derivative.h:131-190
 *   dScore / dSMScore                dsacstar_derivative.h:209-402
 *   trans2pose / pose2trans / getMax dsacstar_util.h:759-790, 821-834
 *
 * Third-party arithmetic restated from its published behaviour (OpenCV 3.4.2): cv::Rodrigues in both directions with
 * its analytic Jacobian (exact derivative of R = cos t I + (1-cos t) k k^T + sin t [k]x), the pose Jacobian of
 * cv::projectPoints (chain rule through Xc = R X + t, same z guard as the projection), cv::Mat::inv(DECOMP_SVD) of the
 * symmetric 6x6 J^T J as an eigen-decomposition pseudo-inverse with OpenCV's threshold 2*DBL_EPSILON*sum(w).
 *
 * Deliberate, documented deviations (on top of those of the forward restatement): sums over cells use the canonical
 * fixed order (thread-strided partials + butterfly) instead of x-major serial order; sum_c dRepro(c) J(c) is formed
 * before the product with dPNP (associativity); the ground-truth pose is inverted as a rigid transform (the reference
 * takes a general 4x4 inverse and lets cv::Rodrigues re-orthonormalise); R -> rvec uses atan2(sin, cos) where OpenCV
 * uses acos(cos); a non-finite dLoss is zeroed like a NaN one; the entropy print-out is not computed.
 */

#define XO_PROB_THRESH 0.001          /* dsacstar_derivative.h:36 */
#define XO_EPS 0.00000001             /* dsacstar_util.h:45 */
#define XO_MAXLOSS 10000000.0         /* dsacstar_loss.h:35 */
#define XO_PI_REF 3.1415926           /* dsacstar_util.h:46 (calcAngularDistance) */
#define XO_CV_PI 3.1415926535897932384626433832795
#define XO_DBL_EPSILON 2.2204460492503131e-16

/* ------------------------------------------------------------------ deterministic inverse trigonometry */

/* atan2 from + - * / sqrt and xo_sincos only: rational first guess (error < 5e-3), then Newton-like corrections
 * t += asin(sin(target - t)) with the asin series to 5th order (error after one step ~1e-17, two more for margin) */
static double xo_atan2(double y, double x)
{
    if (x == 0.0 && y == 0.0) return 0.0;
    double n = sqrt(x * x + y * y);
    double cn = x / n, sn = y / n;
    double ax = fabs(x), ay = fabs(y);
    double t;
    if (ay <= ax) { double a = ay / ax; t = a / (1.0 + 0.28 * a * a); }
    else { double a = ax / ay; t = 1.5707963267948966 - a / (1.0 + 0.28 * a * a); }
    if (x < 0.0) t = 3.141592653589793 - t;
    if (y < 0.0) t = -t;
    for (int it = 0; it < 3; ++it) {
        double s, c;
        xo_sincos(t, &s, &c);
        double d = sn * c - cn * s;
        double d2 = d * d;
        t = t + d * (1.0 + d2 * (1.0 / 6.0 + d2 * (3.0 / 40.0)));
    }
    return t;
}

static double xo_acos(double v)
{
    return xo_atan2(sqrt((1.0 - v) * (1.0 + v)), v);
}

/* ------------------------------------------------------------------ Rodrigues */

/* cv::Rodrigues(matrix -> vector) for a rotation matrix */
static void xo_log_so3(const double R[9], double r[3])
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    double theta = xo_atan2(s, c);
    if (s < 1e-5) {
        if (c > 0.0) { r[0] = 0.0; r[1] = 0.0; r[2] = 0.0; return; }
        double t;
        t = (R[0] + 1.0) * 0.5; rx = sqrt(t > 0.0 ? t : 0.0);
        t = (R[4] + 1.0) * 0.5; ry = sqrt(t > 0.0 ? t : 0.0) * (R[1] < 0.0 ? -1.0 : 1.0);
        t = (R[8] + 1.0) * 0.5; rz = sqrt(t > 0.0 ? t : 0.0) * (R[2] < 0.0 ? -1.0 : 1.0);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && ((R[5] > 0.0) != (ry * rz > 0.0))) rz = -rz;
        theta = theta / sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
        return;
    }
    double vth = (1.0 / (2.0 * s)) * theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

/* cv::Rodrigues(vector -> matrix) */
static void xo_exp_so3(const double r[3], double R[9])
{
    double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < XO_DBL_EPSILON) {
        for (int i = 0; i < 9; ++i) R[i] = 0.0;
        R[0] = 1.0; R[4] = 1.0; R[8] = 1.0;
        return;
    }
    double s, c;
    xo_sincos(th, &s, &c);
    double c1 = 1.0 - c, ith = 1.0 / th;
    double kx = r[0] * ith, ky = r[1] * ith, kz = r[2] * ith;
    R[0] = c + c1 * kx * kx;      R[1] = c1 * kx * ky - s * kz; R[2] = c1 * kx * kz + s * ky;
    R[3] = c1 * kx * ky + s * kz; R[4] = c + c1 * ky * ky;      R[5] = c1 * ky * kz - s * kx;
    R[6] = c1 * kx * kz - s * ky; R[7] = c1 * ky * kz + s * kx; R[8] = c + c1 * kz * kz;
}

/* dR[(3a+b)*3 + c] = d R[a][b] / d r_c at rotation vector r (the transpose of OpenCV's 3x9 jacobian) */
static void xo_rodrigues_jac(const double r[3], double dR[27])
{
    double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (int i = 0; i < 27; ++i) dR[i] = 0.0;
    if (th < XO_DBL_EPSILON) {
        /* generators: d/dr_x = [[0,0,0],[0,0,-1],[0,1,0]] etc. */
        dR[5 * 3 + 0] = -1.0; dR[7 * 3 + 0] = 1.0;
        dR[2 * 3 + 1] = 1.0;  dR[6 * 3 + 1] = -1.0;
        dR[1 * 3 + 2] = -1.0; dR[3 * 3 + 2] = 1.0;
        return;
    }
    double s, c;
    xo_sincos(th, &s, &c);
    double c1 = 1.0 - c, ith = 1.0 / th;
    double k[3] = { r[0] * ith, r[1] * ith, r[2] * ith };
    for (int i = 0; i < 3; ++i) {
        double dk[3];
        for (int j = 0; j < 3; ++j) dk[j] = ((i == j ? 1.0 : 0.0) - k[i] * k[j]) * ith;
        double ski = s * k[i], cki = c * k[i];
        /* K = [k]x, dK = [dk]x */
        double K[9] = { 0.0, -k[2], k[1], k[2], 0.0, -k[0], -k[1], k[0], 0.0 };
        double dK[9] = { 0.0, -dk[2], dk[1], dk[2], 0.0, -dk[0], -dk[1], dk[0], 0.0 };
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double v = ski * (k[a] * k[b]) + c1 * (dk[a] * k[b] + k[a] * dk[b]) + cki * K[3 * a + b] + s * dK[3 * a + b];
                if (a == b) v -= ski;
                dR[(3 * a + b) * 3 + i] = v;
            }
    }
}

/* ------------------------------------------------------------------ per-cell derivatives */

/* Row of the residual Jacobian of one cell at pose (R, t) with rotation-vector derivative dR:
 * J6 = d max(|proj - pt|, EPS) / d (rvec, tvec), all-zero when that error exceeds maxReproj
 * (dsacstar_util.h:403-434, dsacstar.cpp:386-405).  Returns the (unclamped) error. */
static double xo_resid_row(const xo_pose *p, const double dR[27], const double X[3], float px, float py,
                           double f, double cx, double cy, float maxReproj, double J6[6])
{
    double qx = p->R[0] * X[0] + p->R[1] * X[1] + p->R[2] * X[2];
    double qy = p->R[3] * X[0] + p->R[4] * X[1] + p->R[5] * X[2];
    double qz = p->R[6] * X[0] + p->R[7] * X[1] + p->R[8] * X[2];
    double xc = qx + p->t[0], yc = qy + p->t[1], zc = qz + p->t[2];
    double z = (zc != 0.0) ? 1.0 / zc : 1.0;
    double xn = xc * z, yn = yc * z;
    float uf = (float)(xn * f + cx), vf = (float)(yn * f + cy);
    float dxf = uf - px, dyf = vf - py;
    double err = sqrt((double)dxf * (double)dxf + (double)dyf * (double)dyf);
    if (err < XO_EPS) err = XO_EPS;
    for (int i = 0; i < 6; ++i) J6[i] = 0.0;
    if (err > (double)maxReproj) return err;
    double nx = 1.0 / err * (double)dxf, ny = 1.0 / err * (double)dyf;
    double fa = f * z;                 /* du/dXc = dv/dYc */
    double fc = -(fa * xn);            /* du/dZc */
    double fd = -(fa * yn);            /* dv/dZc */
    for (int c = 0; c < 3; ++c) {
        double dX = dR[0 * 3 + c] * X[0] + dR[1 * 3 + c] * X[1] + dR[2 * 3 + c] * X[2];
        double dY = dR[3 * 3 + c] * X[0] + dR[4 * 3 + c] * X[1] + dR[5 * 3 + c] * X[2];
        double dZ = dR[6 * 3 + c] * X[0] + dR[7 * 3 + c] * X[1] + dR[8 * 3 + c] * X[2];
        double ju = fa * dX + fc * dZ, jv = fa * dY + fd * dZ;
        J6[c] = nx * ju + ny * jv;
    }
    J6[3] = nx * fa;
    J6[4] = ny * fa;
    J6[5] = nx * fc + ny * fd;
    return err;
}

/* dProjectdObj, dsacstar_derivative.h:51-106 (expression order kept) */
static void xo_dproject_dobj(const xo_pose *p, const double X[3], float ptx, float pty,
                             double f, double ppx, double ppy, float maxReproj, double out[3])
{
    out[0] = 0.0; out[1] = 0.0; out[2] = 0.0;
    double ox = p->R[0] * X[0] + p->R[1] * X[1] + p->R[2] * X[2] + p->t[0];
    double oy = p->R[3] * X[0] + p->R[4] * X[1] + p->R[5] * X[2] + p->t[1];
    double oz = p->R[6] * X[0] + p->R[7] * X[1] + p->R[8] * X[2] + p->t[2];
    if (fabs(oz) < XO_EPS) return;
    double px = f * ox / oz + ppx;
    double py = f * oy / oz + ppy;
    double ex = (double)ptx - px, ey = (double)pty - py;
    double err = sqrt(ex * ex + ey * ey);
    if (err > (double)maxReproj) return;
    err += XO_EPS;
    for (int k = 0; k < 3; ++k) {
        double pxd = f * p->R[k] / oz - f * ox / oz / oz * p->R[6 + k];
        double pyd = f * p->R[3 + k] / oz - f * oy / oz / oz * p->R[6 + k];
        out[k] = 0.5 / err * (2.0 * ex * -pxd + 2.0 * ey * -pyd);
    }
}

/* ------------------------------------------------------------------ 6x6 symmetric pseudo-inverse */

/* cv::Mat::inv(DECOMP_SVD) of a symmetric positive semi-definite 6x6: cyclic Jacobi (fixed 12 sweeps),
 * eigenvalues <= 2*DBL_EPSILON*sum|w| are dropped */
static void xo_pinv6(const double A_[36], double Ainv[36])
{
    double A[6][6], V[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) { A[i][j] = A_[6 * i + j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 12; ++sweep)
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                double apq = A[p][q];
                if (apq == 0.0) continue;
                double tau = (A[q][q] - A[p][p]) / (2.0 * apq);
                double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < 6; ++k) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; ++k) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    double sum = 0.0;
    for (int i = 0; i < 6; ++i) sum += fabs(A[i][i]);
    double thr = sum * (2.0 * XO_DBL_EPSILON);
    double wi[6];
    for (int i = 0; i < 6; ++i) wi[i] = (fabs(A[i][i]) > thr) ? 1.0 / A[i][i] : 0.0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double v = 0.0;
            for (int k = 0; k < 6; ++k) v += V[i][k] * wi[k] * V[j][k];
            Ainv[6 * i + j] = v;
        }
}

/* ------------------------------------------------------------------ pose loss and its derivative */

typedef struct { double Rc2w[9]; double C[3]; double R2[9]; double t2[3]; } xo_gt;   /* ground truth, both forms */

static void xo_gt_from_pose16(const float *gt16, xo_gt *g)
{
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) g->Rc2w[3 * i + j] = (double)gt16[4 * i + j];
        g->C[i] = (double)gt16[4 * i + 3];
    }
    /* trans2pose (dsacstar_util.h:777-790) as a rigid inverse: world->camera.  The reference passes the rotation
     * through cv::Rodrigues and back (dsacstar_loss.h:107-108), which makes it exactly orthonormal; float ground
     * truth is not, and trace(R1 R2^T) > 3 would otherwise hit the clamp where the angle derivative is infinite. */
    double Rt[9], r2[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rt[3 * i + j] = g->Rc2w[3 * j + i];
    xo_log_so3(Rt, r2);
    xo_exp_so3(r2, g->R2);
    for (int i = 0; i < 3; ++i)
        g->t2[i] = -(g->R2[3 * i] * g->C[0] + g->R2[3 * i + 1] * g->C[1] + g->R2[3 * i + 2] * g->C[2]);
}

/* loss(pose2trans(est), gtTrans), dsacstar_loss.h:47-88 */
static double xo_pose_loss(const xo_pose *est, const xo_gt *g, double wRot, double wTrans, double cut)
{
    /* estTrans = [R t]^-1: rot1 = R^T, centre c1 = -R^T t;  rotDiff = rot2 * rot1^T = Rc2w * R */
    double trace = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) trace += g->Rc2w[3 * i + k] * est->R[3 * k + i];
    trace = trace > 3.0 ? 3.0 : (trace < -1.0 ? -1.0 : trace);
    double rotErr = 180.0 * xo_acos((trace - 1.0) / 2.0) / XO_PI_REF;
    double d2 = 0.0;
    for (int i = 0; i < 3; ++i) {
        double c1 = -(est->R[i] * est->t[0] + est->R[3 + i] * est->t[1] + est->R[6 + i] * est->t[2]);
        double d = c1 - g->C[i];
        d2 += d * d;
    }
    double tErr = sqrt(d2);
    double loss = wRot * rotErr + wTrans * tErr;
    if (loss > cut) loss = sqrt(cut * loss);
    return loss < XO_MAXLOSS ? loss : XO_MAXLOSS;
}

/* dLoss, dsacstar_loss.h:99-212: 1x6 derivative w.r.t. (rvec, tvec) of the estimate; dR = rodrigues jacobian at it */
static void xo_dloss(const xo_pose *est, const double dR[27], const xo_gt *g, double wRot, double wTrans, double cut,
                     double jac[6])
{
    for (int i = 0; i < 6; ++i) jac[i] = 0.0;
    const double *R1 = est->R, *R2 = g->R2;
    double trace = 0.0;                                  /* trace(R1 * R2^T) */
    for (int a = 0; a < 3; ++a)
        for (int k = 0; k < 3; ++k) trace += R1[3 * a + k] * R2[3 * a + k];
    trace = trace > 3.0 ? 3.0 : (trace < -1.0 ? -1.0 : trace);
    double rotErr = 180.0 * xo_acos((trace - 1.0) / 2.0) / XO_CV_PI;
    double invT1[3], invT2[3], diff[3];
    for (int i = 0; i < 3; ++i) {
        invT1[i] = R1[i] * est->t[0] + R1[3 + i] * est->t[1] + R1[6 + i] * est->t[2];
        invT2[i] = R2[i] * g->t2[0] + R2[3 + i] * g->t2[1] + R2[6 + i] * g->t2[2];
        diff[i] = invT1[i] - invT2[i];
    }
    double tErr = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
    double loss = wRot * rotErr + wTrans * tErr;
    int cutLoss = 0;
    if (loss > cut) { loss = sqrt(loss); cutLoss = 1; }
    if (loss > XO_MAXLOSS) return;
    if ((tErr + rotErr) < XO_EPS) return;
    double dD[3];
    for (int i = 0; i < 3; ++i) dD[i] = diff[i] / tErr;
    /* translation part: dDist_dInvT1 * invRot1 */
    for (int j = 0; j < 3; ++j)
        jac[3 + j] += (dD[0] * R1[j * 3 + 0] + dD[1] * R1[j * 3 + 1] + dD[2] * R1[j * 3 + 2]) * wTrans;
    /* rotation through invT1 = R1^T t1: d invT1_i / d R1[j][i] = t1_j */
    for (int c = 0; c < 3; ++c) {
        double v = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) v += dD[i] * est->t[j] * dR[(3 * j + i) * 3 + c];
        jac[c] += v * wTrans;
    }
    /* rotation angle: d trace / d R1[a][k] = R2[a][k] */
    double fac = 180.0 / XO_CV_PI * -1.0 / sqrt(3.0 - trace * trace + 2.0 * trace);
    for (int c = 0; c < 3; ++c) {
        double v = 0.0;
        for (int m = 0; m < 9; ++m) v += R2[m] * dR[m * 3 + c];
        jac[c] += fac * v * wRot;
    }
    if (cutLoss)
        for (int i = 0; i < 6; ++i) jac[i] *= 0.5 / loss;
    /* the reference tests for NaN only (loss.h:207-208); an infinite entry (estimate == ground truth to rounding,
     * where d acos is unbounded) would turn into NaN one product later, so it is treated the same way here */
    for (int i = 0; i < 6; ++i)
        if (!(jac[i] == jac[i]) || fabs(jac[i]) > 1.0e300) { for (int k = 0; k < 6; ++k) jac[k] = 0.0; return; }
}

/* ------------------------------------------------------------------ dPNP (central differences over P3P) */

/* 6x12 jacobian (row-major, [6][12]) of the P3P pose w.r.t. the four sampled scene coordinates; columns 9..11 stay
 * zero (the 4th point only disambiguates).  dsacstar_derivative.h:131-190: float points, float eps, the
 * += eps / -= 2 eps / += eps sequence is carried from column to column exactly as written there. */
static void xo_dpnp(const float obj[4][3], const double uv[4][2], double f, double cx, double cy, double J[72])
{
    const float eps = 0.001f;
    float pts[4][3];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) pts[i][j] = obj[i][j];
    for (int i = 0; i < 72; ++i) J[i] = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double P[4][3];
            xo_pose fwd, bwd;
            pts[i][j] += eps;
            for (int a = 0; a < 4; ++a) for (int b = 0; b < 3; ++b) P[a][b] = (double)pts[a][b];
            int okf = xo_p3p(P, uv, f, cx, cy, &fwd);
            pts[i][j] -= 2 * eps;
            for (int a = 0; a < 4; ++a) for (int b = 0; b < 3; ++b) P[a][b] = (double)pts[a][b];
            int okb = okf ? xo_p3p(P, uv, f, cx, cy, &bwd) : 0;
            if (!okf || !okb) { for (int k = 0; k < 72; ++k) J[k] = 0.0; return; }
            pts[i][j] += eps;
            double rf[3], rb[3];
            xo_log_so3(fwd.R, rf);
            xo_log_so3(bwd.R, rb);
            double den = (double)(2 * eps);
            int col = i * 3 + j, bad = 0;
            for (int k = 0; k < 3; ++k) {
                double a = (rf[k] - rb[k]) / den, b = (fwd.t[k] - bwd.t[k]) / den;
                J[k * 12 + col] = a;
                J[(3 + k) * 12 + col] = b;
                if (!(a == a) || !(b == b)) bad = 1;
            }
            if (bad) { for (int k = 0; k < 72; ++k) J[k] = 0.0; return; }
        }
}

/* ------------------------------------------------------------------ refinement shared with the forward pass */

/* refineHyp (dsacstar_util.h:522-597): pose is refined in place; inlAcc receives the inlier map of the last
 * successful re-fit (all zero if none), *nAcc its size.  errs/inl are N-sized scratch. */
static void xo_refine(const xo_coords *co, float thr, float maxReproj, double f, double cx, double cy, int sub,
                      xo_pose *pose, unsigned char *inlAcc, unsigned *nAcc, float *errs, unsigned char *inl,
                      int *roundsOut, int *evalsOut)
{
    int N = co->Ho * co->Wo, Wo = co->Wo;
    for (int i = 0; i < N; ++i) {
        int y = i / Wo, x = i - y * Wo;
        double X[3];
        xo_fetch(co, x, y, X);
        errs[i] = xo_cell_err(pose, X, (float)(x * sub + sub / 2), (float)(y * sub + sub / 2), f, cx, cy, maxReproj);
    }
    if (inlAcc) memset(inlAcc, 0, (size_t)N);
    unsigned best = 4, finalInl = 0;
    int rounds = 0, evals = 0;
    for (int step = 0; step < XO_MAX_REF_STEPS; ++step) {
        unsigned cnt = 0;
        for (int i = 0; i < N; ++i) { inl[i] = (errs[i] < thr) ? 1 : 0; cnt += inl[i]; }
        if (cnt <= best) break;
        best = cnt;
        xo_pose upd = *pose;
        if (!xo_lm_pnp(co, inl, f, cx, cy, sub, &upd, &evals)) break;
        *pose = upd;
        finalInl = cnt;
        if (inlAcc) memcpy(inlAcc, inl, (size_t)N);
        ++rounds;
        for (int i = 0; i < N; ++i) {
            int y = i / Wo, x = i - y * Wo;
            double X[3];
            xo_fetch(co, x, y, X);
            errs[i] = xo_cell_err(pose, X, (float)(x * sub + sub / 2), (float)(y * sub + sub / 2), f, cx, cy, maxReproj);
        }
    }
    if (nAcc) *nAcc = finalInl;
    if (roundsOut) *roundsOut = rounds;
    if (evalsOut) *evalsOut = evals;
}

/* ------------------------------------------------------------------ the entry point */

/* per-hypothesis record exported for parity tests (doubles) */
#define XO_BWD_REC 64
/*  [0] prob  [1] loss  [2] active  [3] accepted inliers  [4] clampI (path I zeroed)  [5] sog
 *  [6..17] refined pose (R, t)  [18..23] dLoss/dHyp  [24..29] w = pinv(JtJ) dLoss^T  [30..41] support-point gradients
 *  [42] max |jacobeanR|  [43] max |dPNP|  [44..49] g6 = sum_c dRepro(c) J_init(c)  [50..52] rvec of the refined pose */

int xo_dsac_backward_rgb(const float *coords, int64_t sc, int64_t sy, int64_t sx, int Ho, int Wo,
                         float *grad, int64_t gsc, int64_t gsy, int64_t gsx,      /* accumulated (+=) */
                         const float *gtPose16, int nHyp, float thr, float focal, float ppx, float ppy,
                         float wRot, float wTrans, float softClamp, float alpha, float maxReproj, int sub,
                         uint64_t seed, uint64_t image, uint32_t maxTries,
                         double *outLoss, double *rec /*[nHyp*XO_BWD_REC] or NULL*/)
{
    if (!coords || !grad || !gtPose16 || !outLoss || nHyp <= 0 || Ho <= 0 || Wo <= 0 || sub <= 0 || maxTries == 0) return -1;
    xo_coords co = { coords, sc, sy, sx, Ho, Wo };
    const double f = (double)focal, cx = (double)ppx, cy = (double)ppy;
    const int N = Ho * Wo;
    xo_gt gt;
    xo_gt_from_pose16(gtPose16, &gt);

    xo_pose *init = (xo_pose *)malloc(sizeof(xo_pose) * (size_t)nHyp);
    xo_pose *ref = (xo_pose *)malloc(sizeof(xo_pose) * (size_t)nHyp);
    double *scores = (double *)malloc(sizeof(double) * (size_t)nHyp);
    double *probs = (double *)malloc(sizeof(double) * (size_t)nHyp);
    double *losses = (double *)malloc(sizeof(double) * (size_t)nHyp);
    double *sog = (double *)calloc((size_t)nHyp, sizeof(double));
    int32_t *cells = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)nHyp);
    unsigned char *inlAcc = (unsigned char *)calloc((size_t)nHyp * (size_t)N, 1);
    unsigned *nAcc = (unsigned *)calloc((size_t)nHyp, sizeof(unsigned));
    double *dRinit = (double *)calloc((size_t)nHyp * 27, sizeof(double));
    double *dRref = (double *)calloc((size_t)nHyp * 27, sizeof(double));
    double *wv = (double *)calloc((size_t)nHyp * 6, sizeof(double));
    double *spg = (double *)calloc((size_t)nHyp * 12, sizeof(double));
    int *clampI = (int *)calloc((size_t)nHyp, sizeof(int));
    double *recs = rec;
    if (recs) memset(recs, 0, sizeof(double) * (size_t)nHyp * XO_BWD_REC);

    /* 1. sampleHypotheses + soft-inlier scores (same sampler as the forward pass, keyed by randomSeed) */
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
        init[h] = p;
        for (int j = 0; j < 4; ++j) cells[4 * h + j] = c4[j];
        scores[h] = xo_score(&co, &p, thr, alpha, maxReproj, f, cx, cy, sub);
    }

    /* 2. softMax (dsacstar_util.h:684-704) */
    {
        double maxScore = 0.0, sum = 0.0;
        for (int i = 0; i < nHyp; ++i) if (i == 0 || scores[i] > maxScore) maxScore = scores[i];
        for (int i = 0; i < nHyp; ++i) { probs[i] = xo_exp(scores[i] - maxScore); sum += probs[i]; }
        for (int i = 0; i < nHyp; ++i) probs[i] /= sum;
    }

    /* 3. refinement of every hypothesis that matters + 4. losses + path I per-hypothesis quantities */
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < nHyp; ++h) {
        ref[h] = init[h];
        int active = !(probs[h] < XO_PROB_THRESH);
        double dLossH[6] = { 0, 0, 0, 0, 0, 0 }, maxJR = 0.0, rv[3] = { 0, 0, 0 };
        if (active) {
            float *errs = (float *)malloc(sizeof(float) * (size_t)N);
            unsigned char *inl = (unsigned char *)malloc((size_t)N);
            xo_refine(&co, thr, maxReproj, f, cx, cy, sub, &ref[h], inlAcc + (size_t)h * N, &nAcc[h], errs, inl, NULL, NULL);
            free(errs); free(inl);
        }
        losses[h] = xo_pose_loss(&ref[h], &gt, (double)wRot, (double)wTrans, (double)softClamp);
        if (active) {
            double r0[3];
            xo_log_so3(init[h].R, r0);
            xo_rodrigues_jac(r0, dRinit + (size_t)h * 27);
            xo_log_so3(ref[h].R, rv);
            xo_rodrigues_jac(rv, dRref + (size_t)h * 27);
            xo_dloss(&ref[h], dRref + (size_t)h * 27, &gt, (double)wRot, (double)wTrans, (double)softClamp, dLossH);
            if (nAcc[h] >= 4) {
                /* J^T J over the accepted inliers, canonical reduction */
                xo_vec28 *part = (xo_vec28 *)malloc(sizeof(xo_vec28) * XO_T);
                const unsigned char *ia = inlAcc + (size_t)h * N;
                for (int tdx = 0; tdx < XO_T; ++tdx) {
                    double a[28];
                    for (int k = 0; k < 28; ++k) a[k] = 0.0;
                    for (int i = tdx; i < N; i += XO_T) {
                        if (!ia[i]) continue;
                        int y = i / Wo, x = i - y * Wo;
                        double X[3], J6[6];
                        xo_fetch(&co, x, y, X);
                        xo_resid_row(&ref[h], dRref + (size_t)h * 27, X, (float)(x * sub + sub / 2), (float)(y * sub + sub / 2),
                                     f, cx, cy, maxReproj, J6);
                        int k = 0;
                        for (int r = 0; r < 6; ++r)
                            for (int c = r; c < 6; ++c) a[k++] += J6[r] * J6[c];
                    }
                    for (int k = 0; k < 28; ++k) part[tdx].v[k] = a[k];
                }
                double ne[28], A[36], Ainv[36];
                xo_reduce28(part, ne);
                free(part);
                int k = 0;
                for (int r = 0; r < 6; ++r)
                    for (int c = r; c < 6; ++c) { A[6 * r + c] = ne[k]; A[6 * c + r] = ne[k]; ++k; }
                xo_pinv6(A, Ainv);
                /* jacobeanR = -(JtJ)^-1 J^T: clamp if any entry exceeds 10 (dsacstar.cpp:408-412) */
                for (int i = 0; i < N; ++i) {
                    if (!ia[i]) continue;
                    int y = i / Wo, x = i - y * Wo;
                    double X[3], J6[6];
                    xo_fetch(&co, x, y, X);
                    xo_resid_row(&ref[h], dRref + (size_t)h * 27, X, (float)(x * sub + sub / 2), (float)(y * sub + sub / 2),
                                 f, cx, cy, maxReproj, J6);
                    for (int r = 0; r < 6; ++r) {
                        double u = 0.0;
                        for (int c = 0; c < 6; ++c) u += Ainv[6 * r + c] * J6[c];
                        u = fabs(u);
                        if (u > maxJR) maxJR = u;
                    }
                }
                clampI[h] = (maxJR > 10.0) ? 1 : 0;
                for (int r = 0; r < 6; ++r) {
                    double v = 0.0;
                    for (int c = 0; c < 6; ++c) v += Ainv[6 * r + c] * dLossH[c];
                    wv[(size_t)h * 6 + r] = v;
                }
            }
        }
        if (recs) {
            double *q = recs + (size_t)h * XO_BWD_REC;
            q[0] = probs[h]; q[1] = losses[h]; q[2] = (double)active; q[3] = (double)nAcc[h]; q[4] = (double)clampI[h];
            for (int i = 0; i < 9; ++i) q[6 + i] = ref[h].R[i];
            for (int i = 0; i < 3; ++i) q[15 + i] = ref[h].t[i];
            for (int i = 0; i < 6; ++i) { q[18 + i] = dLossH[i]; q[24 + i] = wv[(size_t)h * 6 + i]; }
            q[42] = maxJR;
            for (int i = 0; i < 3; ++i) q[50 + i] = rv[i];
        }
    }

    /* 5. expected loss; gradient of the soft-max selection (dsacstar_derivative.h:345-356) */
    double expected = 0.0;
    for (int h = 0; h < nHyp; ++h) expected += probs[h] * losses[h];
    *outLoss = expected;
    for (int i = 0; i < nHyp; ++i) {
        if (probs[i] < XO_PROB_THRESH) continue;
        double g = probs[i] * losses[i];
        for (int j = 0; j < nHyp; ++j) g -= probs[i] * probs[j] * losses[j];
        sog[i] = g;
    }

    /* 6. path II per-hypothesis quantities: g6 = sum_c dRepro(c) J_init(c), dPNP, support-point gradients */
    const float beta = 5.0f / thr;
    const float facf = alpha / (float)Wo / (float)Ho;
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < nHyp; ++h) {
        if (probs[h] < XO_PROB_THRESH) continue;
        xo_vec28 *part = (xo_vec28 *)malloc(sizeof(xo_vec28) * XO_T);
        for (int tdx = 0; tdx < XO_T; ++tdx) {
            double a[28];
            for (int k = 0; k < 28; ++k) a[k] = 0.0;
            for (int i = tdx; i < N; i += XO_T) {
                int y = i / Wo, x = i - y * Wo;
                double X[3], J6[6];
                xo_fetch(&co, x, y, X);
                float px = (float)(x * sub + sub / 2), py = (float)(y * sub + sub / 2);
                float e = xo_cell_err(&init[h], X, px, py, f, cx, cy, maxReproj);
                float stf = beta * (e - thr);
                double st = 1.0 / (1.0 + xo_exp(-(double)stf));
                double dRep = -st * (1.0 - st) * (double)beta * sog[h];
                dRep *= (double)facf;
                xo_resid_row(&init[h], dRinit + (size_t)h * 27, X, px, py, f, cx, cy, maxReproj, J6);
                for (int k = 0; k < 6; ++k) a[k] += dRep * J6[k];
            }
            for (int k = 0; k < 28; ++k) part[tdx].v[k] = a[k];
        }
        double g28[28];
        xo_reduce28(part, g28);
        free(part);
        float obj[4][3];
        double uv[4][2];
        for (int j = 0; j < 4; ++j) {
            int cidx = cells[4 * h + j];
            int y = cidx / Wo, x = cidx - y * Wo;
            const float *p = co.base + (int64_t)y * co.sy + (int64_t)x * co.sx;
            obj[j][0] = p[0]; obj[j][1] = p[co.sc]; obj[j][2] = p[2 * co.sc];
            uv[j][0] = (double)(float)(x * sub + sub / 2);
            uv[j][1] = (double)(float)(y * sub + sub / 2);
        }
        double J[72], maxH = 0.0;
        xo_dpnp(obj, uv, f, cx, cy, J);
        for (int k = 0; k < 72; ++k) { double v = fabs(J[k]); if (v > maxH) maxH = v; }
        if (maxH > 10.0) for (int k = 0; k < 72; ++k) J[k] = 0.0;      /* dsacstar_derivative.h:288 */
        for (int c = 0; c < 12; ++c) {
            double v = 0.0;
            for (int r = 0; r < 6; ++r) v += g28[r] * J[r * 12 + c];
            spg[(size_t)h * 12 + c] = v;
        }
        if (recs) {
            double *q = recs + (size_t)h * XO_BWD_REC;
            q[5] = sog[h]; q[43] = maxH;
            for (int c = 0; c < 12; ++c) q[30 + c] = spg[(size_t)h * 12 + c];
            for (int k = 0; k < 6; ++k) q[44 + k] = g28[k];
        }
    }

    /* 7. assemble: per cell, hypotheses in ascending order, float accumulation like the reference's tensor += */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        int y = i / Wo, x = i - y * Wo;
        double X[3];
        xo_fetch(&co, x, y, X);
        float px = (float)(x * sub + sub / 2), py = (float)(y * sub + sub / 2);
        float *g = grad + (int64_t)y * gsy + (int64_t)x * gsx;
        float acc[3] = { g[0], g[gsc], g[2 * gsc] };
        for (int h = 0; h < nHyp; ++h) {
            if (probs[h] < XO_PROB_THRESH) continue;
            double gI[3] = { 0.0, 0.0, 0.0 };
            if (nAcc[h] >= 4 && !clampI[h] && inlAcc[(size_t)h * N + i]) {
                double J6[6], dNdO[3];
                xo_resid_row(&ref[h], dRref + (size_t)h * 27, X, px, py, f, cx, cy, maxReproj, J6);
                double s = 0.0;
                for (int k = 0; k < 6; ++k) s += J6[k] * wv[(size_t)h * 6 + k];
                s = -s;
                xo_dproject_dobj(&ref[h], X, px, py, f, cx, cy, maxReproj, dNdO);
                for (int k = 0; k < 3; ++k) gI[k] = s * dNdO[k];
            }
            float e = xo_cell_err(&init[h], X, px, py, f, cx, cy, maxReproj);
            float stf = beta * (e - thr);
            double st = 1.0 / (1.0 + xo_exp(-(double)stf));
            double dRep = -st * (1.0 - st) * (double)beta * sog[h];
            dRep *= (double)facf;
            double dPdO[3];
            xo_dproject_dobj(&init[h], X, px, py, f, cx, cy, maxReproj, dPdO);
            double jac[3] = { dPdO[0] * dRep, dPdO[1] * dRep, dPdO[2] * dRep };
            for (int j = 0; j < 4; ++j)
                if (cells[4 * h + j] == i)
                    for (int k = 0; k < 3; ++k) jac[k] += spg[(size_t)h * 12 + 3 * j + k];
            for (int k = 0; k < 3; ++k)
                acc[k] = (float)((double)acc[k] + (probs[h] * gI[k] + jac[k]));
        }
        g[0] = acc[0]; g[gsc] = acc[1]; g[2 * gsc] = acc[2];
    }

    free(init); free(ref); free(scores); free(probs); free(losses); free(sog); free(cells); free(inlAcc); free(nAcc);
    free(dRinit); free(dRref); free(wv); free(spg); free(clampI);
    return 0;
}

/* --- unit-test hooks --- */
double xo_test_atan2(double y, double x) { return xo_atan2(y, x); }
void xo_test_log_so3(const double *R9, double *r3) { xo_log_so3(R9, r3); }
void xo_test_rodrigues_jac(const double *r3, double *dR27) { xo_rodrigues_jac(r3, dR27); }
void xo_test_pinv6(const double *A36, double *Ainv36) { xo_pinv6(A36, Ainv36); }
double xo_test_resid_row(const double *Rt12, const double *r3, const double *X3, float px, float py,
                         double f, double cx, double cy, float maxReproj, double *J6)
{
    xo_pose p;
    double dR[27];
    for (int i = 0; i < 9; ++i) p.R[i] = Rt12[i];
    for (int i = 0; i < 3; ++i) p.t[i] = Rt12[9 + i];
    xo_rodrigues_jac(r3, dR);
    return xo_resid_row(&p, dR, X3, px, py, f, cx, cy, maxReproj, J6);
}
void xo_test_dproject_dobj(const double *Rt12, const double *X3, float px, float py, double f, double cx, double cy,
                           float maxReproj, double *out3)
{
    xo_pose p;
    for (int i = 0; i < 9; ++i) p.R[i] = Rt12[i];
    for (int i = 0; i < 3; ++i) p.t[i] = Rt12[9 + i];
    xo_dproject_dobj(&p, X3, px, py, f, cx, cy, maxReproj, out3);
}
double xo_test_pose_loss(const double *Rt12, const float *gt16, double wRot, double wTrans, double cut)
{
    xo_pose p;
    xo_gt g;
    for (int i = 0; i < 9; ++i) p.R[i] = Rt12[i];
    for (int i = 0; i < 3; ++i) p.t[i] = Rt12[9 + i];
    xo_gt_from_pose16(gt16, &g);
    return xo_pose_loss(&p, &g, wRot, wTrans, cut);
}
void xo_test_dloss(const double *Rt12, const double *r3, const float *gt16, double wRot, double wTrans, double cut, double *jac6)
{
    xo_pose p;
    xo_gt g;
    double dR[27];
    for (int i = 0; i < 9; ++i) p.R[i] = Rt12[i];
    for (int i = 0; i < 3; ++i) p.t[i] = Rt12[9 + i];
    xo_gt_from_pose16(gt16, &g);
    xo_rodrigues_jac(r3, dR);
    xo_dloss(&p, dR, &g, wRot, wTrans, cut, jac6);
}
void xo_test_dpnp(const float *obj12, const double *uv8, double f, double cx, double cy, double *J72)
{
    float o[4][3];
    double uv[4][2];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 3; ++j) o[i][j] = obj12[3 * i + j];
        uv[i][0] = uv8[2 * i]; uv[i][1] = uv8[2 * i + 1];
    }
    xo_dpnp(o, uv, f, cx, cy, J72);
}
