// xl_loss.hip — fused forward+backward kernels of CrossLoc's per-pixel regression losses.
//
// Reference (PyTorch eager, ~40 elementwise launches and 5 host syncs per call):
//   loss/coord.py:87-188   scene_coords_regression_loss (+ :20-84 helpers, utils/learning.py:49-71)
//   loss/depth.py:7-76     depth_regression_loss
//   loss/normal.py:8-127   normal_regression_loss (+ utils/learning.py:401-440 angle helpers)
// Each loss is ONE streaming kernel here (HBM-bound: reads prediction + label once, writes the gradient once)
// plus a tiny finalisation kernel; per-cell loss terms and the analytic gradients of SURVEY.md Appendix B are
// produced together, sums are reduced in a fixed order (wave butterfly -> LDS -> per-block fp64 partials ->
// sequential finalise), and nothing synchronises with the host.
//
// Layout: NCHW as the network emits it; pred [B,C,N], sigma [B,N], labels [B,C,N], N = Ho*Wo, contiguous.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/crossloc_loss.h"
#include "../../include/crossloc_dsac.h"

namespace {

constexpr int kT = 256;
constexpr float kPi = 3.14159265358979323846f;

struct Sums { double a, b, c, d; };

// fixed-order block reduction of 4 doubles; result valid in thread 0
__device__ __forceinline__ Sums block_reduce4(Sums v, double *sm)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        v.a += __shfl_xor(v.a, off); v.b += __shfl_xor(v.b, off);
        v.c += __shfl_xor(v.c, off); v.d += __shfl_xor(v.d, off);
    }
    if (lane == 0) { sm[wave * 4] = v.a; sm[wave * 4 + 1] = v.b; sm[wave * 4 + 2] = v.c; sm[wave * 4 + 3] = v.d; }
    __syncthreads();
    Sums r{ sm[0], sm[1], sm[2], sm[3] };
#pragma unroll
    for (int w = 1; w < kT / 64; ++w) { r.a += sm[w * 4]; r.b += sm[w * 4 + 1]; r.c += sm[w * 4 + 2]; r.d += sm[w * 4 + 3]; }
    return r;
}

// inverse of a 4x4 (row-major float) -> first three rows as float (gt_poses.inverse()[:, 0:3, :], coord.py:29)
__device__ void inverse_rows3(const float *m, float *P /*12*/)
{
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = (double)m[i];
    double inv[16];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    const double id = 1.0 / det;
    for (int i = 0; i < 12; ++i) P[i] = (float)(inv[i] * id);
}

// ---------------------------------------------------------------------------------------------- coord

struct CoordArgs {
    const float *pred, *unc, *poses, *gt;
    float *dpred, *dunc;
    double *partials;
    int B, Ho, Wo, N, nblk, mode;
    float f, cx, cy, sub, minDepth, soft, hard, tol, nodata, gscale;
};

__global__ __launch_bounds__(kT)
void coord_loss_kernel(CoordArgs a)
{
    __shared__ float sP[12];
    __shared__ double sRed[16];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) inverse_rows3(a.poses + b * 16, sP);
    __syncthreads();
    const int i = blockIdx.x * kT + threadIdx.x;
    Sums s{ 0.0, 0.0, 0.0, 0.0 };
    if (i < a.N) {
        const long long base = (long long)b * 3 * a.N + i;
        const float X = a.pred[base], Y = a.pred[base + a.N], Z = a.pred[base + 2 * a.N];
        const float gx = a.gt[base], gy = a.gt[base + a.N], gz = a.gt[base + 2 * a.N];
        const bool g = (gx != a.nodata) && (gy != a.nodata) && (gz != a.nodata);             // learning.py:63
        // coords_world_to_cam (coord.py:20-38)
        const float xc = sP[0] * X + sP[1] * Y + sP[2] * Z + sP[3];
        const float yc = sP[4] * X + sP[5] * Y + sP[6] * Z + sP[7];
        const float zc = sP[8] * X + sP[9] * Y + sP[10] * Z + sP[11];
        const float xg = sP[0] * gx + sP[1] * gy + sP[2] * gz + sP[3];
        const float yg = sP[4] * gx + sP[5] * gy + sP[6] * gz + sP[7];
        const float zg = sP[8] * gx + sP[9] * gy + sP[10] * gz + sP[11];
        const float dx = xc - xg, dy = yc - yg, dz = zc - zg;
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);                                   // coord.py:120
        // get_repro_err (coord.py:41-57)
        const float px = a.f * xc + a.cx * zc, py = a.f * yc + a.cy * zc, pz = zc;
        const float zt = fmaxf(pz, a.minDepth);
        const float u = px / zt, v = py / zt;
        const int yy = i / a.Wo, xx = i - yy * a.Wo;
        const float ru = u - (xx * a.sub + a.sub * 0.5f), rv = v - (yy * a.sub + a.sub * 0.5f);
        const float rn = sqrtf(ru * ru + rv * rv);
        const float e = fmaxf(rn, 1e-7f);
        // check_constraints (coord.py:60-84): NOT coupled with valid_gt except for the tolerance test
        const bool m = !(zc < a.minDepth) && !(e > a.hard) && !((d > a.tol) && g);
        // reprojection term (coord.py:141-148)
        const float ep = m ? e : 0.f;
        const bool small = ep <= a.soft;
        const float l1 = fmaxf(small ? ep : 0.f, 1e-7f);
        const float lqIn = fmaxf(small ? 0.f : ep, 1e-7f);
        const float lq = fmaxf(sqrtf(a.soft * lqIn + 1e-7f), 1e-7f);
        const float lr = l1 + lq;
        float dLde = 0.f;
        if (m) {
            if (small) dLde = (ep >= 1e-7f) ? 1.f : 0.f;
            else dLde = a.soft / (2.f * sqrtf(a.soft * ep + 1e-7f));
        }
        // 3-D / uncertainty term (coord.py:152-167)
        float lu = 0.f, dLdd = 0.f, dLds = 0.f;
        if (a.mode == 1) {
            const float sraw = a.unc[(long long)b * a.N + i];
            const float sg = fmaxf(sraw, 1e-7f);
            const float d2 = d * d, d2c = fmaxf(d2, 1e-7f);
            const float s2 = sg * sg, s2c = fmaxf(s2, 1e-7f);
            if (g) {
                lu = 3.0f * logf(sg) + d2c / (2.0f * s2c);
                if (d2 >= 1e-7f) dLdd = d / s2c;
                if (sraw >= 1e-7f) {
                    dLds = 3.0f / sg;
                    if (s2 >= 1e-7f) dLds -= d2c / (s2c * sg);
                }
            }
            if (a.dunc) a.dunc[(long long)b * a.N + i] = dLds * a.gscale;
        } else {
            if (g) { lu = d; dLdd = 1.f; }
            if (a.dunc) a.dunc[(long long)b * a.N + i] = 0.f;
        }
        if (a.dpred) {
            // d e / d (u,v), then through the projection, K and the rigid transform
            float gu = 0.f, gv = 0.f;
            if (rn >= 1e-7f) { gu = dLde * ru / rn; gv = dLde * rv / rn; }
            const float gpx = gu / zt, gpy = gv / zt;
            float gpz = 0.f;
            if (pz >= a.minDepth) gpz = -(gu * px + gv * py) / (zt * zt);
            float gxc = a.f * gpx, gyc = a.f * gpy, gzc = a.cx * gpx + a.cy * gpy + gpz;
            if (d > 0.f) { const float k = dLdd / d; gxc += k * dx; gyc += k * dy; gzc += k * dz; }
            a.dpred[base] = (sP[0] * gxc + sP[4] * gyc + sP[8] * gzc) * a.gscale;
            a.dpred[base + a.N] = (sP[1] * gxc + sP[5] * gyc + sP[9] * gzc) * a.gscale;
            a.dpred[base + 2 * a.N] = (sP[2] * gxc + sP[6] * gyc + sP[10] * gzc) * a.gscale;
        }
        s.a = (double)lu; s.b = (double)lr; s.c = m ? 1.0 : 0.0; s.d = g ? 1.0 : 0.0;
    }
    const Sums r = block_reduce4(s, sRed);
    if (threadIdx.x == 0) {
        double *o = a.partials + ((long long)b * a.nblk + blockIdx.x) * 4;
        o[0] = r.a; o[1] = r.b; o[2] = r.c; o[3] = r.d;
    }
}

// ---------------------------------------------------------------------------------------------- depth

struct DepthArgs {
    const float *pred, *unc, *gt;
    float *dpred, *dunc;
    double *partials;
    int B, N, nblk, mode;
    float minDepth, hard, nodata, gscale;
};

__global__ __launch_bounds__(kT)
void depth_loss_kernel(DepthArgs a)
{
    __shared__ double sRed[16];
    const int b = blockIdx.y;
    const int i = blockIdx.x * kT + threadIdx.x;
    Sums s{ 0.0, 0.0, 0.0, 0.0 };
    if (i < a.N) {
        const long long idx = (long long)b * a.N + i;
        const float D = a.pred[idx], Dg = a.gt[idx];
        const bool g = Dg != a.nodata;
        const float diff = D - Dg;
        const float err = fabsf(diff);                                                          // depth.py:27
        const bool valid = !(D < a.minDepth) && !(err > a.hard) && g;                           // depth.py:34-40
        const float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
        float l = 0.f, dD = 0.f, dS = 0.f;
        if (a.mode == 1) {
            const float sraw = a.unc[idx];
            const float sg = fmaxf(sraw, 1e-7f);
            const float e2 = err * err, e2c = fmaxf(e2, 1e-7f);
            const float s2 = sg * sg, s2c = fmaxf(s2, 1e-7f);
            if (g) {
                l = logf(sg) + e2c / (2.0f * s2c);                                              // depth.py:51-54
                if (e2 >= 1e-7f) dD = err * sgn / s2c;
                if (sraw >= 1e-7f) {
                    dS = 1.0f / sg;
                    if (s2 >= 1e-7f) dS -= e2c / (s2c * sg);
                }
            }
        } else if (g) { l = err; dD = sgn; }
        if (a.dpred) a.dpred[idx] = dD * a.gscale;
        if (a.dunc) a.dunc[idx] = dS * a.gscale;
        s.a = (double)l; s.c = valid ? 1.0 : 0.0; s.d = g ? 1.0 : 0.0;
    }
    const Sums r = block_reduce4(s, sRed);
    if (threadIdx.x == 0) {
        double *o = a.partials + ((long long)b * a.nblk + blockIdx.x) * 4;
        o[0] = r.a; o[1] = r.b; o[2] = r.c; o[3] = r.d;
    }
}

// ---------------------------------------------------------------------------------------------- normal

struct NormalArgs {
    const float *logits, *unc, *gt;
    float *dlogits, *dunc;
    double *partials;
    int B, N, nblk, mode;
    float hard, nodata, gscale;
};

__device__ __forceinline__ float logit_to_rad(float l, float &dadl)
{
    // utils/learning.py:431-440: (2*clamp(sigmoid(l), 1e-7, 1-1e-7) - 1) * pi
    const float sg = 1.0f / (1.0f + expf(-l));
    const float lo = 1e-7f, hi = 1.0f - 1e-7f;
    const bool inside = (sg >= lo) && (sg <= hi);
    const float sc = fminf(fmaxf(sg, lo), hi);
    dadl = inside ? 2.0f * kPi * sg * (1.0f - sg) : 0.f;
    return (sc * 2.0f - 1.0f) * kPi;
}

__global__ __launch_bounds__(kT)
void normal_loss_kernel(NormalArgs a)
{
    __shared__ double sRed[16];
    const int b = blockIdx.y;
    const int i = blockIdx.x * kT + threadIdx.x;
    Sums s{ 0.0, 0.0, 0.0, 0.0 };
    if (i < a.N) {
        const long long b2 = (long long)b * 2 * a.N + i, b3 = (long long)b * 3 * a.N + i, b1 = (long long)b * a.N + i;
        float daz, del;
        const float az = logit_to_rad(a.logits[b2], daz);
        const float el = logit_to_rad(a.logits[b2 + a.N], del);
        const float gx = a.gt[b3], gy = a.gt[b3 + a.N], gz = a.gt[b3 + 2 * a.N];
        const bool g = (gx != a.nodata) && (gy != a.nodata) && (gz != a.nodata);
        const float azg = atan2f(gy, gx);                                                       // learning.py:409-413
        const float elg = atan2f(gz, sqrtf(gx * gx + gy * gy));
        // circular azimuth loss (normal.py:37-43)
        const float dlt = fabsf(azg - az);
        const float other = 2.0f * kPi - dlt;
        const bool first = dlt <= other;
        const float mn = first ? dlt : other;
        const float laz = 2.0f * fabsf(mn);
        const float lel = fabsf(el - elg);
        const float Eraw = laz + lel;
        const float E = fmaxf(Eraw, 1e-7f);
        // validity via angular error (normal.py:65-73); detached
        const float cxy = cosf(el);
        float nx = cosf(az) * cxy, ny = sinf(az) * cxy, nz = sinf(el);
        const float nn = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
        nx /= nn; ny /= nn; nz /= nn;
        const float n1 = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-8f);
        const float n2 = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-8f);
        float cs = (nx * gx + ny * gy + nz * gz) / (n1 * n2);
        cs = fminf(fmaxf(cs, -1.0f + 1e-7f), 1.0f - 1e-7f);
        const float ang = acosf(cs) / kPi * 180.0f;
        const bool valid = !(ang > a.hard) && g;
        // gradient of E wrt az / el
        float dEdaz = 0.f, dEdel = 0.f;
        if (Eraw >= 1e-7f) {
            const float sd = (azg - az > 0.f) ? 1.f : ((azg - az < 0.f) ? -1.f : 0.f);         // d|x|/dx
            const float smn = (mn > 0.f) ? 1.f : ((mn < 0.f) ? -1.f : 0.f);
            dEdaz = 2.0f * smn * (first ? 1.f : -1.f) * (-sd);
            dEdel = (el - elg > 0.f) ? 1.f : ((el - elg < 0.f) ? -1.f : 0.f);
        }
        float l = 0.f, dE = 0.f, dS = 0.f;
        if (a.mode == 1) {
            const float sraw = a.unc[b1];
            const float sg = fmaxf(sraw, 1e-7f);
            const float e2 = E * E, e2c = fmaxf(e2, 1e-7f);
            const float s2 = sg * sg, s2c = fmaxf(s2, 1e-7f);
            if (g) {
                l = 2.0f * logf(sg) + e2c / (2.0f * s2c);                                       // normal.py:103-105
                if (e2 >= 1e-7f) dE = E / s2c;
                if (sraw >= 1e-7f) {
                    dS = 2.0f / sg;
                    if (s2 >= 1e-7f) dS -= e2c / (s2c * sg);
                }
            }
        } else if (g) { l = E; dE = 1.f; }
        if (a.dlogits) {
            a.dlogits[b2] = dE * dEdaz * daz * a.gscale;
            a.dlogits[b2 + a.N] = dE * dEdel * del * a.gscale;
        }
        if (a.dunc) a.dunc[b1] = dS * a.gscale;
        s.a = (double)l; s.c = valid ? 1.0 : 0.0; s.d = g ? 1.0 : 0.0;
    }
    const Sums r = block_reduce4(s, sRed);
    if (threadIdx.x == 0) {
        double *o = a.partials + ((long long)b * a.nblk + blockIdx.x) * 4;
        o[0] = r.a; o[1] = r.b; o[2] = r.c; o[3] = r.d;
    }
}

// ---------------------------------------------------------------------------------------------- semantics

// CrossEntropyLoss2d (loss/semantics.py:10-18) inside semantics_classification_loss (:44-91): per pixel
// -log softmax(logits)[label]; the "valid" count is the number of pixels whose arg-max class (first maximum, like
// torch.argmax) equals the label.  Gradient: (softmax - onehot) * scale.  logits NCHW [B,C,N], labels [B,N] as floats.
struct SemArgs {
    const float *logits, *labels;
    float *dlogits;
    double *partials;
    int B, C, N, nblk;
    float gscale;
};

__global__ __launch_bounds__(kT)
void semantics_loss_kernel(SemArgs a)
{
    __shared__ double sRed[16];
    const int b = blockIdx.y;
    const int i = blockIdx.x * kT + threadIdx.x;
    Sums s{ 0.0, 0.0, 0.0, 0.0 };
    if (i < a.N) {
        const float *l = a.logits + (long long)b * a.C * a.N + i;
        const int lab = (int)a.labels[(long long)b * a.N + i];
        float m = l[0];
        int arg = 0;
        for (int c = 1; c < a.C; ++c) {
            const float v = l[(long long)c * a.N];
            if (v > m) { m = v; arg = c; }
        }
        float se = 0.f;
        for (int c = 0; c < a.C; ++c) se += expf(l[(long long)c * a.N] - m);
        const float lse = logf(se);
        const float xl = (lab >= 0 && lab < a.C) ? l[(long long)lab * a.N] : m;
        const float loss = (lab >= 0 && lab < a.C) ? -((xl - m) - lse) : 0.f;
        if (a.dlogits) {
            float *d = a.dlogits + (long long)b * a.C * a.N + i;
            const float inv = 1.0f / se;
            for (int c = 0; c < a.C; ++c) {
                float g = expf(l[(long long)c * a.N] - m) * inv;
                if (c == lab) g -= 1.0f;
                d[(long long)c * a.N] = (lab >= 0 && lab < a.C) ? g * a.gscale : 0.f;
            }
        }
        s.a = (double)loss; s.c = (arg == lab) ? 1.0 : 0.0; s.d = 1.0;
    }
    const Sums r = block_reduce4(s, sRed);
    if (threadIdx.x == 0) {
        double *o = a.partials + ((long long)b * a.nblk + blockIdx.x) * 4;
        o[0] = r.a; o[1] = r.b; o[2] = r.c; o[3] = r.d;
    }
}

// ---------------------------------------------------------------------------------------------- finalise

// out[0] = loss (mean over B*N), out[1] = valid rate, out[2..2+B) = per-image mean loss (reduction=None).
// `gate`: 1 -> the sum in slot b only counts if any cell of the batch is valid (coord.py:141).
// One workgroup of 256 threads: per image, thread t sums the block partials t, t+256, ... and a fixed-order tree
// finishes (a single thread walking B*nblk partials is a serial chain of load latencies: 21600 of them for a
// full-resolution semantics batch).
__global__ __launch_bounds__(256)
void finalize_kernel(const double *partials, int B, int nblk, int N, int gate, float *out)
{
    __shared__ double sR[3][256];
    __shared__ double sImgA[64], sImgB[64];           // per-image sums when B <= 64 (else recomputed below)
    const int tid = threadIdx.x;
    double totA = 0.0, totB = 0.0, totC = 0.0;
    for (int b = 0; b < B; ++b) {
        double sa = 0.0, sb = 0.0, sc = 0.0;
        for (int k = tid; k < nblk; k += 256) {
            const double *p = partials + ((long long)b * nblk + k) * 4;
            sa += p[0]; sb += p[1]; sc += p[2];
        }
        sR[0][tid] = sa; sR[1][tid] = sb; sR[2][tid] = sc;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) { sR[0][tid] += sR[0][tid + w]; sR[1][tid] += sR[1][tid + w]; sR[2][tid] += sR[2][tid + w]; }
            __syncthreads();
        }
        if (tid == 0) {
            totA += sR[0][0]; totB += sR[1][0]; totC += sR[2][0];
            if (b < 64) { sImgA[b] = sR[0][0]; sImgB[b] = sR[1][0]; }
        }
        __syncthreads();
    }
    if (tid != 0) return;
    const bool useB = !gate || totC > 0.0;
    for (int b = 0; b < B && b < 64; ++b) out[2 + b] = (float)((sImgA[b] + (useB ? sImgB[b] : 0.0)) / (double)N);
    for (int b = 64; b < B; ++b) {                    // beyond the LDS table: recompute serially (not a hot case)
        double sa = 0.0, sb = 0.0;
        for (int k = 0; k < nblk; ++k) { const double *p = partials + ((long long)b * nblk + k) * 4; sa += p[0]; sb += p[1]; }
        out[2 + b] = (float)((sa + (useB ? sb : 0.0)) / (double)N);
    }
    out[0] = (float)((totA + (useB ? totB : 0.0)) / ((double)B * (double)N));
    out[1] = (float)(totC / ((double)B * (double)N));
}

int launch_ok()
{
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}

}  // namespace

extern "C" {

int xl_loss_workspace_doubles(int B, int Ho, int Wo)
{
    const int N = Ho * Wo;
    return B * ((N + kT - 1) / kT) * 4;
}

int xl_loss_coord(const float *pred, const float *unc, const float *gt_poses, const float *gt_coords,
                  int B, int Ho, int Wo, float focal, float cx, float cy, float subsample,
                  float min_depth, float soft_clamp, float hard_clamp, float init_tolerance, float nodata,
                  int mode, int per_image_scale, float *dpred, float *dunc, double *workspace, float *out, void *stream)
{
    if (!pred || !gt_poses || !gt_coords || !workspace || !out || B <= 0 || Ho <= 0 || Wo <= 0) return XL_ERR_ARG;
    if (mode == 1 && !unc) return XL_ERR_ARG;
    CoordArgs a;
    a.pred = pred; a.unc = unc; a.poses = gt_poses; a.gt = gt_coords; a.dpred = dpred; a.dunc = dunc;
    a.partials = workspace; a.B = B; a.Ho = Ho; a.Wo = Wo; a.N = Ho * Wo; a.nblk = (a.N + kT - 1) / kT; a.mode = mode;
    a.f = focal; a.cx = cx; a.cy = cy; a.sub = subsample; a.minDepth = min_depth; a.soft = soft_clamp;
    a.hard = hard_clamp; a.tol = init_tolerance; a.nodata = nodata;
    a.gscale = per_image_scale ? 1.0f / (float)a.N : 1.0f / ((float)B * (float)a.N);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(coord_loss_kernel, dim3(a.nblk, B), dim3(kT), 0, st, a);
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, workspace, B, a.nblk, a.N, 1, out);
    return launch_ok();
}

int xl_loss_depth(const float *pred, const float *unc, const float *gt_depth, int B, int Ho, int Wo,
                  float min_depth, float hard_clamp, float nodata, int mode, int per_image_scale,
                  float *dpred, float *dunc, double *workspace, float *out, void *stream)
{
    if (!pred || !gt_depth || !workspace || !out || B <= 0 || Ho <= 0 || Wo <= 0) return XL_ERR_ARG;
    if (mode == 1 && !unc) return XL_ERR_ARG;
    DepthArgs a;
    a.pred = pred; a.unc = unc; a.gt = gt_depth; a.dpred = dpred; a.dunc = dunc; a.partials = workspace;
    a.B = B; a.N = Ho * Wo; a.nblk = (a.N + kT - 1) / kT; a.mode = mode;
    a.minDepth = min_depth; a.hard = hard_clamp; a.nodata = nodata;
    a.gscale = per_image_scale ? 1.0f / (float)a.N : 1.0f / ((float)B * (float)a.N);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(depth_loss_kernel, dim3(a.nblk, B), dim3(kT), 0, st, a);
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, workspace, B, a.nblk, a.N, 0, out);
    return launch_ok();
}

int xl_loss_normal(const float *logits, const float *unc, const float *gt_normals, int B, int Ho, int Wo,
                   float hard_clamp, float nodata, int mode, int per_image_scale,
                   float *dlogits, float *dunc, double *workspace, float *out, void *stream)
{
    if (!logits || !gt_normals || !workspace || !out || B <= 0 || Ho <= 0 || Wo <= 0) return XL_ERR_ARG;
    if (mode == 1 && !unc) return XL_ERR_ARG;
    NormalArgs a;
    a.logits = logits; a.unc = unc; a.gt = gt_normals; a.dlogits = dlogits; a.dunc = dunc; a.partials = workspace;
    a.B = B; a.N = Ho * Wo; a.nblk = (a.N + kT - 1) / kT; a.mode = mode; a.hard = hard_clamp; a.nodata = nodata;
    a.gscale = per_image_scale ? 1.0f / (float)a.N : 1.0f / ((float)B * (float)a.N);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(normal_loss_kernel, dim3(a.nblk, B), dim3(kT), 0, st, a);
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, workspace, B, a.nblk, a.N, 0, out);
    return launch_ok();
}

int xl_loss_semantics(const float *logits, const float *labels, int B, int C, int H, int W, int per_image_scale,
                      float *dlogits, double *workspace, float *out, void *stream)
{
    if (!logits || !labels || !workspace || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return XL_ERR_ARG;
    SemArgs a;
    a.logits = logits; a.labels = labels; a.dlogits = dlogits; a.partials = workspace;
    a.B = B; a.C = C; a.N = H * W; a.nblk = (a.N + kT - 1) / kT;
    a.gscale = per_image_scale ? 1.0f / (float)a.N : 1.0f / ((float)B * (float)a.N);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(semantics_loss_kernel, dim3(a.nblk, B), dim3(kT), 0, st, a);
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, workspace, B, a.nblk, a.N, 0, out);
    return launch_ok();
}

}  // extern "C"
