// xl_data.hip — GPU-side frame / label preparation (include/crossloc_data.h): the reference's CPU-worker transform
// pipelines (dataloader/dataloader.py:189-232, 349-393) and its batch_resize collate (:512-563) as streaming kernels.
// Integer / byte arithmetic is reproduced exactly (Pillow's fixed-point resampler and uint8 blends); the float steps use
// the operation order of the torch ops they replace.  HBM-bound elementwise passes: one thread per output pixel, the
// three channels of a pixel together, consecutive lanes on consecutive pixels.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/crossloc_data.h"
#include "../../include/crossloc_dsac.h"   // status codes

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;          // Pillow: PRECISION_BITS

// ---------------------------------------------------------------------------------------------- resampling tables

struct Table { int32_t *bounds; int32_t *coef; int ksize; };   // device: bounds [out][2], coef [out][ksize]

// Pillow precompute_coeffs() + normalize_coeffs_8bpc() for the triangle filter over a whole axis
void host_coefficients(int inSize, int outSize, std::vector<int32_t> &bounds, std::vector<int32_t> &kk, int &ksize)
{
    const double scale = (double)inSize / (double)outSize;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)outSize * 2, 0);
    kk.assign((size_t)outSize * ksize, 0);
    const double ss = 1.0 / filterscale;
    std::vector<double> k((size_t)ksize);
    for (int xx = 0; xx < outSize; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > inSize) xmax = inSize;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            const double w = t < 1.0 ? 1.0 - t : 0.0;
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? k[x] / ww : k[x];
            kk[(size_t)xx * ksize + x] = v < 0 ? (int32_t)(-0.5 + v * (1 << kPrecisionBits)) : (int32_t)(0.5 + v * (1 << kPrecisionBits));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
}

std::mutex g_tableMutex;
std::map<std::tuple<int, int, int>, Table> g_tables;            // (device, in, out) -> device table, built once

int get_table(int inSize, int outSize, Table &t)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return XL_ERR_HIP;
    std::lock_guard<std::mutex> lock(g_tableMutex);
    auto key = std::make_tuple(dev, inSize, outSize);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) { t = it->second; return XL_OK; }
    std::vector<int32_t> b, k;
    int ksize = 0;
    host_coefficients(inSize, outSize, b, k, ksize);
    Table nt;
    nt.ksize = ksize;
    if (hipMalloc(&nt.bounds, b.size() * sizeof(int32_t)) != hipSuccess) return XL_ERR_HIP;
    if (hipMalloc(&nt.coef, k.size() * sizeof(int32_t)) != hipSuccess) return XL_ERR_HIP;
    if (hipMemcpy(nt.bounds, b.data(), b.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) return XL_ERR_HIP;
    if (hipMemcpy(nt.coef, k.data(), k.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) return XL_ERR_HIP;
    g_tables[key] = nt;
    t = nt;
    return XL_OK;
}

__device__ __forceinline__ uint8_t clip8(int ss)
{
    const int v = ss >> kPrecisionBits;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [B][Hs][Ws][Cs] -> dst [B][Hs][W][3]
__global__ __launch_bounds__(256)
void resample_h_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, long long rows, int Ws, int Cs, int W,
                       const int32_t *__restrict__ bounds, const int32_t *__restrict__ coef, int ksize)
{
    const long long total = rows * W;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
        const long long row = p / W;
        const int xx = (int)(p - row * W);
        const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
        const uint8_t *s = src + (row * Ws + x0) * Cs;
        int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
        for (int x = 0; x < n; ++x) {
            const int k = coef[xx * ksize + x];
            a0 += (int)s[x * Cs] * k; a1 += (int)s[x * Cs + 1] * k; a2 += (int)s[x * Cs + 2] * k;
        }
        uint8_t *o = dst + p * 3;
        o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
    }
}

// vertical pass: src [B][Hs][W][Cs] -> dst [B][H][W][3]
__global__ __launch_bounds__(256)
void resample_v_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int B, int Hs, int W, int Cs, int H,
                       const int32_t *__restrict__ bounds, const int32_t *__restrict__ coef, int ksize)
{
    const long long total = (long long)B * H * W;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
        const int x = (int)(p % W);
        const long long r = p / W;
        const int yy = (int)(r % H);
        const int b = (int)(r / H);
        const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
        const uint8_t *s = src + (((long long)b * Hs + y0) * W + x) * Cs;
        int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
        for (int y = 0; y < n; ++y) {
            const int k = coef[yy * ksize + y];
            const uint8_t *q = s + (long long)y * W * Cs;
            a0 += (int)q[0] * k; a1 += (int)q[1] * k; a2 += (int)q[2] * k;
        }
        uint8_t *o = dst + p * 3;
        o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
    }
}

// ---------------------------------------------------------------------------------------------- jitter + ToTensor

// PIL.Image.blend(degenerate, image, alpha) on uint8: truncation inside [0,1], clip then truncation outside
__device__ __forceinline__ int blend_u8(int degenerate, int v, float alpha, bool interp)
{
    const float t = (float)degenerate + alpha * ((float)v - (float)degenerate);
    if (interp) return (int)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

struct Jitter { float brightness, contrast; int contrastFirst, active; };
// the per-frame jitter records of one call travel as a by-value kernel argument (1 KB for 64 frames): no device table to
// fill, no host synchronisation - the call stays asynchronous on its stream.  Larger mini-batches run in chunks of 64.
constexpr int kJitMax = 64;
struct JitterPack { Jitter j[kJitMax]; };

// per-image sum of the 'L' conversion of the image the contrast step sees (the brightened one when brightness runs first)
__global__ __launch_bounds__(256)
void gray_sum_kernel(const uint8_t *__restrict__ img, int Cs, long long pixels, JitterPack jit,
                     unsigned long long *__restrict__ sums, int gray)
{
    const int b = blockIdx.y;
    const Jitter j = jit.j[b];
    if (!j.active) return;
    const bool bInterp = j.brightness >= 0.f && j.brightness <= 1.f;
    unsigned long long acc = 0;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256) {
        const uint8_t *s = img + ((long long)b * pixels + p) * Cs;
        int r = s[0], g = s[1], bl = s[2];
        if (gray) {                                   // Grayscale() runs before ColorJitter: the image IS its 'L' conversion
            int l = luma(r, g, bl);
            if (!j.contrastFirst) l = blend_u8(0, l, j.brightness, bInterp);
            acc += (unsigned long long)l;
            continue;
        }
        if (!j.contrastFirst) {
            r = blend_u8(0, r, j.brightness, bInterp); g = blend_u8(0, g, j.brightness, bInterp); bl = blend_u8(0, bl, j.brightness, bInterp);
        }
        acc += (unsigned long long)luma(r, g, bl);
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[b], acc);         // integer sum: order-independent
}

__global__ __launch_bounds__(256)
void jitter_to_tensor_kernel(const uint8_t *__restrict__ img, int Cs, long long pixels, JitterPack jit,
                             const unsigned long long *__restrict__ sums, float m0, float m1, float m2, float s0, float s1,
                             float s2, int normalize, float *__restrict__ out, int gray)
{
    const int b = blockIdx.y;
    const Jitter j = jit.j[b];
    int mean = 0;
    if (j.active) mean = (int)((double)sums[b] / (double)pixels + 0.5);       // int(ImageStat mean + 0.5)
    const bool bInterp = j.brightness >= 0.f && j.brightness <= 1.f, cInterp = j.contrast >= 0.f && j.contrast <= 1.f;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256) {
        const uint8_t *s = img + ((long long)b * pixels + p) * Cs;
        int v[3] = { s[0], s[1], s[2] };
        if (gray) {                                   // transforms.Grayscale(): one channel, (x - mean[0]) / std[0]
            int l = luma(v[0], v[1], v[2]);
            if (j.active) {
                if (j.contrastFirst) l = blend_u8(0, blend_u8(mean, l, j.contrast, cInterp), j.brightness, bInterp);
                else l = blend_u8(mean, blend_u8(0, l, j.brightness, bInterp), j.contrast, cInterp);
            }
            float f = (float)l / 255.f;
            if (normalize) f = (f - m0) / s0;
            out[(long long)b * pixels + p] = f;
            continue;
        }
        if (j.active) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (j.contrastFirst) v[c] = blend_u8(0, blend_u8(mean, v[c], j.contrast, cInterp), j.brightness, bInterp);
                else v[c] = blend_u8(mean, blend_u8(0, v[c], j.brightness, bInterp), j.contrast, cInterp);
            }
        }
        float f[3] = { (float)v[0] / 255.f, (float)v[1] / 255.f, (float)v[2] / 255.f };
        if (normalize) { f[0] = (f[0] - m0) / s0; f[1] = (f[1] - m1) / s1; f[2] = (f[2] - m2) / s2; }
        float *o = out + (long long)b * 3 * pixels + p;
        o[0] = f[0]; o[pixels] = f[1]; o[2 * pixels] = f[2];
    }
}

// ---------------------------------------------------------------------------------------------- batch_resize

struct AugArgs {
    const float *in; float *out;
    int B, C, H, W, oh, ow, bilinear;
    float r00, r10, r01, r11;        // rescaled inverse rotation (torchvision `rescaled_theta`), float32
    float scaleH, scaleW;            // (float)in / out of the resize
    float fill;
};

__global__ __launch_bounds__(256)
void batch_augment_kernel(AugArgs a)
{
    const long long plane = (long long)a.oh * a.ow;
    const long long total = (long long)a.B * plane;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
        const int b = (int)(p / plane);
        const int rem = (int)(p - (long long)b * plane);
        const int i = rem / a.ow, jx = rem - i * a.ow;
        // rotation: source pixel of the resized frame (grid_sample nearest, align_corners = false)
        const float X = (float)jx + 0.5f - (float)a.ow * 0.5f, Y = (float)i + 0.5f - (float)a.oh * 0.5f;
        const float gx = X * a.r00 + Y * a.r10, gy = X * a.r01 + Y * a.r11;
        const float sx = ((gx + 1.f) * (float)a.ow - 1.f) / 2.f, sy = ((gy + 1.f) * (float)a.oh - 1.f) / 2.f;
        const int ix = (int)rintf(sx), iy = (int)rintf(sy);
        const bool inside = ix >= 0 && ix < a.ow && iy >= 0 && iy < a.oh;
        const float *src = a.in + (long long)b * a.C * a.H * a.W;
        float *dst = a.out + (long long)b * a.C * plane + rem;
        if (!inside) {
            for (int c = 0; c < a.C; ++c) dst[c * plane] = a.fill;
            continue;
        }
        if (a.bilinear) {                                 // upsample_bilinear2d, align_corners = false
            float fy = a.scaleH * ((float)iy + 0.5f) - 0.5f, fx = a.scaleW * ((float)ix + 0.5f) - 0.5f;
            if (fy < 0.f) fy = 0.f;
            if (fx < 0.f) fx = 0.f;
            const int y0 = (int)fy, x0 = (int)fx;
            const int yp = y0 < a.H - 1 ? 1 : 0, xp = x0 < a.W - 1 ? 1 : 0;
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const float hy = 1.f - ly, hx = 1.f - lx;
            for (int c = 0; c < a.C; ++c) {
                const float *q = src + ((long long)c * a.H + y0) * a.W + x0;
                dst[c * plane] = hy * (hx * q[0] + lx * q[xp]) + ly * (hx * q[(long long)yp * a.W] + lx * q[(long long)yp * a.W + xp]);
            }
        } else {                                          // F.interpolate 'nearest': floor(dst * scale)
            int y0 = (int)floorf((float)iy * a.scaleH), x0 = (int)floorf((float)ix * a.scaleW);
            if (y0 > a.H - 1) y0 = a.H - 1;
            if (x0 > a.W - 1) x0 = a.W - 1;
            for (int c = 0; c < a.C; ++c) dst[c * plane] = src[((long long)c * a.H + y0) * a.W + x0];
        }
    }
}

unsigned grid_for(long long items)
{
    long long blocks = (items + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

extern "C" {

int xl_data_resized_shape(int Hs, int Ws, int image_height, int *H, int *W)
{
    if (Hs < 1 || Ws < 1 || image_height < 1 || !H || !W) return XL_ERR_ARG;
    if ((Ws <= Hs && Ws == image_height) || (Hs <= Ws && Hs == image_height)) { *H = Hs; *W = Ws; }
    else if (Ws < Hs) { *H = (int)((long long)image_height * Hs / Ws); *W = image_height; }
    else { *H = image_height; *W = (int)((long long)image_height * Ws / Hs); }
    return XL_OK;
}

long long xl_data_prepare_workspace_bytes(int B, int Hs, int Ws, int H, int W)
{
    // horizontal-pass result, resized frame, per-image jitter records, per-image luma sums (each 256-byte aligned)
    auto al = [](long long v) { return (v + 255) / 256 * 256; };
    return al((long long)B * Hs * W * 3) + al((long long)B * H * W * 3) + al((long long)B * sizeof(Jitter)) +
           al((long long)B * sizeof(unsigned long long));
}

static int prepare_images(const uint8_t *src, int B, int Hs, int Ws, int Cs, int image_height,
                          const float *jitter_host, const float *mean_host, const float *std_host,
                          float *out, void *workspace, void *stream, int gray)
{
    if (!src || !out || !workspace || B < 1 || (Cs != 3 && Cs != 4) || (mean_host == nullptr) != (std_host == nullptr)) return XL_ERR_ARG;
    int H = 0, W = 0;
    if (xl_data_resized_shape(Hs, Ws, image_height, &H, &W) != XL_OK) return XL_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    auto al = [](long long v) { return (v + 255) / 256 * 256; };
    uint8_t *tmp = (uint8_t *)workspace;
    uint8_t *res = tmp + al((long long)B * Hs * W * 3);
    Jitter *jit = (Jitter *)(res + al((long long)B * H * W * 3));
    unsigned long long *sums = (unsigned long long *)((uint8_t *)jit + al((long long)B * sizeof(Jitter)));

    const uint8_t *cur = src;
    int curC = Cs;
    if (Ws != W) {
        Table t;
        const int rc = get_table(Ws, W, t);
        if (rc != XL_OK) return rc;
        hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for((long long)B * Hs * W)), dim3(256), 0, st, cur, tmp,
                           (long long)B * Hs, Ws, curC, W, t.bounds, t.coef, t.ksize);
        cur = tmp; curC = 3;
    }
    if (Hs != H) {
        Table t;
        const int rc = get_table(Hs, H, t);
        if (rc != XL_OK) return rc;
        hipLaunchKernelGGL(resample_v_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, st, cur, res, B, Hs, W, curC,
                           H, t.bounds, t.coef, t.ksize);
        cur = res; curC = 3;
    }
    (void)jit;
    const long long pixels = (long long)H * W;
    const int nm = gray ? 1 : 3;                                       // entries of mean_host / std_host
    const float m0 = mean_host ? mean_host[0] : 0.f, m1 = (mean_host && nm > 1) ? mean_host[1] : 0.f, m2 = (mean_host && nm > 2) ? mean_host[2] : 0.f;
    const float s0 = std_host ? std_host[0] : 1.f, s1 = (std_host && nm > 1) ? std_host[1] : 1.f, s2 = (std_host && nm > 2) ? std_host[2] : 1.f;
    if (jitter_host && hipMemsetAsync(sums, 0, sizeof(unsigned long long) * B, st) != hipSuccess) return XL_ERR_HIP;
    for (int b0 = 0; b0 < B; b0 += kJitMax) {                          // the jitter records ride in the kernel arguments
        const int nb = B - b0 < kJitMax ? B - b0 : kJitMax;
        JitterPack pack;
        for (int b = 0; b < kJitMax; ++b) {
            const bool on = jitter_host && b < nb;
            pack.j[b].active = on ? 1 : 0;
            pack.j[b].brightness = on ? jitter_host[3 * (b0 + b)] : 1.f;
            pack.j[b].contrast = on ? jitter_host[3 * (b0 + b) + 1] : 1.f;
            pack.j[b].contrastFirst = on ? (jitter_host[3 * (b0 + b) + 2] != 0.f) : 0;
        }
        const uint8_t *img = cur + (long long)b0 * pixels * curC;
        if (jitter_host) {
            unsigned gx = grid_for(pixels);
            if (gx > 1024) gx = 1024;
            hipLaunchKernelGGL(gray_sum_kernel, dim3(gx, nb), dim3(256), 0, st, img, curC, pixels, pack, sums + b0, gray);
        }
        unsigned gx = grid_for(pixels);
        if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(jitter_to_tensor_kernel, dim3(gx, nb), dim3(256), 0, st, img, curC, pixels, pack, sums + b0, m0, m1, m2,
                           s0, s1, s2, mean_host ? 1 : 0, out + (long long)b0 * (gray ? 1 : 3) * pixels, gray);
    }
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}

int xl_data_prepare_images(const uint8_t *src, int B, int Hs, int Ws, int Cs, int image_height,
                           const float *jitter_host, const float *mean_host, const float *std_host,
                           float *out, void *workspace, void *stream)
{
    return prepare_images(src, B, Hs, Ws, Cs, image_height, jitter_host, mean_host, std_host, out, workspace, stream, 0);
}

int xl_data_prepare_images_gray(const uint8_t *src, int B, int Hs, int Ws, int Cs, int image_height,
                                const float *jitter_host, const float *mean_host, const float *std_host,
                                float *out, void *workspace, void *stream)
{
    return prepare_images(src, B, Hs, Ws, Cs, image_height, jitter_host, mean_host, std_host, out, workspace, stream, 1);
}

int xl_data_batch_augment(const float *in, float *out, int B, int C, int H, int W, int oh, int ow,
                          double angle_deg, float fill, int bilinear, void *stream)
{
    if (!in || !out || B < 1 || C < 1 || H < 1 || W < 1 || oh < 1 || ow < 1) return XL_ERR_ARG;
    AugArgs a;
    a.in = in; a.out = out; a.B = B; a.C = C; a.H = H; a.W = W; a.oh = oh; a.ow = ow; a.bilinear = bilinear; a.fill = fill;
    const double th = angle_deg * (3.14159265358979323846 / 180.0);    // math.radians: x * (pi / 180)
    // torchvision: inverse affine matrix of -angle as float32, divided by the half sizes in float32
    a.r00 = (float)cos(th) / (0.5f * (float)ow);
    a.r10 = (float)(-sin(th)) / (0.5f * (float)ow);
    a.r01 = (float)sin(th) / (0.5f * (float)oh);
    a.r11 = (float)cos(th) / (0.5f * (float)oh);
    a.scaleH = (float)H / (float)oh;
    a.scaleW = (float)W / (float)ow;
    hipLaunchKernelGGL(batch_augment_kernel, dim3(grid_for((long long)B * oh * ow)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? XL_OK : XL_ERR_HIP;
}

}  // extern "C"
