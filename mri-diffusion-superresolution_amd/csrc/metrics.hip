// Image metrics of the reference's evaluator (src/eval/eval.py:15-51) on the device: PSNR and SSIM
// with torchmetrics' defaults (data_range 1; 11x11 Gaussian, sigma 1.5, k1 .01, k2 .03, mean over the fully-inside windows),
// HFEN (Laplacian of a sigma-1.5 Gaussian, skimage / scipy boundary rules) and NMSE.  Bandwidth-trivial kernels; sums are
// accumulated in double precision.
#include "common.h"
#include "prof.h"

namespace mrisr {

__device__ __forceinline__ double warp_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ void block_atomic_add(double* dst, double v) {
    __shared__ double red[4];
    v = warp_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dst, red[0] + red[1] + red[2] + red[3]);
}

// sums[b][0] = sum (p - t)^2, sums[b][1] = sum t^2
__global__ __launch_bounds__(256) void metric_sq_kernel(const float* __restrict__ p, const float* __restrict__ t, double* sums, int HW) {
    const int b = blockIdx.y;
    double e = 0.0, n = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const double pv = p[(size_t)b * HW + i], tv = t[(size_t)b * HW + i];
        e += (pv - tv) * (pv - tv);
        n += tv * tv;
    }
    block_atomic_add(&sums[b * 6 + 0], e);
    block_atomic_add(&sums[b * 6 + 1], n);
}

// SSIM map summed over the (H-10) x (W-10) fully-inside window positions -> sums[b][2]
__global__ __launch_bounds__(256) void metric_ssim_kernel(const float* __restrict__ p, const float* __restrict__ t, double* sums, int H,
                                                          int W) {
    __shared__ float sp[26][27], st[26][27];
    __shared__ float gw[11];
    const int b = blockIdx.z;
    const int ox0 = blockIdx.x * 16, oy0 = blockIdx.y * 16;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    if (threadIdx.x < 11) {
        float s = 0.f;
        for (int k = 0; k < 11; ++k) s += expf(-0.5f * ((k - 5) / 1.5f) * ((k - 5) / 1.5f));
        gw[threadIdx.x] = expf(-0.5f * ((threadIdx.x - 5.f) / 1.5f) * ((threadIdx.x - 5.f) / 1.5f)) / s;
    }
    const float* pb = p + (size_t)b * H * W;
    const float* tb = t + (size_t)b * H * W;
    for (int i = threadIdx.x; i < 26 * 26; i += 256) {
        const int y = i / 26, x = i - y * 26;
        const int iy = min(oy0 + y, H - 1), ix = min(ox0 + x, W - 1);  // clamped reads are never used by a valid output
        sp[y][x] = pb[(size_t)iy * W + ix];
        st[y][x] = tb[(size_t)iy * W + ix];
    }
    __syncthreads();
    double acc = 0.0;
    const int ox = ox0 + tx, oy = oy0 + ty;
    if (ox < W - 10 && oy < H - 10) {
        float mx = 0.f, my = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
        for (int ky = 0; ky < 11; ++ky) {
            float rx = 0.f, ry = 0.f, rxx = 0.f, ryy = 0.f, rxy = 0.f;
#pragma unroll
            for (int kx = 0; kx < 11; ++kx) {
                const float a = sp[ty + ky][tx + kx], c = st[ty + ky][tx + kx], g = gw[kx];
                rx += g * a; ry += g * c; rxx += g * a * a; ryy += g * c * c; rxy += g * a * c;
            }
            const float g = gw[ky];
            mx += g * rx; my += g * ry; xx += g * rxx; yy += g * ryy; xy += g * rxy;
        }
        const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
        const float sxx = xx - mx * mx, syy = yy - my * my, sxy = xy - mx * my;
        acc = (double)(((2.f * mx * my + c1) * (2.f * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2)));
    }
    block_atomic_add(&sums[b * 6 + 2], acc);
}

// sigma-1.5 Gaussian (radius 6, edge values repeated: scipy mode "nearest") of (p - t) and of t -> scratch[0], scratch[1]
__global__ __launch_bounds__(256) void metric_gauss_kernel(const float* __restrict__ p, const float* __restrict__ t, float* sd, float* stt, int H,
                                                           int W) {
    __shared__ float gw[13];
    if (threadIdx.x < 13) {
        float s = 0.f;
        for (int k = 0; k < 13; ++k) s += expf(-0.5f * ((k - 6) / 1.5f) * ((k - 6) / 1.5f));
        gw[threadIdx.x] = expf(-0.5f * ((threadIdx.x - 6.f) / 1.5f) * ((threadIdx.x - 6.f) / 1.5f)) / s;
    }
    __syncthreads();
    const int b = blockIdx.y;
    const size_t base = (size_t)b * H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        float ad = 0.f, at = 0.f;
        for (int ky = -6; ky <= 6; ++ky) {
            const int iy = min(max(y + ky, 0), H - 1);
            float rd = 0.f, rt = 0.f;
            for (int kx = -6; kx <= 6; ++kx) {
                const int ix = min(max(x + kx, 0), W - 1);
                const float tv = t[base + (size_t)iy * W + ix], pv = p[base + (size_t)iy * W + ix];
                rd += gw[kx + 6] * (pv - tv);
                rt += gw[kx + 6] * tv;
            }
            ad += gw[ky + 6] * rd;
            at += gw[ky + 6] * rt;
        }
        sd[base + i] = ad;
        stt[base + i] = at;
    }
}
// 3x3 Laplacian (4 at the centre, -1 on the 4-neighbours; scipy mode "reflect": index -1 -> 0, H -> H-1) of both smoothed
// images; sums[b][3] = sum lap(G(p - t))^2, sums[b][4] = sum lap(G t)^2
__global__ __launch_bounds__(256) void metric_lap_kernel(const float* __restrict__ sd, const float* __restrict__ stt, double* sums, int H, int W) {
    const int b = blockIdx.y;
    const size_t base = (size_t)b * H * W;
    double ed = 0.0, et = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        const int ym = max(y - 1, 0), yp = min(y + 1, H - 1), xm = max(x - 1, 0), xp = min(x + 1, W - 1);
        const float ld = 4.f * sd[base + i] - sd[base + (size_t)ym * W + x] - sd[base + (size_t)yp * W + x] - sd[base + (size_t)y * W + xm] -
                         sd[base + (size_t)y * W + xp];
        const float lt = 4.f * stt[base + i] - stt[base + (size_t)ym * W + x] - stt[base + (size_t)yp * W + x] - stt[base + (size_t)y * W + xm] -
                         stt[base + (size_t)y * W + xp];
        ed += (double)ld * ld;
        et += (double)lt * lt;
    }
    block_atomic_add(&sums[b * 6 + 3], ed);
    block_atomic_add(&sums[b * 6 + 4], et);
}
__global__ void metric_finish_kernel(const double* sums, float* out, int B, int H, int W) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* s = sums + b * 6;
    const double mse = s[0] / ((double)H * W);
    out[b * 4 + 0] = mse > 0.0 ? (float)(10.0 * log10(1.0 / mse)) : INFINITY;      // PSNR, data_range 1
    out[b * 4 + 1] = (float)(s[2] / ((double)(H - 10) * (W - 10)));                 // SSIM
    out[b * 4 + 2] = (float)(sqrt(s[3]) / (sqrt(s[4]) + 1e-8));                     // HFEN
    out[b * 4 + 3] = (float)(s[0] / (s[1] + 1e-8));                                 // NMSE
}

int launch_image_metrics(const float* pred, const float* gt, int B, int H, int W, float* scratch, double* sums, float* out, hipStream_t st) {
    MRISR_REQUIRE(pred && gt && scratch && sums && out && B >= 1, "metrics: null argument");
    MRISR_REQUIRE(H > 10 && W > 10, "metrics: the image must be larger than the 11x11 SSIM window");
    MRISR_CHECK_HIP(hipMemsetAsync(sums, 0, (size_t)B * 6 * sizeof(double), st));
    const int HW = H * W;
    int gx = (HW + 255) / 256;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(metric_sq_kernel, dim3(gx, B), dim3(256), 0, st, pred, gt, sums, HW);
    hipLaunchKernelGGL(metric_ssim_kernel, dim3((W - 10 + 15) / 16, (H - 10 + 15) / 16, B), dim3(256), 0, st, pred, gt, sums, H, W);
    float* sd = scratch;
    float* stt = scratch + (size_t)B * HW;
    hipLaunchKernelGGL(metric_gauss_kernel, dim3(gx, B), dim3(256), 0, st, pred, gt, sd, stt, H, W);
    hipLaunchKernelGGL(metric_lap_kernel, dim3(gx, B), dim3(256), 0, st, sd, stt, sums, H, W);
    hipLaunchKernelGGL(metric_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, st, sums, out, B, H, W);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace mrisr

extern "C" int mrisr_image_metrics(const float* pred_dev, const float* gt_dev, int batch, int height, int width, float* scratch_dev,
                                   double* sums_dev, float* out_dev, void* stream) {
    return mrisr::launch_image_metrics(pred_dev, gt_dev, batch, height, width, scratch_dev, sums_dev, out_dev, (hipStream_t)stream);
}
