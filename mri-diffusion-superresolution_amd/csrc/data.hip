// Slice degradation of the reference's data pipeline on the device: the notebook dataset builds each
// low-field input from its high-field slice with scipy's gaussian_filter (sigma = 0.5 * scale) followed by Pillow BICUBIC
// down- and up-sampling, and brings the slice to the target size with a Pillow LANCZOS resize (nb ResDif c22:102-154).  Here a
// whole batch of slices [B][H][W] f32 goes through the same arithmetic in a handful of HBM-bound passes:
//   * gaussian_filter: separable, scipy boundary mode "reflect" (d c b a | a b c d | d c b a), radius int(truncate*sigma+.5),
//     double accumulation, f32 intermediate between the two axes (rows first) as scipy keeps its output dtype;
//   * Pillow resize on mode "F": separable, horizontal pass first, per-output windows [xmin, xmin+n) of normalised filter
//     weights computed in double (Pillow's precompute_coeffs: support scaled by max(1, in/out), centre (i+.5)*in/out), double
//     accumulation, f32 intermediate.  The window tables are built on the device, one thread per output coordinate.
#include "common.h"
#include "prof.h"

namespace mrisr {

constexpr int kMaxTaps = 64;

__device__ __forceinline__ double pil_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0;
    if (x < 2.0) return (((x - 5.0) * x + 8.0) * x - 4.0) * a;
    return 0.0;
}
__device__ __forceinline__ double pil_sinc(double x) {
    if (x == 0.0) return 1.0;
    x *= 3.14159265358979323846;
    return sin(x) / x;
}
__device__ __forceinline__ double pil_lanczos(double x) { return (-3.0 <= x && x < 3.0) ? pil_sinc(x) * pil_sinc(x / 3.0) : 0.0; }

// bounds[i] = {first input index, tap count}; coef[i][0..ksize) normalised weights (zero past the count)
__global__ void resample_table_kernel(int in_size, int out_size, int filter, int ksize, int2* bounds, double* coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out_size) return;
    const double scale = (double)in_size / (double)out_size;
    const double fscale = scale < 1.0 ? 1.0 : scale;
    const double support = (filter == 1 ? 3.0 : 2.0) * fscale;
    const double center = (i + 0.5) * scale;
    const double ss = 1.0 / fscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    const int n = xmax - xmin;
    double* k = coef + (size_t)i * ksize;
    double ww = 0.0;
    for (int x = 0; x < n; ++x) {
        const double a = (x + xmin - center + 0.5) * ss;
        const double w = filter == 1 ? pil_lanczos(a) : pil_bicubic(a);
        k[x] = w;
        ww += w;
    }
    for (int x = 0; x < n; ++x)
        if (ww != 0.0) k[x] /= ww;
    for (int x = n; x < ksize; ++x) k[x] = 0.0;
    bounds[i] = make_int2(xmin, n);
}

// out[r][i] = sum_x in[r][xmin_i + x] * k_i[x]   (rows = B*H, contiguous along W)
__global__ __launch_bounds__(256) void resample_h_kernel(const float* __restrict__ in, float* __restrict__ out, long long rows, int W, int OW,
                                                         const int2* __restrict__ bounds, const double* __restrict__ coef, int ksize) {
    const long long n = rows * OW;
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const long long r = idx / OW;
        const int i = (int)(idx - r * OW);
        const int2 b = bounds[i];
        const double* k = coef + (size_t)i * ksize;
        const float* src = in + r * W + b.x;
        double s = 0.0;
        for (int x = 0; x < b.y; ++x) s += (double)src[x] * k[x];
        out[idx] = (float)s;
    }
}
// out[b][j][x] = sum_y in[b][ymin_j + y][x] * k_j[y]
__global__ __launch_bounds__(256) void resample_v_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int OH, int W,
                                                         const int2* __restrict__ bounds, const double* __restrict__ coef, int ksize) {
    const long long n = (long long)B * OH * W;
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const int x = (int)(idx % W);
        const long long t = idx / W;
        const int j = (int)(t % OH);
        const int b = (int)(t / OH);
        const int2 bd = bounds[j];
        const double* k = coef + (size_t)j * ksize;
        const float* src = in + ((size_t)b * H + bd.x) * W + x;
        double s = 0.0;
        for (int y = 0; y < bd.y; ++y) s += (double)src[(size_t)y * W] * k[y];
        out[idx] = (float)s;
    }
}

__device__ __forceinline__ int reflect_index(int i, int n) {
    // scipy "reflect": ... 1 0 | 0 1 ... n-1 | n-1 n-2 ...   (period 2n)
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}
// one axis of scipy.ndimage.gaussian_filter;  axis 0: along H (stride W), axis 1: along W (stride 1)
__global__ __launch_bounds__(256) void gauss_axis_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int axis,
                                                         int radius, double sigma) {
    __shared__ double wts[kMaxTaps + 1];
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int k = 0; k <= radius; ++k) {
            wts[k] = exp(-0.5 / (sigma * sigma) * (double)k * (double)k);
            sum += k == 0 ? wts[k] : 2.0 * wts[k];
        }
        for (int k = 0; k <= radius; ++k) wts[k] /= sum;
    }
    __syncthreads();
    const long long n = (long long)B * H * W;
    const int len = axis == 0 ? H : W;
    const int stride = axis == 0 ? W : 1;
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const int p = axis == 0 ? y : x;
        const float* line = in + idx - (size_t)p * stride;
        double s = (double)line[(size_t)p * stride] * wts[0];
        if (p >= radius && p + radius < len) {
            for (int k = 1; k <= radius; ++k) s += ((double)line[(size_t)(p - k) * stride] + (double)line[(size_t)(p + k) * stride]) * wts[k];
        } else {
            for (int k = 1; k <= radius; ++k)
                s += ((double)line[(size_t)reflect_index(p - k, len) * stride] + (double)line[(size_t)reflect_index(p + k, len) * stride]) * wts[k];
        }
        out[idx] = (float)s;
    }
}

static inline int data_blocks(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
static inline int resample_ksize(int in_size, int out_size, int filter) {
    const double scale = (double)in_size / (double)out_size;
    const double support = (filter == 1 ? 3.0 : 2.0) * (scale < 1.0 ? 1.0 : scale);
    return (int)std::ceil(support) * 2 + 1;
}
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t table_bytes(int in_size, int out_size, int filter) {
    return align256((size_t)out_size * sizeof(int2)) + align256((size_t)out_size * resample_ksize(in_size, out_size, filter) * sizeof(double));
}

size_t resize_scratch_bytes(int B, int H, int W, int OH, int OW, int filter) {
    return table_bytes(W, OW, filter) + table_bytes(H, OH, filter) + align256((size_t)B * H * OW * sizeof(float));
}

int launch_resize_slices(const float* in, int B, int H, int W, float* out, int OH, int OW, int filter, void* scratch, size_t scratch_bytes,
                         hipStream_t st) {
    MRISR_REQUIRE(in && out && scratch && B >= 1 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1, "resize: bad argument");
    MRISR_REQUIRE(filter == 0 || filter == 1, "resize: filter must be 0 (bicubic) or 1 (lanczos)");
    MRISR_REQUIRE(scratch_bytes >= resize_scratch_bytes(B, H, W, OH, OW, filter), "resize: scratch too small");
    ProfScope ps("data_resize", 0.0, 4.0 * B * ((double)H * W + 2.0 * H * OW + (double)OH * OW), st);
    char* p = static_cast<char*>(scratch);
    const int kh = resample_ksize(W, OW, filter), kv = resample_ksize(H, OH, filter);
    int2* bh = reinterpret_cast<int2*>(p); p += align256((size_t)OW * sizeof(int2));
    double* ch = reinterpret_cast<double*>(p); p += align256((size_t)OW * kh * sizeof(double));
    int2* bv = reinterpret_cast<int2*>(p); p += align256((size_t)OH * sizeof(int2));
    double* cv = reinterpret_cast<double*>(p); p += align256((size_t)OH * kv * sizeof(double));
    float* tmp = reinterpret_cast<float*>(p);
    // Pillow skips a pass whose size does not change; the other pass then works on the input directly
    const bool need_h = OW != W, need_v = OH != H;
    if (!need_h && !need_v) {
        MRISR_CHECK_HIP(hipMemcpyAsync(out, in, (size_t)B * H * W * sizeof(float), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    const float* src = in;
    if (need_h) {
        hipLaunchKernelGGL(resample_table_kernel, dim3((OW + 63) / 64), dim3(64), 0, st, W, OW, filter, kh, bh, ch);
        float* dst = need_v ? tmp : out;
        hipLaunchKernelGGL(resample_h_kernel, dim3(data_blocks((long long)B * H * OW)), dim3(256), 0, st, src, dst, (long long)B * H, W, OW, bh, ch, kh);
        src = dst;
    }
    if (need_v) {
        hipLaunchKernelGGL(resample_table_kernel, dim3((OH + 63) / 64), dim3(64), 0, st, H, OH, filter, kv, bv, cv);
        hipLaunchKernelGGL(resample_v_kernel, dim3(data_blocks((long long)B * OH * OW)), dim3(256), 0, st, src, out, B, H, OH, OW, bv, cv, kv);
    }
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gaussian_blur(const float* in, int B, int H, int W, float sigma, float truncate, float* tmp, float* out, hipStream_t st) {
    MRISR_REQUIRE(in && tmp && out && B >= 1 && H >= 1 && W >= 1, "gaussian blur: bad argument");
    MRISR_REQUIRE(sigma > 0.f && truncate > 0.f, "gaussian blur: sigma and truncate must be positive");
    const int radius = (int)(truncate * sigma + 0.5f);
    MRISR_REQUIRE(radius >= 1 && radius <= kMaxTaps, "gaussian blur: radius out of range (1..64)");
    ProfScope ps("data_blur", 0.0, 16.0 * B * H * W, st);
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(gauss_axis_kernel, dim3(data_blocks(n)), dim3(256), 0, st, in, tmp, B, H, W, 0, radius, (double)sigma);
    hipLaunchKernelGGL(gauss_axis_kernel, dim3(data_blocks(n)), dim3(256), 0, st, tmp, out, B, H, W, 1, radius, (double)sigma);
    MRISR_CHECK_HIP(hipGetLastError());
    return 0;
}

size_t low_field_scratch_bytes(int B, int H, int W, float scale) {
    const int sh = (int)(W / scale), sw = (int)(H / scale);  // (sic) see launch_simulate_low_field
    const size_t img = align256((size_t)B * H * W * sizeof(float));
    const size_t small = align256((size_t)B * (sh > 0 ? sh : 1) * (sw > 0 ? sw : 1) * sizeof(float));
    size_t rs = resize_scratch_bytes(B, H, W, sh > 0 ? sh : 1, sw > 0 ? sw : 1, 0);
    const size_t up = resize_scratch_bytes(B, sh > 0 ? sh : 1, sw > 0 ? sw : 1, H, W, 0);
    if (up > rs) rs = up;
    return 2 * img + small + rs;
}

// nb ResDif c22:140-154: blur(sigma = 0.5 * scale) -> BICUBIC to small_size -> BICUBIC back to target_size.  The reference
// builds small_size as (target_size[1] // scale, target_size[0] // scale) and hands both tuples to PIL, which reads them as
// (width, height): the intermediate image is therefore (W // scale) rows by (H // scale) columns - transposed proportions for
// a non-square slice.  Reproduced as is (no difference for the square targets the reference uses).
int launch_simulate_low_field(const float* hr, int B, int H, int W, float scale, float* lr, void* scratch, size_t scratch_bytes, hipStream_t st) {
    MRISR_REQUIRE(hr && lr && scratch && B >= 1, "simulate_low_field: bad argument");
    MRISR_REQUIRE(scale >= 1.f, "simulate_low_field: scale_factor must be >= 1");
    const int sh = (int)(W / scale), sw = (int)(H / scale);
    MRISR_REQUIRE(sh >= 1 && sw >= 1, "simulate_low_field: slice smaller than the scale factor");
    MRISR_REQUIRE(scratch_bytes >= low_field_scratch_bytes(B, H, W, scale), "simulate_low_field: scratch too small");
    char* p = static_cast<char*>(scratch);
    const size_t img = align256((size_t)B * H * W * sizeof(float));
    float* t0 = reinterpret_cast<float*>(p); p += img;
    float* t1 = reinterpret_cast<float*>(p); p += img;
    float* small = reinterpret_cast<float*>(p); p += align256((size_t)B * sh * sw * sizeof(float));
    const size_t rest = scratch_bytes - (size_t)(p - static_cast<char*>(scratch));
    int rc = launch_gaussian_blur(hr, B, H, W, 0.5f * scale, 4.0f, t0, t1, st);
    if (rc) return rc;
    rc = launch_resize_slices(t1, B, H, W, small, sh, sw, 0, p, rest, st);
    if (rc) return rc;
    return launch_resize_slices(small, B, sh, sw, lr, H, W, 0, p, rest, st);
}

}  // namespace mrisr

extern "C" {
size_t mrisr_resize_scratch_bytes(int batch, int height, int width, int out_height, int out_width, int filter) {
    return mrisr::resize_scratch_bytes(batch, height, width, out_height, out_width, filter);
}
int mrisr_resize_slices(const float* in_dev, int batch, int height, int width, float* out_dev, int out_height, int out_width, int filter,
                        void* scratch_dev, size_t scratch_bytes, void* stream) {
    return mrisr::launch_resize_slices(in_dev, batch, height, width, out_dev, out_height, out_width, filter, scratch_dev, scratch_bytes,
                                       (hipStream_t)stream);
}
int mrisr_gaussian_blur_slices(const float* in_dev, int batch, int height, int width, float sigma, float truncate, float* tmp_dev,
                               float* out_dev, void* stream) {
    return mrisr::launch_gaussian_blur(in_dev, batch, height, width, sigma, truncate, tmp_dev, out_dev, (hipStream_t)stream);
}
size_t mrisr_low_field_scratch_bytes(int batch, int height, int width, float scale_factor) {
    return mrisr::low_field_scratch_bytes(batch, height, width, scale_factor);
}
int mrisr_simulate_low_field(const float* hr_dev, int batch, int height, int width, float scale_factor, float* lr_dev, void* scratch_dev,
                             size_t scratch_bytes, void* stream) {
    return mrisr::launch_simulate_low_field(hr_dev, batch, height, width, scale_factor, lr_dev, scratch_dev, scratch_bytes, (hipStream_t)stream);
}
}
