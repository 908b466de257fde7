// Small kernels around FPS: the colour gather of dp3_point_cloud_processing.py:48 and the
// (K,6) row packing that chunking_data.py:134-144 performs before the zarr write (CHK:195);
// plus the parameter defaults and status strings of the C ABI.
#include "pcp_internal.h"

__global__ void __launch_bounds__(256)
gather_rgb_k(const float* __restrict__ rgb, const long long* __restrict__ offsets, long long total_pts,
             const long long* __restrict__ idx, int K, int compat_neg, float* __restrict__ out)
{
    const int b = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const long long base = offsets[b], M = offsets[b + 1] - base;
    long long i = idx[(size_t)b * K + k];
    float* o = out + ((size_t)b * K + k) * 3;
    if (i < 0) {
        if (compat_neg && M > 0 && i + M >= 0) i += M;        // numpy negative indexing, DP3:48
        else { o[0] = o[1] = o[2] = 0.f; return; }
    }
    // a row outside the cloud or outside the caller's array (a cloud stage 1 could not store: capacity overflow)
    // reads nothing: zeros.  numpy would raise IndexError here; the flag of the producing call is the error report.
    if (i >= M || base < 0 || base + i >= total_pts) { o[0] = o[1] = o[2] = 0.f; return; }
    const float* s = rgb + 3 * (base + i);
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
}

__global__ void __launch_bounds__(256)
pack_dp3_k(const float* __restrict__ xyz, const float* __restrict__ rgb, long long n,
           float* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[6 * i + c] = xyz[3 * i + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = (double)rgb[3 * i + c];
        v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        const double q = rint(v * 255.0);                      // PLY uchar quantisation
        out[6 * i + 3 + c] = (float)((q / 255.0) * 255.0);     // read back /255, CHK:139 *255
    }
}

// extract_topics_one_camera.py:72 (XTP): depth = depth_u16.astype(np.float32) / 1000.0 — one f32 division
__global__ void __launch_bounds__(256)
depth_mm_k(const uint16_t* __restrict__ mm, long long n, float* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)mm[i] / 1000.0f;
}

extern "C" int pcp_depth_mm_to_m(const uint16_t* depth_mm, int64_t n, float* depth_m, void* stream)
{
    if (n < 0) return PCP_ERR_INVALID_ARG;
    if (n == 0) return PCP_OK;
    if (!depth_mm || !depth_m) return PCP_ERR_INVALID_ARG;
    hipLaunchKernelGGL(depth_mm_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       depth_mm, (long long)n, depth_m);
    return pcp_launch_status();
}

extern "C" int pcp_gather_rgb(const float* rgb, const int64_t* offsets, int64_t total_pts, const int64_t* idx,
                              int B, int K, int compat_neg_index, float* out, void* stream)
{
    if (B < 0 || K < 0 || total_pts < 0) return PCP_ERR_INVALID_ARG;
    if (B == 0 || K == 0) return PCP_OK;
    if ((total_pts > 0 && !rgb) || !offsets || !idx || !out) return PCP_ERR_INVALID_ARG;
    hipLaunchKernelGGL(gather_rgb_k, dim3((K + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, rgb,
                       (const long long*)offsets, (long long)total_pts, (const long long*)idx, K, compat_neg_index, out);
    return pcp_launch_status();
}

extern "C" int pcp_pack_dp3(const float* xyz, const float* rgb, int B, int K, float* out, void* stream)
{
    if (B < 0 || K < 0) return PCP_ERR_INVALID_ARG;
    const long long n = (long long)B * K;
    if (n == 0) return PCP_OK;
    if (!xyz || !rgb || !out) return PCP_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pack_dp3_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       xyz, rgb, n, out);
    return pcp_launch_status();
}

extern "C" void pcp_default_params(pcp_params* p)
{
    if (!p) return;
    *p = pcp_params{};
    for (int i = 0; i < 16; ++i) p->T_d2c[i] = (i % 5 == 0) ? 1.0 : 0.0;
    p->T_d2c[3] = (double)(-0.059f);                           // XPC:17-22 is float32
    static const double ext[16] = {-0.5777, 0.4385, -0.6884, 0.7814,     // S1:13-18
                                   0.7992, 0.1327, -0.5862, 0.4271,
                                   -0.1657, -0.8889, -0.4271, 0.5267,
                                   0.0, 0.0, 0.0, 1.0};
    for (int i = 0; i < 16; ++i) p->ext[i] = ext[i];
    p->ws_lo[0] = -0.05;  p->ws_hi[0] = 0.4;                   // DP3:29-33
    p->ws_lo[1] = -0.12;  p->ws_hi[1] = 0.35;
    p->ws_lo[2] = -0.008; p->ws_hi[2] = 0.6;
    p->z_min = 0.01f; p->z_max = 5.0f;                         // XPC:109
    p->zc_min = 0.01; p->zc_clamp = 0.001;                     // XPC:75, XPC:70-71
    p->apply_crop = 1;
    p->transform_fma = 1;
}

extern "C" const char* pcp_strerror(int status)
{
    switch (status) {
        case PCP_OK: return "ok";
        case PCP_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or bad enum)";
        case PCP_ERR_WORKSPACE: return "workspace missing, misaligned (256 B) or too small";
        case PCP_ERR_CAPACITY: return "output capacity too small";
        case PCP_ERR_LAUNCH: return "HIP kernel launch failed";
        case PCP_ERR_NO_DEVICE: return "no HIP device";
        case PCP_ERR_EMPTY: return "empty point cloud";
        default: return "unknown pcp status";
    }
}

extern "C" const char* pcp_version(void) { return "pcp-hip 0.2 gfx950"; }

extern "C" int pcp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
