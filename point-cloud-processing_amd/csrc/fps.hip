// Batched farthest-point sampling on gfx950.
//
// Replaces farthest_point_sampling, dp3_point_cloud_processing.py:7-21, i.e.
// pytorch3d.ops.sample_farthest_points(points=(1,M,3) f32, K=[K]) with the default
// random_start_point=False: start at index 0, K-1 rounds of
//     min_d[p] = min(min_d[p], |p - last|^2);  last = argmax_p min_d[p]
// with f32 distances accumulated in x,y,z order and ties broken toward the LOWEST index
// (strict '>' scan on the CPU build, cub::ArgMax on the CUDA build).
//
// dist_mode 0: ((dx*dx)+(dy*dy))+(dz*dz), each op rounded (file is built -ffp-contract=off)
// dist_mode 1: fmaf(dz,dz,fmaf(dy,dy,dx*dx)), nvcc's default contraction of the same source
//
// Arg-max keys: min_d is never negative, so its f32 bit pattern orders like the value.
// key = (bits(min_d) << 32) | ~index  makes "largest value, then lowest index" one u64 max;
// key 0 means "no candidate".
#include "pcp_internal.h"
#include <float.h>

template <int MODE>
__device__ __forceinline__ float dist2(float sx, float sy, float sz, float x, float y, float z)
{
    const float dx = sx - x, dy = sy - y, dz = sz - z;
    if (MODE == 0) return ((dx * dx) + (dy * dy)) + (dz * dz);
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

__device__ __forceinline__ unsigned long long make_key(float v, unsigned idx)
{
    return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(~idx);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(k, d, 64);
        k = o > k ? o : k;
    }
    return k;
}

// ---- brute force: one 1024-thread workgroup per cloud, min_d streamed through `md` ----------
template <int MODE>
__global__ void __launch_bounds__(1024)
fps_brute(const float* __restrict__ xyz, const long long* __restrict__ offsets, int K,
          long long* __restrict__ idx, float* __restrict__ out_xyz, float* __restrict__ md)
{
    __shared__ unsigned long long part[2][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = offsets[b];
    const long long Mll = offsets[b + 1] - base;
    const int M = (int)Mll;
    const int kn = K < M ? K : (M > 0 ? M : 0);
    long long* irow = idx + (size_t)b * K;
    float* orow = out_xyz ? out_xyz + (size_t)b * K * 3 : nullptr;
    for (int k = kn + tid; k < K; k += 1024) {           // pytorch3d padding: -1 / zeros
        irow[k] = -1;
        if (orow) { orow[3 * k] = 0.f; orow[3 * k + 1] = 0.f; orow[3 * k + 2] = 0.f; }
    }
    if (kn == 0) return;
    const float* P = xyz + 3 * base;
    float* mdp = md + base;
    for (int p = tid; p < M; p += 1024) mdp[p] = FLT_MAX;
    int last = 0;
    for (int k = 0;; ++k) {
        const float sx = P[3 * last], sy = P[3 * last + 1], sz = P[3 * last + 2];
        if (tid == 0) {
            irow[k] = last;
            if (orow) { orow[3 * k] = sx; orow[3 * k + 1] = sy; orow[3 * k + 2] = sz; }
        }
        if (k + 1 >= kn) break;
        float best = -1.0f;
        int besti = 0;
        for (int p = tid; p < M; p += 1024) {
            const float d = dist2<MODE>(sx, sy, sz, P[3 * p], P[3 * p + 1], P[3 * p + 2]);
            float m = mdp[p];
            m = d < m ? d : m;
            mdp[p] = m;
            if (m > best) { best = m; besti = p; }
        }
        unsigned long long key = best >= 0.0f ? make_key(best, (unsigned)besti) : 0ull;
        key = wave_max_u64(key);
        if (lane == 0) part[k & 1][wave] = key;
        __syncthreads();
        unsigned long long kk = part[k & 1][lane & 15];
        kk = wave_max_u64(kk);
        last = kk ? (int)(~(unsigned)(kk & 0xffffffffull)) : 0;   // kk == 0 only for NaN input
    }
}

// ---- host entry points -------------------------------------------------------------------------
extern "C" size_t pcp_fps_workspace_bytes(int B, int64_t total_pts, int64_t max_pts, int K, int algo)
{
    (void)B; (void)max_pts; (void)K; (void)algo;
    if (total_pts < 0) total_pts = 0;
    return pcp_align256((size_t)total_pts * sizeof(float)) + 256;
}

extern "C" int pcp_fps_batched(const float* xyz, const int64_t* offsets, int B, int64_t total_pts,
                               int64_t max_pts, int K, int dist_mode, int algo, int64_t* idx,
                               float* out_xyz, void* ws, size_t ws_bytes, void* stream)
{
    if (B < 0 || K < 0 || total_pts < 0 || max_pts < 0 || !offsets || (dist_mode != 0 && dist_mode != 1) ||
        algo < 0 || algo > 2)
        return PCP_ERR_INVALID_ARG;
    if ((total_pts > 0 && !xyz) || (B > 0 && K > 0 && !idx)) return PCP_ERR_INVALID_ARG;
    if (max_pts > 0x7fffffffLL) return PCP_ERR_INVALID_ARG;
    if (!ws || ((uintptr_t)ws & 255) || ws_bytes < pcp_fps_workspace_bytes(B, total_pts, max_pts, K, algo))
        return PCP_ERR_WORKSPACE;
    if (B == 0 || K == 0) return PCP_OK;
    hipStream_t st = (hipStream_t)stream;
    float* md = (float*)ws;
    if (dist_mode == 0)
        hipLaunchKernelGGL(fps_brute<0>, dim3(B), dim3(1024), 0, st, xyz, (const long long*)offsets, K,
                           (long long*)idx, out_xyz, md);
    else
        hipLaunchKernelGGL(fps_brute<1>, dim3(B), dim3(1024), 0, st, xyz, (const long long*)offsets, K,
                           (long long*)idx, out_xyz, md);
    return pcp_launch_status();
}
