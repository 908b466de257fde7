// Stage-2 workspace crop on points that are already in the base frame (the PLY-input
// drop-in path).  Replaces the mask and boolean gathers of process_point_cloud,
// dp3_point_cloud_processing.py:35-42, and the .float() cast of :10 for the FPS input.
// Six strict FP64 compares; compaction preserves row order (numpy boolean-mask order).
#include "pcp_internal.h"

struct Box { double lo[3], hi[3]; };

__device__ __forceinline__ int in_box3(const Box& bx, const double* __restrict__ p)
{
    const double x = p[0], y = p[1], z = p[2];
    return x > bx.lo[0] && x < bx.hi[0] && y > bx.lo[1] && y < bx.hi[1] &&
           z > bx.lo[2] && z < bx.hi[2];                                       // DP3:35-39
}

// grid tiles, block 256; optional byte mask, ballots and tile counts
__global__ void __launch_bounds__(256)
crop_flags(const double* __restrict__ xyz, long long M1, Box bx, uint8_t* __restrict__ mask,
           unsigned long long* __restrict__ masks, int* __restrict__ tile_cnt)
{
    __shared__ int wave_cnt[4];
    const int lane = pcp_lane(), wave = threadIdx.x >> 6;
    const long long tile = blockIdx.x;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int word = wave * 4 + r;
        const long long i = tile * PCP_TILE + word * 64 + lane;
        int keep = 0;
        if (i < M1) {
            keep = in_box3(bx, xyz + 3 * i);
            if (mask) mask[i] = (uint8_t)keep;
        }
        const unsigned long long m = __ballot(keep);
        if (masks && lane == 0) masks[tile * PCP_TILE_WORDS + word] = m;
        cnt += __popcll(m);
    }
    if (lane == 0) wave_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0 && tile_cnt)
        tile_cnt[tile] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ void __launch_bounds__(256)
crop_scatter(const double* __restrict__ xyz, const double* __restrict__ rgb,
             const unsigned long long* __restrict__ masks, const int* __restrict__ tile_off,
             double* out_xyz64, float* out_xyz32, double* out_rgb64, long long cap)
{
    __shared__ unsigned long long m[PCP_TILE_WORDS];
    __shared__ int pre[PCP_TILE_WORDS + 1];
    const long long tile = blockIdx.x;
    if (threadIdx.x < PCP_TILE_WORDS) m[threadIdx.x] = masks[tile * PCP_TILE_WORDS + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int w = 0; w < PCP_TILE_WORDS; ++w) { pre[w] = acc; acc += __popcll(m[w]); }
        pre[PCP_TILE_WORDS] = acc;
    }
    __syncthreads();
    const int n = pre[PCP_TILE_WORDS];
    const long long base = tile_off[tile];
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        int w = 0;
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1)
            if (pre[w + s] <= j) w += s;
        const long long i = tile * PCP_TILE + w * 64 + pcp_select_bit(m[w], j - pre[w]);
        const long long row = base + j;
        if (row >= cap) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = xyz[3 * i + c];
            if (out_xyz64) out_xyz64[3 * row + c] = v;
            if (out_xyz32) out_xyz32[3 * row + c] = (float)v;                 // DP3:10
            if (out_rgb64) out_rgb64[3 * row + c] = rgb[3 * i + c];
        }
    }
}

static Box make_box(const double lo[3], const double hi[3])
{
    Box b;
    for (int i = 0; i < 3; ++i) { b.lo[i] = lo[i]; b.hi[i] = hi[i]; }
    return b;
}

// workspace: masks u64[tiles*16] | tile_cnt i32[tiles] | frame_cnt i64[1] | offsets i64[2]
extern "C" size_t pcp_crop_workspace_bytes(int64_t M1)
{
    if (M1 <= 0) return 256;
    const size_t tiles = ((size_t)M1 + PCP_TILE - 1) / PCP_TILE;
    return pcp_align256(tiles * PCP_TILE_WORDS * 8) + pcp_align256(tiles * 4) + 512;
}

extern "C" int pcp_crop_mask(const double* xyz, int64_t M1, const double lo[3], const double hi[3],
                             uint8_t* mask, void* stream)
{
    if (M1 < 0 || !lo || !hi || (M1 > 0 && (!xyz || !mask))) return PCP_ERR_INVALID_ARG;
    if (M1 == 0) return PCP_OK;
    const long long tiles = (M1 + PCP_TILE - 1) / PCP_TILE;
    hipLaunchKernelGGL(crop_flags, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, xyz,
                       (long long)M1, make_box(lo, hi), mask, (unsigned long long*)nullptr, (int*)nullptr);
    return pcp_launch_status();
}

extern "C" int pcp_crop_compact(const double* xyz, const double* rgb, int64_t M1, const double lo[3],
                                const double hi[3], double* out_xyz64, float* out_xyz32,
                                double* out_rgb64, int64_t cap, int64_t* count, void* ws,
                                size_t ws_bytes, void* stream)
{
    if (M1 < 0 || !lo || !hi || !count || cap < 0 || (M1 > 0 && !xyz) || (out_rgb64 && !rgb))
        return PCP_ERR_INVALID_ARG;
    if (!ws || ((uintptr_t)ws & 255) || ws_bytes < pcp_crop_workspace_bytes(M1)) return PCP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (M1 == 0) {
        PCP_FILL_TRY(count, 0u, 2, st);
        return PCP_OK;
    }
    const long long tiles = (M1 + PCP_TILE - 1) / PCP_TILE;
    if (tiles > 0x7fffffffLL) return PCP_ERR_INVALID_ARG;
    char* w = (char*)ws;
    unsigned long long* masks = (unsigned long long*)w;
    w += pcp_align256((size_t)tiles * PCP_TILE_WORDS * 8);
    int* tile_cnt = (int*)w;
    hipLaunchKernelGGL(crop_flags, dim3((unsigned)tiles), dim3(256), 0, st, xyz, (long long)M1,
                       make_box(lo, hi), (uint8_t*)nullptr, masks, tile_cnt);
    hipLaunchKernelGGL(pcp_scan_tiles, dim3(1), dim3(256), 0, st, tile_cnt, (int)tiles, (long long*)count);
    hipLaunchKernelGGL(crop_scatter, dim3((unsigned)tiles), dim3(256), 0, st, xyz, rgb, masks, tile_cnt,
                       out_xyz64, out_xyz32, out_rgb64, (long long)cap);
    return pcp_launch_status();
}

// B clouds in CSR form at once (the stage-2 script reads many PLYs and crops them in one go): the same
// flag / scan / scatter passes over the concatenated rows, plus the CSR offsets of the result.
extern "C" int pcp_crop_compact_batched(const double* xyz, const double* rgb, const int64_t* offsets, int B,
                                        int64_t total, const double lo[3], const double hi[3],
                                        double* out_xyz64, float* out_xyz32, double* out_rgb64,
                                        int64_t* out_offsets, int64_t cap, void* ws, size_t ws_bytes,
                                        void* stream)
{
    if (B < 0 || total < 0 || !lo || !hi || !offsets || !out_offsets || cap < 0 || (total > 0 && !xyz) ||
        (out_rgb64 && !rgb))
        return PCP_ERR_INVALID_ARG;
    if (!ws || ((uintptr_t)ws & 255) || ws_bytes < pcp_crop_workspace_bytes(total)) return PCP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (total == 0 || B == 0) {
        PCP_FILL_TRY(out_offsets, 0u, 2 * (size_t)(B + 1), st);
        return PCP_OK;
    }
    const long long tiles = (total + PCP_TILE - 1) / PCP_TILE;
    if (tiles > 0x7fffffffLL) return PCP_ERR_INVALID_ARG;
    char* w = (char*)ws;
    unsigned long long* masks = (unsigned long long*)w;
    w += pcp_align256((size_t)tiles * PCP_TILE_WORDS * 8);
    int* tile_cnt = (int*)w;
    w += pcp_align256((size_t)tiles * 4);
    long long* count = (long long*)w;
    hipLaunchKernelGGL(crop_flags, dim3((unsigned)tiles), dim3(256), 0, st, xyz, (long long)total,
                       make_box(lo, hi), (uint8_t*)nullptr, masks, tile_cnt);
    hipLaunchKernelGGL(pcp_scan_tiles, dim3(1), dim3(256), 0, st, tile_cnt, (int)tiles, count);
    hipLaunchKernelGGL(crop_scatter, dim3((unsigned)tiles), dim3(256), 0, st, xyz, rgb, masks, tile_cnt,
                       out_xyz64, out_xyz32, out_rgb64, (long long)cap);
    hipLaunchKernelGGL(pcp_csr_offsets, dim3((B + 256) / 256), dim3(256), 0, st, (const long long*)offsets, B,
                       masks, tile_cnt, count, (long long*)out_offsets);
    return pcp_launch_status();
}
