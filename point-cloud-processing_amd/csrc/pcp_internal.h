// Internal helpers shared by the gfx950 kernels of libpcp_hip.so.  Not installed.
// Built with -ffp-contract=off: every written floating-point operation is rounded
// on its own; fused multiply-adds appear only where fma()/fmaf() is written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pcp.h"

#define PCP_WAVE 64
#define PCP_TILE 1024          // pixels / points per compaction tile (16 ballots of 64)
#define PCP_TILE_WORDS 16

// Device-side copy of pcp_params, passed by value as a kernel argument.
struct DevParams {
    double dfx, dfy, dcx, dcy;      // depth intrinsics
    double cfx, cfy, ccx, ccy;      // colour intrinsics
    double T[12];                   // rows 0..2 of the depth->colour 4x4
    double E[12];                   // rows 0..2 of the base<-camera 4x4
    double lo[3], hi[3];            // workspace box
    double zc_min, zc_clamp;
    float z_min, z_max;
    int apply_crop, transform_fma;
};

static inline DevParams pcp_to_dev(const pcp_params* p)
{
    DevParams d;
    d.dfx = p->depth.fx; d.dfy = p->depth.fy; d.dcx = p->depth.cx; d.dcy = p->depth.cy;
    d.cfx = p->color.fx; d.cfy = p->color.fy; d.ccx = p->color.cx; d.ccy = p->color.cy;
    for (int i = 0; i < 12; ++i) { d.T[i] = p->T_d2c[i]; d.E[i] = p->ext[i]; }
    for (int i = 0; i < 3; ++i) { d.lo[i] = p->ws_lo[i]; d.hi[i] = p->ws_hi[i]; }
    d.zc_min = p->zc_min; d.zc_clamp = p->zc_clamp;
    d.z_min = p->z_min; d.z_max = p->z_max;
    d.apply_crop = p->apply_crop; d.transform_fma = p->transform_fma;
    return d;
}

#define PCP_HIP_TRY(expr)                                   \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) { (void)hipGetLastError(); return PCP_ERR_LAUNCH; } \
    } while (0)

static inline int pcp_launch_status()
{
    return hipGetLastError() == hipSuccess ? PCP_OK : PCP_ERR_LAUNCH;
}

static inline size_t pcp_align256(size_t n) { return (n + 255) & ~(size_t)255; }

// ---- wave / block primitives (64-wide wavefronts) ---------------------------------------

__device__ __forceinline__ int pcp_lane() { return threadIdx.x & 63; }

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ int pcp_wave_incl_scan(int v)
{
    const int lane = pcp_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive prefix sum of one int per thread over a block of up to 1024 threads.
// `lds` needs 17 ints. Returns the exclusive prefix; *total gets the block sum.
__device__ __forceinline__ int pcp_block_excl_scan(int v, int* lds, int* total)
{
    const int lane = pcp_lane(), wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = pcp_wave_incl_scan(v);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int w = lane < nw ? lds[lane] : 0;
        int winc = pcp_wave_incl_scan(w);
        if (lane < nw) lds[lane] = winc - w;     // exclusive wave offsets
        if (lane == 63) lds[16] = winc;          // block total
    }
    __syncthreads();
    int out = lds[wave] + inc - v;
    *total = lds[16];
    __syncthreads();                             // lds reusable afterwards
    return out;
}

// (i / W, i % W) for 0 <= i < 2^31 and i / W < 2^20 without the ~40-instruction integer division: a float estimate of
// the quotient (off by at most one in that range) and an exact integer fix-up.
__device__ __forceinline__ void pcp_divmod(int i, int W, float invW, int* q, int* r)
{
    int v = (int)((float)i * invW);
    int u = i - v * W;
    if (u < 0) { v -= 1; u += W; }
    else if (u >= W) { v += 1; u -= W; }
    *q = v; *r = u;
}

// position (0..63) of the n-th (0-based) set bit of m; m must have more than n bits set
__device__ __forceinline__ int pcp_select_bit(unsigned long long m, int n)
{
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        unsigned long long lowmask = (w == 64) ? ~0ull : ((1ull << w) - 1ull);
        int c = __popcll((m >> pos) & lowmask);
        if (n >= c) { n -= c; pos += w; }
    }
    return pos;
}
