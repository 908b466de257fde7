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
    double rdfx, rdfy;              // RN(1/dfx), RN(1/dfy) for pcp_quot(); 0 = this launch divides the plain way
    double cfx, cfy, ccx, ccy;      // colour intrinsics
    double T[12];                   // rows 0..2 of the depth->colour 4x4
    double E[12];                   // rows 0..2 of the base<-camera 4x4
    double lo[3], hi[3];            // workspace box
    double zc_min, zc_clamp;
    float z_min, z_max;
    int apply_crop, transform_fma;
};

// Correctly rounded reciprocal of a launch-constant divisor (IEEE division on the host), or 0 when the divisor is
// not a plain mid-range number — the kernels then use true divisions throughout.
static inline double pcp_safe_recip(double d)
{
    const double a = d < 0 ? -d : d;
    if (!(a > 0x1p-200 && a < 0x1p200)) return 0.0;
    return 1.0 / d;
}

static inline DevParams pcp_to_dev(const pcp_params* p)
{
    DevParams d;
    d.dfx = p->depth.fx; d.dfy = p->depth.fy; d.dcx = p->depth.cx; d.dcy = p->depth.cy;
    d.rdfx = pcp_safe_recip(d.dfx); d.rdfy = pcp_safe_recip(d.dfy);
    // the principal point must be tame too (zero, or a mid-range number): pcp_quot_valid() relies on |u - cx| being
    // either zero or far from the denormals
    const bool tame_c = (d.dcx == 0.0 || pcp_safe_recip(d.dcx) != 0.0) && (d.dcy == 0.0 || pcp_safe_recip(d.dcy) != 0.0) &&
                        (d.dcx < 0x1p100 && d.dcx > -0x1p100) && (d.dcy < 0x1p100 && d.dcy > -0x1p100);
    if (d.rdfx == 0.0 || d.rdfy == 0.0 || !tame_c) d.rdfx = d.rdfy = 0.0;
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

// Flags, counters and small tables are cleared by a kernel of the library's own, not by hipMemsetAsync: recorded into a
// HIP graph, a memset node in the middle of the graph replayed with a wrong fill value from the second replay on
// (ROCm 7.2; the first node of the graph was fine) — tests/test_gpu_graph.py caught the FPS overflow flag coming back
// as the low bits of a pointer.  A kernel node has no such surprise, and the cost is the same launch.
static __global__ void pcp_fill_u32_k(unsigned int* __restrict__ p, unsigned int v, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

static inline int pcp_fill_u32(void* p, unsigned int v, size_t n_words, hipStream_t st)
{
    if (n_words == 0) return PCP_OK;
    hipLaunchKernelGGL(pcp_fill_u32_k, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, (unsigned int*)p, v, n_words);
    return hipGetLastError() == hipSuccess ? PCP_OK : PCP_ERR_LAUNCH;
}
#define PCP_FILL_TRY(p, v, n_words, st) do { if (pcp_fill_u32((p), (v), (n_words), (st)) != PCP_OK) return PCP_ERR_LAUNCH; } while (0)

// ---- exact FP64 division by a launch constant ------------------------------------------------------
// n / d, correctly rounded — bit for bit the IEEE division the reference's numpy expression performs — given
// r = RN(1/d), d a mid-range number (pcp_safe_recip).  q0 = RN(n*r) is within a few ulp; one residual step
// (e = n - q*d is exact in an FMA) makes q1 a faithful quotient, and a second one is Markstein's final step: with r
// the correctly rounded reciprocal and q1 faithful, RN(q1 + r*(n - q1*d)) = RN(n/d) provided nothing over- or
// underflows on the way.  That proviso is checked on the result's exponent: a quotient outside a generous safe range
// (infinities and NaN included) sets `bad`, and the caller recomputes with true divisions — one wave-uniform,
// practically never taken branch per wave instead of one per quotient.  A zero numerator keeps q0, which carries the
// sign n/d would have.  The FP64 division sequence (v_div_scale x2, v_rcp_f64, Newton steps, v_div_fmas, v_div_fixup)
// costs about three times as many issue cycles.
__device__ __forceinline__ double pcp_quot(double n, double d, double r, bool& bad)
{
    const double q0 = n * r;
    double e = fma(-q0, d, n);
    const double q1 = fma(e, r, q0);
    e = fma(-q1, d, n);
    const double q = fma(e, r, q1);
    const unsigned ex = ((unsigned)__double2hiint(q) >> 20) & 0x7ffu;      // biased exponent
    const bool zero = n == 0.0;
    bad = bad || !(ex - 523u < 1000u || zero);                            // 2^-500 <= |q| < 2^500, or an exact zero
    return zero ? q0 : q;
}

// The same quotient without the range check, for numerators that are KNOWN to be tame: n = (u - c) * z with an
// integer pixel coordinate u < 2^31, a principal point c that is zero or mid-range (pcp_to_dev), and a depth z that
// passed the float range test z_min < z < z_max, i.e. a FINITE float (zero or |z| in [2^-149, 2^128)).  Then |n| is zero
// or in [2^-401, 2^229], every intermediate of the two residual steps is a normal number, the FMA residuals are exact
// and Markstein's argument goes through; for any other z (NaN, infinite — the pixel fails the range test and its
// coordinates are never looked at) the value is unspecified.  A zero numerator may lose the sign of its zero, which no
// comparison can see.  Five FP64 instructions.
__device__ __forceinline__ double pcp_quot_valid(double n, double d, double r)
{
    const double q0 = n * r;
    double e = fma(-q0, d, n);
    const double q1 = fma(e, r, q0);
    e = fma(-q1, d, n);
    return fma(e, r, q1);
}

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
