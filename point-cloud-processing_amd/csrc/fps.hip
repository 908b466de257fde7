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
#include <pthread.h>
#include <type_traits>

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

// A cloud the launch cannot serve: CSR rows outside the caller's arrays (stage 1 reported a capacity overflow, or
// the offsets are corrupt) or more points than the caller's `max_pts` bound let any launched shape take.
__device__ __forceinline__ bool fps_cloud_out_of_bounds(long long base, long long M, long long total_pts, int m_cap)
{
    return base < 0 || M < 0 || base > total_pts || M > total_pts - base || M > (long long)m_cap;
}

// ---- brute force: one 1024-thread workgroup per cloud, min_d streamed through `md` ----------
template <int MODE>
__global__ void __launch_bounds__(1024)
fps_brute(const float* __restrict__ xyz, const long long* __restrict__ offsets, int K,
          long long* __restrict__ idx, float* __restrict__ out_xyz, float* __restrict__ md,
          long long total_pts, int* __restrict__ overflow)
{
    __shared__ unsigned long long part[2][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = offsets[b];
    const long long Mll = offsets[b + 1] - base;
    // rows outside [0, total_pts) do not exist in xyz / the workspace: such a cloud gets an all -1 row and the flag
    const bool bad = fps_cloud_out_of_bounds(base, Mll, total_pts, 0x7fffffff);
    if (bad && tid == 0 && overflow) *overflow = 1;
    const int M = bad ? 0 : (int)Mll;
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

// ---- wave primitives ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

// wave-uniform maximum of one u32 per lane (DPP butterfly, no LDS traffic)
__device__ __forceinline__ unsigned wave_umax_u32(unsigned v)
{
    v = max(v, dpp_u32<0xB1>(v));          // quad_perm [1,0,3,2]
    v = max(v, dpp_u32<0x4E>(v));          // quad_perm [2,3,0,1]
    v = max(v, dpp_u32<0x141>(v));         // row_half_mirror
    v = max(v, dpp_u32<0x140>(v));         // row_mirror: every lane of a row holds the row max
    v = max(v, dpp_u32<0x142, 0xA>(v));    // row_bcast15 → rows 1,3
    v = max(v, dpp_u32<0x143, 0xC>(v));    // row_bcast31 → rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Wave arg-max of one u64 key per lane.  Returns the wave-uniform maximum key and, in *src,
// the (uniform) lane that holds it.  Keys of real points are unique (they embed the index),
// so the second reduction over the low word only runs when two lanes tie on the value.
__device__ __forceinline__ unsigned long long wave_argmax_key(unsigned long long k, int* src)
{
    const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    const unsigned m = wave_umax_u32(hi);
    unsigned long long hit = __ballot(hi == m);
    unsigned l;
    if (__popcll(hit) == 1) {                              // wave-uniform branch, the common case
        *src = __ffsll((long long)hit) - 1;
        l = (unsigned)__builtin_amdgcn_readlane((int)lo, *src);
    } else {
        l = wave_umax_u32(hi == m ? lo : 0u);
        hit = __ballot(hi == m && lo == l);
        *src = __ffsll((long long)hit) - 1;
    }
    return ((unsigned long long)m << 32) | l;
}

// G independent wave maxima at once: the G DPP chains interleave, so the two wait states every
// DPP read needs after a VALU write are filled with useful work instead of s_nop.
template <int G>
__device__ __forceinline__ void wave_umax_u32_x(unsigned (&v)[G])
{
#define PCP_DPP_STEP(CTRL, RM)                                              \
    {                                                                       \
        unsigned t_[G];                                                     \
        _Pragma("unroll") for (int u = 0; u < G; ++u) t_[u] = dpp_u32<CTRL, RM>(v[u]); \
        _Pragma("unroll") for (int u = 0; u < G; ++u) v[u] = max(v[u], t_[u]);         \
    }
    PCP_DPP_STEP(0xB1, 0xf)
    PCP_DPP_STEP(0x4E, 0xf)
    PCP_DPP_STEP(0x141, 0xf)
    PCP_DPP_STEP(0x140, 0xf)
    PCP_DPP_STEP(0x142, 0xA)
    PCP_DPP_STEP(0x143, 0xC)
#undef PCP_DPP_STEP
#pragma unroll
    for (int u = 0; u < G; ++u) v[u] = (unsigned)__builtin_amdgcn_readlane((int)v[u], 63);
}

// G wave arg-maxes at once (see wave_argmax_key); every key set must hold at least one non-zero key
template <int G>
__device__ __forceinline__ void wave_argmax_key_x(const unsigned long long (&k)[G],
                                                  unsigned long long (&wk)[G], int (&src)[G])
{
    unsigned m[G];
    unsigned long long hit[G];
    bool simple = true;
#pragma unroll
    for (int u = 0; u < G; ++u) m[u] = (unsigned)(k[u] >> 32);
    wave_umax_u32_x<G>(m);
#pragma unroll
    for (int u = 0; u < G; ++u) {
        hit[u] = __ballot((unsigned)(k[u] >> 32) == m[u]);
        simple = simple && __popcll(hit[u]) == 1;
    }
    if (simple) {                                          // wave-uniform, the common case
#pragma unroll
        for (int u = 0; u < G; ++u) {
            src[u] = __ffsll((long long)hit[u]) - 1;
            wk[u] = ((unsigned long long)m[u] << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k[u], src[u]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < G; ++u) wk[u] = wave_argmax_key(k[u], &src[u]);
    }
}

// Wave-uniform TOP TWO of the multiset union of one (a1 >= a2) pair per lane: a1 = the largest value, a2 = the second
// largest (equal to a1 when the maximum occurs twice).  Same DPP butterfly as wave_umax_u32; every step merges two
// disjoint lane sets, so the multiset semantics are exact.  Lanes a row-masked step leaves out read 0, the identity.
__device__ __forceinline__ void wave_top2_u32(unsigned& a1, unsigned& a2)
{
#define PCP_TOP2_STEP(CTRL, RM)                                             \
    {                                                                       \
        const unsigned t1_ = dpp_u32<CTRL, RM>(a1), t2_ = dpp_u32<CTRL, RM>(a2); \
        const unsigned lo_ = min(a1, t1_);                                  \
        a1 = max(a1, t1_);                                                  \
        a2 = max(max(a2, t2_), lo_);                                        \
    }
    PCP_TOP2_STEP(0xB1, 0xf)
    PCP_TOP2_STEP(0x4E, 0xf)
    PCP_TOP2_STEP(0x141, 0xf)
    PCP_TOP2_STEP(0x140, 0xf)
    PCP_TOP2_STEP(0x142, 0xA)
    PCP_TOP2_STEP(0x143, 0xC)
#undef PCP_TOP2_STEP
    a1 = (unsigned)__builtin_amdgcn_readlane((int)a1, 63);
    a2 = (unsigned)__builtin_amdgcn_readlane((int)a2, 63);
}

// G of them at once, chains interleaved (see wave_umax_u32_x)
template <int G>
__device__ __forceinline__ void wave_top2_u32_x(unsigned (&a1)[G], unsigned (&a2)[G])
{
#define PCP_TOP2_STEP(CTRL, RM)                                             \
    {                                                                       \
        unsigned t1_[G], t2_[G];                                            \
        _Pragma("unroll") for (int u = 0; u < G; ++u) { t1_[u] = dpp_u32<CTRL, RM>(a1[u]); t2_[u] = dpp_u32<CTRL, RM>(a2[u]); } \
        _Pragma("unroll") for (int u = 0; u < G; ++u) {                     \
            const unsigned lo_ = min(a1[u], t1_[u]);                        \
            a1[u] = max(a1[u], t1_[u]);                                     \
            a2[u] = max(max(a2[u], t2_[u]), lo_);                           \
        }                                                                   \
    }
    PCP_TOP2_STEP(0xB1, 0xf)
    PCP_TOP2_STEP(0x4E, 0xf)
    PCP_TOP2_STEP(0x141, 0xf)
    PCP_TOP2_STEP(0x140, 0xf)
    PCP_TOP2_STEP(0x142, 0xA)
    PCP_TOP2_STEP(0x143, 0xC)
#undef PCP_TOP2_STEP
#pragma unroll
    for (int u = 0; u < G; ++u) {
        a1[u] = (unsigned)__builtin_amdgcn_readlane((int)a1[u], 63);
        a2[u] = (unsigned)__builtin_amdgcn_readlane((int)a2[u], 63);
    }
}

// G wave arg-maxes at once that also return the RUNNER-UP VALUE of each key set (the largest value among the other
// lanes' keys and the per-lane runner-up values r[]; equal to the top value on a tie).  Every key set must hold at
// least one non-zero key.
template <int G>
__device__ __forceinline__ void wave_argmax2_key_x(const unsigned long long (&k)[G], const unsigned (&r)[G],
                                                   unsigned long long (&wk)[G], unsigned (&ws)[G], int (&src)[G])
{
    unsigned m[G], m2[G];
#pragma unroll
    for (int u = 0; u < G; ++u) { m[u] = (unsigned)(k[u] >> 32); m2[u] = r[u]; }
    wave_top2_u32_x<G>(m, m2);
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const unsigned h = (unsigned)(k[u] >> 32), lo = (unsigned)k[u];
        unsigned long long hit = __ballot(h == m[u]);
        unsigned l;
        if (__popcll(hit) == 1) {                              // wave-uniform, the common case
            src[u] = __ffsll((long long)hit) - 1;
            l = (unsigned)__builtin_amdgcn_readlane((int)lo, src[u]);
        } else {
            l = wave_umax_u32(h == m[u] ? lo : 0u);
            hit = __ballot(h == m[u] && lo == l);
            src[u] = __ffsll((long long)hit) - 1;
        }
        wk[u] = ((unsigned long long)m[u] << 32) | l;
        ws[u] = m2[u];
    }
}

// arg-max over the keys held by lanes 0..N-1 (N = 4, 8 or 16).  The reduction stays inside row 0 and the ballots
// are masked to the low N lanes, so the other lanes may hold anything (no exec masking around the loads).
template <int N>
__device__ __forceinline__ unsigned long long lowlanes_argmax_key(unsigned long long k, int* src)
{
    constexpr unsigned long long LOW = (1ull << N) - 1ull;
    unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    unsigned m = hi;
    m = max(m, dpp_u32<0xB1>(m));
    m = max(m, dpp_u32<0x4E>(m));
    if (N > 4) m = max(m, dpp_u32<0x141>(m));
    if (N > 8) m = max(m, dpp_u32<0x140>(m));
    m = (unsigned)__builtin_amdgcn_readlane((int)m, 0);
    unsigned long long hit = __ballot(hi == m) & LOW;
    unsigned l;
    if (__popcll(hit) == 1) {
        *src = __ffsll((long long)hit) - 1;
        l = (unsigned)__builtin_amdgcn_readlane((int)lo, *src);
    } else {
        unsigned t = hi == m ? lo : 0u;
        t = max(t, dpp_u32<0xB1>(t));
        t = max(t, dpp_u32<0x4E>(t));
        if (N > 4) t = max(t, dpp_u32<0x141>(t));
        if (N > 8) t = max(t, dpp_u32<0x140>(t));
        l = (unsigned)__builtin_amdgcn_readlane((int)t, 0);
        hit = __ballot(hi == m && lo == l) & LOW;
        *src = __ffsll((long long)hit) - 1;
    }
    return ((unsigned long long)m << 32) | l;
}

__device__ __forceinline__ float wave_min_f32(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float rl(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Diagnostic build only (make stamps): per-phase cycle sums printed by workgroup 0.
#ifdef PCP_FPS_STAMPS
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

// ---- exact bucket-pruned FPS ------------------------------------------------------------------
// Same picks as the brute-force scan with ≈40·M instead of (K-1)·M distance evaluations on
// the benchmark frames.  One workgroup owns one cloud.
//
//   prologue  bounding box → 4096-cell Morton grid → LDS counting sort of the points into
//             {x,y,z,orig_idx} records (+ one min_d float per point) → BUCKETS of BS consecutive
//             sorted rows (BS = 64 when the cloud has ≤ 64·T·CPL points).  Bucket j belongs to
//             wave j % NW, lane (j / NW) % 64, slot j / (64·NW); the owning lane keeps the
//             bucket's tight box, its arg-max key and that point's xyz in registers.
//   round     every lane tests its buckets: lb = dist2(s, clamp(s, box)).  lb is computed with
//             the same rounded operation sequence as a point distance, and rounding is
//             monotone, so lb <= dist2(s,p) for every p in the box: a bucket with
//             lb >= bucket_max cannot change and is skipped.  A ballot gives the wave its own
//             active buckets; it streams them four at a time (one coalesced 1 KB + 256 B load
//             per bucket), updates min_d in place and hands the new arg-max back to the owning
//             lane with readlane — no queue, no LDS atomics.  One wave arg-max over the lanes'
//             keys, one LDS partial per wave, ONE workgroup barrier, and every wave reduces the
//             NW partials to the next sample (xyz included, so nothing global is on the
//             critical path).
// LDS slot of bucket j's arg-max point: wave-major (bucket j belongs to wave j % NW, its (j / NW)-th), so that the owning
// lanes of one wave read consecutive 16-byte entries — the bucket-major layout put all 64 of them on the same four banks
#ifndef PCP_FPS_MAXP
#define PCP_FPS_MAXP 4           // picks one multi-pick round may certify.  8 was measured (CPU model: 296 instead of 350 rounds per 1,023
                                 // picks): bit-exact, but every round pays for eight samples' worth of registers and unrolled code (108
                                 // scalar spills in the sliced kernel) - 300 clouds 2.54 instead of 2.40 ms, 1 cloud 1.51 / 1.48, 256: 2.01 / 1.94
#endif
// control words of the quantum-sliced launches (below), also read by the resident multi-pick kernel
#define PCP_FPS_SLICE_MAX_SLOTS 256
#define PCP_FPS_SLICE_CTL_WORDS (PCP_FPS_SLICE_MAX_SLOTS + 64)    // per wide shape: the waiting slots, then the number of clouds the shape serves
#define PCP_BX(j, NW, PER_WAVE) (((j) % (NW)) * (PER_WAVE) + (j) / (NW))
#define PCP_FPS_RED 128          // floats of LDS scratch: bounding-box partials (6 per wave, <= 16 waves), scan scratch
#define PCP_FPS_CELL_BITS 12
#define PCP_FPS_NCELL (1 << PCP_FPS_CELL_BITS)
#ifndef PCP_FPS_HILBERT
#define PCP_FPS_HILBERT 1
#endif

#include "hilbert_lut.h"

struct MortonPlan {          // cell id = deposit of each axis' quantised coordinate into its id-bit mask
    int hilbert;             // 0: adaptive Morton; 1: 16^3 grid along the 3-D Hilbert curve; 2: 64^2 grid over the
                             //    two long axes (ax[0], ax[1]) along the 2-D Hilbert curve (LUTs in hilbert_lut.h)
    int ax[2];
    unsigned mask[3];
    int qmax[3];
    float scale[3];
    float lo[3];
};

__device__ __forceinline__ int morton_cell(const MortonPlan& mp, float x, float y, float z)
{
    const float v[3] = {x, y, z};
    unsigned id = 0;
    if (mp.hilbert == 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            id = (id << 4) | (unsigned)min(15, max(0, (int)((v[c] - mp.lo[c]) * mp.scale[c])));
        return (int)pcp_hilbert_lut[id];
    }
    if (mp.hilbert == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned q = (unsigned)min(63, max(0, (int)((v[c] - mp.lo[c]) * mp.scale[c])));
            id |= (c == mp.ax[0]) ? (q << 6) : ((c == mp.ax[1]) ? q : 0u);
        }
        return (int)pcp_hilbert2_lut[id];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        unsigned q = (unsigned)min(mp.qmax[c], max(0, (int)((v[c] - mp.lo[c]) * mp.scale[c])));
        for (unsigned m = mp.mask[c]; m; m &= m - 1) {      // wave-uniform trip count
            id |= (q & 1u) ? (m & (0u - m)) : 0u;
            q >>= 1;
        }
    }
    return (int)id;
}

// Prologue shared by the bucket kernels: bounding box → 4096-cell Morton grid → LDS counting sort of
// the cloud's points into S = {x,y,z,bits(orig_idx)} records, D = FLT_MAX.  Ends with a barrier.
template <int T>
__device__ __forceinline__ void fps_sort_points(const float* __restrict__ P, int M, float4* __restrict__ S,
                                                float* __restrict__ D, int* cellcnt, float* red, int tid,
                                                int lane, int wave)
{
    constexpr int NW = T / 64, NCELL = PCP_FPS_NCELL;
    static_assert(NCELL % T == 0, "cells per thread must be integral");
#ifdef PCP_FPS_STAMPS
    unsigned long long q0, q1, q2, q3, q4;
    STAMP(q0);
#endif
    // ---- prologue 1: bounding box -------------------------------------------------------------
    constexpr int PU = 8;                               // points in flight per thread in the prologue
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int p0 = tid; p0 < M; p0 += T * PU) {
        float v[PU][3];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int p = min(p0 + u * T, M - 1);           // clamped duplicates do not change min/max
#pragma unroll
            for (int c = 0; c < 3; ++c) v[u][c] = P[3 * p + c];
        }
#pragma unroll
        for (int u = 0; u < PU; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], v[u][c]); hi[c] = fmaxf(hi[c], v[u][c]); }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = wave_min_f32(lo[c]); hi[c] = wave_max_f32(hi[c]); }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { red[wave * 6 + c] = lo[c]; red[wave * 6 + 3 + c] = hi[c]; }
    }
    for (int c = tid; c < NCELL; c += T) cellcnt[c] = 0;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        lo[c] = red[c]; hi[c] = red[3 + c];
        for (int w = 1; w < NW; ++w) { lo[c] = fminf(lo[c], red[w * 6 + c]); hi[c] = fmaxf(hi[c], red[w * 6 + 3 + c]); }
    }
    // Morton plan: from the top id bit down, give each bit to the axis whose cells are longest;
    // low id bits are fine coordinate bits, so consecutive ids are spatial neighbours
    MortonPlan mp;
    {
        float e[3];
        int bits[3] = {0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            e[c] = hi[c] - lo[c];
            if (!(e[c] > 0.f) || !(e[c] < INFINITY)) e[c] = 0.f;
            mp.lo[c] = lo[c];
            mp.mask[c] = 0u;
        }
        for (int t = PCP_FPS_CELL_BITS - 1; t >= 0; --t) {
            const float l0 = e[0] / (float)(1 << bits[0]), l1 = e[1] / (float)(1 << bits[1]),
                        l2 = e[2] / (float)(1 << bits[2]);
            if (!(l0 > 0.f) && !(l1 > 0.f) && !(l2 > 0.f)) break;
            if (l0 >= l1 && l0 >= l2) { mp.mask[0] |= 1u << t; bits[0]++; }
            else if (l1 >= l2) { mp.mask[1] |= 1u << t; bits[1]++; }
            else { mp.mask[2] |= 1u << t; bits[2]++; }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mp.qmax[c] = (1 << bits[c]) - 1;
            mp.scale[c] = e[c] > 0.f ? (float)(1 << bits[c]) / e[c] : 0.f;
        }
        // Hilbert order where the plan allows it: consecutive cells are always face neighbours, so 64-row
        // buckets are more compact than along the Morton order (measured: 22 % fewer active buckets per
        // round on the bench clouds).  Every axis with >= 3 bits: 16^3 cells, 3-D curve.  A flat cloud
        // (one axis with <= 1 bit, the others >= 5): 64^2 cells over the long axes, 2-D curve.  Anything
        // else keeps the adaptive Morton plan.
        mp.hilbert = 0; mp.ax[0] = 0; mp.ax[1] = 1;
        if (PCP_FPS_HILBERT) {
            const int bmin = min(bits[0], min(bits[1], bits[2]));
            const int cmin = bits[0] == bmin ? 0 : (bits[1] == bmin ? 1 : 2);
            const int ca = cmin == 0 ? 1 : 0, cb = cmin == 2 ? 1 : 2;
            if (bmin >= 3) {
                mp.hilbert = 1;
#pragma unroll
                for (int c = 0; c < 3; ++c) mp.scale[c] = e[c] > 0.f ? 16.f / e[c] : 0.f;
            } else if (bmin <= 1 && bits[ca] >= 5 && bits[cb] >= 5) {
                mp.hilbert = 2; mp.ax[0] = ca; mp.ax[1] = cb;
#pragma unroll
                for (int c = 0; c < 3; ++c) mp.scale[c] = (c != cmin && e[c] > 0.f) ? 64.f / e[c] : 0.f;
            }
        }
    }

#if defined(PCP_FPS_STOP_AFTER) && PCP_FPS_STOP_AFTER == 1
    return;
#endif
#ifdef PCP_FPS_STAMPS
    STAMP(q1);
#endif
    // ---- prologue 2: histogram, scan, scatter (LDS counting sort by Morton cell) ------------------
    for (int p0 = tid; p0 < M; p0 += T * PU) {
        float v[PU][3];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int p = min(p0 + u * T, M - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[u][c] = P[3 * p + c];
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int cell = morton_cell(mp, v[u][0], v[u][1], v[u][2]);
            if (p0 + u * T < M) atomicAdd(&cellcnt[cell], 1);
        }
    }
    __syncthreads();
    {
        constexpr int CPT = NCELL / T;
        int local[CPT];
        int sum = 0;
#pragma unroll
        for (int j = 0; j < CPT; ++j) { local[j] = cellcnt[tid * CPT + j]; sum += local[j]; }
        int total;
        int ex = pcp_block_excl_scan(sum, (int*)red, &total);
#pragma unroll
        for (int j = 0; j < CPT; ++j) { cellcnt[tid * CPT + j] = ex; ex += local[j]; }
    }
    __syncthreads();
#if defined(PCP_FPS_STOP_AFTER) && PCP_FPS_STOP_AFTER == 2
    return;
#endif
#ifdef PCP_FPS_STAMPS
    STAMP(q2);
#endif
    {
        // coalesced fill instead of a scattered one, 16 bytes per lane where D allows it (D is only 4-byte aligned:
        // head and tail go by single floats) — a quarter of the store instructions of a float-per-lane loop
        const int head = min(M, (int)((16u - ((unsigned)(uintptr_t)D & 15u)) & 15u) >> 2);
        const int body = (M - head) >> 2;
        if (tid < head) D[tid] = FLT_MAX;
        float4* D4 = (float4*)(D + head);
        for (int p = tid; p < body; p += T) D4[p] = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX);
        const int done = head + 4 * body;
        if (tid < M - done) D[done + tid] = FLT_MAX;
    }
    for (int p0 = tid; p0 < M; p0 += T * PU) {
        float v[PU][3];
        int pos[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int p = min(p0 + u * T, M - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[u][c] = P[3 * p + c];
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int cell = morton_cell(mp, v[u][0], v[u][1], v[u][2]);
            pos[u] = (p0 + u * T < M) ? atomicAdd(&cellcnt[cell], 1) : -1;
        }
#pragma unroll
        for (int u = 0; u < PU; ++u)
            if (pos[u] >= 0) {
                S[pos[u]] = make_float4(v[u][0], v[u][1], v[u][2], __int_as_float(p0 + u * T));
            }
    }
    __syncthreads();
#ifdef PCP_FPS_STAMPS
    STAMP(q3);
    if (blockIdx.x == 0 && tid == 0)
        printf("[stamps] prologue: bbox=%llu hist+scan=%llu scatter=%llu cycles\n", q1 - q0, q2 - q1, q3 - q2);
    (void)q4;
#endif
}

// ---- deferred bucket updates for the early, bandwidth-bound rounds ---------------------------------
// Same picks as the eager round loop of fps_bucket_body, about half the bucket loads.  A bucket the new
// sample could change (lb < bucket_max) is not streamed at once: the sample's round number joins the
// bucket's PENDING list (up to three 10-bit round numbers + a 2-bit count in one register of the owning
// lane) and the bucket's key stays what it was — now an UPPER BOUND of its arg-max key, because min_d
// only ever decreases.  A bucket is streamed ("flushed": every pending sample applied in one pass over
// its rows, the samples read back from an LDS table of the picks so far) when its list is full or when
// its upper bound reaches tau = ALPHA x (the previous round's winning distance), i.e. when it could win
// this round.  After the barrier the winner among the CLEAN buckets (no pending sample: key exact) is
// certain if its distance is >= tau, since every bucket still pending has an upper bound < tau.
// Otherwise (a few % of the rounds) tau drops to that distance, the pending buckets at or above it are
// flushed, and the second pick is certain for the same reason.  The winning distances never increase
// from round to round, so ALPHA only trades early flushes against second iterations; it cannot change a
// pick.  Fewer loads, but a longer instruction chain per flush: it pays while the rounds are bound by
// fabric bandwidth (many clouds resident, the first few hundred rounds) and not after, so the host
// passes the number of rounds to run this way and the eager loop takes over from there.
#ifndef PCP_FPS_MP_DEFER_MIN_B
#define PCP_FPS_MP_DEFER_MIN_B 128   // multi-pick rounds defer bucket updates that leave the key exact from this many clouds per launch
#endif
#ifndef PCP_FPS_LAZY_ALPHA
#define PCP_FPS_LAZY_ALPHA 0.95f
#endif
#ifndef PCP_FPS_LAZY_ROUNDS
#define PCP_FPS_LAZY_ROUNDS 256          // rounds run with deferred updates ...
#endif
#ifndef PCP_FPS_LAZY_MIN_B
#define PCP_FPS_LAZY_MIN_B 192           // ... for batches of at least this many clouds
#endif
#define PCP_FPS_LAZY_MAX_ROUNDS 1024     // 10-bit round numbers; the sample table reuses the 16 KB cell histogram

// Runs rounds 0..ks-1 (ks <= kn-1, ks <= MAX_ROUNDS) for a single-chunk, one-bucket-per-lane workgroup and
// leaves every bucket clean.  On return (sx,sy,sz,last) is pick ks, not yet written to the outputs.
template <int T, int MODE, int BH>
__device__ __forceinline__ void
fps_lazy_rounds(const int ks, const int M, const int NB, float4* __restrict__ S, float* __restrict__ D,
                float4* samp, unsigned long long* pkey, float4* pxyz, float4* bxyz, long long* __restrict__ irow,
                float* __restrict__ orow, const int tid, const int lane, const int wave, unsigned long long& key,
                const float (&blo)[3], const float (&bhi)[3], float& sx, float& sy, float& sz, int& last,
                unsigned long long& wmax, int& wmax_j)
{
    constexpr int NW = T / 64, PEND = 3, BS = 64 * BH;
    unsigned pend = 0u;                                         // [count:2][round:10][round:10][round:10]
    float tau = 0.f;                                            // round 0 flushes every bucket
    bool fresh = true;                                          // a new sample was picked: test + append
    bool draining = false;                                      // last trip: flush whatever is still pending
    for (int k = 0, it = 0;; ++it) {
        const int par = it & 1;
        bool need_idle = false;
        if (fresh) {
            if (k >= ks) { draining = true; tau = 0.f; }
            else {
                if (tid == 0) {
                    irow[k] = last;
                    if (orow) { orow[3 * k] = sx; orow[3 * k + 1] = sy; orow[3 * k + 2] = sz; }
                }
                // every wave writes the (same) table entry itself: a wave's LDS accesses stay in program
                // order, so its own flushes read it back without a barrier
                if (lane == 0) samp[k] = make_float4(sx, sy, sz, 0.f);
                const float cmax = __uint_as_float((unsigned)(key >> 32));
                const float qx = __builtin_amdgcn_fmed3f(sx, blo[0], bhi[0]);
                const float qy = __builtin_amdgcn_fmed3f(sy, blo[1], bhi[1]);
                const float qz = __builtin_amdgcn_fmed3f(sz, blo[2], bhi[2]);
                const bool aff = dist2<MODE>(sx, sy, sz, qx, qy, qz) < cmax;
                if (aff) {
                    const unsigned np = pend >> 30;             // < PEND: a full list was flushed in its round
                    pend = (pend & 0x3fffffffu) | ((unsigned)k << (10u * np)) | ((np + 1u) << 30);
                    // Tighten the bucket's upper bound without touching its rows: after this sample no point of the
                    // bucket is farther from the sample set than the box's farthest corner is from the sample (same
                    // rounded operation sequence as a point distance, and rounding is monotone, so the bound holds
                    // in f32 too).  A pending bucket is not a candidate, so the high word of its key is free to
                    // carry the bound; the flush rewrites the whole key.  Far fewer buckets then "could win" and have
                    // to be streamed in the early rounds.  (A bucket holding a NaN point has an infinite box, see
                    // the prologue: its bound never tightens, like that point's min-distance.)
                    const float fx = fabsf(sx - blo[0]) >= fabsf(sx - bhi[0]) ? blo[0] : bhi[0];
                    const float fy = fabsf(sy - blo[1]) >= fabsf(sy - bhi[1]) ? blo[1] : bhi[1];
                    const float fz = fabsf(sz - blo[2]) >= fabsf(sz - bhi[2]) ? blo[2] : bhi[2];
                    const float ub = fminf(cmax, dist2<MODE>(sx, sy, sz, fx, fy, fz));
                    key = ((unsigned long long)__float_as_uint(ub) << 32) | (key & 0xffffffffull);
                }
                need_idle = (__ballot(aff) >> ((wmax_j / NW) & 63)) & 1ull;  // the wave's clean top is no longer clean
            }
        }
        const unsigned long long amask =
            __ballot((pend >> 30) != 0u && (__uint_as_float((unsigned)(key >> 32)) >= tau || (pend >> 30) == (unsigned)PEND));
        unsigned long long cand = wmax;
        int cand_j = wmax_j;
        auto idle_argmax = [&]() {
            if (!need_idle) return;                             // wave-uniform
            int src;
            cand = wave_argmax_key(pend == 0u ? key : 0ull, &src);           // flushing lanes still hold pend != 0
            cand_j = src * NW + wave;
        };
        auto run_pass = [&](auto GC, auto FC, const int (&ol)[4], bool first) {
            constexpr int G = decltype(GC)::value;
            constexpr bool FULL = decltype(FC)::value;
            int r0[G];
            unsigned pw[G];
            float4 pt[G];
            float mm[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                r0[u] = (ol[u] * NW + wave) * BS + (u % BH) * 64;                     // BH > 1: BH consecutive slots = one bucket's parts
                if (FULL) {
                    pt[u] = (S + r0[u])[lane];
                    mm[u] = (D + r0[u])[lane];
                } else {
                    const int r = min(r0[u] + lane, M - 1);
                    pt[u] = S[r];
                    mm[u] = D[r];
                }
                pw[u] = (unsigned)__builtin_amdgcn_readlane((int)pend, ol[u]);
            }
            // lane 4u+i fetches pending sample i of bucket u (slots past the count read some valid entry, unused)
            unsigned w = pw[0];
#pragma unroll
            for (int u = 1; u < G; ++u) w = (lane >> 2) == u ? pw[u] : w;
            const float4 sm = samp[(w >> (10u * (unsigned)(lane & 3))) & 1023u];
            if (first) idle_argmax();                           // VALU work under the memory latency
            unsigned long long bk[G];
#pragma unroll
            for (int u = 0; u < G; ++u)
                asm volatile("" : "+v"(mm[u]), "+v"(pt[u].w));
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int r = r0[u] + lane;
                const bool live = FULL || r < M;
                const unsigned np = pw[u] >> 30;                // wave-uniform, 1..PEND
                float m = mm[u];
#pragma unroll
                for (int i = 0; i < PEND; ++i) {
                    if (i < (int)np)                            // fminf keeps m when d is NaN, like the `d < m` update
                        m = fminf(m, dist2<MODE>(rl(sm.x, 4 * u + i), rl(sm.y, 4 * u + i), rl(sm.z, 4 * u + i),
                                                 pt[u].x, pt[u].y, pt[u].z));
                }
                if (live && m < mm[u]) D[r] = m;
                bk[u] = live ? make_key(m, (unsigned)__float_as_int(pt[u].w)) : 0ull;
            }
            unsigned long long wk[G];
            int src[G];
            wave_argmax_key_x<G>(bk, wk, src);
            if constexpr (BH > 1 && G < BH) {
                // never called: a split bucket occupies BH slots
            } else if constexpr (BH > 1) {
#pragma unroll
                for (int q = 0; q < G; q += BH) {               // the bucket's key is the largest of its parts' keys
                    const int j = ol[q] * NW + wave;
                    int w = q;                                  // scalar
                    unsigned long long wb = wk[q];
#pragma unroll
                    for (int h = 1; h < BH; ++h)
                        if (wk[q + h] > wb) { wb = wk[q + h]; w = q + h; }
#pragma unroll
                    for (int h = 0; h < BH; ++h)
                        if (w == q + h && lane == src[q + h]) bxyz[PCP_BX(j, NW, 64)] = pt[q + h];
                    if (lane == ol[q]) { key = wb; pend = 0u; }
                    if (wb > cand) { cand = wb; cand_j = j; }
                }
            } else {
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int j = ol[u] * NW + wave;
                    if (lane == src[u]) bxyz[PCP_BX(j, NW, 64)] = pt[u];
                    if (lane == ol[u]) { key = wk[u]; pend = 0u; }
                    if (wk[u] > cand) { cand = wk[u]; cand_j = j; }      // scalar
                }
            }
        };
        bool first = true;
        __builtin_amdgcn_s_setprio(3);
        {
            unsigned long long m = amask;
            while (m) {                                         // wave-uniform
                int ol[4], n = 0;
                if constexpr (BH > 1) {
#pragma unroll
                    for (int u = 0; u < 4; u += BH) {
                        if (m) { ol[u] = __ffsll((long long)m) - 1; m &= m - 1; n = u + BH; }
                        else ol[u] = ol[0];
#pragma unroll
                        for (int h = 1; h < BH; ++h) ol[u + h] = ol[u];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (m) { ol[u] = __ffsll((long long)m) - 1; m &= m - 1; n = u + 1; }
                        else ol[u] = ol[0];
                    }
                }
                using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                using I4 = std::integral_constant<int, 4>;
                const bool full = (M % BS == 0) || (((NB - 1) % NW) != wave) || !((amask >> (((NB - 1) / NW) & 63)) & 1ull);
                if (full) {
                    if (n == 1) run_pass(I1{}, std::true_type{}, ol, first);
                    else if (n == 2) run_pass(I2{}, std::true_type{}, ol, first);
                    else run_pass(I4{}, std::true_type{}, ol, first);
                } else {
                    run_pass(I4{}, std::false_type{}, ol, first);
                }
                first = false;
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (first) idle_argmax();
        wmax = cand;
        wmax_j = cand_j;
        if (draining) break;                                    // every bucket clean, each wave's top exact: no pick needed
        if (lane == 0) { pkey[par * NW + wave] = cand; pxyz[par * NW + wave] = bxyz[PCP_BX(cand_j, NW, 64)]; }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // see the eager loop
        {
            const unsigned long long kk = pkey[par * NW + (lane & (NW - 1))];
            const float4 cc = pxyz[par * NW + (lane & (NW - 1))];
            int src;
            const unsigned long long wk = lowlanes_argmax_key<NW>(kk, &src);
            const float v = __uint_as_float((unsigned)(wk >> 32));
            if (!fresh || v >= tau) {                           // certain: every pending bucket is bounded below tau <= v
                last = wk ? (int)(~(unsigned)wk) : 0;
                sx = rl(cc.x, src); sy = rl(cc.y, src); sz = rl(cc.z, src);
                tau = PCP_FPS_LAZY_ALPHA * v;
                fresh = true;
                ++k;
            } else {                                            // flush the pending buckets that could still beat v
                tau = v;
                fresh = false;
            }
        }
    }
    __syncthreads();                                            // the eager loop restarts the partials' parity
}

// ---- multi-pick rounds ----------------------------------------------------------------------------------------------
// One workgroup round applies the n <= 4 newest samples to every bucket any of them could change (one pass over a bucket's
// rows for all of them) and then certifies up to four further picks at once:
//   every wave publishes its best point (key, xyz) and `rub`, an upper bound of every OTHER min-distance it owns (the
//   second-largest value of the multiset {bucket top, bucket runner-up} over its buckets).  Let a1 > a2 > ... be the
//   wave tops in key order, from waves w1, w2, ...  a1 is the next pick, as ever.  a_i is certainly the pick after
//   a_(i-1) when for every earlier pick a_j of this round
//       rub(w_j) < value(a_i)             nothing else in an earlier winner's wave can overtake a_i,
//       !(dist2(a_j, a_i) < value(a_i))   applying a_j leaves a_i's min-distance as it is,
//       a_j is finite                     its own min-distance drops to 0 once it is applied,
//   and value(a_i) differs from every other wave top (a tie on the value is resolved by index order, by a later round's
//   first pick): all other points only lose min-distance, so a_i stays the arg-max.
// min() is exact and commutative, so applying a round's samples together leaves the same min-distances as applying them
// one by one; a bucket test against the older, larger bucket maximum only streams a superset.  The picks are therefore
// the sequential ones, in order — in a third of the workgroup rounds on the bench clouds
// (tests/model_fps_multipick.py: 350 rounds for 1,023 picks).
// Instruction diet (a round keeps all sixteen waves busy, so it is bound by instruction issue, not by its chain):
//   * a streamed bucket is REDUCED only if a sample lowers the min-distance of its arg-max row (bxyz[j]): otherwise its
//     key cannot change (every other row only loses min-distance); its runner-up value may go stale-high, which is safe;
//   * a row is updated only with the samples whose test the bucket failed (per-bucket sample bits);
//   * a wave rebuilds its partial only when a reduced bucket could be its top or runner-up (old key >= rub);
//   * the picks are certified in parallel: every wave ranks its own top among the sixteen (one ballot), the up to four
//     best check themselves against the better ones (lanes = waves), write their verdict, and after a second barrier
//     every wave reads the verdicts: n = 1 + the number of leading certified ranks.
// What the rounds carry from one round to the next besides the bucket keys: a quantum-sliced launch (below) stops the
// rounds at a pick number, saves this, and another workgroup goes on from it.
struct FpsMpCarry {
    unsigned pend;               // the lane's bucket: pending pick numbers (deferred updates)
    unsigned sec;                // the lane's bucket: runner-up value or an upper bound of it
    int n;                       // picks made by the last round = samples the next round applies
    float4 smv;                  // lane i < n: sample i
};

template <int T, int MODE, int BH, bool DEFER, bool TIMED = false, int MAXP = 4>
__device__ __forceinline__ void
fps_multipick_rounds(int& kd, const int k_stop, const unsigned long long t_stop, const int kn, const int M, const int NB, float4* __restrict__ S, float* __restrict__ D,
                     unsigned long long* pkey, float4* pxyz, float4* bxyz, float4* sel, int* okf, float4* ring,
                     long long* __restrict__ irow, float* __restrict__ orow, const int tid, const int lane, const int wave,
                     unsigned long long& key, const float (&blo)[3], const float (&bhi)[3], FpsMpCarry& cy)
{
    constexpr int NW = T / 64, BS = 64 * BH;
    constexpr unsigned long long LOW = (1ull << NW) - 1ull;
    // DEFERRED UPDATES WITH EXACT KEYS.  A sample that could lower a min-distance of a bucket (it fails the box test) but
    // does not lower the one of its arg-max ROW leaves the bucket's key exact — every other row only loses min-distance — so
    // the bucket stays a candidate as it is and its rows need not be touched now: the sample's pick number joins the
    // bucket's PENDING list (three 10-bit pick numbers + a count in one register of the owning lane; the picks' coordinates
    // sit in a 1,024-entry LDS ring).  The rows are streamed — all pending samples and the new ones applied in one pass —
    // when a sample does reach the arg-max row (then the bucket is reduced too), when the list is full, or before a pending
    // pick number would be overwritten in the ring.  A sample that passes the box test of a bucket can never matter to it
    // later either: its distance to every row is at least the bucket's maximum of that moment.  On the bench clouds this
    // halves the buckets streamed per pick (CPU model: 15.3 instead of 28.0, 7.5 instead of 17.3 in the late rounds).
    // It pays where the fabric is busy — 256 clouds 2.14 -> 1.93 ms, 512 clouds 3.51 -> 3.36 — and costs a little where one
    // cloud has a CU and its memory path to itself (1 cloud 1.49 -> 1.55 ms: the test phase and the passes grow by the list
    // handling), so the host turns it on from PCP_FPS_MP_DEFER_MIN_B clouds per launch.
    constexpr int RING = 1024;
    unsigned pend = cy.pend;                                     // [count:2][pick:10][pick:10][pick:10], oldest first
    float4 smv = cy.smv;                                         // lane i < n: sample i of the round
    unsigned sec = cy.sec;                                       // runner-up VALUE of the bucket, or any upper bound of it
    int n = cy.n;
    static_assert(MAXP == 4 || MAXP == 8, "picks per round: the verdict tables are indexed by lane & (MAXP - 1)");
    float s[MAXP][3];                                            // the round's samples (wave-uniform)
#pragma unroll
    for (int i = 0; i < MAXP; ++i) { s[i][0] = rl(smv.x, i); s[i][1] = rl(smv.y, i); s[i][2] = rl(smv.z, i); }
    if (tid < 2 * MAXP) okf[tid] = 0;
    unsigned long long cand = 0ull;                              // the wave's best point, its bucket, its rub
    int cand_j = wave;
    unsigned c2 = 0u;
    bool cached = false;
    __syncthreads();
#ifdef PCP_FPS_STAMPS
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, sT = 0, sS = 0, sP = 0, sB1 = 0, sK = 0, sB2 = 0, nR = 0, nPass = 0, nAct = 0, nHit = 0, nIdle = 0;
    const int kd0 = kd;
#endif
    for (int round = 0;; ++round) {
        const int par = round & 1;
        const unsigned long long t_now = TIMED ? wall_clock64() : 0ull;      // read here, used by the picker: off its critical path
        STAMP(t0);
        // every wave keeps its own copy of the ring current: its LDS accesses stay in program order, so its flushes read the
        // entries back without a barrier (the round's picks are numbers kd-n .. kd-1)
        if (DEFER && lane < n) ring[(kd - n + lane) & (RING - 1)] = smv;
        // ---- bucket tests by the owning lanes: which samples could change the bucket, and do they reach its arg-max row
        const float4 ps = bxyz[wave * 64 + lane];
        const float cmax = __uint_as_float((unsigned)(key >> 32));
        unsigned affb = 0u;
        bool hit = false;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            if (i < n) {                                         // wave-uniform
                const float qx = __builtin_amdgcn_fmed3f(s[i][0], blo[0], bhi[0]);   // clamp(sample, box)
                const float qy = __builtin_amdgcn_fmed3f(s[i][1], blo[1], bhi[1]);
                const float qz = __builtin_amdgcn_fmed3f(s[i][2], blo[2], bhi[2]);
                const bool a = dist2<MODE>(s[i][0], s[i][1], s[i][2], qx, qy, qz) < cmax;
                affb |= a ? (1u << i) : 0u;
                hit = hit || (a && dist2<MODE>(s[i][0], s[i][1], s[i][2], ps.x, ps.y, ps.z) < cmax);
            }
        }
        bool flush = affb != 0u;
        if (DEFER) {
            const unsigned cnt = pend >> 30;
            // the oldest pending pick must leave before the ring wraps over it (K > 1,024 only)
            const bool aged = cnt != 0u && (((unsigned)kd - (pend & 1023u)) & 1023u) > (unsigned)(RING - 16);
            flush = (affb != 0u && (hit || cnt + (unsigned)__popc(affb) > 3u)) || aged;
            if (!flush && affb != 0u) {                          // append the affecting samples' pick numbers, in order
                unsigned c = cnt, w = pend & 0x3fffffffu;
#pragma unroll
                for (int i = 0; i < MAXP; ++i)
                    if ((affb >> i) & 1u) { w |= ((unsigned)(kd - n + i) & 1023u) << (10u * c); c += 1u; }
                pend = w | (c << 30);
            }
        }
        const unsigned long long amask = __ballot(flush);
        const unsigned long long hmask = __ballot(flush && hit);
        // the wave's partial changes only if a bucket that is reduced this round could be its top or its runner-up
        const bool rebuild = !cached || __ballot(hit && (unsigned)(key >> 32) >= c2) != 0ull;
        if (rebuild) { cand = 0ull; cand_j = wave; c2 = 0u; }
        auto idle_top2 = [&]() {                                 // over the buckets that keep their key this round
            const unsigned long long kb = hit ? 0ull : key;
            const unsigned h = (unsigned)(kb >> 32), lo = (unsigned)kb;
            unsigned a1 = h, a2 = hit ? 0u : sec;
            wave_top2_u32(a1, a2);
            unsigned long long hm = __ballot(h == a1);
            int src;
            unsigned l;
            if (__popcll(hm) == 1) {
                src = __ffsll((long long)hm) - 1;
                l = (unsigned)__builtin_amdgcn_readlane((int)lo, src);
            } else {
                l = wave_umax_u32(h == a1 ? lo : 0u);
                hm = __ballot(h == a1 && lo == l);
                src = __ffsll((long long)hm) - 1;
            }
            cand = ((unsigned long long)a1 << 32) | l;
            cand_j = src * NW + wave;
            c2 = a2;
        };
        auto merge = [&](unsigned long long wk, unsigned ws, int j) {     // all wave-uniform
            if (wk > cand) { c2 = max(c2, max((unsigned)(cand >> 32), ws)); cand = wk; cand_j = j; }
            else c2 = max(c2, (unsigned)(wk >> 32));
        };
        auto run_pass = [&](auto GC, auto FC, const int (&ol)[4], bool first) {
            constexpr int G = decltype(GC)::value;
            constexpr bool FULL = decltype(FC)::value;           // every bucket of the pass has all its rows
            int r0[G];
            float4 pt[G];
            float mm[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {                        // BH > 1: BH consecutive slots are the parts of one bucket
                r0[u] = (ol[u] * NW + wave) * BS + (u % BH) * 64;
                if (FULL) {                                      // scalar base + lane: no per-lane address math
                    pt[u] = (S + r0[u])[lane];
                    mm[u] = (D + r0[u])[lane];
                } else {
                    const int r = min(r0[u] + lane, M - 1);
                    pt[u] = S[r];
                    mm[u] = D[r];
                }
            }
            // lane 4u+i fetches pending sample i of slot u's bucket (slots past the count read some valid entry, unused)
            unsigned pw[G];
            float4 pm = make_float4(0.f, 0.f, 0.f, 0.f);
            if (DEFER) {
#pragma unroll
                for (int u = 0; u < G; ++u) pw[u] = (unsigned)__builtin_amdgcn_readlane((int)pend, ol[u]);
                unsigned w = pw[0];
#pragma unroll
                for (int u = 1; u < G; ++u) w = (lane >> 2) == u ? pw[u] : w;
                pm = ring[(w >> (10u * (unsigned)(lane & 3))) & 1023u];
            }
            if (first && rebuild) {                              // VALU work under the memory latency
                idle_top2();
#ifdef PCP_FPS_STAMPS
                nIdle += 1;
#endif
            }
#pragma unroll
            for (int u = 0; u < G; ++u)                          // keep hipcc from sinking a load into the update branch
                asm volatile("" : "+v"(mm[u]), "+v"(pt[u].w));
            unsigned long long bk[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int r = r0[u] + lane;
                const unsigned ab = (unsigned)__builtin_amdgcn_readlane((int)affb, ol[u]);   // the samples this bucket needs
                float m = mm[u];
                if (DEFER) {
                    const unsigned np = pw[u] >> 30;             // wave-uniform, 0..3
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (i < (int)np)
                            m = fminf(m, dist2<MODE>(rl(pm.x, 4 * u + i), rl(pm.y, 4 * u + i), rl(pm.z, 4 * u + i),
                                                     pt[u].x, pt[u].y, pt[u].z));
                }
#pragma unroll
                for (int i = 0; i < MAXP; ++i) {
                    if ((ab >> i) & 1u)                          // wave-uniform; fminf keeps m when d is NaN, like `d < m`
                        m = fminf(m, dist2<MODE>(s[i][0], s[i][1], s[i][2], pt[u].x, pt[u].y, pt[u].z));
                }
                const bool live = FULL || r < M;
                if (live && m < mm[u]) D[r] = m;
                bk[u] = live ? make_key(m, (unsigned)__float_as_int(pt[u].w)) : 0ull;
            }
#pragma unroll
            for (int q = 0; q < G; q += BH) {
                if (q > 0 && ol[q] == ol[0]) continue;           // a left-over slot repeats slot 0
                if (!((hmask >> ol[q]) & 1ull)) continue;        // the arg-max row kept its value: so did the key
#ifdef PCP_FPS_STAMPS
                nHit += 1;
#endif
                unsigned long long pk[BH], wk[BH];
                unsigned zr[BH], ws[BH];
                int src[BH];
#pragma unroll
                for (int h = 0; h < BH; ++h) { pk[h] = bk[q + h]; zr[h] = 0u; }
                wave_argmax2_key_x<BH>(pk, zr, wk, ws, src);
                const int j = ol[q] * NW + wave;
                int w = 0;                                       // scalar: the part that holds the bucket's top
                unsigned long long wb = wk[0];
                unsigned wsb = ws[0];
#pragma unroll
                for (int h = 1; h < BH; ++h) {
                    if (wk[h] > wb) { wsb = max(wsb, max((unsigned)(wb >> 32), ws[h])); wb = wk[h]; w = h; }
                    else wsb = max(wsb, (unsigned)(wk[h] >> 32));
                }
#pragma unroll
                for (int h = 0; h < BH; ++h)
                    if (w == h && lane == src[h]) bxyz[PCP_BX(j, NW, 64)] = pt[q + h];
                if (lane == ol[q]) { key = wb; sec = wsb; }
                if (rebuild) merge(wb, wsb, j);
            }
        };
        bool first = true;
        STAMP(t1);
        __builtin_amdgcn_s_setprio(3);
        {
            unsigned long long m = amask;
#ifdef PCP_FPS_STAMPS
            nAct += __popcll(m);
#endif
            while (m) {                                          // wave-uniform
#ifdef PCP_FPS_STAMPS
                nPass += 1;
#endif
                int ol[4], nb = 0;
#pragma unroll
                for (int u = 0; u < 4; u += BH) {                // 4 / BH buckets per pass, BH slots each
                    if (m) { ol[u] = __ffsll((long long)m) - 1; m &= m - 1; nb = u + BH; }
                    else ol[u] = ol[0];
#pragma unroll
                    for (int h = 1; h < BH; ++h) ol[u + h] = ol[u];
                }
                using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                using I4 = std::integral_constant<int, 4>;
                // only the cloud's last bucket can be short of rows; it belongs to lane (NB-1)/NW
                const bool full = (M % BS == 0) || (((NB - 1) % NW) != wave) || !((amask >> (((NB - 1) / NW) & 63)) & 1ull);
                if (full) {
                    if constexpr (BH == 1) {
                        if (nb == 1) run_pass(I1{}, std::true_type{}, ol, first);
                        else if (nb == 2) run_pass(I2{}, std::true_type{}, ol, first);
                        else run_pass(I4{}, std::true_type{}, ol, first);
                    } else if constexpr (BH == 2) {
                        if (nb == 2) run_pass(I2{}, std::true_type{}, ol, first);
                        else run_pass(I4{}, std::true_type{}, ol, first);
                    } else {
                        run_pass(I4{}, std::true_type{}, ol, first);
                    }
                } else {
                    run_pass(I4{}, std::false_type{}, ol, first);
                }
                first = false;
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (DEFER && flush) pend = 0u;                           // streamed: nothing pending any more
        if (first && rebuild) idle_top2();                       // nothing to stream, but no partial yet
        cached = true;
        STAMP(t2);
        // only this wave ever writes bxyz[cand_j], so the copy cannot race with another wave's update
        if (rebuild && lane == 0) {
            const float4 bx = bxyz[PCP_BX(cand_j, NW, 64)];
            pkey[wave] = cand;
            pxyz[wave] = make_float4(bx.x, bx.y, bx.z, __uint_as_float(c2));
        }
        STAMP(t3);
        // workgroup barrier WITHOUT draining global stores (see the one-pick loop)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        STAMP(t4);
        // ---- the picks: every wave ranks its own top among the NW partials (lane w = wave w)
        const unsigned long long kk = pkey[lane & (NW - 1)];
        const float4 cc = pxyz[lane & (NW - 1)];
        const unsigned long long gt = __ballot(kk > cand) & LOW;                 // the waves whose top beats this one's
        const int rank = __popcll(gt);
        if (wave == 0 && lane < MAXP) okf[(par ^ 1) * MAXP + lane] = 0;          // next round's verdicts start at "no"
        if (cand != 0ull && rank < MAXP) {                                       // wave-uniform
            const unsigned hi_own = (unsigned)(cand >> 32);
            const float ox = rl(cc.x, wave), oy = rl(cc.y, wave), oz = rl(cc.z, wave);
            bool ok = true;
            if (rank > 0) {
                const unsigned long long tie = __ballot((unsigned)(kk >> 32) == hi_own) & LOW & ~(1ull << wave);
                const float d = dist2<MODE>(cc.x, cc.y, cc.z, ox, oy, oz);       // lane j: the better pick a_j against this one
                const bool fin = dist2<MODE>(cc.x, cc.y, cc.z, cc.x, cc.y, cc.z) == 0.f;
                const unsigned long long bad =
                    __ballot(d < __uint_as_float(hi_own) || !(__float_as_uint(cc.w) < hi_own) || !fin) & gt;
                ok = tie == 0ull && bad == 0ull && hi_own != 0u;
            }
            if (lane == 0) {
                sel[par * MAXP + rank] = make_float4(ox, oy, oz, __int_as_float((int)(~(unsigned)cand)));
                int v = ok ? 1 : 0;
                // a timed quantum (sliced launches): the wave that writes the first verdict also says whether time is up,
                // so that every wave leaves the rounds at the same round
                if (TIMED && rank == 0 && t_now >= t_stop) v |= 2;
                okf[par * MAXP + rank] = v;
            }
        }
        STAMP(t5);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const float4 sm = sel[par * MAXP + (lane & (MAXP - 1))];
        smv = sm;
        const int okv = okf[par * MAXP + (lane & (MAXP - 1))];
        const unsigned okm = (unsigned)(__ballot((okv & 1) != 0) & ((1ull << MAXP) - 1ull));   // bit 0: the arg-max itself, always certain
        const bool time_up = TIMED && (__builtin_amdgcn_readlane(okv, 0) & 2) != 0;
        n = min(__ffs((int)~okm) - 1, min(MAXP, kn - kd));                       // leading certified ranks
        if (cand != 0ull && rank < n && lane == 0) {                             // the picked waves write their pick
            irow[kd + rank] = (long long)(int)(~(unsigned)cand);
            if (orow) {
                orow[3 * (kd + rank)] = rl(cc.x, wave); orow[3 * (kd + rank) + 1] = rl(cc.y, wave); orow[3 * (kd + rank) + 2] = rl(cc.z, wave);
            }
        }
        kd += n;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) { s[i][0] = rl(sm.x, i); s[i][1] = rl(sm.y, i); s[i][2] = rl(sm.z, i); }
#ifdef PCP_FPS_STAMPS
        { unsigned long long t6; STAMP(t6); sT += t1 - t0; sS += t2 - t1; sP += t3 - t2; sB1 += t4 - t3; sK += t5 - t4; sB2 += t6 - t5; nR += 1; }
#endif
        if (kd >= k_stop || time_up) break;
    }
    cy.pend = pend; cy.sec = sec; cy.n = n; cy.smv = smv;
#ifdef PCP_FPS_STAMPS
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 1 || wave == NW - 1))
        printf("[stamps] wave %d M=%d NB=%d picks=%d rounds=%llu | cycles per round: test=%llu stream=%llu publish=%llu barrier1=%llu picker=%llu barrier2+read=%llu | per round: passes=%.2f buckets=%.2f reduced=%.2f rebuilds=%.2f\n",
               wave, M, NB, kn - kd0, nR, sT / nR, sS / nR, sP / nR, sB1 / nR, sK / nR, sB2 / nR, (double)nPass / nR, (double)nAct / nR, (double)nHit / nR, (double)nIdle / nR);
#endif
}

template <int T, int CPL>
constexpr size_t fps_bucket_lds_bytes()
{
    // cell histogram | red | wave partial keys [2][NW] | their xyz [2][NW] | bucket arg-max xyz [T*CPL] | round picks [2][8] + verdicts [2][8]
    return PCP_FPS_NCELL * 4 + PCP_FPS_RED * 4 + 2 * (T / 64) * 8 + 2 * (T / 64) * 16 + (size_t)T * CPL * 16 + 16 * 16 + 16 * 4 + 64;
}

template <int T, int CPL, bool MULTI, int MODE, int BH = 1, bool MP = false, bool DEFER = false, int MAXP = 4>
__device__ __forceinline__ void
fps_bucket_body(const float* __restrict__ xyz, const long long* __restrict__ offsets, int K,
                long long* __restrict__ idx, float* __restrict__ out_xyz, float4* __restrict__ pts4,
                float* __restrict__ md, int m_lo, int m_hi, int lazy_k, long long total_pts, int m_cap,
                int* __restrict__ overflow)
{
    // this launch serves the clouds with m_lo <= M <= m_hi; the others belong to a sibling launch
    constexpr int NW = T / 64, NBMAX = T * CPL, NCELL = PCP_FPS_NCELL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* cellcnt = (int*)smem;                                           // [NCELL] histogram / cursors
    float* red = (float*)(cellcnt + NCELL);                              // [PCP_FPS_RED] bbox partials (6 per wave) / scan scratch (17)
    static_assert(6 * NW <= PCP_FPS_RED && 17 <= PCP_FPS_RED, "red[] holds 6 floats per wave and the block scan's 17 ints");
    unsigned long long* pkey = (unsigned long long*)(red + PCP_FPS_RED); // [2][NW] wave partial keys
    float4* pxyz = (float4*)(pkey + 2 * NW);                             // [2][NW] their xyz
    float4* bxyz = pxyz + 2 * NW;                                        // [NBMAX] xyz of every bucket's arg-max (written by its wave only)

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long base = offsets[b];
    const long long Mll = offsets[b + 1] - base;
    // m_cap = the largest cloud any launch of this call takes.  A cloud beyond it, or one whose rows lie outside
    // [0, total_pts), is nobody's: the launch that owns the top size range answers for it with an all -1 row and
    // raises the caller's flag, so no row of idx / out_xyz is ever left unwritten.
    const bool bad = fps_cloud_out_of_bounds(base, Mll, total_pts, m_cap);
    if (bad) {
        if (m_hi != m_cap) return;
        if (tid == 0 && overflow) *overflow = 1;
    }
    const int M = bad ? 0 : (int)Mll;
    if (!bad && (M < m_lo || M > m_hi)) return;
    const int kn = K < M ? K : (M > 0 ? M : 0);
    long long* irow = idx + (size_t)b * K;
    float* orow = out_xyz ? out_xyz + (size_t)b * K * 3 : nullptr;
    for (int k = kn + tid; k < K; k += T) {                              // pytorch3d padding
        irow[k] = -1;
        if (orow) { orow[3 * k] = 0.f; orow[3 * k + 1] = 0.f; orow[3 * k + 2] = 0.f; }
    }
    if (kn == 0) return;
    const float* P = xyz + 3 * base;
    float4* S = pts4 + base;
    float* D = md + base;

    fps_sort_points<T>(P, M, S, D, cellcnt, red, tid, lane, wave);
#if defined(PCP_FPS_STOP_AFTER) && PCP_FPS_STOP_AFTER <= 3
    return;                                                      // tuning builds: time the prologue's phases
#endif

    // ---- prologue 3: buckets and their owners ---------------------------------------------------------
    // 64-row chunks per bucket; the single-chunk shape is only launched for clouds of <= 64·NBMAX points
    // (BH = 2 or 4, single-chunk shape only: buckets of BH 64-row parts that are streamed side by side, for clouds
    //  of up to 64·BH·NBMAX points)
    static_assert(BH == 1 || ((BH == 2 || BH == 4) && !MULTI && CPL == 1), "split buckets are a variant of the wide shape");
    const int chunks = MULTI ? (M + 64 * NBMAX - 1) / (64 * NBMAX) : 1;
    const int BS = 64 * chunks * BH;
    const int NB = (M + BS - 1) / BS;
    unsigned long long key[CPL];
    float blo[CPL][3], bhi[CPL][3];
#pragma unroll
    for (int s = 0; s < CPL; ++s) {
        const int j = (s * 64 + lane) * NW + wave;
        const int r0 = min(j, NB - 1) * BS;                     // clamped: unowned slots scan a real bucket, then drop it
        const int rn = min(BS, M - r0);
#pragma unroll
        for (int c = 0; c < 3; ++c) { blo[s][c] = INFINITY; bhi[s][c] = -INFINITY; }
        bool nan_pt = false;
        for (int i = 0; i < rn; i += 8) {
            float4 pt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pt[u] = S[r0 + min(i + u, rn - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                blo[s][0] = fminf(blo[s][0], pt[u].x); bhi[s][0] = fmaxf(bhi[s][0], pt[u].x);
                blo[s][1] = fminf(blo[s][1], pt[u].y); bhi[s][1] = fmaxf(bhi[s][1], pt[u].y);
                blo[s][2] = fminf(blo[s][2], pt[u].z); bhi[s][2] = fmaxf(bhi[s][2], pt[u].z);
                nan_pt = nan_pt || !(pt[u].x == pt[u].x && pt[u].y == pt[u].y && pt[u].z == pt[u].z);
            }
        }
        // fminf / fmaxf skip NaN coordinates, so the box would not cover such a point; its distances are NaN and its
        // min-distance never drops — give the bucket an infinite box so that no bound derived from the box ever does
        if (nan_pt) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { blo[s][c] = -INFINITY; bhi[s][c] = INFINITY; }
        }
        key[s] = j < NB ? make_key(FLT_MAX, 0u) : 0ull;         // FLT_MAX: active in round 1; 0: never
    }
    float sx = P[0], sy = P[1], sz = P[2];
    int last = 0;
    // Wave-uniform arg-max over ALL buckets the wave owns (key, bucket), as the deferred rounds track it.
    unsigned long long wmax = __builtin_amdgcn_readfirstlane((int)(key[0] >> 32)) ? make_key(FLT_MAX, 0u) : 0ull;
    int wmax_j = wave;

    // every owned slot's arg-max point starts as the first sample: round 0 then "hits" every bucket (distance 0)
#pragma unroll
    for (int s = 0; s < CPL; ++s) bxyz[PCP_BX((s * 64 + lane) * NW + wave, NW, (NBMAX / NW))] = make_float4(sx, sy, sz, 0.f);

    // ---- rounds -------------------------------------------------------------------------------------
    int k0 = 0;
    if constexpr (!MULTI && CPL == 1) {
        static_assert(PCP_FPS_NCELL * 4 >= PCP_FPS_LAZY_MAX_ROUNDS * 16, "the sample table reuses the cell histogram");
        k0 = min(min(lazy_k, PCP_FPS_LAZY_MAX_ROUNDS), kn - 1);
#ifndef PCP_NO_LAZY
        if (k0 > 0)                                             // workgroup-uniform
            fps_lazy_rounds<T, MODE, BH>(k0, M, NB, S, D, (float4*)cellcnt, pkey, pxyz, bxyz, irow, orow, tid, lane, wave,
                                     key[0], blo[0], bhi[0], sx, sy, sz, last, wmax, wmax_j);
#endif
    }
    if constexpr (!MP) {
        // ONE PICK PER ROUND: the shape for batches that put more than one workgroup on a CU, where the rounds are bound by
        // instruction issue and the leanest round wins (the multi-pick rounds below spend more instructions per pick).
        for (int k = k0;; ++k) {
            if (tid == 0) {
                irow[k] = last;
                if (orow) { orow[3 * k] = sx; orow[3 * k + 1] = sy; orow[3 * k + 2] = sz; }
            }
            if (k + 1 >= kn) break;
            const int par = k & 1;
            // bucket tests by the owning lanes
            unsigned long long amask[CPL];
    #pragma unroll
            for (int s = 0; s < CPL; ++s) {
                const float cmax = __uint_as_float((unsigned)(key[s] >> 32));
                const float qx = __builtin_amdgcn_fmed3f(sx, blo[s][0], bhi[s][0]);   // clamp(s, box)
                const float qy = __builtin_amdgcn_fmed3f(sy, blo[s][1], bhi[s][1]);
                const float qz = __builtin_amdgcn_fmed3f(sz, blo[s][2], bhi[s][2]);
                amask[s] = __ballot(dist2<MODE>(sx, sy, sz, qx, qy, qz) < cmax);
            }
            // Running wave candidate (all wave-uniform): arg-max over the buckets this round leaves
            // untouched — reduced while the first loads are in flight — then raised by every bucket the
            // wave updates.  cand_j is the candidate's bucket: its xyz sits in bxyz[cand_j].
            unsigned long long cand = wmax;
            int cand_j = wmax_j;
            bool need_idle = false;
    #pragma unroll
            for (int s = 0; s < CPL; ++s)                           // is the wave's top bucket about to change?
                if (((wmax_j / NW) >> 6) == s && ((amask[s] >> ((wmax_j / NW) & 63)) & 1ull)) need_idle = true;
            auto idle_argmax = [&]() {
                if (!need_idle) return;                             // wave-uniform
                unsigned long long kb = 0ull;
                int sb = 0;
    #pragma unroll
                for (int s = 0; s < CPL; ++s) {
                    const bool act = (amask[s] >> lane) & 1ull;
                    if (!act && key[s] > kb) { kb = key[s]; sb = s; }
                }
                int src;
                cand = wave_argmax_key(kb, &src);
                cand_j = ((__builtin_amdgcn_readlane(sb, src) * 64 + src) * NW + wave);
            };
            // The wave streams its own active buckets, up to four in flight; the pass body is
            // instantiated for 1, 2 and 4 buckets so a short pass does not pay for dead slots.
            auto run_pass = [&](auto GC, auto FC, const int (&ol)[4], unsigned long long& key_s, int s_idx, bool first) {
                constexpr int G = decltype(GC)::value;
                constexpr bool FULL = decltype(FC)::value;          // every bucket of the pass has all its rows
                int r0[G];
    #pragma unroll
                for (int u = 0; u < G; ++u)                         // BH > 1: BH consecutive slots are the parts of one bucket
                    r0[u] = ((s_idx * 64 + ol[u]) * NW + wave) * BS + (u % BH) * 64;
                unsigned long long bk[G];
                float4 bp[G];
                const int nchunks = MULTI ? chunks : 1;
                for (int c = 0; c < nchunks; ++c) {
                    float4 pt[G];
                    float mm[G];
    #pragma unroll
                    for (int u = 0; u < G; ++u) {                   // unconditional loads: all in flight together
                        if (FULL) {                                 // scalar base + lane: no per-lane address math
                            pt[u] = (S + (r0[u] + c * 64))[lane];
                            mm[u] = (D + (r0[u] + c * 64))[lane];
                        } else {
                            const int r = min(r0[u] + c * 64 + lane, M - 1);
                            pt[u] = S[r];
                            mm[u] = D[r];
                        }
                    }
                    if (first && c == 0) idle_argmax();             // VALU work under the memory latency
    #pragma unroll
                    for (int u = 0; u < G; ++u)                     // keep hipcc from sinking a load into the update branch
                        asm volatile("" : "+v"(mm[u]), "+v"(pt[u].w));
    #pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const int r = r0[u] + c * 64 + lane;
                        const float d = dist2<MODE>(sx, sy, sz, pt[u].x, pt[u].y, pt[u].z);
                        const bool live = FULL || r < M;
                        if (live && d < mm[u]) { mm[u] = d; D[r] = d; }
                        const unsigned long long kk = live ? make_key(mm[u], (unsigned)__float_as_int(pt[u].w)) : 0ull;
                        if (!MULTI || c == 0 || kk > bk[u]) { bk[u] = kk; bp[u] = pt[u]; }
                    }
                }
                unsigned long long wk[G];
                int src[G];
                wave_argmax_key_x<G>(bk, wk, src);
                if constexpr (BH > 1 && G < BH) {
                    // never called: a split bucket occupies BH slots
                } else if constexpr (BH > 1) {
    #pragma unroll
                    for (int q = 0; q < G; q += BH) {               // the bucket's key is the largest of its parts' keys
                        const int j = (s_idx * 64 + ol[q]) * NW + wave;
                        int w = q;                                  // scalar
                        unsigned long long wb = wk[q];
    #pragma unroll
                        for (int h = 1; h < BH; ++h)
                            if (wk[q + h] > wb) { wb = wk[q + h]; w = q + h; }
    #pragma unroll
                        for (int h = 0; h < BH; ++h)
                            if (w == q + h && lane == src[q + h]) bxyz[PCP_BX(j, NW, (NBMAX / NW))] = bp[q + h];
                        if (lane == ol[q]) key_s = wb;
                        if (wb > cand) { cand = wb; cand_j = j; }
                    }
                } else {
    #pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const int j = (s_idx * 64 + ol[u]) * NW + wave;
                        if (lane == src[u]) bxyz[PCP_BX(j, NW, (NBMAX / NW))] = bp[u];
                        if (lane == ol[u]) key_s = wk[u];
                        if (wk[u] > cand) { cand = wk[u]; cand_j = j; }      // scalar
                    }
                }
            };
            bool first = true;
            __builtin_amdgcn_s_setprio(3);                          // the streaming phase is the round's critical path
    #pragma unroll
            for (int s = 0; s < CPL; ++s) {
                unsigned long long m = amask[s];
                while (m) {                                         // wave-uniform
                    int ol[4], n = 0;
                    if constexpr (BH > 1) {                         // 4 / BH buckets per pass, BH slots each
    #pragma unroll
                        for (int u = 0; u < 4; u += BH) {
                            if (m) { ol[u] = __ffsll((long long)m) - 1; m &= m - 1; n = u + BH; }
                            else ol[u] = ol[0];
    #pragma unroll
                            for (int h = 1; h < BH; ++h) ol[u + h] = ol[u];
                        }
                    } else {
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {               // a fourth slot left over repeats slot 0: harmless
                            if (m) { ol[u] = __ffsll((long long)m) - 1; m &= m - 1; n = u + 1; }
                            else ol[u] = ol[0];
                        }
                    }
                    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                    using I4 = std::integral_constant<int, 4>;
                    // only the cloud's last bucket can be short of rows; it belongs to lane (NB-1)/NW % 64
                    const bool full = (M % BS == 0) || (((NB - 1) % NW) != wave) || (((NB - 1) / NW) >> 6) != s ||
                                      !((amask[s] >> (((NB - 1) / NW) & 63)) & 1ull);
                    if (full) {
                        if (n == 1) run_pass(I1{}, std::true_type{}, ol, key[s], s, first);
                        else if (n == 2) run_pass(I2{}, std::true_type{}, ol, key[s], s, first);
                        else run_pass(I4{}, std::true_type{}, ol, key[s], s, first);
                    } else {
                        run_pass(I4{}, std::false_type{}, ol, key[s], s, first);
                    }
                    first = false;
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (first) idle_argmax();                               // nothing to stream this round
            wmax = cand;                                            // = arg-max over all the wave's buckets again
            wmax_j = cand_j;
            // only this wave ever writes bxyz[cand_j], so the copy cannot race with another wave's update
            if (lane == 0) { pkey[par * NW + wave] = cand; pxyz[par * NW + wave] = bxyz[PCP_BX(cand_j, NW, (NBMAX / NW))]; }
            // Workgroup barrier WITHOUT draining global stores: only LDS traffic (the partials) has to be
            // visible to the other waves.  The min_d stores of this round are re-read by this wave alone
            // (a bucket is streamed by its owning wave only), and same-wave accesses stay ordered, so
            // waiting for their acknowledgements — which __syncthreads() does — is wasted serial time.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            {
                // every low lane fetches its wave's key AND xyz in one LDS round trip; the winner's xyz is then a
                // readlane away instead of a second, dependent LDS read
                const unsigned long long kk = pkey[par * NW + (lane & (NW - 1))];
                const float4 cc = pxyz[par * NW + (lane & (NW - 1))];
                int src;
                const unsigned long long wk = lowlanes_argmax_key<NW>(kk, &src);
                last = wk ? (int)(~(unsigned)wk) : 0;
                sx = rl(cc.x, src); sy = rl(cc.y, src); sz = rl(cc.z, src);
            }
        }
        return;
    }
    if (tid == 0) {
        irow[k0] = last;
        if (orow) { orow[3 * k0] = sx; orow[3 * k0 + 1] = sy; orow[3 * k0 + 2] = sz; }
    }
    if (k0 + 1 >= kn) return;
    if constexpr (MP) {
        static_assert(!MULTI && CPL == 1, "multi-pick rounds are a variant of the wide shape");
        float4* sel = bxyz + NBMAX;                                      // [2][8] a round's picks {x,y,z,bits(idx)}
        int* okf = (int*)(sel + 16);                                     // [2][8] their verdicts
        int kd = k0 + 1;
        FpsMpCarry cy = {0u, (unsigned)(key[0] >> 32), 1, make_float4(sx, sy, sz, 0.f)};
        fps_multipick_rounds<T, MODE, BH, DEFER, false, MAXP>(kd, kn, 0ull, kn, M, NB, S, D, pkey, pxyz, bxyz, sel, okf, (float4*)cellcnt, irow, orow, tid, lane, wave,
                                                 key[0], blo[0], bhi[0], cy);
    }
}

// "wide" shape (single-chunk buckets): two workgroups of 1024 (four of 512) per CU, so every frame of a
// 300-frame batch is resident at once.  8 waves/SIMD need <= 64 VGPRs and <= 80 SGPRs per wave (800 per
// SIMD, allocated 16 + 16k); the caps are explicit because __launch_bounds__ alone does not enforce them.
template <int T, int CPL, int MODE, int BH = 1>
__global__ void __launch_bounds__(T, 2048 / T) __attribute__((amdgpu_num_sgpr(80), amdgpu_num_vgpr(64)))
fps_bucket_wide(const float* __restrict__ xyz, const long long* __restrict__ offsets, int K,
                long long* __restrict__ idx, float* __restrict__ out_xyz, float4* __restrict__ pts4,
                float* __restrict__ md, int m_lo, int m_hi, int lazy_k, long long total_pts, int m_cap,
                int* __restrict__ overflow, const unsigned* __restrict__ mixed_ctl)
{
    fps_bucket_body<T, CPL, false, MODE, BH>(xyz, offsets, K, idx, out_xyz, pts4, md, m_lo, m_hi, lazy_k, total_pts,
                                             m_cap, overflow);
}

// The wide shape with MULTI-PICK rounds, for batches of at most one workgroup per CU (<= 256 clouds): there a round is
// bound by its dependent chain (test -> loads -> arg-max -> barrier -> pick), not by instruction issue, and certifying
// up to four picks per round cuts the number of chains to a third.  One workgroup per CU = four waves per SIMD, so the
// kernel may use 128 VGPRs.
template <int T, int CPL, int MODE, int BH = 1, bool DEFER = false>
__global__ void __launch_bounds__(T)
fps_bucket_wide_mp(const float* __restrict__ xyz, const long long* __restrict__ offsets, int K,
                   long long* __restrict__ idx, float* __restrict__ out_xyz, float4* __restrict__ pts4,
                   float* __restrict__ md, int m_lo, int m_hi, int lazy_k, long long total_pts, int m_cap,
                   int* __restrict__ overflow, const unsigned* __restrict__ mixed_ctl)
{
    // A batch with clouds in more than one wide size range (counted by fps_slice_prep): the call's any-shape launch runs
    // it whole, see fps_sliced_body - one launch after the other would cost the sum of their times.
    if (mixed_ctl) {
        const unsigned n0 = mixed_ctl[PCP_FPS_SLICE_MAX_SLOTS], n1 = mixed_ctl[PCP_FPS_SLICE_CTL_WORDS + PCP_FPS_SLICE_MAX_SLOTS],
                       n2 = mixed_ctl[2 * PCP_FPS_SLICE_CTL_WORDS + PCP_FPS_SLICE_MAX_SLOTS];
        if (((n0 != 0u) + (n1 != 0u) + (n2 != 0u)) > 1) return;
    }
    fps_bucket_body<T, CPL, false, MODE, BH, true, DEFER, PCP_FPS_MAXP>(xyz, offsets, K, idx, out_xyz, pts4, md, m_lo, m_hi, lazy_k,
                                                                        total_pts, m_cap, overflow);
}

// The same kernel held to 64 VGPRs / 80 SGPRs (scalar registers spill to VGPR lanes) so that TWO workgroups fit a CU:
// for batches of more than one cloud per CU (the 300-frame episode).  Slightly slower per workgroup, but every cloud is
// resident at once — 2.90 instead of 3.84 ms per 300 frames, and ahead of the one-pick kernel's 3.10.
template <int T, int CPL, int MODE, int BH = 1>
__global__ void __launch_bounds__(T, 8) __attribute__((amdgpu_num_sgpr(80), amdgpu_num_vgpr(64)))
fps_bucket_wide_mp2(const float* __restrict__ xyz, const long long* __restrict__ offsets, int K,
                    long long* __restrict__ idx, float* __restrict__ out_xyz, float4* __restrict__ pts4,
                    float* __restrict__ md, int m_lo, int m_hi, int lazy_k, long long total_pts, int m_cap,
                    int* __restrict__ overflow, const unsigned* __restrict__ mixed_ctl)
{
    if (mixed_ctl) {                                             // a mixed-size batch: the call's any-shape launch runs it (see fps_bucket_wide_mp)
        const unsigned n0 = mixed_ctl[PCP_FPS_SLICE_MAX_SLOTS], n1 = mixed_ctl[PCP_FPS_SLICE_CTL_WORDS + PCP_FPS_SLICE_MAX_SLOTS],
                       n2 = mixed_ctl[2 * PCP_FPS_SLICE_CTL_WORDS + PCP_FPS_SLICE_MAX_SLOTS];
        if (((n0 != 0u) + (n1 != 0u) + (n2 != 0u)) > 1) return;
    }
    fps_bucket_body<T, CPL, false, MODE, BH, true, true>(xyz, offsets, K, idx, out_xyz, pts4, md, m_lo, m_hi, lazy_k,
                                                         total_pts, m_cap, overflow);
}

// ---- QUANTUM-SLICED launch: a little more clouds than CUs, one workgroup per CU ----------------------------------
// A batch of a little more than one cloud per CU (the 300-frame episode on 256 CUs) leaves most CUs idle for the last
// third of a resident launch: 44 CUs host two workgroups, which share the CU's instruction issue and set the pace (256
// clouds: 1.94 ms, 300 clouds: 2.82 ms).  A cloud's picks are one dependent chain, so the only way to use the idle CUs is
// to move clouds between CUs over time.  Here G = one workgroup per CU stays resident and runs clouds a QUANTUM at a
// time.  Workgroup i starts cloud i; the E = B - G extra clouds start in WAITING SLOTS, one per group of
// workgroups.  Slot j and the workgroups of its group carry the same label (j mod 8 = blockIdx mod 8): workgroups that
// differ by a multiple of 8 are observed to share an XCD, and with it the L2 that holds the group's clouds — a preference
// for speed, nothing relies on it.  The 32 workgroups of a label are dealt round-robin over the label's slots.
// A quantum is a stretch of TIME (the early picks of a cloud cost several times the late ones, and it is run time that
// has to be shared out evenly), checked once per round by the wave that writes the round's first verdict.
// At the end of a quantum a workgroup reads its group's slot (one word: the waiting cloud and the time it has run so
// far); if that cloud is behind its own it claims the slot (compare-and-swap to EMPTY), saves its own
// cloud's round state (72 KB: bucket keys, boxes, pending lists, arg-max points, the sample ring), publishes it in the
// slot, and goes on with the other.  So the g workgroups of a group take turns over g + 1 clouds and each cloud runs
// g / (g + 1) of the time.  A workgroup that has finished a cloud takes whatever waits in any slot (its own label first),
// and exits when every slot is empty.  No workgroup ever waits for another: a published cloud is picked up at somebody's
// next quantum end, at the latest by its publisher when that has finished the cloud it took instead.
// The picks are those of the multi-pick rounds, which do not depend on where a round runs: bit-identical output.
// Hand-off between workgroups (placement-independent, as inside any launch): plain stores -> every wave drains
// vmcnt -> workgroup barrier -> one lane's agent-scope release fence (L2 write-back) -> vmcnt -> agent-scope store of
// the slot word; the claimer's compare-and-swap -> agent-scope acquire (drops this CU's L1) -> vmcnt -> barrier ->
// plain loads.  The min-distance and record arrays travel the same way (they are the bulk of a cloud's state and stay
// in place).
#define PCP_FPS_WIDE_PTS (64 * 1024)                  // points a wide-shape workgroup holds in whole (64-row) buckets
#define PCP_FPS_SLICE_STATE_BYTES (73728 + 256)      // per cloud: arg-max xyz 16K | ring 16K | keys 8K | sec 4K | pend 4K | boxes 24K | header
#define PCP_FPS_SLICE_MAX_B 1024                     // slot-word packing: 10 bits of cloud number beside 21 bits of run time
typedef __attribute__((address_space(1))) unsigned int fps_gu32;
// slot word j: 0 = extra cloud G + j, not started yet; 1 = empty; else 1<<31 | microseconds the cloud has run << 10 | cloud
#define FPS_SLOT_EMPTY 1u

__device__ __forceinline__ unsigned wave_umin_u32(unsigned v) { return ~wave_umax_u32(~v); }

// Does this launch (clouds of m_lo..m_hi points) have anything to do for cloud c?  The size rule of fps_bucket_body: a
// cloud out of bounds or beyond every shape belongs to the launch of the top size range, which answers it with a -1 row.
__device__ __forceinline__ bool fps_cloud_is_mine(const long long* __restrict__ offsets, const int c, const long long total_pts,
                                                  const int m_cap, const int m_lo, const int m_hi)
{
    const long long base = offsets[c], Mll = offsets[c + 1] - base;
    if (fps_cloud_out_of_bounds(base, Mll, total_pts, m_cap)) return m_hi == m_cap;
    return Mll >= m_lo && Mll <= m_hi;
}

// Wave 0, all lanes: the cloud this workgroup runs next.  Returns `cur` when it keeps its cloud (-1: exit), else a cloud
// whose slot it has CLAIMED (*got = the slot word: 0 = a fresh cloud).
__device__ __forceinline__ int fps_slice_choose(fps_gu32* slot, const int G, const int E, const int grp,
                                                const int cur, const int cur_run, const int lead, const int lane,
                                                unsigned* got, const long long* __restrict__ offsets,
                                                const long long total_pts, const int m_cap, const int m_lo, const int m_hi)
{
    if (E == 0) return cur;
    if (cur >= 0) {
        if (grp < 0) return cur;                                 // a label without extra clouds
        // a busy workgroup: its own group's slot, one look, one try
        const unsigned w = __hip_atomic_load(slot + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int c = -1, run_c = 0;
        if (w == 0u) { if (fps_cloud_is_mine(offsets, G + grp, total_pts, m_cap, m_lo, m_hi)) c = G + grp; }   // (a sibling launch's: never started here)
        else if (w >> 31) { c = (int)(w & 1023u); run_c = (int)((w >> 10) & 0x1fffffu); }
        if (c < 0 || run_c + lead > cur_run) return cur;         // nothing waits, or not far enough behind to be worth a swap
        int ok = 0;
        if (lane == 0) {
            unsigned e = w;
            ok = __hip_atomic_compare_exchange_strong(slot + grp, &e, FPS_SLOT_EMPTY, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
        }
        if (!__builtin_amdgcn_readfirstlane(ok)) return cur;     // somebody else of the group took it
        *got = w;
        return c;
    }
    // an idle workgroup: whatever waits anywhere, its own label's slots first, the cloud furthest behind first
    for (int attempt = 0; attempt < 64; ++attempt) {
        unsigned best = 0xffffffffu;
        unsigned w[PCP_FPS_SLICE_MAX_SLOTS / 64];
#pragma unroll
        for (int u = 0; u < PCP_FPS_SLICE_MAX_SLOTS / 64; ++u) {
            const int j = u * 64 + lane;
            w[u] = j < E ? __hip_atomic_load(slot + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : FPS_SLOT_EMPTY;
        }
#pragma unroll
        for (int u = 0; u < PCP_FPS_SLICE_MAX_SLOTS / 64; ++u) {
            const int j = u * 64 + lane;
            unsigned run_c = 0xffffffffu;
            if (w[u] == 0u) { if (fps_cloud_is_mine(offsets, G + j, total_pts, m_cap, m_lo, m_hi)) run_c = 0u; }
            else if (w[u] >> 31) run_c = (w[u] >> 10) & 0x1fffffu;
            if (run_c != 0xffffffffu) {
                const unsigned far = ((j ^ (int)blockIdx.x) & 7) ? 1u : 0u;
                best = min(best, (far << 30) | (run_c << 8) | (unsigned)j);
            }
        }
        best = wave_umin_u32(best);
        if (best == 0xffffffffu) return -1;
        const int j = (int)(best & 255u);
        unsigned expect = 0u;
#pragma unroll
        for (int u = 0; u < PCP_FPS_SLICE_MAX_SLOTS / 64; ++u)
            if ((j >> 6) == u) expect = (unsigned)__builtin_amdgcn_readlane((int)w[u], j & 63);
        int ok = 0;
        if (lane == 0) {
            unsigned e = expect;
            ok = __hip_atomic_compare_exchange_strong(slot + j, &e, FPS_SLOT_EMPTY, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
        }
        if (__builtin_amdgcn_readfirstlane(ok)) {
            *got = expect;
            return expect == 0u ? G + j : (int)(expect & 1023u);
        }
    }
    return -1;
}


// Ahead of a call's sliced launches: every waiting slot of the three wide shapes to "extra cloud, not started", and how
// many clouds each shape has to touch at all — a launch for a size range no cloud of the batch falls into (the host only
// knows an upper bound of the sizes, so every shape the bound allows is launched) then exits at once instead of walking
// the slots.
__global__ void __launch_bounds__(256)
fps_slice_prep(const long long* __restrict__ offsets, const int B, const long long total_pts, const int m_cap,
               unsigned* __restrict__ ctl)
{
    __shared__ int cnt[3];
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * PCP_FPS_SLICE_CTL_WORDS; i += 256) ctl[i] = 0u;
    if (tid < 3) cnt[tid] = 0;
    __syncthreads();
    constexpr int W = 64 * 1024;
    for (int c = tid; c < B; c += 256) {
#pragma unroll
        for (int sh = 0; sh < 3; ++sh)
            if (fps_cloud_is_mine(offsets, c, total_pts, m_cap, sh == 0 ? 0 : (W << (sh - 1)) + 1, W << sh)) atomicAdd(&cnt[sh], 1);
    }
    __syncthreads();
    if (tid < 3) ctl[tid * PCP_FPS_SLICE_CTL_WORDS + PCP_FPS_SLICE_MAX_SLOTS] = (unsigned)cnt[tid];
}

template <int T, int MODE, int BH>
__device__ __forceinline__ void
fps_sliced_body(const float* __restrict__ xyz, const long long* __restrict__ offsets, const int B, const int K,
                long long* __restrict__ idx, float* __restrict__ out_xyz, float4* __restrict__ pts4,
                float* __restrict__ md, const int m_lo, const int m_hi, const long long total_pts, const int m_cap,
                int* __restrict__ overflow, unsigned* __restrict__ ctl_all, const int shape, const int defer_mixed,
                char* __restrict__ state, const int quantum, const int lead)
{
    // BH = 0: ANY wide shape, chosen per cloud by its size (whole, halved or quartered buckets) - the launch for batches
    // whose clouds fall into more than one size range.  Launches on one stream run one after the other, so three launches
    // that each serve a part of such a batch cost the SUM of their times with most of the chip idle in each (300 ray-cast
    // tabletop frames of 64-83 k points, one of them below 65,536: 1.4 ms for that one cloud, then 2.9 ms for the rest).
    // fps_slice_prep counts the clouds per shape on the device; with more than one shape present the per-shape launches
    // exit at once and this one runs every cloud, otherwise this one exits and the per-shape launch runs as before.
    constexpr bool ANY = BH == 0;
    constexpr int NW = T / 64, NBMAX = T, NCELL = PCP_FPS_NCELL;
    static_assert(T == 1024, "the saved state is laid out for 1,024 threads: one bucket, one ring entry per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* cellcnt = (int*)smem;
    float* red = (float*)(cellcnt + NCELL);
    unsigned long long* pkey = (unsigned long long*)(red + PCP_FPS_RED);
    float4* pxyz = (float4*)(pkey + 2 * NW);
    float4* bxyz = pxyz + 2 * NW;
    float4* sel = bxyz + NBMAX;
    int* okf = (int*)(sel + 16);
    int* sh = okf + 16;                                          // [4] the scheduler's verdict for the workgroup
    float4* ring = (float4*)cellcnt;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned* ctl = ctl_all + (size_t)shape * PCP_FPS_SLICE_CTL_WORDS;       // this launch's waiting slots (ANY: those of shape 0)
    {
        const unsigned n0 = ctl_all[PCP_FPS_SLICE_MAX_SLOTS], n1 = ctl_all[PCP_FPS_SLICE_CTL_WORDS + PCP_FPS_SLICE_MAX_SLOTS],
                       n2 = ctl_all[2 * PCP_FPS_SLICE_CTL_WORDS + PCP_FPS_SLICE_MAX_SLOTS];
        const bool mixed = ((n0 != 0u) + (n1 != 0u) + (n2 != 0u)) > 1;
        if (ANY) { if (!mixed) return; }                         // one shape (or none): the per-shape launch runs it
        else {
            if (defer_mixed && mixed) return;                    // the ANY launch of this call runs every cloud
            if (ctl[PCP_FPS_SLICE_MAX_SLOTS] == 0u) return;      // no cloud of the batch is this shape's (fps_slice_prep)
        }
    }
    fps_gu32* slot = (fps_gu32*)ctl;
    const int G = (int)gridDim.x;                                // workgroup i starts cloud i; the host keeps G <= B
    const int E = B - G;                                         // extra clouds, E <= min(G, slots): they start in the waiting slots
    const int label = (int)blockIdx.x & 7;
    const int E_l = E > label ? (E - label + 7) >> 3 : 0;        // slots of this label: label, label + 8, ...
    const int grp = E_l > 0 ? label + 8 * (((int)blockIdx.x >> 3) % E_l) : -1;

    int cur = -1, kd = 0, kn = 0, M = 0, NB = 0;
    int bh = ANY ? 1 : BH;                                       // 64-row parts per bucket of the current cloud (wave-uniform)
    unsigned long long run_t = 0ull;                             // 10 ns ticks this workgroup's cloud has been worked on, here or elsewhere
    unsigned long long key = 0ull;
    float blo[3] = {0.f, 0.f, 0.f}, bhi[3] = {0.f, 0.f, 0.f};
    FpsMpCarry cy = {0u, 0u, 1, make_float4(0.f, 0.f, 0.f, 0.f)};
    float4* S = pts4;
    float* D = md;
    long long* irow = idx;
    float* orow = out_xyz;
    bool first = true;

    for (;;) {
        const unsigned long long t_top = wall_clock64();
        // ---- which cloud next (wave 0 decides)
        if (wave == 0) {
            int nxt = cur;
            unsigned got = 0u;
            if (first) nxt = (int)blockIdx.x;                    // nobody else ever starts cloud blockIdx.x
            else nxt = fps_slice_choose(slot, G, E, grp, cur, (int)min(run_t / 100ull, 0x1fffffull), lead, lane, &got,
                                        offsets, total_pts, m_cap, m_lo, m_hi);
            if (lane == 0) { sh[0] = nxt; sh[1] = (int)got; }
        }
        first = false;
        __syncthreads();
        const int nxt = sh[0];
        const unsigned got = (unsigned)sh[1];
        if (nxt < 0) break;                                      // nothing runnable and no cloud of its own (cur < 0)
        if (nxt != cur) {
            if (cur >= 0) {
                // ---- save and release the cloud this workgroup leaves
                char* sb = state + (size_t)cur * PCP_FPS_SLICE_STATE_BYTES;
                ((float4*)sb)[tid] = bxyz[tid];
                ((float4*)(sb + 16384))[tid] = ring[tid];
                ((unsigned long long*)(sb + 32768))[tid] = key;
                ((unsigned*)(sb + 40960))[tid] = cy.sec;
                ((unsigned*)(sb + 45056))[tid] = cy.pend;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    ((float*)(sb + 49152))[c * T + tid] = blo[c];
                    ((float*)(sb + 49152))[(3 + c) * T + tid] = bhi[c];
                }
                if (tid < 8) ((float4*)(sb + 73728))[tid] = cy.smv;
                if (tid == 0) {
                    ((int*)(sb + 73728 + 128))[0] = kd; ((int*)(sb + 73728 + 128))[1] = cy.n;
                    ((unsigned*)(sb + 73728 + 128))[2] = (unsigned)run_t; ((unsigned*)(sb + 73728 + 128))[3] = (unsigned)(run_t >> 32);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave: its state stores AND its min-distance stores
                __syncthreads();
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    // into the slot this workgroup has just emptied: its own group's
                    __hip_atomic_store(slot + grp, (1u << 31) | ((unsigned)min(run_t / 100ull, 0x1fffffull) << 10) | (unsigned)cur,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            cur = nxt;
            const long long base = offsets[cur];
            const long long Mll = offsets[cur + 1] - base;
            irow = idx + (size_t)cur * K;
            orow = out_xyz ? out_xyz + (size_t)cur * K * 3 : nullptr;
            if (got == 0u) {
                // ---- a fresh cloud: the prologue of fps_bucket_body
                run_t = 0ull;
                const bool bad = fps_cloud_out_of_bounds(base, Mll, total_pts, m_cap);
                bool mine = true;
                if (bad) {
                    if (m_hi != m_cap) mine = false;
                    else if (tid == 0 && overflow) *overflow = 1;
                }
                M = bad ? 0 : (int)Mll;
                if (!bad && (M < m_lo || M > m_hi)) mine = false;
                kn = !mine ? 0 : (K < M ? K : (M > 0 ? M : 0));
                if (mine)
                    for (int k = kn + tid; k < K; k += T) {              // pytorch3d padding
                        irow[k] = -1;
                        if (orow) { orow[3 * k] = 0.f; orow[3 * k + 1] = 0.f; orow[3 * k + 2] = 0.f; }
                    }
                kd = kn;                                                 // nothing to run unless set up below
                if (kn > 0) {
                    const float* P = xyz + 3 * base;
                    S = pts4 + base;
                    D = md + base;
                    fps_sort_points<T>(P, M, S, D, cellcnt, red, tid, lane, wave);
                    if (ANY) bh = M > 2 * PCP_FPS_WIDE_PTS ? 4 : (M > PCP_FPS_WIDE_PTS ? 2 : 1);
                    const int BS = 64 * bh;
                    NB = (M + BS - 1) / BS;
                    const int j = lane * NW + wave;
                    const int r0 = min(j, NB - 1) * BS;
                    const int rn = min(BS, M - r0);
#pragma unroll
                    for (int c = 0; c < 3; ++c) { blo[c] = INFINITY; bhi[c] = -INFINITY; }
                    bool nan_pt = false;
                    for (int i = 0; i < rn; i += 8) {
                        float4 pt[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) pt[u] = S[r0 + min(i + u, rn - 1)];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            blo[0] = fminf(blo[0], pt[u].x); bhi[0] = fmaxf(bhi[0], pt[u].x);
                            blo[1] = fminf(blo[1], pt[u].y); bhi[1] = fmaxf(bhi[1], pt[u].y);
                            blo[2] = fminf(blo[2], pt[u].z); bhi[2] = fmaxf(bhi[2], pt[u].z);
                            nan_pt = nan_pt || !(pt[u].x == pt[u].x && pt[u].y == pt[u].y && pt[u].z == pt[u].z);
                        }
                    }
                    if (nan_pt) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) { blo[c] = -INFINITY; bhi[c] = INFINITY; }
                    }
                    key = j < NB ? make_key(FLT_MAX, 0u) : 0ull;
                    const float sx = P[0], sy = P[1], sz = P[2];
                    bxyz[tid] = make_float4(sx, sy, sz, 0.f);
                    if (tid == 0) {
                        irow[0] = 0;
                        if (orow) { orow[0] = sx; orow[1] = sy; orow[2] = sz; }
                    }
                    cy.pend = 0u; cy.sec = (unsigned)(key >> 32); cy.n = 1; cy.smv = make_float4(sx, sy, sz, 0.f);
                    kd = 1;
                }
            } else {
                // ---- a cloud another workgroup (or this one, earlier) has worked on
                if (wave == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                M = (int)Mll;
                kn = K < M ? K : M;
                if (ANY) bh = M > 2 * PCP_FPS_WIDE_PTS ? 4 : (M > PCP_FPS_WIDE_PTS ? 2 : 1);
                NB = (M + 64 * bh - 1) / (64 * bh);
                S = pts4 + base;
                D = md + base;
                const char* sb = state + (size_t)cur * PCP_FPS_SLICE_STATE_BYTES;
                bxyz[tid] = ((const float4*)sb)[tid];
                ring[tid] = ((const float4*)(sb + 16384))[tid];
                key = ((const unsigned long long*)(sb + 32768))[tid];
                cy.sec = ((const unsigned*)(sb + 40960))[tid];
                cy.pend = ((const unsigned*)(sb + 45056))[tid];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    blo[c] = ((const float*)(sb + 49152))[c * T + tid];
                    bhi[c] = ((const float*)(sb + 49152))[(3 + c) * T + tid];
                }
                cy.smv = ((const float4*)(sb + 73728))[lane & 7];
                // (vector loads: a scalar load could be served from the scalar cache, which no acquire refreshes)
                kd = (int)__hip_atomic_load((fps_gu32*)(sb + 73728 + 128), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cy.n = (int)__hip_atomic_load((fps_gu32*)(sb + 73728 + 132), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                kd = __builtin_amdgcn_readfirstlane(kd);
                cy.n = __builtin_amdgcn_readfirstlane(cy.n);
                const unsigned r0 = __hip_atomic_load((fps_gu32*)(sb + 73728 + 136), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned r1 = __hip_atomic_load((fps_gu32*)(sb + 73728 + 140), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                run_t = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)r1) << 32) |
                        (unsigned)__builtin_amdgcn_readfirstlane((int)r0);
            }
        }
        // ---- one quantum of rounds
        if (kd < kn) {
            const unsigned long long t_in = wall_clock64();
            // (inlined: as a real call the rounds took their pointers as generic ones — flat loads and stores, which also count
            //  against the LDS wait in front of every barrier — and ran 6 % slower)
            if constexpr (ANY) {
                if (bh == 1)
                    fps_multipick_rounds<T, MODE, 1, true, true, PCP_FPS_MAXP>(kd, kn, t_in + (unsigned long long)quantum, kn, M, NB, S, D, pkey, pxyz, bxyz,
                                                                  sel, okf, ring, irow, orow, tid, lane, wave, key, blo, bhi, cy);
                else if (bh == 2)
                    fps_multipick_rounds<T, MODE, 2, true, true, PCP_FPS_MAXP>(kd, kn, t_in + (unsigned long long)quantum, kn, M, NB, S, D, pkey, pxyz, bxyz,
                                                                  sel, okf, ring, irow, orow, tid, lane, wave, key, blo, bhi, cy);
                else
                    fps_multipick_rounds<T, MODE, 4, true, true, PCP_FPS_MAXP>(kd, kn, t_in + (unsigned long long)quantum, kn, M, NB, S, D, pkey, pxyz, bxyz,
                                                                  sel, okf, ring, irow, orow, tid, lane, wave, key, blo, bhi, cy);
            } else {
                fps_multipick_rounds<T, MODE, BH, true, true, PCP_FPS_MAXP>(kd, kn, t_in + (unsigned long long)quantum, kn, M, NB, S, D, pkey, pxyz, bxyz,
                                                              sel, okf, ring, irow, orow, tid, lane, wave, key, blo, bhi, cy);
            }
        }
        run_t += wall_clock64() - t_top;
        if (kd >= kn) {
            // finished (or not this launch's): its rows are read only after the launch
            __syncthreads();                                             // every wave is out of the rounds' LDS
            cur = -1;
        }
    }
}

template <int T, int MODE, int BH>
__global__ void __launch_bounds__(T)
fps_bucket_sliced(const float* __restrict__ xyz, const long long* __restrict__ offsets, const int B, const int K,
                  long long* __restrict__ idx, float* __restrict__ out_xyz, float4* __restrict__ pts4,
                  float* __restrict__ md, const int m_lo, const int m_hi, const long long total_pts, const int m_cap,
                  int* __restrict__ overflow, unsigned* __restrict__ ctl_all, const int shape, const int defer_mixed,
                  char* __restrict__ state, const int quantum, const int lead)
{
    fps_sliced_body<T, MODE, BH>(xyz, offsets, B, K, idx, out_xyz, pts4, md, m_lo, m_hi, total_pts, m_cap, overflow, ctl_all, shape,
                                 defer_mixed, state, quantum, lead);
}

// (Measured and dropped: the same body held to 64 VGPRs / 80 SGPRs with TWO resident workgroups per CU, for batches of a
//  little more than two clouds per CU — 121 scalar spills and 224 bytes of scratch per lane; 512 clouds on 512 workgroups
//  4.64 ms against the resident launch's 3.37, 600 clouds 5.40-5.70 against 5.03.)

// "deep" shape (multi-chunk buckets, clouds of more than 64*T*CPL points): register-hungry, 4 waves/SIMD.
template <int T, int CPL, int MODE>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_num_sgpr(80)))
fps_bucket_deep(const float* __restrict__ xyz, const long long* __restrict__ offsets, int K,
                long long* __restrict__ idx, float* __restrict__ out_xyz, float4* __restrict__ pts4,
                float* __restrict__ md, int m_lo, int m_hi, int lazy_k, long long total_pts, int m_cap,
                int* __restrict__ overflow, const unsigned* __restrict__ mixed_ctl)
{
    fps_bucket_body<T, CPL, true, MODE>(xyz, offsets, K, idx, out_xyz, pts4, md, m_lo, m_hi, lazy_k, total_pts, m_cap,
                                        overflow);
}

// ---- host entry points -------------------------------------------------------------------------
static int fps_pick_algo(int algo, int64_t max_pts)
{
    if (algo >= 2) return 2;                 // 3 / 4: deferred rounds forced on / off; 5 / 6: multi-pick rounds on / off; 7 / 8: multi-pick + deferred off / on
    if (algo == 1) return 1;
    return max_pts >= 2048 ? 2 : 1;
}

static size_t fps_base_workspace_bytes(int64_t total_pts)
{
    if (total_pts < 0) total_pts = 0;
    return pcp_align256((size_t)total_pts * 16) + pcp_align256((size_t)total_pts * 4) + 256;
}
static int fps_slice_mode(int algo) { return algo == 10 ? 1 : ((algo == 0 || algo == 2) ? 0 : -1); }
static bool fps_mixed_ctl_wanted(int B, int slice, int64_t max_pts);
static bool fps_slice_wanted(int B, int K, int slice, int64_t max_pts);
static bool fps_split_wanted(int B, int K, int slice, int64_t max_pts);
static size_t fps_slice_bytes(int B);

// workspace: sorted records float4[total] | min_d float[total] (| sliced launches: status words | saved round state [B])
extern "C" size_t pcp_fps_workspace_bytes(int B, int64_t total_pts, int64_t max_pts, int K, int algo)
{
    size_t n = fps_base_workspace_bytes(total_pts);
    // quantum-sliced launches (batches of a little more than one cloud per CU) park the round state of a waiting cloud
    if (fps_slice_wanted(B, K, fps_slice_mode(algo), max_pts)) n += fps_slice_bytes(B);
    else if (fps_split_wanted(B, K, fps_slice_mode(algo), max_pts)) n += fps_slice_bytes((B + 1) / 2);
    else if (fps_mixed_ctl_wanted(B, fps_slice_mode(algo), max_pts)) n += fps_slice_bytes(0);   // the control words only
    return n;
}

// Launch shapes.  "wide": 16 waves, one bucket per lane — 64 rows for clouds of up to 65,536 points (the 640x480
// workspace-cropped case), 128 / 256 rows in two / four parts (BH) up to 131,072 / 262,144 points; "deep": two
// buckets per lane, buckets of any multiple of 64 rows, beyond that.  The host only knows an upper bound on the
// cloud sizes, so every shape the bound allows is launched and every workgroup keeps or drops its cloud by its
// actual size (a dropped workgroup exits at once).
struct FpsCall {                 // everything a launch needs besides its shape
    const float* xyz;
    const long long* offsets;
    int B, K;
    long long* idx;
    float* out_xyz;
    float4* pts4;
    float* md;
    long long total_pts;
    int m_cap;                   // largest cloud any launch of this call takes (the top shape's m_hi)
    int* overflow;
    int lazy;
    int mp;                      // multi-pick rounds: 0 = by batch size, 1 = always, 2 = always and with deferred updates, -1 = never
    int slice;                   // quantum-sliced launch: 0 = by batch size, 1 = always (tests), -1 = never
    int any_shape;               // this call ends its wide launches with the ANY-shape one: the per-shape ones leave mixed batches to it
    int ctl_only;                // the workspace holds the control words but no room for parked state (resident-size batches)
    unsigned* slice_ctl;         // [3 wide shapes][PCP_FPS_SLICE_CTL_WORDS slot words], or NULL: the workspace has no room
    char* slice_state;           // [B][PCP_FPS_SLICE_STATE_BYTES]
    hipStream_t st;
};

#ifndef PCP_FPS_ANY_SHAPE
#define PCP_FPS_ANY_SHAPE 1      // 0: no ANY-shape sliced launch (tuning builds): mixed batches run shape after shape
#endif
#ifndef PCP_FPS_SLICE_MIN_B
#define PCP_FPS_SLICE_MIN_B 257          // sliced launches for batches of more clouds than CUs ...
#endif
#ifndef PCP_FPS_SLICE_TOP_B
#define PCP_FPS_SLICE_TOP_B 336          // ... up to where two resident workgroups per CU do as well (350 clouds: 2.85-2.94 against 2.91 ms)
#endif
#ifndef PCP_FPS_SLICE_TOP_B4
#define PCP_FPS_SLICE_TOP_B4 512         // quartered buckets (clouds of 131,073-262,144 points): the resident launch there is the one-pick
                                         // kernel (config 5, K = 8192: 300 clouds 37.4 -> 22.7 ms, 400: 42.4 -> 33.0, 448: 45.3 -> 33.8)
#endif
#define PCP_FPS_SLICE_MIN_K 256
#ifndef PCP_FPS_SLICE_QUANTUM_US
#define PCP_FPS_SLICE_QUANTUM_US 150      // 300 clouds, K = 1024: 2.59 / 2.49 / 2.45 / 2.40 / 2.42 / 2.40 ms at 25 / 50 / 100 / 150 / 200 / 300 us
#endif

static int fps_cu_count()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
        cus[dev] = n > 0 ? n : -1;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}

// (max_pts: clouds beyond 131,072 points — quartered buckets — are sliced up to a larger batch, see PCP_FPS_SLICE_TOP_B4)
static bool fps_slice_wanted(int B, int K, int slice, int64_t max_pts)
{
    if (slice < 0 || B > PCP_FPS_SLICE_MAX_B || K >= (1 << 21)) return false;
    if (slice > 0) return B >= 2 && B <= 960;
    const int top = max_pts > 2 * 65536 ? PCP_FPS_SLICE_TOP_B4 : PCP_FPS_SLICE_TOP_B;
    return B >= PCP_FPS_SLICE_MIN_B && B <= top && K >= PCP_FPS_SLICE_MIN_K;
}

// two sliced launches over the halves of a batch of 513-624 clouds (by batch size only: algo 0 / 2).  Measured, K = 1024,
// resident launch -> two sliced ones: 513 clouds 4.86 -> 4.17 ms, 560: 4.95 -> 4.59, 600: 5.00 -> 4.83, 672: 5.11 -> 5.38
#ifndef PCP_FPS_SPLIT_TOP_B
#define PCP_FPS_SPLIT_TOP_B 624
#endif
static bool fps_split_wanted(int B, int K, int slice, int64_t max_pts)
{
    return slice == 0 && B > 2 * (PCP_FPS_SLICE_MIN_B - 1) && B <= PCP_FPS_SPLIT_TOP_B &&
           fps_slice_wanted((B + 1) / 2, K, 0, max_pts) && !fps_slice_wanted(B, K, 0, max_pts);
}

// A resident-size batch (one workgroup per CU at most) whose size bound allows more than one wide shape: the workspace
// carries the control words, fps_slice_prep counts the clouds per shape, and a batch that does mix shapes runs as one
// any-shape launch (nobody waits, so no parked state) instead of shape after shape.
static bool fps_mixed_ctl_wanted(int B, int slice, int64_t max_pts)
{
    return PCP_FPS_ANY_SHAPE && slice == 0 && B >= 2 && B < PCP_FPS_SLICE_MIN_B && max_pts > 65536;
}

static size_t fps_slice_bytes(int B)
{
    return pcp_align256((size_t)3 * PCP_FPS_SLICE_CTL_WORDS * 4) + (size_t)B * PCP_FPS_SLICE_STATE_BYTES;
}

template <int MODE, int BH>
static int launch_sliced(const FpsCall& c, int m_lo, int m_hi, int shape)
{
    constexpr int T = 1024;
    const int cus = fps_cu_count();
    if (cus <= 0) return PCP_ERR_NO_DEVICE;
    // forced (tests): fewer workgroups than clouds even for a small batch, so that clouds do change workgroups
    int grid = c.slice > 0 ? std::max(1, std::min(cus, c.B - c.B / 4)) : std::min(cus, c.B);
    // short quanta keep a group's clouds close together; a quantum's end costs a slot read (~2 us)
    int quantum = PCP_FPS_SLICE_QUANTUM_US;
#ifdef PCP_FPS_SLICE_TUNE            // tuning builds only
    if (const char* e = getenv("PCP_FPS_SLICE_GRID")) grid = std::max(1, std::min(atoi(e), c.B));
    if (const char* e = getenv("PCP_FPS_SLICE_Q")) quantum = std::max(1, atoi(e));
#endif
    int lead = quantum / 2;
#ifdef PCP_FPS_SLICE_TUNE
    if (const char* e = getenv("PCP_FPS_SLICE_LEAD")) lead = atoi(e);
#endif
    // every extra cloud needs a workgroup of its group (E <= G) and a waiting slot
    if (c.B - grid > grid) grid = (c.B + 1) / 2;
    if (c.B - grid > PCP_FPS_SLICE_MAX_SLOTS || c.K >= (1 << 21)) return PCP_ERR_INVALID_ARG;
    // more than half a CU's LDS: one workgroup per CU whatever the register count
    const size_t lds = std::max(fps_bucket_lds_bytes<T, 1>(), (size_t)96 * 1024);
    auto kern = fps_bucket_sliced<T, MODE, BH>;
    PCP_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // (the slot words of all three shapes are zeroed by fps_slice_prep; the ANY-shape launch, BH = 0, waits in shape 0's)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, c.st, c.xyz, c.offsets, c.B, c.K, c.idx, c.out_xyz, c.pts4, c.md,
                       m_lo, m_hi, c.total_pts, c.m_cap, c.overflow, c.slice_ctl, shape, (c.any_shape == 1 || c.any_shape == 3) ? 1 : 0, c.slice_state, quantum * 100, lead);
    return pcp_launch_status();
}


#ifndef PCP_FPS_MP_MAX_B
#define PCP_FPS_MP_MAX_B 256     // multi-pick rounds for batches of at most this many clouds (one workgroup per CU)
#endif
template <int T, int CPL, bool MULTI, int MODE, int BH = 1>
static int launch_bucket_as(const FpsCall& c, int m_lo, int m_hi)
{
    if constexpr (!MULTI && CPL == 1) {
        if (c.mp >= 0 && c.slice_ctl && fps_slice_wanted(c.B, c.K, c.slice, BH == 4 ? 4 * 65536 : 65536) &&
            (c.slice > 0 || c.B > fps_cu_count()))
            return launch_sliced<MODE, BH>(c, m_lo, m_hi, BH == 1 ? 0 : (BH == 2 ? 1 : 2));
    }
    constexpr size_t lds = fps_bucket_lds_bytes<T, CPL>();
    void (*kern)(const float*, const long long*, int, long long*, float*, float4*, float*, int, int, int, long long,
                 int, int*, const unsigned*);
    if constexpr (MULTI) kern = fps_bucket_deep<T, CPL, MODE>;
    else {
        kern = fps_bucket_wide<T, CPL, MODE, BH>;
        if constexpr (CPL == 1) {
            if (c.mp >= 0) {
                if (c.B < PCP_FPS_MP_DEFER_MIN_B && c.mp < 2) kern = fps_bucket_wide_mp<T, CPL, MODE, BH, false>;   // one workgroup per CU
                else if (c.B <= PCP_FPS_MP_MAX_B) kern = fps_bucket_wide_mp<T, CPL, MODE, BH, true>;
                else if (BH <= 2) kern = fps_bucket_wide_mp2<T, CPL, MODE, (BH <= 2 ? BH : 1)>;  // two (the quartered-bucket shape spills at 64 VGPRs)
                else if (c.mp > 0) kern = fps_bucket_wide_mp<T, CPL, MODE, BH, true>;
            }
        }
    }
    // deferred updates pay once the batch keeps the fabric busy (measured: from about one cloud per CU)
    // (lazy: 0 = by batch size, 1 = always, -1 = never)
    // (with multi-pick rounds after them only once two workgroups share a CU: measured 1.81 vs 1.90 ms at 192 clouds,
    //  2.14 vs 2.17 at 256, 3.02 vs 2.91 at 300 without / with the deferred rounds)
    const int lazy_min_b = c.mp >= 0 ? PCP_FPS_MP_MAX_B + 1 : PCP_FPS_LAZY_MIN_B;
    // how many: 256 ahead of the one-pick loop; ahead of the multi-pick rounds, which take over the mid-range rounds better,
    // 128 up to 400 clouds (300 clouds: 2.83 ms with 96-128, 2.87 with 192, 2.93 with 256; 512 clouds: 3.57 / 3.51 / 3.51)
    const int lazy_n = (c.mp >= 0 && c.B <= 400) ? PCP_FPS_LAZY_ROUNDS / 2 : PCP_FPS_LAZY_ROUNDS;
    const int lazy_k = (!MULTI && CPL == 1 && c.lazy >= 0 && (c.lazy > 0 || c.B >= lazy_min_b)) ? lazy_n : 0;
    if (lds > 48 * 1024)         // the only process-wide state the library touches: a per-kernel attribute, same value every time
        PCP_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(c.B), dim3(T), lds, c.st, c.xyz, c.offsets, c.K, c.idx, c.out_xyz, c.pts4, c.md,
                       m_lo, m_hi, lazy_k, c.total_pts, c.m_cap, c.overflow,
                       (const unsigned*)(c.any_shape >= 2 && !MULTI && CPL == 1 ? c.slice_ctl : nullptr));
    return pcp_launch_status();
}

#ifndef PCP_FPS_SHAPE
#define PCP_FPS_SHAPE 0          // 0: wide for small clouds + deep for large; 2: deep only (tuning builds)
#endif
#define PCP_FPS_WIDE_MAX (64 * 1024)
#ifndef PCP_FPS_SPLIT4
#define PCP_FPS_SPLIT4 1         // 0: clouds above 131,072 points go to the deep shape (tuning builds)
#endif
#ifndef PCP_FPS_DEEP_1024_MAX_B
#define PCP_FPS_DEEP_1024_MAX_B 256
#endif

template <int MODE>
static int launch_bucket(FpsCall c, int64_t max_pts)
{
    constexpr int W = PCP_FPS_WIDE_MAX;
    constexpr int split_max = PCP_FPS_SPLIT4 ? 4 * W : 2 * W;
    if (PCP_FPS_SHAPE == 2) {
        c.m_cap = W;
        return launch_bucket_as<512, 2, false, MODE>(c, 0, W);
    }
    // the top of the size range the launches below cover: a cloud beyond it (the caller understated max_pts) is
    // answered with an all -1 row and the overflow flag by the last launch, never skipped
    c.m_cap = max_pts <= W ? W : (max_pts <= 2 * W ? 2 * W : (max_pts <= split_max ? split_max : 0x7fffffff));
    // A batch whose clouds fall into more than one wide size range: when all the wide shapes of this call go as sliced
    // launches, an ANY-shape sliced launch follows them and takes such a batch whole (the per-shape ones then exit at once).
    // (1: the per-shape launches are sliced ones; 2: they are resident multi-pick launches of at most one workgroup per CU,
    //  and the any-shape launch has a workgroup for every cloud - nobody waits, no state is parked)
    c.any_shape = 0;
    if (PCP_FPS_ANY_SHAPE && PCP_FPS_SPLIT4 && c.slice_ctl && c.mp >= 0 && max_pts > W) {
        if (fps_slice_wanted(c.B, c.K, c.slice, W) && (c.slice > 0 || c.B > fps_cu_count())) c.any_shape = 1;
        else if (c.ctl_only && c.slice == 0 && c.mp == 0 && c.B >= 2 && c.B <= fps_cu_count() && c.B <= PCP_FPS_MP_MAX_B) c.any_shape = 2;
        // 3: 337-512 clouds with a size bound beyond 131,072 points (the workspace then has room for parked state, see
        //    PCP_FPS_SLICE_TOP_B4): the per-shape launches are two resident workgroups per CU for whole and halved buckets and a
        //    sliced launch for quartered ones; a mixed batch goes as one any-shape SLICED launch instead (400 tabletop frames:
        //    4.65 ms shape after shape)
        else if (!c.ctl_only && c.slice == 0 && c.mp == 0 && c.B > PCP_FPS_SLICE_TOP_B && c.B > fps_cu_count() &&
                 max_pts > 2 * W && fps_slice_wanted(c.B, c.K, 0, 4 * W)) c.any_shape = 3;
    }
    if (c.slice_ctl) {           // every waiting slot of the three sliced shapes: "extra cloud, not started"; clouds per shape
        hipLaunchKernelGGL(fps_slice_prep, dim3(1), dim3(256), 0, c.st, c.offsets, c.B, c.total_pts, c.m_cap, c.slice_ctl);
        if (pcp_launch_status() != PCP_OK) return PCP_ERR_LAUNCH;
    }
    int rc = launch_bucket_as<1024, 1, false, MODE>(c, 0, W);
    if (rc != PCP_OK || max_pts <= W) return rc;
    // up to 131,072 / 262,144 points: the same shape with buckets of two / four 64-row parts, streamed side by side
    rc = launch_bucket_as<1024, 1, false, MODE, 2>(c, W + 1, 2 * W);
    if (rc != PCP_OK) return rc;
    if (max_pts <= 2 * W) return c.any_shape ? launch_sliced<MODE, 0>(c, 0, 2 * W, 0) : rc;
    if (PCP_FPS_SPLIT4) {
        rc = launch_bucket_as<1024, 1, false, MODE, 4>(c, 2 * W + 1, 4 * W);
        if (rc != PCP_OK) return rc;
        if (c.any_shape) rc = launch_sliced<MODE, 0>(c, 0, 4 * W, 0);
        if (rc != PCP_OK || max_pts <= 4 * W) return rc;
    }
    // larger clouds: a batch that leaves CUs free (one workgroup per CU is enough) gets 1024 threads per cloud —
    // twice the waves streaming and half the rows per bucket; bigger batches keep two 512-thread workgroups per CU
    if (c.B <= PCP_FPS_DEEP_1024_MAX_B)
        return launch_bucket_as<1024, 2, true, MODE>(c, split_max + 1, 0x7fffffff);
    return launch_bucket_as<512, 2, true, MODE>(c, split_max + 1, 0x7fffffff);
}

static int fps_launch_batch(const float* xyz, const int64_t* offsets, int B, int64_t total_pts, int64_t max_pts,
                            int K, int dist_mode, int algo, int64_t* idx, float* out_xyz, int32_t* overflow,
                            void* ws, size_t base_bytes, int slice_kind, hipStream_t st);

// shared by pcp_fps_batched and its no-allocation twin: argument checks, workspace carve-up, launches
static int fps_batched_impl(const float* xyz, const int64_t* offsets, int B, int64_t total_pts, int64_t max_pts,
                            int K, int dist_mode, int algo, int64_t* idx, float* out_xyz, int32_t* overflow,
                            void* ws, size_t ws_bytes, hipStream_t st, int slice_allow)
{
    // slice_allow 2: quantum-sliced launches and the control words of the any-shape launch; 1: the control words only (the
    // no-allocation entry point: parked round state would not fit its array); 0: neither
    const bool allow_slice = slice_allow == 2, allow_ctl = slice_allow >= 1;
    if (B < 0 || K < 0 || total_pts < 0 || max_pts < 0 || !offsets || (dist_mode != 0 && dist_mode != 1) ||
        algo < 0 || algo > 10)
        return PCP_ERR_INVALID_ARG;
    if ((total_pts > 0 && !xyz) || (B > 0 && K > 0 && !idx)) return PCP_ERR_INVALID_ARG;
    if (max_pts > 0x7fffffffLL) return PCP_ERR_INVALID_ARG;
    const size_t base_bytes = fps_base_workspace_bytes(total_pts);
    // a batch of a little more than two clouds per CU goes as two launches over its halves, each of them sliced (the
    // resident launch runs the clouds beyond 512 as a second wave: 600 clouds 5.00 ms, two sliced launches of 300: 4.83)
    if (allow_slice && fps_split_wanted(B, K, fps_slice_mode(algo), max_pts) && fps_pick_algo(algo, max_pts) == 2) {
        const int h = (B + 1) / 2;
        if (!ws || ((uintptr_t)ws & 255) || ws_bytes < base_bytes + fps_slice_bytes(h)) return PCP_ERR_WORKSPACE;
        if (overflow) PCP_FILL_TRY(overflow, 0u, 1, st);
        int rc = fps_launch_batch(xyz, offsets, h, total_pts, max_pts, K, dist_mode, algo, idx, out_xyz, overflow, ws, base_bytes,
                                  1, st);
        if (rc != PCP_OK) return rc;
        return fps_launch_batch(xyz, offsets + h, B - h, total_pts, max_pts, K, dist_mode, algo, idx + (size_t)h * K,
                                out_xyz ? out_xyz + (size_t)h * K * 3 : nullptr, overflow, ws, base_bytes,
                                fps_slice_wanted(B - h, K, 0, max_pts) ? 1 : 0, st);
    }
    const bool slice = allow_slice && fps_slice_wanted(B, K, fps_slice_mode(algo), max_pts);
    if (!ws || ((uintptr_t)ws & 255) || ws_bytes < base_bytes + (slice ? fps_slice_bytes(B) : 0))
        return PCP_ERR_WORKSPACE;
    // control words only (mixed-size batches of resident size): used when the workspace has them - pcp_fps_workspace_bytes
    // asks for them -, done without otherwise (shape after shape: slower on a mixed batch, same picks)
    const bool ctl_only = allow_ctl && !slice && fps_mixed_ctl_wanted(B, fps_slice_mode(algo), max_pts) &&
                          fps_pick_algo(algo, max_pts) == 2 && ws_bytes >= base_bytes + fps_slice_bytes(0);
    if (overflow) PCP_FILL_TRY(overflow, 0u, 1, st);
    if (B == 0 || K == 0) return PCP_OK;
    return fps_launch_batch(xyz, offsets, B, total_pts, max_pts, K, dist_mode, algo, idx, out_xyz, overflow, ws, base_bytes,
                            slice ? 1 : (ctl_only ? 2 : 0), st);
}

// the launches for clouds [0, B) of `offsets` (arguments checked, overflow flag cleared by the caller)
static int fps_launch_batch(const float* xyz, const int64_t* offsets, int B, int64_t total_pts, int64_t max_pts,
                            int K, int dist_mode, int algo, int64_t* idx, float* out_xyz, int32_t* overflow,
                            void* ws, size_t base_bytes, int slice_kind, hipStream_t st)
{
    FpsCall c;
    c.xyz = xyz; c.offsets = (const long long*)offsets; c.B = B; c.K = K; c.idx = (long long*)idx; c.out_xyz = out_xyz;
    c.pts4 = (float4*)ws;
    c.md = (float*)((char*)ws + pcp_align256((size_t)total_pts * 16));
    c.total_pts = total_pts; c.m_cap = 0x7fffffff; c.overflow = overflow; c.any_shape = 0; c.ctl_only = 0;
    c.lazy = (algo == 3 || algo == 8) ? 1 : ((algo == 4 || algo == 7) ? -1 : 0);
    c.mp = (algo == 5 || algo == 7 || algo == 8) ? 1 : (algo == 9 ? 2 : (algo == 6 ? -1 : 0));
    c.slice = slice_kind ? fps_slice_mode(algo) : -1;
    c.slice_ctl = slice_kind ? (unsigned*)((char*)ws + base_bytes) : nullptr;
    c.slice_state = slice_kind == 1 ? (char*)ws + base_bytes + pcp_align256((size_t)3 * PCP_FPS_SLICE_CTL_WORDS * 4) : nullptr;
    c.ctl_only = slice_kind == 2 ? 1 : 0;
    c.st = st;
    if (fps_pick_algo(algo, max_pts) == 2)
        return dist_mode == 0 ? launch_bucket<0>(c, max_pts) : launch_bucket<1>(c, max_pts);
    if (dist_mode == 0)
        hipLaunchKernelGGL(fps_brute<0>, dim3(B), dim3(1024), 0, st, xyz, c.offsets, K, c.idx, out_xyz, c.md,
                           c.total_pts, overflow);
    else
        hipLaunchKernelGGL(fps_brute<1>, dim3(B), dim3(1024), 0, st, xyz, c.offsets, K, c.idx, out_xyz, c.md,
                           c.total_pts, overflow);
    return pcp_launch_status();
}

extern "C" int pcp_fps_batched(const float* xyz, const int64_t* offsets, int B, int64_t total_pts,
                               int64_t max_pts, int K, int dist_mode, int algo, int64_t* idx,
                               float* out_xyz, int32_t* overflow, void* ws, size_t ws_bytes, void* stream)
{
    return fps_batched_impl(xyz, offsets, B, total_pts, max_pts, K, dist_mode, algo, idx, out_xyz, overflow, ws,
                            ws_bytes, (hipStream_t)stream, 2);
}

// ---- no-allocation twin of pcp_fps_batched -----------------------------------------------------------------------
// For callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid
// it): the sorted-record / min-distance workspace lives in a file-scope device array instead of a caller buffer.
// Every GPU the library runs on holds its own copy of the array (32 MiB) at its own address until the process ends.
// In host code the array's NAME is only a stand-in; the device address comes from hipGetSymbolAddress, once per
// device, with that device current, and only that pointer is ever handed to kernels.
#define PCP_FPS_STATIC_BYTES (32u * 1024u * 1024u)
static __device__ __align__(256) unsigned char pcp_fps_scratch[PCP_FPS_STATIC_BYTES];

#define PCP_FPS_STATIC_MAX_DEVICES 64
static void* g_fps_scratch_ptr[PCP_FPS_STATIC_MAX_DEVICES];       // device address of pcp_fps_scratch on device i, or NULL
static pthread_mutex_t g_fps_scratch_lock = PTHREAD_MUTEX_INITIALIZER;

// device address of the current device's copy of the scratch array (resolved on the first call per device)
static int fps_static_scratch(void** out)
{
    int dev = -1;
    PCP_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= PCP_FPS_STATIC_MAX_DEVICES) return PCP_ERR_NO_DEVICE;
    pthread_mutex_lock(&g_fps_scratch_lock);
    void* p = g_fps_scratch_ptr[dev];
    if (!p) {
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(pcp_fps_scratch));
        if (e != hipSuccess || !p || ((uintptr_t)p & 255)) {
            (void)hipGetLastError();
            pthread_mutex_unlock(&g_fps_scratch_lock);
            return PCP_ERR_LAUNCH;
        }
        g_fps_scratch_ptr[dev] = p;
    }
    pthread_mutex_unlock(&g_fps_scratch_lock);
    *out = p;
    return PCP_OK;
}

static_assert(((size_t)PCP_FPS_STATIC_MAX_PTS * 16 + 255) / 256 * 256 + ((size_t)PCP_FPS_STATIC_MAX_PTS * 4 + 255) / 256 * 256 + 256
                  <= PCP_FPS_STATIC_BYTES,
              "PCP_FPS_STATIC_MAX_PTS points must fit the static scratch array");

extern "C" int pcp_fps_static_prepare(void)
{
    void* ws = nullptr;
    return fps_static_scratch(&ws);
}

extern "C" int pcp_fps_batched_static(const float* xyz, const int64_t* offsets, int B, int64_t total_pts,
                                      int64_t max_pts, int K, int dist_mode, int algo, int64_t* idx,
                                      float* out_xyz, int32_t* overflow, void* stream)
{
    if (total_pts > (int64_t)PCP_FPS_STATIC_MAX_PTS) return PCP_ERR_CAPACITY;     // beyond the static scratch: use pcp_fps_batched
    if (total_pts < 0) return PCP_ERR_INVALID_ARG;
    void* ws = nullptr;
    const int rc = fps_static_scratch(&ws);
    if (rc != PCP_OK) return rc;
    // Nothing in the workspace has to start at zero: every launch rewrites the records and min-distances of the rows
    // it reads before it reads them, so the previous call's contents are never observed.
    // (no quantum-sliced launches here: their parked round state would not fit the array beside the records.  The 3.8 KB of
    //  control words behind the records do - fps_slice_prep rewrites them in every call -, so a mixed-size batch of 2-256
    //  clouds still runs as one any-shape launch with a workgroup per cloud, where nothing is parked.)
    static_assert((size_t)PCP_FPS_STATIC_MAX_PTS * 20 + 1024 + 3 * PCP_FPS_SLICE_CTL_WORDS * 4 + 256 <= PCP_FPS_STATIC_BYTES,
                  "records, min-distances and the control words fit the static array");
    return fps_batched_impl(xyz, offsets, B, total_pts, max_pts, K, dist_mode, algo, idx, out_xyz, overflow, ws,
                            fps_base_workspace_bytes(total_pts) + fps_slice_bytes(0), (hipStream_t)stream, 1);
}
