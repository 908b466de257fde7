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

// ---- exact bucket-pruned FPS ------------------------------------------------------------------
// Same picks as the brute-force scan, far fewer distance evaluations (≈20-35·M instead of
// (K-1)·M on the benchmark frames).  One workgroup owns one cloud:
//   prologue  bounding box → uniform grid of ≤ T*CPT cells → LDS counting sort of the points
//             into {x,y,z,orig_idx} records grouped by cell (+ one min_d float per point);
//             thread t owns cells t, t+T, … and keeps their tight boxes in registers.
//   round     A  every owner tests its cells: lb = dist2(s, clamp(s, box)).  The same rounded
//                operation sequence as the point distance and monotone rounding give
//                lb <= dist2(s,p) for every p in the box, so a cell with lb >= cell_max cannot
//                change and is skipped; the others are queued.  The arg-max over the owner's
//                skipped cells is reduced per wave.
//             B  waves drain the queue: one cell = ≤64 points per step, min_d updated in
//                place, new (cell_max, lowest index, xyz) written to LDS.
//             C  arg-max over the 2*NW wave partials gives the next sample and its xyz
//                without touching global memory.
// Two workgroup barriers and one global-memory round trip per round.
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

// wave-uniform maximum of one u64 key per lane
__device__ __forceinline__ unsigned long long wave_umax_key(unsigned long long k)
{
    const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    const unsigned m = wave_umax_u32(hi);
    const unsigned l = wave_umax_u32(hi == m ? lo : 0u);
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

__device__ __forceinline__ int grid_cell(float v, float lo, float inv_h, int g)
{
    int c = (int)((v - lo) * inv_h);
    return min(g - 1, max(0, c));
}

#ifndef PCP_BUCKET_TARGET_PTS
#define PCP_BUCKET_TARGET_PTS 16       // aim for about this many points per cell
#endif

// Diagnostic build only (make stamps): per-phase cycle sums printed by workgroup 0.
#ifdef PCP_FPS_STAMPS
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(var) do { } while (0)
#endif
//       // aim for about this many points per cell

template <int T, int CPT>
constexpr size_t fps_bucket_lds_bytes()
{
    constexpr size_t CMAX = (size_t)T * CPT, NW = T / 64;
    return CMAX * 8 + CMAX * 12 + (CMAX + 4) * 4 + CMAX * 2 + 4 * NW * 8 + 4 * NW * 4 + 16 + 6 * NW * 4 + 64;
}

template <int T, int CPT, int MODE>
__global__ void __launch_bounds__(T)
fps_bucket(const float* __restrict__ xyz, const long long* __restrict__ offsets, int K,
           long long* __restrict__ idx, float* __restrict__ out_xyz, float4* __restrict__ pts4,
           float* __restrict__ md)
{
    constexpr int NW = T / 64, CMAX = T * CPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* key = (unsigned long long*)smem;                 // [CMAX] cell arg-max keys
    float* cx = (float*)(key + CMAX);                                    // [CMAX] xyz of each cell's arg-max
    float* cy = cx + CMAX;
    float* cz = cy + CMAX;
    int* cstart = (int*)(cz + CMAX);                                     // [CMAX+1] first sorted row of a cell
    unsigned short* queue = (unsigned short*)(cstart + CMAX + 4);        // [CMAX] active cells of the round
    unsigned long long* pkey = (unsigned long long*)(queue + CMAX);      // [2][2*NW] wave partials
    int* pcell = (int*)(pkey + 4 * NW);                                  // [2][2*NW]
    int* qcount = pcell + 4 * NW;                                        // [2] (+2 pad)
    float* red = (float*)(qcount + 4);                                   // [6*NW] / scan scratch
    int* cursor = (int*)key;                                             // prologue alias

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = offsets[b];
    const int M = (int)(offsets[b + 1] - base);
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

#ifdef PCP_FPS_STAMPS
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, tp0 = 0, tp1 = 0;
    unsigned long long sA = 0, sW1 = 0, sB = 0, sW2 = 0, sC = 0, sQ = 0, sPts = 0, sMem = 0, sPass = 0;
    STAMP(tp0);
#endif
    // ---- prologue 1: bounding box -------------------------------------------------------------
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    constexpr int PU = 8;                               // points in flight per thread in the prologue
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
    for (int c = tid; c < CMAX + 1; c += T) cstart[c] = 0;
    for (int c = tid; c < CMAX; c += T) cursor[c] = 0;
    if (tid < 4) qcount[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        lo[c] = red[c]; hi[c] = red[3 + c];
        for (int w = 1; w < NW; ++w) { lo[c] = fminf(lo[c], red[w * 6 + c]); hi[c] = fmaxf(hi[c], red[w * 6 + 3 + c]); }
    }
    // grid: roughly cubic cells, at most CMAX of them, about PCP_BUCKET_TARGET_PTS points each
    int g[3] = {1, 1, 1};
    float inv_h[3] = {0.f, 0.f, 0.f};
    {
        float e[3];
        float emax = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { e[c] = hi[c] - lo[c]; if (!(e[c] > 0.f) || !(e[c] < INFINITY)) e[c] = 0.f; emax = fmaxf(emax, e[c]); }
        int target = M / PCP_BUCKET_TARGET_PTS;
        target = min(CMAX, max(1, target));
        if (emax > 0.f && target > 1) {
            float vol = 1.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) vol *= fmaxf(e[c], emax * 1e-3f);
            float h = cbrtf(vol / (float)target);
            for (int it = 0; it < 200; ++it) {
                long long prod = 1;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float gc = ceilf(e[c] / h);
                    g[c] = (int)fminf(fmaxf(gc, 1.f), (float)CMAX);
                    prod *= g[c];
                }
                if (prod <= CMAX) break;
                h *= 1.04f;
                if (it == 199) { g[0] = g[1] = g[2] = 1; }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) inv_h[c] = e[c] > 0.f ? (float)g[c] / e[c] : 0.f;
        }
    }
    const int C = g[0] * g[1] * g[2];

    // ---- prologue 2: histogram, scan, scatter (LDS counting sort) ---------------------------------
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
            const int cell = (grid_cell(v[u][0], lo[0], inv_h[0], g[0]) * g[1] +
                              grid_cell(v[u][1], lo[1], inv_h[1], g[1])) * g[2] +
                             grid_cell(v[u][2], lo[2], inv_h[2], g[2]);
            if (p0 + u * T < M) atomicAdd(&cstart[cell], 1);
        }
    }
    __syncthreads();
    {
        int local[CPT];
        int sum = 0;
#pragma unroll
        for (int j = 0; j < CPT; ++j) { local[j] = cstart[tid * CPT + j]; sum += local[j]; }
        __syncthreads();
        int total;
        int ex = pcp_block_excl_scan(sum, (int*)red, &total);
#pragma unroll
        for (int j = 0; j < CPT; ++j) { cstart[tid * CPT + j] = ex; cursor[tid * CPT + j] = ex; ex += local[j]; }
        if (tid == T - 1) cstart[CMAX] = ex;
    }
    __syncthreads();
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
            const int cell = (grid_cell(v[u][0], lo[0], inv_h[0], g[0]) * g[1] +
                              grid_cell(v[u][1], lo[1], inv_h[1], g[1])) * g[2] +
                             grid_cell(v[u][2], lo[2], inv_h[2], g[2]);
            pos[u] = (p0 + u * T < M) ? atomicAdd(&cursor[cell], 1) : -1;
        }
#pragma unroll
        for (int u = 0; u < PU; ++u)
            if (pos[u] >= 0) {
                S[pos[u]] = make_float4(v[u][0], v[u][1], v[u][2], __int_as_float(p0 + u * T));
                D[pos[u]] = FLT_MAX;
            }
    }
    __syncthreads();

    // ---- prologue 3: owners take their cells' tight boxes into registers --------------------------
    float blo[CPT][3], bhi[CPT][3];
    {
        int st[CPT], cn[CPT], cmaxn = 0;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int cell = tid + j * T;
            st[j] = cstart[cell];
            cn[j] = cstart[cell + 1] - st[j];
            cmaxn = max(cmaxn, cn[j]);
#pragma unroll
            for (int c = 0; c < 3; ++c) { blo[j][c] = INFINITY; bhi[j][c] = -INFINITY; }
        }
        for (int i = 0; i < cmaxn; i += 2) {              // CPT x 2 independent loads in flight
            float4 pt[CPT][2];
#pragma unroll
            for (int j = 0; j < CPT; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u)          // unconditional, clamped: loads stay in flight together
                    pt[j][u] = S[min(st[j] + min(i + u, max(cn[j] - 1, 0)), M - 1)];
#pragma unroll
            for (int j = 0; j < CPT; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (i + u < cn[j]) {
                        blo[j][0] = fminf(blo[j][0], pt[j][u].x); bhi[j][0] = fmaxf(bhi[j][0], pt[j][u].x);
                        blo[j][1] = fminf(blo[j][1], pt[j][u].y); bhi[j][1] = fmaxf(bhi[j][1], pt[j][u].y);
                        blo[j][2] = fminf(blo[j][2], pt[j][u].z); bhi[j][2] = fmaxf(bhi[j][2], pt[j][u].z);
                    }
        }
        (void)C;
    }
    __syncthreads();                                   // cursor (aliases key) no longer needed
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int cell = tid + j * T;
        key[cell] = cstart[cell + 1] > cstart[cell] ? make_key(FLT_MAX, 0u) : 0ull;
    }
    float sx = P[0], sy = P[1], sz = P[2];
    int last = 0;
    __syncthreads();

    STAMP(tp1);
    // ---- rounds -------------------------------------------------------------------------------------
    for (int k = 0;; ++k) {
        if (tid == 0) {
            irow[k] = last;
            if (orow) { orow[3 * k] = sx; orow[3 * k + 1] = sy; orow[3 * k + 2] = sz; }
        }
        if (k + 1 >= kn) break;
        const int par = k & 1;
        STAMP(t0);
        // phase A: cell tests by the owners
        unsigned long long ka = 0ull;
        int ka_cell = 0;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int cell = tid + j * T;
            const unsigned long long kj = key[cell];
            const float cmax = __uint_as_float((unsigned)(kj >> 32));
            const float qx = fminf(fmaxf(sx, blo[j][0]), bhi[j][0]);
            const float qy = fminf(fmaxf(sy, blo[j][1]), bhi[j][1]);
            const float qz = fminf(fmaxf(sz, blo[j][2]), bhi[j][2]);
            const float lb = dist2<MODE>(sx, sy, sz, qx, qy, qz);
            if (lb < cmax) {
                const int slot = atomicAdd(&qcount[par], 1);
                queue[slot] = (unsigned short)cell;
            } else if (kj > ka) {
                ka = kj;
                ka_cell = cell;
            }
        }
        {
            const unsigned long long wk = wave_umax_key(ka);
            if (ka == wk) { pkey[par * 2 * NW + wave] = wk; pcell[par * 2 * NW + wave] = ka_cell; }
        }
        STAMP(t1);
        __syncthreads();
        STAMP(t2);
        // phase B: update the points of the queued cells
        if (tid == 0) qcount[par ^ 1] = 0;
        const int nq = qcount[par];
        unsigned long long wbest = 0ull;
        int wbest_cell = 0;
        constexpr int BU = 4;                           // cells in flight per wave
        for (int e0 = wave; e0 < nq; e0 += NW * BU) {
            int cell[BU], st[BU], cnt[BU];
            float4 pt[BU];
            float m[BU];
#pragma unroll
            for (int u = 0; u < BU; ++u) {
                const int e = e0 + u * NW;
                const bool ok = e < nq;
                cell[u] = (int)queue[ok ? e : e0];
                st[u] = cstart[cell[u]];
                cnt[u] = ok ? cstart[cell[u] + 1] - st[u] : 0;
            }
#pragma unroll
            for (int u = 0; u < BU; ++u) {                  // unconditional, clamped (no per-load waits)
                const int r = st[u] + min(lane, max(cstart[cell[u] + 1] - st[u] - 1, 0));
                pt[u] = S[r];
                m[u] = D[r];
            }
#ifdef PCP_FPS_STAMPS
            { unsigned long long ta, tb; STAMP(ta); __builtin_amdgcn_s_waitcnt(0x0F70); STAMP(tb); sMem += tb - ta; sPass += 1; }
#endif
#pragma unroll
            for (int u = 0; u < BU; ++u) {
                if (cnt[u] == 0) continue;              // wave-uniform
                unsigned long long bk = 0ull;
                float bx = 0.f, by = 0.f, bz = 0.f;
                if (lane < cnt[u]) {
                    const float d = dist2<MODE>(sx, sy, sz, pt[u].x, pt[u].y, pt[u].z);
                    float mm = m[u];
                    if (d < mm) { mm = d; D[st[u] + lane] = mm; }
                    bk = make_key(mm, (unsigned)__float_as_int(pt[u].w));
                    bx = pt[u].x; by = pt[u].y; bz = pt[u].z;
                }
                for (int i = lane + 64; i < cnt[u]; i += 64) {       // cells with more than 64 points
                    const float4 q = S[st[u] + i];
                    float mm = D[st[u] + i];
                    const float d = dist2<MODE>(sx, sy, sz, q.x, q.y, q.z);
                    if (d < mm) { mm = d; D[st[u] + i] = mm; }
                    const unsigned long long kk = make_key(mm, (unsigned)__float_as_int(q.w));
                    if (kk > bk) { bk = kk; bx = q.x; by = q.y; bz = q.z; }
                }
                const unsigned long long wk = wave_umax_key(bk);
                if (bk == wk) { key[cell[u]] = wk; cx[cell[u]] = bx; cy[cell[u]] = by; cz[cell[u]] = bz; }
                if (wk > wbest) { wbest = wk; wbest_cell = cell[u]; }
            }
        }
        if (lane == 0) { pkey[par * 2 * NW + NW + wave] = wbest; pcell[par * 2 * NW + NW + wave] = wbest_cell; }
        STAMP(t3);
        __syncthreads();
        STAMP(t4);
        // phase C: arg-max over the 2*NW wave partials
        {
            const unsigned long long kk = lane < 2 * NW ? pkey[par * 2 * NW + lane] : 0ull;
            const int cc = lane < 2 * NW ? pcell[par * 2 * NW + lane] : 0;
            const unsigned long long wk = wave_umax_key(kk);
            const unsigned long long hit = __ballot(kk == wk);
            const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)hit) - 1);
            const int cell = __builtin_amdgcn_readlane(cc, src);
            last = wk ? (int)(~(unsigned)wk) : 0;
            sx = cx[cell]; sy = cy[cell]; sz = cz[cell];
        }
#ifdef PCP_FPS_STAMPS
        STAMP(t5);
        sA += t1 - t0; sW1 += t2 - t1; sB += t3 - t2; sW2 += t4 - t3; sC += t5 - t4; sQ += nq;
#endif
    }
#ifdef PCP_FPS_STAMPS
    if (b == 0 && lane == 0 && (wave == 0 || wave == NW - 1))
        printf("[stamps] wave %d M=%d C=%d g=%dx%dx%d rounds=%d prologue=%llu | per round: A=%llu bar1=%llu B=%llu bar2=%llu C=%llu cycles, queue=%.1f cells, passes/round=%.2f memwait/pass=%llu\n",
               wave, M, C, g[0], g[1], g[2], kn - 1, tp1 - tp0, sA / (kn - 1), sW1 / (kn - 1), sB / (kn - 1),
               sW2 / (kn - 1), sC / (kn - 1), (double)sQ / (kn - 1), (double)sPass / (kn - 1), sMem / (sPass ? sPass : 1));
    (void)sPts;
#endif
}

// ---- host entry points -------------------------------------------------------------------------
static int fps_pick_algo(int algo, int64_t max_pts)
{
    if (algo != 0) return algo;
    return max_pts >= 2048 ? 2 : 1;
}

// workspace: sorted records float4[total] | min_d float[total]
extern "C" size_t pcp_fps_workspace_bytes(int B, int64_t total_pts, int64_t max_pts, int K, int algo)
{
    (void)B; (void)K; (void)algo; (void)max_pts;
    if (total_pts < 0) total_pts = 0;
    return pcp_align256((size_t)total_pts * 16) + pcp_align256((size_t)total_pts * 4) + 256;
}

#define PCP_BUCKET_T 512
#define PCP_BUCKET_CPT 4

template <int MODE>
static int launch_bucket(const float* xyz, const long long* offsets, int B, int K, long long* idx,
                         float* out_xyz, float4* pts4, float* md, hipStream_t st)
{
    constexpr size_t lds = fps_bucket_lds_bytes<PCP_BUCKET_T, PCP_BUCKET_CPT>();
    auto kern = fps_bucket<PCP_BUCKET_T, PCP_BUCKET_CPT, MODE>;
    if (lds > 48 * 1024)
        PCP_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(B), dim3(PCP_BUCKET_T), lds, st, xyz, offsets, K, idx, out_xyz, pts4, md);
    return pcp_launch_status();
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
    float4* pts4 = (float4*)ws;
    float* md = (float*)((char*)ws + pcp_align256((size_t)total_pts * 16));
    const long long* off = (const long long*)offsets;
    if (fps_pick_algo(algo, max_pts) == 2)
        return dist_mode == 0 ? launch_bucket<0>(xyz, off, B, K, (long long*)idx, out_xyz, pts4, md, st)
                              : launch_bucket<1>(xyz, off, B, K, (long long*)idx, out_xyz, pts4, md, st);
    if (dist_mode == 0)
        hipLaunchKernelGGL(fps_brute<0>, dim3(B), dim3(1024), 0, st, xyz, off, K, (long long*)idx, out_xyz, md);
    else
        hipLaunchKernelGGL(fps_brute<1>, dim3(B), dim3(1024), 0, st, xyz, off, K, (long long*)idx, out_xyz, md);
    return pcp_launch_status();
}
