// Stage 1 of the DP3 path on gfx950: pinhole backprojection, colour alignment, validity
// mask, rigid transform into the robot-base frame, optional workspace crop, and an
// ORDER-PRESERVING stream compaction (row-major pixel order == numpy boolean-mask order).
//
// Replaces (reference files under data_processing_pipeline/):
//   deproject       extract_point_clouds_one_cam.py:58-64
//   align_colors    extract_point_clouds_one_cam.py:66-78
//   validity mask   extract_point_clouds_one_cam.py:109-110
//   base transform  extract_point_clouds_one_cam.py:119
//   workspace crop  dp3_point_cloud_processing.py:35-42 (when apply_crop)
//
// All coordinate math is FP64 in the reference's operation order, so the
// masks are bit-exact; the only f32 compares are the depth-range tests of XPC:109.
//
// Compaction is three launches with no inter-workgroup hand-off inside a launch:
//   k1_flags   : every pixel → predicate; one 64-bit ballot per 64 consecutive pixels and
//                one count per 1024-pixel tile                       (reads depth + rgb)
//   k1_scan_*  : per-frame tile offsets, then CSR frame offsets      (tiny)
//   k1_scatter : thread j of a tile finds the tile's j-th survivor from the ballots,
//                recomputes that pixel and writes row offsets[b]+tile_off+j — only
//                survivors cost FP64 work the second time, and rows land coalesced.
#include "pcp_internal.h"

struct PixelOut {
    double cam[3];
    double base[3];
    float col[3];
    int proj, valid, inbox, keep;
};

// (float)((double)i / 255.0) for i = 0..255 (XPC:77), folded at compile time with IEEE rounding: the 256 possible
// colour values instead of three FP64 divisions per survivor
struct K1UnitLut {
    float v[256];
    constexpr K1UnitLut() : v() { for (int i = 0; i < 256; ++i) v[i] = (float)((double)i / 255.0); }
};
__constant__ const K1UnitLut k1_unit_lut = K1UnitLut();

// rows 0..2 of a 4x4 applied to (x,y,z,1)
__device__ __forceinline__ void rigid3(const double* r, int use_fma, double x, double y, double z,
                                       double* o)
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double* q = r + 4 * c;
        double a;
        if (use_fma) {                       // k-ordered chain == OpenBLAS dgemm on FMA hosts
            a = q[0] * x;
            a = fma(q[1], y, a);
            a = fma(q[2], z, a);
            a = fma(q[3], 1.0, a);
        } else {
            a = ((q[0] * x + q[1] * y) + q[2] * z) + q[3];
        }
        o[c] = a;
    }
}

// align_colors for one camera-frame point, XPC:66-78: project into the colour image, nearest pixel
// (half-to-even), in-image + z tests, colour lookup.  Returns proj_mask.
template <bool NEED_COLOR>
__device__ __forceinline__ int align_math(const DevParams& P, const double* cam,
                                          const uint8_t* __restrict__ rgb, int Hc, int Wc, float* col)
{
    double cf[3];
    rigid3(P.T, 1, cam[0], cam[1], cam[2], cf);                   // XPC:67-68
    double zdiv = cf[2] > P.zc_clamp ? cf[2] : P.zc_clamp;        // np.maximum(z, 0.001)
    if (cf[2] != cf[2]) zdiv = cf[2];                             //   ... which propagates NaN
    const double uc = ((cf[0] * P.cfx) / zdiv) + P.ccx;           // XPC:70
    const double vc = ((cf[1] * P.cfy) / zdiv) + P.ccy;           // XPC:71
    const double ur = rint(uc), vr = rint(vc);                    // half-to-even, XPC:73
    const bool inside = (ur >= 0.0) && (ur < (double)Wc) && (vr >= 0.0) && (vr < (double)Hc);
    const int proj = inside && (cf[2] > P.zc_min);                // XPC:74-75
    col[0] = col[1] = col[2] = 0.0f;
    if (NEED_COLOR && proj && rgb) {
        const uint8_t* px = rgb + ((size_t)(long long)vr * Wc + (size_t)(long long)ur) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) col[c] = k1_unit_lut.v[px[c]];             // XPC:77
    }
    return proj;
}

template <bool NEED_COLOR>
__device__ __forceinline__ void pixel_math(const DevParams& P, float d, int u, int v,
                                           const uint8_t* __restrict__ rgb, int Hc, int Wc,
                                           PixelOut& o)
{
    const double z = (double)d;
    o.cam[0] = (((double)u - P.dcx) * z) / P.dfx;                 // XPC:62 multiply, then divide
    o.cam[1] = (((double)v - P.dcy) * z) / P.dfy;                 // XPC:63
    o.cam[2] = z;
    o.proj = align_math<NEED_COLOR>(P, o.cam, rgb, Hc, Wc, o.col);
    o.valid = o.proj && (d > P.z_min) && (d < P.z_max);           // XPC:109 (f32 compares)
    rigid3(P.E, P.transform_fma, o.cam[0], o.cam[1], o.cam[2], o.base);          // XPC:119
    const bool in_box = o.base[0] > P.lo[0] && o.base[0] < P.hi[0] &&
                        o.base[1] > P.lo[1] && o.base[1] < P.hi[1] &&
                        o.base[2] > P.lo[2] && o.base[2] < P.hi[2];              // DP3:35-39
    o.inbox = o.valid && in_box;
    o.keep = o.valid && (!P.apply_crop || in_box);
}

// align_colors on free-standing points (the reference function's own signature, XPC:66)
__global__ void __launch_bounds__(256)
k1_align_points(const double* __restrict__ pts, long long n, const uint8_t* __restrict__ rgb, int Hc,
                int Wc, DevParams P, float* __restrict__ colors, uint8_t* __restrict__ mask)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double cam[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    float col[3];
    const int proj = align_math<true>(P, cam, rgb, Hc, Wc, col);
    if (colors) { colors[3 * i] = col[0]; colors[3 * i + 1] = col[1]; colors[3 * i + 2] = col[2]; }
    if (mask) mask[i] = (uint8_t)proj;
}

// ---- dense per-pixel outputs -------------------------------------------------------------
__global__ void __launch_bounds__(256)
k1_dense(const float* __restrict__ depth, const uint8_t* __restrict__ rgb, int HW, int W, int Hc,
         int Wc, DevParams P, double* pts_cam, float* colors, uint8_t* proj_mask, uint8_t* valid,
         double* pts_base, uint8_t* crop_mask)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const size_t g = (size_t)b * HW + i;
    const uint8_t* img = rgb ? rgb + (size_t)b * Hc * Wc * 3 : nullptr;
    PixelOut o;
    int pv, pu;
    pcp_divmod(i, W, 1.0f / (float)W, &pv, &pu);
    pixel_math<true>(P, depth[g], pu, pv, img, Hc, Wc, o);
    if (pts_cam) { pts_cam[3 * g] = o.cam[0]; pts_cam[3 * g + 1] = o.cam[1]; pts_cam[3 * g + 2] = o.cam[2]; }
    if (pts_base) { pts_base[3 * g] = o.base[0]; pts_base[3 * g + 1] = o.base[1]; pts_base[3 * g + 2] = o.base[2]; }
    if (colors) { colors[3 * g] = o.col[0]; colors[3 * g + 1] = o.col[1]; colors[3 * g + 2] = o.col[2]; }
    if (proj_mask) proj_mask[g] = (uint8_t)o.proj;
    if (valid) valid[g] = (uint8_t)o.valid;
    if (crop_mask) crop_mask[g] = (uint8_t)o.inbox;
}

// ---- compaction pass 1: predicate ballots + tile counts ------------------------------------
// grid (tiles, images), image b = b_first + blockIdx.y*b_stride (one camera of every frame when
// b_stride = cameras); block 256: wave w, round r covers pixels tile*1024 + (w*4+r)*64 + lane.
//
// keep = proj_mask && depth range && (crop off || inside the box).  The colour-frame projection that
// decides proj_mask (a rigid transform, two FP64 divisions, two rints) is more than half of the FP64
// work, and with the crop on only ~20 % of the pixels still need it after the cheap tests.  So each wave
// first runs deproject + base transform + box/range tests for its 256 pixels, packs the camera-frame
// points that passed into LDS, and evaluates the projection for those only — one or two 64-lane batches
// instead of four.  When more than K1_PACK_MAX pixels pass (crop off, or a frame filled with in-box
// points) the wave falls back to evaluating every pixel in place.
#define K1_PACK_MAX 128

// CROP / TFMA: p->apply_crop / p->transform_fma as template parameters — with run-time switches the compiler kept
// both forms of every transform row behind branches and, short of scalar registers, re-fetched the parameter block from
// the kernel-argument segment inside the pixel loop (six scalar loads with a full wait per 64 pixels).
template <bool CROP, bool TFMA>
__global__ void __launch_bounds__(256)
k1_flags(const float* __restrict__ depth, const uint8_t* __restrict__ rgb, int HW, int W, int Hc,
         int Wc, DevParams P, unsigned long long* __restrict__ masks, int* __restrict__ tile_cnt,
         int tiles, int b_first, int b_stride)
{
    __shared__ int wave_cnt[4];
    __shared__ double pk_cam[4][3][K1_PACK_MAX];                  // SoA: consecutive survivors 8 bytes apart, no bank conflicts
    __shared__ int pk_pos[4][K1_PACK_MAX];                        // dword slots: byte-wide ones share banks four to a word
    __shared__ unsigned char pk_res[4][256];                      // projection verdict per original pixel slot
    const int b = b_first + blockIdx.y * b_stride, tile = blockIdx.x;
    const int lane = pcp_lane(), wave = threadIdx.x >> 6;
    const float* dframe = depth + (size_t)b * HW;
    const uint8_t* img = rgb ? rgb + (size_t)b * Hc * Wc * 3 : nullptr;
    unsigned long long* mrow = masks + ((size_t)b * tiles + tile) * PCP_TILE_WORDS;
    const float invW = 1.0f / (float)W;
    const unsigned long long lt = (1ull << lane) - 1ull;

    double cam[4][3];
    unsigned long long pre[4];
    int npre = 0;
    // (row, column) of the lane's first pixel by one division; the next rounds are 64 pixels further on
    int pv, pu;
    pcp_divmod(tile * PCP_TILE + wave * 256 + lane, W, invW, &pv, &pu);
    const int dv64 = 64 / W, du64 = 64 - dv64 * W;                    // 64 = dv64*W + du64 (scalar)
    // Phase by phase over the wave's four rounds instead of pixel by pixel: each phase needs only a few of the ~40
    // FP64 parameters, so they stay in scalar registers instead of being re-fetched for every round.
    bool ok[4];
    float dep[4];
    int ppu[4], ppv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                     // deproject, XPC:58-64
        const int i = tile * PCP_TILE + (wave * 4 + r) * 64 + lane;
        if (r > 0) {
            pu += du64; pv += dv64;
            if (pu >= W) { pu -= W; pv += 1; }
        }
        const float d = i < HW ? dframe[i] : 0.0f;                    // 0 fails the range test below
        const double z = (double)d;
        dep[r] = d; ppu[r] = pu; ppv[r] = pv;
        // only pixels that pass the range test below are ever looked at, and for those the numerators are tame:
        // the exact quotient without its range check (pcp_quot_valid)
        cam[r][0] = pcp_quot_valid(((double)pu - P.dcx) * z, P.dfx, P.rdfx);    // XPC:62, exact
        cam[r][1] = pcp_quot_valid(((double)pv - P.dcy) * z, P.dfy, P.rdfy);    // XPC:63
        cam[r][2] = z;
        ok[r] = (i < HW) & (d > P.z_min) & (d < P.z_max);             // XPC:109 (f32 compares)
    }
    if (P.rdfx == 0.0) {                                              // launch-uniform: odd intrinsics, plain divisions
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double z = (double)dep[r];
            cam[r][0] = (((double)ppu[r] - P.dcx) * z) / P.dfx;
            cam[r][1] = (((double)ppv[r] - P.dcy) * z) / P.dfy;
        }
    }
    if (CROP) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {                                 // one row of XPC:119 and its two DP3:35-39 compares
            const double* q = P.E + 4 * c;
            const double lo = P.lo[c], hi = P.hi[c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double a;
                if (TFMA) {
                    a = q[0] * cam[r][0];
                    a = fma(q[1], cam[r][1], a);
                    a = fma(q[2], cam[r][2], a);
                    a = fma(q[3], 1.0, a);
                } else {
                    a = ((q[0] * cam[r][0] + q[1] * cam[r][1]) + q[2] * cam[r][2]) + q[3];
                }
                ok[r] = ok[r] & (a > lo) & (a < hi);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pre[r] = __ballot(ok[r]);
        npre += __popcll(pre[r]);
    }

    unsigned long long keep[4];
    if (npre <= K1_PACK_MAX) {                                        // wave-uniform
        int off = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if ((pre[r] >> lane) & 1ull) {
                const int c = off + __popcll(pre[r] & lt);
                pk_cam[wave][0][c] = cam[r][0]; pk_cam[wave][1][c] = cam[r][1]; pk_cam[wave][2][c] = cam[r][2];
                pk_pos[wave][c] = r * 64 + lane;
            }
            off += __popcll(pre[r]);
        }
        for (int c = lane; c < npre; c += 64) {                       // one wave: LDS accesses stay in program order
            const double cc[3] = {pk_cam[wave][0][c], pk_cam[wave][1][c], pk_cam[wave][2][c]};
            float col[3];
            // the verdict goes back to the pixel's own slot with a plain byte store (an atomicOr into four mask words
            // serialised the whole wave on one address: 34 % of this kernel's LDS cycles were bank conflicts)
            pk_res[wave][pk_pos[wave][c]] = (unsigned char)align_math<false>(P, cc, img, Hc, Wc, col);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)                                   // slots that did not pass the pre-test were never written: masked by pre
            keep[r] = __ballot(((pre[r] >> lane) & 1ull) && pk_res[wave][r * 64 + lane]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int k = 0;
            if ((pre[r] >> lane) & 1ull) {
                float col[3];
                k = align_math<false>(P, cam[r], img, Hc, Wc, col);
            }
            keep[r] = __ballot(k);
        }
    }
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (lane == 0) mrow[wave * 4 + r] = keep[r];
        cnt += __popcll(keep[r]);
    }
    if (lane == 0) wave_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0)
        tile_cnt[(size_t)b * tiles + tile] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// ---- scan kernels (shared with crop.hip) -----------------------------------------------------
// one block per frame: exclusive prefix of the frame's tile counts (in place) + frame total
__global__ void __launch_bounds__(256)
pcp_scan_tiles(int* __restrict__ tile_cnt, int tiles, long long* __restrict__ frame_cnt)
{
    __shared__ int lds[17];
    int* row = tile_cnt + (size_t)blockIdx.x * tiles;
    int carry = 0;
    for (int t0 = 0; t0 < tiles; t0 += 256) {
        const int t = t0 + threadIdx.x;
        const int v = t < tiles ? row[t] : 0;
        int total;
        const int ex = pcp_block_excl_scan(v, lds, &total);
        if (t < tiles) row[t] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) frame_cnt[blockIdx.x] = carry;
}

// single block: offsets[0..B] = exclusive prefix of frame_cnt (int64)
__global__ void __launch_bounds__(256)
pcp_scan_frames(const long long* __restrict__ frame_cnt, int B, long long* __restrict__ offsets)
{
    __shared__ long long wsum[4];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = pcp_lane(), wave = threadIdx.x >> 6;
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const long long v = b < B ? frame_cnt[b] : 0;
        long long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            long long t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        long long woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const long long carry = carry_s;
        if (b < B) offsets[b] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[B] = carry_s;
}

// CSR offsets of a compacted batch (shared with crop.hip, voxel.hip, sor.hip): rows [offsets[b], offsets[b+1])
// of the input belong to cloud b; out_offsets[b] = number of kept rows (set ballot bits) before row offsets[b]
__global__ void pcp_csr_offsets(const long long* __restrict__ offsets, int B,
                            const unsigned long long* __restrict__ masks, const int* __restrict__ tile_off,
                            const long long* __restrict__ total_heads, long long* __restrict__ out_offsets)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > B) return;
    const long long p = offsets[b], n = offsets[B];
    if (p >= n) { out_offsets[b] = total_heads[0]; return; }
    const long long tile = p / PCP_TILE;
    const int in_tile = (int)(p % PCP_TILE), word = in_tile >> 6, bit = in_tile & 63;
    long long r = tile_off[tile];
    for (int w = 0; w < word; ++w) r += __popcll(masks[(size_t)tile * PCP_TILE_WORDS + w]);
    const unsigned long long low = bit ? (masks[(size_t)tile * PCP_TILE_WORDS + word] & ((1ull << bit) - 1ull)) : 0ull;
    out_offsets[b] = r + __popcll(low);
}


// ---- compaction pass 2: survivors only -----------------------------------------------------
// TFMA: p->transform_fma as a template parameter (see k1_flags).  Every output is optional and only what is asked for
// is computed: without rgb32 the colour-frame projection (more than half of the FP64 work and a scattered byte
// gather) is skipped altogether — the pipeline then fetches the colours of the K sampled points only
// (k1_gather_pixels below).
#define K1_SC_TILES 8          // tiles per workgroup: one barrier and eight independent depth loads per thread instead of
                               // two barriers and one dependent load per tile (the pass is latency-, not bandwidth-bound;
                               // 2 / 4 / 8 / 16 tiles: 161 / 148 / 142 / 149 us per 300 frames, one tile: 208)
template <bool TFMA>
__global__ void __launch_bounds__(256)
k1_scatter(const float* __restrict__ depth, const uint8_t* __restrict__ rgb, int HW, int W, int Hc,
           int Wc, DevParams P, const unsigned long long* __restrict__ masks,
           const int* __restrict__ tile_off, const long long* __restrict__ offsets, int tiles,
           double* xyz_cam64, double* xyz64, float* xyz32, float* rgb32, int32_t* pix,
           long long cap, int* overflow, int b_first, int b_stride)
{
    __shared__ unsigned long long m[K1_SC_TILES][PCP_TILE_WORDS];
    __shared__ int pre[K1_SC_TILES][PCP_TILE_WORDS + 1];
    const int b = b_first + blockIdx.y * b_stride, tile0 = blockIdx.x * K1_SC_TILES;
    if (threadIdx.x < K1_SC_TILES * PCP_TILE_WORDS) {
        // thread (t, w) fetches ballot word w of tile t; the sixteen lanes of a tile scan their popcounts among themselves
        const int t = threadIdx.x / PCP_TILE_WORDS, w = threadIdx.x % PCP_TILE_WORDS;
        const unsigned long long v = tile0 + t < tiles ? masks[((size_t)b * tiles + tile0 + t) * PCP_TILE_WORDS + w] : 0ull;
        m[t][w] = v;
        int incl = __popcll(v);
#pragma unroll
        for (int d = 1; d < PCP_TILE_WORDS; d <<= 1) {
            const int x = __shfl_up(incl, d, PCP_TILE_WORDS);
            if (w >= d) incl += x;
        }
        pre[t][w + 1] = incl;
        if (w == 0) pre[t][0] = 0;
    }
    __syncthreads();
    int n[K1_SC_TILES], nmax = 0;
    long long base[K1_SC_TILES];
#pragma unroll
    for (int t = 0; t < K1_SC_TILES; ++t) {
        n[t] = pre[t][PCP_TILE_WORDS];
        nmax = n[t] > nmax ? n[t] : nmax;
        base[t] = n[t] ? offsets[b] + tile_off[(size_t)b * tiles + tile0 + t] : 0;
    }
    if (nmax == 0) return;
    const float* dframe = depth + (size_t)b * HW;
    const uint8_t* img = rgb ? rgb + (size_t)b * Hc * Wc * 3 : nullptr;
    const float invW = 1.0f / (float)W;
    for (int j = threadIdx.x; j < nmax; j += blockDim.x) {
        int pi[K1_SC_TILES];
        float dep[K1_SC_TILES];
#pragma unroll
        for (int t = 0; t < K1_SC_TILES; ++t) {                       // the four loads go out together
            pi[t] = -1;
            dep[t] = 0.f;
            if (j < n[t]) {
                int w = 0;
#pragma unroll
                for (int s = 8; s >= 1; s >>= 1)
                    if (pre[t][w + s] <= j) w += s;                   // largest w with pre[w] <= j
                const int bit = pcp_select_bit(m[t][w], j - pre[t][w]);
                pi[t] = (tile0 + t) * PCP_TILE + w * 64 + bit;
                dep[t] = dframe[pi[t]];
            }
        }
#pragma unroll
        for (int t = 0; t < K1_SC_TILES; ++t) {
            const int i = pi[t];
            const long long row = base[t] + j;
            const bool live = i >= 0 && row < cap;
            if (i >= 0 && row >= cap && overflow) *overflow = 1;
            int pv, pu;
            pcp_divmod(i < 0 ? 0 : i, W, invW, &pv, &pu);
            const double z = (double)dep[t];
            double cam[3];
            bool bad = false;
            cam[0] = pcp_quot(((double)pu - P.dcx) * z, P.dfx, P.rdfx, bad);    // XPC:62 multiply, then divide (exact)
            cam[1] = pcp_quot(((double)pv - P.dcy) * z, P.dfy, P.rdfy, bad);    // XPC:63
            cam[2] = z;
            if (__any(bad && live) || P.rdfx == 0.0) {                // practically never: extreme values, odd fx
                cam[0] = (((double)pu - P.dcx) * z) / P.dfx;
                cam[1] = (((double)pv - P.dcy) * z) / P.dfy;
            }
            if (!live) continue;
            if (xyz_cam64) { xyz_cam64[3 * row] = cam[0]; xyz_cam64[3 * row + 1] = cam[1]; xyz_cam64[3 * row + 2] = cam[2]; }
            if (xyz64 || xyz32) {
                double g[3];
                rigid3(P.E, TFMA, cam[0], cam[1], cam[2], g);         // XPC:119
                if (xyz64) { xyz64[3 * row] = g[0]; xyz64[3 * row + 1] = g[1]; xyz64[3 * row + 2] = g[2]; }
                if (xyz32) { xyz32[3 * row] = (float)g[0]; xyz32[3 * row + 1] = (float)g[1]; xyz32[3 * row + 2] = (float)g[2]; }   // DP3:10
            }
            if (rgb32) {
                float col[3];
                align_math<true>(P, cam, img, Hc, Wc, col);           // XPC:66-78 (the mask already holds proj)
                rgb32[3 * row] = col[0]; rgb32[3 * row + 1] = col[1]; rgb32[3 * row + 2] = col[2];
            }
            if (pix) pix[row] = i;
        }
    }
}

// Colours of the SAMPLED points only: pc_rgb[idx] (DP3:48) evaluated from each selected point's source pixel —
// deproject + align_colors (XPC:58-78) for K points per cloud instead of storing (and first computing) a colour for
// every one of the cloud's M points.  Same arithmetic, same values as gathering rgb32.
__global__ void __launch_bounds__(256)
k1_gather_pixels(const float* __restrict__ depth, const uint8_t* __restrict__ rgb, int HW, int W, int Hc, int Wc,
                 DevParams P, const int32_t* __restrict__ pix, const long long* __restrict__ offsets,
                 long long total_pts, const long long* __restrict__ idx, int K, int compat_neg,
                 float* __restrict__ out)
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
    if (i >= M || base < 0 || base + i >= total_pts) { o[0] = o[1] = o[2] = 0.f; return; }
    const int p = pix[base + i];
    if (p < 0 || p >= HW) { o[0] = o[1] = o[2] = 0.f; return; }
    int pv, pu;
    pcp_divmod(p, W, 1.0f / (float)W, &pv, &pu);
    const double z = (double)depth[(size_t)b * HW + p];
    const double cam[3] = {(((double)pu - P.dcx) * z) / P.dfx, (((double)pv - P.dcy) * z) / P.dfy, z};
    float col[3];
    align_math<true>(P, cam, rgb + (size_t)b * Hc * Wc * 3, Hc, Wc, col);
    o[0] = col[0]; o[1] = col[1]; o[2] = col[2];
}

// The same walk, but the result is WHERE the colour sits, not the colour: the colour-image pixel v_idx * Wc + u_idx of
// XPC:73-77 for every sampled point, -1 where align_colors masks the point (its colour stays zero, XPC:76) or the
// row does not exist.  A caller that keeps the rgb images in host memory then needs no rgb upload at all: the
// projection mask depends on rgb.shape only (XPC:72-75) and colours are wanted for the K sampled points.
__global__ void __launch_bounds__(256)
k1_sampled_color_pixels(const float* __restrict__ depth, int HW, int W, int Hc, int Wc, DevParams P,
                        const int32_t* __restrict__ pix, const long long* __restrict__ offsets, long long total_pts,
                        const long long* __restrict__ idx, int K, int compat_neg, int32_t* __restrict__ out)
{
    const int b = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const long long base = offsets[b], M = offsets[b + 1] - base;
    long long i = idx[(size_t)b * K + k];
    int32_t* o = out + (size_t)b * K + k;
    if (i < 0) {
        if (compat_neg && M > 0 && i + M >= 0) i += M;        // numpy negative indexing, DP3:48
        else { *o = -1; return; }
    }
    if (i >= M || base < 0 || base + i >= total_pts) { *o = -1; return; }
    const int p = pix[base + i];
    if (p < 0 || p >= HW) { *o = -1; return; }
    int pv, pu;
    pcp_divmod(p, W, 1.0f / (float)W, &pv, &pu);
    const double z = (double)depth[(size_t)b * HW + p];
    const double cam[3] = {(((double)pu - P.dcx) * z) / P.dfx, (((double)pv - P.dcy) * z) / P.dfy, z};
    double cf[3];
    rigid3(P.T, 1, cam[0], cam[1], cam[2], cf);                   // XPC:67-68, as align_math
    double zdiv = cf[2] > P.zc_clamp ? cf[2] : P.zc_clamp;
    if (cf[2] != cf[2]) zdiv = cf[2];
    const double uc = ((cf[0] * P.cfx) / zdiv) + P.ccx;
    const double vc = ((cf[1] * P.cfy) / zdiv) + P.ccy;
    const double ur = rint(uc), vr = rint(vc);
    const bool inside = (ur >= 0.0) && (ur < (double)Wc) && (vr >= 0.0) && (vr < (double)Hc);
    *o = (inside && (cf[2] > P.zc_min)) ? (int32_t)((long long)vr * Wc + (long long)ur) : -1;
}

// fused (all cameras of a frame) CSR offsets: every C-th entry of the per-image offsets
__global__ void k1_fused_offsets(const long long* __restrict__ offsets, int F, int C, long long* __restrict__ fused)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f <= F) fused[f] = offsets[(size_t)f * C];
}

// ---- host entry points -----------------------------------------------------------------------
static bool params_ok(const pcp_params* p)
{
    return p && p->depth.fx != 0.0 && p->depth.fy != 0.0;
}

extern "C" int pcp_align_colors(const double* pts, int64_t n, const uint8_t* rgb, int Hc, int Wc,
                                const pcp_params* p, float* colors, uint8_t* mask, void* stream)
{
    if (n < 0 || !p || Hc <= 0 || Wc <= 0 || (n > 0 && (!pts || !rgb))) return PCP_ERR_INVALID_ARG;
    if (n == 0) return PCP_OK;
    hipLaunchKernelGGL(k1_align_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       pts, (long long)n, rgb, Hc, Wc, pcp_to_dev(p), colors, mask);
    return pcp_launch_status();
}

extern "C" int pcp_backproject_dense(const float* depth, const uint8_t* rgb, int B, int H, int W,
                                     int Hc, int Wc, const pcp_params* p, double* pts_cam,
                                     float* colors, uint8_t* proj_mask, uint8_t* valid,
                                     double* pts_base, uint8_t* crop_mask, void* stream)
{
    if (!depth || !params_ok(p) || B < 0 || H <= 0 || W <= 0 || Hc <= 0 || Wc <= 0)
        return PCP_ERR_INVALID_ARG;
    if ((long long)H * W > 0x7fffffffLL / 4 || H >= (1 << 20)) return PCP_ERR_INVALID_ARG;
    if (B == 0) return PCP_OK;
    const int HW = H * W;
    dim3 grid((HW + 255) / 256, B);
    hipLaunchKernelGGL(k1_dense, grid, dim3(256), 0, (hipStream_t)stream, depth, rgb, HW, W, Hc, Wc,
                       pcp_to_dev(p), pts_cam, colors, proj_mask, valid, pts_base, crop_mask);
    return pcp_launch_status();
}

// workspace layout: masks u64[B*tiles*16] | tile_cnt i32[B*tiles] | frame_cnt i64[B]
extern "C" size_t pcp_backproject_workspace_bytes(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 256;
    const size_t tiles = ((size_t)H * W + PCP_TILE - 1) / PCP_TILE;
    return pcp_align256((size_t)B * tiles * PCP_TILE_WORDS * 8) + pcp_align256((size_t)B * tiles * 4) +
           pcp_align256((size_t)B * 8) + 256;
}

// One implementation for the single-camera and the fused multi-camera entry points: images are
// laid out (frame, camera) and camera c uses cams[c]; the flag and scatter passes are launched once
// per camera over a strided image index, the scans once over all F*C images.
static int backproject_align_crop_impl(const float* depth, const uint8_t* rgb, int F, int C, int H, int W,
                                       int Hc, int Wc, const pcp_params* cams, double* xyz_cam64,
                                       double* xyz64, float* xyz32, float* rgb32, int32_t* pix,
                                       int64_t* offsets, int64_t* fused_offsets, int64_t cap,
                                       int32_t* overflow, void* ws, size_t ws_bytes, void* stream)
{
    if (!depth || !cams || !offsets || F < 0 || C <= 0 || H <= 0 || W <= 0 || Hc <= 0 || Wc <= 0 || cap < 0)
        return PCP_ERR_INVALID_ARG;
    for (int c = 0; c < C; ++c)
        if (!params_ok(cams + c)) return PCP_ERR_INVALID_ARG;
    if (rgb32 && !rgb) return PCP_ERR_INVALID_ARG;
    if ((long long)H * W > 0x7fffffffLL / 4 || H >= (1 << 20) || (long long)F * C > 0x7fffffffLL) return PCP_ERR_INVALID_ARG;
    const int B = F * C;
    if (!ws || ((uintptr_t)ws & 255) || ws_bytes < pcp_backproject_workspace_bytes(B, H, W))
        return PCP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W;
    const int tiles = (HW + PCP_TILE - 1) / PCP_TILE;
    char* w = (char*)ws;
    unsigned long long* masks = (unsigned long long*)w;
    w += pcp_align256((size_t)B * tiles * PCP_TILE_WORDS * 8);
    int* tile_cnt = (int*)w;
    w += pcp_align256((size_t)B * tiles * 4);
    long long* frame_cnt = (long long*)w;
    if (overflow) PCP_FILL_TRY(overflow, 0u, 1, st);
    if (B == 0) {
        PCP_FILL_TRY(offsets, 0u, 2, st);
        if (fused_offsets) PCP_FILL_TRY(fused_offsets, 0u, 2, st);
        return PCP_OK;
    }
    dim3 grid(tiles, F);
    for (int c = 0; c < C; ++c) {
        const bool crop = cams[c].apply_crop != 0, tfma = cams[c].transform_fma != 0;
        auto kern = crop ? (tfma ? k1_flags<true, true> : k1_flags<true, false>)
                         : (tfma ? k1_flags<false, true> : k1_flags<false, false>);
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, depth, rgb, HW, W, Hc, Wc, pcp_to_dev(cams + c),
                           masks, tile_cnt, tiles, c, C);
    }
    hipLaunchKernelGGL(pcp_scan_tiles, dim3(B), dim3(256), 0, st, tile_cnt, tiles, frame_cnt);
    hipLaunchKernelGGL(pcp_scan_frames, dim3(1), dim3(256), 0, st, frame_cnt, B, (long long*)offsets);
    const dim3 sgrid((tiles + K1_SC_TILES - 1) / K1_SC_TILES, F);
    for (int c = 0; c < C; ++c)
        hipLaunchKernelGGL(cams[c].transform_fma ? k1_scatter<true> : k1_scatter<false>, sgrid, dim3(256), 0, st, depth, rgb,
                           HW, W, Hc, Wc, pcp_to_dev(cams + c), masks, tile_cnt, (const long long*)offsets, tiles,
                           xyz_cam64, xyz64, xyz32, rgb32, pix, (long long)cap, overflow, c, C);
    if (fused_offsets)
        hipLaunchKernelGGL(k1_fused_offsets, dim3((F + 256) / 256), dim3(256), 0, st, (const long long*)offsets,
                           F, C, (long long*)fused_offsets);
    return pcp_launch_status();
}

extern "C" int pcp_gather_rgb_pixels(const float* depth, const uint8_t* rgb, int B, int H, int W, int Hc, int Wc,
                                     const pcp_params* p, const int32_t* pix, const int64_t* offsets,
                                     int64_t total_pts, const int64_t* idx, int K, int compat_neg_index,
                                     float* out, void* stream)
{
    if (B < 0 || K < 0 || total_pts < 0 || H <= 0 || W <= 0 || Hc <= 0 || Wc <= 0 || !params_ok(p))
        return PCP_ERR_INVALID_ARG;
    if ((long long)H * W > 0x7fffffffLL / 4 || H >= (1 << 20)) return PCP_ERR_INVALID_ARG;
    if (B == 0 || K == 0) return PCP_OK;
    if (!depth || !rgb || !offsets || !idx || !out || (total_pts > 0 && !pix)) return PCP_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k1_gather_pixels, dim3((K + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, depth, rgb,
                       H * W, W, Hc, Wc, pcp_to_dev(p), pix, (const long long*)offsets, (long long)total_pts,
                       (const long long*)idx, K, compat_neg_index, out);
    return pcp_launch_status();
}

extern "C" int pcp_sampled_color_pixels(const float* depth, int B, int H, int W, int Hc, int Wc, const pcp_params* p,
                                        const int32_t* pix, const int64_t* offsets, int64_t total_pts,
                                        const int64_t* idx, int K, int compat_neg_index, int32_t* cpix, void* stream)
{
    if (B < 0 || K < 0 || total_pts < 0 || H <= 0 || W <= 0 || Hc <= 0 || Wc <= 0 || !params_ok(p))
        return PCP_ERR_INVALID_ARG;
    if ((long long)H * W > 0x7fffffffLL / 4 || H >= (1 << 20) || (long long)Hc * Wc > 0x7fffffffLL) return PCP_ERR_INVALID_ARG;
    if (B == 0 || K == 0) return PCP_OK;
    if (!depth || !offsets || !idx || !cpix || (total_pts > 0 && !pix)) return PCP_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k1_sampled_color_pixels, dim3((K + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, depth,
                       H * W, W, Hc, Wc, pcp_to_dev(p), pix, (const long long*)offsets, (long long)total_pts,
                       (const long long*)idx, K, compat_neg_index, cpix);
    return pcp_launch_status();
}

extern "C" int pcp_backproject_align_crop(const float* depth, const uint8_t* rgb, int B, int H, int W,
                                          int Hc, int Wc, const pcp_params* p, double* xyz_cam64,
                                          double* xyz64, float* xyz32, float* rgb32, int32_t* pix,
                                          int64_t* offsets, int64_t cap, int32_t* overflow, void* ws,
                                          size_t ws_bytes, void* stream)
{
    return backproject_align_crop_impl(depth, rgb, B, 1, H, W, Hc, Wc, p, xyz_cam64, xyz64, xyz32, rgb32, pix,
                                       offsets, nullptr, cap, overflow, ws, ws_bytes, stream);
}

extern "C" int pcp_backproject_align_crop_multicam(const float* depth, const uint8_t* rgb, int F, int C,
                                                   int H, int W, int Hc, int Wc, const pcp_params* cams,
                                                   double* xyz_cam64, double* xyz64, float* xyz32,
                                                   float* rgb32, int32_t* pix, int64_t* offsets,
                                                   int64_t* fused_offsets, int64_t cap, int32_t* overflow,
                                                   void* ws, size_t ws_bytes, void* stream)
{
    if (C > PCP_MAX_CAMERAS) return PCP_ERR_INVALID_ARG;
    return backproject_align_crop_impl(depth, rgb, F, C, H, W, Hc, Wc, cams, xyz_cam64, xyz64, xyz32, rgb32,
                                       pix, offsets, fused_offsets, cap, overflow, ws, ws_bytes, stream);
}
