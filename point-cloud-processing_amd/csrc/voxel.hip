// Voxel-grid downsample of B CSR clouds on gfx950 (BASELINE config 3).
//
// The reference has no voxel filter of its own (its filter_point_clouds step is absent from the
// tree), so the semantics are Open3D's PointCloud::VoxelDownSample, which the
// reference's stack would call: origin = per-cloud min bound − voxel/2, voxel index =
// floor((p − origin)/voxel) per axis, every occupied voxel yields the mean of its points and
// colours, FP64 sums accumulated in ascending point order, then one true division by the count.
// Open3D emits voxels in std::unordered_map order (unspecified); this build defines the order as
// ascending (ix, iy, iz).  PARITY UNPINNED against Open3D (absent here); bit-exact against the CPU restatement the tests hold.
//
// Pipeline (all on the caller's stream):
//   vox_minbound : per-cloud FP64 min via wave reduction + one ordered-u64 atomicMin per wave
//   vox_keys     : key = cloud<<48 | ix<<32 | iy<<16 | iz, value = point row
//   rocprim::radix_sort_pairs (stable → rows ascending inside a voxel)
//   vox_heads    : ballot of "first row of a voxel" per 64 sorted rows + counts per 1024-row tile
//   pcp_scan_tiles (one sequence)
//   vox_reduce   : thread t of a tile owns the tile's t-th voxel: walks its run, sums in order, divides
//   pcp_csr_offsets  : CSR offsets of the output clouds from the ballots
#include "pcp_internal.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#define VOX_AXIS_MAX 65535
#define VOX_KEY_PAD 0xFFFFFFFFFFFFFFFFull

// order-preserving map double <-> u64 (so unsigned atomicMin is a floating-point min)
__device__ __forceinline__ unsigned long long vox_ord(double d)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double vox_unord(unsigned long long u)
{
    u = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
    return __longlong_as_double((long long)u);
}

// cloud that owns row i: largest b with offsets[b] <= i
__device__ __forceinline__ int vox_cloud_of(const long long* __restrict__ offsets, int B, long long i)
{
    int lo = 0, hi = B;                       // invariant: offsets[lo] <= i < offsets[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// Each block walks VOX_MB_ITERS*256 consecutive rows; a wave keeps a running min while the rows stay in one
// cloud and flushes it with three atomics only when the cloud changes or the walk ends.
#define VOX_MB_ITERS 16
__global__ void __launch_bounds__(256)
vox_minbound(const double* __restrict__ xyz, const long long* __restrict__ offsets, int B,
             unsigned long long* __restrict__ minb)
{
    const long long n = offsets[B];
    const int lane = pcp_lane();
    const long long first = (long long)blockIdx.x * (256 * VOX_MB_ITERS) + (threadIdx.x >> 6) * (64 * VOX_MB_ITERS);
    if (first >= n) return;
    int cur = -1;                                  // wave-uniform cloud of the running min
    double m[3] = {0.0, 0.0, 0.0};
    auto flush = [&]() {
        if (cur < 0) return;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = m[c];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const double o = __shfl_xor(v, d, 64);
                v = o < v ? o : v;
            }
            if (lane == 0) atomicMin(&minb[3 * cur + c], vox_ord(v));
        }
    };
    for (int it = 0; it < VOX_MB_ITERS; ++it) {
        const long long i0 = first + (long long)it * 64;
        if (i0 >= n) break;                        // wave-uniform
        const long long i = i0 + lane < n ? i0 + lane : n - 1;            // clamped duplicate: harmless for a min
        const int b = vox_cloud_of(offsets, B, i);
        const double p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        const int b0 = __builtin_amdgcn_readfirstlane(b);
        if (__all(b == b0)) {
            if (b0 != cur) {
                flush();
                cur = b0;
#pragma unroll
                for (int c = 0; c < 3; ++c) m[c] = p[c];
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) m[c] = p[c] < m[c] ? p[c] : m[c];
            }
        } else {                                   // a cloud boundary inside these 64 rows: per-lane atomics
            flush();
            cur = -1;
#pragma unroll
            for (int c = 0; c < 3; ++c) atomicMin(&minb[3 * b + c], vox_ord(p[c]));
        }
    }
    flush();
}

// bf = 0: 64-bit keys cloud<<48 | ix<<32 | iy<<16 | iz.  bf > 0 (the caller promised a bounding-box edge, so every
// grid index fits bf bits): 32-bit keys cloud<<3bf | ix<<2bf | iy<<bf | iz into keys32 — a third of the sort's bytes and
// half its passes; an index that does not fit breaks the promise and raises the overflow flag.
__global__ void __launch_bounds__(256)
vox_keys(const double* __restrict__ xyz, const long long* __restrict__ offsets, int B, long long cap,
         const unsigned long long* __restrict__ minb, double voxel, unsigned long long* __restrict__ keys,
         unsigned int* __restrict__ keys32, int bf, unsigned int* __restrict__ vals, int* __restrict__ overflow)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const long long n = offsets[B];
    vals[i] = (unsigned int)i;
    if (i >= n) {
        if (bf) keys32[i] = 0xFFFFFFFFu; else keys[i] = VOX_KEY_PAD;
        return;
    }
    const int b = vox_cloud_of(offsets, B, i);
    const long long qmax = bf ? (1ll << bf) - 1 : (long long)VOX_AXIS_MAX;
    long long q[3];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double origin = vox_unord(minb[3 * b + c]) - voxel * 0.5;     // GetMinBound() - voxel_size*0.5
        const double r = floor((xyz[3 * i + c] - origin) / voxel);
        q[c] = (long long)r;
        if (!(r >= 0.0) || r > (double)qmax) { bad = true; q[c] = r > 0.0 ? qmax : 0; }
    }
    if (bad && overflow) *overflow = 1;
    if (bf) keys32[i] = (unsigned int)((((((unsigned long long)b << bf) | (unsigned long long)q[0]) << bf) | (unsigned long long)q[1]) << bf | (unsigned long long)q[2]);
    else keys[i] = ((unsigned long long)b << 48) | ((unsigned long long)q[0] << 32) | ((unsigned long long)q[1] << 16) | (unsigned long long)q[2];
}

// sorted 32-bit keys back into the 64-bit layout every later kernel reads
__global__ void __launch_bounds__(256)
vox_unpack_keys(const unsigned int* __restrict__ keys32, const long long* __restrict__ offsets, int B, long long cap,
                int bf, unsigned long long* __restrict__ keys)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    if (i >= offsets[B]) { keys[i] = VOX_KEY_PAD; return; }
    const unsigned int k = keys32[i], m = (1u << bf) - 1u;
    keys[i] = ((unsigned long long)(k >> (3 * bf)) << 48) | ((unsigned long long)((k >> (2 * bf)) & m) << 32) |
              ((unsigned long long)((k >> bf) & m) << 16) | (unsigned long long)(k & m);
}

// grid: tiles of 1024 sorted rows; block 256 (wave w, round r → word w*4+r)
__global__ void __launch_bounds__(256)
vox_heads(const unsigned long long* __restrict__ keys, const long long* __restrict__ offsets, int B,
          unsigned long long* __restrict__ masks, int* __restrict__ tile_cnt)
{
    __shared__ int wave_cnt[4];
    const long long n = offsets[B];
    const int lane = pcp_lane(), wave = threadIdx.x >> 6;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int word = wave * 4 + r;
        const long long j = (long long)blockIdx.x * PCP_TILE + word * 64 + lane;
        int head = 0;
        if (j < n) head = (j == 0) || (keys[j] != keys[j - 1]);
        const unsigned long long m = __ballot(head);
        if (lane == 0) masks[(size_t)blockIdx.x * PCP_TILE_WORDS + word] = m;
        cnt += __popcll(m);
    }
    if (lane == 0) wave_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ void __launch_bounds__(256)
vox_reduce(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ vals,
           const double* __restrict__ xyz, const float* __restrict__ rgb,
           const long long* __restrict__ offsets, int B, const unsigned long long* __restrict__ masks,
           const int* __restrict__ tile_off, double* out64, float* out32, float* out_rgb, int32_t* out_cnt,
           long long cap_out, int* overflow)
{
    __shared__ unsigned long long m[PCP_TILE_WORDS];
    __shared__ int pre[PCP_TILE_WORDS + 1];
    const int tile = blockIdx.x;
    if (threadIdx.x < PCP_TILE_WORDS) m[threadIdx.x] = masks[(size_t)tile * PCP_TILE_WORDS + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int w = 0; w < PCP_TILE_WORDS; ++w) { pre[w] = acc; acc += __popcll(m[w]); }
        pre[PCP_TILE_WORDS] = acc;
    }
    __syncthreads();
    const int nh = pre[PCP_TILE_WORDS];
    const long long n = offsets[B];
    for (int t = threadIdx.x; t < nh; t += blockDim.x) {
        int w = 0;
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1)
            if (pre[w + s] <= t) w += s;
        const int bit = pcp_select_bit(m[w], t - pre[w]);
        const long long j0 = (long long)tile * PCP_TILE + w * 64 + bit;
        const long long row = (long long)tile_off[tile] + t;
        if (row >= cap_out) { if (overflow) *overflow = 1; continue; }
        const unsigned long long key = keys[j0];
        double s[3] = {0.0, 0.0, 0.0}, c[3] = {0.0, 0.0, 0.0};
        int cnt = 0;
        for (long long j = j0; j < n && keys[j] == key; ++j) {      // rows ascending (stable sort)
            const size_t i = vals[j];
            s[0] += xyz[3 * i]; s[1] += xyz[3 * i + 1]; s[2] += xyz[3 * i + 2];
            if (rgb) { c[0] += (double)rgb[3 * i]; c[1] += (double)rgb[3 * i + 1]; c[2] += (double)rgb[3 * i + 2]; }
            ++cnt;
        }
        const double dn = (double)cnt;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a = s[k] / dn;
            if (out64) out64[3 * row + k] = a;
            if (out32) out32[3 * row + k] = (float)a;
            if (out_rgb) out_rgb[3 * row + k] = (float)(c[k] / dn);
        }
        if (out_cnt) out_cnt[row] = cnt;
    }
}

static unsigned int vox_cloud_bits(int B)
{
    unsigned int bits = 1;
    while (bits < 16 && (1 << bits) < B) ++bits;
    return bits;
}

// Bits per axis of the packed 32-bit keys, or 0 for the 64-bit keys: with the caller's promise `max_extent` a grid index
// is at most floor(max_extent / cell + 1/2) + 1 (origin = min bound - cell/2; one more for the rounding of the quotient).
static int vox_pack_bits(int B, double max_extent, double cell)
{
    if (!(max_extent > 0.0) || !(max_extent < 1e300)) return 0;
    const double cells = floor(max_extent / cell + 0.5) + 2.0;       // indices 0 .. cells-1
    if (!(cells < 65536.0)) return 0;
    int bf = 1;
    while ((double)(1 << bf) < cells) ++bf;
    return (int)vox_cloud_bits(B) + 3 * bf <= 32 ? bf : 0;
}

static size_t vox_sort_temp_bytes(int64_t cap)
{
    size_t bytes = 0, bytes32 = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long*)nullptr,
                                             (unsigned long long*)nullptr, (const unsigned int*)nullptr,
                                             (unsigned int*)nullptr, (size_t)cap, 0u, 64u, (hipStream_t)0);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    e = rocprim::radix_sort_pairs(nullptr, bytes32, (const unsigned int*)nullptr, (unsigned int*)nullptr,
                                  (const unsigned int*)nullptr, (unsigned int*)nullptr, (size_t)cap, 0u, 32u,
                                  (hipStream_t)0);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return bytes > bytes32 ? bytes : bytes32;
}

// Keys of every row (pads included), stable sort by key, 64-bit sorted keys in keys_out, source rows in vals_out.
// keys_in doubles as the two 32-bit key arrays of the packed path.
static int vox_sorted_keys(const double* xyz, const long long* off, int B, int64_t cap_pts, const unsigned long long* minb,
                           double cell, double max_extent, unsigned long long* keys_in, unsigned long long* keys_out,
                           unsigned int* vals_in, unsigned int* vals_out, void* temp, size_t temp_bytes, int* overflow,
                           hipStream_t st)
{
    const unsigned nblk = (unsigned)((cap_pts + 255) / 256);
    const int bf = vox_pack_bits(B, max_extent, cell);
    unsigned int* k32_in = (unsigned int*)keys_in;
    unsigned int* k32_out = k32_in + cap_pts;
    hipLaunchKernelGGL(vox_keys, dim3(nblk), dim3(256), 0, st, xyz, off, B, (long long)cap_pts, minb, cell, keys_in, k32_in,
                       bf, vals_in, overflow);
    if (bf) {
        PCP_HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, (const unsigned int*)k32_in, k32_out,
                                              (const unsigned int*)vals_in, vals_out, (size_t)cap_pts, 0u,
                                              vox_cloud_bits(B) + 3u * (unsigned)bf, st));
        hipLaunchKernelGGL(vox_unpack_keys, dim3(nblk), dim3(256), 0, st, (const unsigned int*)k32_out, off, B,
                           (long long)cap_pts, bf, keys_out);
    } else {
        // 48 cell bits + the cloud number's: the padding keys (all ones) stay the largest under any prefix of the bits and
        // real keys are zero above the cloud bits, so the order equals the full 64-bit order
        PCP_HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, (const unsigned long long*)keys_in, keys_out,
                                              (const unsigned int*)vals_in, vals_out, (size_t)cap_pts, 0u,
                                              48u + vox_cloud_bits(B), st));
    }
    return PCP_OK;
}

struct VoxLayout {
    size_t minb, keys_in, keys_out, vals_in, vals_out, masks, tile_cnt, total, temp, temp_bytes, end;
};

static VoxLayout vox_layout(int B, int64_t cap)
{
    VoxLayout L;
    const size_t n = (size_t)(cap > 0 ? cap : 1), tiles = (n + PCP_TILE - 1) / PCP_TILE;
    size_t o = 0;
    L.minb = o;     o += pcp_align256((size_t)(B > 0 ? B : 1) * 3 * 8);
    L.keys_in = o;  o += pcp_align256(n * 8);
    L.keys_out = o; o += pcp_align256(n * 8);
    L.vals_in = o;  o += pcp_align256(n * 4);
    L.vals_out = o; o += pcp_align256(n * 4);
    L.masks = o;    o += pcp_align256(tiles * PCP_TILE_WORDS * 8);
    L.tile_cnt = o; o += pcp_align256(tiles * 4);
    L.total = o;    o += 256;
    L.temp = o;
    L.temp_bytes = vox_sort_temp_bytes((int64_t)n);
    o += pcp_align256(L.temp_bytes) + 256;
    L.end = o;
    return L;
}

extern "C" size_t pcp_voxel_workspace_bytes(int B, int64_t cap_pts)
{
    if (B < 0 || cap_pts < 0) return 0;
    return vox_layout(B, cap_pts).end;
}

extern "C" int pcp_voxel_downsample(const double* xyz, const float* rgb, const int64_t* offsets, int B,
                                    int64_t cap_pts, double voxel_size, double max_extent, double* out_xyz64,
                                    float* out_xyz32,
                                    float* out_rgb32, int32_t* out_count, int64_t* out_offsets,
                                    int64_t cap_out, int32_t* overflow, void* ws, size_t ws_bytes,
                                    void* stream)
{
    if (!offsets || !out_offsets || B < 0 || B > 65534 || cap_pts < 0 || cap_pts > 0xFFFFFFFFLL || cap_out < 0 ||
        !(voxel_size > 0.0))
        return PCP_ERR_INVALID_ARG;
    if (cap_pts > 0 && !xyz) return PCP_ERR_INVALID_ARG;
    if (out_rgb32 && !rgb) return PCP_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (overflow) PCP_FILL_TRY(overflow, 0u, 1, st);
    if (B == 0 || cap_pts == 0) {
        PCP_FILL_TRY(out_offsets, 0u, 2 * (size_t)(B + 1), st);
        return PCP_OK;
    }
    const VoxLayout L = vox_layout(B, cap_pts);
    if (!ws || ((uintptr_t)ws & 255) || ws_bytes < L.end) return PCP_ERR_WORKSPACE;
    char* w = (char*)ws;
    unsigned long long* minb = (unsigned long long*)(w + L.minb);
    unsigned long long* keys_in = (unsigned long long*)(w + L.keys_in);
    unsigned long long* keys_out = (unsigned long long*)(w + L.keys_out);
    unsigned int* vals_in = (unsigned int*)(w + L.vals_in);
    unsigned int* vals_out = (unsigned int*)(w + L.vals_out);
    unsigned long long* masks = (unsigned long long*)(w + L.masks);
    int* tile_cnt = (int*)(w + L.tile_cnt);
    long long* total = (long long*)(w + L.total);
    const long long* off = (const long long*)offsets;
    const unsigned nblk = (unsigned)((cap_pts + 255) / 256);
    const int tiles = (int)((cap_pts + PCP_TILE - 1) / PCP_TILE);

    PCP_FILL_TRY(minb, 0xFFFFFFFFu, (size_t)B * 6, st);
    hipLaunchKernelGGL(vox_minbound, dim3((nblk + VOX_MB_ITERS - 1) / VOX_MB_ITERS), dim3(256), 0, st, xyz, off, B, minb);
    {
        const int rc = vox_sorted_keys(xyz, off, B, cap_pts, minb, voxel_size, max_extent, keys_in, keys_out, vals_in,
                                       vals_out, (void*)(w + L.temp), L.temp_bytes, overflow, st);
        if (rc != PCP_OK) return rc;
    }
    hipLaunchKernelGGL(vox_heads, dim3(tiles), dim3(256), 0, st, keys_out, off, B, masks, tile_cnt);
    hipLaunchKernelGGL(pcp_scan_tiles, dim3(1), dim3(256), 0, st, tile_cnt, tiles, total);
    hipLaunchKernelGGL(vox_reduce, dim3(tiles), dim3(256), 0, st, keys_out, vals_out, xyz, rgb, off, B, masks,
                       tile_cnt, out_xyz64, out_xyz32, out_rgb32, out_count, (long long)cap_out, overflow);
    hipLaunchKernelGGL(pcp_csr_offsets, dim3((B + 256) / 256), dim3(256), 0, st, off, B, masks, tile_cnt, total,
                       (long long*)out_offsets);
    return pcp_launch_status();
}
