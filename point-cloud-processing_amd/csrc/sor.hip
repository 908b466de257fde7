// Statistical outlier removal of B CSR clouds on gfx950 (BASELINE config 3).
//
// Semantics: Open3D's PointCloud::RemoveStatisticalOutliers(nb_neighbors, std_ratio) — the reference
// tree has no outlier filter of its own.  For every point the mean of sqrt(d2)
// over its min(nb, M) nearest neighbours INCLUDING itself, d2 = ((dx*dx)+(dy*dy))+(dz*dz) in FP64
// (nanoflann L2 order), summed in ascending order; cloud mean and sample standard deviation
// accumulated strictly sequentially in point order (std::accumulate / std::inner_product); keep
// avg > 0 && avg < mean + std_ratio*std, order preserved.  PARITY UNPINNED against Open3D (absent
// here); bit-exact against the CPU restatement the tests hold — the k smallest distances are a multiset, so the KNN search
// strategy (grid here, kd-tree in Open3D, brute force in the tests) cannot change the result.
//
// Pipeline: uniform grid of `cell` metres keyed like the voxel filter (cloud | ix | iy | iz), stable
// radix sort, points gathered into sorted order; one thread per sorted point scans the (x,y) rows of the
// (2r+1)^3 neighbourhood (each row is one contiguous key run) for r = 1..3 until the k-th distance is provably inside the scanned block,
// else falls back to the whole cloud; one wave per cloud does the sequential statistics; the keep
// flags are compacted with the ballot / tile-scan machinery.
#include "pcp_internal.h"
#include <float.h>

#define SOR_RMAX 3
#ifndef SOR_RSTART
#define SOR_RSTART 1
#endif

__global__ void __launch_bounds__(256)
sor_gather_sorted(const double* __restrict__ xyz, const unsigned int* __restrict__ vals,
                  const long long* __restrict__ offsets, int B, double* __restrict__ sxyz)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= offsets[B]) return;
    const size_t i = vals[j];
    sxyz[3 * j] = xyz[3 * i]; sxyz[3 * j + 1] = xyz[3 * i + 1]; sxyz[3 * j + 2] = xyz[3 * i + 2];
}

// Sorted insertion, compare-and-shift form: every slot decides by itself, so the instructions are independent of each
// other — the form for the sparsely occupied per-point search below, where a dependent chain is not hidden.
typedef float pcp_f2 __attribute__((ext_vector_type(2)));

template <int KCAP>
__device__ __forceinline__ void sor_insert(double (&best)[KCAP], double d)
{
    if (d < best[KCAP - 1]) {
#pragma unroll
        for (int t = KCAP - 1; t > 0; --t) {
            const double lo = best[t - 1];
            best[t] = lo > d ? lo : (best[t] > d ? d : best[t]);
        }
        best[0] = best[0] > d ? d : best[0];
    }
}

// Sorted insertion as a min/max bubble: the candidate travels up the ascending array, every slot keeps the smaller of
// (slot, traveller) and passes the larger on; the largest value falls off the end.  Two FP64 instructions per slot and
// no compare / select pairs (five per slot); min and max return one of their operands unchanged, so the array holds the
// same values as the compare-and-shift form.  d must not be NaN (callers pass DBL_MAX for "nothing to insert").  The
// traveller makes the slots a dependent chain: good where many waves per SIMD hide it (sor_knn_coop), measured twice as
// slow as the compare form in the per-point search (two waves per SIMD).
template <int KCAP>
__device__ __forceinline__ void sor_insert1(double (&best)[KCAP], double d)
{
#pragma unroll
    for (int t = 0; t < KCAP; ++t) {
        double lo, up;
        asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(best[t]), "v"(d));
        asm("v_max_f64 %0, %1, %2" : "=v"(up) : "v"(best[t]), "v"(d));
        best[t] = lo;
        d = up;
    }
}

// k-th smallest = best[k-1]; written as a running maximum over the first k (ascending) slots so that the
// compiler keeps the array in registers instead of lowering a dynamic index through scratch memory
template <int KCAP>
__device__ __forceinline__ double sor_kth(const double (&best)[KCAP], int k)
{
    double v = best[0];
#pragma unroll
    for (int t = 1; t < KCAP; ++t) v = (t < k && best[t] > v) ? best[t] : v;
    return v;
}

// first sorted position in [lo,hi) whose key is >= key
__device__ __forceinline__ long long sor_lower_bound(const unsigned long long* __restrict__ keys, long long lo,
                                                     long long hi, unsigned long long key)
{
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Sorted insertion is ~150 FP64-wide instructions and, written inline, runs for the whole wave whenever ANY of
// its 64 lanes has a candidate that beats its k-th best — i.e. for almost every candidate (measured: 220
// VALU instructions per candidate).  So candidates that beat the lane's (slightly stale) bound are only
// appended to a small per-lane queue in LDS; when some lane's queue is full, all lanes drain theirs in
// lockstep and the insertion code runs with most lanes doing useful work.
#define SOR_Q 8

#ifdef SOR_DBG
__device__ unsigned int sor_dbg[8];
#endif
template <int KCAP>
__global__ void __launch_bounds__(256, 1)
sor_knn(const double* __restrict__ sxyz, const unsigned long long* __restrict__ keys,
        const unsigned int* __restrict__ vals, const long long* __restrict__ offsets, int B, double cell,
        int nb, const unsigned long long* __restrict__ minb, double* __restrict__ avg,
        const unsigned int* __restrict__ todo, const unsigned int* __restrict__ todo_cnt)
{
    __shared__ double queue[SOR_Q * 256];
    // with a to-do list (the points sor_knn_coop could not finish) thread t serves todo[t]; without, sorted point t
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (todo) {
        if (j >= (long long)*todo_cnt) return;
        j = todo[j];
    }
    if (j >= offsets[B]) return;
    const unsigned long long key = keys[j];
    const int b = (int)(key >> 48);
    const int cx = (int)((key >> 32) & 0xFFFF), cy = (int)((key >> 16) & 0xFFFF), cz = (int)(key & 0xFFFF);
    const long long lo = offsets[b], hi = offsets[b + 1];
    const long long M = hi - lo;
    const int k = (long long)nb < M ? nb : (int)M;
    const double px = sxyz[3 * j], py = sxyz[3 * j + 1], pz = sxyz[3 * j + 2];
    const unsigned long long cloud = (unsigned long long)b << 48;
    // grid origin exactly as vox_keys computes it, and a slack that covers the rounding of the cell-index division
    // there (a few ulps of the coordinates): row (x,y) holds no point closer to p than the gap computed below
    const double ox = vox_unord(minb[3 * b]) - cell * 0.5, oy = vox_unord(minb[3 * b + 1]) - cell * 0.5;
    const double slack = 1e-9 * cell + 1e-14 * (fabs(px) + fabs(py) + fabs(ox) + fabs(oy) + 65536.0 * cell);
    double* q = queue + threadIdx.x;        // this lane's queue: q[slot * 256]
    double best[KCAP];
    double tau = DBL_MAX;                   // candidates >= tau cannot enter the k best (stale between drains: only larger)
    int qn = 0;
    auto reset = [&]() {
#pragma unroll
        for (int t = 0; t < KCAP; ++t) best[t] = DBL_MAX;
        tau = DBL_MAX;
        qn = 0;
    };
    auto drain = [&]() {
#pragma unroll
        for (int sl = 0; sl < SOR_Q; ++sl)
            if (__any(sl < qn)) sor_insert<KCAP>(best, sl < qn ? q[sl * 256] : DBL_MAX);
        qn = 0;
        tau = sor_kth<KCAP>(best, k);
    };
    auto push = [&](double d) {
        if (d < tau) { q[qn * 256] = d; ++qn; }
        if (__any(qn == SOR_Q)) drain();
    };
    bool done = false;
#ifdef SOR_DBG
    int r_done = 0;
#endif
    if (M > k) {
        // a to-do point comes from sor_knn_coop, which has scanned its first ring already and found it too small
        for (int r = todo ? 2 : SOR_RSTART; r <= SOR_RMAX && !done; ++r) {
            reset();
            const int z0 = cz - r < 0 ? 0 : cz - r, z1 = cz + r > 65535 ? 65535 : cz + r;
            // rows nearest the query's own row first (offsets 0,-1,+1,-2,+2,...): the bound tightens early and
            // fewer candidates of the outer rows have to be queued
            for (int ix = 0; ix <= 2 * r; ++ix) {
                const int x = cx + ((ix & 1) ? -((ix + 1) >> 1) : (ix >> 1));
                if (x < 0 || x > 65535) continue;
                double gx = x > cx ? (ox + (double)x * cell) - px : (x < cx ? px - (ox + (double)(x + 1) * cell) : 0.0);
                gx = gx > slack ? gx - slack : 0.0;
                if (gx * gx >= tau) continue;                            // the whole x slab is farther than the k-th best
                for (int iy = 0; iy <= 2 * r; ++iy) {                    // one contiguous key run per (x,y) row
                    const int y = cy + ((iy & 1) ? -((iy + 1) >> 1) : (iy >> 1));
                    if (y < 0 || y > 65535) continue;
                    double gy = y > cy ? (oy + (double)y * cell) - py : (y < cy ? py - (oy + (double)(y + 1) * cell) : 0.0);
                    gy = gy > slack ? gy - slack : 0.0;
                    if (gx * gx + gy * gy >= tau) continue;              // no point of this row can enter the k best
                    const unsigned long long row = cloud | ((unsigned long long)x << 32) | ((unsigned long long)y << 16);
                    const unsigned long long k1 = row | (unsigned long long)z1;
                    for (long long t = sor_lower_bound(keys, lo, hi, row | (unsigned long long)z0); t < hi; ++t) {
                        if (keys[t] > k1) break;
                        const double dx = px - sxyz[3 * t], dy = py - sxyz[3 * t + 1], dz = pz - sxyz[3 * t + 2];
                        push(((dx * dx) + (dy * dy)) + (dz * dz));
                    }
                }
            }
            drain();
            const double lim = ((double)r * cell) * (1.0 - 1e-9);      // every point closer than this was scanned
            done = tau <= lim * lim;
#ifdef SOR_DBG
            if (done) r_done = r;
#endif
        }
    }
#ifdef SOR_DBG
    if (todo) atomicAdd(&sor_dbg[r_done], 1u);
#endif
    if (!done) {                                                     // isolated point or tiny cloud: whole cloud
        reset();
        for (long long t = lo; t < hi; ++t) {
            const double dx = px - sxyz[3 * t], dy = py - sxyz[3 * t + 1], dz = pz - sxyz[3 * t + 2];
            push(((dx * dx) + (dy * dy)) + (dz * dz));
        }
        drain();
    }
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < KCAP; ++t)
        if (t < k) s += sqrt(best[t]);
    avg[vals[j]] = s / (double)k;
}


// ---- wave-cooperative k-NN (first ring) --------------------------------------------------------------------
// One wave serves 64 consecutive SORTED points.  Consecutive sorted points share their (x,y) grid row and walk it in
// z, so their 27-cell neighbourhoods overlap almost completely: the union over a row group is nine contiguous key
// runs (rows x-1..x+1, y-1..y+1, cells zmin-1..zmax+1).  The wave finds the nine runs with eighteen binary searches
// done side by side by eighteen lanes, stages the runs' points in LDS with coalesced loads, and every lane scans
// the staged candidates with LDS broadcast reads — no per-lane global gathers, no per-lane binary searches, no
// divergent row loops.  Candidates that may beat a lane's bound set a bit in its pass words and the lanes drain their
// bits in lockstep into a sorted top-k kept in registers.  Scanning a superset of a lane's own 27
// cells cannot change its result: the k smallest distances are a multiset and the ring proof below only needs the 27
// cells to be covered.  A lane whose k-th distance is not provably inside the first ring (or whose cloud has no more
// than k points) is appended to a to-do list that sor_knn finishes with its wider search.
#ifndef SORC_CAND
#define SORC_CAND 192          // candidates staged per wave and chunk (FP64 x,y,z + float4 + pass words: 9 KB, so that
                               // LDS allows the sixteen waves per CU the registers allow); 128 / 256 / 512 were slower
#endif
#ifndef SORC_WAVES
#define SORC_WAVES 1           // waves per workgroup: the waves share nothing, and single-wave groups pack best (2 / 4: +4 / +10 %)
#endif

__device__ __forceinline__ long long sor_rl64(long long v, int lane)
{
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, lane);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), lane);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// waves per SIMD the register allocation aims at: the sorted top-k (2 x KCAP registers) decides what fits
#define SORC_EU(KCAP) ((KCAP) <= 20 ? 4 : ((KCAP) <= 24 ? 3 : 2))
template <int KCAP>
__global__ void __launch_bounds__(64 * SORC_WAVES) __attribute__((amdgpu_waves_per_eu(SORC_EU(KCAP), SORC_EU(KCAP))))
sor_knn_coop(const double* __restrict__ sxyz, const unsigned long long* __restrict__ keys,
             const unsigned int* __restrict__ vals, const long long* __restrict__ offsets, int B, double cell,
             int nb, double* __restrict__ avg, unsigned int* __restrict__ todo, unsigned int* __restrict__ todo_cnt)
{
    // The scan runs in FLOAT (candidate coordinates rounded once, at staging) and only FILTERS: a candidate whose float
    // distance is within a rigorous error margin of the lane's bound gets a pass bit, everything else is provably
    // farther than the bound.  The drain reads the passed candidates' FP64 coordinates back from LDS (a per-lane
    // gather), computes the exact distance in Open3D's operation order and inserts it into the sorted top-k — every
    // value that decides anything is exact, and a candidate costs 4.5 FP32-rate instructions instead of eight FP64 ones.
    // (Gathering the exact coordinates from global memory instead made the kernel texture-address bound: 64 lanes x 3
    // loads, each its own cache line, per passed candidate.)
    __shared__ double cand_s[SORC_WAVES][3][SORC_CAND];
    // the same candidates rounded to float, for the filter: two per row {x0,x1,y0,y1,z0,z1,-,-} so that a pair goes
    // through the packed FP32 instructions
    __shared__ __attribute__((aligned(16))) float cand32_s[SORC_WAVES][SORC_CAND / 2][8];
    __shared__ unsigned int pass_s[SORC_WAVES][SORC_CAND / 32][64];       // per lane: one bit per staged candidate that passed
    const int lane = pcp_lane(), wave = threadIdx.x >> 6;
    const long long n = offsets[B];
    const long long j0 = ((long long)blockIdx.x * SORC_WAVES + wave) * 64;
    if (j0 >= n) return;                                              // wave-uniform
    const long long j = j0 + lane;
    const bool have = j < n;
    const long long jj = have ? j : n - 1;
    const unsigned long long key = keys[jj];
    const int b = (int)(key >> 48);
    const int cz = (int)(key & 0xFFFF);
    const long long lo = offsets[b], hi = offsets[b + 1];
    const long long M = hi - lo;
    const int k = (long long)nb < M ? nb : (int)M;
    const double px = sxyz[3 * jj], py = sxyz[3 * jj + 1], pz = sxyz[3 * jj + 2];
    double (*cand)[SORC_CAND] = cand_s[wave];
    float (*cand32)[8] = cand32_s[wave];
    unsigned int (*passw)[64] = pass_s[wave];
    double best[KCAP];
#pragma unroll
    for (int t = 0; t < KCAP; ++t) best[t] = DBL_MAX;
    // Only a k-th distance inside the first ring can be proven here (every point closer than `lim` is scanned), so a
    // candidate at or beyond lim^2 is of no use: the bound starts there instead of at infinity.  The early, far
    // candidates are then never queued or inserted (about 44 candidates per lane lie inside the ring at the density
    // sor_cell_size aims for, against ~700 scanned), and a lane that ends with fewer than k of them goes to the to-do list.
    const double lim = cell * (1.0 - 1e-9);
    const double lim2 = lim * lim;
    double tau = lim2;
#ifdef SORC_STATS
    int st_drain = 0, st_ins = 0, st_push = 0, st_grp = 0, st_cand = 0;
    unsigned long long tm_search = 0, tm_stage = 0, tm_scan = 0, tm_drain = 0, tm0 = __builtin_amdgcn_s_memtime(), tmx;
#define SORC_T(acc) do { tmx = __builtin_amdgcn_s_memtime(); acc += tmx - tm0; tm0 = tmx; } while (0)
#else
#define SORC_T(acc) do { } while (0)
#endif
    const bool use = have && M > k;
    unsigned long long rem = __ballot(use);
    while (rem) {                                                     // one trip per (cloud, x, y) row present in the wave
        const int leader = __ffsll((long long)rem) - 1;
        const unsigned long long rowkey = (unsigned long long)sor_rl64((long long)(key >> 16), leader);
        const unsigned long long grp = __ballot(use && (key >> 16) == rowkey);
        rem &= ~grp;
        const bool act = (grp >> lane) & 1ull;
        // the group's lanes are consecutive sorted points of one row: z ascends from its first to its last lane
        const int zfirst = __builtin_amdgcn_readlane(cz, __ffsll((long long)grp) - 1);
        const int zlast = __builtin_amdgcn_readlane(cz, 63 - __clzll((long long)grp));
        const int zA = zfirst > 0 ? zfirst - 1 : 0, zB = zlast < 65535 ? zlast + 1 : 65535;
        const int gx = (int)((rowkey >> 16) & 0xFFFF), gy = (int)(rowkey & 0xFFFF);
        const unsigned long long cloud = (rowkey >> 32) << 48;
        const long long glo = sor_rl64(lo, leader), ghi = sor_rl64(hi, leader);
        // eighteen searches side by side: lane s < 9 finds where row s begins at cell zA, lane 9 + s where it ends at zB
        long long pos = glo;
        {
            // rows in the order they are scanned: the group's own row first, then the four that share a face with it,
            // then the four diagonal ones — near candidates first, so the lanes' bounds tighten early
            const int s = lane < 9 ? lane : lane - 9;
            const int ddx = (0x28215 >> (2 * s)) & 3, ddy = (0x22161 >> (2 * s)) & 3;   // packed {1,1,1,0,2,0,0,2,2} / {1,0,2,1,1,0,2,0,2}
            const int x = gx + ddx - 1, y = gy + ddy - 1;
            if (lane < 18 && x >= 0 && x <= 65535 && y >= 0 && y <= 65535) {
                const unsigned long long row = cloud | ((unsigned long long)x << 32) | ((unsigned long long)y << 16);
                long long a = glo, e = ghi;
                if (lane < 9) {                                       // first key >= row|zA
                    const unsigned long long t = row | (unsigned long long)zA;
                    while (a < e) { const long long mid = (a + e) >> 1; if (keys[mid] < t) a = mid + 1; else e = mid; }
                } else {                                              // first key > row|zB
                    const unsigned long long t = row | (unsigned long long)zB;
                    while (a < e) { const long long mid = (a + e) >> 1; if (keys[mid] <= t) a = mid + 1; else e = mid; }
                }
                pos = a;
            }
        }
        long long seg0[9];
        int pre[10];
        pre[0] = 0;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            seg0[s] = sor_rl64(pos, s);
            const long long e = sor_rl64(pos, 9 + s);
            pre[s + 1] = pre[s] + (int)(e > seg0[s] ? e - seg0[s] : 0);
        }
        const int N = pre[9];
#ifdef SORC_STATS
        ++st_grp; st_cand += N;
        SORC_T(tm_search);
#endif
        auto sorted_pos = [&](int ord) -> long long {                 // position in the sorted arrays of the group's ord-th candidate
            long long g = seg0[0] + ord;
#pragma unroll
            for (int t = 1; t < 9; ++t) g = ord >= pre[t] ? seg0[t] + (ord - pre[t]) : g;
            return g;
        };
        // float error margin: a coordinate c and its float differ by <= 2^-24 |c|, every coordinate here is within
        // Rmax of the group's first point o, so a component difference is off by <= 2^-23 (|o| + Rmax) plus its own
        // rounding; with sqrt(tau) <= cell <= Rmax the margin below covers the three components and the float products
        // and sums several times over.  Far-away clouds (|o| >> cell) make the filter pass everything: slower, never wrong.
        const int first_lane = __ffsll((long long)grp) - 1;
        const double omag = fabs(__longlong_as_double(sor_rl64(__double_as_longlong(px), first_lane))) +
                            fabs(__longlong_as_double(sor_rl64(__double_as_longlong(py), first_lane))) +
                            fabs(__longlong_as_double(sor_rl64(__double_as_longlong(pz), first_lane)));
        const float qx = (float)px, qy = (float)py, qz = (float)pz;
        const double Rmax = (double)(zB - zA + 4) * cell;             // cells zA..zB in z, three cells in x and y, plus slack
        const double emargin = 0x1p-20 * (omag + 2.0 * Rmax);
        auto float_bound = [&](double t) -> float {
            const double r = sqrt(t) + emargin;
            return (float)(r * r * (1.0 + 0x1p-18));       // >= 16 float ulps above r^2: covers the strict sign test too
        };
        float tau32 = float_bound(tau);
        // Drain of one staged chunk: every lane walks ITS pass bits (most significant first = staging order) — two
        // candidates per trip so that the LDS gathers of one overlap the insertion of the other —, reads the FP64
        // coordinates back, computes the exact distance in Open3D's operation order and bubbles it into its top-k.  The
        // trips run in lockstep until the slowest lane is through.
        auto drain = [&](int nwords) {
#ifdef SORC_STATS
            ++st_drain;
            SORC_T(tm_scan);
#endif
            int wi = 0;
            unsigned int m = passw[0][lane];
            for (;;) {
                // lanes whose word is used up move on to their next non-empty one
                while (m == 0u && wi + 1 < nwords) { ++wi; m = passw[wi][lane]; }
                if (!__any(m != 0u)) break;
                double d[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool live = m != 0u;
                    const int bit = live ? __clz(m) : 0;
                    m &= ~(0x80000000u >> bit);                       // no-op on an empty word (bit 31 of 0)
                    const int c = live ? wi * 32 + bit : 0;
                    const double dx = px - cand[0][c], dy = py - cand[1][c], dz = pz - cand[2][c];
                    d[u] = live ? ((dx * dx) + (dy * dy)) + (dz * dz) : DBL_MAX;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    // the bound a value must beat to enter: exact when every slot is in use (k == KCAP), else the
                    // (only larger) bound of the last drain; float-margin false positives stop here
                    const double bound = k == KCAP ? fmin(tau, best[KCAP - 1]) : tau;
                    if (__any(d[u] < bound)) {
#ifdef SORC_STATS
                        ++st_ins;
#endif
                        sor_insert1<KCAP>(best, d[u] < bound ? d[u] : DBL_MAX);
                    }
                }
            }
            tau = fmin(lim2, sor_kth<KCAP>(best, k));
            tau32 = float_bound(tau);
            SORC_T(tm_drain);
        };
        for (int cb = 0; cb < N; cb += SORC_CAND) {
            const int cn = N - cb < SORC_CAND ? N - cb : SORC_CAND;
            for (int i = lane; i < cn; i += 64) {                     // stage: consecutive lanes read consecutive sorted points
                const long long g = sorted_pos(cb + i);
                const double x = sxyz[3 * g], y = sxyz[3 * g + 1], z = sxyz[3 * g + 2];
                cand[0][i] = x; cand[1][i] = y; cand[2][i] = z;
                float* c32 = &cand32[i >> 1][i & 1];
                c32[0] = (float)x; c32[2] = (float)y; c32[4] = (float)z;
            }
            SORC_T(tm_stage);
            // filter: 32 candidates per word; the sign of (float distance - bound) is shifted into the word, one
            // v_alignbit per candidate, so a candidate costs eight VALU instructions and no LDS write.  Strictly below
            // the bound here, at or below it in the margin analysis: float_bound() leaves several ulps of room.
            const int nwords = (cn + 31) >> 5;
            const pcp_f2 q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz}, tau2 = {tau32, tau32};
            for (int w = 0; w < nwords; ++w) {
                unsigned int m = 0u;
#pragma unroll
                for (int u = 0; u < 32; u += 2) {                     // rows past cn hold stale data: masked below
                    const float4 xy = *(const float4*)&cand32[w * 16 + (u >> 1)][0];
                    const float2 zz = *(const float2*)&cand32[w * 16 + (u >> 1)][4];
                    const pcp_f2 dx = q2x - (pcp_f2){xy.x, xy.y}, dy = q2y - (pcp_f2){xy.z, xy.w}, dz = q2z - (pcp_f2){zz.x, zz.y};
                    const pcp_f2 t = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx)) - tau2;
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(t.x), 31);   // m = m << 1 | sign(t)
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(t.y), 31);
                }
                const int left = cn - w * 32;                         // candidates of this word that exist
                const unsigned int valid = left >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> left);
                passw[w][lane] = act ? (m & valid) : 0u;
#ifdef SORC_STATS
                st_push += act ? __popc(m & valid) : 0;
#endif
            }
            drain(nwords);                                            // pass bits refer to this chunk
        }
    }
    const bool done = use && sor_kth<KCAP>(best, k) < lim2;           // k real entries, all inside the first ring
#ifdef SORC_STATS
    {
        int pmax = st_push, psum = st_push;
        for (int dd = 32; dd >= 1; dd >>= 1) { pmax = max(pmax, __shfl_xor(pmax, dd, 64)); psum += __shfl_xor(psum, dd, 64); }
        const int nd = (int)__popcll(__ballot(done));
        if (blockIdx.x % 16000 == 17 && wave == SORC_WAVES - 1 && lane == 0)
            printf("[sorc] blk %d groups %d cand %d drains %d inserts %d push max %d avg %.1f done %d | cycles search %llu stage %llu scan %llu drain %llu\n", blockIdx.x, st_grp, st_cand, st_drain, st_ins, pmax, psum / 64.0, nd, tm_search, tm_stage, tm_scan, tm_drain);
    }
#endif
    if (have && done) {
        double sum = 0.0;
#pragma unroll
        for (int t = 0; t < KCAP; ++t)
            if (t < k) sum += sqrt(best[t]);
        avg[vals[j]] = sum / (double)k;
    }
    // one append per wave: the wave's unfinished points take consecutive to-do slots, so the lanes of the per-point
    // search work on spatial neighbours (similar rows, similar trip counts)
    const unsigned long long left = __ballot(have && !done);
    if (left) {
        unsigned int slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(todo_cnt, (unsigned int)__popcll(left));
        slot0 = __builtin_amdgcn_readfirstlane(slot0);
        if (have && !done) todo[slot0 + (unsigned int)__popcll(left & ((1ull << lane) - 1ull))] = (unsigned int)j;
    }
}

// One workgroup per cloud: strictly sequential sums (point order), as std::accumulate does.  All four waves turn a
// block of SOR_ST_BLOCK values into the addends in parallel (0.0 for a skipped value and for the padding of the last
// group: adding +0.0 to a non-negative running sum is exact) and park them in LDS; wave 0 then runs the serial chain
// over the block — broadcast LDS reads of the NEXT 32 addends are issued ahead of the 32 dependent v_add_f64 of the
// current group, so the chain's own latency is all that is left.   stats[b] = {mean, std, threshold, 0}
#define SOR_ST_BLOCK 8192
#define SOR_ST_GROUP 32
__global__ void __launch_bounds__(256)
sor_stats(const double* __restrict__ avg, const long long* __restrict__ offsets, double std_ratio,
          double* __restrict__ stats)
{
    // 66,048 bytes of static LDS: more than the 64 KiB every gfx9 target before gfx950 allows — this library is built
    // for gfx950 (160 KB of LDS per CU) only, see the Makefile
    static_assert((SOR_ST_BLOCK + 2 * SOR_ST_GROUP) * sizeof(double) <= 160 * 1024, "sor_stats' addend buffer must fit gfx950's LDS");
    __shared__ double term_s[SOR_ST_BLOCK + 2 * SOR_ST_GROUP];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long lo = offsets[b], hi = offsets[b + 1];
    const long long M = hi - lo;
    double res[2] = {0.0, 0.0};
    double mean = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        double acc = 0.0;
        for (long long base = lo; base < hi; base += SOR_ST_BLOCK) {
            const int n = (int)((hi - base) < SOR_ST_BLOCK ? (hi - base) : SOR_ST_BLOCK);
            const int npad = (n + 2 * SOR_ST_GROUP - 1) / (2 * SOR_ST_GROUP) * (2 * SOR_ST_GROUP);
            __syncthreads();                                          // the previous block's chain is through
#pragma unroll 8
            for (int i = tid; i < npad; i += 256) {
                const double y = i < n ? avg[base + i] : 0.0;
                double term;
                if (pass == 0) term = y > 0.0 ? y : 0.0;
                else term = y > 0.0 ? (y - mean) * (y - mean) : 0.0;
                term_s[i] = term;
            }
            __syncthreads();
            if (tid < 64) {
                // two register groups, A and B: the reads of one are in flight under the other's chain.  The
                // sched_barriers keep hipcc from sinking the reads down to their first use.
                double ga[SOR_ST_GROUP], gb[SOR_ST_GROUP];
#pragma unroll
                for (int t = 0; t < SOR_ST_GROUP; ++t) ga[t] = term_s[t];
                for (int g = 0; g < npad; g += 2 * SOR_ST_GROUP) {
#pragma unroll
                    for (int t = 0; t < SOR_ST_GROUP; ++t) gb[t] = term_s[g + SOR_ST_GROUP + t];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < SOR_ST_GROUP; ++t) acc = acc + ga[t];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < SOR_ST_GROUP; ++t) ga[t] = term_s[g + 2 * SOR_ST_GROUP + t];   // last trip: pad rows
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < SOR_ST_GROUP; ++t) acc = acc + gb[t];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        res[pass] = acc;                                              // wave 0 only; the other waves' copies are unused
        if (pass == 0) {
            // every wave needs the mean for its share of the second pass' addends
            __syncthreads();
            if (tid == 0) term_s[0] = acc / (double)M;
            __syncthreads();
            mean = term_s[0];
        }
    }
    if (tid == 0) {
        const double sd = sqrt(res[1] / (double)(M - 1));
        stats[4 * b] = mean; stats[4 * b + 1] = sd; stats[4 * b + 2] = mean + std_ratio * sd; stats[4 * b + 3] = 0.0;
    }
}

__global__ void __launch_bounds__(256)
sor_flags(const double* __restrict__ avg, const long long* __restrict__ offsets, int B,
          const double* __restrict__ stats, unsigned long long* __restrict__ masks, int* __restrict__ tile_cnt)
{
    __shared__ int wave_cnt[4];
    const long long n = offsets[B];
    const int lane = pcp_lane(), wave = threadIdx.x >> 6;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int word = wave * 4 + r;
        const long long i = (long long)blockIdx.x * PCP_TILE + word * 64 + lane;
        int keep = 0;
        if (i < n) {
            const int b = vox_cloud_of(offsets, B, i);
            const double a = avg[i];
            keep = a > 0.0 && a < stats[4 * b + 2];
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) masks[(size_t)blockIdx.x * PCP_TILE_WORDS + word] = m;
        cnt += __popcll(m);
    }
    if (lane == 0) wave_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ void __launch_bounds__(256)
sor_scatter(const double* __restrict__ xyz, const float* __restrict__ rgb, const long long* __restrict__ offsets,
            int B, const unsigned long long* __restrict__ masks, const int* __restrict__ tile_off, double* out64,
            float* out32, float* out_rgb, long long* out_index, long long cap_out, int* overflow)
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
    const int nk = pre[PCP_TILE_WORDS];
    for (int t = threadIdx.x; t < nk; t += blockDim.x) {
        int w = 0;
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1)
            if (pre[w + s] <= t) w += s;
        const long long i = (long long)tile * PCP_TILE + w * 64 + pcp_select_bit(m[w], t - pre[w]);
        const long long row = (long long)tile_off[tile] + t;
        if (row >= cap_out) { if (overflow) *overflow = 1; continue; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = xyz[3 * i + c];
            if (out64) out64[3 * row + c] = v;
            if (out32) out32[3 * row + c] = (float)v;
            if (out_rgb) out_rgb[3 * row + c] = rgb[3 * i + c];
        }
        if (out_index) out_index[row] = i - offsets[vox_cloud_of(offsets, B, i)];
    }
}

struct SorLayout { VoxLayout v; size_t sxyz, avg, stats, todo, todo_cnt, end; };

static SorLayout sor_layout(int B, int64_t cap)
{
    SorLayout L;
    L.v = vox_layout(B, cap);
    const size_t n = (size_t)(cap > 0 ? cap : 1);
    size_t o = L.v.end;
    L.sxyz = o;  o += pcp_align256(n * 24);
    L.avg = o;   o += pcp_align256(n * 8);
    L.stats = o; o += pcp_align256((size_t)(B > 0 ? B : 1) * 32);
    L.todo = o;  o += pcp_align256(n * 4);
    L.todo_cnt = o; o += 256;
    L.end = o;
    return L;
}

extern "C" size_t pcp_sor_workspace_bytes(int B, int64_t cap_pts)
{
    if (B < 0 || cap_pts < 0) return 0;
    return sor_layout(B, cap_pts).end;
}

extern "C" int pcp_statistical_outlier_removal(const double* xyz, const float* rgb, const int64_t* offsets,
                                               int B, int64_t cap_pts, int nb_neighbors, double std_ratio,
                                               double cell_size, double max_extent, double* out_xyz64, float* out_xyz32,
                                               float* out_rgb32, int64_t* out_index, int64_t* out_offsets,
                                               double* avg_dist, int64_t cap_out, int32_t* overflow, void* ws,
                                               size_t ws_bytes, void* stream)
{
    if (!offsets || !out_offsets || B < 0 || B > 65534 || cap_pts < 0 || cap_pts > 0xFFFFFFFFLL || cap_out < 0 ||
        nb_neighbors < 1 || nb_neighbors > 32 || !(cell_size > 0.0))
        return PCP_ERR_INVALID_ARG;
    if (cap_pts > 0 && !xyz) return PCP_ERR_INVALID_ARG;
    if (out_rgb32 && !rgb) return PCP_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (overflow) PCP_FILL_TRY(overflow, 0u, 1, st);
    if (B == 0 || cap_pts == 0) {
        PCP_FILL_TRY(out_offsets, 0u, 2 * (size_t)(B + 1), st);
        return PCP_OK;
    }
    const SorLayout L = sor_layout(B, cap_pts);
    if (!ws || ((uintptr_t)ws & 255) || ws_bytes < L.end) return PCP_ERR_WORKSPACE;
    char* w = (char*)ws;
    unsigned long long* minb = (unsigned long long*)(w + L.v.minb);
    unsigned long long* keys_in = (unsigned long long*)(w + L.v.keys_in);
    unsigned long long* keys_out = (unsigned long long*)(w + L.v.keys_out);
    unsigned int* vals_in = (unsigned int*)(w + L.v.vals_in);
    unsigned int* vals_out = (unsigned int*)(w + L.v.vals_out);
    unsigned long long* masks = (unsigned long long*)(w + L.v.masks);
    int* tile_cnt = (int*)(w + L.v.tile_cnt);
    long long* total = (long long*)(w + L.v.total);
    double* sxyz = (double*)(w + L.sxyz);
    double* avg = avg_dist ? avg_dist : (double*)(w + L.avg);
    double* stats = (double*)(w + L.stats);
    const long long* off = (const long long*)offsets;
    const unsigned nblk = (unsigned)((cap_pts + 255) / 256);
    const int tiles = (int)((cap_pts + PCP_TILE - 1) / PCP_TILE);

    PCP_FILL_TRY(minb, 0xFFFFFFFFu, (size_t)B * 6, st);
    hipLaunchKernelGGL(vox_minbound, dim3((nblk + VOX_MB_ITERS - 1) / VOX_MB_ITERS), dim3(256), 0, st, xyz, off, B, minb);
    {
        const int rc = vox_sorted_keys(xyz, off, B, cap_pts, minb, cell_size, max_extent, keys_in, keys_out, vals_in,
                                       vals_out, (void*)(w + L.v.temp), L.v.temp_bytes, overflow, st);
        if (rc != PCP_OK) return rc;
    }
    hipLaunchKernelGGL(sor_gather_sorted, dim3(nblk), dim3(256), 0, st, xyz, vals_out, off, B, sxyz);
    unsigned int* todo = (unsigned int*)(w + L.todo);
    unsigned int* todo_cnt = (unsigned int*)(w + L.todo_cnt);
    PCP_FILL_TRY(todo_cnt, 0u, 1, st);
    // first ring cooperatively (one wave per 64 sorted points); whatever it cannot prove goes through the per-point search
#define PCP_SOR_LAUNCH(KC)                                                                                              \
    do {                                                                                                                \
        hipLaunchKernelGGL(sor_knn_coop<KC>, dim3((unsigned)((cap_pts + 64 * SORC_WAVES - 1) / (64 * SORC_WAVES))),      \
                           dim3(64 * SORC_WAVES), 0, st, sxyz, keys_out, vals_out, off, B, cell_size,                    \
                           nb_neighbors, avg, todo, todo_cnt);                                                          \
        hipLaunchKernelGGL(sor_knn<KC>, dim3(nblk), dim3(256), 0, st, sxyz, keys_out, vals_out, off, B, cell_size,       \
                           nb_neighbors, minb, avg, (const unsigned int*)todo, (const unsigned int*)todo_cnt);          \
    } while (0)
    if (nb_neighbors <= 8) PCP_SOR_LAUNCH(8);
    else if (nb_neighbors <= 16) PCP_SOR_LAUNCH(16);
    else if (nb_neighbors <= 20) PCP_SOR_LAUNCH(20);
    else if (nb_neighbors <= 24) PCP_SOR_LAUNCH(24);
    else PCP_SOR_LAUNCH(32);
#undef PCP_SOR_LAUNCH
#ifdef SOR_DBG
    {
        unsigned int h[8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(sor_dbg), sizeof(h));
        fprintf(stderr, "[sor_dbg] to-do points finished at ring 1/2/3: %u %u %u, whole cloud: %u\n", h[1], h[2], h[3], h[0]);
        unsigned int z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(sor_dbg), z, sizeof(z));
    }
#endif
    hipLaunchKernelGGL(sor_stats, dim3(B), dim3(256), 0, st, avg, off, std_ratio, stats);
    hipLaunchKernelGGL(sor_flags, dim3(tiles), dim3(256), 0, st, avg, off, B, stats, masks, tile_cnt);
    hipLaunchKernelGGL(pcp_scan_tiles, dim3(1), dim3(256), 0, st, tile_cnt, tiles, total);
    hipLaunchKernelGGL(sor_scatter, dim3(tiles), dim3(256), 0, st, xyz, rgb, off, B, masks, tile_cnt, out_xyz64,
                       out_xyz32, out_rgb32, (long long*)out_index, (long long)cap_out, overflow);
    hipLaunchKernelGGL(pcp_csr_offsets, dim3((B + 256) / 256), dim3(256), 0, st, off, B, masks, tile_cnt, total,
                       (long long*)out_offsets);
    return pcp_launch_status();
}
