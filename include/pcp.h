/*
 * pcp.h — C ABI of libpcp_hip.so: the MI355X (gfx950) implementation of the DP3
 * point-cloud preprocessing hot path of darshil0805/Point-Cloud-Processing.
 *
 * The reference has no FFI of its own: its boundary is a handful of Python
 * functions.  Every entry point below names the reference
 * function (file:line under data_processing_pipeline/) whose arithmetic it
 * replaces.  A maintainer binds these with ctypes.
 *
 * Conventions
 *   - Every data pointer is a DEVICE pointer owned by the caller (e.g. a torch
 *     ROCm tensor's data_ptr()), contiguous, of the stated dtype.  `pcp_params`
 *     is a HOST struct read during the call.
 *   - Work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream).  Calls are asynchronous; nothing here synchronises the
 *     device, allocates device memory or frees caller memory.
 *   - Scratch comes from a caller-provided workspace sized by the matching
 *     *_workspace_bytes query.  Workspaces must be 256-byte aligned.
 *   - Return value: 0 = PCP_OK, negative = pcp_status.  No exception or abort
 *     crosses the ABI.  pcp_strerror() gives a static message.
 *   - Devices: kernels are launched on the CURRENT HIP device.  On a multi-GPU host the caller makes
 *     the device that owns `stream` and the buffers current (hipSetDevice / torch.cuda.device)
 *     before the call; the library never changes the current device.
 *   - No mutable state is shared between calls: calls on different streams / workspaces are
 *     independent and thread-safe.  (Two process-wide footnotes: the FPS shapes that need more than
 *     48 KB of LDS set that per-kernel attribute, to the same value, on every launch; and
 *     pcp_fps_batched_static owns one static device buffer per GPU, shared by all its calls there.)
 */
#ifndef PCP_H
#define PCP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PCP_OK = 0,
    PCP_ERR_INVALID_ARG = -1,   /* NULL pointer, negative size, bad enum value      */
    PCP_ERR_WORKSPACE = -2,     /* workspace NULL / too small / misaligned           */
    PCP_ERR_CAPACITY = -3,      /* caller-declared capacity cannot hold the result   */
    PCP_ERR_LAUNCH = -4,        /* HIP reported an error launching a kernel          */
    PCP_ERR_NO_DEVICE = -5,     /* no gfx950 device visible                          */
    PCP_ERR_EMPTY = -6          /* a frame has zero points where >=1 is required     */
} pcp_status;

/* Pinhole intrinsics, the dict built at extract_point_clouds_one_cam.py:53-54. */
typedef struct { double fx, fy, cx, cy; } pcp_intr;

/*
 * Every constant of the per-frame path in one place.
 * pcp_default_params() fills in the reference's values.
 */
typedef struct {
    pcp_intr depth;        /* depth camera intrinsics                     XPC:106            */
    pcp_intr color;        /* colour camera intrinsics                    XPC:107            */
    double T_d2c[16];      /* depth->colour 4x4, row-major                XPC:17-22 (f32 values widened) */
    double ext[16];        /* base<-camera 4x4, row-major                 S1:13-18 / XPC:24-29 */
    double ws_lo[3];       /* workspace box, strict lower bounds          DP3:29-33          */
    double ws_hi[3];       /* workspace box, strict upper bounds          DP3:29-33          */
    float  z_min;          /* depth validity, f32 compare  (0.01f)        XPC:109            */
    float  z_max;          /* depth validity, f32 compare  (5.0f)         XPC:109            */
    double zc_min;         /* colour-frame z threshold     (0.01)         XPC:75             */
    double zc_clamp;       /* projection divisor floor     (0.001)        XPC:70-71          */
    int32_t apply_crop;    /* 1: fuse the DP3:35-39 crop into the compaction predicate       */
    int32_t transform_fma; /* 1: ext applied as the k-ordered FMA chain that OpenBLAS dgemm
                              produces for XPC:119 (bit-exact vs numpy on x86-64 FMA hosts);
                              0: ((r0*x + r1*y) + r2*z) + t with every op rounded           */
} pcp_params;

/* Reference defaults: T_DEPTH_TO_COLOR XPC:17-22, BASE_TO_EXT_CAM S1:13-18,
 * WORK_SPACE DP3:29-33, depth range XPC:109, z thresholds XPC:70-71,75.
 * Intrinsics are left zero (they come from the bag, XPC:37-56). */
void pcp_default_params(pcp_params* p);

const char* pcp_strerror(int status);
/* Library version string, e.g. "pcp-hip 0.2 gfx950". */
const char* pcp_version(void);
/* Number of visible HIP devices (0 on a CPU-only host; never fails). */
int pcp_device_count(void);

/* ------------------------------------------------------------------------- *
 * Stage 1 — replaces deproject XPC:58-64, align_colors XPC:66-78, the validity
 * mask XPC:109, the boolean compaction XPC:110 and the rigid transform XPC:119;
 * with p->apply_crop also the workspace crop DP3:35-42.
 * ------------------------------------------------------------------------- */

/*
 * Dense (un-compacted) per-pixel outputs for ONE batch of B frames; the direct
 * counterpart of calling deproject() and align_colors() and evaluating
 * XPC:109.  Any output pointer may be NULL.
 *   depth     (B,H,W)      f32 metres
 *   rgb       (B,Hc,Wc,3)  u8, RGB order (may be NULL when colors/proj_mask/valid are NULL)
 *   pts_cam   (B,H*W,3)    f64   deproject()                         XPC:58-64
 *   colors    (B,H*W,3)    f32   align_colors()[0], zero where !proj XPC:76-77
 *   proj_mask (B,H*W)      u8    align_colors()[1]                   XPC:74-75
 *   valid     (B,H*W)      u8    proj & depth>z_min & depth<z_max    XPC:109
 *   pts_base  (B,H*W,3)    f64   ext @ [p,1] for every pixel         XPC:119
 *   crop_mask (B,H*W)      u8    valid & inside workspace box        DP3:35-39
 */
int pcp_backproject_dense(const float* depth, const uint8_t* rgb,
                          int B, int H, int W, int Hc, int Wc, const pcp_params* p,
                          double* pts_cam, float* colors, uint8_t* proj_mask, uint8_t* valid,
                          double* pts_base, uint8_t* crop_mask, void* stream);

/*
 * align_colors on free-standing camera-frame points — the reference function's own signature,
 * XPC:66-78 (uses p->color, p->T_d2c, p->zc_min, p->zc_clamp only).
 *   pts (n,3) f64, rgb (Hc,Wc,3) u8 → colors (n,3) f32 (zero where masked), mask (n) u8
 */
int pcp_align_colors(const double* pts, int64_t n, const uint8_t* rgb, int Hc, int Wc,
                     const pcp_params* p, float* colors, uint8_t* mask, void* stream);

size_t pcp_backproject_workspace_bytes(int B, int H, int W);

/*
 * Raw 16-bit depth in millimetres → metres, the conversion the extraction step applies before it
 * stores depth_%05d.npy (extract_topics_one_camera.py:68-72: astype(float32) / 1000.0, one f32
 * division).  Lets a caller upload half the bytes and keep the rest of the path unchanged.
 *   depth_mm (n) u16 → depth_m (n) f32
 */
int pcp_depth_mm_to_m(const uint16_t* depth_mm, int64_t n, float* depth_m, void* stream);

/*
 * Fused backproject + colour-align + validity (+ crop) + order-preserving
 * compaction for B frames.  Surviving points keep row-major pixel order
 * (numpy boolean-mask order, XPC:110 / DP3:41), frames are concatenated CSR
 * style: frame b owns rows [offsets[b], offsets[b+1]).
 *   xyz_cam64 (cap,3) f64  camera-frame points   (pts_v,  XPC:110)   or NULL
 *   xyz64     (cap,3) f64  base-frame points     (pts_g,  XPC:119)   or NULL
 *   xyz32     (cap,3) f32  float32(xyz64), the FPS input (DP3:10)    or NULL
 *   rgb32     (cap,3) f32  colours in [0,1]      (clrs_v, XPC:110)   or NULL
 *   pix       (cap)   i32  source pixel index v*W+u                  or NULL
 *   offsets   (B+1)   i64  CSR row offsets (written)
 * `cap` is the row capacity of the output arrays.  If the batch produces more
 * rows than `cap`, rows beyond cap are NOT written, offsets still hold the true
 * counts, and `overflow` (device i32, may be NULL) is set to 1.
 */
int pcp_backproject_align_crop(const float* depth, const uint8_t* rgb,
                               int B, int H, int W, int Hc, int Wc, const pcp_params* p,
                               double* xyz_cam64, double* xyz64, float* xyz32, float* rgb32,
                               int32_t* pix, int64_t* offsets, int64_t cap, int32_t* overflow,
                               void* ws, size_t ws_bytes, void* stream);

/*
 * Multi-camera fusion (BASELINE config 5; the reference itself is single-camera,
 * its frame loop XPC:96-125 run once per camera): F frames x C cameras, camera c
 * with its own intrinsics/extrinsics cams[c] (host array of C <= PCP_MAX_CAMERAS
 * structs).  Images are laid out (F,C,H,W) / (F,C,Hc,Wc,3); image f*C+c owns rows
 * [offsets[f*C+c], offsets[f*C+c+1]), so the fused cloud of frame f — cameras
 * concatenated in camera order, pixel order inside a camera — is the contiguous
 * row range [fused_offsets[f], fused_offsets[f+1]) = [offsets[f*C], offsets[(f+1)*C]).
 *   offsets (F*C+1) i64, fused_offsets (F+1) i64 or NULL; everything else as above.
 * Workspace: pcp_backproject_workspace_bytes(F*C, H, W).
 */
#define PCP_MAX_CAMERAS 16
int pcp_backproject_align_crop_multicam(const float* depth, const uint8_t* rgb,
                                        int F, int C, int H, int W, int Hc, int Wc,
                                        const pcp_params* cams,
                                        double* xyz_cam64, double* xyz64, float* xyz32, float* rgb32,
                                        int32_t* pix, int64_t* offsets, int64_t* fused_offsets,
                                        int64_t cap, int32_t* overflow,
                                        void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Stage 2 — replaces process_point_cloud DP3:23-52.
 * ------------------------------------------------------------------------- */

/*
 * Workspace crop on already-transformed points (the PLY-input drop-in path):
 * six strict f64 compares, DP3:35-39.
 *   xyz   (M1,3) f64       mask (M1) u8 written
 */
int pcp_crop_mask(const double* xyz, int64_t M1, const double lo[3], const double hi[3],
                  uint8_t* mask, void* stream);

/*
 * Order-preserving crop + compaction of one cloud, DP3:35-42 (+ the .float()
 * cast of DP3:10 for xyz32).
 *   xyz (M1,3) f64, rgb (M1,3) f64 (may be NULL)
 *   out_xyz64 (cap,3) f64 or NULL, out_xyz32 (cap,3) f32 or NULL, out_rgb64 (cap,3) f64 or NULL
 *   count (1) i64 written: number of surviving rows (true count even if > cap)
 */
size_t pcp_crop_workspace_bytes(int64_t M1);
int pcp_crop_compact(const double* xyz, const double* rgb, int64_t M1,
                     const double lo[3], const double hi[3],
                     double* out_xyz64, float* out_xyz32, double* out_rgb64,
                     int64_t cap, int64_t* count, void* ws, size_t ws_bytes, void* stream);

/*
 * The same crop for B clouds at once, rows concatenated CSR style (cloud b owns rows
 * [offsets[b], offsets[b+1]) of xyz / rgb, total = offsets[B] known to the host): one launch
 * sequence for a whole directory chunk of the stage-2 loop DP3:68-90.  out_offsets (B+1) i64
 * receives the CSR offsets of the cropped clouds; row order is preserved inside every cloud.
 * Workspace: pcp_crop_workspace_bytes(total).
 */
int pcp_crop_compact_batched(const double* xyz, const double* rgb, const int64_t* offsets, int B,
                             int64_t total, const double lo[3], const double hi[3],
                             double* out_xyz64, float* out_xyz32, double* out_rgb64,
                             int64_t* out_offsets, int64_t cap,
                             void* ws, size_t ws_bytes, void* stream);

/*
 * Batched farthest-point sampling — replaces farthest_point_sampling DP3:7-21,
 * i.e. pytorch3d.ops.sample_farthest_points(points=(1,M,3) f32, K=[K]) with
 * random_start_point=False (contract: SURVEY.md §8c).
 *   xyz     (total,3) f32   concatenated clouds
 *   offsets (B+1)     i64   CSR: cloud b = rows [offsets[b], offsets[b+1])
 *   idx     (B,K)     i64   written; index RELATIVE to the cloud's first row; -1 beyond M
 *   out_xyz (B,K,3)   f32   written; xyz[idx], 0 where idx = -1; may be NULL
 *   dist_mode 0: ((dx*dx)+(dy*dy))+(dz*dz), every op rounded to f32 (x86 CPU build of pytorch3d)
 *             1: fmaf(dz,dz,fmaf(dy,dy,dx*dx))   (what nvcc's default contraction emits)
 *   algo      0: auto   1: brute force (every point every round)
 *             2: exact bucket-pruned FPS (same indices, far fewer distance evaluations); its first
 *                rounds defer bucket updates when the batch is large enough for that to pay
 *             3: as 2 with the deferred rounds always on   4: as 2 with them always off
 *             5: as 2 with MULTI-PICK rounds (one workgroup round certifies up to four consecutive picks)
 *             6: as 2 with one pick per round, whatever the batch size
 *             7 / 8: as 5 with the deferred rounds always off / on   9: as 5 with exact-key deferred updates inside the
 *                multi-pick rounds whatever the batch size
 *            10: as 2 as a QUANTUM-SLICED launch whatever the batch size: one resident workgroup per CU, clouds take
 *                turns on them a stretch of time at a time.  Algo 0 and 2 launch that way by themselves for batches of a
 *                little more than one cloud per CU (257-336 clouds on 256 CUs, the 300-frame episode: 2.40 instead of
 *                2.83 ms; up to 512 clouds when max_pts is beyond 131,072; 513-624 clouds as two such launches over the
 *                halves of the batch); pcp_fps_workspace_bytes then asks for 72 KB more per cloud (per cloud of the larger
 *                half), where waiting clouds park their round state.  3-9 never launch that way.
 *                A batch of up to 336 clouds (to 512 when max_pts > 131,072; algo 0 / 2) whose sizes fall on both sides of 65,536 or 131,072 points runs as ONE
 *                launch that chooses the bucket split per cloud, instead of one launch per size class after the other; for
 *                2-256 clouds pcp_fps_workspace_bytes adds a few KB of control words for it when max_pts > 65,536 (a
 *                workspace without them is accepted: the size classes then run one after the other, same picks).
 *                (same indices in every case; 3-10 are for tests and tuning)
 * `total_pts` is the row count of xyz (and of the workspace).  `max_pts` is an upper bound on the
 * largest cloud (host-known, e.g. H*W): it decides which size-bucketed launch shapes run, never
 * the picks.
 * Ties are broken toward the lowest index; idx[b][0] = 0.  Empty clouds give
 * an all -1 row (the reference raises inside pytorch3d; the Python wrapper
 * raises too).
 * `overflow` (device i32, may be NULL) is cleared by the call and set to 1 by the kernels when a
 * cloud could not be sampled: more points than `max_pts` promised (beyond the shapes that were
 * launched), or rows outside [0, total_pts) — the state pcp_backproject_align_crop leaves behind
 * when ITS overflow flag is set (true counts in offsets, rows beyond cap unwritten).  Such a cloud
 * gets an all -1 / zero row; nothing is read or written out of bounds and no output row is left
 * unwritten.  The offsets live on the device, so the host cannot know earlier: check the flag
 * wherever the results are consumed.
 */
size_t pcp_fps_workspace_bytes(int B, int64_t total_pts, int64_t max_pts, int K, int algo);
int pcp_fps_batched(const float* xyz, const int64_t* offsets, int B, int64_t total_pts,
                    int64_t max_pts, int K, int dist_mode, int algo,
                    int64_t* idx, float* out_xyz, int32_t* overflow,
                    void* ws, size_t ws_bytes, void* stream);

/*
 * pcp_fps_batched without a workspace argument and WITHOUT ANY ALLOCATION.  It has no counterpart in the
 * reference: it exists for callers that cannot allocate device memory — code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  The sorted-record / min-distance workspace is a static
 * device array inside the library (32 MiB per GPU the library runs on, held until the process ends), so
 *   total_pts <= PCP_FPS_STATIC_MAX_PTS, else PCP_ERR_CAPACITY (use pcp_fps_batched with your own workspace).
 * Inputs, outputs, `overflow`, results and every other rule are those of pcp_fps_batched; the call itself makes
 * no hipMalloc / hipFree / allocating library call.  All calls on one GPU share that GPU's array: calls that
 * overlap on the same GPU (different streams or threads) race — serialise them.  Nothing has to be cleared
 * between calls (a launch rewrites every row it reads).  Quantum-sliced launches (algo 10, or 0 / 2 at 257-336
 * clouds) are not made here — their parked round state would not fit the array — so such a batch runs with two
 * resident workgroups per CU instead, same results (and one size class after the other when it mixes them); a mixed-size
 * batch of 2-256 clouds does run as one any-shape launch, as in pcp_fps_batched: its 3.8 KB of control words sit in the
 * array behind the records and are rewritten by every call.
 * The FIRST call on a device resolves the array's device address (hipGetSymbolAddress, which may also make the
 * runtime load the code object that holds the 32 MiB array): make that call — or pcp_fps_static_prepare(), which does
 * nothing else — once per device OUTSIDE graph capture, as a warm-up; every later call only enqueues kernels.
 * A recorded graph that contains this call must not be replayed while another static call (eager or replayed) runs on
 * the same GPU: they share the array.
 */
#define PCP_FPS_STATIC_MAX_PTS 1600000
int pcp_fps_static_prepare(void);   /* current device; PCP_OK or PCP_ERR_LAUNCH / PCP_ERR_NO_DEVICE */
int pcp_fps_batched_static(const float* xyz, const int64_t* offsets, int B, int64_t total_pts,
                           int64_t max_pts, int K, int dist_mode, int algo,
                           int64_t* idx, float* out_xyz, int32_t* overflow, void* stream);

/*
 * Colour gather pc_rgb[idx], DP3:48.
 *   rgb (total_pts,3) f32, offsets (B+1) i64, idx (B,K) i64, out (B,K,3) f32
 *   compat_neg_index 1: idx = -1 selects the cloud's LAST row (numpy negative
 *                       indexing, exactly what DP3:48 does with pytorch3d's padding)
 *                    0: idx = -1 yields zeros
 * An index outside its cloud, or a row outside [0, total_pts) (a cloud that overflowed stage 1's
 * capacity), yields zeros instead of an out-of-bounds read.
 */
int pcp_gather_rgb(const float* rgb, const int64_t* offsets, int64_t total_pts, const int64_t* idx,
                   int B, int K, int compat_neg_index, float* out, void* stream);

/*
 * The same gather with the colour lookup fused in, for clouds made by pcp_backproject_align_crop: the colours of
 * the K sampled points are evaluated from each point's SOURCE PIXEL (`pix` as written by that call; deproject +
 * align_colors, XPC:58-78, for those points only) instead of being read from a (total,3) colour array — so stage 1
 * need not compute, store or gather a colour for the other M-K points of every cloud (pass rgb32 = NULL there).
 * Values are bit-identical to pcp_gather_rgb over rgb32.  depth / rgb / p are the stage-1 inputs of the same B
 * frames (single camera); pix (total_pts) i32; everything else as pcp_gather_rgb.
 */
int pcp_gather_rgb_pixels(const float* depth, const uint8_t* rgb, int B, int H, int W, int Hc, int Wc,
                          const pcp_params* p, const int32_t* pix, const int64_t* offsets, int64_t total_pts,
                          const int64_t* idx, int K, int compat_neg_index, float* out, void* stream);

/*
 * WHERE the colour of every sampled point sits instead of the colour itself: the colour-image pixel
 * v_idx * Wc + u_idx that align_colors looks up (XPC:73, 77), or -1 where it leaves the colour zero (the point
 * is outside the colour image or too close, XPC:74-76) or the index does not exist.  The projection mask needs
 * rgb.shape only (XPC:72-75) and colours are wanted for K points per frame (DP3:48), so a caller that keeps the
 * decoded rgb images in HOST memory uploads depth alone (0.6 MB instead of 1.5 MB per 640x480 frame with u16
 * depth; pass rgb = NULL and rgb32 = NULL to pcp_backproject_align_crop) and finishes on the host:
 *     colour[b][k] = cpix[b][k] < 0 ? 0 : rgb_b[cpix[b][k]] / 255   (XPC:77; * 255 again for the CHK:141 rows)
 * Arguments as pcp_gather_rgb_pixels (idx = -1 follows compat_neg_index the same way); cpix (B,K) i32.
 */
int pcp_sampled_color_pixels(const float* depth, int B, int H, int W, int Hc, int Wc, const pcp_params* p,
                             const int32_t* pix, const int64_t* offsets, int64_t total_pts,
                             const int64_t* idx, int K, int compat_neg_index, int32_t* cpix, void* stream);

/*
 * (K,6) rows exactly as chunking_data.py:134-144 stacks them for the zarr
 * `data/point_cloud` array (CHK:195): [x, y, z, r*255, g*255, b*255] as f32.
 *   xyz (B,K,3) f32, rgb (B,K,3) f32 in [0,1] → out (B,K,6) f32
 * Colours pass through the PLY uchar quantisation the file round trip applies
 * (round(clamp(c,0,1)*255)) so the values equal what CHK reads.
 */
int pcp_pack_dp3(const float* xyz, const float* rgb, int B, int K, float* out, void* stream);

/* ------------------------------------------------------------------------- *
 * Config-3 filters (BASELINE configs[2]).  The reference tree holds no voxel or
 * outlier filter (north_star's filter_point_clouds.py is absent, SURVEY.md §0
 * item 3); the semantics are Open3D's — the library the reference's stack calls
 * for point-cloud work (DP3:54-58, CHK:135-142).  Parity against Open3D is
 * UNPINNED; the GPU path is bit-exact against oracle/.
 * Both take B clouds in CSR form (rows [offsets[b], offsets[b+1]) of xyz / rgb,
 * offsets on the device) and write B compacted clouds in the same form.
 * `cap_pts` is a host-known upper bound of offsets[B] (the workspace and the
 * sort are sized by it; at most 2^32-1), `cap_out` the row capacity of the
 * outputs; *overflow (device i32, may be NULL) is set when rows were dropped or a
 * grid index left [0,65535] on an axis (results are then unspecified).
 * `max_extent` (metres) is an optional promise that lets the grid sort run on
 * 32-bit keys: no cloud's bounding box is longer than this along any axis (for
 * workspace-cropped clouds: the longest workspace edge).  0 = unknown — always
 * valid, the sort then covers the full 16 bits per axis.  A cloud that breaks the
 * promise sets *overflow (results unspecified), it is never silently mis-sorted.
 * ------------------------------------------------------------------------- */

/*
 * Open3D PointCloud::VoxelDownSample(voxel_size): per cloud, origin = min bound -
 * voxel_size/2, voxel index = floor((p - origin)/voxel_size); every occupied voxel
 * yields the mean point and mean colour of its members (FP64 sums in ascending
 * row order, one division by the count).  Voxels are emitted in ascending
 * (ix,iy,iz) order (Open3D's own order is unordered_map order, i.e. unspecified).
 *   xyz (cap_pts,3) f64, rgb (cap_pts,3) f32 or NULL
 *   out_xyz64 / out_xyz32 / out_rgb32 (cap_out,3), out_count (cap_out) i32: any may be NULL
 *   out_offsets (B+1) i64
 */
size_t pcp_voxel_workspace_bytes(int B, int64_t cap_pts);
int pcp_voxel_downsample(const double* xyz, const float* rgb, const int64_t* offsets, int B,
                         int64_t cap_pts, double voxel_size, double max_extent,
                         double* out_xyz64, float* out_xyz32, float* out_rgb32, int32_t* out_count,
                         int64_t* out_offsets, int64_t cap_out, int32_t* overflow,
                         void* ws, size_t ws_bytes, void* stream);

/*
 * Open3D PointCloud::RemoveStatisticalOutliers(nb_neighbors, std_ratio): per point
 * the mean Euclidean distance to its min(nb_neighbors, M) nearest neighbours
 * (itself included, FP64, ascending summation); cloud mean and sample standard
 * deviation accumulated sequentially in row order; rows with
 * 0 < avg < mean + std_ratio*std survive, order preserved.  1 <= nb_neighbors <= 32.
 * `cell_size` (metres) is the edge of the uniform search grid; it only affects
 * speed (a cell about the radius that holds nb_neighbors points is a good choice).
 *   out_index (cap_out) i64: cloud-local source row of every survivor, or NULL
 *   avg_dist  (cap_pts) f64: the per-point mean distances, or NULL
 */
size_t pcp_sor_workspace_bytes(int B, int64_t cap_pts);
int pcp_statistical_outlier_removal(const double* xyz, const float* rgb, const int64_t* offsets, int B,
                                    int64_t cap_pts, int nb_neighbors, double std_ratio, double cell_size,
                                    double max_extent,
                                    double* out_xyz64, float* out_xyz32, float* out_rgb32,
                                    int64_t* out_index, int64_t* out_offsets, double* avg_dist,
                                    int64_t cap_out, int32_t* overflow,
                                    void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCP_H */
