/*
 * Calling libpcp_hip.so from plain C — no Python, no torch: device memory from the HIP runtime, plain
 * pointers and sizes across the boundary (include/pcp.h).  Reads an (M,3) float32 cloud from a raw file,
 * runs pcp_fps_batched for K points and writes the K int64 indices to another raw file.
 *
 *   gcc -std=c11 -O2 examples/fps_from_c.c -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude \
 *       -Lpoint-cloud-processing_amd/lib -L/opt/rocm/lib -lpcp_hip -lamdhip64 \
 *       -Wl,-rpath,'$ORIGIN/../point-cloud-processing_amd/lib' -Wl,-rpath,/opt/rocm/lib -o examples/fps_from_c
 *   (__graft_entry__.build() runs exactly this)
 *   ./examples/fps_from_c cloud.f32 M K idx.i64
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "pcp.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv)
{
    if (argc != 5) { fprintf(stderr, "usage: %s cloud.f32 M K idx.i64\n", argv[0]); return 1; }
    const int64_t M = atoll(argv[2]);
    const int K = atoi(argv[3]);
    float* h_xyz = (float*)malloc((size_t)M * 3 * sizeof(float));
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(h_xyz, sizeof(float), (size_t)M * 3, f) != (size_t)M * 3) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    fclose(f);
    printf("%s, %d device(s)\n", pcp_version(), pcp_device_count());

    const int64_t h_off[2] = {0, M};
    float *d_xyz, *d_out;
    int64_t *d_off, *d_idx;
    int32_t* d_flag;
    void* d_ws;
    const size_t ws_bytes = pcp_fps_workspace_bytes(1, M, M, K, 0);
    HIP_OK(hipMalloc((void**)&d_xyz, (size_t)M * 12));
    HIP_OK(hipMalloc((void**)&d_off, sizeof h_off));
    HIP_OK(hipMalloc((void**)&d_idx, (size_t)K * 8));
    HIP_OK(hipMalloc((void**)&d_out, (size_t)K * 12));
    HIP_OK(hipMalloc((void**)&d_flag, sizeof(int32_t)));
    HIP_OK(hipMalloc(&d_ws, ws_bytes));                       /* hipMalloc returns >= 256-byte aligned memory */
    HIP_OK(hipMemcpy(d_xyz, h_xyz, (size_t)M * 12, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_off, h_off, sizeof h_off, hipMemcpyHostToDevice));

    const int rc = pcp_fps_batched(d_xyz, d_off, 1, M, M, K, /*dist_mode*/ 0, /*algo*/ 0, d_idx, d_out, d_flag, d_ws,
                                   ws_bytes, /*stream*/ NULL);
    if (rc != PCP_OK) { fprintf(stderr, "pcp_fps_batched: %s\n", pcp_strerror(rc)); return 3; }
    HIP_OK(hipDeviceSynchronize());
    int32_t h_flag = 0;
    HIP_OK(hipMemcpy(&h_flag, d_flag, sizeof h_flag, hipMemcpyDeviceToHost));
    if (h_flag) { fprintf(stderr, "pcp_fps_batched: a cloud exceeded max_pts / total_pts\n"); return 4; }

    int64_t* h_idx = (int64_t*)malloc((size_t)K * 8);
    HIP_OK(hipMemcpy(h_idx, d_idx, (size_t)K * 8, hipMemcpyDeviceToHost));
    f = fopen(argv[4], "wb");
    fwrite(h_idx, 8, (size_t)K, f);
    fclose(f);
    printf("first picks: %lld %lld %lld %lld\n", (long long)h_idx[0], (long long)h_idx[K > 1], (long long)h_idx[K > 2 ? 2 : 0],
           (long long)h_idx[K > 3 ? 3 : 0]);
    hipFree(d_xyz); hipFree(d_off); hipFree(d_idx); hipFree(d_out); hipFree(d_flag); hipFree(d_ws);
    free(h_xyz); free(h_idx);
    return 0;
}
