/* tools/slabs_c_test.c -- a plain C caller of the throughput entry point (built into sqeazy_amd/bin/slabs_c_test):
 *   slabs_c_test <z> <y> <x> <nslabs> [pipeline] [inflight]
 * fills a z*y*x uint16 volume with the bench's synthetic stack (sqeazy_amd/synth.py, same integers) on the host cores, uploads it,
 * encodes it with ONE call of SQYAMD_PipelineEncode_Slabs_UI16_Device (nslabs z-slab blobs, `inflight` slab calls at a time on
 * library-owned streams), then encodes every slab on its own with SQYAMD_PipelineEncode_UI16_DeviceAt and compares the blobs byte
 * for byte.  Prints the rate of both (input bytes / wall time, inputs resident in HBM).  Exit code 0 = every blob equal.
 * The reference encodes one volume of < 2^31 voxels per call (src/cpp/src/sqeazy.cpp:108-142); 2048^3 is 8 such calls. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../include/sqeazy_amd.h"

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int64_t floordiv(int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv)
{
    if (argc < 5) { fprintf(stderr, "usage: slabs_c_test z y x nslabs [pipeline] [inflight]\n"); return 2; }
    /* the HIP runtime deals streams to four hardware queues unless told otherwise, and kernels behind the same queue never overlap:
     * eight queues for the slab calls in flight (read once, at the runtime's first call -- so first thing here) */
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const long Z = atol(argv[1]), Y = atol(argv[2]), X = atol(argv[3]);
    const int nslabs = atoi(argv[4]);
    const char* pipeline = argc > 5 ? argv[5] : "bitswap1->lz4";
    const int inflight = argc > 6 ? atoi(argv[6]) : 3;
    const size_t nvox = (size_t)Z * Y * X, nbytes = nvox * 2;
    uint16_t* vol = (uint16_t*)malloc(nbytes);
    if (!vol) return 2;
    const int64_t ax = (X / 4 > 1 ? X / 4 : 1) * (X / 4 > 1 ? X / 4 : 1), ay = (Y / 4 > 1 ? Y / 4 : 1) * (Y / 4 > 1 ? Y / 4 : 1);
    const int64_t az0 = (6 * Z) / 10 > 1 ? (6 * Z) / 10 : 1, az = az0 * az0;
#pragma omp parallel for schedule(static)
    for (long z = 0; z < Z; ++z) {
        const int64_t dz = 2 * z - Z;
        for (long y = 0; y < Y; ++y) {
            const int64_t dy = 2 * y - Y;
            for (long x = 0; x < X; ++x) {
                const uint64_t i = ((uint64_t)z * Y + y) * X + x;
                const uint64_t r = splitmix64(0x5EA2ull ^ i);
                const uint32_t noise = (uint32_t)(r & 0xff) + (uint32_t)((r >> 8) & 0xff) + (uint32_t)((r >> 16) & 0xff) + (uint32_t)((r >> 24) & 0xff);
                const int64_t dx = 2 * x - X;
                const int64_t q = floordiv(64 * dx * dx, ax) + floordiv(64 * dy * dy, ay) + floordiv(64 * dz * dz, az);
                const int shell = (q - 64 < 0 ? 64 - q : q - 64) < 5;
                vol[i] = (uint16_t)(100 + (noise >> 2) + shell * 6000);
            }
        }
    }
    /* slab capacity: the bound of the largest slab */
    long shape[3] = {Z, Y, X}, slab_shape[3] = {Z / nslabs + (Z % nslabs ? 1 : 0), Y, X};
    long cap = (long)strlen(pipeline);
    if (SQY_Pipeline_Max_Compressed_Length_3D_UI16(pipeline, slab_shape, 3, &cap)) { fprintf(stderr, "pipeline refused\n"); return 2; }
    cap = (cap + 255) & ~255l;
    void *d_src = NULL, *d_dst = NULL, *d_one = NULL;
    if (hipMalloc(&d_src, nbytes) != hipSuccess || hipMalloc(&d_dst, (size_t)cap * nslabs) != hipSuccess || hipMalloc(&d_one, (size_t)cap) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 2;
    }
    if (hipMemcpy(d_src, vol, nbytes, hipMemcpyHostToDevice) != hipSuccess) return 2;
    long* offs = (long*)calloc(nslabs, sizeof(long)), *lens = (long*)calloc(nslabs, sizeof(long));
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {                      /* (the first pass lets every context allocate its workspace) */
        const double t0 = now();
        const int rc = SQYAMD_PipelineEncode_Slabs_UI16_Device(pipeline, d_src, shape, 3, nslabs, d_dst, cap, offs, lens, 0, inflight);
        const double dt = now() - t0;
        if (rc) { fprintf(stderr, "SQYAMD_PipelineEncode_Slabs_UI16_Device returned %d\n", rc); return 3; }
        if (rep && dt < best) best = dt;
    }
    printf("one call, %d slabs, %d in flight: %.3f ms = %.1f GB/s of input voxels\n", nslabs, inflight, best * 1e3, nbytes / best / 1e9);
    /* every slab on its own, one call at a time */
    int bad = 0;
    double t_single = 0;
    long total_out = 0;
    const long base = Z / nslabs, rem = Z % nslabs;
    for (int i = 0; i < nslabs; ++i) {
        const long z0 = i * base + (i < rem ? i : rem), nz = base + (i < rem ? 1 : 0);
        long shp[3] = {nz, Y, X}, at = 0, len = 0;
        const double t0 = now();
        const int rc = SQYAMD_PipelineEncode_UI16_DeviceAt(pipeline, (const char*)d_src + (size_t)z0 * Y * X * 2, shp, 3, d_one, cap, &at, &len, 0, NULL);
        t_single += now() - t0;
        if (rc) { fprintf(stderr, "single call %d returned %d\n", i, rc); return 3; }
        char* a = (char*)malloc((size_t)len), *b = (char*)malloc((size_t)lens[i]);
        if (hipMemcpy(a, (char*)d_one + at, (size_t)len, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(b, (char*)d_dst + offs[i], (size_t)lens[i], hipMemcpyDeviceToHost) != hipSuccess) return 2;
        const int same = len == lens[i] && memcmp(a, b, (size_t)len) == 0;
        if (!same) { fprintf(stderr, "slab %d: blob of the slabs call differs from the single call (%ld vs %ld bytes)\n", i, lens[i], len); bad = 1; }
        total_out += len;
        free(a); free(b);
    }
    printf("%d single calls one after the other: %.3f ms = %.1f GB/s; %ld blob bytes; blobs %s\n", nslabs, t_single * 1e3, nbytes / t_single / 1e9,
           total_out, bad ? "DIFFERENT" : "equal");
    return bad;
}
