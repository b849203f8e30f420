// tools/bsw_bench.hip -- the in-place bit-plane transposer alone on 1 GiB of noise-like voxels, with the frame gaps the product uses (15 bytes:
// every 1 KiB piece of the plane stream straddles two 128-byte lines it shares with its neighbours) and with gaps that keep the pieces
// line-aligned (-DSQY_EXP_GAP_BYTES=128: wrong layout, right timing).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w [-DSQY_EXP_GAP_BYTES=128] -I sqeazy_amd/csrc tools/bsw_bench.hip -o tools/bsw_bench_15
#include "../sqeazy_amd/csrc/sqy_kernels.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(uint16_t* p, uint64_t n, uint32_t hi_planes)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = (uint16_t)(x & ((1u << hi_planes) - 1u));
    }
}
int main(int argc, char** argv)
{
    const uint64_t n = 512ull << 20;                    // voxels (1 GiB)
    const uint32_t planes = argc > 1 ? std::atoi(argv[1]) : 10;     // bit planes that hold data (the others are holes: not written)
    uint16_t* in; uint8_t* out; uint32_t* ph;
    CK(hipMalloc(&in, n * 2)); CK(hipMalloc(&out, n * 2 + (n * 2 / 262144 + 2) * 256 + 4096)); CK(hipMalloc(&ph, (n / 8192) * 16 * 4 * 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, in, n, planes);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t n_tiles = n / 8192;
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((sqy::bitswap1_u16_regs<true>), dim3(256 * 32), dim3(128), 0, 0, in, reinterpret_cast<uint16_t*>(out + 16), n_tiles, n / 16, ph, 18u,
                           (const uint16_t*)nullptr, 0u, 0u);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
#ifdef SQY_EXP_GAP_BYTES
    const int gap = SQY_EXP_GAP_BYTES;
#else
    const int gap = 15;
#endif
    std::printf("gap %3d bytes, %2u planes written: %.3f ms = %.2f TB/s (read %.2f GB + written %.2f GB)\n", gap, planes, best,
                (n * 2.0 + n * 2.0 * planes / 16) / best / 1e9, n * 2.0 / 1e9, n * 2.0 * planes / 16 / 1e9);
    return 0;
}
