// tools/read_bench.hip -- what a streaming READ of 1 GiB reaches on this chip, in the shapes the bit-plane transposer could use.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/read_bench.hip -o tools/read_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
// A: coalesced, 16 B per lane, 1 KiB per wave instruction, UNROLL loads in flight per lane; tile = 16 KiB per wave, grid-stride over tiles
template <int MODE, int NT>
__global__ __launch_bounds__(128) void rd(const v4u* __restrict__ in, uint64_t n_tiles, uint32_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), stride = (uint64_t)gridDim.x * (blockDim.x >> 6);
    uint32_t acc = 0;
    for (uint64_t t = wave; t < n_tiles; t += stride) {
        const v4u* p = in + t * 1024;                      // 16 KiB tile = 1024 vectors
        v4u v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) v[j] = NT ? __builtin_nontemporal_load(p + j * 64 + lane) : p[j * 64 + lane];          // coalesced: 1 KiB per instruction
            else if (MODE == 1) v[j] = NT ? __builtin_nontemporal_load(p + lane * 16 + j) : p[lane * 16 + j];     // the transposer's: a lane's own 256 bytes
            else { const v4u* q = p + (lane >> 4) * 256 + j * 16 + (lane & 15); v[j] = NT ? __builtin_nontemporal_load(q) : *q; }   // 16 lanes share 256 bytes, four such segments per instruction
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const uint64_t bytes = 1ull << 30;
    v4u* in; uint32_t* out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, 64)); CK(hipMemset(in, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t n_tiles = bytes / 16384;
    for (int blocks_per_cu : {4, 32}) {
        for (int v = 0; v < 6; ++v) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0, 0));
                const dim3 g(256 * blocks_per_cu), b(128);
                if (v == 0) hipLaunchKernelGGL((rd<0, 0>), g, b, 0, 0, in, n_tiles, out);
                if (v == 1) hipLaunchKernelGGL((rd<1, 0>), g, b, 0, 0, in, n_tiles, out);
                if (v == 2) hipLaunchKernelGGL((rd<0, 1>), g, b, 0, 0, in, n_tiles, out);
                if (v == 3) hipLaunchKernelGGL((rd<1, 1>), g, b, 0, 0, in, n_tiles, out);
                if (v == 4) hipLaunchKernelGGL((rd<2, 0>), g, b, 0, 0, in, n_tiles, out);
                if (v == 5) hipLaunchKernelGGL((rd<2, 1>), g, b, 0, 0, in, n_tiles, out);
                CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            std::printf("%2d blocks of 2 waves per CU, %s%s: %.3f ms = %.2f TB/s\n", blocks_per_cu, v >= 4 ? "4 x 256 B per instr   " : (v & 1) ? "lane-owns-256-bytes   " : "coalesced 1 KiB/instr ", (v == 2 || v == 3 || v == 5) ? " nt" : "   ",
                        best, bytes / best / 1e9);
        }
    }
    return 0;
}
