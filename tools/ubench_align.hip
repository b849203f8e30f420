// microbenchmark (GPU box): what do byte-misaligned 16-byte global stores / loads / LDS-DMA fills cost on gfx950?
// Decides how the bit-plane transpose writes chunk bodies that sit 11 + 15 k bytes off a 16-byte boundary (frames in place).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_align.hip -o tools/ubench_align && tools/ubench_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include <vector>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v4u_any __attribute__((ext_vector_type(4), aligned(1)));
#define SQY_LDS __attribute__((address_space(3)))

// every wave copies 1 KiB pieces: 16-byte aligned loads, stores at dst + mis (bytes)
template <bool NT>
__global__ __launch_bounds__(256) void copy_store_mis(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t npieces, uint32_t mis)
{
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 4;
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t p = wave; p + 3 * nw < npieces; p += 4 * nw) {
        v4u v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const v4u*>(src + (p + j * nw) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v4u_any* q = reinterpret_cast<v4u_any*>(dst + (p + j * nw) * 1024 + lane * 16 + mis);
            if (NT) __builtin_nontemporal_store(v[j], q); else *q = v[j];
        }
    }
}

// the register-shift form: aligned stores of {previous lane's tail, own head}; the piece's first 16 - mis ... are left to the
// two edge lanes' own unaligned stores (identical bytes where they overlap an aligned one)
__global__ __launch_bounds__(256) void copy_store_shift(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t npieces, uint32_t mis)
{
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 4;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t dsh = mis >> 2, bsh = mis & 3u;          // uniform
    for (uint64_t p = wave; p + 3 * nw < npieces; p += 4 * nw) {
        v4u v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const v4u*>(src + (p + j * nw) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint8_t* base = dst + (p + j * nw) * 1024;        // aligned; data goes to base + mis + lane * 16
            // eight dwords: previous lane's four, then mine
            uint32_t e[9];
            e[0] = __builtin_amdgcn_update_dpp(0u, v[j].x, 0x138, 0xf, 0xf, false);   // wave_shr:1
            e[1] = __builtin_amdgcn_update_dpp(0u, v[j].y, 0x138, 0xf, 0xf, false);
            e[2] = __builtin_amdgcn_update_dpp(0u, v[j].z, 0x138, 0xf, 0xf, false);
            e[3] = __builtin_amdgcn_update_dpp(0u, v[j].w, 0x138, 0xf, 0xf, false);
            e[4] = v[j].x; e[5] = v[j].y; e[6] = v[j].z; e[7] = v[j].w; e[8] = 0;
            // aligned slot `lane` holds stream bytes [16 lane - mis, 16 lane - mis + 16): dwords starting at e-index 4 - dsh (- 1 when bsh)
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // low = e[4 - dsh - (bsh ? 1 : 0) + i], high = next
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if (dsh == (uint32_t)d) { lo = e[4 - d - 1 + i]; hi = e[4 - d + i]; }
                }
                o[i] = bsh ? __builtin_amdgcn_alignbyte(hi, lo, 4u - bsh) : hi;
            }
            const v4u ov = {o[0], o[1], o[2], o[3]};
            if (lane >= 1) __builtin_nontemporal_store(ov, reinterpret_cast<v4u*>(base + lane * 16));
            if (lane == 0 || lane == 63) *reinterpret_cast<v4u_any*>(base + mis + lane * 16) = v[j];
        }
    }
}

// loads at src + mis, aligned stores
__global__ __launch_bounds__(256) void copy_load_mis(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t npieces, uint32_t mis)
{
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 4;
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t p = wave; p + 3 * nw < npieces; p += 4 * nw) {
        v4u v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const v4u_any*>(src + (p + j * nw) * 1024 + lane * 16 + mis);
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v[j], reinterpret_cast<v4u*>(dst + (p + j * nw) * 1024 + lane * 16));
    }
}

// LDS-DMA from a misaligned global address: does it work, what lands?
__global__ __launch_bounds__(64) void ldsdma_mis(const uint8_t* __restrict__ src, uint8_t* __restrict__ out, uint32_t mis)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[2048];
    const uint32_t lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) ring[i] = 0xEE;
    __syncthreads();
    const uint32_t lds_dst = (uint32_t)(uintptr_t)(SQY_LDS uint8_t*)ring;
    typedef __attribute__((address_space(1))) const uint8_t glb_u8;
    glb_u8* a = (glb_u8*)(src + mis + lane * 16);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(a), "s"(lds_dst) : "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = ring[i];
}

int main()
{
    const uint64_t bytes = 1ull << 30, npieces = bytes / 1024;
    uint8_t *a, *b;
    hipMalloc(&a, bytes + 4096); hipMalloc(&b, bytes + 4096);
    hipMemset(a, 1, bytes + 4096); hipMemset(b, 2, bytes + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 5; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-34s %.3f ms  %.0f GB/s (read+write)\n", name, best, 2.0 * bytes / best / 1e6);
    };
    const int grid = 256 * 8;
    for (uint32_t mis : {0u, 4u, 8u, 1u, 11u, 15u}) {
        char nm[64];
        snprintf(nm, 64, "store mis %2u plain", mis); timeit(nm, [&] { copy_store_mis<false><<<grid, 256>>>(a, b, npieces, mis); });
        snprintf(nm, 64, "store mis %2u nt", mis); timeit(nm, [&] { copy_store_mis<true><<<grid, 256>>>(a, b, npieces, mis); });
        snprintf(nm, 64, "store mis %2u shift+aligned nt", mis); timeit(nm, [&] { copy_store_shift<<<grid, 256>>>(a, b, npieces, mis); });
        snprintf(nm, 64, "load  mis %2u", mis); timeit(nm, [&] { copy_load_mis<<<grid, 256>>>(a, b, npieces, mis); });
    }
    // correctness of the shift form and of the LDS-DMA
    std::vector<uint8_t> h(1 << 20), g(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131 + (i >> 8) * 7);
    hipMemcpy(a, h.data(), h.size(), hipMemcpyHostToDevice);
    for (uint32_t mis : {0u, 1u, 4u, 7u, 11u, 15u}) {
        hipMemset(b, 0xAA, 1 << 20);
        copy_store_shift<<<1, 256>>>(a, b, 64, mis);      // 4 waves x 4 pieces x (p + 3 nw < 64)
        hipMemcpy(g.data(), b, 1 << 20, hipMemcpyDeviceToHost);
        size_t bad = 0, checked = 0;
        for (uint64_t p = 0; p + 12 < 64; ++p) if (true) {
            for (int i = 0; i < 1024; ++i) { ++checked; if (g[p * 1024 + mis + i] != h[p * 1024 + i]) ++bad; }
        }
        printf("shift form mis %2u: %zu / %zu bytes wrong (pieces overlap their successor's head by design when mis > 0: expect errors only there)\n", mis, bad, checked);
    }
    for (uint32_t mis : {0u, 1u, 11u}) {
        ldsdma_mis<<<1, 64>>>(a, b, mis);
        hipMemcpy(g.data(), b, 1024, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int i = 0; i < 1024; ++i) if (g[i] != h[mis + i]) ++bad;
        printf("LDS-DMA from a source %2u bytes off: %zu / 1024 bytes wrong\n", mis, bad);
    }
    return 0;
}
