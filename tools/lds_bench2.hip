// microbenchmark (round 4): dependent-chain latency of single DS read instructions at chosen byte alignments, one wave.
// What the LZ4 lean loop wants to know: which read shapes are replayed when their address is off the natural alignment
// (SQ_LDS_UNALIGNED_STALL was 28 % of the wave-cycles of round 3's lz4_chunks), and what the aligned alternatives cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>

#define STAMP(i) __builtin_amdgcn_sched_barrier(0); t[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0);

// each macro: one dependent step -- read at (base + (acc & 0x30)) + mis, fold the result into acc
#define STEP1(...)                                                                             \
    {                                                                                                   \
        uint32_t a = base + (acc & 0x30u) + mis;                                                        \
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;                                                        \
        __VA_ARGS__                                                                                      \
        acc += r0 ^ r1 ^ r2 ^ r3;                                                                       \
    }

__global__ void k(unsigned long long* out, uint32_t* sink, uint32_t mis, uint32_t lane_stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[16384];
    const int lane = threadIdx.x;
    for (int i = lane; i < 16384; i += 64) buf[i] = (uint8_t)(i * 7 + (i >> 8));
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)buf + 256u + (uint32_t)lane * lane_stride;
    uint32_t acc = 0;
    unsigned long long t[12];
    const int R = 32;
    STAMP(0)
    for (int r = 0; r < R; ++r) STEP1(asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r0) : "v"(a) : "memory");)
    STAMP(1)
    for (int r = 0; r < R; ++r) STEP1({ uint64_t v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); r0 = (uint32_t)v; r1 = (uint32_t)(v >> 32); })
    STAMP(2)
    for (int r = 0; r < R; ++r) STEP1({ typedef uint32_t v4 __attribute__((ext_vector_type(4))); v4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); r0 = v.x; r1 = v.y; r2 = v.z; r3 = v.w; })
    STAMP(3)
    for (int r = 0; r < R; ++r) STEP1({ uint64_t v; asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); r0 = (uint32_t)v; r1 = (uint32_t)(v >> 32); })
    STAMP(4)
    for (int r = 0; r < R; ++r) STEP1({ typedef uint32_t v4 __attribute__((ext_vector_type(4))); v4 v; asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); r0 = v.x; r1 = v.y; r2 = v.z; r3 = v.w; })
    STAMP(5)
    // four dwords as two ds_read2_b32 issued back to back (one wait)
    for (int r = 0; r < R; ++r) STEP1({ uint64_t v, w; asm volatile("ds_read2_b32 %0, %2 offset0:0 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v), "=&v"(w) : "v"(a) : "memory"); r0 = (uint32_t)v; r1 = (uint32_t)(v >> 32); r2 = (uint32_t)w; r3 = (uint32_t)(w >> 32); })
    STAMP(6)
    // 16 bytes as two ds_read_b64 back to back (round 3's lds_ld_u128)
    for (int r = 0; r < R; ++r) STEP1({ uint64_t v, w; asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v), "=&v"(w) : "v"(a) : "memory"); r0 = (uint32_t)v; r1 = (uint32_t)(v >> 32); r2 = (uint32_t)w; r3 = (uint32_t)(w >> 32); })
    STAMP(7)
    // 16 bytes as four ds_read_b32 back to back
    for (int r = 0; r < R; ++r) STEP1(asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a) : "memory");)
    STAMP(8)
    // 32 aligned bytes (two ds_read_b128 at a & ~15) -- the window a register funnel shift would work on
    for (int r = 0; r < R; ++r) STEP1({ typedef uint32_t v4 __attribute__((ext_vector_type(4))); v4 v, w; uint32_t aa = a & ~15u; asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v), "=&v"(w) : "v"(aa) : "memory"); r0 = v.x ^ w.x; r1 = v.y ^ w.y; r2 = v.z ^ w.z; r3 = v.w ^ w.w; })
    STAMP(9)
    for (int r = 0; r < R; ++r) STEP1(asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r0) : "v"(a) : "memory");)
    STAMP(10)
    if (lane == 0) for (int i = 0; i < 10; ++i) out[i] = (t[i + 1] - t[i]) / R;
    sink[lane] = acc;
}

// correctness of byte-misaligned reads: what does ds_read_b32 / b64 / b128 return at address a + mis?
__global__ void check(uint32_t* out, uint32_t mis)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) buf[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)buf + 16u + mis;
    uint32_t r0; uint64_t r1; typedef uint32_t v4 __attribute__((ext_vector_type(4))); v4 r2;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r0) : "v"(a) : "memory");
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r1) : "v"(a) : "memory");
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r2) : "v"(a) : "memory");
    if (lane == 0) { out[0] = r0; out[1] = (uint32_t)r1; out[2] = (uint32_t)(r1 >> 32); out[3] = r2.x; out[4] = r2.y; out[5] = r2.z; out[6] = r2.w; }
}

int main()
{
    unsigned long long* d; uint32_t* s; hipMalloc(&d, 128); hipMalloc(&s, 256);
    printf("dependent-chain cycles per step (one wave; includes ~10 cycles of address arithmetic)\n");
    printf("%-22s %6s %6s %6s %8s %8s %9s %8s %8s %9s %5s\n", "mis / lane stride", "b32", "b64", "b128", "rd2_b32", "rd2_b64", "2xrd2_b32", "2xb64", "4xb32", "2xb128al", "u8");
    const uint32_t strides[] = {0, 1, 16};
    for (uint32_t st : strides)
        for (uint32_t mis = 0; mis < 16; ++mis) {
            if (!(mis <= 5 || mis == 8 || mis == 12)) continue;
            k<<<1, 64>>>(d, s, mis, st); hipDeviceSynchronize();
            k<<<1, 64>>>(d, s, mis, st); hipDeviceSynchronize();
            unsigned long long h[10]; hipMemcpy(h, d, 80, hipMemcpyDeviceToHost);
            printf("mis %2u stride %2u       %6llu %6llu %6llu %8llu %8llu %9llu %8llu %8llu %9llu %5llu\n", mis, st, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
        }
    uint32_t* c; hipMalloc(&c, 64);
    for (uint32_t mis = 0; mis < 8; ++mis) {
        check<<<1, 64>>>(c, mis); hipDeviceSynchronize();
        uint32_t h[7]; hipMemcpy(h, c, 28, hipMemcpyDeviceToHost);
        printf("check mis %u: b32 %08x  b64 %08x %08x  b128 %08x %08x %08x %08x\n", mis, h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    }
    return 0;
}
