// microbenchmark: latency of LDS access shapes used by the LZ4 kernel (one wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
#define SQY_LDS __attribute__((address_space(3)))
struct __attribute__((packed)) pk_u128 { uint32_t x, y, z, w; };
struct __attribute__((packed)) pk_u64 { uint64_t v; };
struct __attribute__((packed)) pk_u32 { uint32_t v; };
__global__ void k(unsigned long long* out, uint32_t* sink, int stride, int base)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[16384];
    int lane = threadIdx.x;
    for (int i = lane; i < 16384; i += 64) buf[i] = (uint8_t)(i * 7);
    __syncthreads();
    SQY_LDS uint8_t* b = (SQY_LDS uint8_t*)buf;
    uint32_t acc = 0;
    unsigned long long t[8];
    uint32_t off = base + lane * stride;
#define STAMP(i) __builtin_amdgcn_sched_barrier(0); t[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0);
    STAMP(0)
    for (int r = 0; r < 16; ++r) { const SQY_LDS pk_u128* q = (const SQY_LDS pk_u128*)(b + ((off + acc) & 8191)); acc += q->x ^ q->y ^ q->z ^ q->w; acc &= 15; }
    STAMP(1)
    for (int r = 0; r < 16; ++r) { const SQY_LDS pk_u64* q = (const SQY_LDS pk_u64*)(b + ((off + acc) & 8191)); acc += (uint32_t)q->v; acc &= 15; }
    STAMP(2)
    for (int r = 0; r < 16; ++r) { const SQY_LDS pk_u32* q = (const SQY_LDS pk_u32*)(b + ((off + acc) & 8191)); acc += q->v; acc &= 15; }
    STAMP(3)
    for (int r = 0; r < 16; ++r) { acc += b[(off + acc) & 8191]; acc &= 15; }
    STAMP(4)
    for (int r = 0; r < 16; ++r) { b[8192 + ((off + acc + r) & 4095)] = (uint8_t)acc; acc = (acc + b[8192 + ((off + r) & 4095)]) & 15; }
    STAMP(5)
    if (lane == 0) for (int i = 0; i < 5; ++i) out[i] = (t[i + 1] - t[i]) / 16;
    sink[lane] = acc;
}
int main()
{
    unsigned long long* d; uint32_t* s; hipMalloc(&d, 64); hipMalloc(&s, 256);
    int cfgs[][2] = {{16, 0}, {16, 3}, {1, 0}, {1, 5}, {4, 0}, {8, 1}, {64, 0}, {3, 0}};
    for (auto& c : cfgs) {
        k<<<1, 64>>>(d, s, c[0], c[1]); hipDeviceSynchronize();
        k<<<1, 64>>>(d, s, c[0], c[1]); hipDeviceSynchronize();
        unsigned long long h[5]; hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
        printf("stride %2d base %d : dependent-chain cycles per op  b128 %4llu  b64 %4llu  b32 %4llu  u8 %4llu  st8+ld8 %4llu\n", c[0], c[1], h[0], h[1], h[2], h[3], h[4]);
    }
    return 0;
}
