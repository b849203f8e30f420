// tools/lz4_diag.hip -- diagnostic build of the LZ4 chunk kernel: cycles of one region of the parse loop.
// Not part of the product.  Build one binary per region (marks A -> B, see SQY_STAMP in sqy_kernels.hip):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSQY_DIAG_A=0 -DSQY_DIAG_B=9 -I sqeazy_amd/csrc tools/lz4_diag.hip -o tools/lz4_diag_0_9
// tools/lz4_diag_all.sh builds and runs the usual set.
#define SQY_LZ4_DIAG 1
#include "../sqeazy_amd/csrc/sqy_kernels.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv)
{
    if (argc < 2) { std::printf("usage: lz4_diag file [chunk]\n"); return 1; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> h(n);
    if (std::fread(h.data(), 1, n, f) != (size_t)n) return 1;
    std::fclose(f);
    const uint32_t chunk = argc > 2 ? std::atoi(argv[2]) : 262144;
    const uint64_t nch = (n + chunk - 1) / chunk;
    uint8_t *din, *dscr; uint32_t *dcs, *dredo; unsigned long long* ddg;
    CK(hipMalloc(&din, n + 64)); CK(hipMalloc(&dscr, nch * chunk)); CK(hipMalloc(&dcs, nch * 4)); CK(hipMalloc(&dredo, (nch + 1) * 4)); CK(hipMalloc(&ddg, nch * 32 * 8)); CK(hipMemset(ddg, 0, nch * 32 * 8));
    CK(hipMemcpy(din, h.data(), n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        CK(hipMemset(dredo, 0, 4));
        hipLaunchKernelGGL((sqy::lz4_chunks_kernel<false, false>), dim3((unsigned)nch), dim3(64), 0, 0, din, (uint64_t)n, chunk, (uint64_t)chunk, dscr, (uint64_t)chunk, dcs, (const uint64_t*)nullptr, (uint64_t)0,
                           (const sqy::Lz4Block*)nullptr, (const uint32_t*)nullptr, 0u, dredo, (const uint32_t*)nullptr, 1u, sqy::Lz4DedupeArgs{}, ddg);
        uint32_t nredo = 0; CK(hipMemcpy(&nredo, dredo, 4, hipMemcpyDeviceToHost));
        if (nredo) hipLaunchKernelGGL((sqy::lz4_chunks_kernel<false, true>), dim3(nredo), dim3(64), 0, 0, din, (uint64_t)n, chunk, (uint64_t)chunk, dscr, (uint64_t)chunk, dcs, (const uint64_t*)nullptr, (uint64_t)0,
                           (const sqy::Lz4Block*)nullptr, (const uint32_t*)nullptr, 0u, dredo, (const uint32_t*)nullptr, 1u, sqy::Lz4DedupeArgs{}, ddg);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // checksum of what the kernel produced (sizes and bytes): a quick "did the output change" check between builds
    {
        std::vector<uint32_t> cs(nch);
        std::vector<uint8_t> scr(nch * (size_t)chunk);
        CK(hipMemcpy(cs.data(), dcs, nch * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(scr.data(), dscr, nch * (size_t)chunk, hipMemcpyDeviceToHost));
        unsigned long long hsh = 1469598103934665603ull, tot = 0;
        for (uint64_t k = 0; k < nch; ++k) {
            tot += cs[k];
            hsh = (hsh ^ cs[k]) * 1099511628211ull;
            for (uint32_t i = 0; i < cs[k]; ++i) hsh = (hsh ^ scr[k * (size_t)chunk + i]) * 1099511628211ull;
        }
        std::printf("output: %llu compressed bytes in %llu chunks, fnv %016llx\n", tot, (unsigned long long)nch, hsh);
    }
    std::vector<unsigned long long> dg(nch * 32);
    CK(hipMemcpy(dg.data(), ddg, nch * 32 * 8, hipMemcpyDeviceToHost));
    unsigned long long acc = 0, cnt = 0, rs[16] = {0};
    for (uint64_t k = 0; k < nch; ++k) { acc += dg[k * 32]; cnt += dg[k * 32 + 1]; for (int i = 0; i < 16; ++i) rs[i] += dg[k * 32 + 8 + i]; }
    if (std::getenv("SQY_DIAG_DENSE")) std::printf("dense: batches %llu, sequences %llu, mates resolved %llu, ended by long/lit/catch-up %llu, empty %llu, walk steps %llu, of them same-bucket lanes %llu | lean->generic U!=0 %llu U==0 %llu\n",
                rs[8] / nch, rs[9] / nch, rs[10] / nch, rs[11] / nch, rs[12] / nch, rs[13] / nch, rs[14] / nch, rs[6] / nch, rs[7] / nch);
    std::printf("region %2d -> %2d: kernel %.3f ms, %llu passes/chunk, avg %.0f cycles, total %.0f cycles/chunk | events/chunk: no-hit %llu f0>14 %llu far-fetch %llu hazard %llu long %llu slow-back %llu genericU!=0 %llu genericU==0 %llu\n",
                SQY_DIAG_A, SQY_DIAG_B, best, cnt / nch, cnt ? (double)acc / cnt : 0.0, (double)acc / nch,
                rs[0] / nch, rs[1] / nch, rs[2] / nch, rs[3] / nch, rs[4] / nch, rs[5] / nch, rs[6] / nch, rs[7] / nch);
    return 0;
}
