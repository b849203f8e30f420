// tools/lz4_diag.hip -- diagnostic build of the LZ4 chunk kernel with per-phase s_memtime accounting.
// Not part of the product: reads chunks from a file, runs the kernel, prints cycle shares per phase.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sqeazy_amd/csrc tools/lz4_diag.hip -o tools/lz4_diag
#define SQY_LZ4_DIAG 1
#include "../sqeazy_amd/csrc/sqy_kernels.hip"
#include <cstdio>
#include <vector>
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
    uint8_t *din, *dscr; uint32_t* dcs; unsigned long long* ddg;
    hipMalloc(&din, n + 64); hipMalloc(&dscr, nch * chunk); hipMalloc(&dcs, nch * 4); hipMalloc(&ddg, nch * 24 * 8);
    hipMemcpy(din, h.data(), n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(sqy::lz4_chunks_kernel, dim3((unsigned)nch), dim3(64), 0, 0, din, (uint64_t)n, chunk, dscr, (uint64_t)chunk, dcs, ddg);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::printf("launch %d: %.3f ms wall\n", rep, ms);
    }
    std::vector<unsigned long long> dg(nch * 24); std::vector<uint32_t> cs(nch);
    hipMemcpy(dg.data(), ddg, nch * 24 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(cs.data(), dcs, nch * 4, hipMemcpyDeviceToHost);
    const char* names[8] = {"ensure", "lean: seq reads+hash+put2", "lean: table read", "lean: cand reads+eval+ballots", "lean: winner+wide round", "lean: commit+emit+advance", "generic iteration (whole)", "loop-top"};
    for (uint64_t k = 0; k < nch; ++k) {
        unsigned long long tot = 0;
        for (int i = 0; i < 8; ++i) tot += dg[k * 24 + i];
        std::printf("chunk %llu csize %u total %llu cycles, matches %llu, batches %llu\n", (unsigned long long)k, cs[k], tot, dg[k * 24 + 8 + 6], dg[k * 24 + 8 + 0]);
        for (int i = 0; i < 8; ++i)
            std::printf("   %-20s %10llu cyc  %5.1f%%  n=%llu  avg %.0f\n", names[i], dg[k * 24 + i], 100.0 * dg[k * 24 + i] / (tot ? tot : 1),
                        dg[k * 24 + 8 + i], dg[k * 24 + 8 + i] ? (double)dg[k * 24 + i] / dg[k * 24 + 8 + i] : 0.0);
        std::printf("   lean misses: no-hit-in-64 %llu, f0>14 %llu, far-candidate %llu, hazard %llu | handover long-match %llu, slow-back %llu | generic batches: U!=0 %llu, U==0 %llu\n",
                    dg[k * 24 + 16], dg[k * 24 + 17], dg[k * 24 + 18], dg[k * 24 + 19], dg[k * 24 + 20], dg[k * 24 + 21], dg[k * 24 + 22], dg[k * 24 + 23]);
    }
    return 0;
}
