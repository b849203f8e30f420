// sqy_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the sqeazy hot path.
//
//   bitswap1 (u16, u8)      bit-plane transpose of the flat buffer, LDS-staged
//                           ref: src/cpp/src/encoders/bitswap_scheme_impl.hpp:97-145,
//                                bitplane_reorder_scalar.hpp:27-74, sse_utils.hpp:1365-1433
//   diff3x3x1 (u16, u8)     voxel - mean(3x3 in plane z-1), reference geometry quirks kept
//                           ref: encoders/diff_scheme_impl.hpp:78-139, diff_scheme_utils.hpp:70-99
//   lz4 block compress      one wavefront per 256 KiB chunk, bit-exact to liblz4 1.9.3
//                           LZ4_compress_fast_continue(fresh stream, byU32 table, accel 1) as reached
//                           from encoders/lz4_utils.hpp:99-173 via LZ4F_compressUpdate
//   frame compaction        per-chunk LZ4 frames concatenated (encoders/lz4_utils.hpp:193-274)
//
// No CUDA compatibility layer, no dual paths: this file only targets gfx950.
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdint.h>
#include <cstring>

#include "sqy_kernels.h"

namespace sqy {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
struct __attribute__((packed)) pk_u16 { uint16_t v; };
struct __attribute__((packed)) pk_u32 { uint32_t v; };
struct __attribute__((packed)) pk_u64 { uint64_t v; };
struct __attribute__((packed)) pk_u128 { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { return reinterpret_cast<const pk_u32*>(p)->v; }
__device__ __forceinline__ uint64_t ld_u64(const uint8_t* p) { return reinterpret_cast<const pk_u64*>(p)->v; }
__device__ __forceinline__ uint4 ld_u128(const uint8_t* p)
{
    const pk_u128 t = *reinterpret_cast<const pk_u128*>(p);
    return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st_u32(uint8_t* p, uint32_t v) { reinterpret_cast<pk_u32*>(p)->v = v; }
__device__ __forceinline__ void st_u128(uint8_t* p, uint4 v)
{
    pk_u128 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *reinterpret_cast<pk_u128*>(p) = t;
}

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// explicit address spaces: without them the compiler merges "read from the LDS ring or from global memory"
// into one pointer select and emits flat_load + s_waitcnt vmcnt(0) lgkmcnt(0) (seen in the ISA: ~1000 cycles each)
#define SQY_LDS __attribute__((address_space(3)))
#define SQY_GLB __attribute__((address_space(1)))
typedef SQY_LDS uint8_t lds_u8;
typedef SQY_GLB const uint8_t glb_u8;

__device__ __forceinline__ uint32_t lds_ld_u8(const lds_u8* p) { return *p; }
__device__ __forceinline__ uint32_t lds_ld_u32(const lds_u8* p) { return reinterpret_cast<const SQY_LDS pk_u32*>(p)->v; }
__device__ __forceinline__ uint64_t lds_ld_u64(const lds_u8* p) { return reinterpret_cast<const SQY_LDS pk_u64*>(p)->v; }
typedef uint32_t v4u_any __attribute__((ext_vector_type(4), aligned(1)));
// What a DS read off its natural alignment costs (round 4, tools/lds_bench2.hip, dependent chain of one wave): EVERY instruction
// with a lane off the natural alignment of its width is replayed, +64 cycles each -- ds_read_b32 73 -> 137, ds_read_b64 75 -> 139,
// ds_read_b128 87 -> 147, two ds_read_b64 back to back 87 -> 211, four ds_read_b32 95 -> 339; ds_read_u8 never (73); two
// ds_read2_b32 at a multiple of 4 (four dwords) 95.  Round 3 issued a 16-byte ring read as two ds_read_b64, which is the better
// shape only at multiples of 8: at byte granularity ONE ds_read_b128 is replayed once (147), the pair twice (211).
// SQ_LDS_UNALIGNED_STALL was 28 % of the wave-cycles of round 3's lz4_chunks.
// ONE 16-byte load at any alignment (volatile: four field loads get split by the optimiser into a b64 now and a conditional
// b64 later when only the low half decides a branch -- a second dependent LDS round trip on the parse's critical path).
__device__ __forceinline__ uint4 lds_ld_u128(const lds_u8* p)
{
    const v4u_any t = *reinterpret_cast<const volatile SQY_LDS v4u_any*>(p);
    return make_uint4(t.x, t.y, t.z, t.w);
}
// four dwords at a multiple of 4: two ds_read2_b32, never replayed (95 cycles against 147 for a ds_read_b128 off a multiple of 16)
struct __attribute__((aligned(4))) al4_u64 { uint32_t x, y; };
__device__ __forceinline__ uint4 lds_ld_4dw(const lds_u8* p)
{
    const volatile SQY_LDS al4_u64* q = reinterpret_cast<const volatile SQY_LDS al4_u64*>(p);
    const al4_u64 a = { q[0].x, q[0].y }, b = { q[1].x, q[1].y };
    return make_uint4(a.x, a.y, b.x, b.y);
}
// two dwords at a multiple of 4 (one ds_read2_b32)
__device__ __forceinline__ uint2 lds_ld_2dw(const lds_u8* p)
{
    const volatile SQY_LDS al4_u64* q = reinterpret_cast<const volatile SQY_LDS al4_u64*>(p);
    return make_uint2(q->x, q->y);
}
// the four bytes at any LDS address out of two aligned dwords (75 cycles + one v_alignbyte against 137 for a replayed ds_read_b32)
__device__ __forceinline__ uint32_t lds_ld_u32_via_aligned(const lds_u8* base, uint32_t at)
{
    const uint2 d = lds_ld_2dw(base + (at & ~3u));
    return __builtin_amdgcn_alignbyte(d.y, d.x, at & 3u);
}
__device__ __forceinline__ uint32_t glb_ld_u8(glb_u8* p) { return *p; }
// 16 bytes past this CU's L1 (agent scope: the line may sit there from before this wave's own later stores to it), waited for
__device__ __forceinline__ uint4 ld_u128_agent(const uint8_t* p)
{
    v4u v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t glb_ld_u32(glb_u8* p) { return reinterpret_cast<SQY_GLB const pk_u32*>(p)->v; }
__device__ __forceinline__ uint64_t glb_ld_u64(glb_u8* p) { return reinterpret_cast<SQY_GLB const pk_u64*>(p)->v; }
__device__ __forceinline__ uint4 glb_ld_u128(glb_u8* p)
{
    SQY_GLB const pk_u128* q = reinterpret_cast<SQY_GLB const pk_u128*>(p);
    return make_uint4(q->x, q->y, q->z, q->w);
}

// One lane's LDS write must be seen by another lane's later LDS read.  The hardware keeps a wave's LDS operations in
// order; what this stops is the COMPILER treating lanes as independent threads (it may turn "if (lane == k) t[h] = x;
// y = t[h];" into a diamond whose load side runs first).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, l); }
// (the builtin on a bool: __ballot() first turns the lane mask into 0 / 1 per lane and compares that again -- two vector
// instructions and their latency per ballot on the parse's critical path)
// v_writelane_b32: lane `l` of v takes the (uniform) value x.  (lane select through m0: two different SGPRs exceed the constant bus)
__device__ __forceinline__ uint32_t lane_write(uint32_t v, uint32_t x, uint32_t l)
{
    uint32_t km;
    asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(v), "=&s"(km) : "s"(x), "s"(l));
    return v;
}
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t ctz64(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }

// ------------------------------------------------------------------------------------------------
// bitswap1, 16-bit.  Tile = 8192 voxels (16 KiB) per wavefront:
//   16 coalesced 1 KiB loads -> LDS (row pitch 272 B, conflict-free for the row reads below)
//   lane L owns voxels [128 L, 128 L + 128) of the tile = 8 groups of 16 voxels
//   two groups at a time go through a 16x16 bit transpose held as (row of g | row of g+1 << 16)
//   -> for every plane the lane ends with 8 consecutive output words (16 B)
//   -> 16 coalesced 1 KiB stores, one per plane segment.
// Output word of plane b (segment 15-b) for voxels 16w..16w+15 carries voxel 16w+j at bit 15-j.
// ------------------------------------------------------------------------------------------------
constexpr int BSW_TILE_VOX = 8192;

// rows r[i] = (voxel i of group A) | (voxel i of group B) << 16, i = 0..15, voxel order REVERSED by the
// caller (r[i] holds voxel 15-i) so that a plain transpose yields msb-first plane words.
__device__ __forceinline__ void transpose16x16_pairs(uint32_t r[16])
{
    // 8x8 blocks: rows i <-> i+8, byte granularity
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t a = r[i], b = r[i + 8];
        // per 16-bit half: a' = (a & 0x00ff) | (b & 0x00ff) << 8 ; b' = (a >> 8 & 0x00ff) | (b & 0xff00)
        r[i] = __builtin_amdgcn_perm(b, a, 0x06020400u);
        r[i + 8] = __builtin_amdgcn_perm(b, a, 0x07030501u);
    }
#pragma unroll
    for (int blk = 0; blk < 16; blk += 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t a = r[blk + i], b = r[blk + i + 4];
            r[blk + i] = (a & 0x0f0f0f0fu) | ((b << 4) & 0xf0f0f0f0u);
            r[blk + i + 4] = ((a >> 4) & 0x0f0f0f0fu) | (b & 0xf0f0f0f0u);
        }
    }
#pragma unroll
    for (int blk = 0; blk < 16; blk += 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t a = r[blk + i], b = r[blk + i + 2];
            r[blk + i] = (a & 0x33333333u) | ((b << 2) & 0xccccccccu);
            r[blk + i + 2] = ((a >> 2) & 0x33333333u) | (b & 0xccccccccu);
        }
    }
#pragma unroll
    for (int blk = 0; blk < 16; blk += 2) {
        const uint32_t a = r[blk], b = r[blk + 1];
        r[blk] = (a & 0x55555555u) | ((b << 1) & 0xaaaaaaaau);
        r[blk + 1] = ((a >> 1) & 0x55555555u) | (b & 0xaaaaaaaau);
    }
}

// (round 6, built, exact, measured, not kept: the tile as sixteen non-temporal loads of 4 x 256 contiguous bytes -- 7.1 TB/s for a 1 GiB
// read on this chip against 5.2 for "every lane its own 256 bytes", tools/read_bench.hip -- and the 16 x 16 transpose of sixteen-byte elements
// across the sixteen lanes of a DPP row that puts a lane's own 256 bytes back into its registers (four butterfly stages of row_shr / row_shl
// moves, the bank mask as the select: ~400 vector instructions, no LDS).  The kernel that only reads: 0.249 -> 0.214 ms; the real one, ten
// planes written: 0.380 -> 0.402; in flight -2 %.  At ~1000 vector instructions per tile the transposer is no longer only memory-bound:
// profiles/r06_experiments.txt.)
// No LDS: every lane loads its own 256 contiguous bytes (16 x 16 B at a lane stride of 256 B; the eight loads that share
// a 128-byte line are issued back to back, so the line is fetched once and hit in L1 after that).  Needs no LDS allocation
// at all, which matters when the CUs' LDS is held by resident LZ4 chunk waves of other calls in flight: this kernel then
// still finds room (wave slots and registers only).
// piece_hash != nullptr: a 32-bit hash of every 1 KiB piece of plane data the wave writes goes to piece_hash (four partial
// sums per piece, one per 16-lane row) -- what the LZ4 stage's duplicate-chunk detection is built on (lz4_dedupe_*).
// GAP (frames in place, see lz4_tail_*): the plane stream is written as the BODIES of the LZ4 frames it will be cut into --
// chunk k of 2^gap_shift bytes starts at out + k * (chunk + 15), the 15 bytes in between being the frame header, size field and
// end mark of a stored frame.  A chunk the LZ4 stage stores raw then never moves again.  The 1 KiB pieces land 0..15 bytes
// off a 16-byte boundary; plain unaligned 16-byte stores (measured within 8 % of aligned ones, tools/ubench_align.hip).
// side != nullptr (diff3x3x1 right in front, see diff3x3x1_u16_rows_kernel): columns x < side_w of every row (X voxels, a multiple
// of the 128 a lane owns) come from the compact buffer `side` (rows side_w voxels apart), everything else from `in`.
__device__ __forceinline__ uint32_t lz4_hash5_32(uint32_t lo, uint32_t byte4);      // (liblz4's 5-byte hash, defined with the LZ4 kernels below)
template <bool GAP>
__global__ __launch_bounds__(256)       // (launched in blocks of 128; bounds of 128 let the compiler take 173 registers instead of 127)
void bitswap1_u16_regs(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, uint64_t n_tiles, uint64_t seg_words,
                       uint32_t* __restrict__ piece_hash, uint32_t gap_shift, const uint16_t* __restrict__ side, uint32_t side_w, uint32_t X,
                       uint32_t* __restrict__ digest, uint32_t digest_stride)
{
    const int lane = threadIdx.x & 63;
    // (the wave number through readfirstlane: tile and every piece address below are then scalar, the lane part a 32-bit offset)
    const uint32_t wpb = blockDim.x >> 6;
    const uint64_t wave_global = (uint64_t)blockIdx.x * wpb + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave_stride = (uint64_t)gridDim.x * wpb;
    // (round 6, measured and not kept: consecutive tiles on one wave, so that the line two neighbouring 1 KiB pieces share is written
    // whole -- 3 % slower, alone and in flight; non-temporal loads -- the eight loads that share a line no longer hit in L1: 0.47 -> 0.80 ms)
    for (uint64_t tile = wave_global; tile < n_tiles; tile += wave_stride) {
        const v4u* src = reinterpret_cast<const v4u*>(in + tile * BSW_TILE_VOX) + lane * 16;
        if (side) {
            const uint64_t i0 = tile * BSW_TILE_VOX + (uint64_t)lane * 128u;
            const uint64_t row = i0 / X;
            const uint32_t x0 = (uint32_t)(i0 - row * X);
            if (x0 < side_w) src = reinterpret_cast<const v4u*>(side + row * side_w + x0);
        }
        v4u v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = src[j];
        uint32_t pl[16][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4u a0 = v[4 * q], a1 = v[4 * q + 1], b0 = v[4 * q + 2], b1 = v[4 * q + 3];
            const uint32_t ga[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const uint32_t gb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            uint32_t r[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                r[15 - 2 * k] = __builtin_amdgcn_perm(gb[k], ga[k], 0x05040100u);
                r[14 - 2 * k] = __builtin_amdgcn_perm(gb[k], ga[k], 0x07060302u);
            }
            transpose16x16_pairs(r);
#pragma unroll
            for (int b = 0; b < 16; ++b) pl[b][q] = r[b];
        }
        // ---- the NOISE DIGEST (round 6; GAP only, the host offers it when every plane segment is a whole number of chunks) ----
        // While liblz4's search finds nothing, WHERE it probes is a closed form: probe u of a search that starts with a chunk sits at
        // B_s + (u - u_s) * s for u_s = 64 s - 62 <= u <= 64 s + 1, B_s = 2 + 32 s (s - 1) (s = the step; tools/lz4_parse_stats.c, checked
        // against the loop itself).  A chunk of noise is parsed -- to find that it has to be stored -- by ~5760 such probes, each of which
        // needs the bucket and the tag of five bytes: 0.6 GB of plane stream read again per GiB of voxels for 23 KB per chunk of answers.
        // Those answers are left HERE, where the bytes are in registers: from probe 961 on (step >= 16: at most one probe per lane's
        // sixteen bytes) lane L works out which probe, if any, starts inside its bytes [x, x + 16) of the chunk, takes the five bytes
        // from there (the next lane's first dword through a DPP shift; the wave's last lane cannot see behind its bytes: such an entry
        // says "look yourself"), and stores bucket << 16 | tag at digest[chunk][u - 961].  The chunk's parse wave (lz4_chunks_kernel, the
        // batches proved empty) takes its batches from there as long as nothing has matched.  Exact by construction: the entries are what
        // the parse computes from the same bytes.  Pieces that are all zero write nothing; a chunk with such a piece is parsed from its bytes.
        uint32_t dg = 0;                       // one register across the planes: bit 31 = a probe starts in my bytes | probe number - 961 << 4 | its byte
        if (GAP && digest) {
            const uint32_t x = (((uint32_t)tile << 10) & ((1u << gap_shift) - 1u)) + (uint32_t)lane * 16u;     // my bytes inside the chunk
            if (x + 16u > 7667u) {                                              // (probe 961 sits at byte 7667)
                const uint32_t xx = x > 7667u ? x : 7667u;
                uint32_t st = (uint32_t)((1.0f + __builtin_sqrtf(1.0f + (float)(xx - 2u) * 0.125f)) * 0.5f);
                if (2u + 32u * (st + 1u) * st <= xx) ++st;                      // B_{s+1} <= xx
                if (2u + 32u * st * (st - 1u) > xx) --st;                       // B_s > xx
                const uint32_t Bs = 2u + 32u * st * (st - 1u), num = xx - Bs;  // num < 64 s
                uint32_t kk = (uint32_t)((float)num / (float)st);
                if (kk * st > num) --kk;
                if ((kk + 1u) * st <= num) ++kk;
                if (kk * st != num) ++kk;                                      // ceil
                const uint32_t cand = Bs + kk * st, u = 64u * st - 62u + kk;
                if (cand < x + 16u && u >= 961u && u - 961u < digest_stride) dg = 0x80000000u | ((u - 961u) << 4) | (cand - x);
            }
        }
        const uint32_t dg_tsh = 31u - gap_shift;                            // the parse's tag width for a whole chunk (positions take gap_shift bits)
        // One pass over the planes: the piece's hash, then the piece (round 6: hash and store of a plane next to each other, its four
        // registers are free behind them; two separate loops kept all sixty-four alive across sixteen branches).
        const uint32_t pm = 2u * (uint32_t)lane + 1u;                    // position inside the piece
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            uint32_t vx = pl[b][0], vy = pl[b][1], vz = pl[b][2], vw = pl[b][3];
            // (nothing of plane b is computed before plane b - 1 has left: hoisted, the sixteen planes' hash terms cost 40 registers)
            asm volatile("" : "+v"(vx), "+v"(vy), "+v"(vz), "+v"(vw) :: "memory");
            const v4u val = {vx, vy, vz, vw};
            const uint64_t nzm = ballot((val.x | val.y | val.z | val.w) != 0u);
            if (piece_hash) {
                uint32_t a = val.x * 0x9E3779B1u + val.y * 0x85EBCA77u + val.z * 0xC2B2AE3Du + val.w * 0x27D4EB2Fu;
                a = (a ^ (a >> 15)) * pm;
                // EXACT zero marker: a row of 16 lanes stores 0 if and only if all its 256 bytes are zero (its hash terms are
                // all 0 then; a row with any non-zero word gets bit 0 forced on) -- all-zero chunks need no byte compare
                const bool row_nz = ((nzm >> (lane & 48)) & 0xffffull) != 0ull;
                a += __builtin_amdgcn_update_dpp(0u, a, 0x111, 0xf, 0xf, false);      // row_shr:1  (sum over the 16-lane row ends in its last lane)
                a += __builtin_amdgcn_update_dpp(0u, a, 0x112, 0xf, 0xf, false);
                a += __builtin_amdgcn_update_dpp(0u, a, 0x114, 0xf, 0xf, false);
                a += __builtin_amdgcn_update_dpp(0u, a, 0x118, 0xf, 0xf, false);
                // piece = 1 KiB of segment 15-b: (byte offset of the piece in the plane stream) / 1024
                const uint64_t piece = ((uint64_t)(15 - b) * seg_words * 2u + tile * 1024u) >> 10;
                if ((lane & 15) == 15) piece_hash[piece * 4u + (uint32_t)(lane >> 4)] = row_nz ? (a | 1u) : 0u;
            }
            if (GAP) {
                // HOLES: a piece that is all zero is not written at all -- its hash says so (the exact zero marker above), and
                // lz4_dedupe_verify_kernel fills such pieces in for the chunks somebody is going to read.  Chunks that are all
                // zero are never read but for the first: bit planes above the largest voxel value are HBM traffic nobody needs.
                if (nzm == 0ull) continue;
                const uint64_t B = (uint64_t)(15 - b) * seg_words * 2u + tile * 1024u;       // byte offset of the piece in the plane stream
                if (digest) {
                    const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)val.x, 0x130, 0xf, 0xf, true);      // wave_shl:1: the next lane's first dword
                    const uint32_t dg_off = dg & 15u, dg_q = dg_off >> 2, dg_r = dg_off & 3u;
                    const uint32_t d0 = dg_q == 0u ? val.x : dg_q == 1u ? val.y : dg_q == 2u ? val.z : val.w;
                    const uint32_t d1 = dg_q == 0u ? val.y : dg_q == 1u ? val.z : dg_q == 2u ? val.w : nx;
                    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, dg_r);                  // bytes off .. off + 3
                    const uint32_t b5 = d1 >> (8u * dg_r);                                          // byte off + 4 (in its low byte)
                    uint32_t e = (lz4_hash5_32(lo, b5) << 16) | ((lo * 2654435761u) >> (32u - dg_tsh));
                    if (lane == 63 && dg_off > 11u) e = 0xffffffffu;                                // its five bytes end in the next tile's piece
                    // (the chunk's row of the digest as a SCALAR base, the lane's entry a 32-bit offset: see the piece's address below)
                    const uint64_t drow = (B >> gap_shift) * (uint64_t)digest_stride * 4u;
                    const uint32_t drow_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)drow);
                    const uint32_t drow_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(drow >> 32));
                    uint8_t* const dbase = reinterpret_cast<uint8_t*>(digest) + (((uint64_t)drow_hi << 32) | drow_lo);
                    if ((int32_t)dg < 0) *reinterpret_cast<uint32_t*>(dbase + ((dg >> 2) & 0x1ffffffcu)) = e;
                }
                // (round 6) the piece's address stays SCALAR: "* 15" as shift and subtract -- there is no 64-bit scalar multiply, the
                // compiler moved the product, and with it sixteen 64-bit store addresses, into vector registers
                const uint64_t kq = B >> gap_shift;
#ifdef SQY_EXP_GAP_BYTES
                const uint64_t off = B + kq * SQY_EXP_GAP_BYTES;         // (tools/bsw_bench.hip: what the 15-byte gaps' misalignment costs)
#else
                const uint64_t off = B + (kq << 4) - kq;
#endif
                const uint32_t off_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
                const uint32_t off_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(off >> 32));
                uint8_t* dst = reinterpret_cast<uint8_t*>(out) + (((uint64_t)off_hi << 32) | off_lo);
                // (non-temporal, round 6: the plane stream is read again much later and by other XCDs; kept out of this XCD's L2 it leaves the
                // lines the LZ4 chunk waves of the calls in flight read twice -- candidates behind their LDS window -- where they are: with the
                // noise digest in place +3 %; without it, when the HBM's bandwidth was the limit, +-0)
                __builtin_nontemporal_store((v4u_any)val, reinterpret_cast<v4u_any*>(dst + (uint32_t)lane * 16u));
            } else {
                v4u* dst = reinterpret_cast<v4u*>(out + (uint64_t)(15 - b) * seg_words + tile * (BSW_TILE_VOX / 16));
                __builtin_nontemporal_store(val, dst + lane);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Duplicate chunks in front of LZ4.  Bit planes of small-magnitude data repeat: the planes above the largest value are all
// zero, the sign-extension planes of a residual (diff3x3x1) are copies of each other -- whole 256 KiB chunks of the plane
// stream are byte-identical, and identical chunks compress to identical frames (every chunk starts from a fresh table).
// key kernel: chunk key from the piece hashes the bit-plane transpose left behind, inserted into an open-addressing table
// that keeps the SMALLEST chunk number per key.  verify kernel: a chunk whose key belongs to an earlier chunk is compared
// with it byte for byte; only then is it marked dup_of[k] = that chunk.  The LZ4 kernel skips marked chunks, scan and gather
// take size and bytes from the chunk they duplicate.  Exact by construction: the hash only nominates, the compare decides.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64)
void lz4_dedupe_key_kernel(const uint32_t* __restrict__ piece_hash, uint32_t pieces_per_chunk, uint64_t nchunks_full,
                           uint64_t* __restrict__ chunk_key, uint64_t* __restrict__ tab_key, uint32_t* __restrict__ tab_val, uint32_t tab_mask,
                           uint64_t* __restrict__ holes_map, uint32_t pieces_last)
{
    const uint64_t k = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t* ph = piece_hash + k * pieces_per_chunk * 4u;
    // (holes_map: the grid also covers the ragged last chunk, which takes no part in the duplicate search)
    const uint32_t np = k < nchunks_full ? pieces_per_chunk : pieces_last;
    uint64_t* hm = holes_map ? holes_map + k * (1u + (pieces_per_chunk + 63u) / 64u) : nullptr;
    uint32_t h1 = 0, h2 = 0, any = 0, nzero = 0;
    for (uint32_t base = 0; base < np; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        uint4 q = make_uint4(1u, 0u, 0u, 0u);
        if (i < np) q = *reinterpret_cast<const uint4*>(ph + i * 4u);
        any |= i < np ? (q.x | q.y | q.z | q.w) : 0u;
        const uint32_t s_ = i < np ? q.x + q.y + q.z + q.w : 0u;
        h1 += s_ * ((2u * i + 1u) * 0x9E3779B1u);
        h2 += ((s_ << 13) | (s_ >> 19)) * ((2u * i + 3u) * 0x85EBCA77u);
        if (hm) {
            // which pieces are all zero (the exact marker) -- the pieces the transpose left unwritten, see lz4_dedupe_verify_kernel
            const uint64_t zm = ballot((q.x | q.y | q.z | q.w) == 0u);
            nzero += (uint32_t)__builtin_popcountll(zm);
            if (lane == 0) hm[1u + base / 64u] = zm;
        }
    }
    if (hm && lane == 0) hm[0] = nzero;
    if (k >= nchunks_full) return;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { h1 += __shfl_xor(h1, d); h2 += __shfl_xor(h2, d); }
    const bool some = ballot(any != 0u) != 0ull;
    if (lane == 0) {
        // key 1 = "every byte of the chunk is zero" (exact, see the piece hashes); other keys have bit 1 set; 0 = empty slot
        const uint64_t key = some ? ((((uint64_t)h2 << 32) | h1) | 3ull) : 1ull;
        chunk_key[k] = key;
        uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & tab_mask;
        for (uint32_t tries = 0; tries <= tab_mask; ++tries, slot = (slot + 1u) & tab_mask) {
            const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&tab_key[slot]), 0ull, (unsigned long long)key);
            if (prev == 0ull || prev == key) { atomicMin(&tab_val[slot], (uint32_t)k); break; }
        }
    }
}

constexpr uint32_t DEDUPE_THREADS = 256;
__global__ __launch_bounds__(DEDUPE_THREADS)
void lz4_dedupe_verify_kernel(const uint8_t* __restrict__ in, uint32_t chunk, uint64_t in_stride, uint64_t nchunks_full, uint64_t nchunks,
                              const uint64_t* __restrict__ chunk_key, const uint64_t* __restrict__ tab_key,
                              const uint32_t* __restrict__ tab_val, uint32_t tab_mask, uint32_t* __restrict__ dup_of,
                              const uint32_t* __restrict__ piece_hash, const uint64_t* __restrict__ holes_map, uint64_t total)
{
    const uint64_t k = blockIdx.x;
    __shared__ uint32_t s_rep;
    __shared__ uint32_t s_diff;
    if (threadIdx.x == 0) {
        uint32_t r = (uint32_t)k;
        if (k < nchunks_full) {
            const uint64_t key = chunk_key[k];
            uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & tab_mask;
            for (uint32_t tries = 0; tries <= tab_mask; ++tries, slot = (slot + 1u) & tab_mask) {
                const uint64_t tk = tab_key[slot];
                if (tk == key) { r = tab_val[slot]; break; }
                if (tk == 0) break;
            }
        }
        s_rep = r;
        s_diff = 0;
    }
    __syncthreads();
    const uint32_t r = s_rep;
    // HOLES (frames in place, see bitswap1_u16_regs): the all-zero 1 KiB pieces of the plane stream were never written; the map
    // (lz4_dedupe_key_kernel: their number per chunk, then one bit per piece) says which.  They are filled in here for every chunk
    // somebody is going to read: all but the chunks that are all zero AND a duplicate -- those are neither parsed (their frame is
    // the first all-zero chunk's) nor ever stored raw.  Bit planes above the data's range are then neither written nor read; what
    // is left to fill are the zero pieces inside chunks that also hold data.  (Nobody reads a hole before this kernel is over: the
    // compare below goes by the markers.)
    auto fill_holes = [&]() {
        if (!holes_map) return;
        const uint64_t left = total - k * chunk;
        const uint32_t np = (uint32_t)((left < chunk ? left : chunk) >> 10);
        const uint64_t* hm = holes_map + k * (1u + ((chunk >> 10) + 63u) / 64u);
        if ((uint32_t)hm[0] == 0u) return;
        uint8_t* body = const_cast<uint8_t*>(in) + k * in_stride;
        const uint32_t lane = threadIdx.x & 63u;
        for (uint32_t pc = threadIdx.x >> 6; pc < np; pc += DEDUPE_THREADS / 64u) {
            if (!((hm[1u + pc / 64u] >> (pc & 63u)) & 1ull)) continue;
            const v4u zero = {0, 0, 0, 0};
            *reinterpret_cast<v4u_any*>(body + (uint64_t)pc * 1024u + lane * 16u) = zero;
        }
    };
    if (r >= k) { if (threadIdx.x == 0) dup_of[k] = (uint32_t)k; fill_holes(); return; }     // (uniform) first of its kind
    if (chunk_key[k] == 1ull) { if (threadIdx.x == 0) dup_of[k] = r; return; } // all zero, exactly: equal to the first all-zero chunk
    fill_holes();
    // compare chunk k with chunk r (chunk is a multiple of 1 KiB here; chunk bodies may sit at any byte alignment)
    const v4u_any* a = reinterpret_cast<const v4u_any*>(in + k * in_stride);
    const v4u_any* b = reinterpret_cast<const v4u_any*>(in + (uint64_t)r * in_stride);
    const uint32_t nvec = chunk >> 4;
    uint32_t diff = 0;
    // holes: pieces whose hash is the zero marker count as zeros, written or not (a wavefront's 64 x 16 bytes are exactly one piece)
    const uint4* hk = reinterpret_cast<const uint4*>(piece_hash) + k * (chunk >> 10);
    const uint4* hr = reinterpret_cast<const uint4*>(piece_hash) + (uint64_t)r * (chunk >> 10);
    for (uint32_t i = threadIdx.x; i < nvec; i += DEDUPE_THREADS * 4u) {
        // four independent 16-byte pairs in flight per thread
        v4u x[4], y[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint32_t j = i + u * DEDUPE_THREADS;
            const v4u zero = {0, 0, 0, 0};
            bool za = false, zb = false;
            if (holes_map && j < nvec) {
                const uint4 qa = hk[j >> 6], qb = hr[j >> 6];
                za = (qa.x | qa.y | qa.z | qa.w) == 0u;
                zb = (qb.x | qb.y | qb.z | qb.w) == 0u;
            }
            x[u] = (j < nvec && !za) ? (v4u)a[j] : zero;
            y[u] = (j < nvec && !zb) ? (v4u)b[j] : zero;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) diff |= (x[u].x ^ y[u].x) | (x[u].y ^ y[u].y) | (x[u].z ^ y[u].z) | (x[u].w ^ y[u].w);
    }
    if (diff) atomicOr(&s_diff, 1u);
    __syncthreads();
    if (threadIdx.x == 0) dup_of[k] = s_diff ? (uint32_t)k : r;
}

// generic (any length / alignment) path: one thread per output word, used for the part of the buffer
// the tile kernel does not cover and for the copied tail (bitswap_scheme_impl.hpp:99-103).
__global__ __launch_bounds__(256)
void bitswap1_u16_generic(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                          uint64_t len, uint64_t first_word, uint64_t seg_words)
{
    const uint64_t L = seg_words * 16;
    const uint64_t w = first_word + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < seg_words) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = in[w * 16 + j];
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc |= ((v[j] >> b) & 1u) << (15 - j);
            out[(uint64_t)(15 - b) * seg_words + w] = (uint16_t)acc;
        }
    }
    // tail elements [L, len) are copied verbatim
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && t < len - L) out[L + t] = in[L + t];
}

// 8-bit: one thread per output byte (8 input bytes -> one byte in each of 8 planes).
// (8 coalesced byte stores per thread; measured 4.9 TB/s of read + write on a 1 GiB volume: at the streaming kernels' rate)
__global__ __launch_bounds__(256)
void bitswap1_u8_generic(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t len, uint64_t seg_bytes)
{
    const uint64_t L = seg_bytes * 8;
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < seg_bytes) {
        const uint64_t x = *reinterpret_cast<const uint64_t*>(in + w * 8); // byte j = voxel j
        // 8x8 bit transpose (rows = voxels, little-endian bytes); wanted: plane b byte with voxel j at bit 7-j
        uint64_t t = x;
        uint64_t y;
        y = (t ^ (t >> 7)) & 0x00AA00AA00AA00AAull; t = t ^ y ^ (y << 7);
        y = (t ^ (t >> 14)) & 0x0000CCCC0000CCCCull; t = t ^ y ^ (y << 14);
        y = (t ^ (t >> 28)) & 0x00000000F0F0F0F0ull; t = t ^ y ^ (y << 28);
        // now byte b of t holds bit b of every voxel with voxel j at bit j -> reverse bits in each byte
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint32_t byte = (uint32_t)(t >> (8 * b)) & 0xffu;
            out[(uint64_t)(7 - b) * seg_bytes + w] = (uint8_t)(__brev(byte) >> 24);
        }
    }
    if (blockIdx.x == 0 && w < len - L) out[L + w] = in[L + w];
}

// ------------------------------------------------------------------------------------------------
// diff3x3x1.  out = in, except at flat indices idx in  U_{z in [1,min(X,Z)), y in [1,Y-1)}
// [z*Y*X + y*X + 1, +Z-2)  where out[idx] = in[idx] - (wrapping 9-neighbour sum of plane z-1)/9.
// (halo quirk: the per-row extent is Z-2 and comes from the depth; rows may run into the next row.)
// ------------------------------------------------------------------------------------------------
// idx is rewritten iff the LARGEST row start s <= idx (row starts: z*frame + y*X + 1, 1 <= z < zlim,
// 1 <= y <= Y-2) satisfies idx < s + hx  (any covering row implies the nearest one covers too).
__device__ __forceinline__ bool diff_touched(uint64_t idx, uint64_t length, uint64_t Y, uint64_t X, uint64_t hx,
                                             uint64_t zlim, bool single)
{
    const uint64_t frame = Y * X;
    const uint64_t z = idx / frame;
    const uint64_t r = idx - z * frame;
    uint64_t s;
    if (z >= 1 && z < zlim && r >= X + 1) {
        uint64_t y = (r - 1) / X;
        if (y > Y - 2) y = Y - 2;
        s = z * frame + y * X + 1;
    } else {
        const uint64_t zz = z < zlim ? z : zlim;   // last row of the nearest earlier frame that has rows
        if (zz < 2) return false;
        s = (zz - 1) * frame + (Y - 2) * X + 1;
    }
    const uint64_t reach = single ? (length - s) : hx;   // diff_scheme_impl.hpp:97-101
    return idx - s < reach;
}

// SCHAR: the stage as a TAIL filter on the sink's `char` output (src/sqeazy_pipelines.hpp:64-77; char is signed): the wrapping
// 8-bit sum is SIGN-EXTENDED into the reference's unsigned short sum_type (diff_scheme_impl.hpp:24, traits.hpp:29) before the
// division -- a sum of -5 divides as 65531.
template <typename T, typename ST, bool SCHAR = false>
__global__ __launch_bounds__(256)
void diff3x3x1_kernel(const T* __restrict__ in, T* __restrict__ out, uint64_t Z, uint64_t Y, uint64_t X,
                      uint64_t hx, uint64_t zlim, int single)
{
    const uint64_t length = Z * Y * X;
    const uint64_t frame = Y * X;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < length; idx += (uint64_t)gridDim.x * blockDim.x) {
        T v = in[idx];
        if (diff_touched(idx, length, Y, X, hx, zlim, single != 0)) {
            const T* p = in + idx - frame;
            T sum = 0;
            sum = (T)(sum + p[-(int64_t)X - 1]); sum = (T)(sum + p[-(int64_t)X]); sum = (T)(sum + p[-(int64_t)X + 1]);
            sum = (T)(sum + p[-1]);              sum = (T)(sum + p[0]);           sum = (T)(sum + p[1]);
            sum = (T)(sum + p[X - 1]);           sum = (T)(sum + p[X]);           sum = (T)(sum + p[X + 1]);
            const uint32_t mean = SCHAR ? (uint32_t)(uint16_t)(int16_t)(int8_t)sum / 9u : (uint32_t)sum / 9u;
            v = (T)(ST)((uint32_t)v - mean);
        }
        out[idx] = v;
    }
}

// 16-bit fast path for the common geometry (every row's reach 1 + hx stays inside its row, i.e. Z-2 <= X-1):
// thread = 8 consecutive voxels of one row (one 16-byte load per source row), grid = (row pieces, Y, Z), so no
// coordinate is ever recovered by division.  Rewritten voxels: 1 <= z < zlim, 1 <= y <= Y-2, 1 <= x < 1 + hx.
// out_stride / xlim: the output rows are out_stride voxels apart and only columns below xlim are written (X / X: the whole volume,
// the stage on its own).  With a bit-plane transpose right behind, only the columns the stage can touch at all -- x < 1 + hx,
// and hx comes from the DEPTH of the stack -- go to a compact side buffer, the transpose takes everything else
// from the stage's input: for a 2048 x 2048 x 256 slab that is 1/8 of the volume instead of a full read-and-write pass.
// DECODE: the inverse, one frame (or a run of untouched frames) per launch: voxel + mean of the DECODED frame z-1, read from `out`
// (rows X apart there); frames z0 + blockIdx.z, columns [xbeg, xlim).
template <bool DECODE>
__global__ __launch_bounds__(256)
void diff3x3x1_u16_rows_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, uint32_t Y, uint32_t X,
                               uint32_t hx, uint32_t zlim, uint32_t out_stride, uint32_t xlim, uint32_t rows_per_block, uint32_t z0, uint32_t xbeg)
{
    // rows_per_block (a power of two up to 16): a block covers that many rows with 256 / rows_per_block threads each (narrow xlim)
    const uint32_t tpr = 256u / rows_per_block;
    const uint32_t z = z0 + blockIdx.z, y = blockIdx.y * rows_per_block + threadIdx.x / tpr;
    const uint32_t x0 = xbeg + (blockIdx.x * tpr + threadIdx.x % tpr) * 8u;
    if (x0 >= xlim || y >= Y) return;
    const uint64_t frame = (uint64_t)Y * X;
    const uint64_t row = (uint64_t)z * frame + (uint64_t)y * X;
    const uint16_t* __restrict__ cur = in + row;
    uint16_t* __restrict__ dst = out + ((uint64_t)z * Y + y) * out_stride;
    const bool full = x0 + 8u <= X;
    uint32_t v[8];
    if (full && (((uintptr_t)(cur + x0)) & 15) == 0) {
        const uint4 c = *reinterpret_cast<const uint4*>(cur + x0);
        v[0] = c.x & 0xffffu; v[1] = c.x >> 16; v[2] = c.y & 0xffffu; v[3] = c.y >> 16;
        v[4] = c.z & 0xffffu; v[5] = c.z >> 16; v[6] = c.w & 0xffffu; v[7] = c.w >> 16;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (x0 + j < X) ? cur[x0 + j] : 0u;
    }
    const bool row_touched = z >= 1u && z < zlim && y >= 1u && y + 2u <= Y && hx > 0u && x0 < 1u + hx && x0 + 8u > 1u;
    if (row_touched) {
        // column sums over the three rows of plane z-1 for columns x0-1 .. x0+8 (wrapping 16-bit arithmetic)
        uint32_t colsum[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) colsum[j] = 0;
        const uint16_t* up = (DECODE ? (const uint16_t*)out : in) + row - frame;       // (decode: `out` holds whole frames, rows X apart)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const uint16_t* r = up + (int64_t)dy * X;
            uint32_t e[10];
            e[0] = (x0 > 0u) ? r[x0 - 1] : 0u;                       // only used for x = x0 >= 1
            if (full && (((uintptr_t)(r + x0)) & 15) == 0) {
                const uint4 c = *reinterpret_cast<const uint4*>(r + x0);
                e[1] = c.x & 0xffffu; e[2] = c.x >> 16; e[3] = c.y & 0xffffu; e[4] = c.y >> 16;
                e[5] = c.z & 0xffffu; e[6] = c.z >> 16; e[7] = c.w & 0xffffu; e[8] = c.w >> 16;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) e[1 + j] = (x0 + j < X) ? r[x0 + j] : 0u;
            }
            e[9] = (x0 + 8u < X) ? r[x0 + 8] : 0u;
#pragma unroll
            for (int j = 0; j < 10; ++j) colsum[j] += e[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t x = x0 + j;
            if (x >= 1u && x < 1u + hx && x < X) {
                const uint32_t sum = (colsum[j] + colsum[j + 1] + colsum[j + 2]) & 0xffffu;   // T sum wraps mod 2^16
                const uint32_t mean = (sum * 58255u) >> 19;                                    // sum / 9 for sum < 65536
                v[j] = (DECODE ? v[j] + mean : v[j] - mean) & 0xffffu;
            }
        }
    }
    if (full && (((uintptr_t)(dst + x0)) & 15) == 0) {
        *reinterpret_cast<uint4*>(dst + x0) = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (x0 + j < X) dst[x0 + j] = (uint16_t)v[j];
    }
}

// diff3x3x1 decode, SEVERAL frames per launch (round 4).  Frame z needs the DECODED frame z-1, so round 3 ran one launch per frame
// (255 launch-bound kernels of 1 MiB each on a 2048 x 2048 x 256 slab: 1.6 ms).  Here a block owns a strip of R rows and decodes K
// frames in a row without waiting for anybody: what it needs of its neighbours' strips it computes itself -- frame z+j of rows
// [y0 - (K-1-j), y1 + (K-1-j)), a halo that shrinks by one row per frame -- starting from the decoded frame z-1, which the launch
// before has finished.  Two images of the (R + 2K) x w touched columns take turns in LDS; only the block's own rows go to `out`.
// (K - 1) / R of the work is redundant (K = 8, R = 32: 22 %), the chain is Z / K launches instead of Z.
// w columns (a multiple of 8, the touched ones and their right-hand neighbour) are decoded; rewritten: 1 <= z < zlim, 1 <= y <= Y - 2,
// 1 <= x < 1 + hx.  `out` rows are X voxels apart and hold the decoded frames.
constexpr uint32_t DDK_K = 8, DDK_R = 32, DDK_ROWS = DDK_R + 2 * DDK_K;

constexpr uint32_t DDK_THREADS = 1024, DDK_ITEMS = 2;          // an image is at most DDK_ITEMS * DDK_THREADS (row, vector) items (launcher)

__global__ __launch_bounds__(DDK_THREADS)
void diff3x3x1_u16_decode_frames_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, uint32_t Y, uint32_t X, uint32_t hx,
                                        uint32_t zlim, uint32_t w, uint32_t z0, uint32_t nframes)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t ddk_lds[];   // 2 images of DDK_ROWS rows
    const uint32_t pitch = w + 16u;                            // voxels per LDS row: [8 .. 8 + w) = columns 0 .. w-1, [8 + w] = column w (16-byte aligned rows)
    uint16_t* img[2] = {ddk_lds, ddk_lds + (size_t)DDK_ROWS * pitch};
    const int32_t y0 = (int32_t)(blockIdx.x * DDK_R);
    const uint64_t frame = (uint64_t)Y * X;
    const uint32_t vpr = w / 8u;                               // 8-voxel vectors per row; item vpr of a row is the single column w
    // A thread owns the same (image row, vector) items in every frame: image row i <-> volume row y0 - K + i.  The encoded voxels of
    // the NEXT frame are fetched while the frame in hand is computed (a block is 16 waves on one CU: nothing else hides the latency).
    uint32_t it_i[DDK_ITEMS], it_v[DDK_ITEMS];
    int32_t it_y[DDK_ITEMS];
    bool it_on[DDK_ITEMS];
#pragma unroll
    for (uint32_t k = 0; k < DDK_ITEMS; ++k) {
        const uint32_t t = threadIdx.x + k * DDK_THREADS;
        it_i[k] = t / (vpr + 1u); it_v[k] = t % (vpr + 1u);
        it_y[k] = y0 - (int32_t)DDK_K + (int32_t)it_i[k];
        it_on[k] = it_i[k] < DDK_ROWS && it_y[k] >= 0 && it_y[k] < (int32_t)Y;
    }
    auto active = [&](uint32_t k, uint32_t j) -> bool {        // frame j of this launch needs rows [y0 - halo, y0 + R + halo), halo = nframes - 1 - j
        const int32_t halo = (int32_t)(nframes - 1u - j);
        return it_on[k] && it_y[k] >= y0 - halo && it_y[k] < y0 + (int32_t)DDK_R + halo;
    };
    auto fetch = [&](uint32_t k, uint32_t j) -> uint4 {
        const uint16_t* src = in + (uint64_t)(z0 + j) * frame + (uint64_t)it_y[k] * X;
        if (it_v[k] < vpr) return *reinterpret_cast<const uint4*>(src + it_v[k] * 8u);
        return make_uint4(w < X ? (uint32_t)src[w] : 0u, 0u, 0u, 0u);
    };
    // 1. the decoded frame z0 - 1 (z0 >= 1), all image rows
    {
        const uint16_t* prev = out + (uint64_t)(z0 - 1u) * frame;
#pragma unroll
        for (uint32_t k = 0; k < DDK_ITEMS; ++k) {
            if (!it_on[k]) continue;
            uint16_t* d = img[0] + (size_t)it_i[k] * pitch + 8u;
            if (it_v[k] < vpr) *reinterpret_cast<uint4*>(d + it_v[k] * 8u) = *reinterpret_cast<const uint4*>(prev + (uint64_t)it_y[k] * X + it_v[k] * 8u);
            else d[w] = w < X ? prev[(uint64_t)it_y[k] * X + w] : (uint16_t)0;
        }
    }
    // (measured: ONE frame ahead, 34 us per launch of 8 frames; all 8 frames' voxels fetched up front, 16 loads in flight per thread:
    // 41 us -- the column copy next to this chain keeps the HBM busy, 0.75 ms of the stage's 1.1 ms are that copy's)
    uint4 cin[DDK_ITEMS];
#pragma unroll
    for (uint32_t k = 0; k < DDK_ITEMS; ++k) cin[k] = active(k, 0) ? fetch(k, 0) : make_uint4(0, 0, 0, 0);
    __syncthreads();
    int cur = 0;
    for (uint32_t j = 0; j < nframes; ++j) {
        const uint32_t z = z0 + j;
        uint4 nin[DDK_ITEMS];
#pragma unroll
        for (uint32_t k = 0; k < DDK_ITEMS; ++k) nin[k] = (j + 1u < nframes && active(k, j + 1u)) ? fetch(k, j + 1u) : make_uint4(0, 0, 0, 0);
        uint16_t* dst = out + (uint64_t)z * frame;
        const uint16_t* A = img[cur];
        uint16_t* B = img[cur ^ 1];
        const bool frame_touched = z >= 1u && z < zlim && hx > 0u;
#pragma unroll
        for (uint32_t k = 0; k < DDK_ITEMS; ++k) {
            if (!active(k, j)) continue;
            const uint32_t i = it_i[k], v = it_v[k];
            const int32_t y = it_y[k];
            if (v == vpr) {                                    // column w is never rewritten (w >= 2 + hx, or w == X and nobody reads it)
                B[(size_t)i * pitch + 8u + w] = (uint16_t)cin[k].x;
                continue;
            }
            const uint32_t x0 = v * 8u;
            const uint4 c = cin[k];
            uint32_t val[8] = {c.x & 0xffffu, c.x >> 16, c.y & 0xffffu, c.y >> 16, c.z & 0xffffu, c.z >> 16, c.w & 0xffffu, c.w >> 16};
            if (frame_touched && y >= 1 && y + 2 <= (int32_t)Y && x0 < 1u + hx) {
                uint32_t colsum[10];
#pragma unroll
                for (int q = 0; q < 10; ++q) colsum[q] = 0;
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    const uint16_t* r = A + (size_t)((int32_t)i + dy) * pitch + 8u + x0;    // columns x0 - 1 .. x0 + 8 of the decoded frame z - 1
                    const uint4 e = *reinterpret_cast<const uint4*>(r);
                    colsum[0] += r[-1];
                    colsum[1] += e.x & 0xffffu; colsum[2] += e.x >> 16; colsum[3] += e.y & 0xffffu; colsum[4] += e.y >> 16;
                    colsum[5] += e.z & 0xffffu; colsum[6] += e.z >> 16; colsum[7] += e.w & 0xffffu; colsum[8] += e.w >> 16;
                    colsum[9] += r[8];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t x = x0 + (uint32_t)q;
                    if (x >= 1u && x < 1u + hx) {
                        const uint32_t sum = (colsum[q] + colsum[q + 1] + colsum[q + 2]) & 0xffffu;     // the reference's sum wraps in 16 bits
                        val[q] = (val[q] + ((sum * 58255u) >> 19)) & 0xffffu;                          // + sum / 9
                    }
                }
            }
            const uint4 o = make_uint4(val[0] | (val[1] << 16), val[2] | (val[3] << 16), val[4] | (val[5] << 16), val[6] | (val[7] << 16));
            *reinterpret_cast<uint4*>(B + (size_t)i * pitch + 8u + x0) = o;
            if (y >= y0 && y < y0 + (int32_t)DDK_R) *reinterpret_cast<uint4*>(dst + (uint64_t)y * X + x0) = o;
        }
        __syncthreads();
        cur ^= 1;
#pragma unroll
        for (uint32_t k = 0; k < DDK_ITEMS; ++k) cin[k] = nin[k];
    }
}

// ------------------------------------------------------------------------------------------------
// LZ4 block compressor: one wavefront per chunk, hash table (4096 x u32) in LDS.
//
// Greedy parse of liblz4 1.9.3's LZ4_compress_generic (byU32, limitedOutput, acceleration 1) -- the
// parse is sequential by definition, so the wavefront evaluates 64 consecutive PROBES of the search
// loop at once and commits the prefix up to the first probe that matches:
//   probe u (unified index) sits at position q_u; q_0 is the "test next position" probe that follows a
//   match (equivalent to a search probe with zero literals), q_1.. are the search probes with upstream's
//   step schedule  step = searchMatchNb++ >> 6.
//   - table reads see the state before the batch; probes that share a hash bucket with an EARLIER probe
//     of the same batch ("hazard" lanes, found with a flagged ds_max through the table itself) take that
//     probe as candidate instead, resolved in lane order;
//   - the table is then restored and the committed prefix re-inserted with ds_max (positions grow).
// Match extension, backward catch-up, literal copies and token emission are wave-parallel.
// Two implementations of the batch share the loop: the LEAN loop for the batch right after a match (15 probes + the
// pending ip-2 insertion in one round of LDS reads, see there) and the GENERIC path (64 probes, any step schedule).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t LZ4_MFLIMIT = 12, LZ4_LASTLITERALS = 5, LZ4_MINLENGTH = 13, LZ4_MAXD = 65535;

__device__ __forceinline__ uint32_t lz4_hash5(uint64_t seq)
{
    return (uint32_t)(((seq << 24) * 889523592379ULL) >> 52);
}

// the same hash from the low dword and the 5th byte without 64-bit multiplies:
// ((seq << 24) * P) >> 52 = bits 28..39 of (seq40 * P) mod 2^40, P = 0xCF1BBCDCBB
__device__ __forceinline__ uint32_t lz4_hash5_32(uint32_t lo, uint32_t byte4)
{
    const uint32_t plo = 0x1BBCDCBBu;
    const uint32_t l = lo * plo;
    uint32_t h = __umulhi(lo, plo);
    h += (lo & 0xffu) * 0xCFu + (byte4 & 0xffu) * 0xBBu;      // only bits 0..7 of this sum reach the result
    return ((h << 4) | (l >> 28)) & 0xfffu;
}

// Sliding window of the chunk's source bytes in LDS (ring of WIN bytes + a 16-byte mirror of its head so
// that unaligned 16-byte reads never have to split at the wrap, filled FB bytes at a time).
// The parse only ever needs bytes around ip (probe sequences, catch-up, match extension, literals) and
// candidates a short distance back; everything resident is read with LDS latency instead of a dependent
// HBM/L2 round trip.  Positions outside [wlo, whi) -- far candidates, skip-accelerated probes on
// incompressible data, very long matches -- fall back to global loads.
// The block [whi, whi+FB) is always on its way: one global_load_lds_dwordx4 (64 lanes x 16 B = 1 KiB straight
// into the ring slot, no registers).  The instruction is issued through inline asm so that the compiler's
// s_waitcnt bookkeeping does not see it -- otherwise every ring read in the parse loop waits for the block in
// flight (vmcnt(0)) and the prefetch hides nothing.  commit() waits for it explicitly when ip gets within AHEAD
// bytes of whi, one refill later.  (The compiler's own vmcnt(N) waits only become stricter by the extra
// outstanding operation, never too weak: VMEM operations retire in order.)
// The first pass keeps the ring at 8 KiB (26 KiB of LDS per chunk wave = 6 waves per CU).  The DENSE kernel, which only
// runs the chunks the first pass gave up, takes 32 KiB: with 64 probes per batch nearly every batch would otherwise have
// a candidate behind the ring, i.e. a global round trip per batch.
#ifdef SQY_EXP_STATS
// EXPERIMENT BUILDS ONLY (tools/exp_stats.py; SQY_EXTRA_FLAGS=-DSQY_EXP_STATS): every parse wave of the first pass logs when it ran and
// what it waited for -- g_exp_buf[0] = records written, then records of eight 64-bit words.
__device__ unsigned long long* g_exp_buf;
__device__ unsigned long long g_exp_cap;
extern "C" __attribute__((visibility("default"))) int SQYAMD_Exp_Set_Buffer(void* p, unsigned long long cap_records)
{
    unsigned long long* q = static_cast<unsigned long long*>(p);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_exp_buf), &q, sizeof q) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_exp_cap), &cap_records, sizeof cap_records) != hipSuccess;
}
#define SQY_EXP(...) __VA_ARGS__
#else
#define SQY_EXP(...)
#endif
constexpr uint32_t LZ4_RINGLESS_U = 64u * 15u;   // probes since the last match from which on the batches stride over the ring (step >= 16)
#ifndef SQY_EXP_WIN_LEAN
#define SQY_EXP_WIN_LEAN 8192
#endif
#ifndef SQY_EXP_DG_AHEAD
#define SQY_EXP_DG_AHEAD 8
#endif
constexpr uint32_t LZ4_WIN_LEAN = SQY_EXP_WIN_LEAN, LZ4_WIN_DENSE = 32768, LZ4_FB = 1024, LZ4_AHEAD = 2048, LZ4_MIRROR = 16;

template <uint32_t LZ4_WIN>
struct Lz4WindowT {
    glb_u8* src;           // chunk source (global)
    lds_u8* win;           // LDS ring (WIN + MIRROR bytes)
    uint32_t n;            // chunk bytes
    uint32_t whi, wlo;     // resident range [wlo, whi), whi a multiple of FB
    uint32_t nif;          // blocks [whi, whi + nif * FB) are in flight into their slots (1 in the steady state)
    uint32_t lane16;       // lane * 16
    uint32_t pmin;         // first position of the block being parsed (0, or 65536 in a block-linked frame): the ring never holds less
    SQY_EXP(unsigned long long x_wait = 0; uint32_t x_nwait = 0;)

    __device__ __forceinline__ void issue()
    {
        const uint32_t b = whi + nif * LZ4_FB;                 // block to fetch
        // its slot still holds [b - WIN, b - WIN + FB): give that up before the copy starts
        if (b + LZ4_FB - wlo > LZ4_WIN) wlo = b + LZ4_FB - LZ4_WIN;
        const uint32_t a = b + lane16;
        if (a + 16u <= n) {                                    // lanes past the chunk's last whole 16 bytes stay off
            const uint32_t lds_dst = (uint32_t)(uintptr_t)win + (b & (LZ4_WIN - 1));   // wave-uniform; the copy adds lane * 16
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src + a), "s"(lds_dst) : "memory");
        }
        nif += 1;
    }
    // everything in flight has landed and becomes readable
    __device__ __forceinline__ void commit()
    {
        SQY_EXP(const unsigned long long x_t0 = __builtin_amdgcn_s_memtime();)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SQY_EXP(x_wait += __builtin_amdgcn_s_memtime() - x_t0; x_nwait += 1;)
        wave_lds_sync();
        const uint32_t end = whi + nif * LZ4_FB;
        if (((end - 1u) & ~(LZ4_WIN - 1)) != ((whi - 1u) & ~(LZ4_WIN - 1)) || whi == 0) {   // a block with ring offset 0 among them
            if (lane16 < 64u) *reinterpret_cast<SQY_LDS uint32_t*>(win + LZ4_WIN + (lane16 >> 2)) = *reinterpret_cast<const SQY_LDS uint32_t*>(win + (lane16 >> 2));
            wave_lds_sync();
        }
        whi = end;
        nif = 0;
    }
    // make [ip - some history, ip + AHEAD) resident as far as the chunk goes (uniform control flow)
    __device__ __forceinline__ void ensure(uint32_t ip)
    {
        if (ip + LZ4_AHEAD <= whi || whi >= n) return;
        if (ip >= whi + (LZ4_WIN - LZ4_FB)) {                 // jumped past everything resident: restart the ring
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (a copy in flight must not land on top of the new ones)
            uint32_t start = ip & ~(LZ4_FB - 1);
            if (start >= pmin + LZ4_FB) start -= LZ4_FB;       // keep one block of history
            whi = wlo = start;
            nif = 0;
        }
        // fetch what is missing in one go (one wait), then leave the next block in flight
        while (whi + nif * LZ4_FB < ip + LZ4_AHEAD && whi + nif * LZ4_FB < n) issue();
        commit();
        if (whi < n) issue();
    }
    // end of the bytes that can be read from the ring: the last (n & 15) bytes are never filled
    __device__ __forceinline__ uint32_t hi_valid() const { return whi < (n & ~15u) ? whi : (n & ~15u); }
    __device__ __forceinline__ bool inside(uint32_t p, uint32_t len) const { return p >= wlo && p + len <= hi_valid(); }
    __device__ __forceinline__ const lds_u8* at(uint32_t p) const { return win + (p & (LZ4_WIN - 1)); }

    // ring-only reads (caller has established residency)
    __device__ __forceinline__ uint4 lds128(uint32_t p) const { return lds_ld_u128(at(p)); }
    __device__ __forceinline__ uint32_t lds32(uint32_t p) const { return lds_ld_u32(at(p)); }
    __device__ __forceinline__ uint32_t lds8(uint32_t p) const { return lds_ld_u8(at(p)); }

    // per-lane "ring or global" reads; the two loads live in different address spaces and stay separate
    __device__ __forceinline__ uint64_t rd64(uint32_t p) const
    {
        uint64_t v;
        if (inside(p, 8)) v = lds_ld_u64(at(p)); else v = glb_ld_u64(src + p);
        return v;
    }
    __device__ __forceinline__ uint32_t rd32(uint32_t p) const
    {
        uint32_t v;
        if (inside(p, 4)) v = lds_ld_u32(at(p)); else v = glb_ld_u32(src + p);
        return v;
    }
    __device__ __forceinline__ uint32_t rd8(uint32_t p) const
    {
        uint32_t v;
        if (inside(p, 1)) v = lds_ld_u8(at(p)); else v = glb_ld_u8(src + p);
        return v;
    }
    __device__ __forceinline__ uint4 rd128(uint32_t p) const   // caller guarantees p + 16 <= n
    {
        uint4 v;
        if (inside(p, 16)) v = lds_ld_u128(at(p)); else v = glb_ld_u128(src + p);
        return v;
    }
};

// copy `len` bytes (uniform) of the chunk starting at position `from` to d, any alignment; all 64 lanes call it
template <class Lz4Window>
__device__ __forceinline__ void wave_copy(uint8_t* __restrict__ d, const Lz4Window& w, uint32_t from, uint32_t len, int lane)
{
    if (len <= 64) {
        if ((uint32_t)lane < len) d[lane] = (uint8_t)w.rd8(from + lane);
        return;
    }
    // head: bring d to 16-byte alignment
    const uint32_t head = (uint32_t)((16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15);
    if ((uint32_t)lane < head) d[lane] = (uint8_t)w.rd8(from + lane);
    d += head; from += head; len -= head;
    const uint32_t nvec = len >> 4;
    for (uint32_t i = lane; i < nvec; i += 64) {
        *reinterpret_cast<uint4*>(d + (size_t)i * 16) = w.rd128(from + i * 16);   // from + 16 i + 16 <= from + len <= n
    }
    const uint32_t done = nvec << 4;
    if ((uint32_t)lane < len - done) d[done + lane] = (uint8_t)w.rd8(from + done + lane);
}

// index of the first differing byte of two 16-byte values (16 when equal)
// v_ffbl_b32 itself: bit index of the lowest set bit, ~0 for 0.  (__builtin_ffs(v) - 1 means the same, but the compiler does
// not know that the instruction already yields ~0 for 0 and adds a compare and a select per dword -- 30 instead of 16
// instructions on the parse's critical path.)
__device__ __forceinline__ uint32_t ffbl(uint32_t v)
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// the same from the four dwords of x ^ y
__device__ __forceinline__ uint32_t first_nonzero16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
    const uint32_t t0 = ffbl(x0);
    const uint32_t t1 = ffbl(x1) | 32u;
    const uint32_t t2 = ffbl(x2) | 64u;
    const uint32_t t3 = ffbl(x3) | 96u;
    const uint32_t a = t0 < t1 ? t0 : t1, b = t2 < t3 ? t2 : t3;
    const uint32_t m = a < b ? a : b;
    return (m < 128u ? m : 128u) >> 3;
}
__device__ __forceinline__ uint32_t first_diff16(uint4 x, uint4 y)
{
    // "| 32 k" adds the dword offset to a real index and leaves the "no difference in this dword" marker (~0) above every
    // index.  Straight-line on purpose: the branchy form (low half first, high half only if equal) costs two exec-mask
    // regions per call.
    const uint32_t t0 = ffbl(x.x ^ y.x);
    const uint32_t t1 = ffbl(x.y ^ y.y) | 32u;
    const uint32_t t2 = ffbl(x.z ^ y.z) | 64u;
    const uint32_t t3 = ffbl(x.w ^ y.w) | 96u;
    const uint32_t a = t0 < t1 ? t0 : t1, b = t2 < t3 ? t2 : t3;
    const uint32_t m = a < b ? a : b;
    return (m < 128u ? m : 128u) >> 3;
}

// number of equal leading bytes of chunk[a..a+maxlen) and chunk[b..b+maxlen), maxlen <= 16, b < a
template <class Lz4Window>
__device__ __forceinline__ uint32_t common16(const Lz4Window& w, uint32_t a, uint32_t b, uint32_t maxlen)
{
    if (a + 16u <= w.n) {
        const uint32_t c = first_diff16(w.rd128(a), w.rd128(b));
        return c < maxlen ? c : maxlen;
    }
    uint32_t c = 0;
    while (c < maxlen && glb_ld_u8(w.src + a + c) == glb_ld_u8(w.src + b + c)) ++c;
    return c;
}

// Compressed bytes are staged in LDS and written out in 16-byte pieces when the stage fills: the parse then
// issues almost no global stores, so the s_waitcnt vmcnt(0) in front of the occasional global LOAD (far
// candidate, window refill) no longer queues behind a stream of tiny stores on the critical path.
constexpr uint32_t LZ4_OB = 2048;

struct Lz4Out {
    SQY_GLB uint8_t* dst;        // chunk's scratch (global)
    lds_u8* ob;                  // LDS stage
    uint32_t base;               // dst offset of ob[0]; bytes [base, op) live in the stage
    int lane;

    // write everything staged; afterwards base == op
    __device__ __forceinline__ void flush(uint32_t op)
    {
        wave_lds_sync();                                       // staged bytes were written by other lanes
        const uint32_t cnt = op - base;
        const uint32_t nvec = cnt >> 4;
        for (uint32_t i = lane; i < nvec; i += 64) {
            const v4u v = *reinterpret_cast<const SQY_LDS v4u*>(ob + i * 16u);
            SQY_GLB pk_u128* q = reinterpret_cast<SQY_GLB pk_u128*>(dst + base + i * 16u);
            q->x = v.x; q->y = v.y; q->z = v.z; q->w = v.w;
        }
        const uint32_t done = nvec << 4;
        if ((uint32_t)lane < cnt - done) dst[base + done + lane] = ob[done + lane];
        base = op;
    }
    // room for `len` more staged bytes at op (uniform)
    __device__ __forceinline__ void reserve(uint32_t op, uint32_t len)
    {
        if (op - base + len > LZ4_OB) flush(op);
    }
    __device__ __forceinline__ lds_u8* at(uint32_t o) const { return ob + (o - base); }
};

// SQY_LZ4_DIAG: defined only by tools/lz4_diag.hip, which #includes this file into a stand-alone diagnostic program; the
// product build (sqeazy_amd/build.py passes -DSQY_PRODUCT_BUILD and nothing else) cannot turn it on.  Cycles of ONE
// region of the parse loop per build, between mark SQY_DIAG_A and mark SQY_DIAG_B (s_memtime; one region at a time keeps
// the probe's own register and issue cost out of the number), plus event counters.
#if defined(SQY_PRODUCT_BUILD) && defined(SQY_LZ4_DIAG)
#error "SQY_LZ4_DIAG is a tools/ build switch; libsqeazy_amd.so has exactly one configuration"
#endif
#ifdef SQY_LZ4_DIAG
#define SQY_DIAG_ARG , unsigned long long* __restrict__ diag
#define SQY_REASON(i) do { dreason[i] += 1; } while (0)
#define SQY_STAMP(i) do { \
        if ((i) == SQY_DIAG_A) { __builtin_amdgcn_sched_barrier(0); dt0 = __builtin_amdgcn_s_memtime(); darmed = true; __builtin_amdgcn_sched_barrier(0); } \
        else if ((i) == SQY_DIAG_B) { __builtin_amdgcn_sched_barrier(0); if (darmed) { dacc += __builtin_amdgcn_s_memtime() - dt0; dcnt += 1; darmed = false; } __builtin_amdgcn_sched_barrier(0); } \
    } while (0)
#else
#define SQY_DIAG_ARG
#define SQY_STAMP(i) do { } while (0)
#define SQY_REASON(i) do { } while (0)
#endif

// One block of a block-linked frame (LINKED kernel): where it sits in the stream and how far liblz4's backward catch-up
// may move a match that starts inside the block (low_in) or in the history in front of it (low_dict) -- the one place
// where liblz4's prefix and external-dictionary modes differ (host: sqy::lz4_plan_blocks, LZ4F's buffer management).
// (struct Lz4Block: sqy_kernels.h)
constexpr uint32_t LZ4_HIST = 65536;     // LINKED: a block is parsed at positions [HIST, HIST + n), its history sits below

// Round 4: the duplicate decision of a chunk (lz4_dedupe_verify_kernel's work: look the chunk's key up, compare the bytes with the
// first chunk of that key, fill the chunk's holes) done by the chunk's own parse wavefront in front of its parse.  The verify kernel
// was 0.1 ms of HBM traffic on its own and 0.28 ms of every call's chain with other calls' transposes next to it; a wavefront does
// the same for its one chunk in a few microseconds, and the parse of the other chunks no longer waits for the slowest compare.
// (struct Lz4DedupeArgs: sqy_kernels.h.)  Returns true when the chunk is a duplicate (dup_of[k] = the chunk it equals): nothing to parse.
// noinline: the registers of this cold code stay out of the parse loop's allocation.
__device__ __attribute__((noinline)) bool lz4_chunk_dedupe(const uint8_t* __restrict__ in, uint32_t chunk, uint64_t in_stride, uint64_t total,
                                                           uint64_t k, const Lz4DedupeArgs& dd, uint32_t lane)
{
    uint32_t r = (uint32_t)k;
    if (k < dd.nchunks_full) {
        if (lane == 0) {
            const uint64_t key = dd.chunk_key[k];
            uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & dd.tab_mask;
            for (uint32_t tries = 0; tries <= dd.tab_mask; ++tries, slot = (slot + 1u) & dd.tab_mask) {
                const uint64_t tk = dd.tab_key[slot];
                if (tk == key) { r = dd.tab_val[slot]; break; }
                if (tk == 0) break;
            }
        }
        r = sgpr(r);
    }
    // the all-zero 1 KiB pieces the transpose left unwritten (see lz4_dedupe_verify_kernel): one 1 KiB store per hole
    auto fill_holes = [&]() {
        if (!dd.holes_map) return;
        const uint64_t left = total - k * chunk;
        const uint32_t np = (uint32_t)((left < chunk ? left : chunk) >> 10);
        const uint64_t* hm = dd.holes_map + k * (1u + ((chunk >> 10) + 63u) / 64u);
        if ((uint32_t)hm[0] == 0u) return;
        uint8_t* body = const_cast<uint8_t*>(in) + k * in_stride;
        for (uint32_t w0 = 0; w0 < np; w0 += 64) {
            uint64_t m = hm[1u + w0 / 64u];
            while (m) {
                const uint32_t pc = w0 + ctz64(m);
                m &= m - 1;
                const v4u zero = {0, 0, 0, 0};
                *reinterpret_cast<v4u_any*>(body + (uint64_t)pc * 1024u + lane * 16u) = zero;
            }
        }
    };
    if (r >= k) { if (lane == 0) dd.dup_of[k] = (uint32_t)k; fill_holes(); return false; }     // first of its kind
    if (dd.chunk_key[k] == 1ull) { if (lane == 0) dd.dup_of[k] = r; return true; }           // all zero, exactly: the first all-zero chunk's twin
    fill_holes();
    // byte compare with chunk r; pieces whose hash is the zero marker count as zeros, written or not (64 lanes x 16 bytes = one piece)
    const v4u_any* a = reinterpret_cast<const v4u_any*>(in + k * in_stride);
    const v4u_any* b = reinterpret_cast<const v4u_any*>(in + (uint64_t)r * in_stride);
    const uint4* hk = reinterpret_cast<const uint4*>(dd.piece_hash) + k * (chunk >> 10);
    const uint4* hr = reinterpret_cast<const uint4*>(dd.piece_hash) + (uint64_t)r * (chunk >> 10);
    const uint32_t npieces = chunk >> 10;
    uint32_t diff = 0;
    for (uint32_t p0 = 0; p0 < npieces && !diff; p0 += 4) {             // four pieces (4 KiB per side) in flight
        v4u x[4], y[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint32_t pc = p0 + u;
            const v4u zero = {0, 0, 0, 0};
            bool za = true, zb = true;
            if (pc < npieces) {
                za = zb = false;
                if (dd.holes_map) {
                    const uint4 qa = hk[pc], qb = hr[pc];
                    za = (qa.x | qa.y | qa.z | qa.w) == 0u;
                    zb = (qb.x | qb.y | qb.z | qb.w) == 0u;
                }
            }
            x[u] = za ? zero : (v4u)a[pc * 64u + lane];
            y[u] = zb ? zero : (v4u)b[pc * 64u + lane];
        }
        uint32_t d = 0;
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) d |= (x[u].x ^ y[u].x) | (x[u].y ^ y[u].y) | (x[u].z ^ y[u].z) | (x[u].w ^ y[u].w);
        diff = ballot(d != 0u) != 0ull ? 1u : 0u;
    }
    if (lane == 0) dd.dup_of[k] = diff ? (uint32_t)k : r;
    return diff == 0u;
}

// LINKED = false: one independent chunk per wavefront (the chunked layout's single-block frames).
// LINKED = true: wavefront f walks the blocks [frame_first[f], frame_first[f+1]) of one frame in order; the table lives on
// across blocks (positions are re-based by the previous block's size, entries that fall more than 64 KiB behind the new
// block die), candidates may sit in the 64 KiB in front of the block (far fetches from global memory).
// DENSE = false: the first pass over every chunk.  A chunk that turns out to be a stream of short sequences (32 matches
// of less than 16 bytes in a row) is given up -- it is appended to the redo list (redo[0] = count, redo[1..] = chunks) --
// and parsed again by the DENSE = true kernel, which resolves several sequences per batch; the host launches that one over
// exactly the listed chunks, and only when there are any.  Two kernels instead of one keep the lean loop of the first pass
// free of the dense batches' registers (and the dense batches free to use a larger window).
// ACCEL = true (round 4): liblz4's acceleration above 1 (sqeazy's lz4(accel=-k): LZ4F turns a negative compression level into
// acceleration k + 1, lz4frame.c LZ4F_compressBlock / _continue) -- the search then strides: searchMatchNb starts at
// acceleration << 6, so the increments between probes are 1, 1, a, a, .. (a = acceleration) growing by one every 64 probes.
// A separate instantiation that only takes the generic path with the general probe positions (the lean loop, the dense
// batches and the no-hit batches are built on the stride-1 start and stay out): exact, not fast -- a rarely used setting --
// and the acceleration-1 kernels keep their code.
// (the last argument: the duplicate search's tables for the chunked layout, the block-parallel walk's for block-linked frames)
template <bool LINKED> struct Lz4ExtraArgs { typedef Lz4DedupeArgs type; };
template <> struct Lz4ExtraArgs<true> { typedef Lz4SpecArgs type; };
__device__ __forceinline__ uint32_t lz4_linked_tag_shift(uint32_t max_block)
{
    const uint32_t pmax = LZ4_HIST + max_block;
    return 31u - (32u - (uint32_t)__builtin_clz(pmax - 1));
}

template <bool LINKED, bool DENSE, bool ACCEL = false>
__global__ __launch_bounds__(64)
void lz4_chunks_kernel(const uint8_t* __restrict__ in, uint64_t total, uint32_t chunk, uint64_t in_stride,
                       uint8_t* __restrict__ scratch, uint64_t stride, uint32_t* __restrict__ csize,
                       const uint64_t* __restrict__ fmap, uint64_t fbytes,
                       const Lz4Block* __restrict__ blocks, const uint32_t* __restrict__ frame_first, uint32_t max_block,
                       uint32_t* __restrict__ redo_list, const uint32_t* __restrict__ dup_of, uint32_t accel, typename Lz4ExtraArgs<LINKED>::type dd SQY_DIAG_ARG)
{
#ifdef SQY_LZ4_DIAG
    unsigned long long dacc = 0, dt0 = 0, dcnt = 0, dreason[16] = {0};
    bool darmed = false;
#endif
    __shared__ uint32_t table[4096];
    constexpr uint32_t LZ4_WIN = DENSE ? LZ4_WIN_DENSE : LZ4_WIN_LEAN;
    typedef Lz4WindowT<LZ4_WIN> Lz4Window;
    __shared__ __attribute__((aligned(16))) uint8_t ring[LZ4_WIN + LZ4_MIRROR];
    __shared__ __attribute__((aligned(16))) uint8_t stage[LZ4_OB];
    const int lane = threadIdx.x;
    if (!LINKED && !DENSE && dup_of && dup_of[blockIdx.x] != blockIdx.x) return;     // byte-identical to an earlier chunk (lz4_dedupe_*): its frame is that chunk's
    if constexpr (!LINKED && !DENSE && !ACCEL) if (dd.chunk_key) {                     // the same decision, made here (round 4)
        if (lz4_chunk_dedupe(in, chunk, in_stride, total, blockIdx.x, dd, threadIdx.x)) return;
        // The holes filled above are read back below by this same wavefront (loads and LDS-DMA): its stores have to have left the
        // wave (vmcnt) -- nothing more: nobody has read those addresses since the kernel began, so no cache holds an older copy.
        // (An agent-scope release here writes back the whole XCD's L2 once per chunk, next to other calls' transposes filling it:
        // measured, the bench lost a quarter.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SQY_EXP(const unsigned long long x_start = __builtin_amdgcn_s_memrealtime(); const unsigned long long x_c0 = __builtin_amdgcn_s_memtime();
            unsigned long long x_far = 0, x_commit = 0; uint32_t x_nfar = 0, x_ncommit = 0, x_nseq = 0, x_ngen = 0;
            unsigned long long x_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long x_last = x_c0;
            auto x_lap = [&](int i) { const unsigned long long t = __builtin_amdgcn_s_memtime(); x_acc[i] += t - x_last; x_last = t; };)
    uint32_t b_first, b_last;
    uint32_t b_out = 0;                                        // LINKED: the first block of the walk whose output counts
    uint32_t spec_mode = 0;
    if constexpr (LINKED) {
        spec_mode = dd.mode;
        if (spec_mode) {
            b_first = dd.wave_first[blockIdx.x];
            b_last = dd.wave_last[blockIdx.x] + 1u;
            b_out = spec_mode == 1u ? b_last - 1u : b_first;
        } else {
            b_first = frame_first[blockIdx.x];
            b_last = frame_first[blockIdx.x + 1];
            b_out = b_first;
        }
    } else {
        b_first = DENSE ? redo_list[1 + blockIdx.x] : blockIdx.x;
        b_last = b_first + 1;
    }
    uint32_t n_prev = 0;
  for (uint32_t bi = b_first; bi < b_last; ++bi) {
    // (a warm-up block's output lands where the counted block's will, and is overwritten by it)
    const uint64_t blk = LINKED && bi < b_out ? b_last - 1u : bi;
    // position of the block's first byte (p0) and of the stream's first byte (p_lo) in the block's own coordinates
    const uint32_t p0 = LINKED ? LZ4_HIST : 0u;
    uint32_t n, p_lo = 0;
    int64_t low_in = 0, low_dict = 0;                          // catch-up limits, same coordinates
    bool fresh = true;
    const uint8_t* __restrict__ src;
    if (LINKED) {
        const Lz4Block bd = blocks[bi];
        n = bd.n;
        src = in + bd.start - LZ4_HIST;                        // (never dereferenced below `in`: p_lo)
        p_lo = bd.start >= LZ4_HIST ? 0u : (uint32_t)(LZ4_HIST - bd.start);
        low_in = bd.low_in - (int64_t)bd.start + LZ4_HIST;
        low_dict = bd.low_dict - (int64_t)bd.start + LZ4_HIST;
        fresh = (bd.flags & 1u) != 0;
        if constexpr (LINKED) {
            if (spec_mode == 1u && bi == b_first) fresh = true;                  // the guess: nothing older than this walk matters
            if (spec_mode == 2u && bi == b_first && !fresh) {                   // the table the block in front really left
                const uint32_t* __restrict__ tf = dd.tables + (uint64_t)(bi - 1u) * kLz4SpecTableWords + 4096u;
#pragma unroll 4
                for (int i = 0; i < 64; ++i) table[i * 64 + lane] = tf[i * 64 + lane];
                n_prev = blocks[bi - 1u].n;
            }
        }
    } else {
        // frame_shuffle in front of the sink: the stream is the frames of `in` in the order fmap gives (a chunk never straddles
        // two frames, the host checks fbytes % chunk == 0), read in place instead of gathered into a copy first
        const uint64_t lin = blk * chunk;
        // (in_stride: chunk k of the stream starts at in + k * in_stride -- chunk bytes apart, or chunk + 15 when the stage in
        // front wrote the stream as frame bodies in place)
        src = fmap ? in + fmap[lin / fbytes] * fbytes + lin % fbytes : in + blk * in_stride;
        const uint64_t left = total - blk * chunk;
        n = (uint32_t)(left < chunk ? left : chunk);
    }
    const uint32_t pend = p0 + n;                              // end of the block
    // bytes the backward catch-up may take on the match side (liblz4: match > lowLimit)
    auto back_room = [&](uint32_t mt) -> uint32_t {
        if (!LINKED) return mt;
        const int64_t r = (int64_t)mt - (mt >= p0 ? low_in : low_dict);
        return r < 0 ? 0u : (r > 0x7fffffff ? 0x7fffffffu : (uint32_t)r);
    };
    uint8_t* __restrict__ dst = scratch + blk * stride;

    Lz4Window w;
    w.src = (glb_u8*)src; w.win = (lds_u8*)ring; w.n = pend; w.whi = p0; w.wlo = p0; w.nif = 0; w.lane16 = (uint32_t)lane * 16u;
    w.pmin = p0;
    w.issue();
    Lz4Out o;
    o.dst = (SQY_GLB uint8_t*)dst; o.ob = (lds_u8*)stage; o.base = 0; o.lane = lane;
    // Table entry = position << tsh | tag, tag = tsh-bit hash of the 4 bytes at that position.  Positions stay in the
    // high bits, so entries order like positions (ds_max commit, flag bit 31 free) and a tag mismatch proves that the
    // candidate's first 4 bytes differ -- no read of a far candidate just to reject it.  An empty bucket means
    // "position 0" in liblz4, so the table starts out as entry(first byte of the stream).
    const uint32_t pmax = LINKED ? LZ4_HIST + max_block : n;
    const uint32_t pos_bits = pmax > 1 ? 32u - (uint32_t)__builtin_clz(pmax - 1) : 1u;
    const uint32_t tsh = 31u - pos_bits;                       // >= 8 for blocks up to 4 MiB
    const uint32_t tmask = (1u << tsh) - 1u;
    auto tag_of = [&](uint32_t seq32) -> uint32_t { return (seq32 * 2654435761u) >> (32u - tsh); };
    if (fresh) {
        const uint32_t e0 = (p0 << tsh) | (n >= 4 ? tag_of(glb_ld_u32((glb_u8*)src + p0)) : 0u);
        uint4* t4 = reinterpret_cast<uint4*>(table);
#pragma unroll
        for (int i = 0; i < 16; ++i) t4[i * 64 + lane] = make_uint4(e0, e0, e0, e0);
        if constexpr (LINKED) if (spec_mode && bi >= b_out && !(blocks[bi].flags & 1u)) {
            // a guess without any warm-up (the host's knob): the table on record is the empty one, which no block leaves behind
            uint4* g4 = reinterpret_cast<uint4*>(dd.tables + (uint64_t)bi * kLz4SpecTableWords);
#pragma unroll
            for (int i = 0; i < 16; ++i) g4[i * 64 + lane] = make_uint4(e0, e0, e0, e0);
        }
    } else {
        // the stream moved on by n_prev bytes: positions shift down, whatever falls below 0 is more than 64 KiB behind
        // every position of this block ("too far" for good) and parks at position 0
#pragma unroll 4
        for (int i = 0; i < 64; ++i) {
            const uint32_t e = table[i * 64 + lane];
            const uint32_t pp = e >> tsh;
            const uint32_t e2 = pp >= n_prev ? (((pp - n_prev) << tsh) | (e & tmask)) : 0u;
            table[i * 64 + lane] = e2;
            if constexpr (LINKED) if (spec_mode && bi >= b_out) dd.tables[(uint64_t)bi * kLz4SpecTableWords + i * 64 + lane] = e2;
        }
    }
    __syncthreads();
    n_prev = n;

    // (round 6) the noise digest the transpose left for this chunk (see bitswap1_u16_regs): usable for a whole chunk none of whose 1 KiB
    // pieces is all zero (those write no entries: what lies there is a former call's)
    const uint32_t* __restrict__ dgp = nullptr;
    uint32_t dg_words = 0;
    // the digest's batches are fetched DG_AHEAD batches ahead of their use (round 6: a batch's entries are 256 bytes in a row, where they
    // lie is known as long as nothing is found, and with calls in flight a load takes a microsecond -- a chunk of noise spent three
    // quarters of its 120 us waiting for its 76 loads one after the other): dgq[0] holds the batch that starts at probe dgq_U
    constexpr int DG_AHEAD = SQY_EXP_DG_AHEAD;
    uint32_t dgq[DG_AHEAD];
    uint32_t dgq_U = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < DG_AHEAD; ++k) dgq[k] = 0u;
    if constexpr (!LINKED && !DENSE && !ACCEL) {
        if (dd.digest && n == chunk && dd.holes_map &&
            (uint32_t)dd.holes_map[blk * (1u + ((chunk >> 10) + 63u) / 64u)] == 0u) {
            dgp = dd.digest + blk * (uint64_t)dd.digest_stride;
            dg_words = dd.digest_stride;
        }
    }
    const uint32_t olimit = n - 1;       // capacity n-1 (LZ4F_makeBlock); offsets into dst
    uint32_t op = 0, anchor = p0;
    bool failed = false;
    bool redo_dense = false;             // first pass: this chunk is left to the DENSE kernel
    // (round 6, built, exact, measured, not kept: COUNT ONLY -- a chunk of noise with a short match every kilobyte or two, such as plane 8
    // of the bench stack where the shell is tangent, ends up stored, but until the output limit says so every sequence copies its
    // kilobytes of literals: 1.1 ms of such a chunk's 1.8 with calls in flight.  From the first long literal run that found the chunk
    // 256 bytes behind its input on, nothing was written any more, sizes and limit checks went on, a chunk that fit after all went onto
    // the dense kernel's list.  That chunk: 1.84 -> 0.75 ms in flight -- and the bench did not move (the transposes set the pace, not the
    // slowest chunk), the C3 slab's launch was 3 % slower for the extra code: profiles/r06_experiments.txt.)

    if (n >= LZ4_MINLENGTH) {
        const uint32_t mflimitPlusOne = pend - LZ4_MFLIMIT + 1;
        const uint32_t matchlimit = pend - LZ4_LASTLITERALS;
        if (LINKED && !fresh) {
            // LZ4_putPosition(ip) on the block's first byte (a no-op on a fresh table, which already says "first byte")
            const uint64_t s0 = glb_ld_u64((glb_u8*)src + p0);
            if (lane == 0) table[lz4_hash5(s0)] = (p0 << tsh) | tag_of((uint32_t)s0);
            wave_lds_sync();
        }
        uint32_t P = p0 + 1, U = 1;      // first probe of the block: search from ip = first byte + 1
        uint32_t put2 = 0xffffffffu;     // position whose hash has to enter the table before the next batch (ip - 2)

        // A sequence found by the lean path is written out one iteration LATER, between the issue of the next batch's
        // ring reads and their first use: its ~70 instructions then run in the shadow of that LDS round trip.
        // dense batches (several short sequences resolved from one batch of 64 probes, see below): switched on by the lean
        // loop after two short matches in a row, off again by anything a dense batch does not handle
        bool dense_next = DENSE;
        uint32_t shorts = 0;
        bool pend = false;
        uint32_t pe_lit = 0, pe_mcode = 0, pe_off = 0;
        uint32_t pe_litv = 0;            // per lane: literal byte k-1 for lane k
        // two halves, so that the lean loop can run each in the shadow of a different LDS round trip: the output-limit checks
        // and the room in the stage (scalar), then the bytes (one per lane)
        uint32_t pe_bytes = 0, pe_ext = 0;
        auto emit_pending_checks = [&]() {
            const uint32_t lit = pe_lit, matchCode = pe_mcode;
            pe_ext = (matchCode + 240u) / 255u;                                     // = (matchCode - 15) / 255 + 1 from 15 on, else 0
            pe_bytes = 1u + lit + 2u + pe_ext;                                      // <= 1 + 14 + 2 + 4
            // upstream's two limit checks; lit < 15 so lit/255 == 0 and there is no literal-length extension.  Both are
            // below op + 15 + 8 + 5 (at most 4 extension bytes for the matches the lean loop settles): far from the end of
            // the output nothing has to be looked at
            if (op + 33u > olimit &&
                (op + 1u + lit + (2 + 1 + LZ4_LASTLITERALS) > olimit ||
                 op + 1u + lit + 2u + (1 + LZ4_LASTLITERALS) + pe_ext > olimit)) { failed = true; return; }
            o.reserve(op, pe_bytes);
        };
        auto emit_pending_bytes = [&]() {
            pend = false;
            const uint32_t lit = pe_lit, matchCode = pe_mcode, ml_ext = pe_ext;
            // everything behind the literals as one little-endian value: offset, then ml_ext - 1 bytes of 255 and the rest
            // (ml_ext <= 4 here: matches of at most 1 KiB + catch-up).  No lane-dependent branches: the select chain this
            // replaces was compiled into exec-mask regions.
            const uint32_t rest = matchCode - 15u - (ml_ext - 1u) * 255u;           // (unused when ml_ext == 0)
            const uint64_t ones = (1ull << (8u * (ml_ext ? ml_ext - 1u : 0u))) - 1ull;
            const uint64_t tail = (uint64_t)pe_off | (ml_ext ? ((ones | ((uint64_t)rest << (8u * (ml_ext - 1u)))) << 16) : 0ull);
            const uint32_t k = (uint32_t)lane;
            const uint32_t j = k - (lit + 1u);                                      // byte of `tail` for the lanes behind the literals
            const uint32_t tb = (uint32_t)(tail >> (8u * (j & 7u))) & 0xffu;
            uint32_t v = (lit << 4) | (matchCode < 15u ? matchCode : 15u);
            v = (k - 1u) < lit ? pe_litv : v;
            v = k > lit ? tb : v;
            if (k < pe_bytes) *o.at(op + k) = (uint8_t)v;
            op += pe_bytes;
        };
        auto emit_pending = [&]() {
            emit_pending_checks();
            if (failed) { pend = false; return; }
            emit_pending_bytes();
        };

        for (;;) {
            SQY_STAMP(0);
            SQY_EXP(x_lap(0);)                                  // 0: set-up, loop top, everything not named below
            // Skip-accelerated probing (no match for a while: incompressible data) strides over the ring: at a step of 16 bytes a
            // batch of 64 probes spans a KiB, later several -- most probes lie behind the resident range and are read from global
            // memory anyway, and refilling the ring up to P + AHEAD would cost an HBM round trip per batch for bytes nobody reads.
            // The ring is left behind then; the first match restarts it (ensure's "jumped past everything" path).
            if (ACCEL || U < LZ4_RINGLESS_U) w.ensure(P);

            uint32_t f = 64, fcand = 0;      // first matching probe of the batch and its candidate
            uint32_t ipf = 0;                // its position
            uint32_t fwl = 0;                // forward bytes beyond MINMATCH for the winner: exact when fw_exact, else "at least"
            bool fw_exact = false;
            uint32_t bkl = 0xffffffffu;      // bytes known equal in front of (ip, match) for the winner (0..3 exact), ~0 = unknown
            bool batch_done = false;         // the lean loop found the winner and committed the table; the tail below finishes the match
            uint32_t nvalid = 64, next_P = 0;

            // ---------------------------------------------------------------------------------------
            // lean loop: batches right after a match (anchor == P, put2 == P-2) whose winner sits among the first 15
            // probes, with fewer than 15 literals and a match settled by the speculative 16-byte compare or one wide
            // round.  Straight-line; ring reads are unconditional (ring offsets are always in bounds, the values only
            // count where the masks say so).  Lane 15 does not probe: it hashes position P-2 alongside the others and
            // performs LZ4_putPosition(ip - 2) with the same instructions.  Anything unusual leaves the loop for the
            // generic path below, which redoes the batch from P.
            // ---------------------------------------------------------------------------------------
            bool finished = false;
            bool redo = false;               // the lean loop handed over to the dense batches: start the round again
            // ---------------------------------------------------------------------------------------
            // dense batches: streams of SHORT sequences (a few literals, a match of less than 16 bytes -- noisy bit
            // planes, quantised data) spend one lean iteration per 5..10 bytes.  Here all 64 lanes probe P .. P+63 against
            // the table as it stands, every lane judges its own candidate (16 bytes forward, 4 back), and a scalar walk over
            // the two ballots then resolves sequence after sequence without touching the LDS: first hit at or behind the
            // cursor -> literals, catch-up, match length from that lane's registers -> cursor behind the match.  Probes
            // that share a bucket with an EARLIER lane of the batch (their true candidate may be that lane) end the batch
            // in front of them; so do long matches, long literal runs and long catch-ups, which the lean / generic paths
            // take over.  The sequences found are written in one go (lane k writes sequence k), the probes the walk passed
            // over enter the table with one ds_max.  Not used inside block-linked frames.
            // ---------------------------------------------------------------------------------------
            if (DENSE && U == 0 && dense_next) {
                for (;;) {
                    if (!(P >= w.wlo + 4u && P + 112u <= w.hi_valid() && P + 112u <= matchlimit)) { dense_next = false; break; }
                    SQY_STAMP(20);
                    if (pend) { emit_pending(); if (failed) break; }
                    const uint32_t pos = P + (uint32_t)lane;
                    const uint32_t wlo4 = w.wlo + 4u;
                    const uint4 s16 = w.lds128(pos);
                    const uint32_t b4 = w.lds32(pos - 4u);
                    // LZ4_putPosition(P - 2) of the match in front: its five bytes sit in lane 0's b4 / s16
                    {
                        const uint32_t lo2 = (b4 >> 16) | (s16.x << 16);
                        const uint32_t h2 = lz4_hash5_32(lo2, s16.x >> 16);
                        if (lane == 0) atomicMax(&table[h2], ((P - 2u) << tsh) | tag_of(lo2));
                        wave_lds_sync();
                        put2 = 0xffffffffu;
                    }
                    const uint32_t h = lz4_hash5_32(s16.x, s16.y);
                    const uint32_t mytag = tag_of(s16.x);
                    const uint32_t mine = (pos << tsh) | mytag;
                    const uint32_t oe = table[h];
                    // lanes that are not the first of their bucket inside this batch (flagged ds_max through the table, then restored)
                    atomicMax(&table[h], 0x80000000u | (uint32_t)(63 - lane));
                    const uint32_t fl = table[h];
                    table[h] = oe;
                    wave_lds_sync();
                    const bool dup = (63u - (fl & 63u)) != (uint32_t)lane;
                    const uint32_t old = oe >> tsh;
                    const bool near = (pos - old) <= LZ4_MAXD && (oe & tmask) == mytag;
                    const bool cin = near && old >= wlo4;
                    uint4 c16 = w.lds128(old);
                    uint32_t cb4 = w.lds32(old - 4u);
                    if (ballot(near && !cin)) {
                        if (near && !cin) {                                             // candidates behind the ring: one round trip for all of them
                            c16 = glb_ld_u128(w.src + old);                             // old + 16 <= pos + 15 < matchlimit
                            // (the stream begins at p_lo: 0, or inside the history window of a block-linked frame's first blocks)
                            cb4 = old >= p_lo + 4u ? glb_ld_u32(w.src + old - 4u) : (glb_ld_u32(w.src + p_lo) << (8u * (4u - (old - p_lo))));
                        }
                    }
                    const uint32_t d = first_diff16(s16, c16);                          // 0..16 equal bytes forward
                    const uint32_t xb = b4 ^ cb4;
                    const uint32_t bkv = xb ? ((uint32_t)__builtin_clz(xb) >> 3) : 4u;  // equal bytes in front, 4 = maybe more
                    // what the walk needs of a lane in one word: forward bytes | bytes in front << 5 | literal limit << 8 | hit << 12 |
                    // shares-a-bucket << 13 | offset << 16.
                    // The literal limit folds every reason to leave the dense batches into one compare per sequence: fewer than 15
                    // literals; fewer than 5 when the catch-up may run past the 4 bytes looked at; none at all (0: the match goes to
                    // the lean / generic paths) when it runs past the 16 bytes looked at or its candidate sits within 15 bytes of the
                    // stream's start (the catch-up limit needs care there; a block-linked frame's history limits it likewise).
                    auto verdict_word = [&](uint32_t fwd, uint32_t bk, uint32_t cand, uint32_t off) -> uint32_t {
                        const bool edge = cand < p_lo + 16u || (LINKED && back_room(cand) < 4u);
                        const uint32_t ml1 = (fwd == 16u || edge) ? 0u : (bk == 4u ? 5u : 15u);
                        return fwd | (bk << 5) | (ml1 << 8) | (fwd >= 4u ? 1u << 12 : 0u) | (off << 16);
                    };
                    const uint32_t info = (verdict_word(d, bkv, old, pos - old) & (near ? ~0u : ~(1u << 12))) | (dup ? 1u << 13 : 0u);
                    const uint64_t M = ballot(near && d >= 4u);
                    const uint64_t D = ballot(dup);
                    SQY_STAMP(21);
                    // ---- the walk.  Sequence after sequence: first event lane at or behind the cursor -> its verdict word -> cursor behind the
                    // match.  Round 3 profiled it at three quarters of a batch's time and hand-wrote the common step -- an event lane that is
                    // a plain hit -- as a scalar loop of ~20 instructions; round 5 takes the step apart: WHAT the parse does when its cursor
                    // stands at lane i (which lane matches, with what verdict, where the cursor goes) depends, for a plain hit, on i alone, so
                    // every lane works that out for its own i at once (a 64-bit shift of the event mask, one lane permute, a few compares:
                    // tools/dense_walk_model.c), and what is left in a row is the chain 0 -> next(0) -> next(next(0)) ..: one lane read, a
                    // bit set and a compare per sequence.  The sequences are recorded where they begin: lane c = the sequence whose anchor is
                    // lane c (bit c of `chain`), no second numbering.  Lanes that share a bucket with an earlier lane of the batch (their true
                    // candidate depends on which of those lanes have entered the table by then: 14 % of the steps on quantised data) stop
                    // the chain; they are settled below from the chain so far and written into the cursor lane's record, and the chain goes on.
                    uint32_t cur = 0;                                                   // lane units; the cursor is also the anchor of the next sequence
                    uint64_t chain = 0;                                                 // bit c: a sequence starts (has its anchor) at lane c
                    bool keep_dense = true;
                    uint64_t evm = M | D;                                               // lanes the walk has to look at
                    // per cursor lane: status | probe lane << 8 | next cursor << 16, and the verdict word of that probe lane
                    constexpr uint32_t W_OK = 0u, W_NONE = 1u, W_LIMIT = 2u, W_MATE = 3u;
                    uint32_t epk, einf;
                    {
                        const uint64_t x = evm >> (uint32_t)lane;
                        const uint32_t eo = x ? (uint32_t)__builtin_ctzll(x) : 0u;
                        const uint32_t e = (uint32_t)lane + eo;                        // first event lane at or behind me (me, when there is none)
                        einf = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(e << 2), (int)info);
                        const uint32_t st = !x ? W_NONE : ((einf >> 13) & 1u) ? W_MATE : (eo >= ((einf >> 8) & 15u)) ? W_LIMIT : W_OK;
                        epk = st | (e << 8) | ((e + (einf & 31u)) << 16);
                    }
                    // is lane q in the table by now?  (q in front of the cursor: it belongs to the sequence of the last chain lane at or in
                    // front of it -- a literal or the probe that matched: yes; strictly inside the match: no, but for its ip - 2)
                    auto entered = [&](uint32_t q) -> bool {
                        if (q >= cur) return true;                                      // the literals of the sequence under way: probes the parse passed
                        const uint64_t cs = chain & ((2ull << q) - 1ull);              // (cur > 0: lane 0 is a chain lane)
                        const uint32_t c = 63u - (uint32_t)__builtin_clzll(cs);
                        const uint32_t wq = lane_read(epk, c);
                        const uint32_t fqc = (wq >> 8) & 63u, nxc = wq >> 16;
                        return q <= fqc || q + 2u == nxc;
                    };
                    for (;;) {
                        uint32_t st, t0;
                        SQY_STAMP(24);
                        asm volatile(
                            "1:\n\t"
                            "v_readlane_b32 %[t0], %[epk], %[cur]\n\t"
                            "s_and_b32 %[st], %[t0], 0xff\n\t"
                            "s_cmp_lg_u32 %[st], 0\n\t"
                            "s_cbranch_scc1 2f\n\t"
                            "s_bitset1_b64 %[ch], %[cur]\n\t"
                            "s_lshr_b32 %[cur], %[t0], 16\n\t"
                            "s_cmp_lt_u32 %[cur], 64\n\t"
                            "s_cbranch_scc1 1b\n"
                            "2:"
                            : [t0] "=&s"(t0), [st] "=&s"(st), [cur] "+s"(cur), [ch] "+s"(chain)
                            : [epk] "v"(epk)
                            : "scc");
                        SQY_STAMP(25);
                        if (st == W_OK || st == W_NONE) break;                          // cursor past the batch / no event lane left
                        if (st == W_LIMIT) { keep_dense = false; SQY_REASON(11); break; }
                        // ---- W_MATE: the first event lane at or behind the cursor shares its bucket with an earlier lane of this batch ----
                        SQY_REASON(13);
                        uint32_t fq = (t0 >> 8) & 63u;
                        uint32_t inf = 0;
                        bool found = false;
                        for (;;) {
                            inf = lane_read(info, fq);
                            if ((inf >> 13) & 1u) {
                                SQY_REASON(14);
                                // If one of the earlier lanes of the bucket has entered the table by now (a probe the parse passed over, or an
                                // ip - 2), the LATEST such lane is this probe's true candidate; else the table entry from before the batch is.
                                const uint32_t hf = lane_read(h, fq);
                                uint64_t mates = ballot(h == hf) & ((1ull << fq) - 1ull);
                                uint32_t qm = 64u;
                                while (mates) {
                                    const uint32_t c = 63u - (uint32_t)__builtin_clzll(mates);
                                    mates &= ~(1ull << c);
                                    if (entered(c)) { qm = c; break; }
                                }
                                if (qm < 64u) {
                                    // both sequences sit in registers: lane qm's 16 + 4 bytes go to every lane, each compares its own against
                                    // them (one vector pass), lane fq's result counts
                                    const uint4 cq = make_uint4(lane_read(s16.x, qm), lane_read(s16.y, qm), lane_read(s16.z, qm), lane_read(s16.w, qm));
                                    const uint32_t dq = lane_read(first_diff16(s16, cq), fq);
                                    const uint32_t xbq = lane_read(b4, fq) ^ lane_read(b4, qm);
                                    const uint32_t bq = xbq ? ((uint32_t)__builtin_clz(xbq) >> 3) : 4u;
                                    inf = verdict_word(dq, bq, P + qm, fq - qm);
                                    SQY_REASON(10);
                                }
                            }
                            if ((inf >> 12) & 1u) { found = true; break; }
                            // (a same-bucket lane that is no match: one more probe passed) -- the next event lane behind it, same cursor
                            const uint64_t rest = fq < 63u ? evm >> (fq + 1u) : 0ull;
                            if (!rest) break;
                            fq += 1u + (uint32_t)__builtin_ctzll(rest);
                        }
                        fq = sgpr(fq); inf = sgpr(inf);
                        if (!found) break;                                               // no event lane left behind the cursor
                        if (fq - cur >= ((inf >> 8) & 15u)) { keep_dense = false; SQY_REASON(11); break; }
                        // the cursor lane's record, settled: the chain reads it again and goes on
                        epk = lane_write(epk, sgpr(W_OK | (fq << 8) | ((fq + (inf & 31u)) << 16)), sgpr(cur));
                        einf = lane_write(einf, inf, sgpr(cur));
                        SQY_STAMP(26);
                    }
                    const uint32_t nseq = (uint32_t)__builtin_popcountll(chain);
                    SQY_REASON(8);
#ifdef SQY_LZ4_DIAG
                    dreason[9] += nseq;
#endif
                    SQY_STAMP(22);
                    if (nseq == 0) { dense_next = false; SQY_REASON(12); break; }        // (P - 2 is in the table: put2 stays empty)
                    // the probes the parse passed over enter the table: every lane in front of the cursor that is not strictly inside a match
                    // (but for a match's ip - 2) -- each lane asks the record of the chain lane it belongs to
                    {
                        const uint64_t cs = chain & ((2ull << (uint32_t)lane) - 1ull);
                        const uint32_t c = 63u - (uint32_t)__builtin_clzll(cs | 1ull);
                        const uint32_t wq = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)epk);
                        const uint32_t fqc = (wq >> 8) & 63u, nxc = wq >> 16;
                        const bool ins = (uint32_t)lane < cur && ((uint32_t)lane <= fqc || (uint32_t)lane + 2u == nxc);
                        if (ins) atomicMax(&table[h], mine);
                    }
                    wave_lds_sync();
                    // ---- write the sequences: lane c of `chain` = the sequence whose anchor is lane c ----
                    {
                        const bool on = (chain >> (uint32_t)lane) & 1ull;
                        // the catch-up takes what the literals and the bytes known equal in front allow
                        const uint32_t q_fq = (epk >> 8) & 63u;
                        const uint32_t q_df = einf & 31u, q_bk = (einf >> 5) & 7u, q_off = einf >> 16;
                        const uint32_t q_lit0 = q_fq - (uint32_t)lane;
                        const uint32_t q_back = q_bk < q_lit0 ? q_bk : q_lit0;
                        const uint32_t q_lit = q_lit0 - q_back, q_mc = q_df - 4u + q_back;
                        const uint32_t ext = q_mc >= 15u ? 1u : 0u;                     // match code <= 15: at most one extension byte (0)
                        const uint32_t sb = on ? 1u + q_lit + 2u + ext : 0u;
                        uint32_t inc = sb;                                              // inclusive prefix sum over the wave
                        inc += __builtin_amdgcn_update_dpp(0u, inc, 0x111, 0xf, 0xf, false);   // row_shr:1
                        inc += __builtin_amdgcn_update_dpp(0u, inc, 0x112, 0xf, 0xf, false);   // row_shr:2
                        inc += __builtin_amdgcn_update_dpp(0u, inc, 0x114, 0xf, 0xf, false);   // row_shr:4
                        inc += __builtin_amdgcn_update_dpp(0u, inc, 0x118, 0xf, 0xf, false);   // row_shr:8
                        inc += __builtin_amdgcn_update_dpp(0u, inc, 0x142, 0xa, 0xf, false);   // row_bcast:15
                        inc += __builtin_amdgcn_update_dpp(0u, inc, 0x143, 0xc, 0xf, false);   // row_bcast:31
                        const uint32_t total = lane_read(inc, 63);
                        const uint32_t my_op = op + inc - sb;
                        // upstream's two limit checks per sequence (no literal-length extension below 15 literals)
                        const bool bad = on && (my_op + 1u + q_lit + (2 + 1 + LZ4_LASTLITERALS) > olimit ||
                                                my_op + 1u + q_lit + 2u + (1 + LZ4_LASTLITERALS) + ext > olimit);
                        if (ballot(bad)) { failed = true; break; }
                        o.reserve(op, total);
                        const uint4 l16 = s16;                                          // the literals start at the sequence's anchor: my own lane
                        // the sequence as five little-endian words: token, literals, offset, (extension byte 0), exact length sb;
                        // whole words leave as (unaligned) word stores, the 0..3 bytes behind them as byte stores
                        const uint32_t tok = (q_lit << 4) | q_mc;                       // (q_mc <= 15 is its own token nibble)
                        uint32_t wv[6];
                        wv[0] = tok | (l16.x << 8);
                        wv[1] = __builtin_amdgcn_alignbyte(l16.y, l16.x, 3);
                        wv[2] = __builtin_amdgcn_alignbyte(l16.z, l16.y, 3);
                        wv[3] = __builtin_amdgcn_alignbyte(l16.w, l16.z, 3);
                        wv[4] = l16.w >> 24;
                        wv[5] = 0;
                        const uint32_t hb = 1u + q_lit;                                 // the offset sits behind token + literals
                        const uint32_t hw = hb >> 2, hs = (hb & 3u) * 8u;
                        const uint64_t hdr = (uint64_t)q_off << hs;                     // (extension byte, when present, is 0)
                        const uint32_t keep = (1u << hs) - 1u;
#pragma unroll
                        for (uint32_t i = 0; i < 6u; ++i)
                            wv[i] = i < hw ? wv[i] : (i == hw ? ((wv[i] & keep) | (uint32_t)hdr) : (i == hw + 1u ? (uint32_t)(hdr >> 32) : 0u));
                        lds_u8* const my = o.at(my_op);
#pragma unroll
                        for (uint32_t i = 0; i < 4u; ++i)
                            if (4u * (i + 1u) <= sb) reinterpret_cast<SQY_LDS pk_u32*>(my + 4u * i)->v = wv[i];
                        {
                            const uint32_t tw = sb >> 2, tb = sb & 3u;                  // first word that is not stored whole, its bytes
                            const uint32_t tv = tw == 0 ? wv[0] : tw == 1 ? wv[1] : tw == 2 ? wv[2] : tw == 3 ? wv[3] : wv[4];
                            if (tb >= 1u) my[4u * tw] = (uint8_t)tv;
                            if (tb >= 2u) my[4u * tw + 1u] = (uint8_t)(tv >> 8);
                            if (tb >= 3u) my[4u * tw + 2u] = (uint8_t)(tv >> 16);
                        }
                        op += total;
                    }
                    anchor = P + cur;
                    put2 = anchor - 2u;                                                 // (already in the table when it lay inside the batch: harmless)
                    P = anchor;
                    SQY_STAMP(23);
                    w.ensure(P);
                    if (!keep_dense) { dense_next = false; break; }
                }
                if (failed) break;
            }
            if (!ACCEL && U == 0) {
                // Enter the loop with no load of the compiler's own in flight (results the generic path left unused
                // count): otherwise its waitcnt pass puts a vmcnt(0) in front of the loop's first ring read, and that
                // one would wait for the ring block in flight on EVERY iteration.
                __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
                // per lane: offset of its position from P (probes 0..14; lane 15 performs the pending LZ4_putPosition(P - 2) with
                // the same instructions), and a word that is non-zero for lanes that do not probe
                const bool putlane = lane == 15;
                const uint32_t posoff = putlane ? 0xfffffffeu : (uint32_t)lane;
                const uint32_t notprobe = (uint32_t)lane < 15u ? 0u : 1u;
                const uint32_t lane0_off = lane == 0 ? 0u : 0xffffffffu;
                for (;;) {
                    {
                        // (all scalar: a three-way vector minimum here costs the loop a round through VCC per iteration)
                        const uint32_t hv = sgpr(w.hi_valid());
                        const uint32_t top = hv < matchlimit ? hv : matchlimit;
                        if (!(P >= sgpr(w.wlo) + 4u && P + 1200u <= top)) break;
                    }
                    SQY_STAMP(1);
                    const uint32_t pos = P + posoff;
                    const uint32_t wlo8 = w.wlo + 8u;
                    // The twelve bytes [pos - 4, pos + 8) of every probe (the hash takes five bytes, the tag four, the four in front
                    // serve the catch-up and the literals) out of the four dwords that hold them: reads at a multiple of 4 are not
                    // replayed, the byte shift is three v_alignbyte.  (Round 3: a 4-byte and an 8-byte read at byte granularity,
                    // two replays = +128 cycles per sequence.)
                    const uint32_t sh1 = pos & 3u;
                    const uint4 d1 = lds_ld_4dw(w.win + ((pos - 4u - sh1) & (LZ4_WIN - 1u)));
                    const uint32_t b4 = __builtin_amdgcn_alignbyte(d1.y, d1.x, sh1);
                    const uint2 s16 = make_uint2(__builtin_amdgcn_alignbyte(d1.z, d1.y, sh1), __builtin_amdgcn_alignbyte(d1.w, d1.z, sh1));
                    // (in the shadow of those reads: limit checks / stage room of the sequence found in the last iteration)
                    if (pend) { emit_pending_checks(); if (failed) break; }
                    SQY_STAMP(2);
                    const uint32_t h = lz4_hash5_32(s16.x, s16.y);
                    const uint32_t mytag = tag_of(s16.x);
                    const uint32_t mine = (pos << tsh) | mytag;
                    if (putlane) table[h] = mine;                                       // LZ4_putPosition(P - 2), before anybody looks
                    wave_lds_sync();
                    put2 = 0xffffffffu;
                    SQY_STAMP(3);
                    const uint32_t oe = table[h];
                    // (in the shadow of the table read: its bytes)
                    if (pend) emit_pending_bytes();
                    const uint32_t old = oe >> tsh;
                    // tag differs: the candidate's first four bytes differ, it cannot match.  Tag equal: they are equal but for
                    // one case in 2^tsh -- the FIRST such probe is taken as the winner right away and one wide round (64 lanes x
                    // 16 bytes from ip / match, ring or global memory) then settles in a single round trip whether the match is
                    // real and how far it goes (round 2 read every probe's candidate first and only then, for matches of 16
                    // bytes and more, started the wide round: one dependent round trip more per long match, two when the
                    // candidate sits behind the ring).
                    // One word that is zero exactly for a probing lane whose candidate is within reach and carries the tag
                    // (a single compare feeds the ballot; a boolean expression costs a select and a second compare)
                    const uint32_t rej = ((oe & tmask) ^ mytag) | ((pos - old) >> 16) | notprobe;
                    SQY_STAMP(4);
                    const uint64_t nm = ballot(rej == 0u);
                    if (nm == 0) { SQY_REASON(0); break; }
                    const uint32_t f0 = ctz64(nm);                                      // <= 14
                    const uint32_t mt0 = lane_read(old, f0);
                    const uint32_t ip0 = P + f0;
                    // (the wide compare below starts up to 7 bytes in front of the match: a candidate at the stream's first bytes
                    // -- the empty buckets of a fresh table say "first byte" -- goes to the generic path)
                    if (mt0 < p_lo + 8u) { SQY_REASON(3); break; }
                    SQY_STAMP(5);
                    // Commit probes 0..f0 now and read the buckets back: a probe that does not find its own entry shares its
                    // bucket with another probe of the batch -- the later one's true candidate would be the earlier one --,
                    // which is left to the generic path below after the buckets are put back (both hold the same old entry).
                    // (Round 2 compared the hashes lane by lane with readlane before committing: ~16 instructions per probe.)
                    const bool commit = (uint32_t)lane <= f0;
                    if (commit) table[h] = mine;
                    wave_lds_sync();
                    const uint32_t rb = table[h];
                    // One frame for both sides: it starts 4 + r bytes in front of (ip0, mt0), r = ip0 & 3, so that the ip side is read
                    // at multiples of 4 (never replayed) and the 4 bytes in front of the match -- the catch-up -- are part of lane 0's
                    // 16 bytes instead of a read of their own; the match side is ONE 16-byte read per lane (replayed once unless the
                    // offset happens to be a multiple of 16).  Round 3 read 4 + 2 x 8 + 2 x 8 bytes at byte granularity: five replays,
                    // +320 cycles per sequence.
                    const uint32_t r3 = ip0 & 3u;
                    const uint32_t hd = 4u + r3;                                        // bytes of lane 0 in front of ip0
                    const uint32_t dd = (uint32_t)lane * 16u - hd;
                    uint32_t x0, x1, x2, x3;                                            // ip side ^ match side
                    // (two complete branches: a merged load would make the common, ring-only side wait on vmcnt too)
                    if (mt0 >= wlo8) {
                        uint32_t oi = (ip0 + dd) & (LZ4_WIN - 1u), om = (mt0 + dd) & (LZ4_WIN - 1u);
                        asm volatile("" : "+v"(oi), "+v"(om));                          // (both reads issued back to back, see the loop top)
                        const uint4 ci = lds_ld_4dw(w.win + oi), cm = lds_ld_u128(w.win + om);   // ip side resident and clear of matchlimit (loop condition)
                        x0 = ci.x ^ cm.x; x1 = ci.y ^ cm.y; x2 = ci.z ^ cm.z; x3 = ci.w ^ cm.w;
                    } else {
                        SQY_EXP(const unsigned long long x_t0 = __builtin_amdgcn_s_memtime();)
                        const uint4 cg = glb_ld_u128(w.src + (uint32_t)(mt0 + dd));     // (32-bit sum: dd wraps for lane 0)  mt0 - 7 >= p_lo; mt0 + 1024 <= ip0 + 1023 < matchlimit
                        SQY_EXP(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); x_far += __builtin_amdgcn_s_memtime() - x_t0; x_nfar += 1;)
                        const uint4 ci = lds_ld_4dw(w.win + ((ip0 + dd) & (LZ4_WIN - 1u)));
                        x0 = ci.x ^ cg.x; x1 = ci.y ^ cg.y; x2 = ci.z ^ cg.z; x3 = ci.w ^ cg.w;
                        asm volatile("" : "+v"(x0));                                    // (keeps the optimiser from re-merging the branches)
                        SQY_REASON(2);
                    }
                    // the 4 bytes in front of (ip0, mt0): bytes r .. r+3 of lane 0
                    const uint32_t xbv = __builtin_amdgcn_alignbyte(x1, x0, r3);
                    // forward: lane 0 leaves out its first 4 + r bytes
                    x0 &= lane0_off;
                    x1 &= lane0_off | (0xffffffffu << (8u * r3));
                    const uint32_t g = first_nonzero16(x0, x1, x2, x3);
                    SQY_STAMP(6);
                    const uint64_t nf = ballot(g != 16u);
                    uint32_t eq = 1024u - hd;                                           // equal bytes from (ip0, mt0) on
                    bool settled = false;
                    if (nf) {
                        const uint32_t l = ctz64(nf);
                        eq = l * 16u + lane_read(g, l) - hd;
                        settled = true;
                    }
                    SQY_STAMP(7);
                    if ((ballot(rb != mine) & ((2ull << f0) - 1ull)) != 0ull || eq < 4u) {
                        // same-bucket probes, or the tag lied (eq < 4): the table as it was, the generic path redoes the batch
                        if (eq < 4u) SQY_REASON(1); else SQY_REASON(3);
                        if (commit) table[h] = oe;
                        wave_lds_sync();
                        break;
                    }
                    const uint32_t ml = eq - 4u;                                        // exact when settled, else "at least"
                    // catch-up of the winner: ip - anchor = f0 literals, match > 0
                    const uint32_t xb = lane_read(xbv, 0);
                    const uint32_t bk = xb ? ((uint32_t)__builtin_clz(xb) >> 3) : 4u;   // equal bytes in front, 4 = maybe more
                    const uint32_t room0 = back_room(mt0);
                    const uint32_t lim = f0 < room0 ? f0 : room0;
                    const uint32_t bck = bk < lim ? bk : lim;
                    const bool slow_back = bk == 4u && lim > 4u;
                    SQY_STAMP(8);
                    if (!settled || slow_back) {
                        // winner known, but the match runs past the wide round or the catch-up past 4 bytes: generic tail
                        if (!settled) SQY_REASON(4); else SQY_REASON(5);
                        f = f0; fcand = mt0; ipf = ip0; fwl = ml; fw_exact = settled;
                        bkl = slow_back ? 0xffffffffu : bck;
                        batch_done = true;
                        break;
                    }
                    SQY_EXP(x_nseq += 1;)
                    pend = true; pe_lit = f0 - bck; pe_mcode = ml + bck; pe_off = ip0 - mt0;    // < 15 literals
                    pe_litv = b4 >> 24;                                                 // literal k-1 sits at anchor + k - 1 = pos - 1
                    const uint32_t ipn = ip0 + 4u + ml;
                    anchor = ipn;
                    if (ipn >= mflimitPlusOne) { finished = true; break; }
                    put2 = ipn - 2u;
                    P = ipn;
                    SQY_STAMP(9);
                    w.ensure(P);
                    if (DENSE) {
                        // two matches of less than 16 bytes in a row: back to the dense batches
                        shorts = ml < 12u ? shorts + 1u : 0u;
                        if (shorts >= 2u) { shorts = 0; dense_next = true; redo = true; break; }
                    } else if (!LINKED) {
                        // 32 in a row: a stream of short sequences, this chunk goes to the DENSE kernel
                        // (round 5: a leaky count -- a short match adds one, any other takes four away -- up to 128.  Rounds 2-4 gave a chunk up
                        // after 32 short matches IN A ROW: streams that are four fifths short matches or more (the top plane of quantised
                        // data) get there within a few hundred sequences either way, but 32 in a row also happen in chunks that are mostly
                        // ordinary -- 36 chunks of a 2048 x 2048 x 256 slab of the bench stack, which then waited for a second launch that
                        // was no faster on them: that slab 2.46 -> 1.84 ms, the C3 slab 6.29 -> 5.65, the "+ 40000" stack 3.80 -> 2.30.)
                        shorts = ml < 12u ? shorts + 1u : (shorts > 4u ? shorts - 4u : 0u);
                        if (shorts >= 128u && redo_list) { redo_dense = true; break; }
                    } else if (LINKED) {
                        // (round 5) the block-parallel parse's guesses (mode 1): a stream of short sequences -- the top plane of a quantised
                        // stack, ~35 000 sequences per block -- is given up here, warm-up or not: its guess fails anyway (on such data nearly
                        // every look-up finds a candidate, any difference between the guessed and the true table changes a match and with it
                        // the positions that enter the table: tools/segment_convergence.c), and the run it belongs to is parsed again in order
                        // by the kernel with the dense batches (mode 2).  The block's tables on record are made impossible below, so that it
                        // and the block behind it fail the check.  (A leaky count of 2048: the sparse planes of a diff3x3x1 residual have
                        // stretches of short matches too, and are better off here.)
                        if constexpr (LINKED) if (spec_mode == 1u) {
                            shorts = ml < 12u ? shorts + 1u : (shorts > 4u ? shorts - 4u : 0u);
                            if (shorts >= 2048u && !(blocks[b_last - 1u].flags & 1u)) { redo_dense = true; break; }
                        }
                    }
                }
                SQY_EXP(x_lap(1);)                              // 1: lean loop (and dense batches)
                if (finished || failed || redo_dense) break;
                if (redo) continue;
            }

            // ---------------------------------------------------------------------------------------
            // generic path: any step schedule, chunk borders, hazards -- exact but slower
            // ---------------------------------------------------------------------------------------
            if (pend) { emit_pending(); if (failed) break; }
            // (Round 3, measured without gain and not kept:
            // the sequences of the next eight batches of incompressible data kept in flight (DMA into the idle ring's slots, a batch reads its slot)
            // -- exact, and no faster: a batch of these chunks is ~1 us of instructions and LDS round trips in this path, the
            // HBM round trip of its strided reads is not what it waits for.
            // And the lean loop's scalar bookkeeping (loop bounds cached, ONE arithmetic test for everything unusual about the pending
            // sequence, exit reasons as one word instead of flags): ~20 scalar instructions fewer per sequence, the sparse planes 1.5 %
            // faster, the dense kernel 2.7 % slower (A/B on one box) -- an extra scalar instruction costs ~1.6 cycles of an
            // iteration's ~730 (20 s_nop anywhere in the loop: +4.4 %), the loop is bound by its dependent LDS / VALU chain.)
            // Batches that find nothing (data that does not compress; the step schedule is at 16 bytes and more): proved empty from the
            // tags, one LDS round trip per batch.  Every probe reads its bucket and enters itself with ds_max (returning what was there):
            // nothing can match unless some probe's entry -- the one from before the batch, or an earlier probe of the batch in the
            // same bucket -- carries the probe's tag within reach.  A probe that finds a LATER probe of the batch in its bucket (the
            // atomics of one instruction did not run in lane order) proves nothing; then, and on any tag hit, the buckets are put
            // back and the batch goes through the generic code below.
            // (first pass and linked frames only: the dense kernel sees chunks of short sequences, and is 3 % slower with this loop compiled in)
            // (round 6, built, exact, measured, not kept: the same proof from the second batch of a search on -- steps below 16, the
            // sequences out of the ring -- and the loads of the batches at step 16 and more issued four batches ahead (where the search probes
            // is a closed form while nothing is found).  A chunk of noise 75 -> 52 -> 35 us alone, the noise planes' share of the slot time
            // -40 %; with calls in flight the bench stayed where it was (without the loads ahead) or lost 3 % (with them: they take HBM
            // bandwidth from the other calls' transposes, which set the pace), and the sparse planes of a diff3x3x1 residual lost up to 12 %
            // when the proof was tried right behind the lean loop: profiles/r06_experiments.txt.)
            while (!DENSE && !ACCEL && !batch_done && U >= LZ4_RINGLESS_U && put2 == 0xffffffffu) {
                // where this batch's probes are; false = the batch reaches the end of the search (the generic path takes it)
                auto geometry = [&](uint32_t& pos, uint32_t& nxt) -> bool {
                    const uint32_t s_first = (62 + U) >> 6;
                    const uint32_t ustar = 64 * (s_first + 1) - 62;
                    const uint32_t u = U + lane;
                    pos = P + s_first * lane + (u > ustar ? u - ustar : 0u);
                    nxt = pos + ((u >= ustar) ? s_first + 1 : s_first);
                    return ballot(nxt <= mflimitPlusOne) == ~0ull;
                };
                // the batch's probes against the table; false = one of them may match (the table is as before: the generic path decides)
                auto enter = [&](uint32_t h, uint32_t mytag, uint32_t pos, uint32_t nxt) -> bool {
                    const uint32_t mine = (pos << tsh) | mytag;
                    const uint32_t oe = table[h];
                    const uint32_t seen = atomicMax(&table[h], mine);
                    wave_lds_sync();
                    const bool hit_before = (oe & tmask) == mytag && (pos - (oe >> tsh)) <= LZ4_MAXD;
                    const uint32_t sp = seen >> tsh;
                    const bool in_batch = sp >= P && seen != oe;
                    const bool trouble = hit_before || (in_batch && (sp >= pos || (seen & tmask) == mytag));
                    if (ballot(trouble)) {
                        table[h] = oe;                                      // (same value from every probe of a bucket)
                        wave_lds_sync();
                        return false;
                    }
                    P = lane_read(nxt, 63);
                    U += 64;
                    return true;
                };
                // Nothing has matched since the chunk began (anchor == p0: the search started at probe 1 and these are the positions the
                // transpose foresaw): bucket and tag of a batch's probes come from the digest, 256 bytes instead of 64 scattered reads
                // of the plane stream.  An entry of all ones: the probe's five bytes end in another wave's piece -- that lane looks itself.
                // The entries are fetched DG_AHEAD batches ahead: a pass of DG_AHEAD batches, batch k out of slot k of the queue, the slot
                // filled again at once for the pass behind (fixed registers: a queue that shifts would have to wait for what it moves).
                auto digest_ok = [&]() -> bool { return sgpr((uint32_t)(dgp && anchor == p0 && U >= 961u && U - 961u + 64u <= dg_words)) != 0u; };   // (uniform: it decides about P and U)
                if (digest_ok()) {
                    SQY_GLB const uint32_t* const dg = (SQY_GLB const uint32_t*)dgp;
                    // (loads behind the row's end are clamped to its last word: never used -- digest_ok() guards every batch)
                    auto dg_fetch = [&](uint32_t ahead) -> uint32_t {
                        const uint32_t ix = U - 961u + 64u * ahead + (uint32_t)lane;
                        return dg[ix < dg_words ? ix : dg_words - 1u];
                    };
                    if (dgq_U != U) {                                       // the first pass, or back from the generic path: fill the queue
#pragma unroll
                        for (int k = 0; k < DG_AHEAD; ++k) dgq[k] = dg_fetch((uint32_t)k);
                    }
                    dgq_U = 0xffffffffu;
                    bool on = true;
#pragma unroll
                    for (int k = 0; k < DG_AHEAD; ++k) {
                        if (!on) continue;
                        uint32_t pos, nxt;
                        if (!digest_ok() || !geometry(pos, nxt)) { on = false; continue; }
                        const uint32_t e = dgq[k];
                        uint32_t h = e >> 16, mytag = e & 0xffffu;
                        if (ballot(e == 0xffffffffu)) {
                            if (e == 0xffffffffu) {
                                const uint64_t seq = glb_ld_u64(w.src + pos);
                                h = lz4_hash5(seq); mytag = tag_of((uint32_t)seq);
                            }
                        }
                        if (!enter(h, mytag, pos, nxt)) { on = false; continue; }
                        // the batch DG_AHEAD behind the one just taken, into its slot (U has moved on; fetched only now, when nothing
                        // reads the slot's old value any more: fetched in front, it went to another register and was MOVED here at the
                        // end of the step -- behind a wait for the load)
                        dgq[k] = dg_fetch((uint32_t)DG_AHEAD - 1u);
                    }
                    if (!on) break;                                         // (whatever ended the pass: the generic path looks at the batch)
                    dgq_U = U;                                              // a whole pass: slot k holds the batch k behind U again
                    continue;
                }
                uint32_t pos, nxt;
                if (!geometry(pos, nxt)) break;
                const uint64_t seq = glb_ld_u64(w.src + pos);              // (the ring was left behind at these strides)
                if (!enter(lz4_hash5(seq), tag_of((uint32_t)seq), pos, nxt)) break;
            }
            SQY_EXP(x_lap(2);)                                  // 2: batches proved empty
            SQY_EXP(x_ngen += 1;)
            if (!batch_done) {
                if (U != 0) SQY_REASON(6); else SQY_REASON(7);
                uint32_t pos, nxt;
                if (!ACCEL) {
                    const uint32_t s_first = (62 + U) >> 6 ? (62 + U) >> 6 : 1;
                    const uint32_t ustar = 64 * (s_first + 1) - 62;      // first unified index with step s_first+1
                    const uint32_t u = U + lane;
                    const uint32_t bump = u > ustar ? u - ustar : 0;      // #earlier lanes already at the larger step
                    pos = P + s_first * lane + bump;
                    const uint32_t adv = (u >= ustar) ? s_first + 1 : s_first;
                    nxt = pos + adv;
                } else {
                    // probe u (0 = the position right behind a match, 1 = the next byte / a block's first probe) sits S(u) bytes behind
                    // probe 0: increments 1, 1, then (64 a + j) >> 6 for j = 0, 1, ..  (64-bit: a may be as large as 65537)
                    auto S = [&](uint64_t u) -> uint64_t {
                        if (u <= 2) return u;
                        const uint64_t m = u - 2, q = m >> 6, r = m & 63;
                        return 2 + 64 * ((uint64_t)accel * q + q * (q - 1) / 2) + r * ((uint64_t)accel + q);
                    };
                    const uint64_t s0 = S(U), s1 = S((uint64_t)U + lane) - s0, s2 = S((uint64_t)U + lane + 1) - s0;
                    const uint64_t lim = (uint64_t)mflimitPlusOne + 1;    // (anything beyond is "not valid"; keeps the sums inside 32 bits)
                    pos = (uint32_t)(P + s1 < lim ? P + s1 : lim);
                    nxt = (uint32_t)(P + s2 < lim ? P + s2 : lim);
                }
                const bool valid = nxt <= mflimitPlusOne;
                const uint64_t vmask = ballot(valid);                     // a prefix of the lanes
                nvalid = (uint32_t)__builtin_popcountll(vmask);

                uint64_t seq = 0;
                if (valid) seq = w.rd64(pos);
                if (put2 != 0xffffffffu) {
                    // LZ4_putPosition(ip - 2) of the previous match, fetched together with this batch's sequences
                    const uint64_t s2 = w.rd64(put2);
                    const uint32_t h2 = lz4_hash5(s2);
                    if (lane == 0) table[h2] = (put2 << tsh) | tag_of((uint32_t)s2);
                    wave_lds_sync();
                    put2 = 0xffffffffu;
                }
                uint32_t h = 0, oe = 0, fl = 0;
                if (valid) {
                    h = lz4_hash5(seq);
                    oe = table[h];
                    atomicMax(&table[h], 0x80000000u | (uint32_t)(63 - lane));
                    fl = table[h];
                }
                const uint32_t seq32 = (uint32_t)seq;
                const uint32_t mytag = tag_of(seq32);
                const uint32_t old = oe >> tsh;
                const bool hazard = valid && ((63u - (fl & 63u)) != (uint32_t)lane);
                const uint64_t hz = ballot(hazard);

                // non-hazard lanes: candidate is the pre-batch table entry; a differing tag rules it out without a read
                const bool near = valid && !hazard && (old + LZ4_MAXD >= pos) && (oe & tmask) == mytag;
                uint32_t m32 = ~seq32;
                if (near) m32 = w.rd32(old);
                const uint64_t mm = ballot(near && m32 == seq32);

                f = mm ? ctz64(mm) : 64u;
                if (f < 64) fcand = lane_read(old, f);

                // hazard lanes in front of f, in lane order: candidate = latest earlier probe with the same hash
                uint64_t hzq = hz & ((f < 64) ? ((1ull << f) - 1ull) : ~0ull);
                while (hzq) {
                    const uint32_t c = ctz64(hzq);
                    hzq &= hzq - 1;
                    const uint32_t hc = lane_read(h, c);
                    const uint64_t grp = ballot(valid && h == hc) & ((1ull << c) - 1ull);
                    const uint32_t j = 63u - (uint32_t)__builtin_clzll(grp);
                    const uint32_t pj = lane_read(pos, j), pc = lane_read(pos, c);
                    if (pj + LZ4_MAXD >= pc && lane_read(seq32, j) == lane_read(seq32, c)) {
                        f = c;
                        fcand = pj;
                        break;
                    }
                }

                // table: restore, then insert the committed prefix (lanes <= f, or every valid lane)
                const uint32_t ncommit = (f < 64) ? f + 1 : nvalid;
                if (valid) table[h] = oe;
                if ((uint32_t)lane < ncommit) atomicMax(&table[h], (pos << tsh) | mytag);
                wave_lds_sync();
                if (f < 64) ipf = lane_read(pos, f);
                next_P = lane_read(nxt, 63);
                
            } else {
                next_P = P + 64u;
            }

            if (f >= 64) {
                if (nvalid < 64) break;                               // forwardIp > mflimitPlusOne -> last literals
                P = sgpr(next_P);
                U += 64;
                SQY_EXP(x_lap(3);)
                continue;
            }

            SQY_EXP(x_lap(3);)                                  // 3: generic batch search
            // ---- a match: ip = pos[f], match = fcand ----
            const uint32_t ip0 = sgpr(ipf);
            const uint32_t mt0 = sgpr(fcand);
            fwl = sgpr(fwl);
            bkl = sgpr(bkl);
            const uint32_t q = ip0 + 4, r = mt0 + 4;

            // forward extension: bytes equal beyond MINMATCH (upstream's LZ4_count from ip+4), up to matchlimit
            uint32_t ml;
            if (fw_exact) {
                ml = fwl;                                                 // settled by the speculative compares
            } else {
                ml = fwl;                                                 // bytes already known equal; continue from there
                // rounds of 64 lanes x 16 bytes per KiB.  Out of the ring as far as it is resident (at least AHEAD bytes beyond
                // ip: matches of up to ~2 KiB never touch global memory); beyond that both sides stream from global memory,
                // 4 KiB per round and the next round's loads in flight while this one is looked at (a zero run of 256 KiB is
                // 64 rounds: 64 dependent HBM round trips one after the other, or one and a stream behind it); the last
                // bytes in front of matchlimit take the exact byte-wise path.
                auto settle = [&](const uint32_t (&got)[4], uint32_t T) -> bool {   // adds the round's equal bytes to ml; true = mismatch found
#pragma unroll
                    for (uint32_t t = 0; t < 4; ++t) {
                        if (t < T) {
                            const uint64_t nf = ballot(got[t] != 16u);
                            if (nf) {
                                const uint32_t l = ctz64(nf);
                                ml += l * 16u + lane_read(got[t], l);
                                return true;
                            }
                            ml += 1024u;
                        }
                    }
                    return false;
                };
                const uint32_t src_end = matchlimit < (w.n & ~15u) ? matchlimit : (w.n & ~15u);   // 16-byte loads stay below this
                for (;;) {
                    uint32_t got[4] = {0, 0, 0, 0};
                    const uint32_t qa = q + ml, ra = r + ml;
                    const uint32_t lds_end = w.hi_valid() < matchlimit ? w.hi_valid() : matchlimit;
                    if (ra >= w.wlo && qa + 1024u + 16u <= lds_end) {
                        // whole KiBs resident on both sides (ra < qa)
                        uint32_t T = (lds_end - 16u - qa) >> 10;
                        T = T < 4u ? T : 4u;
                        got[0] = first_diff16(w.lds128(qa + (uint32_t)lane * 16u), w.lds128(ra + (uint32_t)lane * 16u));
                        if (T > 1u && !ballot(got[0] != 16u)) {              // most matches end inside the first KiB: look at it before the others
#pragma unroll
                            for (uint32_t t = 1; t < 4; ++t)
                                if (t < T) got[t] = first_diff16(w.lds128(qa + t * 1024u + (uint32_t)lane * 16u), w.lds128(ra + t * 1024u + (uint32_t)lane * 16u));
                        } else
                            T = 1;
                        if (settle(got, T)) break;
                        continue;
                    }
                    if (qa + 4096u + 16u <= src_end) {
                        // streaming: round k+1 is loaded while round k is compared.  Two register sets taking turns (the loop body
                        // twice), both loaded unconditionally: handed over at the loop's end, or loaded under a condition ("maybe
                        // defined" over the whole parse loop around this), the sets cost 64 registers more, and this cold loop
                        // then sets the whole kernel's register count (172 -> 108) -- which decides how many waves of OTHER kernels
                        // (the bit-plane transpose of the next call in flight) fit next to a resident chunk wave
                        uint4 ca[4], cb[4], na[4], nb[4];
                        auto load_round = [&](uint4 (&xa)[4], uint4 (&xb)[4], uint32_t ipos) {
#pragma unroll
                            for (uint32_t t = 0; t < 4; ++t) {
                                xa[t] = glb_ld_u128(w.src + ipos + t * 1024u + (uint32_t)lane * 16u);
                                xb[t] = glb_ld_u128(w.src + (ipos - qa + ra) + t * 1024u + (uint32_t)lane * 16u);
                            }
                        };
                        auto judge_round = [&](const uint4 (&xa)[4], const uint4 (&xb)[4]) -> bool {
#pragma unroll
                            for (uint32_t t = 0; t < 4; ++t) got[t] = first_diff16(xa[t], xb[t]);
                            return settle(got, 4u);
                        };
                        load_round(ca, cb, qa);
                        bool stop = false;
                        uint32_t at = qa;                                        // ip-side position of the round about to be judged
                        for (;;) {
                            bool more = at + 8192u + 16u <= src_end;             // another whole round behind this one
                            load_round(na, nb, more ? at + 4096u : at);          // (no other: the same round once more)
                            if (judge_round(ca, cb)) { stop = true; break; }
                            if (!more) break;
                            at += 4096u;
                            more = at + 8192u + 16u <= src_end;
                            load_round(ca, cb, more ? at + 4096u : at);
                            if (judge_round(na, nb)) { stop = true; break; }
                            if (!more) break;
                            at += 4096u;
                        }
                        if (stop) break;
                        continue;
                    }
                    // the tail: lanes stop at matchlimit, positions without 16 readable bytes compare byte by byte
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint32_t a = qa + (uint32_t)t * 1024u + (uint32_t)lane * 16u;
                        const uint32_t room = a < matchlimit ? matchlimit - a : 0u;
                        const uint32_t maxlen = room < 16u ? room : 16u;
                        got[t] = (maxlen == 0) ? 0u : common16(w, a, a - q + r, maxlen);
                        if (t == 0) {
                            if (ballot(got[0] != 16u)) { got[1] = got[2] = got[3] = 0; break; }
                        }
                    }
                    if (settle(got, 4u)) break;
                }
                ml = sgpr(ml);
            }

            // backward catch-up: while (ip > anchor && match > 0 && ip[-1] == match[-1])
            uint32_t back;
            {
                const uint32_t lim_a = ip0 - anchor, lim_m = back_room(mt0);
                const uint32_t lim = lim_a < lim_m ? lim_a : lim_m;
                if (bkl < 4u || lim <= bkl) {
                    back = bkl < lim ? bkl : lim;                          // settled by the speculative 4-byte compare
                    if (bkl == 0xffffffffu) back = 0xffffffffu;
                } else {
                    back = 0xffffffffu;
                }
                if (back == 0xffffffffu) {
                    // 64 bytes per round: lane l tests byte `done + l + 1` in front of (ip0, mt0); the first mismatch or the
                    // limit (literals available / liblz4's lowLimit on the match side) ends it
                    uint32_t done = 0;
                    for (;;) {
                        const uint32_t k = done + (uint32_t)lane + 1u;
                        bool ok = k <= lim;
                        const uint32_t pi = ip0 - k;
                        // LINKED: the match side may reach below position 0 (history more than 64 KiB in front of the block,
                        // liblz4's prefix mode inside one LZ4F_compressUpdate) -- signed, read from global memory there
                        const int32_t pm = (int32_t)mt0 - (int32_t)k;
                        if (mt0 >= done + w.wlo + 64u && ip0 - done <= w.hi_valid()) {       // uniform: both sides resident
                            if (ok) ok = w.lds8(pi) == w.lds8((uint32_t)pm);
                        } else if (ok) {
                            const uint32_t bm = (pm >= (int32_t)w.wlo && (uint32_t)pm < w.hi_valid()) ? w.lds8((uint32_t)pm)
                                                                                                      : glb_ld_u8(w.src + pm);
                            ok = w.rd8(pi) == bm;
                        }
                        const uint64_t bad = ~ballot(ok);
                        const uint32_t nb = bad ? ctz64(bad) : 64u;
                        done += nb;
                        if (nb < 64) break;
                    }
                    back = done;
                }
            }
            

            const uint32_t ip = ip0 - back;                           // start of the match after catch-up
            const uint32_t lit = ip - anchor;
            const uint32_t matchCode = ml + back;                     // match length - MINMATCH as upstream counts it
            const uint32_t offset = ip0 - mt0;

            SQY_EXP(x_lap(4);)                                  // 4: generic match extension + catch-up
            // upstream's two output-limit checks (token + literals, then offset + match length)
            if (op + 1 + lit + (2 + 1 + LZ4_LASTLITERALS) + lit / 255 > olimit) { failed = true; break; }
            const uint32_t lit_ext = lit >= 15 ? (lit - 15) / 255 + 1 : 0;
            const uint32_t ml_ext = matchCode >= 15 ? (matchCode - 15) / 255 + 1 : 0;
            if (op + 1 + lit_ext + lit + 2 + (1 + LZ4_LASTLITERALS) + (matchCode + 240) / 255 > olimit) { failed = true; break; }
            const uint32_t token = ((lit < 15 ? lit : 15u) << 4) | (matchCode < 15 ? matchCode : 15u);
            const uint32_t seq_bytes = 1 + lit_ext + lit + 2 + ml_ext;

            if (seq_bytes <= 64 && lit < 15) {
                // whole sequence at once into the LDS stage: lane k writes byte k
                o.reserve(op, seq_bytes);
                const uint32_t k = (uint32_t)lane;
                uint32_t v = token;
                if (k >= 1 && k <= lit) {
                    if (anchor >= w.wlo && ip0 <= w.hi_valid()) v = w.lds8(anchor + k - 1);   // uniform: literals resident
                    else v = w.rd8(anchor + k - 1);
                }
                else if (k == lit + 1) v = offset & 0xffu;
                else if (k == lit + 2) v = offset >> 8;
                else if (k > lit + 2) {
                    const uint32_t j = k - (lit + 3);
                    v = (j + 1 < ml_ext) ? 255u : (matchCode - 15u - (ml_ext - 1) * 255u);
                }
                if (k < seq_bytes) *o.at(op + k) = (uint8_t)v;
                op += seq_bytes;
            }
            else {
                o.flush(op);
                if (lane == 0) dst[op] = (uint8_t)token;
                op += 1;
                if (lit >= 15) {
                    const uint32_t rest = lit - 15;
                    const uint32_t n255 = rest / 255;
                    for (uint32_t i = lane; i < n255; i += 64) dst[op + i] = 255;
                    if (lane == 0) dst[op + n255] = (uint8_t)(rest - n255 * 255);
                    op += n255 + 1;
                }
                wave_copy(dst + op, w, anchor, lit, lane);
                op += lit;
                if (lane == 0) { dst[op] = (uint8_t)offset; dst[op + 1] = (uint8_t)(offset >> 8); }
                op += 2;
                if (matchCode >= 15) {
                    const uint32_t rest = matchCode - 15;
                    const uint32_t n255 = rest / 255;
                    for (uint32_t i = lane; i < n255; i += 64) dst[op + i] = 255;
                    if (lane == 0) dst[op + n255] = (uint8_t)(rest - n255 * 255);
                    op += n255 + 1;
                }
                o.base = op;
            }
            

            const uint32_t ipn = q + ml;                              // = original ip + 4 + forward count
            anchor = ipn;
            
            if (ipn >= mflimitPlusOne) break;

            // LZ4_putPosition(ip - 2) happens at the top of the next batch; the unified batch then starts with
            // the "test next position" probe at ip
            put2 = ipn - 2;
            P = ipn;
            U = 0;
            SQY_EXP(x_lap(5);)                                  // 5: generic emission
            SQY_STAMP(10);
        }
        if (pend && !failed) emit_pending();
    }

    if (!failed && !redo_dense) {
        o.flush(op);
        const uint32_t lastRun = pend - anchor;
        if (op + lastRun + 1 + (lastRun + 255 - 15) / 255 > olimit) {
            failed = true;
        } else {
            if (lastRun >= 15) {
                const uint32_t rest = lastRun - 15;
                const uint32_t n255 = rest / 255;
                if (lane == 0) dst[op] = 0xF0;
                for (uint32_t i = lane; i < n255; i += 64) dst[op + 1 + i] = 255;
                if (lane == 0) dst[op + 1 + n255] = (uint8_t)(rest - n255 * 255);
                op += n255 + 2;
            } else {
                if (lane == 0) dst[op] = (uint8_t)(lastRun << 4);
                op += 1;
            }
            wave_copy(dst + op, w, anchor, lastRun, lane);
            op += lastRun;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no ring copy may still be in flight when the LDS is released / refilled
#ifdef SQY_EXP_STATS
    if constexpr (!LINKED && !DENSE && !ACCEL) if (lane == 0 && g_exp_buf) {
        const unsigned long long slot = atomicAdd(g_exp_buf, 1ull);
        if (slot < g_exp_cap) {
            unsigned long long* r = g_exp_buf + 8 + slot * 8;
            uint32_t hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            r[0] = x_start; r[1] = __builtin_amdgcn_s_memrealtime(); r[2] = __builtin_amdgcn_s_memtime() - x_c0;
            r[3] = w.x_wait | ((unsigned long long)w.x_nwait << 40); r[4] = x_far | ((unsigned long long)x_nfar << 40);
            r[5] = (unsigned long long)blockIdx.x | ((unsigned long long)x_nseq << 32); r[6] = (((unsigned long long)(uintptr_t)scratch >> 12) & 0xffffffffull) | ((unsigned long long)(xcc & 15u) << 32) | ((unsigned long long)hwid << 36);
            r[7] = (unsigned long long)x_ngen | ((unsigned long long)(failed ? 0u : op) << 32);
            x_lap(6);                                            // 6: tail (last literals, flush)
            r[2] = (x_acc[0] >> 8) | ((x_acc[1] >> 8) << 21) | ((x_acc[2] >> 8) << 42);         // (units of 256 ticks, 21 bits each)
            r[3] |= 0;                                           // (ring waits stay)
            r[4] |= 0;
            r[6] = (r[6] & 0xffffffffull) | (((x_acc[3] >> 8) & 0xffffull) << 32) | (((x_acc[4] >> 8) & 0xffffull) << 48);
            r[7] = (r[7] & 0xffffffff0000ffffull) | (((x_acc[5] >> 8) & 0xffffull) << 16);
        }
    }
#endif
    if (lane == 0) {
        if (!LINKED && !DENSE && redo_dense) redo_list[1u + atomicAdd(&redo_list[0], 1u)] = (uint32_t)blk;
        else if (!LINKED || (bi >= b_out && !redo_dense)) csize[blk] = failed ? 0u : op;      // (LINKED && redo_dense: given up, see below)
    }
    if (LINKED) __syncthreads();
    if constexpr (LINKED) if (spec_mode && bi >= b_out) {
        uint32_t* __restrict__ tf = dd.tables + (uint64_t)bi * kLz4SpecTableWords + 4096u;
#pragma unroll 4
        for (int i = 0; i < 64; ++i) tf[i * 64 + lane] = table[i * 64 + lane];
    }
    if constexpr (LINKED && !DENSE) if (redo_dense) {
        // given up (mode 1, see the lean loop): a table entry with bit 31 set exists in no table -- the block this wavefront was to
        // deliver neither passes the check nor lets the block behind it pass
        if (lane == 0) {
            uint32_t* __restrict__ t0 = dd.tables + (uint64_t)(b_last - 1u) * kLz4SpecTableWords;
            __hip_atomic_store(&t0[0], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&t0[4096], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        break;
    }
  }
#ifdef SQY_LZ4_DIAG
    if (lane == 0) { diag[blockIdx.x * 32] = dacc; diag[blockIdx.x * 32 + 1] = dcnt; for (int i = 0; i < 16; ++i) diag[blockIdx.x * 32 + 8 + i] = dreason[i]; }
#endif
}

// ------------------------------------------------------------------------------------------------
// frame layout: exclusive scan of frame sizes, then scatter  [7 B header][u32 size][data][u32 0]
// ------------------------------------------------------------------------------------------------
// blocks == nullptr: every chunk is a single-block frame.  Otherwise block k of the list contributes its 4-byte size
// field and body, plus the 7-byte frame header when it opens a frame and the 4-byte end mark when it closes one.
// tail_info != nullptr (frames in place): [0] = j, the first chunk of the run of stored chunks that ends the stream (nchunks when
// the last chunk compressed), [1] = bytes of the frames in front of it, [2] = stored chunks among those, [3] = payload bytes
// Round 4 (one host round trip per call): `guard` != nullptr and guard[0] != 0 -- the parse left chunks to the dense second pass,
// the sizes are not final -- makes this kernel and the ones behind it (stash, gather, finish) return at once; the host runs the
// second pass and launches them again without a guard.  body0 != nullptr: the tail marks (frame header, size field, end mark of
// the stored chunks that end the stream, lz4_tail_marks_kernel) are written here as well.
__global__ __launch_bounds__(1024)
void lz4_frame_scan_kernel(const uint32_t* __restrict__ csize, uint64_t nchunks, uint64_t total, uint32_t chunk,
                           uint64_t* __restrict__ frame_off /* nchunks + 1 */, const Lz4Block* __restrict__ blocks,
                           const uint32_t* __restrict__ dup_of, uint64_t* __restrict__ tail_info, const uint32_t* __restrict__ guard,
                           uint8_t* __restrict__ body0, uint64_t in_stride, uint32_t bd_byte, uint32_t hc_byte)
{
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry_s;
    __shared__ uint32_t last_comp_s, raw_head_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nt = blockDim.x;                            // (round 6: launched with 256 threads -- a block of sixteen waves waits for a CU to drain when calls are in flight)
    if (guard && guard[0] != 0u) return;
    if (tid == 0) { carry_s = 0; last_comp_s = 0; raw_head_s = 0; }
    __syncthreads();
    for (uint64_t base = 0; base < nchunks; base += nt) {
        const uint64_t k = base + tid;
        uint64_t sz = 0;
        if (k < nchunks) {
            const uint32_t c = csize[dup_of ? dup_of[k] : k];
            if (tail_info && c) atomicMax(&last_comp_s, (uint32_t)k + 1u);
            if (blocks) {
                const Lz4Block bd = blocks[k];
                sz = ((bd.flags & 1u) ? 7 : 0) + 4 + (c ? c : bd.n) + ((bd.flags & 2u) ? 4 : 0);
            } else {
                const uint64_t left = total - k * chunk;
                const uint64_t nk = left < chunk ? left : chunk;
                sz = 7 + 4 + (c ? c : nk) + 4;
            }
        }
        // inclusive scan inside the wave
        uint64_t x = sz;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint64_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const uint64_t carry = carry_s;
        if (k < nchunks) frame_off[k] = carry + woff + x - sz;
        __syncthreads();
        if ((uint32_t)tid == nt - 1u) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0) frame_off[nchunks] = carry_s;
    if (tail_info) {
        __syncthreads();
        const uint32_t j = last_comp_s;
        uint32_t nraw = 0;
        for (uint64_t k = tid; k < j; k += nt) nraw += csize[dup_of ? dup_of[k] : k] == 0u ? 1u : 0u;
        if (nraw) atomicAdd(&raw_head_s, nraw);
        __syncthreads();
        if (tid == 0) {
            tail_info[0] = j;
            tail_info[1] = frame_off[j];
            tail_info[2] = raw_head_s;
            tail_info[3] = carry_s;
        }
        if (body0) {
            for (uint64_t k = (uint64_t)j + tid; k < nchunks; k += nt) {
                const uint64_t left = total - k * chunk;
                const uint32_t nk = (uint32_t)(left < chunk ? left : chunk);
                uint8_t* b = body0 + k * in_stride;
                const uint32_t field = nk | 0x80000000u;
                b[-11] = 0x04; b[-10] = 0x22; b[-9] = 0x4D; b[-8] = 0x18; b[-7] = 0x40; b[-6] = (uint8_t)bd_byte; b[-5] = (uint8_t)hc_byte;
                b[-4] = (uint8_t)field; b[-3] = (uint8_t)(field >> 8); b[-2] = (uint8_t)(field >> 16); b[-1] = (uint8_t)(field >> 24);
                b[nk] = 0; b[nk + 1] = 0; b[nk + 2] = 0; b[nk + 3] = 0;
            }
        }
    }
}

// Frames in place: the stage in front of lz4 wrote chunk k of the stream at body0 + k * in_stride (in_stride = chunk + 15).  The
// stored chunks that END the stream (k >= tail_info[0]) are final where they are -- they only get their 7-byte frame header,
// size field and end mark written around them; the frames in front are gathered so that they end where that run begins.
__global__ __launch_bounds__(256)
void lz4_tail_marks_kernel(uint8_t* __restrict__ body0, uint64_t in_stride, uint64_t total, uint32_t chunk, uint64_t nchunks,
                           uint32_t bd_byte, uint32_t hc_byte, const uint64_t* __restrict__ tail_info)
{
    const uint64_t k = tail_info[0] + (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (k >= nchunks) return;
    const uint64_t left = total - k * chunk;
    const uint32_t nk = (uint32_t)(left < chunk ? left : chunk);
    uint8_t* b = body0 + k * in_stride;
    const uint32_t field = nk | 0x80000000u;
    b[-11] = 0x04; b[-10] = 0x22; b[-9] = 0x4D; b[-8] = 0x18; b[-7] = 0x40; b[-6] = (uint8_t)bd_byte; b[-5] = (uint8_t)hc_byte;
    b[-4] = (uint8_t)field; b[-3] = (uint8_t)(field >> 8); b[-2] = (uint8_t)(field >> 16); b[-1] = (uint8_t)(field >> 24);
    b[nk] = 0; b[nk + 1] = 0; b[nk + 2] = 0; b[nk + 3] = 0;
}

// Stored chunks IN FRONT of that run move (towards higher addresses, over their own and their neighbours' old places): they are
// first put aside in their -- unused -- compressed-output slots; the gather then reads every frame body from the scratch.
// tail_info != nullptr (round 4): how many chunks sit in front of the stored tail, and whether any of them is stored at all, is read
// on the device (tail_info[0], [2]); the grid is a fixed number of blocks that share the (chunk, slice) items among them.
__global__ __launch_bounds__(256)
void lz4_stash_raw_kernel(const uint8_t* __restrict__ body0, uint64_t in_stride, uint64_t total, uint32_t chunk,
                          uint8_t* __restrict__ scratch, uint64_t stride, const uint32_t* __restrict__ csize,
                          const uint32_t* __restrict__ dup_of, uint32_t slices_per_chunk, const uint64_t* __restrict__ tail_info,
                          const uint32_t* __restrict__ guard)
{
    uint64_t nitems = gridDim.x;
    if (tail_info) {
        if (guard && guard[0] != 0u) return;
        if (tail_info[2] == 0) return;
        nitems = tail_info[0] * slices_per_chunk;
    }
    for (uint64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        const uint64_t k = item / slices_per_chunk;
        const uint32_t slice = (uint32_t)(item % slices_per_chunk);
        if (csize[dup_of ? dup_of[k] : k] != 0u) continue;
        const uint64_t left = total - k * chunk;
        const uint32_t nk = (uint32_t)(left < chunk ? left : chunk);
        const uint32_t begin = slice * 32768u;
        if (begin >= nk) continue;
        const uint32_t end = begin + 32768u < nk ? begin + 32768u : nk;
        const uint8_t* __restrict__ sp = body0 + k * in_stride + begin;
        uint8_t* __restrict__ dp = scratch + k * stride + begin;            // 16-byte aligned (stride and slices are)
        const uint32_t len = end - begin, nvec = len >> 4;
        for (uint32_t i = threadIdx.x; i < nvec; i += 256) *reinterpret_cast<uint4*>(dp + (size_t)i * 16) = ld_u128(sp + (size_t)i * 16);
        const uint32_t done = nvec << 4;
        if (threadIdx.x < len - done) dp[done + threadIdx.x] = sp[done + threadIdx.x];
    }
}

// one workgroup per (chunk, slice): copies its slice of the frame body, slice 0 also writes header/trailer
constexpr uint32_t GATHER_SLICE = 32768;

__global__ __launch_bounds__(256)
void lz4_frame_gather_kernel(const uint8_t* __restrict__ in, uint64_t total, uint32_t chunk,
                             const uint8_t* __restrict__ scratch, uint64_t stride,
                             const uint32_t* __restrict__ csize, const uint64_t* __restrict__ frame_off,
                             uint8_t* __restrict__ out, uint32_t bd_byte, uint32_t hc_byte, uint32_t slices_per_chunk,
                             const uint64_t* __restrict__ fmap, uint64_t fbytes, const Lz4Block* __restrict__ blocks,
                             const uint32_t* __restrict__ dup_of, uint64_t in_stride, int raw_from_scratch,
                             const uint64_t* __restrict__ tail_info, uint64_t t0, const uint32_t* __restrict__ guard)
{
    // tail_info != nullptr (round 4, frames in place): how many frames there are to gather (tail_info[0]), where they go (they end
    // where the stored tail begins: out + t0 + [0] * in_stride - [1]) and whether stored chunks among them were put aside in the
    // scratch ([2]) is read on the device; the grid is a fixed number of blocks that share the (chunk, slice) items among them
    uint64_t nitems = gridDim.x;
    if (tail_info) {
        if (guard && guard[0] != 0u) return;
        nitems = tail_info[0] * slices_per_chunk;
        out += t0 + tail_info[0] * in_stride - tail_info[1];
        raw_from_scratch = tail_info[2] != 0;
    }
  for (uint64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
    const uint64_t k = item / slices_per_chunk;
    const uint32_t slice = (uint32_t)(item % slices_per_chunk);
    uint32_t nk, flags = 3u;
    uint64_t lin;
    if (blocks) { const Lz4Block bd = blocks[k]; nk = bd.n; lin = bd.start; flags = bd.flags; }
    else {
        const uint64_t left = total - k * chunk;
        nk = (uint32_t)(left < chunk ? left : chunk);
        lin = k * chunk;
    }
    const uint64_t ks = dup_of ? dup_of[k] : k;                      // the chunk whose compressed bytes this one shares
    const uint32_t c = csize[ks];
    const uint32_t body = c ? c : nk;
    // a stored chunk: from the stream (chunk k at in + k * in_stride in the chunked layout), or from its own scratch slot when
    // it was put aside there (lz4_stash_raw_kernel)
    const uint8_t* __restrict__ s = c ? scratch + ks * stride
                                      : raw_from_scratch ? scratch + k * stride
                                      : fmap ? in + fmap[lin / fbytes] * fbytes + lin % fbytes
                                      : blocks ? in + lin : in + k * in_stride;
    uint8_t* __restrict__ d = out + frame_off[k];
    const int tid = threadIdx.x;
    const uint32_t hdr = (flags & 1u) ? 7u : 0u;                     // frame header in front of the block's size field

    if (slice == 0 && tid < 15) {
        const uint32_t field = c ? c : (nk | 0x80000000u);
        uint8_t v = 0;
        uint32_t o = tid;
        bool on = true;
        switch (tid) {
            case 0: v = 0x04; break; case 1: v = 0x22; break; case 2: v = 0x4D; break; case 3: v = 0x18; break;
            case 4: v = 0x40; break; case 5: v = (uint8_t)bd_byte; break; case 6: v = (uint8_t)hc_byte; break;
            case 7: v = (uint8_t)field; break; case 8: v = (uint8_t)(field >> 8); break;
            case 9: v = (uint8_t)(field >> 16); break; case 10: v = (uint8_t)(field >> 24); break;
            default: v = 0; o = 11 + body + (tid - 11); on = (flags & 2u) != 0; break;       // end mark
        }
        if (tid < 7) on = hdr != 0;
        if (on) d[o - (7u - hdr)] = v;
    }
    const uint32_t begin = slice * GATHER_SLICE;
    if (begin >= body) continue;
    const uint32_t end = (begin + GATHER_SLICE < body) ? begin + GATHER_SLICE : body;
    uint8_t* __restrict__ dd = d + hdr + 4 + begin;
    const uint8_t* __restrict__ ss = s + begin;
    uint32_t len = end - begin;
    // head to 16-byte alignment of the destination
    const uint32_t head0 = (uint32_t)((16 - (reinterpret_cast<uintptr_t>(dd) & 15)) & 15);
    const uint32_t head = head0 < len ? head0 : len;
    if ((uint32_t)tid < head) dd[tid] = ss[tid];
    dd += head; ss += head; len -= head;
    const uint32_t nvec = len >> 4;
    // four loads in flight per thread before the first store (a 32 KiB slice is 8 vectors per thread)
    for (uint32_t i0 = tid; i0 < nvec; i0 += 1024) {
        uint4 v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) if (i0 + j * 256u < nvec) v[j] = ld_u128(ss + (size_t)(i0 + j * 256u) * 16);
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) if (i0 + j * 256u < nvec) *reinterpret_cast<uint4*>(dd + (size_t)(i0 + j * 256u) * 16) = v[j];
    }
    const uint32_t done = nvec << 4;
    if ((uint32_t)tid < len - done) dd[done + tid] = ss[done + tid];
  }
}

// Frames in place, last kernel of a call (round 4): the sqy header is written on the device -- the text in front of and behind the
// payload's byte count comes as a kernel argument, the count itself (tail_info[3]) is formatted here, the header ends where the
// payload begins -- and what the host has to know goes to `record` (pinned host memory): [0] status (1 done, 2 the dense second
// pass is needed, 3 the header does not fit in front of the payload), [1] blob offset, [2] blob bytes, [3] payload bytes,
// [4] chunks in front of the stored tail, [5] stored chunks among them, [6] chunks left to the dense pass.
// One host round trip per call instead of three (dense list, tail, header).
struct Lz4HeaderParts { uint32_t prefix_len, suffix_len; char text[3000]; };

__global__ __launch_bounds__(256)
void lz4_inplace_finish_kernel(uint8_t* __restrict__ out, uint64_t t0, uint64_t in_stride, const uint64_t* __restrict__ tail_info,
                               Lz4HeaderParts hp, uint32_t elem_size, const uint32_t* __restrict__ guard, uint64_t* __restrict__ record)
{
    const uint32_t tid = threadIdx.x;
    if (guard && guard[0] != 0u) {
        if (tid == 0) { record[6] = guard[0]; record[0] = 2; }
        return;
    }
    const uint64_t tail_j = tail_info[0], head_bytes = tail_info[1], payload = tail_info[3];
    const uint64_t payload_at = t0 + tail_j * in_stride - head_bytes;
    char digits[20];
    uint32_t nd = 0;
    {
        uint64_t v = payload;
        do { digits[nd++] = (char)('0' + v % 10); v /= 10; } while (v);       // (least significant first)
    }
    const uint64_t text = (uint64_t)hp.prefix_len + nd + hp.suffix_len;
    const uint64_t pad = (elem_size - text % elem_size) % elem_size;           // sqeazy_header.hpp:172-178: the header's size is a multiple of the voxel's
    const uint64_t hdr_len = text + pad;
    if (hdr_len > payload_at) {
        if (tid == 0) record[0] = 3;
        return;
    }
    uint8_t* h = out + payload_at - hdr_len;
    for (uint64_t i = tid; i < hdr_len; i += 256) {
        uint8_t c;
        if (i < pad) c = ' ';
        else if (i < pad + hp.prefix_len) c = (uint8_t)hp.text[i - pad];
        else if (i < pad + hp.prefix_len + nd) c = (uint8_t)digits[nd - 1 - (i - pad - hp.prefix_len)];
        else c = (uint8_t)hp.text[hp.prefix_len + (i - pad - hp.prefix_len - nd)];
        h[i] = c;
    }
    if (tid == 0) {
        record[1] = payload_at - hdr_len;
        record[2] = hdr_len + payload;
        record[3] = payload;
        record[4] = tail_j;
        record[5] = tail_info[2];
        record[6] = 0;
        record[0] = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// quantiser (encoders/quantiser_scheme_impl.hpp:176-226): 65536-bin histogram, then LUT apply.
//
// Histogram: every workgroup keeps a private 16384-bin window of the value range in LDS (64 KiB) and
// counts there with LDS atomics; microscopy stacks concentrate in a narrow band, so nearly every voxel
// lands in the window.  Values outside go to the global histogram directly.  The window a workgroup uses
// is voted from the first voxels it sees (quarter of the range with the most hits).
// ------------------------------------------------------------------------------------------------
constexpr int HIST_WIN_BINS = 16384;

__global__ __launch_bounds__(256)
void histogram_u16_kernel(const uint16_t* __restrict__ in, uint64_t len, uint32_t* __restrict__ histo /* 65536, zeroed */)
{
    extern __shared__ uint32_t hwin[];      // HIST_WIN_BINS counters
    __shared__ uint32_t votes[4];
    const int tid = threadIdx.x;
    for (int i = tid; i < HIST_WIN_BINS; i += 256) hwin[i] = 0;
    if (tid < 4) votes[tid] = 0;
    __syncthreads();

    const uint64_t nvec = len / 8;                                   // 8 voxels per 16-byte load
    const uint64_t per_block = (nvec + gridDim.x - 1) / gridDim.x;
    const uint64_t v0 = (uint64_t)blockIdx.x * per_block;
    const uint64_t v1 = (v0 + per_block < nvec) ? v0 + per_block : nvec;
    const uint4* src = reinterpret_cast<const uint4*>(in);

    if (v0 + tid < v1) {
        const uint4 x = src[v0 + tid];
        atomicAdd(&votes[(x.x & 0xffffu) >> 14], 1u);
    }
    __syncthreads();
    uint32_t best = 0;
#pragma unroll
    for (uint32_t q = 1; q < 4; ++q) if (votes[q] > votes[best]) best = q;
    const uint32_t wbase = best * HIST_WIN_BINS;

    for (uint64_t vb = v0 + tid; vb < v1; vb += 1024) {                 // four loads in flight per thread
        uint4 xs[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) if (vb + u * 256u < v1) xs[u] = src[vb + u * 256u];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            if (vb + u * 256u >= v1) continue;
            const uint4 x = xs[u];
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t a = w[j] & 0xffffu, b = w[j] >> 16;
                const uint32_t ra = a - wbase, rb = b - wbase;
                if (ra < (uint32_t)HIST_WIN_BINS) atomicAdd(&hwin[ra], 1u); else atomicAdd(&histo[a], 1u);
                if (rb < (uint32_t)HIST_WIN_BINS) atomicAdd(&hwin[rb], 1u); else atomicAdd(&histo[b], 1u);
            }
        }
    }
    // tail voxels (len % 8) by the last block
    if (blockIdx.x == gridDim.x - 1) {
        for (uint64_t i = nvec * 8 + tid; i < len; i += 256) atomicAdd(&histo[in[i]], 1u);
    }
    __syncthreads();
    for (int i = tid; i < HIST_WIN_BINS; i += 256) {
        const uint32_t c = hwin[i];
        if (c) atomicAdd(&histo[wbase + i], c);
    }
}

// out[i] = lut[in[i]] with the 64 KiB encode LUT staged in LDS (quantiser_utils.hpp:26-42)
__global__ __launch_bounds__(256)
void quantiser_apply_u16_kernel(const uint16_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t len,
                                const uint8_t* __restrict__ lut /* 65536 */)
{
    extern __shared__ uint8_t slut[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 65536 / 16; i += 256)
        reinterpret_cast<uint4*>(slut)[i] = reinterpret_cast<const uint4*>(lut)[i];
    __syncthreads();
    const uint64_t nvec = len / 8;
    const uint4* src = reinterpret_cast<const uint4*>(in);
    uint2* dst = reinterpret_cast<uint2*>(out);
    const uint64_t step = (uint64_t)gridDim.x * 256;
    for (uint64_t v0 = (uint64_t)blockIdx.x * 256 + tid; v0 < nvec; v0 += 4 * step) {           // four loads in flight per thread
        uint4 xs[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) if (v0 + j * step < nvec) xs[j] = src[v0 + j * step];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            if (v0 + j * step >= nvec) continue;
            const uint4 x = xs[j];
            uint2 y;
            y.x = (uint32_t)slut[x.x & 0xffffu] | ((uint32_t)slut[x.x >> 16] << 8) | ((uint32_t)slut[x.y & 0xffffu] << 16) |
                  ((uint32_t)slut[x.y >> 16] << 24);
            y.y = (uint32_t)slut[x.z & 0xffffu] | ((uint32_t)slut[x.z >> 16] << 8) | ((uint32_t)slut[x.w & 0xffffu] << 16) |
                  ((uint32_t)slut[x.w >> 16] << 24);
            dst[v0 + j * step] = y;
        }
    }
    if (blockIdx.x == 0) {
        for (uint64_t i = nvec * 8 + tid; i < len; i += 256) out[i] = slut[in[i]];
    }
}

// (round 5) quantiser with an 8-bit bitswap1 right behind it (quantiser->bitswap1->..: the BASELINE pipeline): the look-up and the
// bit-plane transpose of the sink's bytes in one pass -- out is the 8 plane segments of seg = len / 8 bytes, MSB plane first, byte w of
// plane b holds voxels 8w .. 8w+7 with voxel 8w + j at bit 7 - j (bitswap_scheme_impl.hpp:97-145 on `char`), the len % 8 tail bytes
// copied behind them.  One thread per 8 voxels: 16 bytes in, one byte into each plane (coalesced across the threads); the pass that
// wrote the sink's bytes and the pass that read them again (1 + 1 bytes per voxel of HBM traffic) are gone.
__global__ __launch_bounds__(256)
void quantiser_apply_bitswap1_u8_kernel(const uint16_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t len,
                                        const uint8_t* __restrict__ lut /* 65536 */)
{
    extern __shared__ uint8_t slut[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 65536 / 16; i += 256)
        reinterpret_cast<uint4*>(slut)[i] = reinterpret_cast<const uint4*>(lut)[i];
    __syncthreads();
    const uint64_t seg = len / 8;
    const uint4* src = reinterpret_cast<const uint4*>(in);
    const uint64_t step = (uint64_t)gridDim.x * 256;
    for (uint64_t v0 = (uint64_t)blockIdx.x * 256 + tid; v0 < seg; v0 += 4 * step) {           // four loads in flight per thread
        uint4 xs[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) if (v0 + j * step < seg) xs[j] = src[v0 + j * step];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint64_t w = v0 + j * step;
            if (w >= seg) continue;
            const uint4 x = xs[j];
            // byte j of t = the quantised voxel 8w + j
            uint64_t t = (uint64_t)slut[x.x & 0xffffu] | ((uint64_t)slut[x.x >> 16] << 8) | ((uint64_t)slut[x.y & 0xffffu] << 16) |
                         ((uint64_t)slut[x.y >> 16] << 24) | ((uint64_t)slut[x.z & 0xffffu] << 32) | ((uint64_t)slut[x.z >> 16] << 40) |
                         ((uint64_t)slut[x.w & 0xffffu] << 48) | ((uint64_t)slut[x.w >> 16] << 56);
            // 8x8 bit transpose (bitswap1_u8_generic): afterwards byte b of t holds bit b of every voxel, voxel j at bit j
            uint64_t y;
            y = (t ^ (t >> 7)) & 0x00AA00AA00AA00AAull; t = t ^ y ^ (y << 7);
            y = (t ^ (t >> 14)) & 0x0000CCCC0000CCCCull; t = t ^ y ^ (y << 14);
            y = (t ^ (t >> 28)) & 0x00000000F0F0F0F0ull; t = t ^ y ^ (y << 28);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint32_t byte = (uint32_t)(t >> (8 * b)) & 0xffu;
                out[(uint64_t)(7 - b) * seg + w] = (uint8_t)(__brev(byte) >> 24);
            }
        }
    }
    if (blockIdx.x == 0) {
        for (uint64_t i = seg * 8 + tid; i < len; i += 256) out[i] = slut[in[i]];
    }
}

// ------------------------------------------------------------------------------------------------
// frame_shuffle (encoders/frame_shuffle_utils.hpp:91-172).
// metric[z] = (float sum of frame z accumulated SEQUENTIALLY in binary32) / (Y*X): the additions round once
// the sum passes 2^24, so the order is part of the result -- one lane walks one frame in index order.
// Lanes of a wave own 64 consecutive frames and read 16 bytes at a time (the 128-byte line is reused by the
// lane's next 7 loads).  The permuted copy is a plain frame gather.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64)
void frame_metric_kernel(const T* __restrict__ in, uint64_t Z, uint64_t per_frame, float* __restrict__ metric)
{
    const uint64_t z = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (z >= Z) return;
    const T* p = in + z * per_frame;
    float sum = 0.f;
    uint64_t i = 0;
    constexpr uint32_t per_vec = 16 / sizeof(T);
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const uint4* pv = reinterpret_cast<const uint4*>(p);
        const uint64_t nv = per_frame / per_vec;
        uint64_t v = 0;
        // one cache line (8 x 16 B) per step, two steps ahead in registers: the loads of steps k+1 and k+2 travel while
        // step k is added -- strictly in index order, the additions are the reference's rounding sequence
        if (nv >= 24) {
            uint4 a[8], b[8], c[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { a[k] = pv[k]; b[k] = pv[8 + k]; }
            for (; v + 24 <= nv; v += 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) c[k] = pv[v + 16 + k];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t w[4] = {a[k].x, a[k].y, a[k].z, a[k].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (sizeof(T) == 2) {
                            sum = sum + (float)(w[j] & 0xffffu);
                            sum = sum + (float)(w[j] >> 16);
                        } else {
                            sum = sum + (float)(w[j] & 0xffu);
                            sum = sum + (float)((w[j] >> 8) & 0xffu);
                            sum = sum + (float)((w[j] >> 16) & 0xffu);
                            sum = sum + (float)(w[j] >> 24);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) { a[k] = b[k]; b[k] = c[k]; }
            }
        }
        for (; v < nv; ++v) {
            const uint4 x = pv[v];
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (sizeof(T) == 2) {
                    sum = sum + (float)(w[j] & 0xffffu);
                    sum = sum + (float)(w[j] >> 16);
                } else {
                    sum = sum + (float)(w[j] & 0xffu);
                    sum = sum + (float)((w[j] >> 8) & 0xffu);
                    sum = sum + (float)((w[j] >> 16) & 0xffu);
                    sum = sum + (float)(w[j] >> 24);
                }
            }
        }
        i = nv * per_vec;
    }
    for (; i < per_frame; ++i) sum = sum + (float)p[i];
    metric[z] = sum;                           // the division by Y*X happens on the host (sqy::frame_shuffle_order)
}

// frames of SIGNED bytes (frame_shuffle as a tail filter on the sink's `char` output): the running sum is not monotone, the
// block-parallel evaluation below does not apply -- one lane adds one frame in index order
__global__ __launch_bounds__(64)
void frame_metric_i8_kernel(const int8_t* __restrict__ in, uint64_t Z, uint64_t per_frame, float* __restrict__ metric)
{
    const uint64_t z = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (z >= Z) return;
    const int8_t* __restrict__ p = in + z * per_frame;
    float sum = 0.f;
    uint64_t i = 0;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const uint4* pv = reinterpret_cast<const uint4*>(p);
        const uint64_t nv = per_frame / 16;
        for (uint64_t v = 0; v < nv; ++v) {
            const uint4 x = pv[v];
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sum = sum + (float)(int8_t)(w[j] & 0xffu);
                sum = sum + (float)(int8_t)((w[j] >> 8) & 0xffu);
                sum = sum + (float)(int8_t)((w[j] >> 16) & 0xffu);
                sum = sum + (float)(int8_t)(w[j] >> 24);
            }
        }
        i = nv * 16;
    }
    for (; i < per_frame; ++i) sum = sum + (float)p[i];
    metric[z] = sum;
}

// Block-parallel EXACT evaluation of the same sequential binary32 sum.
// While the running sum S (an integer) stays inside one binade [2^e, 2^(e+1)) with ulp u = 2^(e-23) >= 2, adding an integer v
// rounds S + v to a multiple of u (ties to the even multiple): how it rounds depends on v, on the binade and on the PARITY of S / u,
// on nothing else of S.  So what a run of elements does to the sum is a map  parity in -> (increment, parity out), maps compose, and a
// map is obtained by letting the hardware's float add do the rounding on a stand-in for S with the right binade and parity
// (FmBlock::lane_map).  A wave splits a 4 KiB block over its 64 lanes, each lane adds its 64 bytes to both stand-ins, the 64 lane maps
// are composed in lane order by a tree (FmBlock::block_map).  Blocks that may cross into the next binade (at most a dozen per frame)
// are added by the lanes one after the other, starting from the real sum.  Below 2^24 everything is exact.
// Checked against numpy's sequential float32 cumsum (tests) and, end to end, against the oracle's C loop.
// building blocks of the exact evaluation, one 4 KiB block of a frame per wavefront step
template <typename T>
struct FmBlock {
    static constexpr uint32_t EPL = 64 / sizeof(T);          // elements per lane and block (64 bytes)
    static constexpr uint32_t BLK = 64 * EPL;

    // this lane's 64 bytes of block b (zero padded past the frame: adding 0 never changes the sum)
    static __device__ __forceinline__ void load(const T* __restrict__ p, uint64_t per_frame, uint64_t b, int lane, uint32_t x[EPL])
    {
        const uint64_t e0 = b * BLK + (uint64_t)lane * EPL;
        if (e0 + EPL <= per_frame) {
            const uint4* q = reinterpret_cast<const uint4*>(p + e0);
            uint32_t w[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const uint4 t = q[j]; w[4 * j] = t.x; w[4 * j + 1] = t.y; w[4 * j + 2] = t.z; w[4 * j + 3] = t.w; }
#pragma unroll
            for (uint32_t i = 0; i < EPL; ++i)
                x[i] = sizeof(T) == 1 ? (w[i >> 2] >> (8 * (i & 3))) & 0xffu : (w[i >> 1] >> (16 * (i & 1))) & 0xffffu;
        } else {
#pragma unroll
            for (uint32_t i = 0; i < EPL; ++i) x[i] = (e0 + i < per_frame) ? (uint32_t)p[e0 + i] : 0u;
        }
    }
    static __device__ __forceinline__ uint32_t wave_sum(const uint32_t x[EPL])     // <= 4096 * 255 or 2048 * 65535 < 2^28
    {
        uint32_t tot = 0;
#pragma unroll
        for (uint32_t i = 0; i < EPL; ++i) tot += x[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);
        return sgpr(tot);
    }
    // What this lane's elements do to the running sum while it stays in the binade of regime k (ulp u = 2^k): a map
    //   parity of S / u coming in  ->  (exact increment of S, parity going out).
    // Only that parity and the binade decide how S + v rounds, so the lane simply ADDS ITS ELEMENTS WITH THE HARDWARE'S FLOAT ADD to
    // two stand-ins for S -- 2^e (parity 0) and 2^e + u (parity 1), e = 23 + k -- : same binade, same parity, hence the same
    // roundings as the real sum, 64 bytes cannot leave the binade from there, and the stand-in's last mantissa bit is the parity going
    // out.  Two v_add_f32 and a convert per element.  (Round 2 tracked remainders, ties and parities symbolically: ~14 instructions per
    // element, the block-record kernel was compute-bound at 0.48 ms per GiB.)
    static __device__ __forceinline__ void lane_map(const uint32_t x[EPL], uint32_t k, uint32_t& q, int32_t& i0, int32_t& i1)
    {
        const uint32_t bits = (k + 23u + 127u) << 23;
        const float b0 = __builtin_bit_cast(float, bits), b1 = __builtin_bit_cast(float, bits | 1u);
        float s0 = b0, s1 = b1;
#pragma unroll
        for (uint32_t i = 0; i < EPL; ++i) {
            const float f = (float)x[i];
            s0 = s0 + f;
            s1 = s1 + f;
        }
        q = (__builtin_bit_cast(uint32_t, s0) & 1u) | ((__builtin_bit_cast(uint32_t, s1) & 1u) << 1);   // parity out for parity in 0 | parity in 1 << 1
        i0 = (int32_t)(s0 - b0);                                       // exact: both multiples of u, less than 2^24 apart
        i1 = (int32_t)(s1 - b1);
    }
    // The 64 lane maps in lane order as an ordered tree reduction: maps compose associatively, six rounds of "lower lanes' map, then
    // upper lanes' map" leave the whole block's map in lane 0 (~70 instructions): increment and parity out for both parities in.
    static __device__ __forceinline__ void block_map(uint32_t q, int32_t d0, int32_t d1, int32_t& inc0, uint32_t& par0, int32_t& inc1, uint32_t& par1)
    {
#pragma unroll
        for (int s_ = 1; s_ < 64; s_ <<= 1) {
            // the map of the 2 s_ lanes that start here = (upper s_ lanes) after (lower s_ lanes); only lanes that are a multiple of
            // 2 s_ hold a meaningful result, the others compute along
            const uint32_t uq = (uint32_t)__shfl_down((int)q, s_);
            const int32_t ud0 = __shfl_down(d0, s_), ud1 = __shfl_down(d1, s_);
            const uint32_t m0 = q & 1u, m1 = (q >> 1) & 1u;               // parity between the halves for parity in 0 / 1
            d0 += m0 ? ud1 : ud0;
            d1 += m1 ? ud1 : ud0;
            q = ((uq >> m0) & 1u) | (((uq >> m1) & 1u) << 1);
        }
        par0 = lane_read(q, 0) & 1u; par1 = (lane_read(q, 0) >> 1) & 1u;
        inc0 = (int32_t)lane_read((uint32_t)d0, 0); inc1 = (int32_t)lane_read((uint32_t)d1, 0);
    }
    // the block may change binade: lanes add their elements one after the other with the hardware's float add
    static __device__ __forceinline__ uint64_t sequential(const uint32_t x[EPL], uint64_t S, int lane)
    {
        float sf = (float)S;                                           // S is representable: it IS the float sum
        for (int L = 0; L < 64; ++L) {
            if (lane == L) {
#pragma unroll
                for (uint32_t i = 0; i < EPL; ++i) sf = sf + (float)x[i];
            }
            sf = __builtin_bit_cast(float, lane_read(__builtin_bit_cast(uint32_t, sf), L));
        }
        return (uint64_t)sf;
    }
    // one block onto the exact running sum S
    static __device__ __forceinline__ uint64_t apply(const uint32_t x[EPL], uint32_t tot, uint64_t S, int lane)
    {
        if (S + tot < (1ull << 24)) return S + tot;                    // exact regime
        if (S >= (1ull << 24)) {
            const uint32_t e = 63u - (uint32_t)__builtin_clzll(S);
            const uint32_t k = e - 23u;
            if (S + tot + (uint64_t)BLK * ((1ull << k) >> 1) < (1ull << (e + 1))) {
                uint32_t q, p0, p1; int32_t i0, i1, inc0, inc1;
                lane_map(x, k, q, i0, i1);
                block_map(q, i0, i1, inc0, p0, inc1, p1);
                return (uint64_t)((int64_t)S + (((uint32_t)(S >> k) & 1u) ? inc1 : inc0));
            }
        }
        return sequential(x, S, lane);
    }
};

// one wavefront per frame, blocks one after the other (short frames, or no scratch for the planned path below)
template <typename T>
__global__ __launch_bounds__(64)
void frame_metric_scan_kernel(const T* __restrict__ in, uint64_t per_frame, float* __restrict__ metric)
{
    using B = FmBlock<T>;
    const int lane = threadIdx.x;
    const T* __restrict__ p = in + (uint64_t)blockIdx.x * per_frame;
    const uint64_t nblk = (per_frame + B::BLK - 1) / B::BLK;
    uint64_t S = 0;                                        // the float sum, held exactly (it is an integer < 2^40)
    for (uint64_t b = 0; b < nblk; ++b) {
        uint32_t x[B::EPL];
        B::load(p, per_frame, b, lane, x);
        S = B::apply(x, B::wave_sum(x), S, lane);
    }
    if (lane == 0) metric[blockIdx.x] = (float)S;          // exact: S is the float sum
}

// Planned path for long frames: almost all of the work leaves the per-frame serial chain.
//   A  frame_block_sums_kernel        integer sum of every block (parallel over all blocks of all frames)
//   B  frame_block_summaries_kernel   per block: predict the regime from the INTEGER prefix sum of the frame (the float sum
//                                     differs from it by far less than a binade except close to a power of two) and store
//                                     what the block does to the sum for both incoming parities {delta, parity after}
//   C  frame_chain_kernel             one wavefront per frame walks the block records: the true S says whether a record's
//                                     regime holds and the block stays inside the binade -- then it is one add; otherwise
//                                     (a dozen blocks per frame) the block is evaluated from the data as above.
struct FmRecord { int32_t d0, d1; uint32_t info; };          // info: bit 0 valid, bits 8..15 k, bit 16/17 parity after for parity-in 0/1

template <typename T>
__global__ __launch_bounds__(256)
void frame_block_sums_kernel(const T* __restrict__ in, uint64_t per_frame, uint32_t nb, uint64_t nblocks, uint32_t* __restrict__ bsum)
{
    using B = FmBlock<T>;
    const int lane = threadIdx.x & 63;
    const uint64_t g = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= nblocks) return;
    const uint64_t z = g / nb, b = g - z * nb;
    uint32_t x[B::EPL];
    B::load(in + z * per_frame, per_frame, b, lane, x);
    const uint32_t tot = B::wave_sum(x);
    if (lane == 0) bsum[g] = tot;
}

template <typename T>
__global__ __launch_bounds__(256)
void frame_block_summaries_kernel(const T* __restrict__ in, uint64_t per_frame, uint32_t nb, uint64_t nblocks,
                                  const uint32_t* __restrict__ bsum, FmRecord* __restrict__ rec)
{
    using B = FmBlock<T>;
    const int lane = threadIdx.x & 63;
    const uint64_t g = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= nblocks) return;
    const uint64_t z = g / nb, b = g - z * nb;
    // integer prefix of the frame in front of this block
    uint64_t P = 0;
    for (uint64_t i = (uint64_t)lane; i < b; i += 64) P += bsum[z * nb + i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) P += __shfl_xor(P, o);
    const uint32_t tot = bsum[g];
    FmRecord r; r.d0 = 0; r.d1 = 0; r.info = 0;
    if (P >= (1ull << 24)) {
        const uint32_t e = 63u - (uint32_t)__builtin_clzll(P);
        const uint32_t k = e - 23u;
        if (P + tot + (uint64_t)B::BLK * ((1ull << k) >> 1) < (1ull << (e + 1))) {
            uint32_t x[B::EPL];
            B::load(in + z * per_frame, per_frame, b, lane, x);
            uint32_t q, p0, p1; int32_t i0, i1, inc0, inc1;
            B::lane_map(x, k, q, i0, i1);
            B::block_map(q, i0, i1, inc0, p0, inc1, p1);
            r.d0 = inc0 - (int32_t)tot; r.d1 = inc1 - (int32_t)tot;   // (the chain adds the block's integer sum itself)
            r.info = 1u | (k << 8) | (p0 << 16) | (p1 << 17);
        }
    }
    if (lane == 0) rec[g] = r;
}

template <typename T>
__global__ __launch_bounds__(64)
void frame_chain_kernel(const T* __restrict__ in, uint64_t per_frame, uint32_t nb, const uint32_t* __restrict__ bsum,
                        const FmRecord* __restrict__ rec, float* __restrict__ metric)
{
    using B = FmBlock<T>;
    const int lane = threadIdx.x;
    const uint64_t z = blockIdx.x;
    const T* __restrict__ p = in + z * per_frame;
    uint64_t S = 0;
    for (uint32_t base = 0; base < nb; base += 64) {
        // records of the next 64 blocks, one per lane
        const uint32_t mine = base + (uint32_t)lane;
        uint32_t mtot = 0; FmRecord mr; mr.d0 = 0; mr.d1 = 0; mr.info = 0;
        if (mine < nb) { mtot = bsum[z * nb + mine]; mr = rec[z * nb + mine]; }
        const uint32_t cnt = nb - base < 64u ? nb - base : 64u;
        for (uint32_t l = 0; l < cnt; ++l) {
            const uint32_t tot = lane_read(mtot, l);
            if (S + tot < (1ull << 24)) { S += tot; continue; }
            const uint32_t info = lane_read(mr.info, l);
            if (S >= (1ull << 24) && (info & 1u)) {
                const uint32_t e = 63u - (uint32_t)__builtin_clzll(S);
                const uint32_t k = e - 23u;
                if (k == ((info >> 8) & 0xffu) && S + tot + (uint64_t)B::BLK * ((1ull << k) >> 1) < (1ull << (e + 1))) {
                    const uint32_t par = (uint32_t)(S >> k) & 1u;
                    const int32_t d = par ? (int32_t)lane_read((uint32_t)mr.d1, l) : (int32_t)lane_read((uint32_t)mr.d0, l);
                    S = (uint64_t)((int64_t)(S + tot) + (int64_t)d);
                    continue;
                }
            }
            uint32_t x[B::EPL];                                        // the record does not apply: evaluate the block itself
            B::load(p, per_frame, base + l, lane, x);
            S = B::apply(x, tot, S, lane);
        }
    }
    if (lane == 0) metric[z] = (float)S;
}

// ------------------------------------------------------------------------------------------------
// raster_reorder (encoders/raster_reorder_utils.hpp:36-367).  One thread per run of a row that lies inside one tile
// (tile_size voxels, fewer in a remainder tile): a contiguous piece of the volume <-> a contiguous piece of a tile.
//   tile offset(tz,ty,tx) = tz*ts*Y*X + ez*(ty*ts*X + ey*tx*ts)     (ez, ey, ex = extents of the tile)
//   in-tile offset        = (z%ts)*ey*ex + (y%ts)*ex
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void raster_reorder_kernel(const T* __restrict__ in, T* __restrict__ out, uint64_t Z, uint64_t Y, uint64_t X, uint64_t ts,
                           uint64_t TX, bool decode)
{
    const uint64_t run = (uint64_t)blockIdx.x * 256 + threadIdx.x;       // (z, y, tx)
    const uint64_t rows = Z * Y;
    if (run >= rows * TX) return;
    const uint64_t row = run / TX, tx = run - row * TX;
    const uint64_t z = row / Y, y = row - z * Y;
    const uint64_t tz = z / ts, ty = y / ts;
    const uint64_t ez = (tz + 1) * ts <= Z ? ts : Z - tz * ts;
    const uint64_t ey = (ty + 1) * ts <= Y ? ts : Y - ty * ts;
    const uint64_t ex = (tx + 1) * ts <= X ? ts : X - tx * ts;
    const uint64_t lin = row * X + tx * ts;
    const uint64_t til = tz * ts * Y * X + ez * (ty * ts * X + ey * tx * ts) + (z - tz * ts) * ey * ex + (y - ty * ts) * ex;
    const T* s = decode ? in + til : in + lin;
    T* d = decode ? out + lin : out + til;
    if (ex * sizeof(T) == 16 && ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
        *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
    } else {
        for (uint64_t i = 0; i < ex; ++i) d[i] = s[i];
    }
}

// out frame i = in frame map[i]
__global__ __launch_bounds__(256)
void frame_gather_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t frame_bytes,
                         const uint64_t* __restrict__ map, uint32_t blocks_per_frame)
{
    const uint64_t f = blockIdx.x / blocks_per_frame;
    const uint32_t part = blockIdx.x % blocks_per_frame;
    uint8_t* d = out + f * frame_bytes;
    const uint64_t nvec = frame_bytes / 16;
    if (map[f] == ~0ull) {                                       // (tile_shuffle decode: a slot no encoded tile maps to stays zero)
        for (uint64_t i = (uint64_t)part * 256 + threadIdx.x; i < frame_bytes; i += (uint64_t)blocks_per_frame * 256) d[i] = 0;
        return;
    }
    const uint8_t* s = in + map[f] * frame_bytes;
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
        const uint64_t step = (uint64_t)blocks_per_frame * 256;
        for (uint64_t v0 = (uint64_t)part * 256 + threadIdx.x; v0 < nvec; v0 += 4 * step) {   // four loads in flight per thread
            uint4 t[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) if (v0 + j * step < nvec) t[j] = reinterpret_cast<const uint4*>(s)[v0 + j * step];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) if (v0 + j * step < nvec) reinterpret_cast<uint4*>(d)[v0 + j * step] = t[j];
        }
        if (part == 0)
            for (uint64_t i = nvec * 16 + threadIdx.x; i < frame_bytes; i += 256) d[i] = s[i];
    } else {
        for (uint64_t i = (uint64_t)part * 256 + threadIdx.x; i < frame_bytes; i += (uint64_t)blocks_per_frame * 256) d[i] = s[i];
    }
}

// ------------------------------------------------------------------------------------------------
// bitshuffle (encoders/bitshuffle_scheme_impl.hpp:91-100 -> bshuf_bitshuffle of kiyo-masui/bitshuffle, which the reference
// fetches at configure time; not in its tree -- written from the published algorithm).  Per block of `bs` elements: bit row
// r = 8 * (byte of the element) + (bit of that byte) holds that bit of every element, 8 elements per byte, element 8k+j at
// bit j; rows follow each other, bs / 8 bytes each.  The last block is rounded down to a multiple of 8 elements, up to 7
// elements behind it are copied.
// Fast kernels: 16-bit elements, blocks of 4096 (the default): one wavefront per block, lane L owns elements 64L..64L+63
// (128 contiguous bytes in, 8 bytes of every row out); two 16x16 bit transposes on pairs of 16-element groups.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void bitshuffle_u16_4096_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, uint64_t nblocks, bool decode)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave_global = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t wave_stride = (uint64_t)gridDim.x * 4;
    for (uint64_t blk = wave_global; blk < nblocks; blk += wave_stride) {
        const uint16_t* src = in + blk * 4096;
        uint16_t* dst = out + blk * 4096;
        uint32_t ra[16], rb[16];                                 // element pairs of groups (0,1) and (2,3) / row words
        if (!decode) {
            const v4u* p = reinterpret_cast<const v4u*>(src) + lane * 8;
            v4u v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[j];
            // group g = elements 16g..16g+15 = v[2g], v[2g+1]; r[i] = element i of the first group | element i of the second << 16
            const uint32_t g0[8] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
            const uint32_t g1[8] = {v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w};
            const uint32_t g2[8] = {v[4].x, v[4].y, v[4].z, v[4].w, v[5].x, v[5].y, v[5].z, v[5].w};
            const uint32_t g3[8] = {v[6].x, v[6].y, v[6].z, v[6].w, v[7].x, v[7].y, v[7].z, v[7].w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                ra[2 * k] = __builtin_amdgcn_perm(g1[k], g0[k], 0x05040100u);
                ra[2 * k + 1] = __builtin_amdgcn_perm(g1[k], g0[k], 0x07060302u);
                rb[2 * k] = __builtin_amdgcn_perm(g3[k], g2[k], 0x05040100u);
                rb[2 * k + 1] = __builtin_amdgcn_perm(g3[k], g2[k], 0x07060302u);
            }
            transpose16x16_pairs(ra);
            transpose16x16_pairs(rb);
            // row b: 512 bytes per block, this lane's 8 bytes = groups 0..3 (2 bytes each)
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                uint2 w;
                w.x = ra[b];
                w.y = rb[b];
                *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(dst) + b * 512 + lane * 8) = w;
            }
        } else {
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(src) + b * 512 + lane * 8);
                ra[b] = w.x;
                rb[b] = w.y;
            }
            transpose16x16_pairs(ra);                            // (its own inverse)
            transpose16x16_pairs(rb);
            v4u v[8];
            uint32_t g0[8], g1[8], g2[8], g3[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // ra[2k] = elem 2k of g0 | elem 2k of g1 << 16 ; ra[2k+1] likewise -> dword k of g0 = elem 2k | elem 2k+1 << 16
                g0[k] = __builtin_amdgcn_perm(ra[2 * k + 1], ra[2 * k], 0x05040100u);
                g1[k] = __builtin_amdgcn_perm(ra[2 * k + 1], ra[2 * k], 0x07060302u);
                g2[k] = __builtin_amdgcn_perm(rb[2 * k + 1], rb[2 * k], 0x05040100u);
                g3[k] = __builtin_amdgcn_perm(rb[2 * k + 1], rb[2 * k], 0x07060302u);
            }
            v[0] = v4u{g0[0], g0[1], g0[2], g0[3]}; v[1] = v4u{g0[4], g0[5], g0[6], g0[7]};
            v[2] = v4u{g1[0], g1[1], g1[2], g1[3]}; v[3] = v4u{g1[4], g1[5], g1[6], g1[7]};
            v[4] = v4u{g2[0], g2[1], g2[2], g2[3]}; v[5] = v4u{g2[4], g2[5], g2[6], g2[7]};
            v[6] = v4u{g3[0], g3[1], g3[2], g3[3]}; v[7] = v4u{g3[4], g3[5], g3[6], g3[7]};
            v4u* p = reinterpret_cast<v4u*>(dst) + lane * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = v[j];
        }
    }
}

// any element size (1, 2), any block size (multiple of 8), partial last block, copied tail: one thread per byte of the
// shuffled side (it collects / scatters the bit of its 8 elements)
__global__ __launch_bounds__(256)
void bitshuffle_generic_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t first_elem, uint64_t n_elems,
                               uint32_t elem_size, uint64_t bs, bool decode)
{
    // elements [first_elem, n_elems): whole blocks of bs, then one block of the remainder rounded down to 8, then the copied tail
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;            // byte index inside [first_elem * E, n_elems * E)
    const uint64_t total_bytes = (n_elems - first_elem) * elem_size;
    if (t >= total_bytes) return;
    const uint64_t rel_elems = n_elems - first_elem;
    const uint64_t full = rel_elems / bs;
    const uint64_t last = (rel_elems - full * bs) / 8 * 8;
    const uint64_t shuffled_bytes = (full * bs + last) * elem_size;
    const uint8_t* ib = in + first_elem * elem_size;
    uint8_t* ob = out + first_elem * elem_size;
    if (t >= shuffled_bytes) { ob[t] = ib[t]; return; }                            // tail: up to 7 elements, verbatim
    const uint64_t blk = t / (bs * elem_size);
    const uint64_t cnt = blk < full ? bs : last;                                   // elements of this block
    const uint64_t base = blk * bs * elem_size;                                    // byte offset of the block
    const uint64_t o = t - base;                                                   // byte inside the block, shuffled side
    const uint64_t row = o / (cnt / 8), col = o % (cnt / 8);                       // bit row, byte in the row
    const uint32_t byte_of = (uint32_t)(row >> 3), bit = (uint32_t)(row & 7);
    if (!decode) {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) v |= ((uint32_t)(ib[base + (col * 8 + j) * elem_size + byte_of] >> bit) & 1u) << j;
        ob[t] = (uint8_t)v;
    } else {
        // decode: thread t owns plain byte t of the block: element e = o / E, byte b = o % E; its 8 bits come from 8 rows
        const uint64_t e = o / elem_size;
        const uint32_t b = (uint32_t)(o % elem_size);
        uint32_t v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) v |= ((uint32_t)(ib[base + (uint64_t)(b * 8 + k) * (cnt / 8) + e / 8] >> (e & 7)) & 1u) << k;
        ob[t] = (uint8_t)v;
    }
}

// ------------------------------------------------------------------------------------------------
// decode (src/sqeazy.cpp:281-335 -> dynamic_pipeline.hpp:740-846): LZ4 frames, inverse filters.
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// Frame index.  blk[i] = {offset of block i's data in the stream (lo, hi), size | raw flag << 31, block id in its frame},
// frame_first[f] = index of the first block of frame f, counts[0..3] = #frames, #blocks, error code, #compressed blocks.
//
// The frames of the chunked layout form a linked list through their size fields -- walking it is a pointer chase
// through cold memory (one lane, about 0.75 us per frame: 3 ms for the 4096 frames of a 1 GiB stack).  The fast path
// ranks the list in parallel instead:
//   1. lz4_frame_candidates_kernel: every position of the stream that holds the frame magic and a FLG byte sqeazy writes
//      is a candidate {pos, first block size field, FLG, "the four bytes in front are zero"}, appended to a list and
//      entered into an open-addressing table pos -> list index (one pass at HBM speed);
//   2. lz4_frame_rank_kernel (one workgroup): successor of a candidate = the candidate that starts right behind its
//      block's end mark; pointer doubling (log2 N rounds) gives every candidate its distance to the end of the stream
//      and marks the candidates reachable from position 0 -- those are the frames, rank = distance(head) - distance.
// Anything the fast path does not cover (multi-block frames = the serial block-linked layout, empty frames, more
// candidates than the table holds, a chain that does not end exactly at the end of the stream) sets code 100 and the
// host runs the serial walk (lz4_frame_index_kernel), which also produces the exact error codes for corrupt streams.
// ------------------------------------------------------------------------------------------------
struct FrameCand { unsigned long long pos; uint32_t field; uint32_t flags; };     // flags: bit 0 = 4 zero bytes in front, bits 8..15 = FLG
struct FrameSlot { unsigned long long key; uint32_t idx; uint32_t pad; };           // key = pos + 1, 0 = empty

__device__ __forceinline__ uint32_t cand_slot(uint64_t pos, uint32_t mask)
{
    return (uint32_t)(((pos + 1) * 0x9E3779B97F4A7C15ull) >> 40) & mask;
}

// A look at the first frame in front of everything (round 4): when a second block follows its first block, the stream is the serial
// layout's ONE block-linked frame (or chunks of several blocks) -- nothing the ranking covers, it would scan the whole payload
// (0.2-0.3 ms per GiB) only to give up.  *hint = 1 makes the two kernels below return at once; the host then takes the walk, as it does
// whenever the ranking gives up.
__global__ void lz4_frame_probe_kernel(const uint8_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ hint)
{
    uint32_t multi = 0;
    if (n >= 19 && ld_u32(in) == 0x184D2204u) {
        const uint32_t flg = in[4];
        const uint32_t field = ld_u32(in + 7);
        if ((flg >> 6) == 1 && !(flg & 0x0D) && field != 0) {
            const uint64_t next = 11ull + (field & 0x7fffffffu) + (((flg >> 4) & 1u) ? 4u : 0u);
            if (next + 4 <= n && ld_u32(in + next) != 0u) multi = 1;
        }
    }
    *hint = multi;
}

// a window that holds the magic's first two bytes somewhere: the sixteen start positions one by one (a call: it is rare)
__device__ __noinline__ void frame_cand_record(uint64_t n, FrameSlot* __restrict__ table, uint32_t mask, FrameCand* __restrict__ list, uint32_t cap,
                                               uint32_t* __restrict__ ncand, uint64_t base, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3,
                                               uint32_t w4, uint32_t w5, uint32_t w6, uint32_t w7, uint32_t w8, uint32_t kmax)
{
    const uint32_t w[9] = {w0, w1, w2, w3, w4, w5, w6, w7, w8};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if ((uint32_t)k >= kmax) continue;                             // (only the window in front of the first aligned vector stops early)
        // little-endian window of 12 bytes starting at base + k: dwords d0 (bytes 0..3), d1 (4..7), d2 (8..11)
        const int i = 1 + (k >> 2), sh = k & 3;
        const uint32_t d0 = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
        if (d0 != 0x184D2204u) continue;
        const uint32_t d1 = __builtin_amdgcn_alignbyte(w[i + 2], w[i + 1], sh);
        const uint32_t d2 = __builtin_amdgcn_alignbyte(w[i + 3], w[i + 2], sh);
        const uint32_t flg = d1 & 0xffu;
        const uint64_t pos = base + (uint64_t)k;
        if ((flg >> 6) != 1 || (flg & 0x0D) || pos + 11 > n) continue;
        const uint32_t pz = __builtin_amdgcn_alignbyte(w[i], w[i - 1], sh);       // the four bytes in front
        const uint32_t idx = atomicAdd(ncand, 1u);
        if (idx >= cap) continue;                                        // the rank kernel sees ncand > cap and gives up
        FrameCand c;
        c.pos = pos; c.field = (d1 >> 24) | (d2 << 8); c.flags = ((pos >= 4 && pz == 0u) ? 1u : 0u) | (flg << 8);
        list[idx] = c;
        uint32_t s_ = cand_slot(pos, mask);
        for (uint32_t probe = 0; probe <= mask; ++probe, s_ = (s_ + 1) & mask) {
            if (atomicCAS(&table[s_].key, 0ull, (unsigned long long)(pos + 1)) == 0ull) { table[s_].idx = idx; break; }
        }
    }
}

// The stored tail (round 6).  Bit planes of noise end up as stored frames: 99.5 % of the bench stack's blob, all of them the same size,
// and the scan above reads every byte of them to find nothing.  The decoder knows the chunk size, so it knows where the LAST frames would
// start if they were stored ones: n - (15 + last), and from there every 15 + chunk bytes towards the front.  One workgroup looks at those
// places -- magic, a FLG the encoder writes (no block checksum), the size field of a stored block of exactly that size, the end mark
// behind it, the four zero bytes in front -- and takes the longest run of them that reaches the end of the stream: those m places go into
// the candidate list as the scan would have put them, *scan_end = the first of them, and the scan stops there.  Nothing is taken on
// trust: the ranking still has to reach the first of these from position 0 and the last one's end mark has to sit on the stream's end; a
// stream built to fool this (stored payloads that hold such headers at those very places) makes the ranking give up, and the host scans
// again with last = 0 (no tail), then walks.
__global__ __launch_bounds__(1024)
void lz4_frame_tail_kernel(const uint8_t* __restrict__ in, uint64_t n, uint64_t chunk, uint64_t last, uint32_t kmax, FrameSlot* __restrict__ table,
                           uint32_t mask, FrameCand* __restrict__ list, uint32_t cap, uint32_t* __restrict__ ncand,
                           const uint32_t* __restrict__ hint, uint32_t* __restrict__ tail_frames, unsigned long long* __restrict__ scan_end)
{
    __shared__ uint32_t first_bad;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { first_bad = kmax + 1u; *scan_end = n; *tail_frames = 0u; }
    __syncthreads();
    if (*hint || last == 0 || last > chunk || chunk >= 0x7fffffffull || n < last + 15) return;
    // place k = 1 .. kmax from the end: start(k) = n - (15 + last) - (k - 1) * (15 + chunk)
    auto start_of = [&](uint32_t k) -> int64_t { return (int64_t)n - (int64_t)(15 + last) - (int64_t)(k - 1u) * (int64_t)(15 + chunk); };
    // (every load of a thread's four places in flight at once, at clamped addresses: a place costs one trip to memory, not five)
    for (uint32_t k0 = 1u; k0 <= kmax && k0 < first_bad; k0 += 4096u) {
        uint32_t pz[4], mg[4], w1[4], w2[4], em[4];
        bool in_range[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t k = k0 + tid + 1024u * (uint32_t)j;
            const int64_t p = start_of(k);
            in_range[j] = k <= kmax && p >= 0 && p != 1 && p != 2 && p != 3;
            const uint64_t q = in_range[j] ? (uint64_t)p : (uint64_t)start_of(1u);
            const uint64_t raw = (in_range[j] && k != 1u) ? chunk : last;
            const uint8_t* f = in + q;
            pz[j] = q >= 4 ? ld_u32(f - 4) : 0u;
            mg[j] = ld_u32(f); w1[j] = ld_u32(f + 4); w2[j] = ld_u32(f + 8);
            em[j] = ld_u32(f + 11 + raw);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t k = k0 + tid + 1024u * (uint32_t)j;
            if (k > kmax) continue;
            const uint32_t raw = (uint32_t)(k == 1u ? last : chunk);
            const uint32_t flg = w1[j] & 0xffu, field = (w1[j] >> 24) | (w2[j] << 8);
            // magic, a FLG the encoder writes without block checksum (as frame_cand_record), a stored block of that size, the end mark
            // behind it (the last one's ends the stream), what the frame in front ends with
            const bool ok = in_range[j] && mg[j] == 0x184D2204u && (flg >> 6) == 1u && !(flg & 0x1Du) && field == (0x80000000u | raw) && em[j] == 0u && pz[j] == 0u;
            if (!ok) atomicMin(&first_bad, k);
        }
        __syncthreads();
    }
    __syncthreads();
    const uint32_t m = first_bad - 1u;
    if (m == 0u) return;
    for (uint32_t k = 1u + tid; k <= m; k += 1024u) {
        const uint64_t pos = (uint64_t)start_of(k);
        const uint32_t idx = atomicAdd(ncand, 1u);
        if (idx >= cap) continue;                                                     // (the rank kernel sees ncand > cap and gives up)
        FrameCand c;
        c.pos = pos; c.field = 0x80000000u | (uint32_t)(k == 1u ? last : chunk); c.flags = (pos >= 4 ? 1u : 0u) | ((uint32_t)in[pos + 4] << 8);
        list[idx] = c;
        uint32_t s_ = cand_slot(pos, mask);
        for (uint32_t probe = 0; probe <= mask; ++probe, s_ = (s_ + 1) & mask) {
            if (atomicCAS(&table[s_].key, 0ull, (unsigned long long)(pos + 1)) == 0ull) { table[s_].idx = idx; break; }
        }
    }
    if (tid == 0) { *scan_end = (unsigned long long)start_of(m); *tail_frames = m; }
}

__global__ __launch_bounds__(256)
void lz4_frame_candidates_kernel(const uint8_t* __restrict__ in, uint64_t n_stream, FrameSlot* __restrict__ table, uint32_t mask,
                                 FrameCand* __restrict__ list, uint32_t cap, uint32_t* __restrict__ ncand, const uint32_t* __restrict__ hint,
                                 const unsigned long long* __restrict__ scan_end)
{
    if (*hint) return;
    // (the stored tail, above: its frames are on the list already; a frame that starts in front of it ends in front of it)
    const uint64_t n = *scan_end < n_stream ? (uint64_t)*scan_end : n_stream;
    // a thread inspects 16 start positions [base, base + 16); it needs the bytes [base - 4, base + 28).  Four such windows per thread and
    // step, all their loads issued before the first is looked at
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    auto inspect = [&](uint64_t base, const uint32_t (&w)[9], uint32_t kmax) {
        // the magic's first TWO bytes (04 22) at one of the 16 positions?  The reject has to hold for whole wavefronts: a lone 04 sits in
        // one window of sixteen, i.e. in nearly every wavefront's 64 windows -- a test for that byte alone sends every wave through the
        // sixteen compares below.  Byte flags by the has-zero trick (never misses a zero byte, may flag a byte above one: fine for a
        // reject).  (Round 3 measured: the scan stays at 2.6 TB/s either way -- its 16-byte loads started 4 bytes in front of a window and
        // hit two sectors each; round 4 loads aligned vectors, below.)
        uint32_t pair = 0;
#pragma unroll
        for (int i = 1; i <= 4; ++i) {
            const uint32_t x = w[i] ^ 0x04040404u;
            const uint32_t y = __builtin_amdgcn_alignbyte(w[i + 1], w[i], 1) ^ 0x22222222u;      // the byte behind each byte
            pair |= ((x - 0x01010101u) & ~x) & ((y - 0x01010101u) & ~y);
        }
        if (!(pair & 0x80808080u)) return;
        // (rare, and out of line: unrolled in place, once for each of a step's windows, the sixteen compares cost the kernel 145 registers)
        frame_cand_record(n, table, mask, list, cap, ncand, base, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], kmax);
    };
    // Round 4: aligned 16-byte vectors; what a window needs from outside its vector comes out of registers, not memory.  Round 5: a
    // wavefront takes 4 KiB, every lane the 64 bytes [64 * lane, 64 * lane + 64) of it as four vectors (the access pattern of
    // frame_block_sums_kernel, which reads at 5.6 TB/s where the grid-stride loop of single vectors stayed at 3): three of a lane's four
    // windows find their neighbours in the lane's own registers, the first takes one dword from the lane before and the last three from
    // the lane behind (ds_bpermute), a wave's first and last lane fetch those themselves -- with the wave's own loads, all in flight at once.
    // Start positions [0, A) in front of the first aligned vector and the ragged end go through the byte-wise window.
    const uint64_t A = (16u - (reinterpret_cast<uintptr_t>(in) & 15u)) & 15u;
    const uint64_t nal = n > A + 32 ? (n - A - 16) / 16 : 0;            // aligned vectors whose window [base - 4, base + 28) lies inside the stream (base >= 4 checked below)
    auto slow_window = [&](uint64_t base, uint32_t kmax) {
        uint32_t w[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t p = (int64_t)base - 4 + 4 * i + k;
                const uint32_t byte = (p >= 0 && (uint64_t)p < n) ? in[p] : 0xFFu;
                v |= byte << (8 * k);
            }
            w[i] = v;
        }
        inspect(base, w, kmax);
    };
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t gthread = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    {
        // a wavefront's pieces: 256 vectors in a row each, the grid's wavefronts take them in turn, the NEXT piece's loads are in flight
        // while this one is looked at (a wavefront that loads 64 bytes per lane once and then computes keeps the memory system busy for
        // half its life: 3 TB/s; what bounds a read-only scan is the bytes in flight)
        // (every load unconditional, at a clamped address: a load under a condition with a default value behind it makes the compiler
        // wait for it on the spot)
        struct Piece { uint4 a[4]; uint4 edge; };
        const uint64_t npieces = (nal + 255) / 256, nwaves = (uint64_t)gridDim.x * 4;
        const uint4* __restrict__ vec = reinterpret_cast<const uint4*>(in + A);
        auto issue = [&](uint64_t pc, Piece& q) {
            const uint64_t t0 = pc * 256 + (uint64_t)lane * 4;           // this lane's first vector
#pragma unroll
            for (int j = 0; j < 4; ++j) q.a[j] = vec[t0 + j < nal ? t0 + j : nal];       // (vector nal exists too: its first dwords are the tail of the window before)
            // the vector behind the wave's last (lane 63) and the one in front of its first (lane 0: its last dword)
            const uint64_t te = lane == 63u ? (t0 + 4 < nal ? t0 + 4 : nal) : (t0 ? t0 - 1 : 0);
            if (lane == 63u || lane == 0u) q.edge = vec[te];
        };
        auto look = [&](uint64_t pc, const Piece& q) {
            const uint64_t t0 = pc * 256 + (uint64_t)lane * 4;
            const int up = (int)(((lane + 1u) & 63u) * 4u), dn = (int)(((lane + 63u) & 63u) * 4u);
            uint32_t nx0 = (uint32_t)__builtin_amdgcn_ds_bpermute(up, (int)q.a[0].x);
            uint32_t nx1 = (uint32_t)__builtin_amdgcn_ds_bpermute(up, (int)q.a[0].y);
            uint32_t nx2 = (uint32_t)__builtin_amdgcn_ds_bpermute(up, (int)q.a[0].z);
            uint32_t pv = (uint32_t)__builtin_amdgcn_ds_bpermute(dn, (int)q.a[3].w);
            if (lane == 63u) { nx0 = q.edge.x; nx1 = q.edge.y; nx2 = q.edge.z; }
            if (lane == 0u) pv = t0 ? q.edge.w : (A >= 4 ? ld_u32(in + A - 4) : 0xffffffffu);   // (A < 4: the window goes the byte-wise way below)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (t0 + j >= nal) continue;
                const uint64_t base = A + (t0 + j) * 16;
                if (base < 4) { slow_window(base, 16u); continue; }       // (a payload that starts within 4 bytes of an aligned address)
                const uint32_t w[9] = {j ? q.a[(j + 3) & 3].w : pv, q.a[j].x, q.a[j].y, q.a[j].z, q.a[j].w,
                                       j < 3 ? q.a[(j + 1) & 3].x : nx0, j < 3 ? q.a[(j + 1) & 3].y : nx1, j < 3 ? q.a[(j + 1) & 3].z : nx2, 0u};
                inspect(base, w, 16u);
            }
        };
        uint64_t pc = (gthread - lane) >> 6;                              // (whole waves stay together: bpermute)
        if (pc < npieces) {
            Piece p0, p1;
            const uint64_t last = npieces - 1;
            issue(pc, p0);
            for (;;) {
                const uint64_t pn = pc + nwaves;
                issue(pn < last ? pn : last, p1);                         // (behind the end: the last piece once more, not looked at)
                look(pc, p0);
                if (pn >= npieces) break;
                const uint64_t pm = pn + nwaves;
                issue(pm < last ? pm : last, p0);
                look(pn, p1);
                if (pm >= npieces) break;
                pc = pm;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && A) slow_window(0, (uint32_t)(A < n ? A : n));          // start positions [0, A)
    // the ragged end: start positions from the first vector not taken above
    for (uint64_t base = A + nal * 16 + gthread * 16; base < n; base += stride * 16) slow_window(base, 16u);
}

// one workgroup of 1024 threads; work arrays succ[2][N+2], dist[2][N+2], mark[N+2]: in LDS as 16-bit indices when the candidates fit
// the dynamic allocation the host made (lds_entries; twelve rounds of pointer doubling with a workgroup barrier each: 0.16 ms through
// global memory, a fraction of that in LDS), else in global memory (`work`)
template <typename IT>
__device__ __forceinline__ void frame_rank_body(const uint8_t* __restrict__ in, uint64_t n, const FrameSlot* __restrict__ table, uint32_t mask,
                                                const FrameCand* __restrict__ list, uint32_t N, IT* succ0, IT* succ1, IT* dist0, IT* dist1,
                                                uint8_t* mark, uint4* __restrict__ blk, uint32_t* __restrict__ frame_first,
                                                uint64_t max_blocks, uint32_t* __restrict__ counts, uint32_t* head_s)
{
    const uint32_t tid = threadIdx.x;
    auto give_up = [&]() { if (tid == 0) { counts[0] = 0; counts[1] = 0; counts[2] = 100; } };
    const uint32_t E = N, X = N + 1, M = N + 2;                          // sentinels: end of stream, no successor
    IT* succ[2] = {succ0, succ1};
    IT* dist[2] = {dist0, dist1};
    auto lookup = [&](uint64_t pos) -> uint32_t {
        uint32_t s_ = cand_slot(pos, mask);
        for (uint32_t probe = 0; probe <= mask; ++probe, s_ = (s_ + 1) & mask) {
            const unsigned long long k = table[s_].key;
            if (k == 0ull) return X;
            if (k == (unsigned long long)(pos + 1)) return table[s_].idx;
        }
        return X;
    };
    if (tid == 0) *head_s = lookup(0);
    for (uint32_t i = tid; i < M; i += 1024) {
        uint32_t sc = X;
        if (i < N) {
            const FrameCand c = list[i];
            const uint64_t sz = c.field & 0x7fffffffu;
            const uint64_t e = c.pos + 11 + sz + (((c.flags >> 12) & 1u) ? 4 : 0);   // where the end mark must sit (FLG bit 4: block checksum)
            if (c.field != 0 && e + 4 <= n) {
                if (e + 4 == n) sc = (ld_u32(in + e) == 0u) ? E : X;
                else { const uint32_t j = lookup(e + 4); if (j < N && (list[j].flags & 1u)) sc = j; }
            }
        } else if (i == E) sc = E;
        succ[0][i] = (IT)sc;
        dist[0][i] = (IT)(i < N ? 1u : 0u);
        mark[i] = 0;
    }
    if (sizeof(IT) == 4) __threadfence();                              // (work arrays in global memory; the 16-bit ones live in LDS)
    __syncthreads();
    const uint32_t head = *head_s;
    if (head >= N) { give_up(); return; }
    if (tid == 0) mark[head] = 1;
    if (sizeof(IT) == 4) __threadfence();                              // (work arrays in global memory; the 16-bit ones live in LDS)
    __syncthreads();
    int cur = 0;
    for (uint32_t span = 1; span < M; span <<= 1) {
        // reachable set doubles: everything marked marks its (2^b-th) successor, then the pointers jump
        for (uint32_t i = tid; i < N; i += 1024) if (mark[i]) { const uint32_t sc = succ[cur][i]; if (sc < N) mark[sc] = 1; }
        for (uint32_t i = tid; i < M; i += 1024) {
            const uint32_t sc = succ[cur][i];
            dist[cur ^ 1][i] = (IT)((uint32_t)dist[cur][i] + (uint32_t)dist[cur][sc]);
            succ[cur ^ 1][i] = succ[cur][sc];
        }
        if (sizeof(IT) == 4) __threadfence();
        __syncthreads();
        cur ^= 1;
    }
    // every pointer now rests on a sentinel; the head's chain must end at the end of the stream
    if ((uint32_t)succ[cur][head] != E) { give_up(); return; }
    const uint32_t L = dist[cur][head];                                  // number of frames
    if (L > max_blocks) { give_up(); return; }
    for (uint32_t i = tid; i < N; i += 1024) {
        if (!mark[i]) continue;
        const uint32_t r = L - (uint32_t)dist[cur][i];
        const FrameCand c = list[i];
        const uint64_t off = c.pos + 11;
        blk[r] = make_uint4((uint32_t)off, (uint32_t)(off >> 32), c.field, 0u);
        frame_first[r] = r;
        if (!(c.field >> 31)) atomicAdd(&counts[3], 1u);                  // (compressed blocks: picks the decode kernel's ring)
    }
    if (tid == 0) { frame_first[L] = L; counts[0] = L; counts[1] = L; counts[2] = 0; }
}

__global__ __launch_bounds__(1024)
void lz4_frame_rank_kernel(const uint8_t* __restrict__ in, uint64_t n, const FrameSlot* __restrict__ table, uint32_t mask,
                           const FrameCand* __restrict__ list, uint32_t cap, const uint32_t* __restrict__ ncand,
                           uint32_t* __restrict__ work, uint4* __restrict__ blk, uint32_t* __restrict__ frame_first,
                           uint64_t max_blocks, uint32_t* __restrict__ counts, uint32_t lds_entries, const uint32_t* __restrict__ hint)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t rank_lds[];
    __shared__ uint32_t head_s;
    if (*hint) { if (threadIdx.x == 0) { counts[0] = 0; counts[1] = 0; counts[2] = 100; } return; }     // (lz4_frame_probe_kernel: frames of several blocks)
    const uint32_t N = *ncand;
    if (N == 0 || N > cap) { if (threadIdx.x == 0) { counts[0] = 0; counts[1] = 0; counts[2] = 100; } return; }
    const uint32_t M = N + 2;
    if (M <= lds_entries && M <= 65535u) {
        uint16_t* w16 = reinterpret_cast<uint16_t*>(rank_lds);
        frame_rank_body<uint16_t>(in, n, table, mask, list, N, w16, w16 + lds_entries, w16 + 2 * lds_entries, w16 + 3 * lds_entries,
                                  rank_lds + 8u * lds_entries, blk, frame_first, max_blocks, counts, &head_s);
    } else {
        frame_rank_body<uint32_t>(in, n, table, mask, list, N, work, work + M, work + 2 * M, work + 3 * M,
                                  reinterpret_cast<uint8_t*>(work + 4 * M), blk, frame_first, max_blocks, counts, &head_s);
    }
}

// Serial walk (the size fields are chased one after the other): any layout, exact error codes.  The wavefront's lanes all run the
// same walk (uniform values, lane 0 stores); they part only behind a STORED block, where lane j looks at the place the (j+1)-th next
// size field would be if the blocks that follow were stored blocks of the same size -- as in the noise planes of a block-linked frame,
// runs of hundreds of them: the leading lanes that find exactly that field are that many blocks, taken in one step (round 4).
__global__ __launch_bounds__(64)
void lz4_frame_index_kernel(const uint8_t* __restrict__ in, uint64_t n, uint4* __restrict__ blk, uint32_t* __restrict__ frame_first,
                            uint64_t max_blocks, uint32_t* __restrict__ counts)
{
    const uint32_t lane = threadIdx.x;
    // every step fetches 16 bytes at once: behind a block's data sit the end mark (or the next block's size) and --
    // after an end mark -- the next frame's 7 header bytes and its first block size: one dependent load per
    // single-block frame.
    uint64_t off = 0;
    uint32_t nframes = 0, nblocks = 0, err = 0, ncomp = 0;
    auto load16 = [&](uint64_t o, uint32_t w[4]) {
        // bytes past the end of the stream read as 0xFF (never a valid header, size fields fail the bounds checks)
        if (o + 16 <= n) { const uint4 v = ld_u128(in + o); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; return; }
        uint8_t b[16];
        for (int i = 0; i < 16; ++i) b[i] = (o + i < n) ? in[o + i] : 0xFF;
        for (int i = 0; i < 4; ++i) w[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    };
    uint32_t w[4];
    bool have = false;                       // w[0..2] hold the 12 bytes at `off`
    while (off < n && !err) {
        if (!have) load16(off, w);
        have = false;
        // frame header: magic, FLG, BD, HC, then the first block size (bytes 7..10)
        if (off + 7 > n || w[0] != 0x184D2204u) { err = 1; break; }
        const uint32_t flg = w[1] & 0xffu;
        if ((flg >> 6) != 1 || (flg & 0x0D)) { err = 2; break; }       // only what sqeazy writes: no content size / checksum / dictID
        const bool block_checksum = (flg >> 4) & 1;
        // frame_first has max_blocks + 2 entries; a stream of block-less frames must not run past it
        if (nframes >= max_blocks) { err = 5; break; }
        if (lane == 0) frame_first[nframes] = nblocks;
        uint32_t field = (w[1] >> 24) | (w[2] << 8);                    // bytes 7..10
        off += 7;
        uint32_t j = 0;
        for (;;) {
            if (off + 4 > n) { err = 3; break; }
            off += 4;
            if (field == 0) break;                                      // end mark (a frame without blocks)
            const uint32_t sz = field & 0x7fffffffu;
            if (off + sz > n || nblocks >= max_blocks) { err = 4; break; }
            if (lane == 0) blk[nblocks] = make_uint4((uint32_t)off, (uint32_t)(off >> 32), field, j);
            ++nblocks; ++j;
            ncomp += !(field >> 31);
            off += sz + (block_checksum ? 4 : 0);
            if ((field >> 31) && sz) {
                // a run of stored blocks of this size?  lane j: the field at off + j * (4 + sz [+ 4]) must be this very field, its block inside the stream
                const uint64_t step = 4ull + sz + (block_checksum ? 4 : 0);
                const uint64_t pj = off + (uint64_t)lane * step;
                const bool ok = pj + step <= n && ld_u32(in + pj) == field;
                const uint64_t okm = ballot(ok);
                uint32_t run = okm == ~0ull ? 64u : ctz64(~okm);
                if ((uint64_t)run > max_blocks - nblocks) run = (uint32_t)(max_blocks - nblocks);
                if (lane < run) { const uint64_t o = pj + 4; blk[nblocks + lane] = make_uint4((uint32_t)o, (uint32_t)(o >> 32), field, j + lane); }
                nblocks += run; j += run;
                off += (uint64_t)run * step;
            }
            load16(off, w);                                             // next size field or end mark, and 12 bytes behind it
            field = w[0];
            if (field == 0 && off + 4 <= n) {                           // end mark: the 12 bytes behind it open the next frame
                off += 4;
                w[0] = w[1]; w[1] = w[2]; w[2] = w[3];
                have = true;
                break;
            }
        }
        if (err) break;
        ++nframes;
    }
    if (lane == 0) {
        frame_first[nframes] = nblocks;
        counts[0] = nframes; counts[1] = nblocks; counts[2] = err; counts[3] = ncomp;
    }
}

// One wavefront per frame; the last 64 KiB of decoded output live in an LDS ring so that match copies (which may
// overlap their own output and, in block-linked frames, reach into the previous block) never read global memory
// the wave has just written.  Output leaves through the ring in 16-byte pieces.  The compressed bytes are parsed out of
// a 3-4 KiB LDS stage refilled 64 x 16 B at a time (a token/length/offset byte costs an LDS broadcast read instead of a
// dependent global load); stored blocks of single-block frames are copied straight from the stream to the output.
// DEC_RING = 64 KiB: every match source is in the ring (68 KiB of LDS per frame wave: 2 waves per CU) -- right when few frames
// are compressed and each is a long chain of short sequences (latency-bound).  DEC_RING = 16 KiB (20 KiB: 8 waves per CU): for
// streams with thousands of compressed frames (throughput-bound); a match that reaches further back than the ring reads its
// source from the output buffer, where those bytes have long been flushed (one global round trip for such a match).

// Round 5: `remap` != nullptr -- the stage in front of `lz4` on the encoder's side was frame_shuffle, and every LZ4 chunk lies inside ONE of
// its frames of remap_bytes: the frames are decoded straight to where the shuffle's inverse would move them (frame i of the sorted
// stream is frame remap[i] of the volume, frame_shuffle_utils.hpp:337-344), one pass over the decoded bytes less.
// (a map that names a place several times -- frames of equal metric on the encoder's side -- reaches the device with all but the LAST frame
// named for a place struck (~0, sqy_capi.cpp frame_shuffle_prepare): those frames are not decoded at all, what they would write is
// overwritten by definition (the reference's loop: the last one stays), and no two frames ever decode into one place at once)
__device__ __forceinline__ bool lz4_decode_frame_struck(uint64_t o, const uint64_t* __restrict__ remap, uint64_t remap_bytes)
{
    return remap && remap[o / remap_bytes] == ~0ull;
}
__device__ __forceinline__ uint64_t lz4_decode_frame_out(uint64_t o, const uint64_t* __restrict__ remap, uint64_t remap_bytes)
{
    if (!remap) return o;
    const uint64_t fi = o / remap_bytes;
    return remap[fi] * remap_bytes + (o - fi * remap_bytes);
}

// The decode kernels' rings leave for global memory in pieces of DEC_FLUSH_PIECE bytes (flush()), a copy step adds at most DEC_STEP_MAX:
// up to DEC_FLUSH_PIECE - 1 + DEC_STEP_MAX decoded bytes are in the ring only.  A match that reaches behind the ring reads them back
// from global memory; what it reads must have been STORED (offset > DEC_RING covers that: the static_asserts) and the stores must have
// LANDED: the wait at the match's start covers everything older, and a step whose source may lie within the bytes this very match has
// flushed since -- j + cnt + DEC_FLUSH_PIECE + DEC_STEP_MAX > offset -- waits again (round-5 advice: the guard still said 2048 from the
// time when pieces were 1 KiB).
constexpr uint32_t DEC_FLUSH_PIECE = 4096u, DEC_STEP_MAX = 1024u;
template <uint32_t DEC_RING>
__global__ __launch_bounds__(64)
void lz4_frames_decode_kernel(const uint8_t* __restrict__ in, const uint4* __restrict__ blk, const uint32_t* __restrict__ frame_first,
                              uint8_t* __restrict__ out, uint64_t out_bytes, uint64_t frame_stride, uint64_t block_bytes,
                              uint32_t* __restrict__ errflag, const uint64_t* __restrict__ remap, uint64_t remap_bytes)
{
    // stage of compressed bytes: 4 KiB, or 3 KiB beside the small ring (ring + stage + marks < 20 KiB: 8 waves per CU, not 7)
    constexpr uint32_t DEC_IN = DEC_RING < 65536u ? 3072u : 4096u;
    static_assert(DEC_RING >= DEC_FLUSH_PIECE + 2u * DEC_STEP_MAX + 1024u, "a match behind the ring must find its source flushed: ring >= flush piece + two copy steps + margin");
    __shared__ __attribute__((aligned(16))) uint8_t dring_raw[DEC_RING + DEC_IN + 64];   // static: 68 KiB (dynamic LDS stops at 64 KiB by default)
    lds_u8* ring = (lds_u8*)dring_raw;
    lds_u8* stage = (lds_u8*)dring_raw + DEC_RING;
    lds_u8* owner_mark = (lds_u8*)dring_raw + DEC_RING + DEC_IN;                   // 64 bytes: which sequence starts at an output byte
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x;
    const uint32_t b0 = frame_first[f], b1 = frame_first[f + 1];
    if (lz4_decode_frame_struck((uint64_t)f * frame_stride, remap, remap_bytes)) return;
    const uint64_t frame_out = lz4_decode_frame_out((uint64_t)f * frame_stride, remap, remap_bytes);   // frame f decodes to [f*chunk, ...)
    uint32_t pos = 0;                                          // decoded bytes of this frame so far
    uint32_t flushed = 0;                                      // bytes of this frame already written to global memory
    bool bad = false;

    auto flush = [&](bool all) {
        // write ring bytes [flushed, pos) (all) or whole pieces of 4 KiB of it (round 5: four reads of the ring in flight, then four stores;
        // before, a KiB per call, each behind its own LDS round trip.  What has not left yet stays inside the ring -- a step adds at most
        // 1 KiB, the smallest ring has 8 --, and a match that reaches behind the ring ends more than 7 KiB back: flushed)
        while (flushed + DEC_FLUSH_PIECE <= pos) {
            const uint64_t o = frame_out + flushed;
            if (o + DEC_FLUSH_PIECE > out_bytes) { bad = true; flushed = pos; break; }
            v4u v[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const SQY_LDS v4u*>(ring + ((flushed + q * 1024u + (uint32_t)lane * 16u) & (DEC_RING - 1)));
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) st_u128(out + o + q * 1024u + (uint32_t)lane * 16u, make_uint4(v[q].x, v[q].y, v[q].z, v[q].w));
            flushed += DEC_FLUSH_PIECE;
        }
        while (all && flushed < pos) {
            const uint32_t cnt = (pos - flushed >= 1024u) ? 1024u : (pos - flushed);
            const uint64_t o = frame_out + flushed;
            if (o + cnt > out_bytes) { bad = true; flushed = pos; break; }
            const uint32_t nvec = cnt >> 4;
            if ((uint32_t)lane < nvec) {
                const v4u v = *reinterpret_cast<const SQY_LDS v4u*>(ring + ((flushed + (uint32_t)lane * 16u) & (DEC_RING - 1)));
                st_u128(out + o + (uint32_t)lane * 16u, make_uint4(v.x, v.y, v.z, v.w));
            }
            const uint32_t done = nvec << 4;
            if ((uint32_t)lane < cnt - done) out[o + done + lane] = ring[(flushed + done + lane) & (DEC_RING - 1)];
            flushed += cnt;
        }
    };

    for (uint32_t b = b0; b < b1 && !bad; ++b) {
        const uint4 e = blk[b];
        const uint8_t* __restrict__ src = in + (((uint64_t)e.y << 32) | e.x);
        const uint32_t sz = e.z & 0x7fffffffu;
        if (e.z >> 31) {
            if (pos + sz > block_bytes * (b - b0 + 1)) { bad = true; break; }
            if (b1 - b0 == 1) {
                // stored block of a single-block frame: nothing will ever reference it; lz4_stored_frames_copy_kernel
                // moves it from the stream to the output, this wave only vouches for the bounds
                if (frame_out + sz > out_bytes) { bad = true; break; }
                pos += sz;
                flushed = pos;
                continue;
            }
            // stored block of a linked frame: through the ring, later blocks may reference it
            for (uint32_t i = 0; i < sz; i += 64) {
                const uint32_t cnt = sz - i < 64 ? sz - i : 64;
                if ((uint32_t)lane < cnt) ring[(pos + lane) & (DEC_RING - 1)] = src[i + lane];
                wave_lds_sync();
                pos += cnt;
                flush(false);
            }
            continue;
        }
        const uint32_t block_start = pos;
        uint32_t ip = 0;
        uint32_t sbase = 0, shi = 0;                               // stage holds block bytes [sbase, shi)
        uint32_t one_by_one = 0, backoff = 8;                      // sequences to take singly before the next batch attempt
        auto fill = [&](uint32_t at) {
            sbase = at & ~15u;
#pragma unroll
            for (uint32_t j = 0; j < DEC_IN / 1024; ++j) {
                const uint32_t a = sbase + j * 1024u + (uint32_t)lane * 16u;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (a + 16u <= sz) v = ld_u128(src + a);
                else if (a < sz) {                                  // the block's last, partial 16 bytes: never read past it
                    uint32_t wv[4] = {0, 0, 0, 0};
                    for (uint32_t k = 0; a + k < sz; ++k) wv[k >> 2] |= (uint32_t)src[a + k] << (8u * (k & 3u));
                    v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
                }
                const v4u vv = {v.x, v.y, v.z, v.w};
                *reinterpret_cast<SQY_LDS v4u*>(stage + j * 1024u + (uint32_t)lane * 16u) = vv;
            }
            wave_lds_sync();
            shi = sbase + DEC_IN < sz ? sbase + DEC_IN : sz;
        };
        auto need = [&](uint32_t at, uint32_t cnt) { if (at < sbase || at + cnt > shi) fill(at); };   // cnt <= DEC_IN - 16, at + cnt <= sz
        auto sbyte_at = [&](uint32_t at) -> uint32_t { return stage[at - sbase]; };
        // A run of length-extension bytes from `at` on (255 .. 255, closed by a byte below 255), 64 bytes per LDS round trip (round 5: byte by
        // byte, a round trip each, the thousand 255s of a chunk of zeros were a seventh of a millisecond -- the frames of the nearly empty
        // planes, a few hundred sequences of kilobyte matches, were the slowest of the bench stack).  acc += the bytes; at -> behind the run;
        // false: the block ends inside the run.
        auto ext_run = [&](uint32_t& at, uint32_t& acc) -> bool {
            for (;;) {
                if (at >= sz) return false;
                const uint32_t cnt = sz - at < 64u ? sz - at : 64u;
                need(at, cnt);
                const uint32_t bv = (uint32_t)lane < cnt ? (uint32_t)stage[at - sbase + (uint32_t)lane] : 0u;
                const uint64_t closing = ballot((uint32_t)lane < cnt && bv != 255u);
                if (closing) {
                    const uint32_t k = ctz64(closing);
                    acc += 255u * k + lane_read(bv, k);
                    at += k + 1u;
                    return true;
                }
                acc += 255u * cnt;
                at += cnt;
            }
        };

        // cnt <= 1024 bytes from LDS (ring or stage: `from`, contiguous) onto the ring at dp (dp + cnt <= DEC_RING), 16 bytes per lane and
        // the last cnt % 16 one per lane; every read is issued before the first write.  (A lane's 16-byte read may run up to 15 bytes
        // past the source -- still inside this kernel's LDS array, and not used.)
        auto wide_copy = [&](const lds_u8* from, uint32_t dp, uint32_t cnt) {
            const uint32_t full = cnt >> 4, r = cnt & 15u;
            v4u_any v = {0, 0, 0, 0};
            uint32_t t = 0;
            if ((uint32_t)lane < full) v = *reinterpret_cast<const SQY_LDS v4u_any*>(from + (uint32_t)lane * 16u);
            if ((uint32_t)lane < r) t = from[full * 16u + (uint32_t)lane];
            if ((uint32_t)lane < full) *reinterpret_cast<SQY_LDS v4u_any*>(ring + dp + (uint32_t)lane * 16u) = v;
            if ((uint32_t)lane < r) ring[dp + full * 16u + (uint32_t)lane] = (uint8_t)t;
            wave_lds_sync();
        };
        // match copy: `ml` bytes from `offset` bytes back, onto the ring at pos
        auto copy_match = [&](uint32_t offset, uint32_t ml) {
            if (DEC_RING < 65536u && offset > DEC_RING) {
                // behind the ring: those bytes have been handed to global memory (at most DEC_FLUSH_PIECE - 1 + DEC_STEP_MAX bytes are
                // not, and offset > DEC_RING is more than that).  Loads bypass the L1 (sc1): the line may have been read before this
                // wave's later stores to it.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint8_t* const gsrc = out + frame_out;
                for (uint32_t j = 0; j < ml;) {                       // up to 1 KiB per step (round 4; before: 64 bytes)
                    const uint32_t dp = pos & (DEC_RING - 1);
                    uint32_t cnt = ml - j < 1024u ? ml - j : 1024u;
                    cnt = cnt < DEC_RING - dp ? cnt : DEC_RING - dp;
                    if (j + cnt + DEC_FLUSH_PIECE + DEC_STEP_MAX > offset) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the source was written by this very match)
                    const uint32_t full = cnt >> 4, r = cnt & 15u;
                    const uint8_t* const g = gsrc + (pos - offset);
                    uint4 v = make_uint4(0, 0, 0, 0);
                    uint32_t t = 0;
                    if ((uint32_t)lane < full) v = ld_u128_agent(g + (uint32_t)lane * 16u);
                    if ((uint32_t)lane < r) t = __hip_atomic_load(g + full * 16u + (uint32_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)lane < full) { const v4u_any vv = {v.x, v.y, v.z, v.w}; *reinterpret_cast<SQY_LDS v4u_any*>(ring + dp + (uint32_t)lane * 16u) = vv; }
                    if ((uint32_t)lane < r) ring[dp + full * 16u + (uint32_t)lane] = (uint8_t)t;
                    wave_lds_sync();
                    pos += cnt;
                    j += cnt;
                    flush(false);
                }
                return;
            }
            // inside the ring.  A short match is one byte per lane (periodic extension when it overlaps its own output).  Longer ones
            // go up to 1 KiB per step, 16 bytes per lane (round 4; before: 64 bytes per step up to 2 KiB): a step may copy as many
            // bytes as are known to repeat in front of the cursor -- `period` of them, a multiple of the offset that doubles with every
            // step that uses it up (byte p equals byte p - offset, hence p - period) --, so its source ends where its destination
            // begins and all reads come before the writes: one LDS round trip per step.
            if (ml <= 64u) {
                uint32_t lm = (uint32_t)lane;
                if (offset < 64u && offset < ml) lm = (uint32_t)lane % offset;    // (a branch: the division is forty instructions)
                uint32_t v = 0;
                if ((uint32_t)lane < ml) v = ring[(pos - offset + lm) & (DEC_RING - 1)];
                if ((uint32_t)lane < ml) ring[(pos + lane) & (DEC_RING - 1)] = (uint8_t)v;
                wave_lds_sync();
                pos += ml;
                flush(false);
                return;
            }
            uint32_t rem = ml, period = offset;
            if (offset < 64u) {
                // the first 64 bytes by periodic extension, then whole periods: the largest multiple of the offset inside what is written
                const uint32_t lm = (uint32_t)lane % offset;
                const uint32_t v = ring[(pos - offset + lm) & (DEC_RING - 1)];
                ring[(pos + lane) & (DEC_RING - 1)] = (uint8_t)v;
                wave_lds_sync();
                pos += 64u;
                rem -= 64u;
                flush(false);
                period = ((64u + offset) / offset) * offset;            // <= 64 + offset bytes repeat in front of the cursor
            }
            while (rem) {
                const uint32_t sp = (pos - period) & (DEC_RING - 1), dp = pos & (DEC_RING - 1);
                uint32_t cnt = rem < 1024u ? rem : 1024u;
                cnt = cnt < period ? cnt : period;
                cnt = cnt < DEC_RING - sp ? cnt : DEC_RING - sp;            // neither side wraps inside a step
                cnt = cnt < DEC_RING - dp ? cnt : DEC_RING - dp;
                wide_copy(ring + sp, dp, cnt);
                pos += cnt;
                rem -= cnt;
                flush(false);
                if (cnt == period && period < 1024u) period <<= 1;
            }
        };
        while (ip < sz) {
            // Batch: the next 64 compressed bytes, one per lane.  Every lane parses the sequence that WOULD start at its byte
            // (token, literal count < 15, offset, at most one match-length extension byte); a scalar walk from lane 0 along
            // the "next start" links marks the real starts; their output positions are one prefix sum; the output then comes
            // 64 bytes per step (below).  A single wave's serial chain pays for every instruction, taken branch and LDS round
            // trip: ~100 cycles per short sequence this way against ~800 when each is parsed and copied on its own.
            if (one_by_one) --one_by_one;
            else if (ip + 96u <= sz) {
                need(ip, 96);
                const uint32_t wi = ip - sbase + (uint32_t)lane;
                const uint32_t tokb = stage[wi];
                const uint32_t flit = tokb >> 4, fml = tokb & 15u;
                const uint32_t w = lds_ld_u32_via_aligned(stage, wi + 1u + flit);  // offset, first extension byte (stage: 16-byte aligned)
                const uint32_t offs = w & 0xffffu, ext = (w >> 16) & 0xffu;
                const bool okl = flit < 15u && (fml < 15u || ext < 255u);
                const uint32_t mlen = fml < 15u ? fml + 4u : 19u + ext;           // <= 273
                const uint32_t nxt = (uint32_t)lane + 3u + flit + (fml == 15u ? 1u : 0u);
                // the walk along the "next start" links: a lane read, a bit set and a compare per sequence (round 5; the compiler's loop
                // took fifteen scalar instructions and three branches per step -- a fifth of a batch's time on streams of short sequences).
                // step = where the next sequence starts, 0 = this one does not fit the mould or ends behind the 64 bytes: the walk stops
                const uint32_t step = (okl && nxt <= 64u) ? nxt : 0u;
                uint64_t starts = 0;
                uint32_t cur = 0;
                {
                    uint32_t t0;
                    asm volatile(
                        "1:\n\t"
                        "v_readlane_b32 %[t0], %[step], %[cur]\n\t"
                        "s_cmp_eq_u32 %[t0], 0\n\t"
                        "s_cbranch_scc1 2f\n\t"
                        "s_bitset1_b64 %[st], %[cur]\n\t"
                        "s_mov_b32 %[cur], %[t0]\n\t"
                        "s_cmp_lt_u32 %[cur], 64\n\t"
                        "s_cbranch_scc1 1b\n"
                        "2:"
                        : [t0] "=&s"(t0), [cur] "+s"(cur), [st] "+s"(starts)
                        : [step] "v"(step)
                        : "scc");
                }
                if (starts) {
                    const bool is_start = (starts >> lane) & 1ull;
                    uint32_t inc = is_start ? flit + mlen : 0u;                    // inclusive prefix sum of the output lengths
                    inc += __builtin_amdgcn_update_dpp(0u, inc, 0x111, 0xf, 0xf, false);   // row_shr:1
                    inc += __builtin_amdgcn_update_dpp(0u, inc, 0x112, 0xf, 0xf, false);   // row_shr:2
                    inc += __builtin_amdgcn_update_dpp(0u, inc, 0x114, 0xf, 0xf, false);   // row_shr:4
                    inc += __builtin_amdgcn_update_dpp(0u, inc, 0x118, 0xf, 0xf, false);   // row_shr:8
                    inc += __builtin_amdgcn_update_dpp(0u, inc, 0x142, 0xa, 0xf, false);   // row_bcast:15
                    inc += __builtin_amdgcn_update_dpp(0u, inc, 0x143, 0xc, 0xf, false);   // row_bcast:31
                    const uint32_t total = lane_read(inc, 63);
                    const uint32_t rel = inc - (is_start ? flit + mlen : 0u);      // output of this start begins at pos + rel
                    const uint32_t base = pos;
                    // refuse (to the one-by-one path below, which answers exactly): an impossible offset, a block that overflows
                    const bool wrong = is_start && (offs == 0u || offs > base + rel + flit);
                    const uint32_t info = flit | (mlen << 4) | (offs << 16);
                    const uint32_t nst = (uint32_t)__builtin_popcountll(starts);
                    const bool fits = !ballot(wrong) && base - block_start + total <= block_bytes;
                    if (fits && total < 64u * nst) {
                        // short sequences: the batch's output is produced 64 bytes per step, one byte per lane.  A lane finds
                        // the sequence that owns its byte (marks at the first output byte of every sequence, running maximum),
                        // then takes a literal from the stage or a match byte from `offset` back -- out of the ring, or out of
                        // this very step (a lane further left; chains such as runs resolve by pointer doubling).
                        const uint32_t wbase = ip - sbase;
                        uint32_t carry = 0;
                        for (uint32_t r0 = 0; r0 < total; r0 += 64u) {
                            const uint32_t q = r0 + (uint32_t)lane;
                            owner_mark[lane] = 0;
                            wave_lds_sync();
                            if (is_start && rel - r0 < 64u) owner_mark[rel - r0] = (uint8_t)(lane + 1);
                            wave_lds_sync();
                            uint32_t mx = owner_mark[lane];
                            if (lane == 0 && mx == 0u) mx = carry;
                            {
                                uint32_t t;
                                t = __builtin_amdgcn_update_dpp(0u, mx, 0x111, 0xf, 0xf, false); mx = mx > t ? mx : t;
                                t = __builtin_amdgcn_update_dpp(0u, mx, 0x112, 0xf, 0xf, false); mx = mx > t ? mx : t;
                                t = __builtin_amdgcn_update_dpp(0u, mx, 0x114, 0xf, 0xf, false); mx = mx > t ? mx : t;
                                t = __builtin_amdgcn_update_dpp(0u, mx, 0x118, 0xf, 0xf, false); mx = mx > t ? mx : t;
                                t = __builtin_amdgcn_update_dpp(0u, mx, 0x142, 0xa, 0xf, false); mx = mx > t ? mx : t;
                                t = __builtin_amdgcn_update_dpp(0u, mx, 0x143, 0xc, 0xf, false); mx = mx > t ? mx : t;
                            }
                            carry = lane_read(mx, 63);
                            const uint32_t s_of = mx - 1u;                           // (byte 0 of the batch belongs to lane 0's sequence)
                            const uint32_t inf = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s_of * 4u), (int)info);
                            const uint32_t rl = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s_of * 4u), (int)rel);
                            const uint32_t k = q - rl, fl = inf & 15u, off = inf >> 16;
                            const bool active = q < total;
                            const bool is_lit = k < fl;
                            const uint32_t roundpos = base + r0;
                            const uint32_t P = base + q - off;                       // match byte: equals the byte at frame position P
                            const bool in_step = active && !is_lit && P >= roundpos;
                            uint32_t val = 0;
                            if (active && is_lit) val = stage[wbase + s_of + 1u + k];
                            if (DEC_RING < 65536u) {
                                const bool far = active && !is_lit && off > DEC_RING;      // behind the ring: flushed long ago
                                if (ballot(far)) {
                                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                    if (far) val = __hip_atomic_load(out + frame_out + P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                                if (active && !is_lit && !in_step && !far) val = ring[P & (DEC_RING - 1)];
                            } else {
                                if (active && !is_lit && !in_step) val = ring[P & (DEC_RING - 1)];
                            }
                            bool has = !in_step;
                            uint32_t dep = in_step ? P - roundpos : (uint32_t)lane;     // < lane
                            while (ballot(!has)) {
                                const uint32_t g = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(dep * 4u), (int)((has ? 0x80000000u : 0u) | (dep << 8) | val));
                                if (!has) {
                                    if (g >> 31) { val = g & 0xffu; has = true; }
                                    else dep = (g >> 8) & 63u;
                                }
                            }
                            if (active) ring[(roundpos + (uint32_t)lane) & (DEC_RING - 1)] = (uint8_t)val;
                            wave_lds_sync();
                            pos = roundpos + (total - r0 < 64u ? total - r0 : 64u);
                            flush(false);
                        }
                        ip += cur;
                        backoff = 8;
                        continue;
                    }
                    // longer sequences (the batch still knows where each starts, how long it is and where it goes): all literals
                    // of the batch reach the ring in one step -- lane i holds byte i of the window; it belongs to the last start
                    // s <= i and is literal number i - s - 1 of that sequence --, then the matches in order, each with all its
                    // reads in front of its writes.  Not when a match reaches so far back that literals written ahead of it
                    // would land on its source.
                    const bool clash = is_start && offs <= DEC_RING && offs + total > DEC_RING;
                    if (fits && nst >= 8u && !ballot(clash)) {              // (fewer do not pay for the batch's parse)
                        {
                            const uint64_t below = starts & (~0ull >> (63u - (uint32_t)lane));
                            const uint32_t s_of = 63u - (uint32_t)__builtin_clzll(below | 1ull);
                            const uint32_t key = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s_of * 4u), (int)(rel | (flit << 16)));
                            const uint32_t k = (uint32_t)lane - s_of - 1u;          // (lane == s_of: wraps, never < flit)
                            if ((uint32_t)lane < cur && k < (key >> 16))
                                ring[(base + (key & 0xffffu) + k) & (DEC_RING - 1)] = (uint8_t)tokb;
                            wave_lds_sync();
                        }
                        uint64_t todo = starts;
                        while (todo) {
                            const uint32_t sl = ctz64(todo);
                            todo &= todo - 1ull;
                            const uint32_t inf = lane_read(info, sl);
                            const uint32_t off = inf >> 16, mln = (inf >> 4) & 0xfffu;
                            pos = base + lane_read(rel, sl) + (inf & 15u);
                            copy_match(off, mln);
                        }
                        ip += cur;
                        backoff = 8;
                        continue;
                    }
                    if (fits) { one_by_one = backoff > nst ? backoff : nst; backoff = backoff < 256u ? backoff * 2u : 256u; }
                } else {
                    // not even the first sequence fits the batch's mould (15+ literals, a long match): the same back-off
                    one_by_one = backoff;
                    backoff = backoff < 256u ? backoff * 2u : 256u;
                }
            }
            // one sequence at a time: whatever the batch does not take (long literal runs, long matches, the block's tail).
            // A sequence whose header (token, a few literals, offset, match-length extension bytes) lies inside 16 bytes
            // is parsed out of ONE 16-byte broadcast read of the stage
            uint32_t token;
            bool have_token = false;
            if (ip + 16u <= sz) {
                need(ip, 16);
                // (five aligned dwords and four v_alignbyte instead of one 16-byte read at a byte address: no replay, ~100 cycles against 147)
                uint4 hw;
                {
                    const uint32_t at = ip - sbase, sh = at & 3u;
                    const uint4 d = lds_ld_4dw(stage + (at & ~3u));
                    const uint32_t d4 = *reinterpret_cast<const volatile SQY_LDS uint32_t*>(stage + (at & ~3u) + 16u);
                    hw = make_uint4(__builtin_amdgcn_alignbyte(d.y, d.x, sh), __builtin_amdgcn_alignbyte(d.z, d.y, sh),
                                    __builtin_amdgcn_alignbyte(d.w, d.z, sh), __builtin_amdgcn_alignbyte(d4, d.w, sh));
                }
                const uint32_t w0 = sgpr(hw.x);
                const uint32_t tok = w0 & 0xffu, flit = tok >> 4, fml = tok & 15u;
                token = tok; have_token = true;
                if (flit <= 13u && fml < 15u) {
                    // no extension bytes at all: token, literals, offset
                    const uint32_t w1 = sgpr(hw.y), w2 = sgpr(hw.z), w3 = sgpr(hw.w);
                    const uint32_t oi = 1u + flit;
                    const uint64_t lo = ((uint64_t)w1 << 32) | w0, hi = ((uint64_t)w3 << 32) | w2;
                    const uint32_t b0 = oi < 8u ? (uint32_t)(lo >> (8u * oi)) & 0xffu : (uint32_t)(hi >> (8u * (oi - 8u))) & 0xffu;
                    const uint32_t oj = oi + 1u;
                    const uint32_t b1 = oj < 8u ? (uint32_t)(lo >> (8u * oj)) & 0xffu : (uint32_t)(hi >> (8u * (oj - 8u))) & 0xffu;
                    const uint32_t offset = b0 | (b1 << 8);
                    const uint32_t ml = fml + 4u;
                    if (offset == 0 || offset > pos + flit || pos - block_start + flit + ml > block_bytes) { bad = true; break; }
                    if (flit) {
                        const uint32_t bi = 1u + (uint32_t)lane;               // literal k is window byte 1 + k
                        const uint32_t wsel = bi < 4u ? w0 : bi < 8u ? w1 : bi < 12u ? w2 : w3;
                        if ((uint32_t)lane < flit) ring[(pos + lane) & (DEC_RING - 1)] = (uint8_t)(wsel >> (8u * (bi & 3u)));
                        wave_lds_sync();
                        pos += flit;
                    }
                    ip += 3u + flit;
                    if (DEC_RING < 65536u && offset > DEC_RING) { copy_match(offset, ml); continue; }
                    const uint32_t lm = offset >= 64u ? (uint32_t)lane : (uint32_t)lane % offset;
                    uint32_t v = 0;
                    if ((uint32_t)lane < ml) v = ring[(pos - offset + lm) & (DEC_RING - 1)];
                    if ((uint32_t)lane < ml) ring[(pos + lane) & (DEC_RING - 1)] = (uint8_t)v;
                    wave_lds_sync();
                    pos += ml;
                    flush(false);
                    continue;
                }
                if (flit <= 11u && fml == 15u) {
                    // match-length extension bytes inside the window
                    const uint32_t w1 = sgpr(hw.y), w2 = sgpr(hw.z), w3 = sgpr(hw.w);
                    const uint64_t lo = ((uint64_t)w1 << 32) | w0, hi = ((uint64_t)w3 << 32) | w2;
                    auto wbyte = [&](uint32_t i) -> uint32_t {      // window byte i (uniform), i < 16
                        return i < 8u ? (uint32_t)(lo >> (8u * i)) & 0xffu : (uint32_t)(hi >> (8u * (i - 8u))) & 0xffu;
                    };
                    const uint32_t offset = wbyte(1u + flit) | (wbyte(2u + flit) << 8);
                    uint32_t ml = 15u, used = 3u + flit;
                    bool parsed = false;
                    while (used < 16u) {
                        const uint32_t sbyte = wbyte(used++);
                        ml += sbyte;
                        if (sbyte != 255u) { parsed = true; break; }
                    }
                    if (parsed) {
                        ml += 4u;
                        if (offset == 0 || offset > pos + flit || pos - block_start + flit + ml > block_bytes) { bad = true; break; }
                        if (flit) {
                            const uint32_t bi = 1u + (uint32_t)lane;
                            const uint32_t wsel = bi < 4u ? w0 : bi < 8u ? w1 : bi < 12u ? w2 : w3;
                            if ((uint32_t)lane < flit) ring[(pos + lane) & (DEC_RING - 1)] = (uint8_t)(wsel >> (8u * (bi & 3u)));
                            wave_lds_sync();
                            pos += flit;
                        }
                        ip += used;
                        copy_match(offset, ml);
                        continue;
                    }
                }
                // The general sequence in TWO reads (round 4): the window holds the token and the literal-length bytes; behind the
                // literals a second window holds the offset and the match-length bytes.  Its read is issued in front of the literal
                // copy and looked at behind it, so the copy's LDS round trip hides it.  (Before: a dependent broadcast read per
                // header byte -- six in a row for 15+ literals and a match of 19+ bytes, what the diff3x3x1 planes are made of.)
                {
                    const uint32_t w1 = sgpr(hw.y), w2 = sgpr(hw.z), w3 = sgpr(hw.w);
                    const uint64_t lo = ((uint64_t)w1 << 32) | w0, hi = ((uint64_t)w3 << 32) | w2;
                    auto wbyte = [&](uint32_t i) -> uint32_t { return i < 8u ? (uint32_t)(lo >> (8u * i)) & 0xffu : (uint32_t)(hi >> (8u * (i - 8u))) & 0xffu; };
                    uint32_t lit = flit, used = 1u;
                    bool lit_ok = flit < 15u;
                    while (!lit_ok && used < 16u) {
                        const uint32_t sb = wbyte(used++);
                        lit += sb;
                        lit_ok = sb != 255u;
                    }
                    const uint32_t ipl = ip + used;                                  // the first literal
                    if (lit_ok && lit <= DEC_IN - 128u && ipl + lit + 16u <= sz) {
                        if (pos - block_start + lit > block_bytes) { bad = true; break; }
                        need(ip, used + lit + 16u);
                        const uint32_t at = ipl + lit - sbase, sh = at & 3u;
                        const uint4 d = lds_ld_4dw(stage + (at & ~3u));
                        const uint32_t d4 = *reinterpret_cast<const volatile SQY_LDS uint32_t*>(stage + (at & ~3u) + 16u);
                        for (uint32_t i = 0; i < lit;) {
                            const uint32_t dp = pos & (DEC_RING - 1);
                            uint32_t cnt = lit - i < 1024u ? lit - i : 1024u;
                            cnt = cnt < DEC_RING - dp ? cnt : DEC_RING - dp;
                            wide_copy(stage + (ipl - sbase + i), dp, cnt);
                            pos += cnt;
                            i += cnt;
                            flush(false);
                        }
                        const uint32_t x0 = sgpr(__builtin_amdgcn_alignbyte(d.y, d.x, sh)), x1 = sgpr(__builtin_amdgcn_alignbyte(d.z, d.y, sh));
                        const uint32_t x2 = sgpr(__builtin_amdgcn_alignbyte(d.w, d.z, sh)), x3 = sgpr(__builtin_amdgcn_alignbyte(d4, d.w, sh));
                        const uint64_t lo2 = ((uint64_t)x1 << 32) | x0, hi2 = ((uint64_t)x3 << 32) | x2;
                        auto xbyte = [&](uint32_t i) -> uint32_t { return i < 8u ? (uint32_t)(lo2 >> (8u * i)) & 0xffu : (uint32_t)(hi2 >> (8u * (i - 8u))) & 0xffu; };
                        const uint32_t offset = x0 & 0xffffu;
                        uint32_t ml = fml, used2 = 2u;
                        bool ml_ok = fml < 15u;
                        while (!ml_ok && used2 < 16u) {
                            const uint32_t sb = xbyte(used2++);
                            ml += sb;
                            ml_ok = sb != 255u;
                        }
                        ip = ipl + lit + used2;
                        if (!ml_ok && !ext_run(ip, ml)) bad = true;                  // (a match of more than 3.3 KiB: the rest of its length bytes)
                        ml += 4u;
                        if (bad || offset == 0 || offset > pos || pos - block_start + ml > block_bytes) { bad = true; break; }
                        copy_match(offset, ml);
                        continue;
                    }
                }
            }
            if (!have_token) { need(ip, 1); token = sbyte_at(ip); }
            ++ip;
            uint32_t lit = token >> 4;
            if (lit == 15 && !ext_run(ip, lit)) bad = true;
            if (bad || ip + lit > sz || pos - block_start + lit > block_bytes) { bad = true; break; }
            if (lit <= DEC_IN - 64u) {
                if (lit) need(ip, lit);
                for (uint32_t i = 0; i < lit;) {                          // up to 1 KiB per step (round 4; before: 64 bytes)
                    const uint32_t dp = pos & (DEC_RING - 1);
                    uint32_t cnt = lit - i < 1024u ? lit - i : 1024u;
                    cnt = cnt < DEC_RING - dp ? cnt : DEC_RING - dp;
                    wide_copy(stage + (ip - sbase + i), dp, cnt);
                    pos += cnt;
                    i += cnt;
                    flush(false);
                }
            } else {
                for (uint32_t i = 0; i < lit; i += 64) {                // long literal run: stream -> ring
                    const uint32_t cnt = lit - i < 64 ? lit - i : 64;
                    if ((uint32_t)lane < cnt) ring[(pos + lane) & (DEC_RING - 1)] = src[ip + i + lane];
                    wave_lds_sync();
                    pos += cnt;
                    flush(false);
                }
            }
            ip += lit;
            if (ip >= sz) break;                                   // last sequence: literals only
            if (ip + 2 > sz) { bad = true; break; }
            need(ip, 2);
            const uint32_t offset = sbyte_at(ip) | (sbyte_at(ip + 1) << 8);
            ip += 2;
            uint32_t ml = token & 15u;
            if (ml == 15 && !ext_run(ip, ml)) bad = true;
            ml += 4;
            if (bad || offset == 0 || offset > pos || pos - block_start + ml > block_bytes) { bad = true; break; }
            copy_match(offset, ml);
        }
    }
    flush(true);
    // a frame has to deliver exactly its share of the output: the whole stream when it is the only frame (serial layout),
    // else one chunk (the last frame the remainder) -- short frames must not leave the destination half-written
    {
        const uint64_t room = frame_out < out_bytes ? out_bytes - frame_out : 0;
        const uint64_t expect = remap ? frame_stride : (gridDim.x == 1 || room < frame_stride) ? room : frame_stride;   // (remap: whole chunks only)
        if ((uint64_t)pos != expect) bad = true;
    }
    if (bad && lane == 0) atomicExch(errflag, 1u);
}

// ---- the same decode by TWO wavefronts per frame (round 5) ------------------------------------------------------------------------------
// A frame's decode is one dependent chain, and a lone wavefront issues an instruction every five cycles or so: 2000 cycles per sequence on
// the sparse planes of the bench stack, 380 per sequence on streams of short ones.  Taken apart (a build whose copies do nothing): 42 % / 58 %
// of that is finding out WHAT the sequences are -- a chain through the compressed bytes alone -- and the rest is moving the bytes.  So two
// wavefronts share a frame: wave 0 walks the compressed bytes and writes what it finds -- {first literal, literals, offset, match length},
// up to 64 sequences to a unit, two units in LDS --, wave 1 takes a unit into its registers, gives the slot back and does the copies (the
// rounds of 64 output bytes for short sequences, one sequence after the other for long ones), while wave 0 is two units ahead.  Frames of
// ONE block only (what the chunked layout writes); everything else, and streams with so many compressed frames that the decode is bound
// by the instructions issued rather than by one chain (the 8 KiB ring's range), stays with the kernel above.
// Hand-over: prod / cons count units in LDS, a wave that waits sleeps and looks again; `stop` ends both (an error on either side); every wait
// is bounded (2^22 looks: seconds), so that the grid drains whatever happens.
// Diagnostic builds (tools/dec_stats.sh; never the product): -DSQY_DEC_STATS makes both waves count what they do and leave the counts in the
// first 256 bytes of their frame's OUTPUT (which is garbage then); -DSQY_DEC_INERT makes wave 1 take its units and do nothing.
#ifdef SQY_DEC_STATS
#define SQY_DST(x) x
#else
#define SQY_DST(x)
#endif
constexpr uint32_t DEC2_PIN = 4096;                   // wave 0's own stage of compressed bytes: a ring of two halves (+ 32 bytes of mirror)
constexpr uint32_t DEC2_UNIT = 64;                    // sequences per unit

// wave 0 of lz4_frames_decode2_kernel (a function of its own: the two roles in one body had the register allocator spill)
__device__ __noinline__ void lz4_decode2_parse(const uint8_t* __restrict__ src, uint32_t sz, lds_u8* pstage, SQY_LDS uint4* units,
                                               volatile SQY_LDS uint32_t* ctrl, int lane, uint64_t* stats)
{
    constexpr uint32_t SPIN = 1u << 22;
    SQY_DST(uint64_t st_t0 = __builtin_amdgcn_s_memtime(); uint64_t st_a = st_t0; uint64_t st_batches = 0; uint64_t st_nst = 0; uint64_t st_singles = 0;
            uint64_t st_cb = 0; uint64_t st_cs = 0; uint64_t st_pub = 0; uint64_t st_cp = 0; uint64_t st_stall = 0; uint64_t st_ext = 0;)
    // ================================================= wave 0: what the sequences are =================================================
    // Its stage is a ring of two halves; the half behind the one it reads is on its way from global memory while it parses (the
    // sparse planes' sequences are 100-300 compressed bytes apart: with a stage filled when it runs out, a third of this wave's
    // time was the wait for the fill) -- by LDS-DMA, as the encoder's window: no registers held across the parse loop.
    // lo_ok .. hi_ok: the block bytes that can be read; the first 32 bytes of the ring are mirrored behind its end, so that the short
    // reads below never wrap.  A half is fetched into the slot of the half BEHIND the one being read, i.e. only once the reads have
    // moved into the newest half.
    constexpr uint32_t PH = DEC2_PIN / 2u, PR = DEC2_PIN;
    uint32_t lo_ok = 0, hi_ok = 0;
    bool pend = false;                                                // [hi_ok, hi_ok + PH) is on its way
    auto dma_issue = [&]() {
#pragma unroll
        for (uint32_t j = 0; j < PH / 1024u; ++j) {
            const uint32_t a = hi_ok + j * 1024u + (uint32_t)lane * 16u;
            if (a + 16u <= sz) {                                      // lanes past the block's last whole 16 bytes stay off
                const uint32_t lds_dst = sgpr((uint32_t)(uintptr_t)pstage + ((hi_ok + j * 1024u) & (PR - 1u)));   // wave-uniform; the copy adds lane * 16
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src + a), "s"(lds_dst) : "memory");
            }
        }
        if (hi_ok >= PH && lo_ok < hi_ok - PH) lo_ok = hi_ok - PH;   // (the slot held [hi_ok - PR, hi_ok - PH))
        pend = true;
    };
    auto dma_commit = [&]() {
        SQY_DST(const uint64_t q0 = __builtin_amdgcn_s_memtime();)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SQY_DST(st_stall += __builtin_amdgcn_s_memtime() - q0;)
        wave_lds_sync();
        // the block's last, partial 16 bytes, when they lie in this half: byte by byte (never a read past the block)
        if (sz < hi_ok + PH && (sz & 15u) && (sz & ~15u) >= hi_ok) {
            const uint32_t a = (sz & ~15u) + (uint32_t)lane;
            if (a < sz) pstage[a & (PR - 1u)] = src[a];
            wave_lds_sync();
        }
        if ((hi_ok & (PR - 1u)) == 0u) {                              // the half with ring offset 0: its first 32 bytes once more behind the end
            if (lane < 8) *reinterpret_cast<SQY_LDS uint32_t*>(pstage + PR + 4u * (uint32_t)lane) = *reinterpret_cast<const SQY_LDS uint32_t*>(pstage + 4u * (uint32_t)lane);
            wave_lds_sync();
        }
        hi_ok += PH;
        pend = false;
    };
    auto need = [&](uint32_t at, uint32_t cnt) {                      // cnt <= PH - 16, at + cnt <= sz
        while (!(at >= lo_ok && at + cnt <= hi_ok)) {
            if (!(at >= lo_ok && at < hi_ok + PH)) {                  // the first use, or a jump over more than the half on its way
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (a copy in flight must not land on top of the new ones)
                lo_ok = hi_ok = at & ~(PH - 1u);
                pend = false;
            }
            if (!pend) dma_issue();
            dma_commit();
        }
        if (!pend && hi_ok < sz && hi_ok >= PH && at >= hi_ok - PH) dma_issue();      // the next half, early
    };
    auto byte_at = [&](uint32_t at) -> uint32_t { need(at, 1); return pstage[at & (PR - 1u)]; };
    // the 16 bytes at `at` (at + 16 <= sz) as two uniform 64-bit words
    auto window = [&](uint32_t at, uint64_t& lo, uint64_t& hi) {
        need(at, 16);
        const uint32_t o = at & (PR - 1u), sh = o & 3u;
        const uint4 d = lds_ld_4dw(pstage + (o & ~3u));
        const uint32_t d4 = *reinterpret_cast<const volatile SQY_LDS uint32_t*>(pstage + (o & ~3u) + 16u);
        const uint32_t x0 = sgpr(__builtin_amdgcn_alignbyte(d.y, d.x, sh)), x1 = sgpr(__builtin_amdgcn_alignbyte(d.z, d.y, sh));
        const uint32_t x2 = sgpr(__builtin_amdgcn_alignbyte(d.w, d.z, sh)), x3 = sgpr(__builtin_amdgcn_alignbyte(d4, d.w, sh));
        lo = ((uint64_t)x1 << 32) | x0; hi = ((uint64_t)x3 << 32) | x2;
    };
    auto wb = [](uint64_t lo, uint64_t hi, uint32_t i) -> uint32_t { return i < 8u ? (uint32_t)(lo >> (8u * i)) & 0xffu : (uint32_t)(hi >> (8u * (i - 8u))) & 0xffu; };
    // a run of length-extension bytes from `at` on, 64 bytes per LDS round trip (see ext_run in the kernel above)
    auto ext_run = [&](uint32_t& at, uint32_t& acc) -> bool {
        for (;;) {
            SQY_DST(++st_ext;)
            if (at >= sz) return false;
            const uint32_t cnt = sz - at < 64u ? sz - at : 64u;
            need(at, cnt);
            const uint32_t bv = (uint32_t)lane < cnt ? (uint32_t)pstage[(at + (uint32_t)lane) & (PR - 1u)] : 0u;
            const uint64_t closing = ballot((uint32_t)lane < cnt && bv != 255u);
            if (closing) {
                const uint32_t k = ctz64(closing);
                acc += 255u * k + lane_read(bv, k);
                at += k + 1u;
                return true;
            }
            acc += 255u * cnt;
            at += cnt;
        }
    };

    uint32_t published = 0, m = 0;                         // units handed over, sequences in the unit under way
    bool stopped = false;
    // the slot of unit `published` is free once unit `published - 2` has been taken
    auto slot_wait = [&]() {
        for (uint32_t spin = 0; published >= ctrl[1] + 2u; ++spin) {
            if (ctrl[2] || spin >= SPIN) { stopped = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto publish = [&](uint32_t flags) {
        wave_lds_sync();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) { ctrl[4u + (published & 1u)] = m | (flags << 8); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) ctrl[0] = published + 1u;
        ++published;
        m = 0;
    };
    uint32_t ip = 0;
    uint32_t one_by_one = 0, backoff = 1;
    bool damaged = false, ended = false;
    slot_wait();
    while (!stopped && !ended && !damaged) {
        SQY_DST(st_a = __builtin_amdgcn_s_memtime();)
        if (ip >= sz) { ended = true; break; }             // (a block that ends behind a match: the byte count decides, as in the kernel above)
        // ---- a batch: the next 64 compressed bytes, every lane the sequence that WOULD start at its byte (see the kernel above) ----
        if (one_by_one) --one_by_one;
        else if (ip + 96u <= sz) {
            if (m + 22u > DEC2_UNIT) {
                SQY_DST(const uint64_t q0 = __builtin_amdgcn_s_memtime();)
                publish(0); slot_wait();
                SQY_DST(st_cp += __builtin_amdgcn_s_memtime() - q0; ++st_pub;)
                if (stopped) break;
            }
            need(ip, 96);
            // (both fields behind ONE read of 24 bytes per lane, picked out of registers: measured, 3.7 -> 4.4 ms on the quantised stack)
            const uint32_t wi = (ip + (uint32_t)lane) & (PR - 1u);
            const uint32_t tokb = pstage[wi];
            const uint32_t flit = tokb >> 4, fml = tokb & 15u;
            const uint32_t w = lds_ld_u32_via_aligned(pstage, (wi + 1u + flit) & (PR - 1u));
            const uint32_t lw = lds_ld_u32_via_aligned(pstage, (wi + 1u) & (PR - 1u));      // the four bytes behind the token: up to four literals travel in the record
            const uint32_t offs = w & 0xffffu, ext = (w >> 16) & 0xffu;
            const bool okl = flit < 15u && (fml < 15u || ext < 255u);
            const uint32_t mlen = fml < 15u ? fml + 4u : 19u + ext;
            const uint32_t nxt = (uint32_t)lane + 3u + flit + (fml == 15u ? 1u : 0u);
            const uint32_t step = (okl && nxt <= 64u) ? nxt : 0u;
            uint64_t starts = 0;
            uint32_t cur = 0;
            {
                uint32_t t0;
                asm volatile(
                    "1:\n\t"
                    "v_readlane_b32 %[t0], %[step], %[cur]\n\t"
                    "s_cmp_eq_u32 %[t0], 0\n\t"
                    "s_cbranch_scc1 2f\n\t"
                    "s_bitset1_b64 %[st], %[cur]\n\t"
                    "s_mov_b32 %[cur], %[t0]\n\t"
                    "s_cmp_lt_u32 %[cur], 64\n\t"
                    "s_cbranch_scc1 1b\n"
                    "2:"
                    : [t0] "=&s"(t0), [cur] "+s"(cur), [st] "+s"(starts)
                    : [step] "v"(step)
                    : "scc");
            }
            const uint32_t nst = (uint32_t)__builtin_popcountll(starts);
            if (nst) {                                                 // (whatever the walk found is parsed: it goes into the unit)
                if ((starts >> lane) & 1ull) {
                    const uint32_t r = (uint32_t)__builtin_popcountll(starts & ((1ull << lane) - 1ull));
                    const v4u rec = {flit <= 4u ? lw : ip + (uint32_t)lane + 1u, flit | (flit <= 4u ? 0x80000000u : 0u), offs, mlen};
                    *reinterpret_cast<SQY_LDS v4u*>(units + (published & 1u) * DEC2_UNIT + m + r) = rec;
                }
                m += nst;
                ip += cur;
            }
            // why the walk stopped: a sequence that runs past the 64 bytes (the next batch takes it), or one that does not fit the mould
            // (15+ literals, a long match) -- that one goes the single way; streams made of such sequences try a batch ever more rarely
            const bool mould = cur >= 64u || lane_read(okl ? 1u : 0u, cur & 63u) != 0u;
            if (nst >= 4u) backoff = 1;                                // (a stretch of short sequences: the next batch right behind the odd one out)
            SQY_DST(++st_batches; st_nst += nst; { const uint64_t q1 = __builtin_amdgcn_s_memtime(); st_cb += q1 - st_a; st_a = q1; })
            if (mould) continue;
            one_by_one = backoff - 1u;                                 // (this one, below, is the first of them)
            if (nst < 4u) backoff = backoff < 256u ? backoff * 2u : 256u;
            if (m >= DEC2_UNIT) { publish(0); slot_wait(); if (stopped) break; }
        }
        // ---- one sequence ----
        if (m >= DEC2_UNIT) { publish(0); slot_wait(); if (stopped) break; }
        uint32_t token, lit, used = 1;
        uint64_t lo = 0, hi = 0;
        bool win = false;
        uint32_t inl = 0;                                      // the four bytes behind the token (when a window holds them)
        if (ip + 16u <= sz) { window(ip, lo, hi); win = true; token = wb(lo, hi, 0); inl = (uint32_t)(lo >> 8); }
        else token = byte_at(ip);
        const bool have_inl = win;
        lit = token >> 4;
        if (lit == 15u) {
            bool ok = false;
            while (win && used < 16u) { const uint32_t sb = wb(lo, hi, used++); lit += sb; if (sb != 255u) { ok = true; break; } }
            if (!ok) {
                uint32_t at = ip + used;
                if (!ext_run(at, lit)) { damaged = true; break; }
                used = at - ip;
                win = false;                                   // (the window in hand is behind us)
            }
        }
        const uint32_t ipl = ip + used;                        // the first literal
        if (ipl > sz || lit > sz - ipl) { damaged = true; break; }
        uint32_t offset = 0, ml = 0;
        if (ipl + lit == sz) ended = true;                     // the block's last sequence: literals only
        else {
            const uint32_t ipo = ipl + lit;
            if (ipo + 2u > sz) { damaged = true; break; }
            uint32_t o2 = 0;                                  // index of the offset's first byte inside the window in hand
            if (win && used + lit + 2u <= 16u) o2 = used + lit;
            else if (ipo + 16u <= sz) { window(ipo, lo, hi); win = true; }
            else win = false;
            uint32_t used2;
            if (win) { offset = wb(lo, hi, o2) | (wb(lo, hi, o2 + 1u) << 8); used2 = o2 + 2u; }
            else { offset = byte_at(ipo) | (byte_at(ipo + 1u) << 8); used2 = 2u; }
            const uint32_t wbase = win ? ipo - o2 : ipo;          // block position of window byte 0 (no window: of the offset)
            ml = token & 15u;
            if (ml == 15u) {
                bool ok = false;
                while (win && used2 < 16u) { const uint32_t sb = wb(lo, hi, used2++); ml += sb; if (sb != 255u) { ok = true; break; } }
                if (!ok) {
                    uint32_t at = wbase + used2;
                    if (!ext_run(at, ml)) { damaged = true; break; }
                    used2 = at - wbase;
                }
            }
            ml += 4u;
            ip = wbase + used2;
        }
        if (lane == 0) {
            const bool in_rec = have_inl && lit <= 4u;         // (lit < 15: the literals begin right behind the token)
            const v4u rec = {in_rec ? inl : ipl, lit | (in_rec ? 0x80000000u : 0u), offset, ml};
            *reinterpret_cast<SQY_LDS v4u*>(units + (published & 1u) * DEC2_UNIT + m) = rec;
        }
        ++m;
        SQY_DST(++st_singles; st_cs += __builtin_amdgcn_s_memtime() - st_a;)
    }
    if (!stopped) publish(damaged ? 2u : 1u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no half of the stage may still be on its way into LDS when this wave is gone
    SQY_DST(if (lane == 0 && stats) {
        stats[0] = __builtin_amdgcn_s_memtime() - st_t0; stats[1] = st_batches; stats[2] = st_nst; stats[3] = st_singles; stats[4] = st_cb; stats[5] = st_cs;
        stats[6] = st_pub; stats[7] = st_cp; stats[8] = sz; stats[9] = st_stall; stats[10] = st_ext;
    })
}

// wave 1 of lz4_frames_decode2_kernel
template <uint32_t DEC_RING>
__device__ __noinline__ void lz4_decode2_copy(const uint8_t* __restrict__ src, uint32_t sz, uint8_t* __restrict__ out, uint64_t frame_out, uint64_t out_bytes,
                                              uint64_t block_bytes, uint64_t expect, lds_u8* ring, lds_u8* stage, lds_u8* owner_mark,
                                              SQY_LDS uint4* units, volatile SQY_LDS uint32_t* ctrl, uint32_t* __restrict__ errflag, int lane)
{
    constexpr uint32_t DEC_IN = 3072u;
    constexpr uint32_t SPIN = 1u << 22;
    static_assert(DEC_RING >= DEC_FLUSH_PIECE + 2u * DEC_STEP_MAX + 1024u, "a match behind the ring must find its source flushed: ring >= flush piece + two copy steps + margin");
    SQY_DST(uint64_t ct_t0 = __builtin_amdgcn_s_memtime(); uint64_t ct_wait = 0; uint64_t ct_units = 0; uint64_t ct_runits = 0; uint64_t ct_rounds = 0;
            uint64_t ct_cr = 0; uint64_t ct_seqs = 0; uint64_t ct_ci = 0; uint64_t ct_fill = 0; uint64_t ct_flush = 0; uint64_t ct_nflush = 0;
            uint64_t ct_short = 0; uint64_t ct_cshort = 0; uint64_t ct_long = 0; uint64_t ct_clong = 0; uint64_t ct_far = 0; uint64_t ct_cfar = 0;)
    // ===================================================== wave 1: the bytes =====================================================
    uint32_t pos = 0, flushed = 0;
    bool bad = false;
    auto flush = [&](bool all) {
        SQY_DST(const uint64_t fl0 = __builtin_amdgcn_s_memtime(); if (flushed + 4096u <= pos) ++ct_nflush;)
        // 4 KiB at a time: four reads of the ring in flight, then four stores (a KiB per call, each behind its own LDS round trip, was a fifth
        // of this wave's time).  What has not left yet stays inside the ring (a step adds at most 1 KiB), and a match that reaches behind the
        // ring (offset > DEC_RING >= 16 KiB) ends more than 15 KiB back: flushed.
        while (flushed + DEC_FLUSH_PIECE <= pos) {
            const uint64_t o = frame_out + flushed;
            if (o + DEC_FLUSH_PIECE > out_bytes) { bad = true; flushed = pos; break; }
            v4u v[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const SQY_LDS v4u*>(ring + ((flushed + q * 1024u + (uint32_t)lane * 16u) & (DEC_RING - 1)));
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) st_u128(out + o + q * 1024u + (uint32_t)lane * 16u, make_uint4(v[q].x, v[q].y, v[q].z, v[q].w));
            flushed += DEC_FLUSH_PIECE;
        }
        while (all && flushed < pos) {
            const uint32_t cnt = (pos - flushed >= 1024u) ? 1024u : (pos - flushed);
            const uint64_t o = frame_out + flushed;
            if (o + cnt > out_bytes) { bad = true; flushed = pos; break; }
            const uint32_t nvec = cnt >> 4;
            if ((uint32_t)lane < nvec) {
                const v4u v = *reinterpret_cast<const SQY_LDS v4u*>(ring + ((flushed + (uint32_t)lane * 16u) & (DEC_RING - 1)));
                st_u128(out + o + (uint32_t)lane * 16u, make_uint4(v.x, v.y, v.z, v.w));
            }
            const uint32_t done = nvec << 4;
            if ((uint32_t)lane < cnt - done) out[o + done + lane] = ring[(flushed + done + lane) & (DEC_RING - 1)];
            flushed += cnt;
        }
        SQY_DST(ct_flush += __builtin_amdgcn_s_memtime() - fl0;)
    };
    uint32_t sbase = 0, shi = 0;
    auto fill = [&](uint32_t at) {
        SQY_DST(const uint64_t f0 = __builtin_amdgcn_s_memtime();)
        sbase = at & ~15u;
#pragma unroll
        for (uint32_t j = 0; j < DEC_IN / 1024; ++j) {
            const uint32_t a = sbase + j * 1024u + (uint32_t)lane * 16u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a + 16u <= sz) v = ld_u128(src + a);
            else if (a < sz) {
                uint32_t wv[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; a + k < sz; ++k) wv[k >> 2] |= (uint32_t)src[a + k] << (8u * (k & 3u));
                v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            }
            const v4u vv = {v.x, v.y, v.z, v.w};
            *reinterpret_cast<SQY_LDS v4u*>(stage + j * 1024u + (uint32_t)lane * 16u) = vv;
        }
        wave_lds_sync();
        shi = sbase + DEC_IN < sz ? sbase + DEC_IN : sz;
        SQY_DST(ct_fill += __builtin_amdgcn_s_memtime() - f0;)
    };
    auto need = [&](uint32_t at, uint32_t cnt) { if (at < sbase || at + cnt > shi) fill(at); };       // cnt <= DEC_IN - 16, at + cnt <= sz
    auto wide_copy = [&](const lds_u8* from, uint32_t dp, uint32_t cnt) {
        const uint32_t full = cnt >> 4, r = cnt & 15u;
        v4u_any v = {0, 0, 0, 0};
        uint32_t t = 0;
        if ((uint32_t)lane < full) v = *reinterpret_cast<const SQY_LDS v4u_any*>(from + (uint32_t)lane * 16u);
        if ((uint32_t)lane < r) t = from[full * 16u + (uint32_t)lane];
        if ((uint32_t)lane < full) *reinterpret_cast<SQY_LDS v4u_any*>(ring + dp + (uint32_t)lane * 16u) = v;
        if ((uint32_t)lane < r) ring[dp + full * 16u + (uint32_t)lane] = (uint8_t)t;
        wave_lds_sync();
    };
    auto copy_match = [&](uint32_t offset, uint32_t ml) {
        SQY_DST(const uint64_t cm0 = __builtin_amdgcn_s_memtime();)
        if (DEC_RING < 65536u && offset > DEC_RING) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint8_t* const gsrc = out + frame_out;
            for (uint32_t j = 0; j < ml;) {
                const uint32_t dp = pos & (DEC_RING - 1);
                uint32_t cnt = ml - j < 1024u ? ml - j : 1024u;
                cnt = cnt < DEC_RING - dp ? cnt : DEC_RING - dp;
                if (j + cnt + DEC_FLUSH_PIECE + DEC_STEP_MAX > offset) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint32_t full = cnt >> 4, r = cnt & 15u;
                const uint8_t* const g = gsrc + (pos - offset);
                uint4 v = make_uint4(0, 0, 0, 0);
                uint32_t t = 0;
                if ((uint32_t)lane < full) v = ld_u128_agent(g + (uint32_t)lane * 16u);
                if ((uint32_t)lane < r) t = __hip_atomic_load(g + full * 16u + (uint32_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)lane < full) { const v4u_any vv = {v.x, v.y, v.z, v.w}; *reinterpret_cast<SQY_LDS v4u_any*>(ring + dp + (uint32_t)lane * 16u) = vv; }
                if ((uint32_t)lane < r) ring[dp + full * 16u + (uint32_t)lane] = (uint8_t)t;
                wave_lds_sync();
                pos += cnt;
                j += cnt;
                flush(false);
            }
            SQY_DST(++ct_far; ct_cfar += __builtin_amdgcn_s_memtime() - cm0;)
            return;
        }
        if (ml <= 64u) {
            uint32_t lm = (uint32_t)lane;
            if (offset < 64u && offset < ml) lm = (uint32_t)lane % offset;    // (a branch: the division is forty instructions)
            uint32_t v = 0;
            if ((uint32_t)lane < ml) v = ring[(pos - offset + lm) & (DEC_RING - 1)];
            if ((uint32_t)lane < ml) ring[(pos + lane) & (DEC_RING - 1)] = (uint8_t)v;
            wave_lds_sync();
            pos += ml;
            flush(false);
            SQY_DST(++ct_short; ct_cshort += __builtin_amdgcn_s_memtime() - cm0;)
            return;
        }
        uint32_t rem = ml, period = offset;
        if (offset < 64u) {
            const uint32_t lm = (uint32_t)lane % offset;
            const uint32_t v = ring[(pos - offset + lm) & (DEC_RING - 1)];
            ring[(pos + lane) & (DEC_RING - 1)] = (uint8_t)v;
            wave_lds_sync();
            pos += 64u;
            rem -= 64u;
            flush(false);
            period = ((64u + offset) / offset) * offset;
        }
        while (rem) {
            const uint32_t sp = (pos - period) & (DEC_RING - 1), dp = pos & (DEC_RING - 1);
            uint32_t cnt = rem < 1024u ? rem : 1024u;
            cnt = cnt < period ? cnt : period;
            cnt = cnt < DEC_RING - sp ? cnt : DEC_RING - sp;
            cnt = cnt < DEC_RING - dp ? cnt : DEC_RING - dp;
            wide_copy(ring + sp, dp, cnt);
            pos += cnt;
            rem -= cnt;
            flush(false);
            if (cnt == period && period < 1024u) period <<= 1;
        }
        SQY_DST(++ct_long; ct_clong += __builtin_amdgcn_s_memtime() - cm0;)
    };

    for (uint32_t taken = 0; !bad;) {
        bool timed_out = false;
        SQY_DST(const uint64_t w0 = __builtin_amdgcn_s_memtime();)
        for (uint32_t spin = 0; ctrl[0] == taken; ++spin) {
            if (ctrl[2] || spin >= SPIN) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        SQY_DST(ct_wait += __builtin_amdgcn_s_memtime() - w0; ++ct_units;)
        if (timed_out) { bad = true; break; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t meta = ctrl[4u + (taken & 1u)];
        const uint32_t m = meta & 0xffu, flags = meta >> 8;
        v4u rec = {0, 0, 0, 0};
        if ((uint32_t)lane < m) rec = *reinterpret_cast<const SQY_LDS v4u*>(units + (taken & 1u) * DEC2_UNIT + (uint32_t)lane);
        wave_lds_sync();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        ++taken;
        if (lane == 0) ctrl[1] = taken;                            // the slot is wave 0's again
        if (flags & 2u) { bad = true; break; }
#ifdef SQY_DEC_INERT
        if (flags & 1u) break;
        continue;
#endif
        if (m) {
            SQY_DST(const uint64_t u0 = __builtin_amdgcn_s_memtime();)
            // (up to four literals travel in the record itself, rec.x, instead of their place in the block: no read of the stage for them)
            const uint32_t lip = rec.x, lit = rec.y & 0x7fffffffu, inl = rec.y >> 31, off = rec.z, mlen = rec.w;
            const uint32_t len = lit + mlen;                       // (lanes >= m: 0)
            uint32_t inc = len;
            inc += __builtin_amdgcn_update_dpp(0u, inc, 0x111, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0u, inc, 0x112, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0u, inc, 0x114, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0u, inc, 0x118, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0u, inc, 0x142, 0xa, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0u, inc, 0x143, 0xc, 0xf, false);
            const uint32_t total = lane_read(inc, 63);
            const uint32_t rel = inc - len;
            const uint32_t base = pos;
            // (64 sequences of up to sz literals and 2^20-ish match bytes each: no 32-bit overflow below -- lit <= sz < 2^23, mlen < 2^23)
            const bool wrong = (uint32_t)lane < m && (lit > 0x7fffffu || mlen > 0x7fffffu || (mlen && (off == 0u || off > base + rel + lit)));
            if (ballot(wrong) || (uint64_t)base + total > block_bytes) { bad = true; break; }
            // the literals that are NOT in their records: from the first to the last of them in the block (they lie in order)
            const uint64_t staged = ballot((uint32_t)lane < m && !inl && lit > 0u);
            const uint32_t lip0 = staged ? lane_read(lip, ctz64(staged)) : 0u;
            const uint32_t lend = staged ? lane_read(lip + lit, 63u - (uint32_t)__builtin_clzll(staged)) : 0u;
            const bool any_inl = ballot((uint32_t)lane < m && inl && lit > 0u) != 0ull;
            // (a round of 64 output bytes costs about as much as one and a half sequences taken on their own: rounds below 32 bytes a sequence)
            if (total < 32u * m && total < 4096u && lend - lip0 <= DEC_IN - 32u) {
                // short sequences: the unit's output 64 bytes per step, one byte per lane (see the kernel above)
                if (lend > lip0) need(lip0, lend - lip0);
                const uint32_t inf = lit | (inl << 15) | (off << 16), rl = rel | ((inl ? 0u : lip - lip0) << 12);
                uint32_t carry = 0;
                for (uint32_t r0 = 0; r0 < total; r0 += 64u) {
                    const uint32_t q = r0 + (uint32_t)lane;
                    owner_mark[lane] = 0;
                    wave_lds_sync();
                    if ((uint32_t)lane < m && len && rel - r0 < 64u) owner_mark[rel - r0] = (uint8_t)(lane + 1);
                    wave_lds_sync();
                    uint32_t mx = owner_mark[lane];
                    if (lane == 0 && mx == 0u) mx = carry;
                    {
                        uint32_t t;
                        t = __builtin_amdgcn_update_dpp(0u, mx, 0x111, 0xf, 0xf, false); mx = mx > t ? mx : t;
                        t = __builtin_amdgcn_update_dpp(0u, mx, 0x112, 0xf, 0xf, false); mx = mx > t ? mx : t;
                        t = __builtin_amdgcn_update_dpp(0u, mx, 0x114, 0xf, 0xf, false); mx = mx > t ? mx : t;
                        t = __builtin_amdgcn_update_dpp(0u, mx, 0x118, 0xf, 0xf, false); mx = mx > t ? mx : t;
                        t = __builtin_amdgcn_update_dpp(0u, mx, 0x142, 0xa, 0xf, false); mx = mx > t ? mx : t;
                        t = __builtin_amdgcn_update_dpp(0u, mx, 0x143, 0xc, 0xf, false); mx = mx > t ? mx : t;
                    }
                    carry = lane_read(mx, 63);
                    const uint32_t s_of = mx - 1u;
                    const uint32_t inf_s = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s_of * 4u), (int)inf);
                    const uint32_t rl_s = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s_of * 4u), (int)rl);
                    uint32_t x_s = 0;
                    if (any_inl) x_s = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s_of * 4u), (int)lip);
                    const uint32_t k = q - (rl_s & 0xfffu), fl = inf_s & 0x7fffu, o = inf_s >> 16;
                    const bool inl_s = (inf_s >> 15) & 1u;
                    const bool active = q < total;
                    const bool is_lit = k < fl;
                    const uint32_t roundpos = base + r0;
                    const uint32_t P = base + q - o;
                    const bool in_step = active && !is_lit && P >= roundpos;
                    uint32_t val = 0;
                    if (active && is_lit && inl_s) val = (x_s >> (8u * (k & 3u))) & 0xffu;
                    if (active && is_lit && !inl_s) val = stage[lip0 - sbase + (rl_s >> 12) + k];
                    if (DEC_RING < 65536u) {
                        const bool far = active && !is_lit && o > DEC_RING;
                        if (ballot(far)) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            if (far) val = __hip_atomic_load(out + frame_out + P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (active && !is_lit && !in_step && !far) val = ring[P & (DEC_RING - 1)];
                    } else {
                        if (active && !is_lit && !in_step) val = ring[P & (DEC_RING - 1)];
                    }
                    bool has = !in_step;
                    uint32_t dep = in_step ? P - roundpos : (uint32_t)lane;
                    while (ballot(!has)) {
                        const uint32_t g = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(dep * 4u), (int)((has ? 0x80000000u : 0u) | (dep << 8) | val));
                        if (!has) {
                            if (g >> 31) { val = g & 0xffu; has = true; }
                            else dep = (g >> 8) & 63u;
                        }
                    }
                    if (active) ring[(roundpos + (uint32_t)lane) & (DEC_RING - 1)] = (uint8_t)val;
                    wave_lds_sync();
                    pos = roundpos + (total - r0 < 64u ? total - r0 : 64u);
                    flush(false);
                    SQY_DST(++ct_rounds;)
                }
                SQY_DST(++ct_runits; ct_cr += __builtin_amdgcn_s_memtime() - u0;)
            } else {
                // one sequence after the other
                for (uint32_t i = 0; i < m && !bad; ++i) {
                    const uint32_t l = lane_read(lit, i), lp = lane_read(lip, i), o = lane_read(off, i), mln = lane_read(mlen, i);
                    if (l && lane_read(inl, i)) {
                        // the literals out of the record: a write, no read (and the match's read behind it sees it: one wave's LDS
                        // operations keep their order)
                        if ((uint32_t)lane < l) ring[(pos + lane) & (DEC_RING - 1)] = (uint8_t)(lp >> (8u * ((uint32_t)lane & 3u)));
                        wave_lds_sync();
                        pos += l;
                        if (!mln) flush(false);
                    } else if (l) {
                        if (l <= DEC_IN - 64u) {
                            need(lp, l);
                            for (uint32_t j = 0; j < l;) {
                                const uint32_t dp = pos & (DEC_RING - 1);
                                uint32_t cnt = l - j < 1024u ? l - j : 1024u;
                                cnt = cnt < DEC_RING - dp ? cnt : DEC_RING - dp;
                                wide_copy(stage + (lp - sbase + j), dp, cnt);
                                pos += cnt;
                                j += cnt;
                                flush(false);
                            }
                        } else {
                            for (uint32_t j = 0; j < l; j += 64) {                // a long literal run: stream -> ring
                                const uint32_t cnt = l - j < 64 ? l - j : 64;
                                if ((uint32_t)lane < cnt) ring[(pos + lane) & (DEC_RING - 1)] = src[lp + j + lane];
                                wave_lds_sync();
                                pos += cnt;
                                flush(false);
                            }
                        }
                    }
                    if (mln) copy_match(o, mln);
                }
                SQY_DST(ct_seqs += m; ct_ci += __builtin_amdgcn_s_memtime() - u0;)
            }
        }
        if (flags & 1u) break;
    }
    flush(true);
    if ((uint64_t)pos != expect) bad = true;
    if (bad) {
        if (lane == 0) { ctrl[2] = 1u; atomicExch(errflag, 1u); }
    }
    SQY_DST(if (lane == 0 && frame_out + 512 <= out_bytes) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint64_t* cs = reinterpret_cast<uint64_t*>(out + frame_out) + 16;
        cs[0] = __builtin_amdgcn_s_memtime() - ct_t0; cs[1] = ct_units; cs[2] = ct_wait; cs[3] = ct_runits; cs[4] = ct_rounds; cs[5] = ct_cr; cs[6] = ct_seqs; cs[7] = ct_ci;
        cs[8] = ct_fill; cs[9] = ct_nflush; cs[10] = ct_flush; cs[11] = ct_short; cs[12] = ct_cshort; cs[13] = ct_long; cs[14] = ct_clong; cs[15] = ct_far;
        cs[16] = ct_cfar;
    })
}

template <uint32_t DEC_RING>
__global__ __launch_bounds__(128)
void lz4_frames_decode2_kernel(const uint8_t* __restrict__ in, const uint4* __restrict__ blk, const uint32_t* __restrict__ frame_first,
                               uint8_t* __restrict__ out, uint64_t out_bytes, uint64_t frame_stride, uint64_t block_bytes,
                               uint32_t* __restrict__ errflag, const uint64_t* __restrict__ remap, uint64_t remap_bytes)
{
    constexpr uint32_t DEC_IN = 3072u;
    __shared__ __attribute__((aligned(16))) uint8_t d2_raw[DEC_RING + DEC_IN + DEC2_PIN + 64 + 2 * DEC2_UNIT * 16 + 32 + 32];
    lds_u8* ring = (lds_u8*)d2_raw;
    lds_u8* stage = (lds_u8*)d2_raw + DEC_RING;                                     // wave 1: the literals come from here
    lds_u8* pstage = (lds_u8*)d2_raw + DEC_RING + DEC_IN;                           // wave 0: tokens, lengths, offsets
    lds_u8* owner_mark = (lds_u8*)d2_raw + DEC_RING + DEC_IN + DEC2_PIN + 32;
    SQY_LDS uint4* units = reinterpret_cast<SQY_LDS uint4*>((lds_u8*)d2_raw + DEC_RING + DEC_IN + DEC2_PIN + 32 + 64);
    volatile SQY_LDS uint32_t* ctrl = reinterpret_cast<volatile SQY_LDS uint32_t*>((lds_u8*)d2_raw + DEC_RING + DEC_IN + DEC2_PIN + 32 + 64 + 2 * DEC2_UNIT * 16);
    // ctrl[0] units published, [1] units taken, [2] stop, [4 + s] slot s: sequences | flags << 8 (1 = the frame's last unit, 2 = damaged)
    const int lane = threadIdx.x & 63;
    const uint32_t role = sgpr(threadIdx.x >> 6);
    const uint32_t f = blockIdx.x;
    const uint32_t b0 = frame_first[f], b1 = frame_first[f + 1];
    if (lz4_decode_frame_struck((uint64_t)f * frame_stride, remap, remap_bytes)) return;      // (the whole workgroup: f is its frame)
    if (threadIdx.x < 8) ctrl[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t frame_out = lz4_decode_frame_out((uint64_t)f * frame_stride, remap, remap_bytes);
    const uint64_t room = frame_out < out_bytes ? out_bytes - frame_out : 0;
    const uint64_t expect = remap ? frame_stride : (gridDim.x == 1 || room < frame_stride) ? room : frame_stride;
    if (b1 - b0 != 1) {                                       // (the host sends such streams to the other kernel)
        if (threadIdx.x == 0) atomicExch(errflag, 1u);
        return;
    }
    const uint4 e = blk[b0];
    const uint8_t* __restrict__ src = in + (((uint64_t)e.y << 32) | e.x);
    const uint32_t sz = e.z & 0x7fffffffu;
    if (e.z >> 31) {
        // a stored frame: lz4_stored_frames_copy_kernel moves it, this kernel only vouches for the bounds
        if (threadIdx.x == 0 && (frame_out + sz > out_bytes || (uint64_t)sz != expect || (uint64_t)sz > block_bytes)) atomicExch(errflag, 1u);
        return;
    }
    uint64_t* stats = nullptr;
    SQY_DST(if (frame_out + 512 <= out_bytes) stats = reinterpret_cast<uint64_t*>(out + frame_out);)
    if (role == 0) lz4_decode2_parse(src, sz, pstage, units, ctrl, lane, stats);
    else lz4_decode2_copy<DEC_RING>(src, sz, out, frame_out, out_bytes, block_bytes, expect, ring, stage, owner_mark, units, ctrl, errflag, lane);
}

// stored (uncompressed) blocks of single-block frames: plain copy stream -> output, one workgroup per 32 KiB slice
constexpr uint32_t DEC_COPY_SLICE = 32768;

__global__ __launch_bounds__(256)
void lz4_stored_frames_copy_kernel(const uint8_t* __restrict__ in, const uint4* __restrict__ blk, const uint32_t* __restrict__ frame_first,
                                   uint8_t* __restrict__ out, uint64_t out_bytes, uint64_t frame_stride, uint32_t slices_per_frame,
                                   const uint64_t* __restrict__ remap, uint64_t remap_bytes)
{
    const uint32_t f = blockIdx.x / slices_per_frame, slice = blockIdx.x % slices_per_frame;
    const uint32_t b0 = frame_first[f];
    if (frame_first[f + 1] - b0 != 1) return;
    const uint4 e = blk[b0];
    if (!(e.z >> 31)) return;
    const uint32_t sz = e.z & 0x7fffffffu;
    if (lz4_decode_frame_struck((uint64_t)f * frame_stride, remap, remap_bytes)) return;
    const uint64_t o = lz4_decode_frame_out((uint64_t)f * frame_stride, remap, remap_bytes);
    if (o + sz > out_bytes) return;                              // the decode kernel reports it
    const uint32_t begin = slice * DEC_COPY_SLICE;
    if (begin >= sz) return;
    const uint32_t len0 = sz - begin < DEC_COPY_SLICE ? sz - begin : DEC_COPY_SLICE;
    const uint8_t* __restrict__ s = in + (((uint64_t)e.y << 32) | e.x) + begin;
    uint8_t* __restrict__ d = out + o + begin;
    const int tid = threadIdx.x;
    const uint32_t head0 = (uint32_t)((16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15);
    const uint32_t head = head0 < len0 ? head0 : len0;
    if ((uint32_t)tid < head) d[tid] = s[tid];
    const uint32_t nvec = (len0 - head) >> 4;
    for (uint32_t i0 = tid; i0 < nvec; i0 += 1024) {             // four loads in flight per thread before the first store
        uint4 v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) if (i0 + j * 256u < nvec) v[j] = ld_u128(s + head + (size_t)(i0 + j * 256u) * 16);
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) if (i0 + j * 256u < nvec) *reinterpret_cast<uint4*>(d + head + (size_t)(i0 + j * 256u) * 16) = v[j];
    }
    const uint32_t done = head + (nvec << 4);
    if ((uint32_t)tid < len0 - done) d[done + tid] = s[done + tid];
}

// ---- ONE block-linked frame (the serial layout, nthreads = 1) decoded block-parallel (round 4) --------------------------------------
// A block of a linked frame may copy from the 64 KiB decoded in front of it, so rounds 2-3 decoded the frame with one wavefront
// (0.34 GB/s on the 1 GiB bench stack).  The dependence is shallow, though: every byte of a block is a literal of the block or equals
// ONE byte of the history in front of the block -- whatever chain of copies inside the block leads there.  So:
//   1. lz4_blocks_decode_sym_kernel, one wavefront per compressed block, all blocks at once: decodes the block with the history as
//      an unknown -- next to every output byte it keeps a 16-bit reference (0: the byte is known; d: the byte equals the byte d in
//      front of the block's first byte), copied along with the bytes by every match.  The history itself enters as bytes with
//      reference = their distance.  Output: the block's bytes (unknown ones as 0) and its references (`refs`, 2 bytes per byte).
//      Stored blocks are plain copies (lz4_linked_stored_copy_kernel).
//   2. lz4_linked_resolve_tails_kernel, one workgroup walking the blocks in order: only the last 64 KiB of block k-1 can be named by
//      block k, so the chain that has to be followed in order is short -- the tail of block k is resolved from the resolved tail of
//      block k-1, kept in LDS (64 LDS gathers per thread and block).
//   3. lz4_linked_resolve_bodies_kernel, every block at once: the remaining bytes that carry a reference take the byte out of the
//      (now final) tail in front of their block.
// Every block but the last must decode to exactly block_bytes (LZ4F without autoFlush only cuts full blocks): checked; a stream that
// is built differently, or is damaged, raises the flag and the host falls back to the one-wavefront walk (which owns the error codes).
constexpr uint32_t SYM_RING = 16384, SYM_IN = 3072, SYM_TAIL = 65536;
constexpr uint32_t SYM_RANGE = 64;                    // blocks per range of the tails' scan

__global__ __launch_bounds__(256)
void lz4_linked_stored_copy_kernel(const uint8_t* __restrict__ in, const uint4* __restrict__ blk, uint8_t* __restrict__ out, uint64_t out_bytes,
                                   uint64_t block_bytes, uint32_t slices_per_block, uint32_t* __restrict__ errflag)
{
    const uint32_t k = blockIdx.x / slices_per_block, slice = blockIdx.x % slices_per_block;
    const uint4 e = blk[k];
    if (!(e.z >> 31)) return;
    const uint32_t sz = e.z & 0x7fffffffu;
    const uint64_t o = (uint64_t)k * block_bytes;
    const uint64_t expect = o < out_bytes ? (out_bytes - o < block_bytes ? out_bytes - o : block_bytes) : 0;
    if ((uint64_t)sz != expect) { if (slice == 0 && threadIdx.x == 0) atomicExch(errflag, 1u); return; }
    const uint32_t begin = slice * DEC_COPY_SLICE;
    if (begin >= sz) return;
    const uint32_t len0 = sz - begin < DEC_COPY_SLICE ? sz - begin : DEC_COPY_SLICE;
    const uint8_t* __restrict__ s = in + (((uint64_t)e.y << 32) | e.x) + begin;
    uint8_t* __restrict__ d = out + o + begin;
    const int tid = threadIdx.x;
    const uint32_t head0 = (uint32_t)((16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15);
    const uint32_t head = head0 < len0 ? head0 : len0;
    if ((uint32_t)tid < head) d[tid] = s[tid];
    const uint32_t nvec = (len0 - head) >> 4;
    for (uint32_t i0 = tid; i0 < nvec; i0 += 1024) {
        uint4 v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) if (i0 + j * 256u < nvec) v[j] = ld_u128(s + head + (size_t)(i0 + j * 256u) * 16);
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) if (i0 + j * 256u < nvec) *reinterpret_cast<uint4*>(d + head + (size_t)(i0 + j * 256u) * 16) = v[j];
    }
    const uint32_t done = head + (nvec << 4);
    if ((uint32_t)tid < len0 - done) d[done + tid] = s[done + tid];
}

__global__ __launch_bounds__(64)
void lz4_blocks_decode_sym_kernel(const uint8_t* __restrict__ in, const uint4* __restrict__ blk, uint8_t* __restrict__ out,
                                  uint16_t* __restrict__ refs, uint64_t out_bytes, uint64_t block_bytes, uint32_t* __restrict__ errflag)
{
    __shared__ __attribute__((aligned(16))) uint8_t sym_raw[SYM_RING + 2 * SYM_RING + SYM_IN + 64];
    lds_u8* ring = (lds_u8*)sym_raw;                                            // the last 16 KiB of the block's output ..
    SQY_LDS uint16_t* rref = (SQY_LDS uint16_t*)((lds_u8*)sym_raw + SYM_RING);  // .. and their references
    lds_u8* stage = (lds_u8*)sym_raw + 3 * SYM_RING;
    constexpr uint32_t MASK = SYM_RING - 1;
    const int lane = threadIdx.x;
    const uint32_t k = blockIdx.x;
    const uint4 e = blk[k];
    if (e.z >> 31) return;                                                      // stored: lz4_linked_stored_copy_kernel
    const uint8_t* __restrict__ src = in + (((uint64_t)e.y << 32) | e.x);
    const uint32_t sz = e.z & 0x7fffffffu;
    const uint64_t base = (uint64_t)k * block_bytes;
    if (base >= out_bytes) { if (lane == 0) atomicExch(errflag, 1u); return; }
    const uint32_t expect = (uint32_t)(out_bytes - base < block_bytes ? out_bytes - base : block_bytes);
    uint8_t* __restrict__ dout = out + base;
    uint16_t* __restrict__ dref = refs + base;
    // the history in front of the block, as far as the ring reaches: unknown bytes that refer to themselves (block 0 has none:
    // its matches cannot reach in front of it, checked below)
    for (uint32_t i = lane; i < SYM_RING; i += 64) { ring[i] = 0; rref[i] = (uint16_t)(k ? SYM_RING - i : 0u); }
    wave_lds_sync();
    const uint32_t reach = k ? 65535u : 0u;                                     // how far in front of the block a match may begin
    uint32_t pos = 0, flushed = 0, ip = 0;
    bool bad = false;

    auto flush = [&](bool all) {
        while (flushed + 1024u <= pos) {                                        // whole KiB pieces: bytes and references
            const uint32_t ri = (flushed & MASK) + (uint32_t)lane * 16u;
            const v4u d = *reinterpret_cast<const SQY_LDS v4u*>(ring + ri);
            const v4u r0 = *reinterpret_cast<const SQY_LDS v4u*>(rref + ri), r1 = *reinterpret_cast<const SQY_LDS v4u*>(rref + ri + 8u);
            *reinterpret_cast<v4u_any*>(dout + flushed + (uint32_t)lane * 16u) = d;
            *reinterpret_cast<v4u_any*>(dref + flushed + (uint32_t)lane * 16u) = r0;
            *reinterpret_cast<v4u_any*>(dref + flushed + (uint32_t)lane * 16u + 8u) = r1;
            flushed += 1024u;
        }
        if (all) {
            for (uint32_t i = flushed + (uint32_t)lane; i < pos; i += 64u) { dout[i] = ring[i & MASK]; dref[i] = rref[i & MASK]; }
            flushed = pos;
        }
    };
    uint32_t sbase = 0, shi = 0;
    auto fill = [&](uint32_t at) {
        sbase = at & ~15u;
#pragma unroll
        for (uint32_t j = 0; j < SYM_IN / 1024; ++j) {
            const uint32_t a = sbase + j * 1024u + (uint32_t)lane * 16u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a + 16u <= sz) v = ld_u128(src + a);
            else if (a < sz) {
                uint32_t wv[4] = {0, 0, 0, 0};
                for (uint32_t q = 0; a + q < sz; ++q) wv[q >> 2] |= (uint32_t)src[a + q] << (8u * (q & 3u));
                v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            }
            const v4u vv = {v.x, v.y, v.z, v.w};
            *reinterpret_cast<SQY_LDS v4u*>(stage + j * 1024u + (uint32_t)lane * 16u) = vv;
        }
        wave_lds_sync();
        shi = sbase + SYM_IN < sz ? sbase + SYM_IN : sz;
    };
    auto need = [&](uint32_t at, uint32_t cnt) { if (at < sbase || at + cnt > shi) fill(at); };   // cnt <= SYM_IN - 16, at + cnt <= sz
    auto sbyte_at = [&](uint32_t at) -> uint32_t { return stage[at - sbase]; };
    // a run of length-extension bytes from `at` on, 64 bytes per LDS round trip (round 5; see ext_run in lz4_frames_decode_kernel)
    auto ext_run = [&](uint32_t& at, uint32_t& acc) -> bool {
        for (;;) {
            if (at >= sz) return false;
            const uint32_t cnt = sz - at < 64u ? sz - at : 64u;
            need(at, cnt);
            const uint32_t bv = (uint32_t)lane < cnt ? (uint32_t)stage[at - sbase + (uint32_t)lane] : 0u;
            const uint64_t closing = ballot((uint32_t)lane < cnt && bv != 255u);
            if (closing) {
                const uint32_t k = ctz64(closing);
                acc += 255u * k + lane_read(bv, k);
                at += k + 1u;
                return true;
            }
            acc += 255u * cnt;
            at += cnt;
        }
    };
    // 16 window bytes at block offset `at` (inside the stage) as four scalars
    auto window = [&](uint32_t at) -> uint4 {
        const uint32_t a = at - sbase, sh = a & 3u;
        const uint4 d = lds_ld_4dw(stage + (a & ~3u));
        const uint32_t d4 = *reinterpret_cast<const volatile SQY_LDS uint32_t*>(stage + (a & ~3u) + 16u);
        return make_uint4(sgpr(__builtin_amdgcn_alignbyte(d.y, d.x, sh)), sgpr(__builtin_amdgcn_alignbyte(d.z, d.y, sh)),
                          sgpr(__builtin_amdgcn_alignbyte(d.w, d.z, sh)), sgpr(__builtin_amdgcn_alignbyte(d4, d.w, sh)));
    };
    // cnt <= 1024 bytes with their references, LDS -> ring at dp (no wrap on either side); from_ref == nullptr: literals (reference 0)
    auto wide_copy = [&](const lds_u8* from, const SQY_LDS uint16_t* from_ref, uint32_t dp, uint32_t cnt) {
        const uint32_t full = cnt >> 4, r = cnt & 15u;
        const uint32_t l16 = (uint32_t)lane * 16u;
        v4u_any v = {0, 0, 0, 0}, ra = {0, 0, 0, 0}, rb = {0, 0, 0, 0};
        uint32_t t = 0, tr = 0;
        if ((uint32_t)lane < full) {
            v = *reinterpret_cast<const SQY_LDS v4u_any*>(from + l16);
            if (from_ref) { ra = *reinterpret_cast<const SQY_LDS v4u_any*>(from_ref + l16); rb = *reinterpret_cast<const SQY_LDS v4u_any*>(from_ref + l16 + 8u); }
        }
        if ((uint32_t)lane < r) { t = from[full * 16u + (uint32_t)lane]; if (from_ref) tr = from_ref[full * 16u + (uint32_t)lane]; }
        if ((uint32_t)lane < full) {
            *reinterpret_cast<SQY_LDS v4u_any*>(ring + dp + l16) = v;
            *reinterpret_cast<SQY_LDS v4u_any*>(rref + dp + l16) = ra;
            *reinterpret_cast<SQY_LDS v4u_any*>(rref + dp + l16 + 8u) = rb;
        }
        if ((uint32_t)lane < r) { ring[dp + full * 16u + (uint32_t)lane] = (uint8_t)t; rref[dp + full * 16u + (uint32_t)lane] = (uint16_t)tr; }
        wave_lds_sync();
    };
    auto copy_literals = [&](uint32_t at, uint32_t lit) {                       // stage -> ring; [at, at + lit) is in the stage
        for (uint32_t i = 0; i < lit;) {
            const uint32_t dp = pos & MASK;
            uint32_t cnt = lit - i < 1024u ? lit - i : 1024u;
            cnt = cnt < SYM_RING - dp ? cnt : SYM_RING - dp;
            wide_copy(stage + (at - sbase + i), nullptr, dp, cnt);
            pos += cnt;
            i += cnt;
            flush(false);
        }
    };
    auto copy_match = [&](uint32_t offset, uint32_t ml) {
        uint32_t rem = ml;
        if (offset > SYM_RING) {
            // behind the ring: the block's own output, flushed at least 15 KiB ago (bytes and references, past this CU's L1) -- or
            // the history in front of the block, where a byte is nothing but its distance
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            while (rem) {
                const uint32_t dp = pos & MASK;
                uint32_t cnt = rem < 1024u ? rem : 1024u;
                cnt = cnt < SYM_RING - dp ? cnt : SYM_RING - dp;
                const uint32_t full = cnt >> 4, r = cnt & 15u;
                const int32_t q0 = (int32_t)(pos - offset) + lane * 16;         // block-relative position of this lane's first source byte
                if ((uint32_t)lane < full) {
                    uint32_t dw[4] = {0, 0, 0, 0}, rw[8];
                    if (q0 + 15 >= 0) {                                          // (may begin up to 15 bytes in front of the block: patched below)
                        const uint4 a = ld_u128_agent(dout + q0);
                        const uint4 b0 = ld_u128_agent(reinterpret_cast<const uint8_t*>(dref + q0)), b1 = ld_u128_agent(reinterpret_cast<const uint8_t*>(dref + q0 + 8));
                        dw[0] = a.x; dw[1] = a.y; dw[2] = a.z; dw[3] = a.w;
                        rw[0] = b0.x; rw[1] = b0.y; rw[2] = b0.z; rw[3] = b0.w; rw[4] = b1.x; rw[5] = b1.y; rw[6] = b1.z; rw[7] = b1.w;
                    }
                    if (q0 < 0) {
                        const uint32_t nneg = (uint32_t)(-q0) < 16u ? (uint32_t)(-q0) : 16u;      // leading bytes that lie in front of the block
#pragma unroll
                        for (uint32_t m = 0; m < 4; ++m) {
                            const uint32_t lo = 4u * m;
                            dw[m] = nneg >= lo + 4u ? 0u : (nneg <= lo ? dw[m] : dw[m] & (0xffffffffu << (8u * (nneg - lo))));
                        }
#pragma unroll
                        for (uint32_t m = 0; m < 8; ++m) {
                            const uint32_t j0 = 2u * m, j1 = 2u * m + 1u;
                            const uint32_t a0 = j0 < nneg ? (uint32_t)(-(q0 + (int32_t)j0)) : (rw[m] & 0xffffu);
                            const uint32_t a1 = j1 < nneg ? (uint32_t)(-(q0 + (int32_t)j1)) : (rw[m] >> 16);
                            rw[m] = a0 | (a1 << 16);
                        }
                    }
                    const v4u_any dv = {dw[0], dw[1], dw[2], dw[3]}, ra = {rw[0], rw[1], rw[2], rw[3]}, rb = {rw[4], rw[5], rw[6], rw[7]};
                    *reinterpret_cast<SQY_LDS v4u_any*>(ring + dp + (uint32_t)lane * 16u) = dv;
                    *reinterpret_cast<SQY_LDS v4u_any*>(rref + dp + (uint32_t)lane * 16u) = ra;
                    *reinterpret_cast<SQY_LDS v4u_any*>(rref + dp + (uint32_t)lane * 16u + 8u) = rb;
                }
                if ((uint32_t)lane < r) {
                    const int32_t q = (int32_t)(pos - offset) + (int32_t)(full * 16u) + lane;
                    uint32_t t = 0, tr;
                    if (q < 0) tr = (uint32_t)(-q);
                    else {
                        t = __hip_atomic_load(dout + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        tr = __hip_atomic_load(dref + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    ring[dp + full * 16u + (uint32_t)lane] = (uint8_t)t;
                    rref[dp + full * 16u + (uint32_t)lane] = (uint16_t)tr;
                }
                wave_lds_sync();
                pos += cnt;
                rem -= cnt;
                flush(false);
                if (rem && (ml - rem) + 3072u > offset) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the next source is what this match wrote itself)
            }
            return;
        }
        // inside the ring (bytes and references alike; the history's last 16 KiB sit there as references to themselves)
        if (ml <= 64u) {
            const uint32_t lm = (offset >= 64u || offset >= ml) ? (uint32_t)lane : (uint32_t)lane % offset;
            uint32_t v = 0, rv = 0;
            if ((uint32_t)lane < ml) { v = ring[(pos - offset + lm) & MASK]; rv = rref[(pos - offset + lm) & MASK]; }
            if ((uint32_t)lane < ml) { ring[(pos + lane) & MASK] = (uint8_t)v; rref[(pos + lane) & MASK] = (uint16_t)rv; }
            wave_lds_sync();
            pos += ml;
            flush(false);
            return;
        }
        uint32_t period = offset;
        if (offset < 64u) {
            const uint32_t lm = (uint32_t)lane % offset;
            const uint32_t v = ring[(pos - offset + lm) & MASK], rv = rref[(pos - offset + lm) & MASK];
            ring[(pos + lane) & MASK] = (uint8_t)v; rref[(pos + lane) & MASK] = (uint16_t)rv;
            wave_lds_sync();
            pos += 64u;
            rem -= 64u;
            flush(false);
            period = ((64u + offset) / offset) * offset;
        }
        while (rem) {
            const uint32_t sp = (pos - period) & MASK, dp = pos & MASK;
            uint32_t cnt = rem < 1024u ? rem : 1024u;
            cnt = cnt < period ? cnt : period;
            cnt = cnt < SYM_RING - sp ? cnt : SYM_RING - sp;
            cnt = cnt < SYM_RING - dp ? cnt : SYM_RING - dp;
            wide_copy(ring + sp, rref + sp, dp, cnt);
            pos += cnt;
            rem -= cnt;
            flush(false);
            if (cnt == period && period < 1024u) period <<= 1;
        }
    };

    while (ip < sz && !bad) {
        // token and literal-length bytes out of one window, the offset and match-length bytes out of a second one behind the
        // literals (the block's last bytes, where a window would run past the block: byte by byte)
        uint32_t token, lit, used = 1u;
        bool lit_ok;
        if (ip + 16u <= sz) {
            need(ip, 16);
            const uint4 hw = window(ip);
            const uint64_t lo = ((uint64_t)hw.y << 32) | hw.x, hi = ((uint64_t)hw.w << 32) | hw.z;
            token = hw.x & 0xffu;
            lit = token >> 4;
            lit_ok = lit < 15u;
            while (!lit_ok && used < 16u) {
                const uint32_t sb = used < 8u ? (uint32_t)(lo >> (8u * used)) & 0xffu : (uint32_t)(hi >> (8u * (used - 8u))) & 0xffu;
                ++used;
                lit += sb;
                lit_ok = sb != 255u;
            }
        } else {
            need(ip, 1);
            token = sbyte_at(ip);
            lit = token >> 4;
            lit_ok = lit < 15u;
        }
        uint32_t ipl = ip + used;
        if (!lit_ok && !ext_run(ipl, lit)) bad = true;
        if (bad || ipl + lit > sz || pos + lit > expect) { bad = true; break; }
        const bool second_window = lit <= SYM_IN - 128u && ipl + lit + 16u <= sz;
        uint4 xw = make_uint4(0, 0, 0, 0);
        if (second_window) {
            need(ip, (ipl - ip) + lit + 16u);                                   // (ipl - ip <= 16 whenever lit is this small)
            xw = window(ipl + lit);
            copy_literals(ipl, lit);
        } else if (lit <= SYM_IN - 64u) {
            if (lit) { need(ipl, lit); copy_literals(ipl, lit); }
        } else {
            for (uint32_t i = 0; i < lit; i += 64) {                            // a literal run longer than the stage: stream -> ring
                const uint32_t cnt = lit - i < 64 ? lit - i : 64;
                if ((uint32_t)lane < cnt) { ring[(pos + lane) & MASK] = src[ipl + i + lane]; rref[(pos + lane) & MASK] = 0; }
                wave_lds_sync();
                pos += cnt;
                flush(false);
            }
        }
        ip = ipl + lit;
        if (ip >= sz) break;                                                    // the block's last sequence: literals only
        uint32_t offset, ml = token & 15u;
        bool ml_ok = ml < 15u;
        if (second_window) {
            const uint64_t lo = ((uint64_t)xw.y << 32) | xw.x, hi = ((uint64_t)xw.w << 32) | xw.z;
            offset = xw.x & 0xffffu;
            uint32_t used2 = 2u;
            while (!ml_ok && used2 < 16u) {
                const uint32_t sb = used2 < 8u ? (uint32_t)(lo >> (8u * used2)) & 0xffu : (uint32_t)(hi >> (8u * (used2 - 8u))) & 0xffu;
                ++used2;
                ml += sb;
                ml_ok = sb != 255u;
            }
            ip += used2;
        } else {
            if (ip + 2u > sz) { bad = true; break; }
            need(ip, 2);
            offset = sbyte_at(ip) | (sbyte_at(ip + 1u) << 8);
            ip += 2u;
        }
        if (!ml_ok && !ext_run(ip, ml)) bad = true;
        ml += 4u;
        if (bad || offset == 0u || offset > pos + reach || pos + ml > expect) { bad = true; break; }
        copy_match(offset, ml);
    }
    flush(true);
    if (pos != expect) bad = true;                                              // (every block of the frame but the last is a full block)
    if (bad && lane == 0) atomicExch(errflag, 1u);
}

// the tails, in order: one workgroup; LDS holds the resolved last 64 KiB of the block in front.  A thread owns 64 consecutive bytes;
// the bytes and references of the NEXT block's tail are fetched while the block in hand is resolved.  Four bytes at a time: references
// that count down (a straight copy out of the history: four consecutive bytes, one LDS read) or are all equal (a run), else byte by byte.
// (range_len != 0: workgroup w walks the blocks [w * range_len, (w + 1) * range_len) only, and starts from the resolved tail in front of
// its range, range_start + w * 64 KiB -- what lz4_linked_chain_ranges_kernel left there; workgroup 0 needs none)
__global__ __launch_bounds__(1024)
void lz4_linked_resolve_tails_kernel(const uint4* __restrict__ blk, uint32_t nblocks, uint8_t* __restrict__ out, const uint16_t* __restrict__ refs,
                                     uint64_t out_bytes, uint64_t block_bytes, uint32_t range_len, const uint8_t* __restrict__ range_start)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tails_lds[];                 // 2 x 64 KiB
    SQY_LDS uint8_t* prev = (SQY_LDS uint8_t*)tails_lds;
    SQY_LDS uint8_t* cur = (SQY_LDS uint8_t*)tails_lds + SYM_TAIL;
    __shared__ uint32_t rawbits[1024];                                                  // bit k: block k is stored (nblocks <= 32768, host)
    const uint32_t t = threadIdx.x;
    {
        uint32_t bits = 0;
        for (uint32_t j = 0; j < 32u; ++j) { const uint32_t b = t * 32u + j; if (b < nblocks && (blk[b].z >> 31)) bits |= 1u << j; }
        rawbits[t] = bits;
    }
    const uint32_t k_first = range_len ? blockIdx.x * range_len : 0u;
    const uint32_t k_end = range_len && (k_first + range_len) < nblocks - 1u ? k_first + range_len : nblocks - 1u;   // tails of [k_first, k_end)
    if (k_first && range_start) {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint4 v = ld_u128(range_start + (uint64_t)blockIdx.x * SYM_TAIL + t * 64u + j * 16u);
            const v4u vv = {v.x, v.y, v.z, v.w};
            *reinterpret_cast<SQY_LDS v4u*>(prev + t * 64u + j * 16u) = vv;
        }
    }
    __syncthreads();
    auto is_raw = [&](uint32_t b) -> bool { return ((rawbits[b >> 5] >> (b & 31u)) & 1u) != 0u; };
    // what block k asks of this thread: 0 nothing (stored, next one stored too), 1 bytes only (known bytes the next block will look at),
    // 2 bytes and references (a compressed block behind the first)
    auto wants = [&](uint32_t b) -> uint32_t {
        if (b + 1 >= nblocks) return 0u;
        if (is_raw(b) || b == 0) return is_raw(b + 1) ? 0u : 1u;
        return 2u;
    };
    uint32_t d[16], rr[32];
    auto fetch = [&](uint32_t b, uint32_t what) {
        if (!what) return;
        const uint64_t tb = (uint64_t)(b + 1) * block_bytes - SYM_TAIL;                 // first byte of the tail (blocks in front of the last are full)
        const uint8_t* o = out + tb + (uint64_t)t * 64u;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) { const uint4 v = ld_u128(o + j * 16u); d[4 * j] = v.x; d[4 * j + 1] = v.y; d[4 * j + 2] = v.z; d[4 * j + 3] = v.w; }
        if (what == 2u) {
            const uint16_t* r = refs + tb + (uint64_t)t * 64u;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) { const uint4 v = ld_u128(reinterpret_cast<const uint8_t*>(r + j * 8u)); rr[4 * j] = v.x; rr[4 * j + 1] = v.y; rr[4 * j + 2] = v.z; rr[4 * j + 3] = v.w; }
        }
    };
    uint32_t what = k_first < k_end ? wants(k_first) : 0u;
    fetch(k_first, what);
    for (uint32_t k = k_first; k < k_end; ++k) {
        // the block in hand, out of the registers
        uint32_t c[16];
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) c[j] = d[j];
        uint32_t any = 0;
        if (what == 2u) {
#pragma unroll
            for (uint32_t j = 0; j < 32; ++j) any |= rr[j];
        }
        uint32_t q[32];
#pragma unroll
        for (uint32_t j = 0; j < 32; ++j) q[j] = rr[j];
        const uint32_t what_now = what;
        // .. and the next one on its way
        what = k + 1 < k_end ? wants(k + 1) : 0u;
        fetch(k + 1, what);
        if (what_now == 2u && any) {
#pragma unroll
            for (uint32_t m = 0; m < 16; ++m) {
                const uint32_t a = q[2 * m], b = q[2 * m + 1];
                if (!(a | b)) continue;
                const uint32_t r0 = a & 0xffffu;
                if (r0 >= 4u && a == r0 * 0x10001u - 0x10000u && b == a - 0x20002u) c[m] = lds_ld_u32(prev + (SYM_TAIL - r0));
                else if (r0 && a == r0 * 0x10001u && b == a) c[m] = (uint32_t)prev[SYM_TAIL - r0] * 0x01010101u;
                else {
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const uint32_t rf = ((j < 2 ? a : b) >> (16u * (j & 1u))) & 0xffffu;
                        if (rf) c[m] = (c[m] & ~(0xffu << (8u * j))) | ((uint32_t)prev[SYM_TAIL - rf] << (8u * j));
                    }
                }
            }
            const uint64_t tb = (uint64_t)(k + 1) * block_bytes - SYM_TAIL;
            uint8_t* o = out + tb + (uint64_t)t * 64u;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) st_u128(o + j * 16u, make_uint4(c[4 * j], c[4 * j + 1], c[4 * j + 2], c[4 * j + 3]));
        }
        if (what_now && !is_raw(k + 1)) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const v4u v = {c[4 * j], c[4 * j + 1], c[4 * j + 2], c[4 * j + 3]};
                *reinterpret_cast<SQY_LDS v4u*>(cur + t * 64u + j * 16u) = v;
            }
        }
        __syncthreads();
        SQY_LDS uint8_t* sw = prev; prev = cur; cur = sw;
    }
}

// The walk over the tails as a scan (long frames).  What block k does to a tail is a map from the tail in front of it to its own
// (byte i = a literal, or the byte at distance d of the tail in front); maps compose.  So the blocks are cut into ranges:
//   lz4_linked_compose_tails_kernel, one workgroup per range, all ranges at once: composes the maps of the range's blocks, from the
//     identity, in LDS -- a symbolic tail: per byte a 16-bit word that is a literal or a distance into the tail IN FRONT OF THE RANGE,
//     and a bit that says which (128 + 8 KiB);
//   lz4_linked_chain_ranges_kernel, one workgroup: applies the composed maps in order -- the true tail in front of every range;
//   lz4_linked_resolve_tails_kernel with range_len: every range walks its blocks from its true start, all ranges at once.
// n / W + W steps in a row instead of n.
constexpr uint32_t SYM_MAP_BYTES = 2 * SYM_TAIL + SYM_TAIL / 8;                         // a composed map in memory: 64 Ki words, 64 Ki bits

__global__ __launch_bounds__(1024)
void lz4_linked_compose_tails_kernel(const uint4* __restrict__ blk, uint32_t nblocks, const uint8_t* __restrict__ out, const uint16_t* __restrict__ refs,
                                     uint64_t block_bytes, uint32_t range_len, uint8_t* __restrict__ maps)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t compose_lds[];
    SQY_LDS uint16_t* sv = (SQY_LDS uint16_t*)compose_lds;                              // word i of the symbolic tail
    SQY_LDS uint32_t* isref = (SQY_LDS uint32_t*)((SQY_LDS uint8_t*)compose_lds + 2 * SYM_TAIL);   // bit i: word i is a distance, not a literal
    const uint32_t t = threadIdx.x;
    const uint32_t k_first = blockIdx.x * range_len;
    const uint32_t k_end = (k_first + range_len) < nblocks - 1u ? k_first + range_len : nblocks - 1u;
    // the identity: byte i of the tail in front of the range is itself, SYM_TAIL - i in front of the range's first byte
    for (uint32_t j = 0; j < 64u; ++j) sv[t * 64u + j] = (uint16_t)(SYM_TAIL - (t * 64u + j));     // (word 0: 65536 -> 0; no reference can name it, distances stop at 65535)
    isref[2u * t] = 0xffffffffu; isref[2u * t + 1u] = 0xffffffffu;
    __syncthreads();
    for (uint32_t k = k_first; k < k_end; ++k) {
        const uint64_t tb = (uint64_t)(k + 1) * block_bytes - SYM_TAIL;
        const uint8_t* o = out + tb + (uint64_t)t * 64u;
        const bool raw = (blk[k].z >> 31) != 0u;
        uint32_t d[16], nv[32], nb[2] = {0u, 0u};
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) { const uint4 v = ld_u128(o + j * 16u); d[4 * j] = v.x; d[4 * j + 1] = v.y; d[4 * j + 2] = v.z; d[4 * j + 3] = v.w; }
        if (raw || k == 0) {
#pragma unroll
            for (uint32_t j = 0; j < 32; ++j) {
                const uint32_t b0 = (d[j >> 1] >> (16u * (j & 1u))) & 0xffu, b1 = (d[j >> 1] >> (16u * (j & 1u) + 8u)) & 0xffu;
                nv[j] = b0 | (b1 << 16);
            }
        } else {
            const uint16_t* r = refs + tb + (uint64_t)t * 64u;
            uint32_t rr[32];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) { const uint4 v = ld_u128(reinterpret_cast<const uint8_t*>(r + j * 8u)); rr[4 * j] = v.x; rr[4 * j + 1] = v.y; rr[4 * j + 2] = v.z; rr[4 * j + 3] = v.w; }
            // four bytes at a time: references that count down (a straight copy: four consecutive words of the symbolic tail, one 8-byte
            // read, and their four bits) or are all equal (a run), else byte by byte
#pragma unroll
            for (uint32_t m = 0; m < 16; ++m) {
                const uint32_t a = rr[2 * m], b = rr[2 * m + 1];
                const uint32_t r0 = a & 0xffffu;
                const uint32_t dm = d[m];
                if (!(a | b)) {
                    nv[2 * m] = (dm & 0xffu) | ((dm & 0xff00u) << 8);
                    nv[2 * m + 1] = ((dm >> 16) & 0xffu) | ((dm >> 24) << 16);
                } else if (r0 >= 4u && a == r0 * 0x10001u - 0x10000u && b == a - 0x20002u) {
                    const uint32_t idx = SYM_TAIL - r0;
                    const uint64_t w4 = lds_ld_u64((const lds_u8*)sv + 2u * idx);
                    nv[2 * m] = (uint32_t)w4; nv[2 * m + 1] = (uint32_t)(w4 >> 32);
                    const uint32_t sh = idx & 31u;
                    uint32_t f4 = isref[idx >> 5] >> sh;
                    if (sh > 28u) f4 |= isref[(idx >> 5) + 1u] << (32u - sh);
                    nb[m >> 3] |= (f4 & 0xfu) << ((4u * m) & 31u);
                } else if (r0 && a == r0 * 0x10001u && b == a) {
                    const uint32_t idx = SYM_TAIL - r0;
                    const uint32_t w = sv[idx], f = (isref[idx >> 5] >> (idx & 31u)) & 1u;
                    nv[2 * m] = w * 0x10001u; nv[2 * m + 1] = w * 0x10001u;
                    nb[m >> 3] |= (f ? 0xfu : 0u) << ((4u * m) & 31u);
                } else {
                    uint32_t w2[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const uint32_t rf = ((j < 2 ? a : b) >> (16u * (j & 1u))) & 0xffffu;
                        uint32_t w = (dm >> (8u * j)) & 0xffu, f = 0u;
                        if (rf) {
                            const uint32_t idx = SYM_TAIL - rf;
                            w = sv[idx];
                            f = (isref[idx >> 5] >> (idx & 31u)) & 1u;
                        }
                        w2[j] = w;
                        nb[m >> 3] |= f << ((4u * m + j) & 31u);
                    }
                    nv[2 * m] = w2[0] | (w2[1] << 16); nv[2 * m + 1] = w2[2] | (w2[3] << 16);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const v4u v = {nv[4 * j], nv[4 * j + 1], nv[4 * j + 2], nv[4 * j + 3]};
            *reinterpret_cast<SQY_LDS v4u*>(sv + t * 64u + j * 8u) = v;
        }
        isref[2u * t] = nb[0]; isref[2u * t + 1u] = nb[1];
        __syncthreads();
    }
    uint8_t* m = maps + (uint64_t)blockIdx.x * SYM_MAP_BYTES;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        const v4u v = *reinterpret_cast<const SQY_LDS v4u*>(sv + t * 64u + j * 8u);
        st_u128(m + (uint64_t)t * 128u + j * 16u, make_uint4(v.x, v.y, v.z, v.w));
    }
    reinterpret_cast<uint32_t*>(m + 2 * SYM_TAIL)[2u * t] = isref[2u * t];
    reinterpret_cast<uint32_t*>(m + 2 * SYM_TAIL)[2u * t + 1u] = isref[2u * t + 1u];
}

// starts[w] (64 KiB each) = the true tail in front of range w, for w = 1 .. nranges - 1 (nothing lies in front of range 0)
__global__ __launch_bounds__(1024)
void lz4_linked_chain_ranges_kernel(const uint8_t* __restrict__ maps, uint32_t nranges, uint8_t* __restrict__ starts)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t chain_lds[];                 // 2 x 64 KiB
    SQY_LDS uint8_t* prev = (SQY_LDS uint8_t*)chain_lds;
    SQY_LDS uint8_t* cur = (SQY_LDS uint8_t*)chain_lds + SYM_TAIL;
    const uint32_t t = threadIdx.x;
    for (uint32_t j = 0; j < 16u; ++j) reinterpret_cast<SQY_LDS uint32_t*>(prev)[t * 16u + j] = 0u;
    __syncthreads();
    for (uint32_t w = 0; w + 1 < nranges; ++w) {
        const uint8_t* m = maps + (uint64_t)w * SYM_MAP_BYTES;
        uint32_t wv[32];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) { const uint4 v = ld_u128(m + (uint64_t)t * 128u + j * 16u); wv[4 * j] = v.x; wv[4 * j + 1] = v.y; wv[4 * j + 2] = v.z; wv[4 * j + 3] = v.w; }
        const uint32_t f0 = reinterpret_cast<const uint32_t*>(m + 2 * SYM_TAIL)[2u * t], f1 = reinterpret_cast<const uint32_t*>(m + 2 * SYM_TAIL)[2u * t + 1u];
        uint32_t c[16];
#pragma unroll
        for (uint32_t m = 0; m < 16; ++m) {                              // four bytes at a time, as in the compose kernel
            const uint32_t a = wv[2 * m], b = wv[2 * m + 1];
            const uint32_t f4 = ((m < 8u ? f0 : f1) >> ((4u * m) & 31u)) & 0xfu;
            const uint32_t x0 = a & 0xffffu;
            if (!f4) c[m] = (a & 0xffu) | ((a >> 8) & 0xff00u) | ((b & 0xffu) << 16) | ((b >> 16) << 24);
            else if (f4 == 0xfu && x0 >= 4u && a == x0 * 0x10001u - 0x10000u && b == a - 0x20002u) c[m] = lds_ld_u32(prev + (SYM_TAIL - x0));
            else if (f4 == 0xfu && x0 && a == x0 * 0x10001u && b == a) c[m] = (uint32_t)prev[SYM_TAIL - x0] * 0x01010101u;
            else {
                uint32_t cm = 0;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t x = ((j < 2 ? a : b) >> (16u * (j & 1u))) & 0xffffu;
                    // (distance 0 -- the identity's word 0, which nothing can name -- never comes out of a composed map of a real block)
                    const uint32_t bb = ((f4 >> j) & 1u) ? (uint32_t)prev[(SYM_TAIL - x) & (SYM_TAIL - 1u)] : (x & 0xffu);
                    cm |= bb << (8u * j);
                }
                c[m] = cm;
            }
        }
        uint8_t* o = starts + (uint64_t)(w + 1) * SYM_TAIL + (uint64_t)t * 64u;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            st_u128(o + j * 16u, make_uint4(c[4 * j], c[4 * j + 1], c[4 * j + 2], c[4 * j + 3]));
            const v4u v = {c[4 * j], c[4 * j + 1], c[4 * j + 2], c[4 * j + 3]};
            *reinterpret_cast<SQY_LDS v4u*>(cur + t * 64u + j * 16u) = v;
        }
        __syncthreads();
        SQY_LDS uint8_t* sw = prev; prev = cur; cur = sw;
    }
}

// everything in front of the tails: a byte that carries a reference takes the byte out of the final tail in front of its block
__global__ __launch_bounds__(256)
void lz4_linked_resolve_bodies_kernel(const uint4* __restrict__ blk, uint32_t nblocks, uint8_t* __restrict__ out, const uint16_t* __restrict__ refs,
                                      uint64_t out_bytes, uint64_t block_bytes, uint32_t pieces_per_block)
{
    const uint32_t k = 1u + blockIdx.x / pieces_per_block, piece = blockIdx.x % pieces_per_block;
    if (k >= nblocks || (blk[k].z >> 31)) return;
    const uint64_t base = (uint64_t)k * block_bytes;
    const uint64_t n = out_bytes - base < block_bytes ? out_bytes - base : block_bytes;
    // (the tail of every block but the last is final already)
    const uint64_t body = k + 1 < nblocks ? n - SYM_TAIL : n;
    const uint64_t nvec = body / 16u;
    for (uint64_t v = (uint64_t)piece * 256u + threadIdx.x; v < nvec; v += (uint64_t)pieces_per_block * 256u) {
        const uint4 ra = ld_u128(reinterpret_cast<const uint8_t*>(refs + base + v * 16u)), rb = ld_u128(reinterpret_cast<const uint8_t*>(refs + base + v * 16u + 8u));
        if (!(ra.x | ra.y | ra.z | ra.w | rb.x | rb.y | rb.z | rb.w)) continue;
        const uint32_t rr[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
        const uint4 dv = ld_u128(out + base + v * 16u);
        uint32_t d[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t rf = (rr[j >> 1] >> (16u * (j & 1u))) & 0xffffu;
            if (rf) {
                const uint32_t b = out[base - rf];
                d[j >> 2] = (d[j >> 2] & ~(0xffu << (8u * (j & 3u)))) | (b << (8u * (j & 3u)));
            }
        }
        st_u128(out + base + v * 16u, make_uint4(d[0], d[1], d[2], d[3]));
    }
    if (piece == 0) {
        for (uint64_t i = nvec * 16u + threadIdx.x; i < body; i += 256u) {
            const uint32_t rf = refs[base + i];
            if (rf) out[base + i] = out[base - rf];
        }
    }
}

// inverse bitswap1 (bitplane_reorder_scalar.hpp:81-116): one thread per group of W voxels
template <typename T>
__global__ __launch_bounds__(256)
void bitswap1_decode_kernel(const T* __restrict__ in, T* __restrict__ out, uint64_t len, uint64_t seg, uint64_t w0)
{
    constexpr uint32_t W = sizeof(T) * 8;
    const uint64_t L = seg * W;
    const uint64_t w = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < seg) {
        uint32_t plane[W];
#pragma unroll
        for (uint32_t b = 0; b < W; ++b) plane[b] = in[(uint64_t)(W - 1 - b) * seg + w];
#pragma unroll
        for (uint32_t j = 0; j < W; ++j) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t b = 0; b < W; ++b) v |= ((plane[b] >> (W - 1 - j)) & 1u) << b;
            out[w * W + j] = (T)v;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < len - L) out[L + threadIdx.x] = in[L + threadIdx.x];
}

// 8-bit planes straight to the quantiser's 16-bit values (decode of `quantiser->bitswap1->...`): the inverse transpose and the
// look-up in one pass -- the 8-bit volume in between (1 byte per voxel written and read again) never exists.  A thread takes 16
// consecutive bytes of every plane (8 coalesced 16-byte loads) = 128 voxels and writes their 256 bytes of 16-bit values.
__global__ __launch_bounds__(256)
void bitswap1_u8_decode_lut_kernel(const uint8_t* __restrict__ in, uint16_t* __restrict__ out, uint64_t nvec, uint64_t seg,
                                   const uint16_t* __restrict__ lut)
{
    __shared__ uint16_t sl[256];
    __shared__ __attribute__((aligned(16))) uint8_t xch[4][16384];     // a wave's 16 KiB of output, so that it leaves 1 KiB per store instruction
    sl[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    lds_u8* const xw = (lds_u8*)xch[wv];
    for (uint64_t v0 = (uint64_t)blockIdx.x * 256 + wv * 64u; v0 < nvec; v0 += (uint64_t)gridDim.x * 256) {
        const uint64_t v = v0 + lane;
        const bool mine = v < nvec;
        if (mine) {
        uint32_t pw[8][4];                                               // pw[b] = 16 bytes of the plane that carries bit b
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(in + (uint64_t)(7 - b) * seg) + v);
            pw[b][0] = t.x; pw[b][1] = t.y; pw[b][2] = t.z; pw[b][3] = t.w;
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            // byte b of (hi:lo) = plane byte of bit b, as stored: voxel j at bit 7 - j.  (Round 5: the eight bytes gathered by six
            // v_perm instead of eight extract / reverse / shift / or chains, no per-byte bit reversal -- the transpose of the
            // unreversed bytes holds voxel j in byte 7 - j, and the look-ups simply take the bytes from the top down --, the transpose
            // on the two 32-bit halves: ~60 instead of ~110 vector instructions per eight voxels; the kernel was bound by them.)
            constexpr uint32_t Z = 0x0c;                                             // v_perm selector: a zero byte
            const uint32_t c = (uint32_t)(g & 3), wdx = (uint32_t)(g >> 2);
            const uint32_t s01 = c | ((4u + c) << 8) | (Z << 16) | (Z << 24), s23 = Z | (Z << 8) | (c << 16) | ((4u + c) << 24);
            uint32_t lo = __builtin_amdgcn_perm(pw[1][wdx], pw[0][wdx], s01) | __builtin_amdgcn_perm(pw[3][wdx], pw[2][wdx], s23);
            uint32_t hi = __builtin_amdgcn_perm(pw[5][wdx], pw[4][wdx], s01) | __builtin_amdgcn_perm(pw[7][wdx], pw[6][wdx], s23);
            // 8x8 bit transpose (its own inverse) of the 64-bit word hi:lo; the first two stages stay inside a half
            uint32_t y;
            y = (lo ^ (lo >> 7)) & 0x00AA00AAu; lo ^= y ^ (y << 7);
            y = (hi ^ (hi >> 7)) & 0x00AA00AAu; hi ^= y ^ (y << 7);
            y = (lo ^ (lo >> 14)) & 0x0000CCCCu; lo ^= y ^ (y << 14);
            y = (hi ^ (hi >> 14)) & 0x0000CCCCu; hi ^= y ^ (y << 14);
            y = (lo ^ (hi << 4)) & 0xF0F0F0F0u; lo ^= y; hi ^= y >> 4;
            // byte i of hi:lo = voxel 7 - i
            v4u o;
            o.x = (uint32_t)sl[hi >> 24] | ((uint32_t)sl[(hi >> 16) & 0xffu] << 16);
            o.y = (uint32_t)sl[(hi >> 8) & 0xffu] | ((uint32_t)sl[hi & 0xffu] << 16);
            o.z = (uint32_t)sl[lo >> 24] | ((uint32_t)sl[(lo >> 16) & 0xffu] << 16);
            o.w = (uint32_t)sl[(lo >> 8) & 0xffu] | ((uint32_t)sl[lo & 0xffu] << 16);
            // piece g of lane t: row t, column g ^ (t & 15) (the columns of sixteen lanes in a row differ: no bank conflict either way)
            *reinterpret_cast<SQY_LDS v4u*>(xw + lane * 256u + (((uint32_t)g ^ (lane & 15u)) << 4)) = o;
        }
        }
        wave_lds_sync();
        // the wave's 16 KiB in order: store k, lane L = piece 64 k + L = piece (L & 15) of lane 4 k + (L >> 4)
        v4u* dst = reinterpret_cast<v4u*>(out + v0 * 128);
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t t = 4u * k + (lane >> 4), g = lane & 15u;
            const v4u o = *reinterpret_cast<const SQY_LDS v4u*>(xw + t * 256u + ((g ^ (t & 15u)) << 4));
            if (v0 + t < nvec) dst[64u * k + lane] = o;
        }
        wave_lds_sync();
    }
}

// 16-bit, whole tiles of 8192 voxels: the mirror image of bitswap1_u16_regs.  A lane takes 8 consecutive words of every plane
// (16 coalesced 1 KiB loads per wave), transposes them two groups at a time (the bit transpose is its own inverse) and has its
// 128 voxels as 16 x 16 B.  Round 5: those leave through LDS, 1 KiB in a row per store instruction -- a lane storing its own 256 bytes
// 16 at a time (sixteen-byte pieces 256 bytes apart across the wave) kept the kernel at 4.9 TB/s of read + write, the 8-bit sibling,
// which writes two bytes for every byte it reads, at 2.8.  (The generic kernel moves 2 bytes per lane and instruction.)
__global__ __launch_bounds__(256)
void bitswap1_decode_u16_regs(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, uint64_t n_tiles, uint64_t seg_words)
{
    __shared__ __attribute__((aligned(16))) uint8_t xch[4][16384];     // a wave's tile of output
    const uint32_t lane = threadIdx.x & 63;
    lds_u8* const xw = (lds_u8*)xch[threadIdx.x >> 6];
    const uint64_t wave_global = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t wave_stride = (uint64_t)gridDim.x * 4;
    for (uint64_t tile = wave_global; tile < n_tiles; tile += wave_stride) {
        uint32_t pl[16][4];
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const v4u* src = reinterpret_cast<const v4u*>(in + (uint64_t)(15 - b) * seg_words + tile * (BSW_TILE_VOX / 16));
            const v4u t = __builtin_nontemporal_load(src + lane);
            pl[b][0] = t.x; pl[b][1] = t.y; pl[b][2] = t.z; pl[b][3] = t.w;
        }
        // piece g (16 bytes) of lane t: row t, column g ^ (t & 15) (the columns of sixteen lanes in a row differ: no bank conflicts)
        auto put = [&](uint32_t g, const v4u& val) { *reinterpret_cast<SQY_LDS v4u*>(xw + lane * 256u + ((g ^ (lane & 15u)) << 4)) = val; };
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t r[16];
#pragma unroll
            for (int b = 0; b < 16; ++b) r[b] = pl[b][q];
            transpose16x16_pairs(r);                                      // r[i] = voxel 15-i of group A | of group B << 16
            uint32_t ga[8], gb[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                ga[k] = __builtin_amdgcn_perm(r[14 - 2 * k], r[15 - 2 * k], 0x05040100u);
                gb[k] = __builtin_amdgcn_perm(r[14 - 2 * k], r[15 - 2 * k], 0x07060302u);
            }
            const v4u a0 = {ga[0], ga[1], ga[2], ga[3]}, a1 = {ga[4], ga[5], ga[6], ga[7]};
            const v4u b0 = {gb[0], gb[1], gb[2], gb[3]}, b1 = {gb[4], gb[5], gb[6], gb[7]};
            put(4u * q, a0); put(4u * q + 1u, a1); put(4u * q + 2u, b0); put(4u * q + 3u, b1);
        }
        wave_lds_sync();
        // the tile's 16 KiB in order: store k, lane L = piece 64 k + L = piece (L & 15) of lane 4 k + (L >> 4)
        v4u* dst = reinterpret_cast<v4u*>(out + tile * BSW_TILE_VOX);
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t t = 4u * k + (lane >> 4), g = lane & 15u;
            dst[64u * k + lane] = *reinterpret_cast<const SQY_LDS v4u*>(xw + t * 256u + ((g ^ (t & 15u)) << 4));
        }
        wave_lds_sync();
    }
}

// inverse diff3x3x1, voxels [r0, r1) of frame z (everything before them must be final): diff_scheme_impl.hpp:143-194.
// The reference decodes in raster order and reads its own output: a rewritten voxel needs the 3x3 neighbourhood one frame back,
// i.e. indices idx - frame - X - 1 .. idx - frame + X + 1.  Those lie in frame z-1 -- except for the last X + 1 voxels of a
// frame, whose lower neighbours are the first voxels of frame z itself (they are rewritten only in geometries whose rows' reach
// spills over the row end: Z - 2 > X - 1, or the single-row case).  The launcher therefore decodes such frames in two steps.
template <typename T, typename ST, bool SCHAR = false>
__global__ __launch_bounds__(256)
void diff3x3x1_decode_plane_kernel(const T* __restrict__ in, T* __restrict__ out, uint64_t z, uint64_t length, uint64_t Y, uint64_t X,
                                   uint64_t hx, uint64_t zlim, int single, uint64_t r0, uint64_t r1)
{
    const uint64_t frame = Y * X;
    const uint64_t r = r0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= r1) return;
    const uint64_t idx = z * frame + r;
    T v = in[idx];
    if (z >= 1 && diff_touched(idx, length, Y, X, hx, zlim, single != 0)) {
        const T* p = out + idx - frame;
        T sum = 0;
        sum = (T)(sum + p[-(int64_t)X - 1]); sum = (T)(sum + p[-(int64_t)X]); sum = (T)(sum + p[-(int64_t)X + 1]);
        sum = (T)(sum + p[-1]);              sum = (T)(sum + p[0]);           sum = (T)(sum + p[1]);
        sum = (T)(sum + p[X - 1]);           sum = (T)(sum + p[X]);           sum = (T)(sum + p[X + 1]);
        v = (T)((uint32_t)(int32_t)(ST)v + (SCHAR ? (uint32_t)(uint16_t)(int16_t)(int8_t)sum / 9u : (uint32_t)sum / 9u));
    }
    out[idx] = v;
}

// (Round 2 had a one-launch inverse here -- strips of rows that waited for their neighbours through an exchange buffer, all
// workgroups resident at once.  A plain launch cannot promise that: two decodes in flight could each hold half the chip and wait
// for strips that never start; a process-local mutex hid it.  A cooperative launch promises it, but concurrent cooperative
// launches from several host threads crash the HIP runtime at process exit (ROCm 7.2, seen in round 3).  The inverse is now a
// chain of ordinary launches, one per frame, over the columns the stage can touch -- no kernel of this library waits for another
// workgroup any more.)


// inverse quantiser: out[i] = lut_decode[in[i]]
__global__ __launch_bounds__(256)
void quantiser_decode_kernel(const uint8_t* __restrict__ in, uint16_t* __restrict__ out, uint64_t len, const uint16_t* __restrict__ lut)
{
    __shared__ uint16_t sl[256];
    sl[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    // 8 voxels per thread and step: one 8-byte load, one 16-byte store
    const uint64_t nvec = ((reinterpret_cast<uintptr_t>(in) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) ? len / 8 : 0;
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (uint64_t)gridDim.x * 256) {
        const uint2 x = reinterpret_cast<const uint2*>(in)[v];
        uint4 y;
        y.x = (uint32_t)sl[x.x & 0xffu] | ((uint32_t)sl[(x.x >> 8) & 0xffu] << 16);
        y.y = (uint32_t)sl[(x.x >> 16) & 0xffu] | ((uint32_t)sl[x.x >> 24] << 16);
        y.z = (uint32_t)sl[x.y & 0xffu] | ((uint32_t)sl[(x.y >> 8) & 0xffu] << 16);
        y.w = (uint32_t)sl[(x.y >> 16) & 0xffu] | ((uint32_t)sl[x.y >> 24] << 16);
        reinterpret_cast<uint4*>(out)[v] = y;
    }
    for (uint64_t i = nvec * 8 + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (uint64_t)gridDim.x * 256) out[i] = sl[in[i]];
}

// inverse frame_shuffle: out frame map[i] = in frame i (frame_shuffle_utils.hpp:313-357)
__global__ __launch_bounds__(256)
void frame_scatter_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t frame_bytes,
                          const uint64_t* __restrict__ map, uint32_t blocks_per_frame)
{
    const uint64_t f = blockIdx.x / blocks_per_frame;
    const uint32_t part = blockIdx.x % blocks_per_frame;
    const uint8_t* s = in + f * frame_bytes;
    const uint64_t to = map[f];
    if (to == ~0ull) return;                                   // a later frame of the stream goes to the same place: that one counts (host)
    uint8_t* d = out + to * frame_bytes;
    const uint64_t nvec = frame_bytes / 16;
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
        const uint64_t step = (uint64_t)blocks_per_frame * 256;
        for (uint64_t v0 = (uint64_t)part * 256 + threadIdx.x; v0 < nvec; v0 += 4 * step) {   // four loads in flight per thread
            uint4 t[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) if (v0 + j * step < nvec) t[j] = reinterpret_cast<const uint4*>(s)[v0 + j * step];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) if (v0 + j * step < nvec) reinterpret_cast<uint4*>(d)[v0 + j * step] = t[j];
        }
        if (part == 0)
            for (uint64_t i = nvec * 16 + threadIdx.x; i < frame_bytes; i += 256) d[i] = s[i];
    } else {
        for (uint64_t i = (uint64_t)part * 256 + threadIdx.x; i < frame_bytes; i += (uint64_t)blocks_per_frame * 256) d[i] = s[i];
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline int num_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

std::atomic<long> g_bsw_blocks_per_cu{32};       // (measurement switch: SQYAMD_Set_Option("transpose_blocks_per_cu"))
void set_bitswap1_blocks_per_cu(long n) { g_bsw_blocks_per_cu.store(n); }

hipError_t launch_bitswap1_u16(const uint16_t* in, uint16_t* out, uint64_t len, hipStream_t stream, uint32_t* piece_hash, uint32_t gap_chunk,
                               const uint16_t* side, uint32_t side_w, uint32_t X, uint32_t* digest, uint32_t digest_stride)
{
    if (len == 0) return hipSuccess;
    if (side && (X == 0 || X % 128u != 0 || side_w % 128u != 0 || side_w > X || len % X != 0 || (reinterpret_cast<uintptr_t>(side) & 15) ||
                 len % BSW_TILE_VOX != 0 || (reinterpret_cast<uintptr_t>(in) & 15) || (!gap_chunk && (reinterpret_cast<uintptr_t>(out) & 15))))
        return hipErrorInvalidValue;
    if (gap_chunk) {
        // frames in place: whole tiles only, chunks a power of two of at least one piece, `out` = body of chunk 0 (any alignment)
        // (and piece hashes: all-zero pieces are left unwritten, the hashes tell the readers which)
        if (len % BSW_TILE_VOX != 0 || (reinterpret_cast<uintptr_t>(in) & 15) || gap_chunk < 1024u || (gap_chunk & (gap_chunk - 1u)) || !piece_hash)
            return hipErrorInvalidValue;
        // (round 6, measured: fewer resident blocks -- 8 .. 2 per CU instead of as many as fit -- let the small kernels of the other calls in
        // flight start sooner, and cost the transposes and with them the step 3 .. 12 %)
        const uint64_t n_tiles = len / BSW_TILE_VOX, want = (n_tiles + 1) / 2, cap = (uint64_t)num_cus() * (g_bsw_blocks_per_cu.load() > 0 ? (uint64_t)g_bsw_blocks_per_cu.load() : 32u);    // (blocks of two waves: see below)
        // (the noise digest: only when every plane segment is a whole number of chunks -- a lane's place inside its chunk is then the same
        // in all sixteen planes -- and the chunk long enough for the search to reach step 16)
        if (digest && ((len / 8) % gap_chunk != 0 || gap_chunk < 16384u || digest_stride == 0)) return hipErrorInvalidValue;
        hipLaunchKernelGGL(bitswap1_u16_regs<true>, dim3((unsigned)(want < cap ? want : cap)), dim3(128), 0, stream, in, out, n_tiles, len / 16,
                           piece_hash, (uint32_t)__builtin_ctz(gap_chunk), side, side_w, X, digest, digest_stride);
        return hipGetLastError();
    }
    const uint64_t seg_words = len / 16;
    uint64_t n_tiles = 0;
    const bool aligned = (seg_words % 8 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (aligned) n_tiles = (seg_words * 16) / BSW_TILE_VOX;
    // piece hashes only when the tile kernel covers the whole buffer (see bitswap1_piece_hash_words)
    if (piece_hash && n_tiles * BSW_TILE_VOX != len) return hipErrorInvalidValue;
    // the register-tile kernel.  It needs no LDS, so it runs next to the LZ4 chunk waves of other calls in flight (those hold nearly
    // all of a CU's LDS) -- in blocks of TWO waves: the waves of a block go to different SIMDs, and with chunk waves resident a block of
    // four seldom finds room on all four at once (measured with three calls in flight: 0.78 ms per launch in blocks of four waves,
    // 0.49 ms in blocks of two or one; alone 0.34 ms either way)
    if (n_tiles) {
        const uint64_t want = (n_tiles + 1) / 2;
        const uint64_t cap = (uint64_t)num_cus() * 32;
        const unsigned grid = (unsigned)(want < cap ? want : cap);
        hipLaunchKernelGGL(bitswap1_u16_regs<false>, dim3(grid), dim3(128), 0, stream, in, out, n_tiles, seg_words, piece_hash, 0u, side, side_w, X,
                           (uint32_t*)nullptr, 0u);
    }
    const uint64_t first_word = n_tiles * (BSW_TILE_VOX / 16);
    const uint64_t rest_words = seg_words - first_word;
    const uint64_t tail = len - seg_words * 16;
    if (rest_words || tail) {
        uint64_t blocks = (rest_words + 255) / 256;
        if (blocks == 0) blocks = 1;
        hipLaunchKernelGGL(bitswap1_u16_generic, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, len, first_word, seg_words);
    }
    return hipGetLastError();
}

uint32_t lz4_noise_digest_stride(uint32_t chunk)
{
    // probes of a search that starts with the chunk and never finds anything (liblz4's step schedule), from probe 961 on; 0: no digest
    if (chunk < 16384u || (chunk & (chunk - 1u))) return 0;
    uint64_t p = 1, st = 1, nb = 64, probes = 0;
    for (;;) {
        ++probes;
        const uint64_t p2 = p + st;
        st = nb >> 6; ++nb;
        if (p2 > (uint64_t)chunk - 12 + 1) break;
        p = p2;
    }
    if (probes <= 961) return 0;
    return (uint32_t)(((probes - 960) + 63) & ~(uint64_t)63);
}

uint64_t bitswap1_piece_hash_words(const void* in, const void* out, uint64_t len)
{
    // one 1 KiB piece per plane and tile, four words each; 0 when the tile kernel does not cover the buffer exactly
    if (len == 0 || len % BSW_TILE_VOX != 0) return 0;
    if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return 0;
    return (len / BSW_TILE_VOX) * 16 * 4;
}

// the duplicate search's table emptied (keys 0, values ~0) and one more word zeroed (the dense pass's list counter) by ONE small kernel
// that a call can launch in front of its bit-plane transpose: three fill dispatches less between the kernels of a call (round 4)
__global__ __launch_bounds__(256)
void lz4_dedupe_clear_kernel(uint64_t* __restrict__ tab_key, uint32_t* __restrict__ tab_val, uint32_t tab, uint32_t* __restrict__ zero_word)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < tab) { tab_key[i] = 0ull; tab_val[i] = 0xffffffffu; }
    if (i == 0 && zero_word) *zero_word = 0u;
}

hipError_t launch_lz4_dedupe_clear(void* work, uint64_t nchunks, uint32_t* zero_word, hipStream_t stream)
{
    if (nchunks == 0) return hipSuccess;
    uint32_t tab = 64;
    while (tab < 2 * nchunks) tab <<= 1;
    uint64_t* chunk_key = static_cast<uint64_t*>(work);
    uint64_t* tab_key = chunk_key + nchunks;
    uint32_t* tab_val = reinterpret_cast<uint32_t*>(tab_key + tab);
    hipLaunchKernelGGL(lz4_dedupe_clear_kernel, dim3((tab + 255u) / 256u), dim3(256), 0, stream, tab_key, tab_val, tab, zero_word);
    return hipGetLastError();
}

hipError_t launch_lz4_dedupe(const uint8_t* in, uint64_t total, uint32_t chunk, const uint32_t* piece_hash, void* work,
                             uint32_t* dup_of, hipStream_t stream, uint64_t in_stride, uint64_t* holes_map, bool table_is_clear,
                             Lz4DedupeArgs* fused)
{
    const uint64_t nchunks = (total + chunk - 1) / chunk, nfull = total / chunk;
    if (nchunks == 0) return hipSuccess;
    if (in_stride == 0) in_stride = chunk;
    if (chunk % 1024 != 0) return hipErrorInvalidValue;
    uint32_t tab = 64;
    while (tab < 2 * nchunks) tab <<= 1;
    uint64_t* chunk_key = static_cast<uint64_t*>(work);
    uint64_t* tab_key = chunk_key + nchunks;
    uint32_t* tab_val = reinterpret_cast<uint32_t*>(tab_key + tab);
    hipError_t e = hipSuccess;
    if (!table_is_clear) {
        e = hipMemsetAsync(tab_key, 0, (size_t)tab * 8, stream);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(tab_val, 0xff, (size_t)tab * 4, stream);
        if (e != hipSuccess) return e;
    }
    if (holes_map && total % 1024u != 0) return hipErrorInvalidValue;
    const uint64_t nkey = holes_map ? nchunks : nfull;
    if (nkey)
        hipLaunchKernelGGL(lz4_dedupe_key_kernel, dim3((unsigned)nkey), dim3(64), 0, stream, piece_hash, chunk / 1024u, nfull, chunk_key, tab_key,
                           tab_val, tab - 1u, holes_map, (uint32_t)((total - nfull * chunk) >> 10));
    if (fused) {
        // the decision per chunk is left to the chunk's parse wavefront (lz4_chunk_dedupe): only the key table is built here
        fused->chunk_key = chunk_key; fused->tab_key = tab_key; fused->tab_val = tab_val; fused->tab_mask = tab - 1u; fused->dup_of = dup_of;
        fused->piece_hash = piece_hash; fused->holes_map = holes_map; fused->nchunks_full = nfull;
        return hipGetLastError();
    }
    hipLaunchKernelGGL(lz4_dedupe_verify_kernel, dim3((unsigned)nchunks), dim3(DEDUPE_THREADS), 0, stream, in, chunk, in_stride, nfull, nchunks, chunk_key,
                       tab_key, tab_val, tab - 1u, dup_of, piece_hash, holes_map, total);
    return hipGetLastError();
}

uint64_t lz4_holes_map_bytes(uint64_t nchunks, uint32_t chunk)
{
    return nchunks * (1u + ((uint64_t)(chunk >> 10) + 63u) / 64u) * 8u;
}

uint64_t lz4_dedupe_work_bytes(uint64_t nchunks)
{
    uint64_t tab = 64;
    while (tab < 2 * nchunks) tab <<= 1;
    return nchunks * 8 + tab * 8 + tab * 4 + 64;
}

hipError_t launch_quantiser_apply_bitswap1_u8(const uint16_t* in, uint8_t* out, uint64_t len, const uint8_t* lut, hipStream_t stream)
{
    if (len == 0) return hipSuccess;
    if (reinterpret_cast<uintptr_t>(in) & 15) return hipErrorInvalidValue;
    uint64_t blocks = (len / 8 + 256 * 16 - 1) / (256 * 16);
    const uint64_t cap = (uint64_t)num_cus() * 2;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(quantiser_apply_bitswap1_u8_kernel, dim3((unsigned)blocks), dim3(256), 65536, stream, in, out, len, lut);
    return hipGetLastError();
}

hipError_t launch_bitswap1_u8(const uint8_t* in, uint8_t* out, uint64_t len, hipStream_t stream)
{
    if (len == 0) return hipSuccess;
    const uint64_t seg = len / 8;
    uint64_t blocks = (seg + 127) / 128;                       // (blocks of two waves: they find room next to resident LZ4 chunk waves)
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(bitswap1_u8_generic, dim3((unsigned)blocks), dim3(128), 0, stream, in, out, len, seg);
    return hipGetLastError();
}

// the rows kernel's geometry (every row's reach 1 + hx stays inside its row) and the width of the compact side buffer a bit-plane
// transpose right behind the stage can take the touched columns from: 0 = not applicable
uint32_t diff3x3x1_side_width(uint64_t Z, uint64_t Y, uint64_t X, int elem_size)
{
    const uint64_t length = Z * Y * X;
    if (length == 0 || elem_size != 2 || X % 128 != 0 || length % BSW_TILE_VOX != 0) return 0;
    const uint64_t zlim = X < Z ? X : Z;
    const uint64_t noff = (zlim >= 1 ? (zlim - 1) : 0) * (Y >= 2 ? (Y - 2) : 0);
    if (noff == 1) return 0;
    const uint64_t hx = Z >= 2 ? Z - 2 : 0;
    if (!(hx + 2 <= X && Y <= 65535 && Z <= 65535 && X <= 0xffffffffull)) return 0;
    const uint64_t w = ((1 + hx) + 127) / 128 * 128;                    // columns [0, 1 + hx) rounded up to whole lanes of the transpose
    return w < X ? (uint32_t)w : 0;                                     // (no saving when every column can be touched)
}

hipError_t launch_diff3x3x1_side(const uint16_t* in, uint16_t* side, uint64_t Z, uint64_t Y, uint64_t X, uint32_t side_w, hipStream_t stream)
{
    if (!side_w || side_w != diff3x3x1_side_width(Z, Y, X, 2)) return hipErrorInvalidValue;
    const uint64_t zlim = X < Z ? X : Z, hx = Z >= 2 ? Z - 2 : 0;
    // 8 voxels per thread: a row of the side buffer is side_w / 8 threads; narrow buffers put several rows into one block
    uint32_t rpb = 1;
    while (rpb < 16u && (side_w / 8u) * rpb * 2u <= 256u) rpb *= 2u;
    const unsigned bx = (unsigned)((side_w / 8u + (256u / rpb) - 1u) / (256u / rpb));
    hipLaunchKernelGGL(diff3x3x1_u16_rows_kernel<false>, dim3(bx, (unsigned)((Y + rpb - 1) / rpb), (unsigned)Z), dim3(256), 0, stream, in, side, (uint32_t)Y,
                       (uint32_t)X, (uint32_t)hx, (uint32_t)zlim, side_w, side_w, rpb, 0u, 0u);
    return hipGetLastError();
}

hipError_t launch_diff3x3x1(const void* in, void* out, uint64_t Z, uint64_t Y, uint64_t X, int elem_size, hipStream_t stream, bool schar)
{
    // geometry of the reference's halo (neighborhood_utils.hpp:160-240), see the kernel header comment
    const uint64_t length = Z * Y * X;
    if (length == 0) return hipSuccess;
    const uint64_t zlim = X < Z ? X : Z;                         // z in [1, min(X, Z))
    const uint64_t noff = (zlim >= 1 ? (zlim - 1) : 0) * (Y >= 2 ? (Y - 2) : 0);
    const int single = (noff == 1);
    const uint64_t hx = single ? 0 : (Z >= 2 ? Z - 2 : 0);
    if (elem_size == 2 && !single && hx + 2 <= X && Y <= 65535 && Z <= 65535 && X <= 0xffffffffull) {   // reach stays inside the row, x+1 too
        const unsigned bx = (unsigned)((X + 2047) / 2048);
        hipLaunchKernelGGL(diff3x3x1_u16_rows_kernel<false>, dim3(bx, (unsigned)Y, (unsigned)Z), dim3(256), 0, stream, (const uint16_t*)in,
                           (uint16_t*)out, (uint32_t)Y, (uint32_t)X, (uint32_t)hx, (uint32_t)zlim, (uint32_t)X, (uint32_t)X, 1u, 0u, 0u);
        return hipGetLastError();
    }
    uint64_t blocks = (length + 255) / 256;
    const uint64_t cap = (uint64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (elem_size == 2)
        hipLaunchKernelGGL((diff3x3x1_kernel<uint16_t, int16_t>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           (const uint16_t*)in, (uint16_t*)out, Z, Y, X, hx, zlim, single);
    else if (schar)
        hipLaunchKernelGGL((diff3x3x1_kernel<uint8_t, int8_t, true>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           (const uint8_t*)in, (uint8_t*)out, Z, Y, X, hx, zlim, single);
    else
        hipLaunchKernelGGL((diff3x3x1_kernel<uint8_t, int8_t>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           (const uint8_t*)in, (uint8_t*)out, Z, Y, X, hx, zlim, single);
    return hipGetLastError();
}

#ifdef SQY_LZ4_DIAG
#define SQY_DIAG_NULL , (unsigned long long*)nullptr
#else
#define SQY_DIAG_NULL
#endif

hipError_t launch_lz4_chunks(const uint8_t* in, uint64_t total, uint32_t chunk, uint8_t* scratch, uint64_t stride,
                             uint32_t* csize, uint64_t nchunks, hipStream_t stream, const uint64_t* frame_map, uint64_t frame_bytes,
                             uint32_t* redo, const uint32_t* dup_of, uint64_t in_stride, uint32_t acceleration, bool redo_is_zero,
                             const Lz4DedupeArgs* dedupe)
{
    if (nchunks == 0) return hipSuccess;
    if (in_stride == 0) in_stride = chunk;
    if (frame_map && (frame_bytes == 0 || frame_bytes % chunk != 0)) return hipErrorInvalidValue;
    if (redo && !redo_is_zero) {
        const hipError_t e = hipMemsetAsync(redo, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    if (acceleration > 1)    // (no dense second pass: the list stays empty)
        hipLaunchKernelGGL((lz4_chunks_kernel<false, false, true>), dim3((unsigned)nchunks), dim3(64), 0, stream, in, total, chunk, in_stride, scratch, stride, csize,
                           frame_map, frame_bytes, (const Lz4Block*)nullptr, (const uint32_t*)nullptr, 0u, (uint32_t*)nullptr, dup_of, acceleration, Lz4DedupeArgs{} SQY_DIAG_NULL);
    else
        hipLaunchKernelGGL((lz4_chunks_kernel<false, false>), dim3((unsigned)nchunks), dim3(64), 0, stream, in, total, chunk, in_stride, scratch, stride, csize,
                           frame_map, frame_bytes, (const Lz4Block*)nullptr, (const uint32_t*)nullptr, 0u, redo, dup_of, 1u, dedupe ? *dedupe : Lz4DedupeArgs{} SQY_DIAG_NULL);
    return hipGetLastError();
}

hipError_t launch_lz4_chunks_dense(const uint8_t* in, uint64_t total, uint32_t chunk, uint8_t* scratch, uint64_t stride,
                                   uint32_t* csize, uint32_t* redo, uint32_t redo_count, hipStream_t stream,
                                   const uint64_t* frame_map, uint64_t frame_bytes, uint64_t in_stride)
{
    if (redo_count == 0) return hipSuccess;
    if (in_stride == 0) in_stride = chunk;
    hipLaunchKernelGGL((lz4_chunks_kernel<false, true>), dim3(redo_count), dim3(64), 0, stream, in, total, chunk, in_stride, scratch, stride, csize,
                       frame_map, frame_bytes, (const Lz4Block*)nullptr, (const uint32_t*)nullptr, 0u, redo, (const uint32_t*)nullptr, 1u, Lz4DedupeArgs{} SQY_DIAG_NULL);
    return hipGetLastError();
}

hipError_t launch_lz4_linked(const uint8_t* in, const Lz4Block* blocks, const uint32_t* frame_first, uint64_t nframes,
                             uint32_t max_block, uint8_t* scratch, uint64_t stride, uint32_t* csize, hipStream_t stream, uint32_t acceleration)
{
    if (nframes == 0) return hipSuccess;
    if (max_block == 0 || max_block > (4u << 20)) return hipErrorInvalidValue;
    if (acceleration > 1)
        hipLaunchKernelGGL((lz4_chunks_kernel<true, false, true>), dim3((unsigned)nframes), dim3(64), 0, stream, in, (uint64_t)0, 0u, (uint64_t)0, scratch, stride, csize,
                           (const uint64_t*)nullptr, (uint64_t)0, blocks, frame_first, max_block, (uint32_t*)nullptr, (const uint32_t*)nullptr, acceleration, Lz4SpecArgs{} SQY_DIAG_NULL);
    else
        hipLaunchKernelGGL((lz4_chunks_kernel<true, false>), dim3((unsigned)nframes), dim3(64), 0, stream, in, (uint64_t)0, 0u, (uint64_t)0, scratch, stride, csize,
                           (const uint64_t*)nullptr, (uint64_t)0, blocks, frame_first, max_block, (uint32_t*)nullptr, (const uint32_t*)nullptr, 1u, Lz4SpecArgs{} SQY_DIAG_NULL);
    return hipGetLastError();
}

hipError_t launch_lz4_linked_spec(const uint8_t* in, const Lz4Block* blocks, const Lz4SpecArgs& spec, uint64_t nwaves,
                                  uint32_t max_block, uint8_t* scratch, uint64_t stride, uint32_t* csize, hipStream_t stream, uint32_t acceleration)
{
    if (nwaves == 0) return hipSuccess;
    if (max_block == 0 || max_block > (4u << 20) || !spec.wave_first || !spec.wave_last || !spec.tables || (spec.mode != 1 && spec.mode != 2))
        return hipErrorInvalidValue;
    // mode 2 (runs of blocks whose guess failed, parsed again in order): the kernel with the dense batches -- it is the lean kernel
    // until it meets two short matches in a row, and 2.4 x faster per block on streams of short sequences, where a run can be
    // hundreds of blocks long (round 5; 50 KiB of LDS per wavefront, which a handful of runs do not mind)
    if (acceleration <= 1 && spec.mode == 2) {
        hipLaunchKernelGGL((lz4_chunks_kernel<true, true>), dim3((unsigned)nwaves), dim3(64), 0, stream, in, (uint64_t)0, 0u, (uint64_t)0, scratch, stride, csize,
                           (const uint64_t*)nullptr, (uint64_t)0, blocks, (const uint32_t*)nullptr, max_block, (uint32_t*)nullptr, (const uint32_t*)nullptr, 1u, spec SQY_DIAG_NULL);
        return hipGetLastError();
    }
    if (acceleration > 1)
        hipLaunchKernelGGL((lz4_chunks_kernel<true, false, true>), dim3((unsigned)nwaves), dim3(64), 0, stream, in, (uint64_t)0, 0u, (uint64_t)0, scratch, stride, csize,
                           (const uint64_t*)nullptr, (uint64_t)0, blocks, (const uint32_t*)nullptr, max_block, (uint32_t*)nullptr, (const uint32_t*)nullptr, acceleration, spec SQY_DIAG_NULL);
    else
        hipLaunchKernelGGL((lz4_chunks_kernel<true, false>), dim3((unsigned)nwaves), dim3(64), 0, stream, in, (uint64_t)0, 0u, (uint64_t)0, scratch, stride, csize,
                           (const uint64_t*)nullptr, (uint64_t)0, blocks, (const uint32_t*)nullptr, max_block, (uint32_t*)nullptr, (const uint32_t*)nullptr, 1u, spec SQY_DIAG_NULL);
    return hipGetLastError();
}

// ok[k]: block k opens a frame, or the table it was parsed from (tables[k][0]) is what block k - 1 left (tables[k - 1][1]) re-based
// by that block's size exactly as the walk re-bases it
__global__ __launch_bounds__(256)
void lz4_linked_verify_kernel(const Lz4Block* __restrict__ blocks, const uint32_t* __restrict__ tables, uint32_t max_block, uint32_t* __restrict__ ok)
{
    const uint64_t k = blockIdx.x;
    __shared__ uint32_t differ;
    if (threadIdx.x == 0) differ = 0;
    __syncthreads();
    if (!(blocks[k].flags & 1u)) {
        const uint32_t tsh = lz4_linked_tag_shift(max_block), tmask = (1u << tsh) - 1u;
        const uint32_t n_prev = blocks[k - 1].n;
        const uint32_t* __restrict__ left = tables + (k - 1) * kLz4SpecTableWords + 4096u;
        const uint32_t* __restrict__ used = tables + k * kLz4SpecTableWords;
        uint32_t d = 0;
        for (uint32_t i = threadIdx.x; i < 4096u; i += 256u) {
            const uint32_t e = left[i];
            const uint32_t pp = e >> tsh;
            const uint32_t e2 = pp >= n_prev ? (((pp - n_prev) << tsh) | (e & tmask)) : 0u;
            d |= e2 ^ used[i];
        }
        if (d) differ = 1;                                      // (benign race: every writer writes 1)
    }
    __syncthreads();
    if (threadIdx.x == 0) ok[k] = differ ? 0u : 1u;
}

hipError_t launch_lz4_linked_verify(const Lz4Block* blocks, uint64_t nblocks, const uint32_t* tables, uint32_t max_block, uint32_t* ok, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    if (max_block == 0 || max_block > (4u << 20)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lz4_linked_verify_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, blocks, tables, max_block, ok);
    return hipGetLastError();
}

hipError_t launch_lz4_frame_scan(const uint32_t* csize, uint64_t nchunks, uint64_t total, uint32_t chunk,
                                 uint64_t* frame_off, hipStream_t stream, const Lz4Block* blocks, const uint32_t* dup_of, uint64_t* tail_info,
                                 const uint32_t* guard, uint8_t* body0, uint64_t in_stride, uint32_t bd_byte, uint32_t hc_byte)
{
    hipLaunchKernelGGL(lz4_frame_scan_kernel, dim3(1), dim3(256), 0, stream, csize, nchunks, total, chunk, frame_off, blocks, dup_of, tail_info,
                       guard, body0, in_stride, bd_byte, hc_byte);
    return hipGetLastError();
}

// ---- frames in place, the tail of a call driven from the device (round 4): no host round trip between the parse and the blob ----
// ------------------------------------------------------------------------------------------------
// Frames in place, the whole tail of a call in ONE kernel (round 6).  Scan, tail marks, stash, gather and header were five dependent
// launches; alone that is 0.07 ms, but with calls in flight every launch waits its turn on a full chip and every kernel's chain of
// dependent loads (duplicate map -> size -> offset, sixteen times in a row in the scan) runs at the loaded HBM latency: 0.3 ms of a
// call's 2.4.  Here every workgroup reads ALL the sizes itself (a few loads per thread, all in flight at once) and reduces them to
// what it needs: where the stored tail begins (j), the bytes of the frames in front of it, the bytes in front of its own chunks.
// No workgroup waits for another.  Workgroup b owns chunks [b * cpb, (b + 1) * cpb): frames in front of j are gathered to where they
// end up, chunks from j on get their marks; workgroup 0 also writes the sqy header and the record the host reads.
// Stored chunks IN FRONT of j (their bodies lie where gathered frames go) need the stash pass first: status 4, the host runs the
// separate kernels.  guard[0] != 0 (chunks left to the dense pass): status 2 as before.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t TAIL_THREADS = 64;         // (one wave: with calls in flight a workgroup of four waited for a CU with room for all four -- 0.23 ms against 0.10)
__global__ __launch_bounds__(TAIL_THREADS)
void lz4_inplace_tail_fused_kernel(uint8_t* __restrict__ out, uint64_t t0, uint64_t in_stride, uint64_t total, uint32_t chunk, uint64_t nchunks,
                                   uint32_t cpb, const uint8_t* __restrict__ scratch, uint64_t stride, const uint32_t* __restrict__ csize,
                                   uint64_t* __restrict__ frame_off, const uint32_t* __restrict__ dup_of, uint64_t* __restrict__ tail_info,
                                   uint32_t bd_byte, uint32_t hc_byte, Lz4HeaderParts hp, uint32_t elem_size, const uint32_t* __restrict__ guard,
                                   uint64_t* __restrict__ record)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (guard && guard[0] != 0u) {
        if (blockIdx.x == 0 && tid == 0) { record[6] = guard[0]; record[0] = 2; }
        return;
    }
    __shared__ uint64_t red_sum[TAIL_THREADS / 64], red_before[TAIL_THREADS / 64];
    __shared__ uint32_t red_j[TAIL_THREADS / 64], red_zero[TAIL_THREADS / 64];
    __shared__ uint64_t own_off[64];                       // frame offsets of this workgroup's chunks (cpb <= 64)
    __shared__ uint32_t own_c[64], own_ks[64];
    const uint64_t k_first = (uint64_t)blockIdx.x * cpb;
    // ---- every size of the call, eight per thread in flight ----
    uint64_t sum = 0, before = 0;
    uint32_t jmax = 0, nzero = 0;
    for (uint64_t base = 0; base < nchunks; base += TAIL_THREADS * 8u) {
        uint32_t ks[8], c[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint64_t k = base + u * TAIL_THREADS + tid;
            ks[u] = k < nchunks ? (dup_of ? dup_of[k] : (uint32_t)k) : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint64_t k = base + u * TAIL_THREADS + tid;
            c[u] = k < nchunks ? csize[ks[u]] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint64_t k = base + u * TAIL_THREADS + tid;
            if (k < nchunks) {
                const uint64_t left = total - k * chunk;
                const uint64_t nk = left < chunk ? left : chunk;
                const uint64_t sz = 15u + (c[u] ? c[u] : nk);
                sum += sz;
                if (k < k_first) before += sz;
                if (c[u]) jmax = jmax > (uint32_t)k + 1u ? jmax : (uint32_t)k + 1u; else nzero += 1u;
                if (k >= k_first && k < k_first + cpb) { own_c[k - k_first] = c[u]; own_ks[k - k_first] = ks[u]; }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_xor(sum, d); before += __shfl_xor(before, d); nzero += __shfl_xor(nzero, d);
        const uint32_t o = __shfl_xor(jmax, d);
        jmax = jmax > o ? jmax : o;
    }
    if (lane == 0) { red_sum[wave] = sum; red_before[wave] = before; red_j[wave] = jmax; red_zero[wave] = nzero; }
    __syncthreads();
    sum = 0; before = 0; jmax = 0; nzero = 0;
#pragma unroll
    for (uint32_t w = 0; w < TAIL_THREADS / 64; ++w) {
        sum += red_sum[w]; before += red_before[w]; nzero += red_zero[w];
        jmax = jmax > red_j[w] ? jmax : red_j[w];
    }
    const uint64_t j = jmax;                                                  // first chunk of the stored run that ends the stream
    const uint64_t tail_bytes = (nchunks - j) * 15u + (total - (j * chunk < total ? j * chunk : total));
    const uint64_t head_bytes = sum - tail_bytes;                              // bytes of the frames in front of j
    const uint32_t nraw = nzero - (uint32_t)(nchunks - j);                    // stored chunks among them
    uint8_t* const body0 = out + t0 + 11;
    if (blockIdx.x == 0 && tid == 0) {
        tail_info[0] = j; tail_info[1] = head_bytes; tail_info[2] = nraw; tail_info[3] = sum;
        frame_off[nchunks] = sum;
    }
    // ---- offsets of this workgroup's chunks (a handful: one lane walks them) ----
    const uint32_t nown = (uint32_t)(k_first < nchunks ? (nchunks - k_first < cpb ? nchunks - k_first : cpb) : 0u);
    if (tid == 0) {
        uint64_t o = before;
        for (uint32_t i = 0; i < nown; ++i) {
            const uint64_t k = k_first + i;
            const uint64_t left = total - k * chunk;
            const uint64_t nk = left < chunk ? left : chunk;
            own_off[i] = o;
            frame_off[k] = o;
            o += 15u + (own_c[i] ? own_c[i] : nk);
        }
    }
    __syncthreads();
    // ---- marks of the stored chunks that end the stream: final where they stand ----
    if (tid < nown && k_first + tid >= j) {
        const uint64_t k = k_first + tid;
        const uint64_t left = total - k * chunk;
        const uint32_t nk = (uint32_t)(left < chunk ? left : chunk);
        uint8_t* b = body0 + k * in_stride;
        const uint32_t field = nk | 0x80000000u;
        b[-11] = 0x04; b[-10] = 0x22; b[-9] = 0x4D; b[-8] = 0x18; b[-7] = 0x40; b[-6] = (uint8_t)bd_byte; b[-5] = (uint8_t)hc_byte;
        b[-4] = (uint8_t)field; b[-3] = (uint8_t)(field >> 8); b[-2] = (uint8_t)(field >> 16); b[-1] = (uint8_t)(field >> 24);
        b[nk] = 0; b[nk + 1] = 0; b[nk + 2] = 0; b[nk + 3] = 0;
    }
    if (nraw != 0u) {                                                          // (uniform over the grid) the stash pass has to run first
        if (blockIdx.x == 0 && tid == 0) { record[4] = j; record[5] = nraw; record[0] = 4; }
        return;
    }
    // ---- the frames in front of j end where frame j begins ----
    uint8_t* const dst0 = out + t0 + j * in_stride - head_bytes;
    for (uint32_t i = 0; i < nown; ++i) {
        const uint64_t k = k_first + i;
        if (k >= j) break;
        const uint32_t c = own_c[i];                                           // (!= 0: no stored chunk in front of j)
        const uint8_t* __restrict__ s = scratch + (uint64_t)own_ks[i] * stride;
        uint8_t* __restrict__ d = dst0 + own_off[i];
        if (tid < 15u) {
            uint8_t v = 0;
            uint32_t o = tid;
            switch (tid) {
                case 0: v = 0x04; break; case 1: v = 0x22; break; case 2: v = 0x4D; break; case 3: v = 0x18; break;
                case 4: v = 0x40; break; case 5: v = (uint8_t)bd_byte; break; case 6: v = (uint8_t)hc_byte; break;
                case 7: v = (uint8_t)c; break; case 8: v = (uint8_t)(c >> 8); break;
                case 9: v = (uint8_t)(c >> 16); break; case 10: v = (uint8_t)(c >> 24); break;
                default: v = 0; o = 11u + c + (tid - 11u); break;               // end mark
            }
            d[o] = v;
        }
        uint8_t* __restrict__ dd = d + 11;
        const uint8_t* __restrict__ ss = s;
        uint32_t len = c;
        const uint32_t head0 = (uint32_t)((16 - (reinterpret_cast<uintptr_t>(dd) & 15)) & 15);
        const uint32_t head = head0 < len ? head0 : len;
        if (tid < head) dd[tid] = ss[tid];
        dd += head; ss += head; len -= head;
        const uint32_t nvec = len >> 4;
        for (uint32_t i0 = tid; i0 < nvec; i0 += TAIL_THREADS * 4u) {
            uint4 v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) if (i0 + u * TAIL_THREADS < nvec) v[u] = ld_u128(ss + (size_t)(i0 + u * TAIL_THREADS) * 16);
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) if (i0 + u * TAIL_THREADS < nvec) *reinterpret_cast<uint4*>(dd + (size_t)(i0 + u * TAIL_THREADS) * 16) = v[u];
        }
        const uint32_t done = nvec << 4;
        if (tid < len - done) dd[done + tid] = ss[done + tid];
    }
    // ---- the sqy header in front of the payload, and what the host wants to know ----
    if (blockIdx.x != 0) return;
    const uint64_t payload = sum;
    const uint64_t payload_at = t0 + j * in_stride - head_bytes;
    char digits[20];
    uint32_t nd = 0;
    {
        uint64_t v = payload;
        do { digits[nd++] = (char)('0' + v % 10); v /= 10; } while (v);       // (least significant first)
    }
    const uint64_t text = (uint64_t)hp.prefix_len + nd + hp.suffix_len;
    const uint64_t pad = (elem_size - text % elem_size) % elem_size;           // sqeazy_header.hpp:172-178: the header's size is a multiple of the voxel's
    const uint64_t hdr_len = text + pad;
    if (hdr_len > payload_at) {
        if (tid == 0) record[0] = 3;
        return;
    }
    uint8_t* h = out + payload_at - hdr_len;
    for (uint64_t i = tid; i < hdr_len; i += TAIL_THREADS) {
        uint8_t cch;
        if (i < pad) cch = ' ';
        else if (i < pad + hp.prefix_len) cch = (uint8_t)hp.text[i - pad];
        else if (i < pad + hp.prefix_len + nd) cch = (uint8_t)digits[nd - 1 - (i - pad - hp.prefix_len)];
        else cch = (uint8_t)hp.text[hp.prefix_len + (i - pad - hp.prefix_len - nd)];
        h[i] = cch;
    }
    if (tid == 0) {
        record[1] = payload_at - hdr_len;
        record[2] = hdr_len + payload;
        record[3] = payload;
        record[4] = j;
        record[5] = 0;
        record[6] = 0;
        record[0] = 1;
    }
}

hipError_t launch_lz4_inplace_tail_fused(uint8_t* out, uint64_t t0, uint64_t in_stride, uint64_t total, uint32_t chunk, uint64_t nchunks,
                                         const uint8_t* scratch, uint64_t stride, const uint32_t* csize, uint64_t* frame_off, const uint32_t* dup_of,
                                         uint64_t* tail_info, uint32_t bd_byte, uint32_t hc_byte, const char* hdr_prefix, uint32_t prefix_len,
                                         const char* hdr_suffix, uint32_t suffix_len, uint32_t elem_size, const uint32_t* guard, uint64_t* record,
                                         hipStream_t stream)
{
    Lz4HeaderParts hp;
    if ((uint64_t)prefix_len + suffix_len > sizeof(hp.text) || nchunks == 0 || nchunks > 0x7fffffffull) return hipErrorInvalidValue;
    hp.prefix_len = prefix_len; hp.suffix_len = suffix_len;
    std::memcpy(hp.text, hdr_prefix, prefix_len);
    std::memcpy(hp.text + prefix_len, hdr_suffix, suffix_len);
    // a workgroup owns up to 64 chunks; as many workgroups as that takes, at least a few per CU's worth of small calls
    uint32_t cpb = (uint32_t)((nchunks + 1023u) / 1024u);
    if (cpb < 4u) cpb = 4u;
    if (cpb > 64u) return hipErrorInvalidValue;                                // (more than 65536 chunks: the separate kernels)
    const unsigned grid = (unsigned)((nchunks + cpb - 1u) / cpb);
    hipLaunchKernelGGL(lz4_inplace_tail_fused_kernel, dim3(grid), dim3(TAIL_THREADS), 0, stream, out, t0, in_stride, total, chunk, nchunks, cpb, scratch,
                       stride, csize, frame_off, dup_of, tail_info, bd_byte, hc_byte, hp, elem_size, guard, record);
    return hipGetLastError();
}

hipError_t launch_lz4_inplace_tail(uint8_t* out, uint64_t t0, uint64_t in_stride, uint64_t total, uint32_t chunk, uint64_t nchunks,
                                   uint8_t* scratch, uint64_t stride, const uint32_t* csize, const uint64_t* frame_off, const uint32_t* dup_of,
                                   const uint64_t* tail_info, uint32_t bd_byte, uint32_t hc_byte, const char* hdr_prefix, uint32_t prefix_len,
                                   const char* hdr_suffix, uint32_t suffix_len, uint32_t elem_size, const uint32_t* guard, uint64_t* record,
                                   hipStream_t stream)
{
    Lz4HeaderParts hp;
    if ((uint64_t)prefix_len + suffix_len > sizeof(hp.text) || nchunks == 0) return hipErrorInvalidValue;
    hp.prefix_len = prefix_len; hp.suffix_len = suffix_len;
    std::memcpy(hp.text, hdr_prefix, prefix_len);
    std::memcpy(hp.text + prefix_len, hdr_suffix, suffix_len);
    uint8_t* body0 = out + t0 + 11;
    const uint32_t slices_stash = (chunk + 32767u) / 32768u, slices = (chunk + GATHER_SLICE - 1) / GATHER_SLICE;
    const uint64_t stash_items = nchunks * slices_stash, gather_items = nchunks * slices;
    hipLaunchKernelGGL(lz4_stash_raw_kernel, dim3((unsigned)(stash_items < 1024 ? stash_items : 1024)), dim3(256), 0, stream, body0, in_stride, total, chunk,
                       scratch, stride, csize, dup_of, slices_stash, tail_info, guard);
    hipLaunchKernelGGL(lz4_frame_gather_kernel, dim3((unsigned)(gather_items < 2048 ? gather_items : 2048)), dim3(256), 0, stream, body0, total, chunk,
                       scratch, stride, csize, frame_off, out, bd_byte, hc_byte, slices, (const uint64_t*)nullptr, (uint64_t)0, (const Lz4Block*)nullptr,
                       dup_of, in_stride, 0, tail_info, t0, guard);
    hipLaunchKernelGGL(lz4_inplace_finish_kernel, dim3(1), dim3(256), 0, stream, out, t0, in_stride, tail_info, hp, elem_size, guard, record);
    return hipGetLastError();
}

hipError_t launch_lz4_tail_marks(uint8_t* body0, uint64_t in_stride, uint64_t total, uint32_t chunk, uint64_t nchunks, uint32_t bd_byte,
                                 uint32_t hc_byte, const uint64_t* tail_info, hipStream_t stream)
{
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(lz4_tail_marks_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, stream, body0, in_stride, total, chunk,
                       nchunks, bd_byte, hc_byte, tail_info);
    return hipGetLastError();
}

hipError_t launch_lz4_stash_raw(const uint8_t* body0, uint64_t in_stride, uint64_t total, uint32_t chunk, uint8_t* scratch, uint64_t stride,
                                const uint32_t* csize, const uint32_t* dup_of, uint64_t nhead, hipStream_t stream)
{
    if (nhead == 0) return hipSuccess;
    const uint32_t slices = (chunk + 32767u) / 32768u;
    hipLaunchKernelGGL(lz4_stash_raw_kernel, dim3((unsigned)(nhead * slices)), dim3(256), 0, stream, body0, in_stride, total, chunk, scratch,
                       stride, csize, dup_of, slices, (const uint64_t*)nullptr, (const uint32_t*)nullptr);
    return hipGetLastError();
}

hipError_t launch_lz4_frame_gather(const uint8_t* in, uint64_t total, uint32_t chunk, const uint8_t* scratch, uint64_t stride,
                                   const uint32_t* csize, const uint64_t* frame_off, uint8_t* out, uint32_t bd_byte,
                                   uint32_t hc_byte, uint64_t nchunks, hipStream_t stream, const uint64_t* frame_map, uint64_t frame_bytes,
                                   const Lz4Block* blocks, const uint32_t* dup_of, uint64_t in_stride, bool raw_from_scratch)
{
    if (nchunks == 0) return hipSuccess;
    if (in_stride == 0) in_stride = chunk;
    const uint32_t slices = (chunk + GATHER_SLICE - 1) / GATHER_SLICE;      // (chunk = largest block of the list when `blocks`)
    hipLaunchKernelGGL(lz4_frame_gather_kernel, dim3((unsigned)(nchunks * slices)), dim3(256), 0, stream, in, total, chunk,
                       scratch, stride, csize, frame_off, out, bd_byte, hc_byte, slices, frame_map, frame_bytes, blocks, dup_of, in_stride,
                       raw_from_scratch ? 1 : 0, (const uint64_t*)nullptr, (uint64_t)0, (const uint32_t*)nullptr);
    return hipGetLastError();
}

hipError_t launch_histogram_u16(const uint16_t* in, uint64_t len, uint32_t* histo, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(histo, 0, 65536 * sizeof(uint32_t), stream);
    if (e != hipSuccess || len == 0) return e;
    uint64_t blocks = (len / 8 + 256 * 64 - 1) / (256 * 64);        // >= 64 loads per thread
    const uint64_t cap = (uint64_t)num_cus() * 2;                    // 64 KiB of LDS per workgroup: 2 resident per CU
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(histogram_u16_kernel, dim3((unsigned)blocks), dim3(256), HIST_WIN_BINS * sizeof(uint32_t), stream, in, len, histo);
    return hipGetLastError();
}

hipError_t launch_quantiser_apply_u16(const uint16_t* in, uint8_t* out, uint64_t len, const uint8_t* lut, hipStream_t stream)
{
    if (len == 0) return hipSuccess;
    uint64_t blocks = (len / 8 + 256 * 16 - 1) / (256 * 16);
    const uint64_t cap = (uint64_t)num_cus() * 2;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(quantiser_apply_u16_kernel, dim3((unsigned)blocks), dim3(256), 65536, stream, in, out, len, lut);
    return hipGetLastError();
}

uint64_t frame_metric_scratch_bytes(uint64_t Z, uint64_t per_frame, int elem_size)
{
    const uint64_t blk = 4096 / (uint64_t)elem_size;                     // FmBlock<T>::BLK
    const uint64_t nb = (per_frame + blk - 1) / blk;
    return Z * nb * (sizeof(uint32_t) + sizeof(FmRecord)) + 64;
}

hipError_t launch_frame_metric(const void* in, uint64_t Z, uint64_t per_frame, int elem_size, float* metric, hipStream_t stream,
                               void* scratch, uint64_t scratch_bytes, bool schar)
{
    if (Z == 0) return hipSuccess;
    if (schar) {
        if (elem_size != 1) return hipErrorInvalidValue;
        hipLaunchKernelGGL(frame_metric_i8_kernel, dim3((unsigned)((Z + 63) / 64)), dim3(64), 0, stream, (const int8_t*)in, Z, per_frame, metric);
        return hipGetLastError();
    }
    const bool vector_ok = (per_frame * (uint64_t)elem_size) % 16 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && Z <= 0x7fffffffull &&
                           per_frame * (elem_size == 2 ? 65535ull : 255ull) < (1ull << 39);
    const uint64_t blk = 4096 / (uint64_t)elem_size;
    const uint64_t nb = (per_frame + blk - 1) / blk;
    if (vector_ok && nb >= 16 && nb <= 0x7fffffffull && scratch && scratch_bytes >= frame_metric_scratch_bytes(Z, per_frame, elem_size) &&
        (Z * nb + 3) / 4 <= 0x7fffffffull) {
        // long frames: block sums, block records, then a short chain per frame
        const uint64_t nblocks = Z * nb;
        uint32_t* bsum = static_cast<uint32_t*>(scratch);
        FmRecord* rec = reinterpret_cast<FmRecord*>(static_cast<uint8_t*>(scratch) + ((nblocks * sizeof(uint32_t) + 15) & ~(uint64_t)15));
        const unsigned grid = (unsigned)((nblocks + 3) / 4);
        if (elem_size == 2) {
            hipLaunchKernelGGL((frame_block_sums_kernel<uint16_t>), dim3(grid), dim3(256), 0, stream, (const uint16_t*)in, per_frame, (uint32_t)nb, nblocks, bsum);
            hipLaunchKernelGGL((frame_block_summaries_kernel<uint16_t>), dim3(grid), dim3(256), 0, stream, (const uint16_t*)in, per_frame, (uint32_t)nb, nblocks,
                               (const uint32_t*)bsum, rec);
            hipLaunchKernelGGL((frame_chain_kernel<uint16_t>), dim3((unsigned)Z), dim3(64), 0, stream, (const uint16_t*)in, per_frame, (uint32_t)nb,
                               (const uint32_t*)bsum, (const FmRecord*)rec, metric);
        } else {
            hipLaunchKernelGGL((frame_block_sums_kernel<uint8_t>), dim3(grid), dim3(256), 0, stream, (const uint8_t*)in, per_frame, (uint32_t)nb, nblocks, bsum);
            hipLaunchKernelGGL((frame_block_summaries_kernel<uint8_t>), dim3(grid), dim3(256), 0, stream, (const uint8_t*)in, per_frame, (uint32_t)nb, nblocks,
                               (const uint32_t*)bsum, rec);
            hipLaunchKernelGGL((frame_chain_kernel<uint8_t>), dim3((unsigned)Z), dim3(64), 0, stream, (const uint8_t*)in, per_frame, (uint32_t)nb,
                               (const uint32_t*)bsum, (const FmRecord*)rec, metric);
        }
        return hipGetLastError();
    }
    if (vector_ok) {
        // one wavefront per frame, block-parallel exact emulation of the sequential float sum
        if (elem_size == 2)
            hipLaunchKernelGGL((frame_metric_scan_kernel<uint16_t>), dim3((unsigned)Z), dim3(64), 0, stream, (const uint16_t*)in, per_frame, metric);
        else
            hipLaunchKernelGGL((frame_metric_scan_kernel<uint8_t>), dim3((unsigned)Z), dim3(64), 0, stream, (const uint8_t*)in, per_frame, metric);
        return hipGetLastError();
    }
    const unsigned blocks = (unsigned)((Z + 63) / 64);
    if (elem_size == 2)
        hipLaunchKernelGGL((frame_metric_kernel<uint16_t>), dim3(blocks), dim3(64), 0, stream, (const uint16_t*)in, Z, per_frame, metric);
    else
        hipLaunchKernelGGL((frame_metric_kernel<uint8_t>), dim3(blocks), dim3(64), 0, stream, (const uint8_t*)in, Z, per_frame, metric);
    return hipGetLastError();
}

hipError_t launch_frame_gather(const void* in, void* out, uint64_t Z, uint64_t frame_bytes, const uint64_t* map, hipStream_t stream)
{
    if (Z == 0 || frame_bytes == 0) return hipSuccess;
    uint64_t bpf = (frame_bytes / 16 + 256 * 8 - 1) / (256 * 8);
    if (bpf == 0) bpf = 1;
    if (bpf > 64) bpf = 64;
    hipLaunchKernelGGL(frame_gather_kernel, dim3((unsigned)(Z * bpf)), dim3(256), 0, stream, (const uint8_t*)in, (uint8_t*)out, frame_bytes,
                       map, (uint32_t)bpf);
    return hipGetLastError();
}

hipError_t launch_raster_reorder(const void* in, void* out, uint64_t Z, uint64_t Y, uint64_t X, uint64_t tile_size, int elem_size,
                                 bool decode, hipStream_t stream)
{
    if (Z * Y * X == 0 || tile_size == 0) return hipSuccess;
    const uint64_t TX = (X + tile_size - 1) / tile_size;
    const uint64_t runs = Z * Y * TX;
    const uint64_t blocks = (runs + 255) / 256;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (elem_size == 2)
        hipLaunchKernelGGL((raster_reorder_kernel<uint16_t>), dim3((unsigned)blocks), dim3(256), 0, stream, (const uint16_t*)in, (uint16_t*)out,
                           Z, Y, X, tile_size, TX, decode);
    else
        hipLaunchKernelGGL((raster_reorder_kernel<uint8_t>), dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t*)in, (uint8_t*)out,
                           Z, Y, X, tile_size, TX, decode);
    return hipGetLastError();
}

hipError_t launch_bitshuffle(const void* in, void* out, uint64_t n_elems, int elem_size, uint64_t block_elems, bool decode, hipStream_t stream)
{
    if (n_elems == 0) return hipSuccess;
    if (block_elems == 0 || block_elems % 8) return hipErrorInvalidValue;
    uint64_t done = 0;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (elem_size == 2 && block_elems == 4096 && aligned && n_elems >= 4096) {
        const uint64_t nblocks = n_elems / 4096;
        const uint64_t want = (nblocks + 3) / 4;
        const uint64_t cap = (uint64_t)num_cus() * 16;
        hipLaunchKernelGGL(bitshuffle_u16_4096_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, stream, (const uint16_t*)in,
                           (uint16_t*)out, nblocks, decode);
        done = nblocks * 4096;
    }
    if (done < n_elems) {
        const uint64_t bytes = (n_elems - done) * (uint64_t)elem_size;
        const uint64_t blocks = (bytes + 255) / 256;
        if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(bitshuffle_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t*)in, (uint8_t*)out, done, n_elems,
                           (uint32_t)elem_size, block_elems, decode);
    }
    return hipGetLastError();
}

hipError_t launch_lz4_frame_index(const uint8_t* in, uint64_t n, void* blk, uint32_t* frame_first, uint64_t max_blocks,
                                  uint32_t* counts, hipStream_t stream)
{
    hipLaunchKernelGGL(lz4_frame_index_kernel, dim3(1), dim3(64), 0, stream, in, n, (uint4*)blk, frame_first, max_blocks, counts);
    return hipGetLastError();
}

uint64_t lz4_frame_rank_scratch_bytes(uint64_t expected_frames)
{
    const uint64_t cap = expected_frames * 4 + 1024;                     // candidates the list holds
    uint64_t slots = 1024;
    while (slots < cap * 4) slots *= 2;
    return slots * sizeof(FrameSlot) + cap * sizeof(FrameCand) + 5 * (cap + 2) * sizeof(uint32_t) + 64;
}

hipError_t launch_lz4_frame_rank(const uint8_t* in, uint64_t n, void* blk, uint32_t* frame_first, uint64_t max_blocks,
                                 uint32_t* counts, uint64_t expected_frames, void* scratch, hipStream_t stream, uint64_t chunk, uint64_t last)
{
    const uint64_t cap = expected_frames * 4 + 1024;
    uint64_t slots = 1024;
    while (slots < cap * 4) slots *= 2;
    uint8_t* p = static_cast<uint8_t*>(scratch);
    FrameSlot* table = reinterpret_cast<FrameSlot*>(p);
    FrameCand* list = reinterpret_cast<FrameCand*>(p + slots * sizeof(FrameSlot));
    uint32_t* work = reinterpret_cast<uint32_t*>(p + slots * sizeof(FrameSlot) + cap * sizeof(FrameCand));
    uint32_t* ncand = work + 5 * (cap + 2);
    hipError_t e = hipMemsetAsync(table, 0, slots * sizeof(FrameSlot), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ncand, 0, 4, stream);
    if (e != hipSuccess) return e;
    const uint64_t nvec = (n + 15) / 16;
    uint64_t blocks = (nvec + 1023) / 1024;                               // a thread takes four vectors at a time, a wavefront 4 KiB
    const uint64_t gcap = (uint64_t)num_cus() * 4;                        // (16 wavefronts per CU, all resident, each with its next piece in flight: 128 KiB per CU)
    if (blocks > gcap) blocks = gcap;
    if (blocks == 0) blocks = 1;
    uint32_t* hint = counts + 5;                                          // (counts: [0..3] the index, [4] the decoders' flag, [5] this)
    uint32_t* tail_frames = counts + 6;                                   // [6] frames of the stored tail the scan skipped, [8..9] where the scan ends
    unsigned long long* scan_end = reinterpret_cast<unsigned long long*>(counts + 8);
    hipLaunchKernelGGL(lz4_frame_probe_kernel, dim3(1), dim3(1), 0, stream, in, n, hint);
    hipLaunchKernelGGL(lz4_frame_tail_kernel, dim3(1), dim3(1024), 0, stream, in, n, chunk, last, (uint32_t)std::min<uint64_t>(expected_frames, 1u << 20), table,
                       (uint32_t)(slots - 1), list, (uint32_t)cap, ncand, (const uint32_t*)hint, tail_frames, scan_end);
    hipLaunchKernelGGL(lz4_frame_candidates_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, n, table, (uint32_t)(slots - 1), list,
                       (uint32_t)cap, ncand, (const uint32_t*)hint, (const unsigned long long*)scan_end);
    // LDS for the rank kernel's work arrays: 9 bytes per candidate (the frames expected + room for false magics), when that fits
    uint64_t lds_entries = (expected_frames + 1024 + 2 + 7) & ~(uint64_t)7;
    if (lds_entries > 16000) lds_entries = 0;                            // (144 KiB of the CU's 160)
    const size_t lds_bytes = (size_t)lds_entries * 9;
    if (lds_bytes > 48 * 1024) {
        static std::atomic<size_t> allowed{0};
        if (allowed.load() < lds_bytes) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_frame_rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            if (e != hipSuccess) { (void)hipGetLastError(); lds_entries = 0; }
            else allowed.store(144 * 1024);
        }
    }
    hipLaunchKernelGGL(lz4_frame_rank_kernel, dim3(1), dim3(1024), lds_entries ? lds_bytes : 0, stream, in, n, (const FrameSlot*)table, (uint32_t)(slots - 1),
                       (const FrameCand*)list, (uint32_t)cap, (const uint32_t*)ncand, work, (uint4*)blk, frame_first, max_blocks, counts,
                       (uint32_t)lds_entries, (const uint32_t*)hint);
    return hipGetLastError();
}

constexpr uint32_t SQY_RING8_MIN = 2560;
hipError_t launch_lz4_frames_decode(const uint8_t* in, const void* blk, const uint32_t* frame_first, uint32_t nframes, uint8_t* out,
                                    uint64_t out_bytes, uint64_t frame_stride, uint64_t block_bytes, uint32_t ncompressed,
                                    uint32_t* errflag, hipStream_t stream, hipStream_t copy_stream, hipEvent_t fork, hipEvent_t join,
                                    const uint64_t* remap, uint64_t remap_bytes, bool two_waves)
{
    if (nframes == 0) return hipSuccess;
    // (remap: frame f goes to remap[f * stride / remap_bytes] * remap_bytes + the rest -- whole chunks inside whole shuffle frames only)
    if (remap && (remap_bytes == 0 || frame_stride == 0 || remap_bytes % frame_stride != 0 || out_bytes % remap_bytes != 0 ||
                  (uint64_t)nframes * frame_stride != out_bytes)) return hipErrorInvalidValue;
    // the stored frames are copied (HBM-bound) while the compressed ones are decoded (latency-bound, HBM idle): disjoint
    // outputs, both only read the stream -- on a second stream when the caller has one to spare.  The decode kernel is
    // launched first: its few long-running waves should get their slots before the copy's many short workgroups fill the CUs.
    const bool side = copy_stream && copy_stream != stream && fork && join;
    hipStream_t cs = side ? copy_stream : stream;
    if (side) {
        hipError_t e = hipEventRecord(fork, stream);
        if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(copy_stream, fork, 0);
        if (e != hipSuccess) return e;
    }
    // more compressed blocks than the 64 KiB-ring kernel keeps resident (2 per CU) plus half a round: the small ring's four-fold
    // occupancy wins; below that the frames are few and long, and every match served from LDS wins
    // (round 4) more compressed frames than even the 16 KiB-ring kernel keeps resident (8 per CU) plus a quarter: the 8 KiB ring's 13 waves
    // per CU win although more matches reach behind the ring -- the C3 slab's 3584 frames 3.37 -> 2.65 ms (a 4 KiB ring: no better)
    // (round 5) frames of one block, few enough for every one of them to be resident: two wavefronts per frame, one finds out what the
    // sequences are, the other moves the bytes (lz4_frames_decode2_kernel)
    // (not beyond that: the C3 slab's 3584 frames through two waves each, 8 KiB ring, take 7.1 ms against 2.6 -- that range is bound by the
    // instructions issued, and two waves issue more of them)
    if (two_waves && !(ncompressed > SQY_RING8_MIN && nframes > SQY_RING8_MIN)) {
        // (a 32 KiB ring here would serve nine in ten of the matches that reach behind 16 KiB -- a seventh of the bench stack's -- out of
        // LDS, but only three frames fit a CU then: measured, 0.63 -> 0.73 ms)
        if (ncompressed > 768u && nframes > 768u)
            hipLaunchKernelGGL(lz4_frames_decode2_kernel<16384>, dim3(nframes), dim3(128), 0, stream, in, (const uint4*)blk, frame_first, out,
                               out_bytes, frame_stride, block_bytes, errflag, remap, remap_bytes);
        else
            hipLaunchKernelGGL(lz4_frames_decode2_kernel<65536>, dim3(nframes), dim3(128), 0, stream, in, (const uint4*)blk, frame_first, out,
                               out_bytes, frame_stride, block_bytes, errflag, remap, remap_bytes);
    } else
    if (ncompressed > SQY_RING8_MIN && nframes > SQY_RING8_MIN)
        hipLaunchKernelGGL(lz4_frames_decode_kernel<8192>, dim3(nframes), dim3(64), 0, stream, in, (const uint4*)blk, frame_first, out,
                           out_bytes, frame_stride, block_bytes, errflag, remap, remap_bytes);
    else if (ncompressed > 768u && nframes > 768u)
        hipLaunchKernelGGL(lz4_frames_decode_kernel<16384>, dim3(nframes), dim3(64), 0, stream, in, (const uint4*)blk, frame_first, out,
                           out_bytes, frame_stride, block_bytes, errflag, remap, remap_bytes);
    else
        hipLaunchKernelGGL(lz4_frames_decode_kernel<65536>, dim3(nframes), dim3(64), 0, stream, in, (const uint4*)blk, frame_first, out,
                           out_bytes, frame_stride, block_bytes, errflag, remap, remap_bytes);
    {   // stored blocks of single-block frames (a block never exceeds block_bytes)
        const uint32_t slices = (uint32_t)((block_bytes + DEC_COPY_SLICE - 1) / DEC_COPY_SLICE);
        if (slices == 0 || (uint64_t)nframes * slices > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(lz4_stored_frames_copy_kernel, dim3(nframes * slices), dim3(256), 0, cs, in, (const uint4*)blk, frame_first,
                           out, out_bytes, frame_stride, slices, remap, remap_bytes);
    }
    if (side) {
        hipError_t e = hipEventRecord(join, copy_stream);
        if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(stream, join, 0);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

// ONE block-linked frame, block-parallel (see lz4_blocks_decode_sym_kernel): blk = the frame's block list (lz4_frame_index), refs =
// out_bytes 16-bit words of scratch.  *errflag != 0 afterwards: the stream is not a frame of full blocks, or is damaged -- decode it
// again with the one-wavefront walk, whose result and error codes count.
bool lz4_linked_decode_parallel_possible(uint32_t nblocks, uint64_t out_bytes, uint64_t block_bytes)
{
    return nblocks >= 3 && nblocks <= 32768u && block_bytes >= SYM_TAIL && block_bytes % 1024u == 0 && block_bytes <= (4u << 20) &&
           out_bytes > (uint64_t)(nblocks - 1) * block_bytes && out_bytes <= (uint64_t)nblocks * block_bytes;
}

uint64_t lz4_linked_decode_scan_scratch_bytes(uint32_t nblocks)
{
    const uint64_t nranges = ((uint64_t)nblocks + SYM_RANGE - 2u) / SYM_RANGE;
    return nranges * ((uint64_t)SYM_MAP_BYTES + SYM_TAIL);
}

hipError_t launch_lz4_linked_decode_parallel(const uint8_t* in, const void* blk, uint32_t nblocks, uint8_t* out, uint16_t* refs, uint64_t out_bytes,
                                             uint64_t block_bytes, uint32_t* errflag, hipStream_t stream, uint8_t* scan_scratch)
{
    if (!lz4_linked_decode_parallel_possible(nblocks, out_bytes, block_bytes) || !refs) return hipErrorInvalidValue;
    // (per call: the attribute belongs to the current device's copy of the kernel)
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_linked_resolve_tails_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SYM_TAIL);
    if (attr != hipSuccess) return attr;
    const uint32_t slices = (uint32_t)((block_bytes + DEC_COPY_SLICE - 1) / DEC_COPY_SLICE);
    hipLaunchKernelGGL(lz4_blocks_decode_sym_kernel, dim3(nblocks), dim3(64), 0, stream, in, (const uint4*)blk, out, refs, out_bytes, block_bytes, errflag);
    hipLaunchKernelGGL(lz4_linked_stored_copy_kernel, dim3(nblocks * slices), dim3(256), 0, stream, in, (const uint4*)blk, out, out_bytes, block_bytes, slices, errflag);
    // the tails: one walk, or (long frames, scratch given) ranges of SYM_RANGE blocks composed at once, chained, and walked at once
    const uint32_t nranges = (nblocks - 1u + SYM_RANGE - 1u) / SYM_RANGE;
    if (scan_scratch && nranges >= 4u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_linked_compose_tails_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SYM_MAP_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_linked_chain_ranges_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SYM_TAIL);
        if (e != hipSuccess) return e;
        uint8_t* maps = scan_scratch;
        uint8_t* starts = scan_scratch + (uint64_t)nranges * SYM_MAP_BYTES;
        hipLaunchKernelGGL(lz4_linked_compose_tails_kernel, dim3(nranges), dim3(1024), SYM_MAP_BYTES, stream, (const uint4*)blk, nblocks, (const uint8_t*)out,
                           (const uint16_t*)refs, block_bytes, SYM_RANGE, maps);
        hipLaunchKernelGGL(lz4_linked_chain_ranges_kernel, dim3(1), dim3(1024), 2 * SYM_TAIL, stream, (const uint8_t*)maps, nranges, starts);
        hipLaunchKernelGGL(lz4_linked_resolve_tails_kernel, dim3(nranges), dim3(1024), 2 * SYM_TAIL, stream, (const uint4*)blk, nblocks, out, (const uint16_t*)refs,
                           out_bytes, block_bytes, SYM_RANGE, (const uint8_t*)starts);
    } else
        hipLaunchKernelGGL(lz4_linked_resolve_tails_kernel, dim3(1), dim3(1024), 2 * SYM_TAIL, stream, (const uint4*)blk, nblocks, out, (const uint16_t*)refs,
                           out_bytes, block_bytes, 0u, (const uint8_t*)nullptr);
    const uint32_t pieces = 16;
    hipLaunchKernelGGL(lz4_linked_resolve_bodies_kernel, dim3((nblocks - 1) * pieces), dim3(256), 0, stream, (const uint4*)blk, nblocks, out, (const uint16_t*)refs,
                       out_bytes, block_bytes, pieces);
    return hipGetLastError();
}

// true when the fused kernel took the job (8-bit planes of whole 16-byte vectors, aligned buffers); false: run the two stages one by one
bool bitswap1_u8_decode_lut_possible(const void* in, const void* out, uint64_t len)
{
    const uint64_t seg = len / 8;
    return len != 0 && len % 128 == 0 && seg % 16 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
}

hipError_t launch_bitswap1_u8_decode_lut(const uint8_t* in, uint16_t* out, uint64_t len, const uint16_t* lut, hipStream_t stream)
{
    const uint64_t seg = len / 8, nvec = seg / 16;
    uint64_t g = (nvec + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(bitswap1_u8_decode_lut_kernel, dim3((unsigned)g), dim3(256), 0, stream, in, out, nvec, seg, lut);
    return hipGetLastError();
}

hipError_t launch_bitswap1_decode(const void* in, void* out, uint64_t len, int elem_size, hipStream_t stream)
{
    if (len == 0) return hipSuccess;
    const uint64_t W = (uint64_t)elem_size * 8, seg = len / W;
    uint64_t blocks = (seg + 255) / 256;
    if (blocks == 0) blocks = 1;
    if (elem_size == 2) {
        // whole tiles of 8192 voxels through the register kernel (16-byte aligned buffers and plane segments), the rest one group per thread
        uint64_t w0 = 0;
        const uint64_t n_tiles = seg / (BSW_TILE_VOX / 16);
        if (n_tiles && seg % 8 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
            uint64_t g = (n_tiles + 3) / 4;
            if (g > 65536) g = 65536;
            hipLaunchKernelGGL(bitswap1_decode_u16_regs, dim3((unsigned)g), dim3(256), 0, stream, (const uint16_t*)in, (uint16_t*)out, n_tiles, seg);
            w0 = n_tiles * (BSW_TILE_VOX / 16);
        }
        const uint64_t rest = seg - w0;
        uint64_t rb = (rest + 255) / 256;
        if (rb == 0) rb = 1;                                            // (block 0 also moves the len % 16 voxels behind the planes)
        hipLaunchKernelGGL((bitswap1_decode_kernel<uint16_t>), dim3((unsigned)rb), dim3(256), 0, stream, (const uint16_t*)in, (uint16_t*)out, len, seg, w0);
    } else
        hipLaunchKernelGGL((bitswap1_decode_kernel<uint8_t>), dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t*)in, (uint8_t*)out, len, seg, (uint64_t)0);
    return hipGetLastError();
}

// scratch for the one-launch kernel: 256 strips x 2 edges x 2 frame parities x ceil(X / 3) words, and the abort word
uint64_t diff3x3x1_decode_scratch_bytes(uint64_t X) { (void)X; return 0; }      // (the frame chain below needs none)

// the usual geometry of the inverse (16-bit, every row's reach inside its row): the columns that go through the chain of launches --
// the ones the stage can touch and their right-hand neighbour, whole 16-byte vectors; 0: another geometry
uint64_t diff3x3x1_decode_chain_columns(uint64_t Z, uint64_t Y, uint64_t X, int elem_size)
{
    if (Z * Y * X == 0) return 0;
    const uint64_t zlim = X < Z ? X : Z;
    const uint64_t noff = (zlim >= 1 ? (zlim - 1) : 0) * (Y >= 2 ? (Y - 2) : 0);
    const bool single = (noff == 1);
    const uint64_t hx = single ? 0 : (Z >= 2 ? Z - 2 : 0);
    if (!(elem_size == 2 && !single && hx + 2 <= X && Y <= 65535 && Z <= 65535 && X <= 0xffffffffull && X % 8 == 0)) return 0;
    const uint64_t w = ((2 + hx) + 7) / 8 * 8;
    return w > X ? X : w;
}

hipError_t launch_diff3x3x1_decode(const void* in, void* out, uint64_t Z, uint64_t Y, uint64_t X, int elem_size, void* scratch,
                                   hipStream_t stream, bool schar, hipStream_t copy_stream, hipEvent_t fork, hipEvent_t join, void* left_tmp)
{
    (void)scratch;
    const uint64_t length = Z * Y * X, frame = Y * X;
    if (length == 0) return hipSuccess;
    const uint64_t zlim = X < Z ? X : Z;
    const uint64_t noff = (zlim >= 1 ? (zlim - 1) : 0) * (Y >= 2 ? (Y - 2) : 0);
    const int single = (noff == 1);
    const uint64_t hx = single ? 0 : (Z >= 2 ? Z - 2 : 0);
    // left_tmp (in == out, the volume decoded where it lies): only promised for the geometry below
    if (left_tmp && !(in == out && diff3x3x1_decode_chain_columns(Z, Y, X, elem_size) &&
                      ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(left_tmp)) & 15) == 0)) return hipErrorInvalidValue;
    // The usual geometry (16-bit, every row's reach inside its row): frame z needs the DECODED frame z-1 -- a chain of one launch
    // per frame, in stream order.  Only columns x < 1 + hx can change (hx from the depth of the stack): those go
    // through the chain (1 MiB per frame of a 2048 x 2048 x 256 slab: launch-bound, ~3.5 us each), everything to the right of
    // them is ONE plain copy of all frames.  No workgroup waits for another one.
    if (elem_size == 2 && !single && hx + 2 <= X && Y <= 65535 && Z <= 65535 && X <= 0xffffffffull && X % 8 == 0 &&
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        // columns of the chain, whole 16-byte vectors: the touched ones and their right-hand neighbour (read from the frame before),
        // so that the chain never reads what the copy writes
        uint64_t w = ((2 + hx) + 7) / 8 * 8;
        if (w > X) w = X;
        uint32_t rpb = 1;
        while (rpb < 16u && (w / 8u) * rpb * 2u <= 256u) rpb *= 2u;
        const unsigned bx = (unsigned)((w / 8u + (256u / rpb) - 1u) / (256u / rpb));
        const dim3 grid(bx, (unsigned)((Y + rpb - 1) / rpb), 1);
        if (left_tmp) {
            // decoded where it lies (in == out; round 4): nothing to the right of the chain's columns has to move at all.  The chain's
            // strips read encoded rows of their neighbours' strips -- which those neighbours overwrite -- so the encoded LEFT columns go
            // to left_tmp first (13 % of a 2048-wide slab) and the chain reads them there.
            hipLaunchKernelGGL(diff3x3x1_u16_rows_kernel<true>, dim3(grid.x, grid.y, (unsigned)Z), dim3(256), 0, stream, (const uint16_t*)in, (uint16_t*)left_tmp,
                               (uint32_t)Y, (uint32_t)X, 0u, 0u, (uint32_t)X, (uint32_t)w, rpb, 0u, 0u);
            in = left_tmp;
        }
        hipStream_t cs = stream;
        if (w < X && !left_tmp && copy_stream && fork && join && hipEventRecord(fork, stream) == hipSuccess && hipStreamWaitEvent(copy_stream, fork, 0) == hipSuccess)
            cs = copy_stream;                                                 // (the copy runs next to the chain)
        if (w < X && !left_tmp) {
            // columns [w, X) of every frame: copy
            const uint64_t cols = X - w;
            uint32_t rpb = 1;
            while (rpb < 16u && (cols / 8u) * rpb * 2u <= 256u) rpb *= 2u;
            const unsigned bx = (unsigned)((cols / 8u + (256u / rpb) - 1u) / (256u / rpb));
            hipLaunchKernelGGL(diff3x3x1_u16_rows_kernel<true>, dim3(bx, (unsigned)((Y + rpb - 1) / rpb), (unsigned)Z), dim3(256), 0, cs,
                               (const uint16_t*)in, (uint16_t*)out, (uint32_t)Y, (uint32_t)X, 0u, 0u, (uint32_t)X, (uint32_t)X, rpb, 0u, (uint32_t)w);
        }
        // frames that cannot change (z = 0, z >= zlim) in one launch each run, the others one by one
        hipLaunchKernelGGL(diff3x3x1_u16_rows_kernel<true>, grid, dim3(256), 0, stream, (const uint16_t*)in, (uint16_t*)out, (uint32_t)Y, (uint32_t)X,
                           (uint32_t)hx, (uint32_t)zlim, (uint32_t)X, (uint32_t)w, rpb, 0u, 0u);
        // frames 1 .. zlim-1: K frames per launch, strips of R rows that recompute the halo rows they need of their neighbours
        // (diff3x3x1_u16_decode_frames_kernel); narrow stripes of columns only -- the two LDS images must fit
        const size_t ddk_lds = 2 * (size_t)DDK_ROWS * (w + 16) * sizeof(uint16_t);
        if (ddk_lds <= (64u << 10) && (uint64_t)DDK_ROWS * (w / 8 + 1) <= (uint64_t)DDK_ITEMS * DDK_THREADS) {
            for (uint64_t z = 1; z < zlim; z += DDK_K) {
                const uint32_t nf = (uint32_t)(zlim - z < DDK_K ? zlim - z : DDK_K);
                hipLaunchKernelGGL(diff3x3x1_u16_decode_frames_kernel, dim3((unsigned)((Y + DDK_R - 1) / DDK_R)), dim3(DDK_THREADS), ddk_lds, stream, (const uint16_t*)in,
                                   (uint16_t*)out, (uint32_t)Y, (uint32_t)X, (uint32_t)hx, (uint32_t)zlim, (uint32_t)w, (uint32_t)z, nf);
            }
        } else
        for (uint64_t z = 1; z < zlim; ++z)
            hipLaunchKernelGGL(diff3x3x1_u16_rows_kernel<true>, grid, dim3(256), 0, stream, (const uint16_t*)in, (uint16_t*)out, (uint32_t)Y, (uint32_t)X,
                               (uint32_t)hx, (uint32_t)zlim, (uint32_t)X, (uint32_t)w, rpb, (uint32_t)z, 0u);
        if (zlim < Z)
            hipLaunchKernelGGL(diff3x3x1_u16_rows_kernel<true>, dim3(grid.x, grid.y, (unsigned)(Z - zlim)), dim3(256), 0, stream, (const uint16_t*)in,
                               (uint16_t*)out, (uint32_t)Y, (uint32_t)X, (uint32_t)hx, (uint32_t)zlim, (uint32_t)X, (uint32_t)w, rpb, (uint32_t)zlim, 0u);
        if (cs != stream) {
            hipError_t e = hipEventRecord(join, cs);
            if (e == hipSuccess) e = hipStreamWaitEvent(stream, join, 0);
            if (e != hipSuccess) return e;
        }
        return hipGetLastError();
    }
    // per frame; the last X + 1 voxels of a frame read the frame's own first voxels when rows spill over (see the kernel)
    const bool spills = single || hx + 1 >= X;      // (hx = X-1: voxel X-1 of row Y-2 already reads voxel 0 of its own frame)
    for (uint64_t z = 0; z < Z; ++z) {
        const uint64_t split = (spills && frame > X + 1) ? frame - X - 1 : frame;
        for (int part = 0; part < 2; ++part) {
            const uint64_t r0 = part ? split : 0, r1 = part ? frame : split;
            if (r0 >= r1) continue;
            const unsigned blocks = (unsigned)((r1 - r0 + 255) / 256);
            if (elem_size == 2)
                hipLaunchKernelGGL((diff3x3x1_decode_plane_kernel<uint16_t, int16_t>), dim3(blocks), dim3(256), 0, stream, (const uint16_t*)in,
                                   (uint16_t*)out, z, length, Y, X, hx, zlim, single, r0, r1);
            else if (schar)
                hipLaunchKernelGGL((diff3x3x1_decode_plane_kernel<uint8_t, int8_t, true>), dim3(blocks), dim3(256), 0, stream, (const uint8_t*)in,
                                   (uint8_t*)out, z, length, Y, X, hx, zlim, single, r0, r1);
            else
                hipLaunchKernelGGL((diff3x3x1_decode_plane_kernel<uint8_t, int8_t>), dim3(blocks), dim3(256), 0, stream, (const uint8_t*)in,
                                   (uint8_t*)out, z, length, Y, X, hx, zlim, single, r0, r1);
        }
    }
    return hipGetLastError();
}

hipError_t launch_quantiser_decode(const uint8_t* in, uint16_t* out, uint64_t len, const uint16_t* lut, hipStream_t stream)
{
    if (len == 0) return hipSuccess;
    uint64_t blocks = (len + 256 * 32 - 1) / (256 * 32);
    const uint64_t cap = (uint64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(quantiser_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, len, lut);
    return hipGetLastError();
}

hipError_t launch_frame_scatter(const void* in, void* out, uint64_t Z, uint64_t frame_bytes, const uint64_t* map, hipStream_t stream)
{
    if (Z == 0 || frame_bytes == 0) return hipSuccess;
    uint64_t bpf = (frame_bytes + 256 * 64 - 1) / (256 * 64);
    if (bpf == 0) bpf = 1;
    if (bpf > 64) bpf = 64;
    hipLaunchKernelGGL(frame_scatter_kernel, dim3((unsigned)(Z * bpf)), dim3(256), 0, stream, (const uint8_t*)in, (uint8_t*)out, frame_bytes,
                       map, (uint32_t)bpf);
    return hipGetLastError();
}

} // namespace sqy
