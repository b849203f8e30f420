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
#include <stdint.h>

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

__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
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
constexpr int BSW_ROW_PITCH = 272;                // 256 B of payload + 16 B pad
constexpr int BSW_LDS_PER_WAVE = 64 * BSW_ROW_PITCH;
constexpr int BSW_WAVES = 4;

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

__global__ __launch_bounds__(64 * BSW_WAVES)
void bitswap1_u16_tiles(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                        uint64_t n_tiles, uint64_t seg_words /* S */)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint8_t* lds = lds_raw + wave * BSW_LDS_PER_WAVE;

    const uint64_t wave_global = (uint64_t)blockIdx.x * BSW_WAVES + wave;
    const uint64_t wave_stride = (uint64_t)gridDim.x * BSW_WAVES;

    for (uint64_t tile = wave_global; tile < n_tiles; tile += wave_stride) {
        const v4u* src = reinterpret_cast<const v4u*>(in + tile * BSW_TILE_VOX);
        v4u v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(src + j * 64 + lane);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = j * 4 + (lane >> 4);
            *reinterpret_cast<v4u*>(lds + row * BSW_ROW_PITCH + (lane & 15) * 16) = v[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // plane words for this lane: pl[b][q], q = 0..3 -> (group 2q | group 2q+1 << 16)
        uint32_t pl[16][4];
        const uint8_t* rowp = lds + lane * BSW_ROW_PITCH;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // groups 2q (voxels 32q..32q+15) and 2q+1 (voxels 32q+16..32q+31) of this lane's row
            const uint4 a0 = *reinterpret_cast<const uint4*>(rowp + q * 64);
            const uint4 a1 = *reinterpret_cast<const uint4*>(rowp + q * 64 + 16);
            const uint4 b0 = *reinterpret_cast<const uint4*>(rowp + q * 64 + 32);
            const uint4 b1 = *reinterpret_cast<const uint4*>(rowp + q * 64 + 48);
            const uint32_t ga[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}; // voxel pairs of group A
            const uint32_t gb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            uint32_t r[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // voxel 2k -> row 15-2k, voxel 2k+1 -> row 14-2k
                r[15 - 2 * k] = __builtin_amdgcn_perm(gb[k], ga[k], 0x05040100u); // lo16(A) | lo16(B) << 16
                r[14 - 2 * k] = __builtin_amdgcn_perm(gb[k], ga[k], 0x07060302u); // hi16(A) | hi16(B) << 16
            }
            transpose16x16_pairs(r);
#pragma unroll
            for (int b = 0; b < 16; ++b) pl[b][q] = r[b];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // plane b lives in segment 15-b; this tile contributes 512 words (1 KiB) per segment
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            v4u* dst = reinterpret_cast<v4u*>(out + (uint64_t)(15 - b) * seg_words + tile * (BSW_TILE_VOX / 16));
            const v4u val = {pl[b][0], pl[b][1], pl[b][2], pl[b][3]};
            __builtin_nontemporal_store(val, dst + lane);
        }
    }
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
// TODO(perf): LDS-tiled variant like the 16-bit kernel.
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

template <typename T, typename ST>
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
            // 8-bit: sum_type = unsigned short of a `char`-typed... the C-ABI only feeds unsigned pixel types here
            const uint32_t mean = (uint32_t)sum / 9u;
            v = (T)(ST)((uint32_t)v - mean);
        }
        out[idx] = v;
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
// ------------------------------------------------------------------------------------------------
constexpr uint32_t LZ4_MFLIMIT = 12, LZ4_LASTLITERALS = 5, LZ4_MINLENGTH = 13, LZ4_MAXD = 65535;

__device__ __forceinline__ uint32_t lz4_hash5(uint64_t seq)
{
    return (uint32_t)(((seq << 24) * 889523592379ULL) >> 52);
}

// copy `len` bytes (uniform) from s to d, any alignment, no overlap; all 64 lanes call it
__device__ __forceinline__ void wave_copy(uint8_t* __restrict__ d, const uint8_t* __restrict__ s, uint32_t len, int lane)
{
    if (len <= 64) {
        if ((uint32_t)lane < len) d[lane] = s[lane];
        return;
    }
    // head: bring d to 16-byte alignment
    const uint32_t head = (uint32_t)((16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15);
    if ((uint32_t)lane < head) d[lane] = s[lane];
    d += head; s += head; len -= head;
    const uint32_t nvec = len >> 4;
    for (uint32_t i = lane; i < nvec; i += 64) {
        *reinterpret_cast<uint4*>(d + (size_t)i * 16) = ld_u128(s + (size_t)i * 16);
    }
    const uint32_t done = nvec << 4;
    if ((uint32_t)lane < len - done) d[done + lane] = s[done + lane];
}

// number of equal leading bytes of a[0..maxlen) and b[0..maxlen), maxlen <= 16; loads stay inside [.., lim)
__device__ __forceinline__ uint32_t common16(const uint8_t* a, const uint8_t* b, uint32_t maxlen, bool wide_ok)
{
    if (wide_ok) {
        const uint4 x = ld_u128(a), y = ld_u128(b);
        const uint64_t lo = ((uint64_t)(x.y ^ y.y) << 32) | (uint64_t)(x.x ^ y.x);
        const uint64_t hi = ((uint64_t)(x.w ^ y.w) << 32) | (uint64_t)(x.z ^ y.z);
        uint32_t n = lo ? (ctz64(lo) >> 3) : (hi ? 8 + (ctz64(hi) >> 3) : 16);
        return n < maxlen ? n : maxlen;
    }
    uint32_t n = 0;
    while (n < maxlen && a[n] == b[n]) ++n;
    return n;
}

__global__ __launch_bounds__(64)
void lz4_chunks_kernel(const uint8_t* __restrict__ in, uint64_t total, uint32_t chunk,
                       uint8_t* __restrict__ scratch, uint64_t stride, uint32_t* __restrict__ csize)
{
    __shared__ uint32_t table[4096];
    const int lane = threadIdx.x;
    const uint64_t blk = blockIdx.x;
    const uint8_t* __restrict__ src = in + blk * chunk;
    const uint64_t left = total - blk * chunk;
    const uint32_t n = (uint32_t)(left < chunk ? left : chunk);
    uint8_t* __restrict__ dst = scratch + blk * stride;

    {
        uint4* t4 = reinterpret_cast<uint4*>(table);
#pragma unroll
        for (int i = 0; i < 16; ++i) t4[i * 64 + lane] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    const uint32_t olimit = n - 1;       // capacity n-1 (LZ4F_makeBlock); offsets into dst
    uint32_t op = 0, anchor = 0;
    bool failed = false;

    if (n >= LZ4_MINLENGTH) {
        const uint32_t mflimitPlusOne = n - LZ4_MFLIMIT + 1;
        const uint32_t matchlimit = n - LZ4_LASTLITERALS;
        uint32_t P = 1, U = 1;           // first probe of the block: search from ip = 1 (table[hash(0)] = 0 is a no-op)

        for (;;) {
            // ---- positions of the 64 probes of this batch ----
            const uint32_t s_first = (62 + U) >> 6 ? (62 + U) >> 6 : 1;
            const uint32_t ustar = 64 * (s_first + 1) - 62;          // first unified index with step s_first+1
            const uint32_t u = U + lane;
            const uint32_t bump = u > ustar ? u - ustar : 0;          // #earlier lanes already at the larger step
            const uint32_t pos = P + s_first * lane + bump;
            const uint32_t adv = (u >= ustar) ? s_first + 1 : s_first;
            const uint32_t nxt = pos + adv;
            const bool valid = nxt <= mflimitPlusOne;
            const uint64_t vmask = ballot(valid);                     // a prefix of the lanes
            const uint32_t nvalid = (uint32_t)__builtin_popcountll(vmask);

            uint64_t seq = 0;
            uint32_t h = 0, old = 0, fl = 0;
            if (valid) {
                seq = ld_u64(src + pos);
                h = lz4_hash5(seq);
                old = table[h];
                atomicMax(&table[h], 0x80000000u | (uint32_t)(63 - lane));
                fl = table[h];
            }
            const uint32_t seq32 = (uint32_t)seq;
            const bool hazard = valid && ((63u - (fl & 63u)) != (uint32_t)lane);
            const uint64_t hz = ballot(hazard);

            // non-hazard lanes: candidate is the pre-batch table entry
            const bool near = valid && !hazard && (old + LZ4_MAXD >= pos);
            uint32_t m32 = ~seq32;
            if (near) m32 = ld_u32(src + old);
            const uint64_t mm = ballot(near && m32 == seq32);

            uint32_t f = mm ? ctz64(mm) : 64u;
            uint32_t fcand = 0;
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

            // ---- table: restore, then insert the committed prefix (lanes <= f, or every valid lane) ----
            const uint32_t ncommit = (f < 64) ? f + 1 : nvalid;
            if (valid) table[h] = old;
            if ((uint32_t)lane < ncommit) atomicMax(&table[h], pos);

            if (f >= 64) {
                if (nvalid < 64) break;                               // forwardIp > mflimitPlusOne -> last literals
                P = sgpr(lane_read(nxt, 63));
                U += 64;
                continue;
            }

            // ---- a match: ip = pos[f], match = fcand ----
            uint32_t ip = sgpr(lane_read(pos, f));
            uint32_t mt = sgpr(fcand);

            // backward catch-up: while (ip > anchor && match > 0 && ip[-1] == match[-1])
            for (;;) {
                const uint32_t k = lane + 1;
                bool ok = (ip >= anchor + k) && (mt >= k);
                if (ok) ok = src[ip - k] == src[mt - k];
                const uint64_t bad = ~ballot(ok);
                const uint32_t nb = bad ? ctz64(bad) : 64u;
                ip -= nb; mt -= nb;
                if (nb < 64) break;
            }

            // ---- literals ----
            const uint32_t lit = ip - anchor;
            const uint32_t token_pos = op;
            op += 1;
            if (op + lit + (2 + 1 + LZ4_LASTLITERALS) + lit / 255 > olimit) { failed = true; break; }
            if (lit >= 15) {
                const uint32_t rest = lit - 15;
                const uint32_t n255 = rest / 255;
                for (uint32_t i = lane; i < n255; i += 64) dst[op + i] = 255;
                if (lane == 0) dst[op + n255] = (uint8_t)(rest - n255 * 255);
                op += n255 + 1;
            }
            wave_copy(dst + op, src + anchor, lit, lane);
            op += lit;

            // ---- offset + match length ----
            const uint32_t offset = ip - mt;
            if (lane == 0) { dst[op] = (uint8_t)offset; dst[op + 1] = (uint8_t)(offset >> 8); }
            op += 2;

            uint32_t ml = 0;                                          // bytes matched beyond MINMATCH
            {
                const uint32_t q = ip + 4, r = mt + 4;
                for (;;) {
                    // 4 x 16 bytes per lane and round = 4 KiB per round; first round only 1 KiB matters most
                    uint32_t got[4];
                    bool full[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint32_t a = q + ml + (uint32_t)t * 1024u + (uint32_t)lane * 16u;
                        const uint32_t room = a < matchlimit ? matchlimit - a : 0u;
                        const uint32_t maxlen = room < 16u ? room : 16u;
                        got[t] = (maxlen == 0) ? 0u : common16(src + a, src + (a - q + r), maxlen, a + 16u <= n);
                        full[t] = got[t] == 16u;
                    }
                    bool stop = false;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (!stop) {
                            const uint64_t nf = ballot(!full[t]);
                            if (nf) {
                                const uint32_t l = ctz64(nf);
                                ml += l * 16u + lane_read(got[t], l);
                                stop = true;
                            } else {
                                ml += 1024u;
                            }
                        }
                    }
                    if (stop) break;
                }
            }
            ml = sgpr(ml);
            ip += ml + 4;
            if (op + (1 + LZ4_LASTLITERALS) + (ml + 240) / 255 > olimit) { failed = true; break; }
            if (ml >= 15) {
                const uint32_t rest = ml - 15;
                const uint32_t n255 = rest / 255;
                for (uint32_t i = lane; i < n255; i += 64) dst[op + i] = 255;
                if (lane == 0) dst[op + n255] = (uint8_t)(rest - n255 * 255);
                op += n255 + 1;
            }
            if (lane == 0) dst[token_pos] = (uint8_t)(((lit < 15 ? lit : 15u) << 4) | (ml < 15 ? ml : 15u));
            anchor = ip;
            if (ip >= mflimitPlusOne) break;

            // LZ4_putPosition(ip - 2), then the unified batch starts with the "test next position" probe at ip
            {
                const uint32_t h2 = lz4_hash5(ld_u64(src + ip - 2));
                if (lane == 0) table[h2] = ip - 2;
            }
            P = ip;
            U = 0;
        }
    }

    if (!failed) {
        const uint32_t lastRun = n - anchor;
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
            wave_copy(dst + op, src + anchor, lastRun, lane);
            op += lastRun;
        }
    }
    if (lane == 0) csize[blk] = failed ? 0u : op;
}

// ------------------------------------------------------------------------------------------------
// frame layout: exclusive scan of frame sizes, then scatter  [7 B header][u32 size][data][u32 0]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void lz4_frame_scan_kernel(const uint32_t* __restrict__ csize, uint64_t nchunks, uint64_t total, uint32_t chunk,
                           uint64_t* __restrict__ frame_off /* nchunks + 1 */)
{
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t base = 0; base < nchunks; base += 1024) {
        const uint64_t k = base + tid;
        uint64_t sz = 0;
        if (k < nchunks) {
            const uint64_t left = total - k * chunk;
            const uint64_t nk = left < chunk ? left : chunk;
            const uint32_t c = csize[k];
            sz = 7 + 4 + (c ? c : nk) + 4;
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
        if (tid == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0) frame_off[nchunks] = carry_s;
}

// one workgroup per (chunk, slice): copies its slice of the frame body, slice 0 also writes header/trailer
constexpr uint32_t GATHER_SLICE = 32768;

__global__ __launch_bounds__(256)
void lz4_frame_gather_kernel(const uint8_t* __restrict__ in, uint64_t total, uint32_t chunk,
                             const uint8_t* __restrict__ scratch, uint64_t stride,
                             const uint32_t* __restrict__ csize, const uint64_t* __restrict__ frame_off,
                             uint8_t* __restrict__ out, uint32_t bd_byte, uint32_t hc_byte, uint32_t slices_per_chunk)
{
    const uint64_t k = blockIdx.x / slices_per_chunk;
    const uint32_t slice = blockIdx.x % slices_per_chunk;
    const uint64_t left = total - k * chunk;
    const uint32_t nk = (uint32_t)(left < chunk ? left : chunk);
    const uint32_t c = csize[k];
    const uint32_t body = c ? c : nk;
    const uint8_t* __restrict__ s = c ? scratch + k * stride : in + k * chunk;
    uint8_t* __restrict__ d = out + frame_off[k];
    const int tid = threadIdx.x;

    if (slice == 0 && tid < 15) {
        const uint32_t field = c ? c : (nk | 0x80000000u);
        uint8_t v = 0;
        uint32_t o = tid;
        switch (tid) {
            case 0: v = 0x04; break; case 1: v = 0x22; break; case 2: v = 0x4D; break; case 3: v = 0x18; break;
            case 4: v = 0x40; break; case 5: v = (uint8_t)bd_byte; break; case 6: v = (uint8_t)hc_byte; break;
            case 7: v = (uint8_t)field; break; case 8: v = (uint8_t)(field >> 8); break;
            case 9: v = (uint8_t)(field >> 16); break; case 10: v = (uint8_t)(field >> 24); break;
            default: v = 0; o = 11 + body + (tid - 11); break;       // end mark
        }
        d[o] = v;
    }
    const uint32_t begin = slice * GATHER_SLICE;
    if (begin >= body) return;
    const uint32_t end = (begin + GATHER_SLICE < body) ? begin + GATHER_SLICE : body;
    uint8_t* __restrict__ dd = d + 11 + begin;
    const uint8_t* __restrict__ ss = s + begin;
    uint32_t len = end - begin;
    // head to 16-byte alignment of the destination
    const uint32_t head0 = (uint32_t)((16 - (reinterpret_cast<uintptr_t>(dd) & 15)) & 15);
    const uint32_t head = head0 < len ? head0 : len;
    if ((uint32_t)tid < head) dd[tid] = ss[tid];
    dd += head; ss += head; len -= head;
    const uint32_t nvec = len >> 4;
    for (uint32_t i = tid; i < nvec; i += 256) {
        *reinterpret_cast<uint4*>(dd + (size_t)i * 16) = ld_u128(ss + (size_t)i * 16);
    }
    const uint32_t done = nvec << 4;
    if ((uint32_t)tid < len - done) dd[done + tid] = ss[done + tid];
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

hipError_t launch_bitswap1_u16(const uint16_t* in, uint16_t* out, uint64_t len, hipStream_t stream)
{
    if (len == 0) return hipSuccess;
    const uint64_t seg_words = len / 16;
    uint64_t n_tiles = 0;
    const bool aligned = (seg_words % 8 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (aligned) n_tiles = (seg_words * 16) / BSW_TILE_VOX;
    if (n_tiles) {
        const uint64_t want = (n_tiles + BSW_WAVES - 1) / BSW_WAVES;
        const uint64_t cap = (uint64_t)num_cus() * 2 * 4;   // 2 workgroups resident per CU (LDS), a few rounds each
        const unsigned grid = (unsigned)(want < cap ? want : cap);
        hipLaunchKernelGGL(bitswap1_u16_tiles, dim3(grid), dim3(64 * BSW_WAVES), BSW_WAVES * BSW_LDS_PER_WAVE, stream,
                           in, out, n_tiles, seg_words);
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

hipError_t launch_bitswap1_u8(const uint8_t* in, uint8_t* out, uint64_t len, hipStream_t stream)
{
    if (len == 0) return hipSuccess;
    const uint64_t seg = len / 8;
    uint64_t blocks = (seg + 255) / 256;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(bitswap1_u8_generic, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, len, seg);
    return hipGetLastError();
}

hipError_t launch_diff3x3x1(const void* in, void* out, uint64_t Z, uint64_t Y, uint64_t X, int elem_size, hipStream_t stream)
{
    // geometry of the reference's halo (neighborhood_utils.hpp:160-240), see the kernel header comment
    const uint64_t length = Z * Y * X;
    if (length == 0) return hipSuccess;
    const uint64_t zlim = X < Z ? X : Z;                         // z in [1, min(X, Z))
    const uint64_t noff = (zlim >= 1 ? (zlim - 1) : 0) * (Y >= 2 ? (Y - 2) : 0);
    const int single = (noff == 1);
    const uint64_t hx = single ? 0 : (Z >= 2 ? Z - 2 : 0);
    uint64_t blocks = (length + 255) / 256;
    const uint64_t cap = (uint64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (elem_size == 2)
        hipLaunchKernelGGL((diff3x3x1_kernel<uint16_t, int16_t>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           (const uint16_t*)in, (uint16_t*)out, Z, Y, X, hx, zlim, single);
    else
        hipLaunchKernelGGL((diff3x3x1_kernel<uint8_t, int8_t>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           (const uint8_t*)in, (uint8_t*)out, Z, Y, X, hx, zlim, single);
    return hipGetLastError();
}

hipError_t launch_lz4_chunks(const uint8_t* in, uint64_t total, uint32_t chunk, uint8_t* scratch, uint64_t stride,
                             uint32_t* csize, uint64_t nchunks, hipStream_t stream)
{
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(lz4_chunks_kernel, dim3((unsigned)nchunks), dim3(64), 0, stream, in, total, chunk, scratch, stride, csize);
    return hipGetLastError();
}

hipError_t launch_lz4_frame_scan(const uint32_t* csize, uint64_t nchunks, uint64_t total, uint32_t chunk,
                                 uint64_t* frame_off, hipStream_t stream)
{
    hipLaunchKernelGGL(lz4_frame_scan_kernel, dim3(1), dim3(1024), 0, stream, csize, nchunks, total, chunk, frame_off);
    return hipGetLastError();
}

hipError_t launch_lz4_frame_gather(const uint8_t* in, uint64_t total, uint32_t chunk, const uint8_t* scratch, uint64_t stride,
                                   const uint32_t* csize, const uint64_t* frame_off, uint8_t* out, uint32_t bd_byte,
                                   uint32_t hc_byte, uint64_t nchunks, hipStream_t stream)
{
    if (nchunks == 0) return hipSuccess;
    const uint32_t slices = (chunk + GATHER_SLICE - 1) / GATHER_SLICE;
    hipLaunchKernelGGL(lz4_frame_gather_kernel, dim3((unsigned)(nchunks * slices)), dim3(256), 0, stream, in, total, chunk,
                       scratch, stride, csize, frame_off, out, bd_byte, hc_byte, slices);
    return hipGetLastError();
}

} // namespace sqy
