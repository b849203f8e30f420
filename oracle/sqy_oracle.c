/*
 * sqy_oracle.c -- CPU restatement of the sqeazy hot path.  TEST INFRASTRUCTURE ONLY (see
 * sqy_oracle.h): nothing under sqeazy_amd/ may include, link or call this file.
 *
 * Compile WITHOUT -ffast-math (the reference's release flags use it, which makes its float
 * paths compiler dependent; the declared parity target is IEEE evaluation in statement order).
 *
 * Paths cited are relative to /root/reference/src/cpp/src.
 */
#include "sqy_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* bitswap1                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* encoders/bitplane_reorder_scalar.hpp:27-74 (scalar_bitplane_reorder_encode<1>) with the tail
 * rule of encoders/bitswap_scheme_impl.hpp:97-103: W = bits per element, L = len - len % W,
 * S = L / W.  Bit b of element i lands in word (W-1-b)*S + i/W at bit W-1-(i%W).
 * Elements [L,len) are copied unchanged. */
void sqo_bitswap1_encode_u16(const uint16_t* in, uint16_t* out, size_t len)
{
    const unsigned W = 16;
    const size_t L = len - (len % W);
    const size_t S = L / W;
    for (size_t i = L; i < len; ++i) out[i] = in[i];
    memset(out, 0, L * sizeof(uint16_t));
    for (size_t i = 0; i < L; ++i) {
        const uint16_t v = in[i];
        const unsigned obit = (W - 1) - (unsigned)(i % W);
        for (unsigned b = 0; b < W; ++b) {
            const size_t oidx = (size_t)(W - 1 - b) * S + i / W;
            out[oidx] = (uint16_t)(out[oidx] | (((v >> b) & 1u) << obit));
        }
    }
}

void sqo_bitswap1_encode_u8(const uint8_t* in, uint8_t* out, size_t len)
{
    const unsigned W = 8;
    const size_t L = len - (len % W);
    const size_t S = L / W;
    for (size_t i = L; i < len; ++i) out[i] = in[i];
    memset(out, 0, L);
    for (size_t i = 0; i < L; ++i) {
        const uint8_t v = in[i];
        const unsigned obit = (W - 1) - (unsigned)(i % W);
        for (unsigned b = 0; b < W; ++b) {
            const size_t oidx = (size_t)(W - 1 - b) * S + i / W;
            out[oidx] = (uint8_t)(out[oidx] | (((v >> b) & 1u) << obit));
        }
    }
}

/* encoders/sse_utils.hpp:1365-1433 (simd_segment_broadcast) + :1150-1217
 * (simd_collect_single_bitplane_impl): one full pass over the input per bit-plane, planes
 * distributed over <= 16 threads, 8 elements gathered per step through a sign-bit mask
 * (the reference shifts the wanted bit to the msb and uses movemask; the scalar loop below
 * collects the same 8 bits).  Output identical to sqo_bitswap1_encode_u16. */
void sqo_bitswap1_encode_u16_planes(const uint16_t* in, uint16_t* out, size_t len, int nthreads)
{
    const unsigned W = 16;
    const size_t L = len - (len % W);
    const size_t S = L / W;
    for (size_t i = L; i < len; ++i) out[i] = in[i];
    if (nthreads > (int)W) nthreads = (int)W;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int seg = 0; seg < (int)W; ++seg) {
        const unsigned b = W - 1 - (unsigned)seg; /* plane 0 of the output is the msb */
        uint16_t* dst = out + (size_t)seg * S;
        for (size_t w = 0; w < S; ++w) {
            const uint16_t* p = in + w * W;
            unsigned acc = 0;
            for (unsigned j = 0; j < W; ++j) acc = (acc << 1) | ((p[j] >> b) & 1u);
            dst[w] = (uint16_t)acc;
        }
    }
}

/* encoders/bitplane_reorder_scalar.hpp:81-116 + bitswap_scheme_impl.hpp:147-165 */
void sqo_bitswap1_decode_u16(const uint16_t* in, uint16_t* out, size_t len)
{
    const unsigned W = 16;
    const size_t L = len - (len % W);
    const size_t S = L / W;
    for (size_t i = L; i < len; ++i) out[i] = in[i];
    for (size_t i = 0; i < L; ++i) {
        unsigned v = 0;
        const unsigned ibit = (W - 1) - (unsigned)(i % W);
        for (unsigned b = 0; b < W; ++b) {
            const size_t iidx = (size_t)(W - 1 - b) * S + i / W;
            v |= ((in[iidx] >> ibit) & 1u) << b;
        }
        out[i] = (uint16_t)v;
    }
}

void sqo_bitswap1_decode_u8(const uint8_t* in, uint8_t* out, size_t len)
{
    const unsigned W = 8;
    const size_t L = len - (len % W);
    const size_t S = L / W;
    for (size_t i = L; i < len; ++i) out[i] = in[i];
    for (size_t i = 0; i < L; ++i) {
        unsigned v = 0;
        const unsigned ibit = (W - 1) - (unsigned)(i % W);
        for (unsigned b = 0; b < W; ++b) {
            const size_t iidx = (size_t)(W - 1 - b) * S + i / W;
            v |= ((in[iidx] >> ibit) & 1u) << b;
        }
        out[i] = (uint8_t)v;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* diff3x3x1                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* neighborhood_utils.hpp:160-240 for last_plane_neighborhood<3> (offsets begin {-1,-1,-1},
 * end {2,2,0} on axes {x,y,z}).  halo::world is indexed world[2]=w(x), world[1]=h(y),
 * world[0]=d(z), but non_halo_end(dim) indexes world[dim] with dim 0 meaning "x" -- so the x
 * extent comes out of the DEPTH and the z extent out of the WIDTH (reference quirk, kept):
 *   non_halo_begin(*) = 1
 *   non_halo_end(0) = world[0]-1 = Z-1      -> halo_size_x = Z-2       (diff_scheme_impl.hpp:97)
 *   non_halo_end(1) = world[1]-1 = Y-1
 *   non_halo_end(2) = world[2]   = X        (z loop runs to X, clipped by offset < length)
 * compute_offsets_in_x: num_offsets_required = (X-1)*(Y-2); when that is <= 1 there is a single
 * offset (=1) and halo_size_x = length - 1 (diff_scheme_impl.hpp:98-101). */
size_t sqo_diff3x3x1_offsets(const size_t shape[3], size_t* out, size_t cap, size_t* halo_size_x)
{
    const size_t Z = shape[0], Y = shape[1], X = shape[2];
    const size_t length = Z * Y * X;
    /* compute_offsets_in_x: int non_halo_length; if(non_halo_length != (int)world.at(i)) num *= ... */
    long num = 1;
    if (((int)X - 1) != (int)X) num *= (long)((int)X - 1);          /* i = 2 */
    if (((int)Y - 2) != (int)Y) num *= (long)((int)Y - 2);          /* i = 1 */
    if (halo_size_x) *halo_size_x = 0;
    if (num <= 1) {
        /* the reference takes its single-offset branch (offset 1, halo_size_x = length-1) and
         * reads in front of the buffer: undefined there, refused here. */
        return (size_t)-1;
    }
    size_t n = 0, first = 0;
    for (size_t z = 1; z < X; ++z) {
        for (size_t y = 1; y + 1 < Y; ++y) {
            const size_t off = z * Y * X + y * X + 1;
            if (off < length) {
                if (n == 0) first = off;
                if (out && n < cap) out[n] = off;
                ++n;
            }
        }
    }
    /* diff_scheme_impl.hpp:97-101; with Z == 1 there are no offsets and the value is unused */
    if (halo_size_x) *halo_size_x = (n == 1) ? (length - first) : (Z >= 2 ? Z - 2 : 0);
    return n;
}

/* diff_scheme_utils.hpp:70-99 (naive_sum): the 9 neighbours in plane z-1 are addressed through
 * the flat index (coordinates are re-derived from it, so idx-frame+dy*X+dx for 16-bit data where
 * coord_t = short holds every coordinate); the sum is accumulated in the PIXEL type (wraps). */
#define SQO_DIFF_BODY(T, ST, SUMT)                                                                       \
    const size_t Z = shape[0], Y = shape[1], X = shape[2];                                               \
    const size_t length = Z * Y * X;                                                                     \
    const size_t frame = X * Y;                                                                          \
    memcpy(out, in, length * sizeof(T));                                                                 \
    size_t hx = 0;                                                                                       \
    size_t noff = sqo_diff3x3x1_offsets(shape, NULL, 0, &hx);                                            \
    if (noff == (size_t)-1) return 1;                                                                    \
    if (noff == 0) return 0;                                                                             \
    size_t* offs = (size_t*)malloc((noff ? noff : 1) * sizeof(size_t));                                  \
    if (!offs) return 1;                                                                                 \
    sqo_diff3x3x1_offsets(shape, offs, noff, &hx);                                                       \
    _Pragma("omp parallel for schedule(static) if(SQO_DIFF_PARALLEL)")                                   \
    for (long o = 0; o < (long)noff; ++o) {                                                              \
        for (size_t k = 0; k < hx; ++k) {                                                                \
            const size_t idx = offs[o] + k;                                                              \
            T sum = 0;                                                                                   \
            for (long dy = -1; dy <= 1; ++dy)                                                            \
                for (long dx = -1; dx <= 1; ++dx)                                                        \
                    sum = (T)(sum + src[(size_t)((long)idx - (long)frame + dy * (long)X + dx)]);         \
            const SUMT local_sum = (SUMT)sum;                                                            \
            BODY_STORE                                                                                   \
        }                                                                                                \
    }                                                                                                    \
    free(offs);                                                                                          \
    return 0;

int sqo_diff3x3x1_encode_u16(const uint16_t* in, uint16_t* out, const size_t shape[3])
{
#define SQO_DIFF_PARALLEL 1   /* encode reads only the raw input: rows are independent */
    const uint16_t* src = in;
#define BODY_STORE out[idx] = (uint16_t)(int16_t)(in[idx] - local_sum / 9u);
    SQO_DIFF_BODY(uint16_t, int16_t, unsigned int)
#undef BODY_STORE
}

/* 8-bit: coord_t = char (traits.hpp:10-11), so z/y/x derived from the flat index are truncated to
 * 8 bits; with every extent <= 127 they are exact and the flat-index form above holds.  Larger
 * 8-bit volumes index out of bounds in the reference (undefined); refused here. */
int sqo_diff3x3x1_encode_u8(const uint8_t* in, uint8_t* out, const size_t shape[3])
{
    if (shape[0] > 127 || shape[1] > 127 || shape[2] > 127) return 1;
    const uint8_t* src = in;
#define BODY_STORE out[idx] = (uint8_t)(int8_t)(in[idx] - local_sum / 9u);
    SQO_DIFF_BODY(uint8_t, int8_t, unsigned short)
#undef BODY_STORE
}

/* diff_scheme_impl.hpp:143-194: decode walks the same offsets in order and reads the ALREADY
 * DECODED output (plane z-1 is complete before plane z only in serial order; restated serially). */
#undef SQO_DIFF_PARALLEL
#define SQO_DIFF_PARALLEL 0   /* decode reads its own output of plane z-1: serial, as the restated reference order */
int sqo_diff3x3x1_decode_u16(const uint16_t* in, uint16_t* out, const size_t shape[3])
{
    const uint16_t* src = out;
#define BODY_STORE out[idx] = (uint16_t)((int16_t)in[idx] + local_sum / 9u);
    SQO_DIFF_BODY(uint16_t, int16_t, unsigned int)
#undef BODY_STORE
}

int sqo_diff3x3x1_decode_u8(const uint8_t* in, uint8_t* out, const size_t shape[3])
{
    if (shape[0] > 127 || shape[1] > 127 || shape[2] > 127) return 1;
    const uint8_t* src = out;
#define BODY_STORE out[idx] = (uint8_t)((int8_t)in[idx] + local_sum / 9u);
    SQO_DIFF_BODY(uint8_t, int8_t, unsigned short)
#undef BODY_STORE
}

/* diff3x3x1 as a TAIL filter: the sink's output type is `char` (sqeazy_pipelines.hpp:64-77), i.e. T = char, signed on x86.
 * naive_sum accumulates in `char` (wraps), and sum_type = add_unsigned<twice_as_wide<char>>::type = unsigned short
 * (diff_scheme_impl.hpp:24, traits.hpp:29): the char sum is SIGN-EXTENDED into the unsigned short before the division by 9
 * (a sum of -5 divides as 65531).  Bytes in, bytes out here; everything else as the 8-bit head filter above. */
#undef SQO_DIFF_PARALLEL
#define SQO_DIFF_PARALLEL 1
int sqo_diff3x3x1_encode_i8(const uint8_t* in, uint8_t* out, const size_t shape[3])
{
    if (shape[0] > 127 || shape[1] > 127 || shape[2] > 127) return 1;
    const uint8_t* src = in;
#define BODY_STORE out[idx] = (uint8_t)((unsigned)(int)(int8_t)in[idx] - (unsigned)(unsigned short)(short)(int8_t)local_sum / 9u);
    SQO_DIFF_BODY(uint8_t, int8_t, unsigned short)
#undef BODY_STORE
}
#undef SQO_DIFF_PARALLEL
#define SQO_DIFF_PARALLEL 0
int sqo_diff3x3x1_decode_i8(const uint8_t* in, uint8_t* out, const size_t shape[3])
{
    if (shape[0] > 127 || shape[1] > 127 || shape[2] > 127) return 1;
    const uint8_t* src = out;
#define BODY_STORE out[idx] = (uint8_t)((unsigned)(int)(int8_t)in[idx] + (unsigned)(unsigned short)(short)(int8_t)local_sum / 9u);
    SQO_DIFF_BODY(uint8_t, int8_t, unsigned short)
#undef BODY_STORE
}

/* ------------------------------------------------------------------------------------------ */
/* LZ4 block compressor -- liblz4 1.9.3, LZ4_compress_generic(byU32, limitedOutput, accel 1)   */
/* as reached from LZ4F_compressUpdate -> LZ4F_makeBlock -> LZ4_compress_fast_continue on a    */
/* fresh stream (sqeazy call sites: encoders/lz4_utils.hpp:118-170).  Third-party dependency,  */
/* not in the reference tree; restated from the block format + upstream behaviour and pinned   */
/* against liblz4.so.1.9.3 by oracle/gen_golden.py.                                            */
/* ------------------------------------------------------------------------------------------ */
#define LZ4_MINMATCH 4
#define LZ4_MFLIMIT 12
#define LZ4_LASTLITERALS 5
#define LZ4_MINLENGTH (LZ4_MFLIMIT + 1)
#define LZ4_MAXD 65535
#define LZ4_MLBITS 4
#define LZ4_MLMASK 15u
#define LZ4_RUNMASK 15u
#define LZ4_SKIPTRIGGER 6
#define LZ4_HASHLOG 12

/* liblz4's `acceleration` (LZ4_compress_fast_continue's last argument): 1 for sqeazy's accel = 0..2; LZ4F turns a negative
 * compression level -k (sqeazy: lz4(accel=-k), encoders/lz4.hpp:103-113) into acceleration k + 1 (lz4frame.c 1.9.3,
 * LZ4F_compressBlock / LZ4F_compressBlock_continue), lz4.c caps it at LZ4_ACCELERATION_MAX = 65537.  The search then starts with
 * searchMatchNb = acceleration << 6 instead of 1 << 6.  Set by the test that drives the oracle (not thread safe: test infrastructure). */
static int sqo_lz4_acceleration = 1;
void sqo_lz4_set_acceleration(int a) { sqo_lz4_acceleration = a < 1 ? 1 : (a > 65537 ? 65537 : a); }

static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

static inline uint32_t lz4_hash5(const uint8_t* p)
{
    return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - LZ4_HASHLOG));
}

int sqo_lz4_block_compress(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    uint32_t table[1 << LZ4_HASHLOG];
    memset(table, 0, sizeof(table));
    if (n <= 0) return 0;

    const uint8_t* const base = src;
    const uint8_t* ip = src;
    const uint8_t* anchor = src;
    const uint8_t* const iend = src + n;
    const uint8_t* const mflimitPlusOne = iend - LZ4_MFLIMIT + 1;
    const uint8_t* const matchlimit = iend - LZ4_LASTLITERALS;
    uint8_t* op = dst;
    uint8_t* const olimit = dst + cap;
    uint32_t forwardH;

    if (n < LZ4_MINLENGTH) goto last_literals;

    table[lz4_hash5(ip)] = 0;
    ip++;
    forwardH = lz4_hash5(ip);

    for (;;) {
        const uint8_t* match;
        uint8_t* token;
        {
            const uint8_t* forwardIp = ip;
            int step = 1;
            int searchMatchNb = sqo_lz4_acceleration << LZ4_SKIPTRIGGER;
            do {
                const uint32_t h = forwardH;
                const uint32_t current = (uint32_t)(forwardIp - base);
                uint32_t matchIndex = table[h];
                ip = forwardIp;
                forwardIp += step;
                step = (searchMatchNb++ >> LZ4_SKIPTRIGGER);
                if (forwardIp > mflimitPlusOne) goto last_literals;
                match = base + matchIndex;
                forwardH = lz4_hash5(forwardIp);
                table[h] = current;
                if (matchIndex + LZ4_MAXD < current) continue;
                if (rd32(match) == rd32(ip)) break;
            } while (1);
        }
        while ((ip > anchor) && (match > src) && (ip[-1] == match[-1])) { ip--; match--; }

        {
            const unsigned litLength = (unsigned)(ip - anchor);
            token = op++;
            if (op + litLength + (2 + 1 + LZ4_LASTLITERALS) + (litLength / 255) > olimit) return 0;
            if (litLength >= LZ4_RUNMASK) {
                int len = (int)(litLength - LZ4_RUNMASK);
                *token = (uint8_t)(LZ4_RUNMASK << LZ4_MLBITS);
                for (; len >= 255; len -= 255) *op++ = 255;
                *op++ = (uint8_t)len;
            } else {
                *token = (uint8_t)(litLength << LZ4_MLBITS);
            }
            memcpy(op, anchor, litLength);
            op += litLength;
        }
    next_match:
        {
            const unsigned offset = (unsigned)(ip - match);
            *op++ = (uint8_t)offset;
            *op++ = (uint8_t)(offset >> 8);
        }
        {
            unsigned matchCode;
            {
                const uint8_t* pi = ip + LZ4_MINMATCH;
                const uint8_t* pm = match + LZ4_MINMATCH;
                while (pi < matchlimit && *pi == *pm) { pi++; pm++; }
                matchCode = (unsigned)(pi - (ip + LZ4_MINMATCH));
            }
            ip += (size_t)matchCode + LZ4_MINMATCH;
            if (op + (1 + LZ4_LASTLITERALS) + (matchCode + 240) / 255 > olimit) return 0;
            if (matchCode >= LZ4_MLMASK) {
                *token = (uint8_t)(*token + LZ4_MLMASK);
                matchCode -= LZ4_MLMASK;
                while (matchCode >= 255) { *op++ = 255; matchCode -= 255; }
                *op++ = (uint8_t)matchCode;
            } else {
                *token = (uint8_t)(*token + matchCode);
            }
        }
        anchor = ip;
        if (ip >= mflimitPlusOne) break;

        table[lz4_hash5(ip - 2)] = (uint32_t)(ip - 2 - base);
        {
            const uint32_t h = lz4_hash5(ip);
            const uint32_t current = (uint32_t)(ip - base);
            const uint32_t matchIndex = table[h];
            match = base + matchIndex;
            table[h] = current;
            if ((matchIndex + LZ4_MAXD >= current) && (rd32(match) == rd32(ip))) {
                token = op++;
                *token = 0;
                goto next_match;
            }
        }
        forwardH = lz4_hash5(++ip);
    }

last_literals:
    {
        const size_t lastRun = (size_t)(iend - anchor);
        if (op + lastRun + 1 + ((lastRun + 255 - LZ4_RUNMASK) / 255) > olimit) return 0;
        if (lastRun >= LZ4_RUNMASK) {
            size_t acc = lastRun - LZ4_RUNMASK;
            *op++ = (uint8_t)(LZ4_RUNMASK << LZ4_MLBITS);
            for (; acc >= 255; acc -= 255) *op++ = 255;
            *op++ = (uint8_t)acc;
        } else {
            *op++ = (uint8_t)(lastRun << LZ4_MLBITS);
        }
        memcpy(op, anchor, lastRun);
        op += lastRun;
    }
    return (int)(op - dst);
}

int sqo_lz4_block_decompress(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    const uint8_t* ip = src;
    const uint8_t* const iend = src + n;
    uint8_t* op = dst;
    uint8_t* const oend = dst + cap;
    while (ip < iend) {
        const unsigned token = *ip++;
        size_t lit = token >> 4;
        if (lit == 15) {
            unsigned s;
            do { if (ip >= iend) return -1; s = *ip++; lit += s; } while (s == 255);
        }
        if ((size_t)(iend - ip) < lit || (size_t)(oend - op) < lit) return -1;
        memcpy(op, ip, lit);
        op += lit;
        ip += lit;
        if (ip >= iend) break;
        if (iend - ip < 2) return -1;
        const size_t offset = (size_t)ip[0] | ((size_t)ip[1] << 8);
        ip += 2;
        if (offset == 0 || offset > (size_t)(op - dst)) return -1;
        size_t ml = token & 15u;
        if (ml == 15) {
            unsigned s;
            do { if (ip >= iend) return -1; s = *ip++; ml += s; } while (s == 255);
        }
        ml += LZ4_MINMATCH;
        if ((size_t)(oend - op) < ml) return -1;
        const uint8_t* m = op - offset;
        for (size_t i = 0; i < ml; ++i) op[i] = m[i];
        op += ml;
    }
    return (int)(op - dst);
}

/* ------------------------------------------------------------------------------------------ */
/* xxh32 (LZ4 frame header checksum) -- public XXH32 definition                               */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

uint32_t sqo_xxh32(const uint8_t* p, size_t len, uint32_t seed)
{
    const uint32_t P1 = 2654435761U, P2 = 2246822519U, P3 = 3266489917U, P4 = 668265263U, P5 = 374761393U;
    const uint8_t* const end = p + len;
    uint32_t h;
    if (len >= 16) {
        const uint8_t* const limit = end - 16;
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 = rotl32(v1 + rd32(p) * P2, 13) * P1; p += 4;
            v2 = rotl32(v2 + rd32(p) * P2, 13) * P1; p += 4;
            v3 = rotl32(v3 + rd32(p) * P2, 13) * P1; p += 4;
            v4 = rotl32(v4 + rd32(p) * P2, 13) * P1; p += 4;
        } while (p <= limit);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl32(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl32(h + (*p) * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

/* ------------------------------------------------------------------------------------------ */
/* LZ4 framing as sqeazy drives it                                                             */
/* ------------------------------------------------------------------------------------------ */
static const size_t lz4f_block_bytes[8] = {0, 0, 0, 0, 64u << 10, 256u << 10, 1u << 20, 4u << 20};

/* LZ4F_compressBound(chunk, prefs) + LZ4F_HEADER_SIZE_MAX for autoFlush=0, no checksums
 * (values pinned by tests/test_lz4_sandbox.cpp:387-430: 256 KiB -> 262152; header max 19). */
static size_t lz4f_compress_bound(size_t srcSize, int blocksize_id)
{
    const size_t blockSize = lz4f_block_bytes[blocksize_id];
    const size_t bufferedSize = blockSize - 1; /* autoFlush == 0: assume a full tmp buffer */
    const size_t maxSrcSize = srcSize + bufferedSize;
    const unsigned nbFullBlocks = (unsigned)(maxSrcSize / blockSize);
    const size_t partialBlockSize = maxSrcSize & (blockSize - 1);
    const int flush = (srcSize == 0); /* autoFlush | (srcSize == 0) */
    const size_t lastBlockSize = flush ? partialBlockSize : 0;
    const unsigned nbBlocks = nbFullBlocks + (lastBlockSize > 0);
    const size_t blockCRCSize = 0, frameEnd = 4; /* no block / content checksum */
    return (4 + blockCRCSize) * nbBlocks + (blockSize * nbFullBlocks) + lastBlockSize + frameEnd;
}

size_t sqo_lz4_max_encoded_size(size_t n, size_t chunk, int blocksize_id, int nthreads)
{
    /* encoders/lz4.hpp:166-188 */
    if (chunk >= n) return 19 + lz4f_compress_bound(chunk, blocksize_id);
    const size_t nchunks = (n + chunk - 1) / chunk;
    const size_t per_thread = (nchunks + (size_t)nthreads - 1) / (size_t)nthreads;
    return per_thread * (lz4f_compress_bound(chunk, blocksize_id) + 19) * (size_t)nthreads;
}

static size_t lz4f_write_single_block_frame(const uint8_t* src, size_t n, uint8_t* dst, int blocksize_id)
{
    /* LZ4F_compressBegin: magic, FLG (version 01, block-linked => B.Indep = 0, no checksums, no
     * content size, no dictID) = 0x40, BD = blocksize_id << 4, HC = (xxh32(FLG,BD) >> 8) & 0xff */
    uint8_t* op = dst;
    op[0] = 0x04; op[1] = 0x22; op[2] = 0x4D; op[3] = 0x18;
    op[4] = 0x40;
    op[5] = (uint8_t)(blocksize_id << 4);
    op[6] = (uint8_t)((sqo_xxh32(op + 4, 2, 0) >> 8) & 0xff);
    op += 7;
    if (n > 0) {
        /* LZ4F_makeBlock: compress with capacity n-1, store raw when that fails */
        int c = sqo_lz4_block_compress(src, (int)n, op + 4, (int)n - 1);
        uint32_t field;
        if (c == 0) {
            field = (uint32_t)n | 0x80000000u;
            memcpy(op + 4, src, n);
            c = (int)n;
        } else {
            field = (uint32_t)c;
        }
        op[0] = (uint8_t)field; op[1] = (uint8_t)(field >> 8); op[2] = (uint8_t)(field >> 16); op[3] = (uint8_t)(field >> 24);
        op += 4 + (size_t)c;
    }
    op[0] = op[1] = op[2] = op[3] = 0; /* LZ4F_compressEnd: end mark, no content checksum */
    op += 4;
    return (size_t)(op - dst);
}

/* ------------------------------------------------------------------------------------------ */
/* Block-linked frames: lz4::encode_serial (encoders/lz4_utils.hpp:99-173) = LZ4F_compressBegin,  */
/* one LZ4F_compressUpdate per `framestep` bytes (options NULL => stableSrc 0), LZ4F_compressEnd, */
/* with prefs {blockLinked, autoFlush 0} (encoders/lz4.hpp:103-113).  What liblz4 1.9.3 does then: */
/*   - one LZ4_stream_t for the frame: the byU32 table and currentOffset carry across blocks;      */
/*   - full blocks are compressed straight from the caller's buffer, a remainder < blockSize is    */
/*     parked in tmpBuff and compressed from there by the next update / by compressEnd;            */
/*   - after an update that compressed from the caller's buffer the last 64 KiB are copied into    */
/*     tmpBuff (LZ4F_localSaveDict), so the first block of the next update runs in liblz4's        */
/*     external-dictionary mode, blocks that follow their dictionary in memory in prefix mode.     */
/* Both modes see the same logically contiguous history (at most 65535 bytes back); they differ in */
/* how far the backward catch-up may move the match (`lowLimit`), which is modelled per block.     */
/* Restated from upstream behaviour; pinned against liblz4.so.1.9.3 (oracle/gen_golden.py,         */
/* tests/test_oracle_golden.py::test_serial_*).                                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t table[1 << LZ4_HASHLOG];
    uint32_t currentOffset;
    int dict_space;          /* 0 = none (NULL), 1 = caller's buffer, 2 = tmpBuff */
    uint64_t dict_addr;      /* address of the dictionary inside its space */
    uint32_t dictSize;
} lz4_stream_model;

/* LZ4_compress_fast_continue(stream, block, dst, n, cap, 1) on the logical stream `s0` (position 0 = first byte of the
 * frame's input); the block is s0[start, start+n) and sits at (space, addr) in memory. */
static int lz4_block_continue(lz4_stream_model* st, const uint8_t* s0, uint64_t start, int n, int space, uint64_t addr,
                              uint8_t* dst, int cap)
{
    /* LZ4_renormDictT */
    if ((uint64_t)st->currentOffset + (uint32_t)n > 0x80000000u) {
        const uint32_t delta = st->currentOffset - (64u << 10);
        for (int i = 0; i < (1 << LZ4_HASHLOG); ++i) st->table[i] = st->table[i] < delta ? 0 : st->table[i] - delta;
        st->currentOffset = 64u << 10;
        if (st->dictSize > (64u << 10)) { st->dict_addr += st->dictSize - (64u << 10); st->dictSize = 64u << 10; }
    }
    int dict_follows = st->dict_space == space && st->dict_addr + st->dictSize == addr;
    if ((st->dictSize - 1u < 4u - 1u) && !dict_follows) {      /* invalidate tiny dictionaries */
        st->dictSize = 0; st->dict_space = space; st->dict_addr = addr; dict_follows = 1;
    }
    const int prefix = dict_follows;
    const int dict_small = (st->dictSize < (64u << 10)) && (st->dictSize < st->currentOffset);
    const uint32_t dictSize = st->dictSize;
    const uint32_t startIndex = st->currentOffset;
    const uint32_t prefixIdxLimit = startIndex - dictSize;
    /* logical position of table index i: start + (i - startIndex) */
    const uint64_t low_dict = start - dictSize;                /* dictionary start (prefix mode: lowLimit) */
    const uint64_t low_in = prefix ? low_dict : start;

    /* context update (top of LZ4_compress_generic), then what LZ4_compress_fast_continue does after an extDict block */
    st->currentOffset += (uint32_t)n;
    if (prefix) st->dictSize += (uint32_t)n;
    else { st->dict_space = space; st->dict_addr = addr; st->dictSize = (uint32_t)n; }

    const uint8_t* const src = s0 + start;
    const uint8_t* ip = src;
    const uint8_t* anchor = src;
    const uint8_t* const iend = src + n;
    const uint8_t* const mflimitPlusOne = iend - LZ4_MFLIMIT + 1;
    const uint8_t* const matchlimit = iend - LZ4_LASTLITERALS;
    uint8_t* op = dst;
    uint8_t* const olimit = dst + cap;
    uint32_t forwardH;
#define IDX(p) ((uint32_t)(startIndex + (uint32_t)((p) - src)))
#define POS(i) (src + (int64_t)(int32_t)((i) - startIndex))

    if (n < LZ4_MINLENGTH) goto last_literals;
    st->table[lz4_hash5(ip)] = IDX(ip);
    ip++;
    forwardH = lz4_hash5(ip);

    for (;;) {
        const uint8_t* match;
        const uint8_t* lowLimit;
        uint8_t* token;
        uint32_t offset;
        {
            const uint8_t* forwardIp = ip;
            int step = 1;
            int searchMatchNb = sqo_lz4_acceleration << LZ4_SKIPTRIGGER;
            do {
                const uint32_t h = forwardH;
                const uint32_t current = IDX(forwardIp);
                const uint32_t matchIndex = st->table[h];
                ip = forwardIp;
                forwardIp += step;
                step = (searchMatchNb++ >> LZ4_SKIPTRIGGER);
                if (forwardIp > mflimitPlusOne) goto last_literals;
                match = POS(matchIndex);
                lowLimit = s0 + ((matchIndex < startIndex) ? low_dict : low_in);
                forwardH = lz4_hash5(forwardIp);
                st->table[h] = current;
                if (dict_small && matchIndex < prefixIdxLimit) continue;
                if (matchIndex + LZ4_MAXD < current) continue;
                if (rd32(match) == rd32(ip)) { offset = current - matchIndex; break; }
            } while (1);
        }
        while ((ip > anchor) && (match > lowLimit) && (ip[-1] == match[-1])) { ip--; match--; }
        {
            const unsigned litLength = (unsigned)(ip - anchor);
            token = op++;
            if (op + litLength + (2 + 1 + LZ4_LASTLITERALS) + (litLength / 255) > olimit) return 0;
            if (litLength >= LZ4_RUNMASK) {
                int len = (int)(litLength - LZ4_RUNMASK);
                *token = (uint8_t)(LZ4_RUNMASK << LZ4_MLBITS);
                for (; len >= 255; len -= 255) *op++ = 255;
                *op++ = (uint8_t)len;
            } else {
                *token = (uint8_t)(litLength << LZ4_MLBITS);
            }
            memcpy(op, anchor, litLength);
            op += litLength;
        }
    next_match:
        *op++ = (uint8_t)offset;
        *op++ = (uint8_t)(offset >> 8);
        {
            unsigned matchCode;
            {
                /* extDict: count to the dictionary's end, then on from the block's first byte -- the same bytes as
                 * one count over the logically contiguous stream */
                const uint8_t* pi = ip + LZ4_MINMATCH;
                const uint8_t* pm = match + LZ4_MINMATCH;
                while (pi < matchlimit && *pi == *pm) { pi++; pm++; }
                matchCode = (unsigned)(pi - (ip + LZ4_MINMATCH));
            }
            ip += (size_t)matchCode + LZ4_MINMATCH;
            if (op + (1 + LZ4_LASTLITERALS) + (matchCode + 240) / 255 > olimit) return 0;
            if (matchCode >= LZ4_MLMASK) {
                *token = (uint8_t)(*token + LZ4_MLMASK);
                matchCode -= LZ4_MLMASK;
                while (matchCode >= 255) { *op++ = 255; matchCode -= 255; }
                *op++ = (uint8_t)matchCode;
            } else {
                *token = (uint8_t)(*token + matchCode);
            }
        }
        anchor = ip;
        if (ip >= mflimitPlusOne) break;
        st->table[lz4_hash5(ip - 2)] = IDX(ip - 2);
        {
            const uint32_t h = lz4_hash5(ip);
            const uint32_t current = IDX(ip);
            const uint32_t matchIndex = st->table[h];
            match = POS(matchIndex);
            lowLimit = s0 + ((matchIndex < startIndex) ? low_dict : low_in);
            st->table[h] = current;
            if ((dict_small ? (matchIndex >= prefixIdxLimit) : 1) && (matchIndex + LZ4_MAXD >= current) &&
                (rd32(match) == rd32(ip))) {
                token = op++;
                *token = 0;
                offset = current - matchIndex;
                goto next_match;
            }
        }
        forwardH = lz4_hash5(++ip);
    }

last_literals:
    {
        const size_t lastRun = (size_t)(iend - anchor);
        if (op + lastRun + 1 + ((lastRun + 255 - LZ4_RUNMASK) / 255) > olimit) return 0;
        if (lastRun >= LZ4_RUNMASK) {
            size_t acc = lastRun - LZ4_RUNMASK;
            *op++ = (uint8_t)(LZ4_RUNMASK << LZ4_MLBITS);
            for (; acc >= 255; acc -= 255) *op++ = 255;
            *op++ = (uint8_t)acc;
        } else {
            *op++ = (uint8_t)(lastRun << LZ4_MLBITS);
        }
        memcpy(op, anchor, lastRun);
        op += lastRun;
    }
#undef IDX
#undef POS
    return (int)(op - dst);
}

typedef struct {
    lz4_stream_model st;
    const uint8_t* s0;
    uint8_t* op;
    size_t blockSize;
} lz4f_model;

/* LZ4F_makeBlock: compress with capacity n-1, store raw when that fails */
static void lz4f_make_block(lz4f_model* f, uint64_t start, size_t n, int space, uint64_t addr)
{
    int c = lz4_block_continue(&f->st, f->s0, start, (int)n, space, addr, f->op + 4, (int)n - 1);
    uint32_t field;
    if (c == 0) { field = (uint32_t)n | 0x80000000u; memcpy(f->op + 4, f->s0 + start, n); c = (int)n; }
    else field = (uint32_t)c;
    f->op[0] = (uint8_t)field; f->op[1] = (uint8_t)(field >> 8); f->op[2] = (uint8_t)(field >> 16); f->op[3] = (uint8_t)(field >> 24);
    f->op += 4 + (size_t)c;
}

/* LZ4_saveDict(stream, tmpBuff, 64 KB) */
static uint32_t lz4f_save_dict(lz4f_model* f)
{
    uint32_t d = 64u << 10;
    if (d > f->st.dictSize) d = f->st.dictSize;
    f->st.dict_space = 2; f->st.dict_addr = 0; f->st.dictSize = d;
    return d;
}

size_t sqo_lz4_encode_serial(const uint8_t* src, size_t n, uint8_t* dst, size_t framestep, int blocksize_id)
{
    if (blocksize_id < 4 || blocksize_id > 7 || framestep == 0) return 0;
    lz4f_model f;
    memset(&f.st, 0, sizeof(f.st));
    f.s0 = src; f.blockSize = lz4f_block_bytes[blocksize_id];
    /* LZ4F_compressBegin */
    uint8_t* op = dst;
    op[0] = 0x04; op[1] = 0x22; op[2] = 0x4D; op[3] = 0x18;
    op[4] = 0x40;
    op[5] = (uint8_t)(blocksize_id << 4);
    op[6] = (uint8_t)((sqo_xxh32(op + 4, 2, 0) >> 8) & 0xff);
    f.op = op + 7;
    const size_t blockSize = f.blockSize;
    const size_t maxBufferSize = blockSize + (128u << 10);
    size_t tmpIn = 0, tmpInSize = 0;           /* offset of tmpIn in tmpBuff, bytes parked there */
    uint64_t tmp_logical = 0;                  /* logical position of the first parked byte */
    const size_t n_steps = (n + framestep - 1) / framestep;
    uint64_t pos = 0;
    for (size_t s = 0; s < n_steps; ++s) {
        const size_t src_size = (n - pos) < framestep ? (size_t)(n - pos) : framestep;
        /* ---- LZ4F_compressUpdate ---- */
        uint64_t srcPtr = pos;
        const uint64_t srcEnd = pos + src_size;
        int last = 0;                          /* 1 = fromTmpBuffer, 2 = fromSrcBuffer */
        if (tmpInSize > 0) {
            const size_t sizeToCopy = blockSize - tmpInSize;
            if (sizeToCopy > src_size) { tmpInSize += src_size; srcPtr = srcEnd; }
            else {
                last = 1;
                srcPtr += sizeToCopy;
                lz4f_make_block(&f, tmp_logical, blockSize, 2, tmpIn);
                tmpIn += blockSize;
                tmpInSize = 0;
            }
        }
        while (srcEnd - srcPtr >= blockSize) {
            last = 2;
            lz4f_make_block(&f, srcPtr, blockSize, 1, srcPtr);
            srcPtr += blockSize;
        }
        if (last == 2) tmpIn = lz4f_save_dict(&f);
        if (tmpIn + blockSize > maxBufferSize) tmpIn = lz4f_save_dict(&f);
        if (srcPtr < srcEnd) { tmp_logical = srcPtr; tmpInSize = (size_t)(srcEnd - srcPtr); }
        pos += src_size;
    }
    /* ---- LZ4F_compressEnd: flush, end mark ---- */
    if (tmpInSize > 0) lz4f_make_block(&f, tmp_logical, tmpInSize, 2, tmpIn);
    f.op[0] = f.op[1] = f.op[2] = f.op[3] = 0;
    f.op += 4;
    return (size_t)(f.op - dst);
}

size_t sqo_lz4_encode_chunked(const uint8_t* src, size_t n, uint8_t* dst, size_t chunk, int blocksize_id)
{
    if (blocksize_id < 4 || blocksize_id > 7) return 0;
    if (chunk == 0) return 0;
    uint8_t* op = dst;
    if (n == 0) return lz4f_write_single_block_frame(src, 0, dst, blocksize_id);
    for (size_t off = 0; off < n; off += chunk) {
        const size_t len = (n - off < chunk) ? (n - off) : chunk;
        /* encode_parallel hands every chunk to encode_serial with framestep = chunk (lz4_utils.hpp:245-251) */
        if (chunk > lz4f_block_bytes[blocksize_id]) op += sqo_lz4_encode_serial(src + off, len, op, chunk, blocksize_id);
        else op += lz4f_write_single_block_frame(src + off, len, op, blocksize_id);
    }
    return (size_t)(op - dst);
}

size_t sqo_lz4_decode_frames(const uint8_t* src, size_t n, uint8_t* dst, size_t cap)
{
    const uint8_t* ip = src;
    const uint8_t* const iend = src + n;
    uint8_t* op = dst;
    while (ip < iend) {
        if (iend - ip < 7) return (size_t)-1;
        if (!(ip[0] == 0x04 && ip[1] == 0x22 && ip[2] == 0x4D && ip[3] == 0x18)) return (size_t)-1;
        const unsigned flg = ip[4];
        if ((flg >> 6) != 1) return (size_t)-1;
        if (flg & 0x0D) return (size_t)-1; /* content size / checksum / dictID variants not produced by sqeazy */
        const int block_checksum = (flg >> 4) & 1;
        ip += 7;
        uint8_t* const frame_start = op;
        for (;;) {
            if (iend - ip < 4) return (size_t)-1;
            const uint32_t field = rd32(ip);
            ip += 4;
            if (field == 0) break;
            const uint32_t bsz = field & 0x7FFFFFFFu;
            if ((size_t)(iend - ip) < bsz) return (size_t)-1;
            if (field & 0x80000000u) {
                if ((size_t)(dst + cap - op) < bsz) return (size_t)-1;
                memcpy(op, ip, bsz);
                op += bsz;
            } else {
                /* linked blocks: matches may reach back into earlier blocks of this frame */
                const uint8_t* bp = ip;
                const uint8_t* const bend = ip + bsz;
                while (bp < bend) {
                    const unsigned token = *bp++;
                    size_t lit = token >> 4;
                    if (lit == 15) { unsigned s; do { if (bp >= bend) return (size_t)-1; s = *bp++; lit += s; } while (s == 255); }
                    if ((size_t)(bend - bp) < lit || (size_t)(dst + cap - op) < lit) return (size_t)-1;
                    memcpy(op, bp, lit); op += lit; bp += lit;
                    if (bp >= bend) break;
                    if (bend - bp < 2) return (size_t)-1;
                    const size_t offset = (size_t)bp[0] | ((size_t)bp[1] << 8);
                    bp += 2;
                    if (offset == 0 || offset > (size_t)(op - frame_start)) return (size_t)-1;
                    size_t ml = token & 15u;
                    if (ml == 15) { unsigned s; do { if (bp >= bend) return (size_t)-1; s = *bp++; ml += s; } while (s == 255); }
                    ml += LZ4_MINMATCH;
                    if ((size_t)(dst + cap - op) < ml) return (size_t)-1;
                    const uint8_t* m = op - offset;
                    for (size_t i = 0; i < ml; ++i) op[i] = m[i];
                    op += ml;
                }
            }
            ip += bsz;
            if (block_checksum) ip += 4;
        }
    }
    return (size_t)(op - dst);
}

/* ------------------------------------------------------------------------------------------ */
/* base64 -- base64.hpp:135-162 (standard alphabet, '=' padding)                               */
/* ------------------------------------------------------------------------------------------ */
size_t sqo_base64_encode(const uint8_t* src, size_t n, char* dst)
{
    static const char tbl[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    size_t o = 0, i = 0;
    for (; i + 2 < n; i += 3) {
        const unsigned v = ((unsigned)src[i] << 16) | ((unsigned)src[i + 1] << 8) | src[i + 2];
        dst[o++] = tbl[(v >> 18) & 63]; dst[o++] = tbl[(v >> 12) & 63]; dst[o++] = tbl[(v >> 6) & 63]; dst[o++] = tbl[v & 63];
    }
    if (n - i == 1) {
        const unsigned v = (unsigned)src[i] << 16;
        dst[o++] = tbl[(v >> 18) & 63]; dst[o++] = tbl[(v >> 12) & 63]; dst[o++] = '='; dst[o++] = '=';
    } else if (n - i == 2) {
        const unsigned v = ((unsigned)src[i] << 16) | ((unsigned)src[i + 1] << 8);
        dst[o++] = tbl[(v >> 18) & 63]; dst[o++] = tbl[(v >> 12) & 63]; dst[o++] = tbl[(v >> 6) & 63]; dst[o++] = '=';
    }
    return o;
}

/* ------------------------------------------------------------------------------------------------
 * raster_reorder (encoders/raster_reorder_scheme_impl.hpp:97-127, raster_reorder_utils.hpp:36-367):
 * the 3-D index space is cut into tiles of tile_size^3 (remainder tiles at the high end of a dimension),
 * tiles are appended one after the other in (z,y,x) tile order, each tile row-major inside.
 *   out[ tile_offset(tz,ty,tx) + (z%ts)*ey*ex + (y%ts)*ex + (x%ts) ] = in[z,y,x]
 * with (ez,ey,ex) the tile's extents and tile_offset the sizes of all tiles before it.
 * The reference's result is undefined (overlapping writes / unwritten output) when only SOME dimensions
 * have a remainder (the last tile of a remainder-free dimension gets extent 0, :271-305) and when
 * tile_size is a proper multiple of the SSE block of 16/sizeof(T) elements on a remainder-free shape
 * (encode_full_simd overwrites the tile row's head, :160-243): both return -1 here.
 * ---------------------------------------------------------------------------------------------- */
static int raster_geometry_ok(const size_t shape[3], size_t ts, int elem_size)
{
    if (ts == 0 || shape[0] == 0 || shape[1] == 0 || shape[2] == 0) return 0;
    const size_t r0 = shape[0] % ts, r1 = shape[1] % ts, r2 = shape[2] % ts;
    const int nrem = (r0 != 0) + (r1 != 0) + (r2 != 0);
    if (nrem != 0 && nrem != 3) return 0;
    const size_t block = 16 / (size_t)elem_size;
    if (nrem == 0 && ts % block == 0 && ts != block) return 0;
    return 1;
}

static size_t raster_offset(const size_t shape[3], size_t ts, size_t z, size_t y, size_t x)
{
    const size_t Z = shape[0], Y = shape[1], X = shape[2];
    const size_t tz = z / ts, ty = y / ts, tx = x / ts;
    const size_t ez = (tz + 1) * ts <= Z ? ts : Z - tz * ts;
    const size_t ey = (ty + 1) * ts <= Y ? ts : Y - ty * ts;
    const size_t ex = (tx + 1) * ts <= X ? ts : X - tx * ts;
    /* tiles before: whole tile layers, whole tile rows of this layer, tiles of this row */
    const size_t tile_off = tz * ts * Y * X + ez * (ty * ts * X + ey * tx * ts);
    return tile_off + (z % ts) * ey * ex + (y % ts) * ex + (x % ts);
}

int sqo_raster_reorder(const void* in, void* out, const size_t shape[3], size_t tile_size, int elem_size, int decode)
{
    if (!raster_geometry_ok(shape, tile_size, elem_size)) return -1;
    const size_t Z = shape[0], Y = shape[1], X = shape[2];
    const uint8_t* s = (const uint8_t*)in;
    uint8_t* d = (uint8_t*)out;
    for (size_t z = 0; z < Z; ++z)
        for (size_t y = 0; y < Y; ++y)
            for (size_t x = 0; x < X; ++x) {
                const size_t lin = (z * Y + y) * X + x, off = raster_offset(shape, tile_size, z, y, x);
                if (decode) memcpy(d + lin * (size_t)elem_size, s + off * (size_t)elem_size, (size_t)elem_size);
                else memcpy(d + off * (size_t)elem_size, s + lin * (size_t)elem_size, (size_t)elem_size);
            }
    return 0;
}
