/*
 * ref_driver.cpp -- thin driver around the parts of the REAL reference that build in this image
 * without any stand-in header or library (TEST INFRASTRUCTURE ONLY; built into oracle/_ref/,
 * which is git-ignored; never linked into the product).
 *
 *  (1) the reference's own SSE bit-plane gather, included in place from
 *      /root/reference/src/cpp/src/encoders/sse_utils.hpp (needs only <emmintrin.h>&co + OpenMP):
 *      sqeazy::detail::simd_segment_broadcast (sse_utils.hpp:1365-1433), entered exactly as
 *      sse_bitplane_reorder_encode<1> does (bitplane_reorder_sse.hpp:281-311).
 *  (2) liblz4 1.9.3 -- the third-party library that holds ALL of the reference's LZ4 arithmetic
 *      (the reference only calls its Frame API).  It is installed in this image
 *      (/usr/lib/x86_64-linux-gnu/liblz4.so.1.9.3, headers /opt/conda/include).  The functions
 *      below drive it with the call sequence and preferences of the reference's call sites:
 *      encoders/lz4.hpp:103-113 (prefs), encoders/lz4_utils.hpp:99-173 (encode_serial),
 *      :193-274 (encode_parallel).  lz4_utils.hpp itself cannot be compiled here (it includes
 *      sqeazy_common.hpp -> Boost.Align, absent from the image), so that sequencing is restated.
 *
 * Everything else in the reference's hot path needs Boost and is therefore unbuildable here
 * (see DESIGN.md).
 */
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>
#include <numeric>
#include <algorithm>
#include <iostream>
#include <thread>
#include <climits>
#include <omp.h>

#include "encoders/sse_utils.hpp" /* from /root/reference/src/cpp/src via -I */

#include "lz4.h"
#include "lz4frame.h"

extern "C" {

int ref_lz4_version() { return LZ4_versionNumber(); }

/* bitswap_scheme<uint16_t,1>::encode, SSE branch (bitswap_scheme_impl.hpp:97-145): requires
 * len % 128 == 0 and a 16-byte aligned input; returns 1 when the reference would have taken the
 * scalar branch instead (which needs Boost to compile and is not available here). */
int ref_bitswap1_encode_u16(const uint16_t* in, uint16_t* out, size_t len, int nthreads)
{
    if (len % 128 != 0 || len == 0) return 1;
    if ((reinterpret_cast<uintptr_t>(in) & 15u) != 0) return 1;
    sqeazy::detail::simd_segment_broadcast(in, in + len, out, nthreads);
    return 0;
}

static LZ4F_preferences_t make_prefs(int accel, int blocksize_id)
{
    /* encoders/lz4.hpp:103-113 */
    LZ4F_preferences_t prefs;
    std::memset(&prefs, 0, sizeof(prefs));
    prefs.frameInfo.blockSizeID = static_cast<LZ4F_blockSizeID_t>(blocksize_id);
    prefs.frameInfo.blockMode = LZ4F_blockLinked;
    prefs.frameInfo.contentChecksumFlag = LZ4F_noContentChecksum;
    prefs.frameInfo.frameType = LZ4F_frame;
    prefs.frameInfo.contentSize = 0;
    prefs.frameInfo.dictID = 0;
    prefs.frameInfo.blockChecksumFlag = LZ4F_noBlockChecksum;
    prefs.compressionLevel = accel;
    prefs.autoFlush = 0;
    prefs.favorDecSpeed = 0;
    return prefs;
}

size_t ref_lz4f_compress_bound(size_t n, int accel, int blocksize_id)
{
    LZ4F_preferences_t prefs = make_prefs(accel, blocksize_id);
    return LZ4F_compressBound(n, &prefs);
}

int ref_lz4f_header_size_max() { return LZ4F_HEADER_SIZE_MAX; }

/* call sequence of lz4::encode_serial (lz4_utils.hpp:99-173); returns bytes written, 0 on error */
size_t ref_lz4_encode_serial(const char* in, size_t n, char* out, size_t out_bytes, size_t framestep,
                             int accel, int blocksize_id)
{
    LZ4F_preferences_t prefs = make_prefs(accel, blocksize_id);
    LZ4F_compressionContext_t ctx;
    size_t rc = LZ4F_createCompressionContext(&ctx, LZ4F_VERSION);
    if (LZ4F_isError(rc)) return 0;
    size_t written = LZ4F_compressBegin(ctx, out, out_bytes, &prefs);
    if (LZ4F_isError(written)) { LZ4F_freeCompressionContext(ctx); return 0; }
    const size_t n_steps = (n + framestep - 1) / framestep;
    const char* src = in;
    const char* src_end = in + n;
    char* dst = out + written;
    for (size_t s = 0; s < n_steps; ++s) {
        const size_t src_size = (size_t)(src_end - src) < framestep ? (size_t)(src_end - src) : framestep;
        const size_t m = LZ4F_compressUpdate(ctx, dst, out_bytes - written, src, src_size, nullptr);
        if (LZ4F_isError(m)) { LZ4F_freeCompressionContext(ctx); return 0; }
        src += src_size;
        written += m;
        dst += m;
    }
    rc = LZ4F_compressEnd(ctx, dst, out_bytes - written, nullptr);
    if (LZ4F_isError(rc)) { LZ4F_freeCompressionContext(ctx); return 0; }
    written += rc;
    LZ4F_freeCompressionContext(ctx);
    return written;
}

/* call sequence of lz4::encode_parallel (lz4_utils.hpp:193-274): chunk k is framed on its own
 * into out + k*maxbytes_encoded_chunk, then the blanks are removed (:175-190). */
size_t ref_lz4_encode_parallel(const char* in, size_t n, char* out, size_t out_bytes, size_t chunk,
                               int accel, int blocksize_id, int nthreads)
{
    const size_t nchunks = (n + chunk - 1) / chunk;
    if (nchunks == 1) return ref_lz4_encode_serial(in, n, out, out_bytes, chunk, accel, blocksize_id);
    if ((size_t)nthreads > nchunks) nthreads = (int)nchunks;
    const size_t stride = ref_lz4f_compress_bound(chunk, accel, blocksize_id) + LZ4F_HEADER_SIZE_MAX;
    if (nchunks * stride > out_bytes) return 0;
    std::vector<size_t> written(nchunks, 0);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long k = 0; k < (long)nchunks; ++k) {
        const char* t_in = in + (size_t)k * chunk;
        const size_t len = std::min(chunk, n - (size_t)k * chunk);
        written[k] = ref_lz4_encode_serial(t_in, len, out + (size_t)k * stride, stride, chunk, accel, blocksize_id);
    }
    char* value = out + written[0];
    for (size_t k = 1; k < nchunks; ++k) {
        std::memmove(value, out + k * stride, written[k]);
        value += written[k];
    }
    return (size_t)(value - out);
}

/* block level: LZ4_compress_fast_continue on a fresh stream (what LZ4F_makeBlock runs for the
 * first block of a block-linked frame), capacity as given. */
int ref_lz4_block_fast_continue(const char* src, int n, char* dst, int cap, int accel)
{
    LZ4_stream_t* s = LZ4_createStream();
    if (!s) return -1;
    const int r = LZ4_compress_fast_continue(s, src, dst, n, cap, accel);
    LZ4_freeStream(s);
    return r;
}

/* decoder: encoders/lz4.hpp:257-339 (LZ4F_decompress over concatenated frames) */
size_t ref_lz4_decode_frames(const char* in, size_t n, char* out, size_t cap)
{
    LZ4F_decompressionContext_t dctx;
    if (LZ4F_isError(LZ4F_createDecompressionContext(&dctx, LZ4F_VERSION))) return (size_t)-1;
    const char* src = in;
    const char* src_end = in + n;
    char* dst = out;
    char* dst_end = out + cap;
    size_t ret = 1;
    while (src < src_end) {
        size_t dsz = (size_t)(dst_end - dst);
        size_t ssz = (size_t)(src_end - src);
        ret = LZ4F_decompress(dctx, dst, &dsz, src, &ssz, nullptr);
        if (LZ4F_isError(ret)) { LZ4F_freeDecompressionContext(dctx); return (size_t)-1; }
        src += ssz;
        dst += dsz;
        if (ssz == 0 && dsz == 0) break;
    }
    LZ4F_freeDecompressionContext(dctx);
    return (size_t)(dst - out);
}

} /* extern "C" */
