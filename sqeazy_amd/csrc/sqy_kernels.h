// sqy_kernels.h -- launchers of the HIP kernels in sqy_kernels.hip (device pointers, explicit stream).
#ifndef SQY_KERNELS_H_
#define SQY_KERNELS_H_

#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace sqy {

// the duplicate search's tables, handed to launch_lz4_chunks when the decision per chunk is made by the chunk's own parse wavefront
// (launch_lz4_dedupe(.., fused) fills it in); chunk_key == nullptr: not in use
struct Lz4DedupeArgs {
    const uint64_t* chunk_key = nullptr;
    const uint64_t* tab_key = nullptr;
    const uint32_t* tab_val = nullptr;
    uint32_t tab_mask = 0;
    uint32_t* dup_of = nullptr;          // out: dup_of[k] = k, or the earlier chunk that chunk k equals byte for byte
    const uint32_t* piece_hash = nullptr;
    const uint64_t* holes_map = nullptr;
    uint64_t nchunks_full = 0;
    const uint32_t* digest = nullptr;    // the noise digest the transpose left (launch_bitswap1_u16), digest_stride words per chunk
    uint32_t digest_stride = 0;
};
// bitswap1: bit-plane transpose of `len` elements (encoders/bitswap_scheme_impl.hpp:97-145)
// piece_hash != nullptr (bitswap1_piece_hash_words(..) words, only offered when that is non-zero): a hash of every 1 KiB piece of
// plane data is left there for launch_lz4_dedupe
// gap_chunk != 0 ("frames in place", see launch_lz4_tail_marks): the plane stream is written as the bodies of the LZ4 frames it
// will be cut into -- chunk k (gap_chunk bytes, a power of two) at out + k * (gap_chunk + 15), `out` any alignment; needs
// len % 8192 == 0
// side != nullptr (launch_diff3x3x1_side): columns x < side_w of every row of X voxels are read from the compact buffer `side`
// digest != nullptr (round 6, frames in place only, len / 8 a multiple of gap_chunk): the NOISE DIGEST -- bucket << 16 | tag of the five bytes at
// every position liblz4's search probes from probe 961 on when it starts with a chunk and finds nothing, digest_stride
// (= lz4_noise_digest_stride(gap_chunk)) words per chunk of the plane stream; the LZ4 parse of a chunk takes its batches from there as long
// as nothing has matched (Lz4DedupeArgs::digest) instead of reading the plane bytes again
hipError_t launch_bitswap1_u16(const uint16_t* in, uint16_t* out, uint64_t len, hipStream_t stream, uint32_t* piece_hash = nullptr,
                               uint32_t gap_chunk = 0, const uint16_t* side = nullptr, uint32_t side_w = 0, uint32_t X = 0,
                               uint32_t* digest = nullptr, uint32_t digest_stride = 0);
void set_bitswap1_blocks_per_cu(long n);      // workgroups (two waves) of the in-place transposer per CU (default 32: as many as fit)
uint32_t lz4_noise_digest_stride(uint32_t chunk);      // words per chunk, 0 = no digest for this chunk size
uint64_t bitswap1_piece_hash_words(const void* in, const void* out, uint64_t len);
// duplicate chunks of a plane stream (chunk a multiple of 1 KiB): dup_of[k] = the earliest chunk with the same bytes (k itself when
// there is none); the hashes only nominate, a byte compare decides.  work: lz4_dedupe_work_bytes(nchunks) bytes
uint64_t lz4_dedupe_work_bytes(uint64_t nchunks);
// in_stride (all LZ4 launchers; 0 = chunk): chunk k of the stream starts at in + k * in_stride
// holes (with gap_chunk): launch_bitswap1_u16 leaves the all-zero 1 KiB pieces of the stream UNWRITTEN (their hash is the exact zero
// marker, 0); launch_lz4_dedupe(.., holes_map: scratch of lz4_holes_map_bytes) compares through the markers and then fills the pieces
// in (writes into `in`) for every chunk anybody will read -- the bit planes above the data's range are neither written nor read
hipError_t launch_lz4_dedupe(const uint8_t* in, uint64_t total, uint32_t chunk, const uint32_t* piece_hash, void* work,
                             uint32_t* dup_of, hipStream_t stream, uint64_t in_stride = 0, uint64_t* holes_map = nullptr,
                             bool table_is_clear = false, Lz4DedupeArgs* fused = nullptr);
// the search's table emptied and *zero_word = 0 by one small kernel (a call launches it in front of its bit-plane transpose)
hipError_t launch_lz4_dedupe_clear(void* work, uint64_t nchunks, uint32_t* zero_word, hipStream_t stream);
uint64_t lz4_holes_map_bytes(uint64_t nchunks, uint32_t chunk);
hipError_t launch_bitswap1_u8(const uint8_t* in, uint8_t* out, uint64_t len, hipStream_t stream);

// diff3x3x1 on a {Z,Y,X} volume of 1- or 2-byte unsigned voxels (encoders/diff_scheme_impl.hpp:78-139)
// diff3x3x1 with a 16-bit bit-plane transpose right behind it: the stage can only touch columns x < 1 + hx, hx = Z - 2 (the reference
// takes the row extent from the DEPTH).  side_width = those columns rounded up to the 128 voxels a lane of the transpose
// owns (0: not applicable -- other geometry, 8-bit, or no column left out); launch_diff3x3x1_side writes just them (rows side_w
// voxels apart, Z*Y*side_w voxels), launch_bitswap1_u16(.., side, side_w, X) reads them from there and the rest from the input.
uint32_t diff3x3x1_side_width(uint64_t Z, uint64_t Y, uint64_t X, int elem_size);
hipError_t launch_diff3x3x1_side(const uint16_t* in, uint16_t* side, uint64_t Z, uint64_t Y, uint64_t X, uint32_t side_w, hipStream_t stream);
// schar (8-bit only): the stage as a tail filter on the sink's `char` output -- signed bytes, the sum sign-extended before the division
hipError_t launch_diff3x3x1(const void* in, void* out, uint64_t Z, uint64_t Y, uint64_t X, int elem_size, hipStream_t stream,
                            bool schar = false);

// LZ4: every `chunk` bytes of in[0,total) compressed on its own into scratch + k*stride (capacity chunk-1);
// csize[k] = compressed bytes, 0 when the chunk has to be stored raw.
// frame_map != nullptr: the stream is frame frame_map[f] of `in` for f = 0, 1, .. (frame_bytes each, a multiple of chunk),
// read in place (frame_shuffle directly in front of lz4)
// redo != nullptr (nchunks + 1 words): chunks that turn out to be streams of short sequences are not finished but listed
// there (redo[0] = count, redo[1..] = chunk numbers); launch_lz4_chunks_dense then parses exactly those
hipError_t launch_lz4_chunks(const uint8_t* in, uint64_t total, uint32_t chunk, uint8_t* scratch, uint64_t stride,
                             uint32_t* csize, uint64_t nchunks, hipStream_t stream, const uint64_t* frame_map = nullptr,
                             uint64_t frame_bytes = 0, uint32_t* redo = nullptr, const uint32_t* dup_of = nullptr, uint64_t in_stride = 0,
                             uint32_t acceleration = 1,      // liblz4's acceleration (1, or k + 1 for sqeazy's lz4(accel=-k))
                             bool redo_is_zero = false,      // redo[0] has been zeroed already (launch_lz4_dedupe_clear)
                             const Lz4DedupeArgs* dedupe = nullptr);
hipError_t launch_lz4_chunks_dense(const uint8_t* in, uint64_t total, uint32_t chunk, uint8_t* scratch, uint64_t stride,
                                   uint32_t* csize, uint32_t* redo, uint32_t redo_count, hipStream_t stream,
                                   const uint64_t* frame_map = nullptr, uint64_t frame_bytes = 0, uint64_t in_stride = 0);
// One block of a block-linked LZ4 frame (liblz4's LZ4F_blockLinked, what lz4::encode_serial produces: lz4_utils.hpp:99-173):
// where it sits in the stream and how far liblz4's backward catch-up may move a match that starts inside the block
// (low_in) or in the history in front of it (low_dict); stream offsets, may lie below `start - 64 KiB`.
struct Lz4Block {
    uint64_t start;          // byte offset of the block in the stream
    uint32_t n;              // bytes
    uint32_t flags;          // bit 0: first block of its frame (fresh LZ4 stream), bit 1: last block of its frame
    int64_t low_in, low_dict;
};
// block-linked frames: wavefront f compresses blocks [frame_first[f], frame_first[f+1]) in order, hash table carried from
// block to block; block k -> scratch + k*stride, csize[k] (0 = store raw).  max_block = largest blocks[k].n (<= 4 MiB)
hipError_t launch_lz4_linked(const uint8_t* in, const Lz4Block* blocks, const uint32_t* frame_first, uint64_t nframes,
                             uint32_t max_block, uint8_t* scratch, uint64_t stride, uint32_t* csize, hipStream_t stream, uint32_t acceleration = 1);
// Block-linked frames, block-parallel (round 4).  liblz4's table only matters as far as it holds positions of the last 64 KiB, so a
// wavefront can rebuild the table a block starts from by parsing the blocks in front of it (>= 64 KiB of them, output thrown away)
// from an empty table -- a guess that is almost always right and is CHECKED: every parse leaves the table it started from and the
// table it ended with in `tables`, launch_lz4_linked_verify compares block k's start with what block k - 1 really left (after the
// same re-basing; entries more than 64 KiB behind are parked at one value by it, so equality is equivalence), and the blocks that
// fail are parsed again from the true table (mode 2, runs of blocks in order).  When every block verifies, induction from the frame's
// first block (a fresh table, no guess) makes every block's output liblz4's.
//   mode 1: wavefront w walks blocks [wave_first[w], wave_last[w]]; the first starts from an empty table; only the last one's output,
//           size and tables count (the others are the warm-up)
//   mode 2: wavefront w walks blocks [wave_first[w], wave_last[w]], all of them for real; the first starts from the table block
//           wave_first[w] - 1 left (tables) unless it opens a frame
// tables: nblocks x 2 x 4096 words ([k][0] = the table block k was parsed from, [k][1] = the table it left)
struct Lz4SpecArgs {
    const uint32_t* wave_first = nullptr;
    const uint32_t* wave_last = nullptr;
    uint32_t* tables = nullptr;
    uint32_t mode = 0;
};
constexpr uint64_t kLz4SpecTableWords = 2 * 4096;
hipError_t launch_lz4_linked_spec(const uint8_t* in, const Lz4Block* blocks, const Lz4SpecArgs& spec, uint64_t nwaves,
                                  uint32_t max_block, uint8_t* scratch, uint64_t stride, uint32_t* csize, hipStream_t stream, uint32_t acceleration = 1);
// ok[k] = 1 when block k opens a frame or started from the table block k - 1 left, else 0
hipError_t launch_lz4_linked_verify(const Lz4Block* blocks, uint64_t nblocks, const uint32_t* tables, uint32_t max_block, uint32_t* ok, hipStream_t stream);
// frame_off[k] = byte offset of frame k in the concatenated stream, frame_off[nchunks] = total payload bytes
// (blocks != nullptr: offset of what block k contributes -- frame header if it opens a frame, size field, body, end mark
// if it closes one)
hipError_t launch_lz4_frame_scan(const uint32_t* csize, uint64_t nchunks, uint64_t total, uint32_t chunk,
                                 uint64_t* frame_off, hipStream_t stream, const Lz4Block* blocks = nullptr,
                                 const uint32_t* dup_of = nullptr, uint64_t* tail_info = nullptr,
                                 // frames in place, one host round trip per call: guard[0] != 0 (chunks left to the dense pass) makes the
                                 // kernel return at once; body0 != nullptr: the stored tail's frame marks are written here as well
                                 const uint32_t* guard = nullptr, uint8_t* body0 = nullptr, uint64_t in_stride = 0, uint32_t bd_byte = 0,
                                 uint32_t hc_byte = 0);
// Frames in place: everything between the frame scan and the finished blob, driven from the device -- the stored chunks in front of
// the tail put aside, the frames in front of the tail gathered up against it, the sqy header (hdr_prefix | payload bytes in decimal |
// hdr_suffix, padded in front to a multiple of elem_size) written in front of them.  record (pinned host memory, 7 words) takes
// [0] 1 done / 2 dense pass needed (guard[0] != 0: nothing was touched) / 3 no room for the header, [1] blob offset in `out`,
// [2] blob bytes, [3] payload bytes, [4] chunks in front of the stored tail, [5] stored chunks among them, [6] guard[0].
// (round 6) the same in one kernel; record[0] = 4 when stored chunks sit in front of the stored tail (then: the kernels above / below)
hipError_t launch_lz4_inplace_tail_fused(uint8_t* out, uint64_t t0, uint64_t in_stride, uint64_t total, uint32_t chunk, uint64_t nchunks,
                                         const uint8_t* scratch, uint64_t stride, const uint32_t* csize, uint64_t* frame_off, const uint32_t* dup_of,
                                         uint64_t* tail_info, uint32_t bd_byte, uint32_t hc_byte, const char* hdr_prefix, uint32_t prefix_len,
                                         const char* hdr_suffix, uint32_t suffix_len, uint32_t elem_size, const uint32_t* guard, uint64_t* record,
                                         hipStream_t stream);
hipError_t launch_lz4_inplace_tail(uint8_t* out, uint64_t t0, uint64_t in_stride, uint64_t total, uint32_t chunk, uint64_t nchunks,
                                   uint8_t* scratch, uint64_t stride, const uint32_t* csize, const uint64_t* frame_off, const uint32_t* dup_of,
                                   const uint64_t* tail_info, uint32_t bd_byte, uint32_t hc_byte, const char* hdr_prefix, uint32_t prefix_len,
                                   const char* hdr_suffix, uint32_t suffix_len, uint32_t elem_size, const uint32_t* guard, uint64_t* record,
                                   hipStream_t stream);
constexpr uint32_t kLz4InplaceHeaderTextMax = 3000;   // prefix + suffix bytes the finish kernel takes as an argument
// Frames in place (chunked layout; the stage in front wrote chunk k of the stream at body0 + k * in_stride, in_stride = chunk + 15):
// tail_info (4 words, from the scan) = {j, bytes of frames 0..j-1, stored chunks among them, payload bytes}, j = first chunk of
// the run of stored chunks that ends the stream.  Those are final where they stand: tail_marks writes header / size field / end
// mark around them.  The frames in front are gathered (launch_lz4_frame_gather over j chunks) to END at body0 - 11 + j * in_stride;
// stored chunks among THEM are first put aside in their scratch slots (stash_raw; gather then with raw_from_scratch).
hipError_t launch_lz4_tail_marks(uint8_t* body0, uint64_t in_stride, uint64_t total, uint32_t chunk, uint64_t nchunks, uint32_t bd_byte,
                                 uint32_t hc_byte, const uint64_t* tail_info, hipStream_t stream);
hipError_t launch_lz4_stash_raw(const uint8_t* body0, uint64_t in_stride, uint64_t total, uint32_t chunk, uint8_t* scratch, uint64_t stride,
                                const uint32_t* csize, const uint32_t* dup_of, uint64_t nhead, hipStream_t stream);
// writes [04 22 4D 18 | 40 | BD | HC][u32 size][data][00 00 00 00] per chunk at out + frame_off[k]
hipError_t launch_lz4_frame_gather(const uint8_t* in, uint64_t total, uint32_t chunk, const uint8_t* scratch, uint64_t stride,
                                   const uint32_t* csize, const uint64_t* frame_off, uint8_t* out, uint32_t bd_byte,
                                   uint32_t hc_byte, uint64_t nchunks, hipStream_t stream, const uint64_t* frame_map = nullptr,
                                   uint64_t frame_bytes = 0, const Lz4Block* blocks = nullptr, const uint32_t* dup_of = nullptr,
                                   uint64_t in_stride = 0, bool raw_from_scratch = false);

// quantiser: 65536-bin histogram of u16 voxels (histo is zeroed by the launcher), and out[i] = lut[in[i]]
hipError_t launch_histogram_u16(const uint16_t* in, uint64_t len, uint32_t* histo, hipStream_t stream);
hipError_t launch_quantiser_apply_u16(const uint16_t* in, uint8_t* out, uint64_t len, const uint8_t* lut, hipStream_t stream);
// .. with the 8-bit bit-plane transpose of the sink's bytes in the same pass (quantiser->bitswap1; `in` 16-byte aligned): out = the 8
// plane segments of len / 8 bytes, MSB plane first, + the len % 8 tail
hipError_t launch_quantiser_apply_bitswap1_u8(const uint16_t* in, uint8_t* out, uint64_t len, const uint8_t* lut, hipStream_t stream);

// raster_reorder (encoders/raster_reorder_utils.hpp): tiles of tile_size^3 (remainder tiles at the high ends) appended in
// (z,y,x) tile order, row-major inside; decode = the inverse permutation
hipError_t launch_raster_reorder(const void* in, void* out, uint64_t Z, uint64_t Y, uint64_t X, uint64_t tile_size, int elem_size,
                                 bool decode, hipStream_t stream);

// bitshuffle (bshuf_bitshuffle / bshuf_bitunshuffle of kiyo-masui/bitshuffle) of n_elems elements of 1 or 2 bytes in blocks of
// block_elems elements (a multiple of 8)
hipError_t launch_bitshuffle(const void* in, void* out, uint64_t n_elems, int elem_size, uint64_t block_elems, bool decode, hipStream_t stream);

// frame_shuffle: per-frame mean in the reference's sequential binary32 order; frame gather out[i] = in[map[i]]
// schar (8-bit only): frames of signed bytes (the stage as a tail filter)
hipError_t launch_frame_metric(const void* in, uint64_t Z, uint64_t per_frame, int elem_size, float* metric, hipStream_t stream,
                               void* scratch = nullptr, uint64_t scratch_bytes = 0, bool schar = false);
// scratch that lets long frames take the block-parallel path (16 bytes per 4 KiB block)
uint64_t frame_metric_scratch_bytes(uint64_t Z, uint64_t per_frame, int elem_size);
hipError_t launch_frame_gather(const void* in, void* out, uint64_t Z, uint64_t frame_bytes, const uint64_t* map, hipStream_t stream);

// ---- decode ----
// index of every LZ4 block of the concatenated frames: blk[i] = {data offset lo, hi, size | raw << 31, block id in frame}
hipError_t launch_lz4_frame_index(const uint8_t* in, uint64_t n, void* blk, uint32_t* frame_first, uint64_t max_blocks,
                                  uint32_t* counts /* frames, blocks, error */, hipStream_t stream);
// parallel variant for the chunked layout (single-block frames): counts[2] == 100 means "not covered, run the serial walk"
uint64_t lz4_frame_rank_scratch_bytes(uint64_t expected_frames);
hipError_t launch_lz4_frame_rank(const uint8_t* in, uint64_t n, void* blk, uint32_t* frame_first, uint64_t max_blocks,
                                 uint32_t* counts, uint64_t expected_frames, void* scratch, hipStream_t stream, uint64_t chunk = 0, uint64_t last = 0);
// (chunk, last: the bytes every frame but the last / the last frame decodes to -- frames at the stream's end that are STORED blocks of
//  those sizes are found where they must start instead of by the scan; 0 = scan everything.  counts: 16 words, [6] = frames found so)
// frame f decodes to out + f*frame_stride; every block decodes to at most block_bytes
// ONE block-linked frame (the serial layout) decoded block-parallel: every block at once with the history as an unknown (16-bit
// references in `refs`, out_bytes words), the tails resolved in order by one workgroup, the rest at once.  blk: the frame's blocks
// (launch_lz4_frame_index).  *errflag != 0 afterwards: not a frame of full blocks, or damaged -- the one-wavefront walk decides.
bool lz4_linked_decode_parallel_possible(uint32_t nblocks, uint64_t out_bytes, uint64_t block_bytes);
// scan_scratch (lz4_linked_decode_scan_scratch_bytes(nblocks) bytes, optional): long frames resolve their tails as a scan over ranges of
// blocks instead of one walk
uint64_t lz4_linked_decode_scan_scratch_bytes(uint32_t nblocks);
hipError_t launch_lz4_linked_decode_parallel(const uint8_t* in, const void* blk, uint32_t nblocks, uint8_t* out, uint16_t* refs, uint64_t out_bytes,
                                             uint64_t block_bytes, uint32_t* errflag, hipStream_t stream, uint8_t* scan_scratch = nullptr);
hipError_t launch_lz4_frames_decode(const uint8_t* in, const void* blk, const uint32_t* frame_first, uint32_t nframes, uint8_t* out,
                                    uint64_t out_bytes, uint64_t frame_stride, uint64_t block_bytes, uint32_t ncompressed,
                                    uint32_t* errflag, hipStream_t stream, hipStream_t copy_stream = nullptr, hipEvent_t fork = nullptr,
                                    hipEvent_t join = nullptr, const uint64_t* remap = nullptr, uint64_t remap_bytes = 0, bool two_waves = false);
// (two_waves: every frame is ONE block -- two wavefronts per frame, lz4_frames_decode2_kernel)
// (remap: the inverse of frame_shuffle folded into the decode -- the chunk that starts at byte o of the sorted stream goes to
// remap[o / remap_bytes] * remap_bytes + o % remap_bytes; remap_bytes a multiple of frame_stride, out_bytes = nframes * frame_stride)
// decode of quantiser->bitswap1: inverse transpose of the 8-bit planes and the quantiser's look-up in one pass
bool bitswap1_u8_decode_lut_possible(const void* in, const void* out, uint64_t len);
hipError_t launch_bitswap1_u8_decode_lut(const uint8_t* in, uint16_t* out, uint64_t len, const uint16_t* lut, hipStream_t stream);
hipError_t launch_bitswap1_decode(const void* in, void* out, uint64_t len, int elem_size, hipStream_t stream);
// one launch per frame over the columns the stage can touch (frame z needs the decoded frame z-1), the other columns one plain copy
// -- on copy_stream next to the chain when that, fork and join are given.  (scratch: unused since round 3)
uint64_t diff3x3x1_decode_scratch_bytes(uint64_t X);
// diff3x3x1_decode_chain_columns: how many leading columns of a row go through that chain in the usual 16-bit geometry (a multiple of 8;
// 0: another geometry).  left_tmp != nullptr (that geometry, in == out, both 16-byte aligned): the volume is decoded where it lies --
// the encoded chain columns are first copied to left_tmp (same indices, a buffer of the volume's size), nothing else moves
uint64_t diff3x3x1_decode_chain_columns(uint64_t Z, uint64_t Y, uint64_t X, int elem_size);
hipError_t launch_diff3x3x1_decode(const void* in, void* out, uint64_t Z, uint64_t Y, uint64_t X, int elem_size, void* scratch,
                                   hipStream_t stream, bool schar = false, hipStream_t copy_stream = nullptr, hipEvent_t fork = nullptr,
                                   hipEvent_t join = nullptr, void* left_tmp = nullptr);
hipError_t launch_quantiser_decode(const uint8_t* in, uint16_t* out, uint64_t len, const uint16_t* lut, hipStream_t stream);
hipError_t launch_frame_scatter(const void* in, void* out, uint64_t Z, uint64_t frame_bytes, const uint64_t* map, hipStream_t stream);

} // namespace sqy
#endif
