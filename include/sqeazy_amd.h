/*
 * sqeazy_amd.h -- C-ABI of libsqeazy_amd.so, the MI355X-native drop-in for sqeazy's pipeline
 * encode path (filter stages + LZ4 block compression).
 *
 * Section A re-declares, with identical names, argument order and meaning, the entry points of the
 * reference's libsqeazy that belong to this path.  Each declaration cites the reference interface
 * it replaces (paths relative to /root/reference/src/cpp): `inc/sqeazy.h` for the declaration and
 * `src/sqeazy.cpp` for the behaviour.  A maintainer binds them exactly as they bind libsqeazy
 * today.  All pointers are HOST pointers; the library stages data through
 * HBM itself.  Return convention: 0 success, 1 failure ("[sqeazy]\t..." on stderr), never throws.
 *
 * Section B adds entry points for callers whose volumes already live in MI355X HBM.
 *
 * No HDF5 entry points (SQY_h5_*): out of scope of the hot path.
 */
#ifndef SQEAZY_AMD_H_
#define SQEAZY_AMD_H_

#ifdef __cplusplus
#define SQY_FUNCTION_PREFIX extern "C" __attribute__((visibility("default")))
#else
#include <stdbool.h>
#define SQY_FUNCTION_PREFIX __attribute__((visibility("default")))
#endif

/* ------------------------------------------------------------------------------------------------
 * Section A -- sqeazy's own C-ABI for the pipeline path
 * ---------------------------------------------------------------------------------------------- */

/* inc/sqeazy.h:26, src/sqeazy.cpp:16-22.  *length in: bytes available at src; out: header bytes
 * (JSON + "|01307#!" delimiter, including leading pad). */
SQY_FUNCTION_PREFIX int SQY_Header_Size(const char* src, long* length);

/* inc/sqeazy.h:41, src/sqeazy.cpp:24-33.  *num in: bytes at src; out: rank of the stored volume. */
SQY_FUNCTION_PREFIX int SQY_Decompressed_NDims(const char* src, long* num);

/* inc/sqeazy.h:56, src/sqeazy.cpp:35-46.  shape[0] in: bytes at src; out: shape[0..rank) = {z,y,x}. */
SQY_FUNCTION_PREFIX int SQY_Decompressed_Shape(const char* src, long* shape);

/* inc/sqeazy.h:70, src/sqeazy.cpp:48-58.  *Sizeof in: bytes at src; out: bytes per voxel. */
SQY_FUNCTION_PREFIX int SQY_Decompressed_Sizeof(const char* src, long* Sizeof);

/* inc/sqeazy.h:81, src/sqeazy.cpp:61-69.  version[0..3) = major, minor, patch. */
SQY_FUNCTION_PREFIX int SQY_Version_Triple(int* version);

/* inc/sqeazy.h:109-115, src/sqeazy.cpp:72-106.  Encode a uint8 volume.
 *   pipeline    e.g. "frame_shuffle->lz4"; must satisfy SQY_Pipeline_Possible_UI8
 *   src         contiguous voxels, row-major {z,y,x}, x fastest
 *   shape       long[shape_size], voxels per dimension
 *   dst         at least SQY_Pipeline_Max_Compressed_Length_* bytes
 *   dstlength   out only: bytes written (header + payload)
 *   nthreads    <=0 or > hardware threads: all hardware threads (src/sqeazy_algorithms.hpp:14-22).  The value selects
 *               the LZ4 LAYOUT exactly as in the reference (encoders/lz4.hpp:227-239): effective 1 -> ONE frame of
 *               block-linked 256 KiB blocks (lz4_utils.hpp:99-173; bit-identical).  Its blocks are parsed in parallel from hash-table
 *               guesses that are verified and parsed again, in order, where they fail: 2-4 x the chunked layout's time on ordinary
 *               stacks (3 ms against 1.4 for a 1 GiB stack, 21 ms against 5.4 for a diff3x3x1 slab of 2 GiB) -- but SECONDS where no
 *               guess holds: a stream of short sequences such as the top plane of quantised data (511 blocks in a row, one wavefront,
 *               ~10 ms per block: 4.7 s for a 2048x2048x256 slab of quantiser->bitswap1->lz4, slower than one CPU core; the library
 *               says so once on stderr).  >=2 -> one independent frame per 256 KiB chunk (lz4_utils.hpp:193-274; byte-identical for
 *               every count >= 2; the fast path: 12 ms for that slab).  Same decoder for both. */
SQY_FUNCTION_PREFIX int SQY_PipelineEncode_UI8(const char* pipeline, const char* src, long* shape, unsigned shape_size,
                                               char* dst, long* dstlength, int nthreads);

/* inc/sqeazy.h:140-146, src/sqeazy.cpp:108-142.  Same for uint16 voxels (little endian). */
SQY_FUNCTION_PREFIX int SQY_PipelineEncode_UI16(const char* pipeline, const char* src, long* shape, unsigned shape_size,
                                                char* dst, long* dstlength, int nthreads);

/* inc/sqeazy.h:159-161 / :190, src/sqeazy.cpp:144-183.  *length in: raw bytes; out: upper bound of
 * the encoded blob = 2*header + max over stages (dynamic_pipeline.hpp:866-890). */
SQY_FUNCTION_PREFIX int SQY_Pipeline_Max_Compressed_Length_UI8(const char* pipeline, long pipeline_length, long* length);
SQY_FUNCTION_PREFIX int SQY_Pipeline_Max_Compressed_Length_UI16(const char* pipeline, long pipeline_length, long* length);

/* inc/sqeazy.h:175-178 / :204-207, src/sqeazy.cpp:185-231.  *length in: strlen(pipeline) (sic); out: bound. */
SQY_FUNCTION_PREFIX int SQY_Pipeline_Max_Compressed_Length_3D_UI8(const char* pipeline, long* shape, unsigned shape_size, long* length);
SQY_FUNCTION_PREFIX int SQY_Pipeline_Max_Compressed_Length_3D_UI16(const char* pipeline, long* shape, unsigned shape_size, long* length);

/* inc/sqeazy.h:219-243, src/sqeazy.cpp:233-268.  true iff the string parses as head filters -> sink -> tail filters
 * (src/sqeazy_pipelines.hpp:31-77) AND every stage is implemented here.  Head filters: diff3x3x1, bitswap1, bitshuffle,
 * raster_reorder, tile_shuffle, frame_shuffle, zcurve_reorder; sinks: pass_through, quantiser (16-bit input; every
 * weighting_function with a finite exponent, decode_lut_path), lz4 (accel <= 2, negative = liblz4's acceleration; last stage);
 * tail filters on the sink's `char` stream: diff3x3x1, bitswap1, bitshuffle, lz4, raster_reorder, tile_shuffle, frame_shuffle,
 * zcurve_reorder -- the reference's whole list but the video codecs.  false where the reference says true: the background filters
 * (remove_background, rmbkrd_neighbor5x5x5, rmestbkrd), the video sinks / filters (h264, hevc), lz4 with accel >= 3 (LZ4HC),
 * stages behind an lz4 sink.  (A geometry the reference leaves undefined for a stage is refused at encode time: error 1.) */
SQY_FUNCTION_PREFIX bool SQY_Pipeline_Possible_UI16(const char* pipeline_string);
SQY_FUNCTION_PREFIX bool SQY_Pipeline_Possible_UI8(const char* pipeline_string);
SQY_FUNCTION_PREFIX bool SQY_Pipeline_Possible(const char* pipeline_string, int sizeofpixel);

/* inc/sqeazy.h:255, src/sqeazy.cpp:270-279.  *length in: bytes at data; out: decoded bytes. */
SQY_FUNCTION_PREFIX int SQY_Decompressed_Length(const char* data, long* length);

/* inc/sqeazy.h:274-299, src/sqeazy.cpp:281-335.  Decode a blob produced by SQY_PipelineEncode_* (either
 * LZ4 layout).  dst must hold SQY_Decompressed_Length bytes. */
SQY_FUNCTION_PREFIX int SQY_Decode_UI16(const char* src, long srclength, char* dst, int nthreads);
SQY_FUNCTION_PREFIX int SQY_Decode_UI8(const char* src, long srclength, char* dst, int nthreads);

/* ------------------------------------------------------------------------------------------------
 * Section B -- HBM-resident variants (no reference counterpart; same semantics as above)
 * ---------------------------------------------------------------------------------------------- */

/* d_src / d_dst are device pointers on the current HIP device; dst_capacity is checked (1 when the
 * blob does not fit).  hip_stream is a hipStream_t (NULL = default stream).  The call returns after the
 * blob is complete in d_dst.  All work is queued on hip_stream: whatever made d_src has to be in front of it there (or complete). */
SQY_FUNCTION_PREFIX int SQYAMD_PipelineEncode_UI16_Device(const char* pipeline, const void* d_src, const long* shape,
                                                          unsigned shape_size, void* d_dst, long dst_capacity,
                                                          long* dstlength, int nthreads, void* hip_stream);
SQY_FUNCTION_PREFIX int SQYAMD_PipelineEncode_UI8_Device(const char* pipeline, const void* d_src, const long* shape,
                                                         unsigned shape_size, void* d_dst, long dst_capacity,
                                                         long* dstlength, int nthreads, void* hip_stream);
/* The same, but the blob may start anywhere inside [d_dst, d_dst + dst_capacity): *dstoffset says where, *dstlength how long it is
 * (bytes identical to the entry points above).  This is the fast path: for `...->bitswap1->lz4` on 16-bit voxels the bit-plane
 * transpose writes the plane stream straight into d_dst as the bodies of the LZ4 frames it will become (one frame per 256 KiB
 * chunk, encoders/lz4_utils.hpp:193-274); the stored frames that end the payload -- the noise planes, 98 % of the payload of a
 * microscopy stack -- then never move, only the compressed frames in front of them are gathered (no second pass over the
 * payload).  dst_capacity as above; every other pipeline returns *dstoffset = 0. */
SQY_FUNCTION_PREFIX int SQYAMD_PipelineEncode_UI16_DeviceAt(const char* pipeline, const void* d_src, const long* shape,
                                                            unsigned shape_size, void* d_dst, long dst_capacity, long* dstoffset,
                                                            long* dstlength, int nthreads, void* hip_stream);
SQY_FUNCTION_PREFIX int SQYAMD_PipelineEncode_UI8_DeviceAt(const char* pipeline, const void* d_src, const long* shape,
                                                           unsigned shape_size, void* d_dst, long dst_capacity, long* dstoffset,
                                                           long* dstlength, int nthreads, void* hip_stream);
/* As _DeviceAt; additionally tells where every `every`-th LZ4 frame of the payload starts (pipelines that end in lz4 in the
 * chunked layout, one frame per chunk: encoders/lz4_utils.hpp:193-274): frame_offsets[i] = start of frame i * every relative to
 * the blob start, i = 0 .. *count - 1, and frame_offsets[*count] = the blob length (max_entries >= *count + 1, else 1).  With
 * every = chunks per bit plane this is the byte range of every bit plane of a `bitswap1->lz4` blob -- what the single-blob mode of
 * the multi-GPU path re-orders (the frame sizes come from the encoder's own table in HBM, nobody walks the frames). */
SQY_FUNCTION_PREFIX int SQYAMD_PipelineEncode_UI16_DeviceAt_Frames(const char* pipeline, const void* d_src, const long* shape,
                                                                   unsigned shape_size, void* d_dst, long dst_capacity,
                                                                   long* dstoffset, long* dstlength, int nthreads, void* hip_stream,
                                                                   int every, long* frame_offsets, int max_entries, int* count);
SQY_FUNCTION_PREFIX int SQYAMD_PipelineEncode_UI8_DeviceAt_Frames(const char* pipeline, const void* d_src, const long* shape,
                                                                  unsigned shape_size, void* d_dst, long dst_capacity,
                                                                  long* dstoffset, long* dstlength, int nthreads, void* hip_stream,
                                                                  int every, long* frame_offsets, int max_entries, int* count);
/* A whole volume as `nslabs` independent z-slab blobs with ONE call (the reference encodes one volume of < 2^31 voxels per call,
 * src/sqeazy.cpp:108-142; larger volumes are cut into z-slabs by its callers).  shape is the WHOLE volume {z,y,x}; slab i holds
 * frames [i*(Z/n) + min(i, Z%n), ...) -- the first Z % nslabs slabs get one frame more -- and is encoded exactly as
 * SQYAMD_PipelineEncode_*_DeviceAt would encode it (every blob is a complete sqeazy blob, bytes identical to the single calls).
 * Blob i lies inside d_dst[i*slab_capacity, (i+1)*slab_capacity) (slab_capacity >= SQY_Pipeline_Max_Compressed_Length_3D_* of the
 * largest slab): offsets[i] = its start relative to d_dst, lengths[i] = its bytes.  `inflight` slab calls (<= 0: three) run at a
 * time on library-owned streams: the transposes, LZ4 parses and gathers of different slabs overlap.  Returns when all are done.
 * Ordering: the slab calls start behind everything queued on the DEFAULT stream when the call is made (the kernels that made d_src, a
 * fill of d_dst); work on other streams that touches d_src or d_dst has to be complete (synchronised) before the call. */
SQY_FUNCTION_PREFIX int SQYAMD_PipelineEncode_Slabs_UI16_Device(const char* pipeline, const void* d_src, const long* shape,
                                                                unsigned shape_size, int nslabs, void* d_dst, long slab_capacity,
                                                                long* offsets, long* lengths, int nthreads, int inflight);
SQY_FUNCTION_PREFIX int SQYAMD_PipelineEncode_Slabs_UI8_Device(const char* pipeline, const void* d_src, const long* shape,
                                                               unsigned shape_size, int nslabs, void* d_dst, long slab_capacity,
                                                               long* offsets, long* lengths, int nthreads, int inflight);
/* host-pointer encode with an explicit destination capacity (returns 1 instead of overflowing dst; the
 * reference-protocol entry points above assume dst holds exactly SQY_Pipeline_Max_Compressed_Length_* bytes) */
SQY_FUNCTION_PREFIX int SQYAMD_PipelineEncode_UI16_Cap(const char* pipeline, const char* src, long* shape, unsigned shape_size,
                                                       char* dst, long dst_capacity, long* dstlength, int nthreads);
SQY_FUNCTION_PREFIX int SQYAMD_PipelineEncode_UI8_Cap(const char* pipeline, const char* src, long* shape, unsigned shape_size,
                                                      char* dst, long dst_capacity, long* dstlength, int nthreads);

SQY_FUNCTION_PREFIX int SQYAMD_Decode_UI16_Device(const void* d_src, long srclength, void* d_dst, long dst_capacity, void* hip_stream);
SQY_FUNCTION_PREFIX int SQYAMD_Decode_UI8_Device(const void* d_src, long srclength, void* d_dst, long dst_capacity, void* hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Section C -- several GPUs (no reference counterpart: sqeazy is a single process with OpenMP loops)
 *
 * z-slabs of a volume are independent sqeazy blobs (one encode call per slab, one process per GPU); the only exchange step
 * of the path is the final gather of the compressed slabs to one rank over RCCL / xGMI.  RCCL is loaded at first use.
 * ---------------------------------------------------------------------------------------------- */

/* A communicator over `world` ranks, one per GPU (wraps ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy, so that a C caller
 * needs no RCCL header): rank 0 makes the 128-byte id, hands it to the other ranks by whatever means it has (MPI, a file, a
 * socket), every rank then calls Comm_Init with its HIP device current. */
SQY_FUNCTION_PREFIX int SQYAMD_Comm_UniqueId(char* id128);
SQY_FUNCTION_PREFIX int SQYAMD_Comm_Init(void** comm, int world, int rank, const char* id128);
SQY_FUNCTION_PREFIX int SQYAMD_Comm_Destroy(void* comm);
/* Variable-length gather to `root`: every rank passes its blob (device pointer, nbytes); sizes[0..world) (host, out on EVERY rank)
 * are the blob sizes in rank order -- the index of a sharded container --; the root receives the blobs back to back, in rank order,
 * at d_recv (blob r at the sum of the sizes in front of it).  One 8-byte all-gather + grouped ncclSend / ncclRecv (point to point
 * over xGMI) on hip_stream; returns when the bytes have arrived.  A root buffer that is too small makes EVERY rank return 1 (the
 * ranks agree before anybody sends).  d_recv / recv_capacity are only read on the root. */
SQY_FUNCTION_PREFIX int SQYAMD_Gather_Blobs(void* comm, int root, const void* d_blob, long nbytes, void* d_recv, long recv_capacity,
                                            long* sizes, void* hip_stream);

/* per-kernel device timing (hipEvents on the call's stream), for bench.py's roofline line.
 *   enable != 0 starts collecting, Reset clears.  Get: i-th kernel name seen since the last reset
 *   (NULL when i is past the end), total milliseconds and number of launches. */
SQY_FUNCTION_PREFIX void SQYAMD_Profile_Enable(int enable);
SQY_FUNCTION_PREFIX void SQYAMD_Profile_Reset(void);
SQY_FUNCTION_PREFIX const char* SQYAMD_Profile_Get(int i, double* total_ms, long* launches);

/* release the cached HBM workspace of the current device */
SQY_FUNCTION_PREFIX void SQYAMD_Release_Workspace(void);

/* Run-time options (measurement / test switches; none changes a byte of any result).  The environment is read once, when the
 * library is loaded (the names in brackets); afterwards only these two calls change / read them -- thread safe.
 *   "transpose_chain"                 1 [SQY_NO_TRANSPOSE_CHAIN=1 -> 0]  the bit-plane transposes of calls in flight on streams the
 *                                     LIBRARY owns (host-pointer entry points, the Slabs workers) run one after the other
 *   "transpose_chain_caller_streams"  0 [SQY_TRANSPOSE_CHAIN_CALLER_STREAMS=1]  .. on streams the CALLER brings as well.  This puts a
 *                                     hipStreamWaitEvent between two caller streams: only for callers whose streams carry nothing but
 *                                     these calls (a backlog or a host function on one stream would hold the other up)
 *   "block_parallel"                  1 [SQY_NO_BLOCK_PARALLEL=1 -> 0]  block-linked frames (nthreads = 1) encoded / decoded block-parallel
 *   "block_parallel_warmup"           65536 [SQY_BLOCK_PARALLEL_WARMUP=<bytes>, 0 .. 2^30]  stream parsed in front of a block to guess its table
 *   "block_parallel_stats"            0 [SQY_BLOCK_PARALLEL_STATS=1]  print the blocks whose guess failed
 *   "tail_scan"                       1 [SQY_NO_TAIL_SCAN=1 -> 0]  serial-layout decode: the walk over the block tails as a scan
 *   "decode_two_waves"                1 [SQY_NO_DECODE_TWO_WAVES=1 -> 0]  chunked-layout decode: two wavefronts per frame (0: one)
 *   "noise_digest"                    1 [SQY_NO_NOISE_DIGEST=1 -> 0]  frames in place: the bit-plane transpose leaves bucket and tag of every position
 *                                     liblz4's search probes in a chunk of noise; the LZ4 parse proves such chunks incompressible from them
 *                                     instead of reading the plane stream again (0: it reads)
 *   "transpose_blocks_per_cu"         32 [SQY_TRANSPOSE_BLOCKS_PER_CU=<1..64>]  frames in place: workgroups (two wavefronts) of the transposer's grid
 *                                     per CU (32: as many as fit; fewer leave room for the small kernels of other calls in flight -- measured: slower)
 *   "stored_tail_index"               1 [SQY_NO_STORED_TAIL_INDEX=1 -> 0]  decode of the chunked layout: the stored frames at the end of the LZ4
 *                                     stream (bit planes of noise) are looked for where they must start, the scan for frame headers stops in
 *                                     front of them (0: it reads the whole stream)
 * Set: 0 = done, 1 = unknown name or value out of range.  Get: the value, -1 for an unknown name. */
SQY_FUNCTION_PREFIX int SQYAMD_Set_Option(const char* name, long value);
SQY_FUNCTION_PREFIX long SQYAMD_Get_Option(const char* name);

/* Header helpers for callers that store blobs in containers of their own (the HDF5 filter's cd_values carry a header:
 * inc/sqeazy_h5_filter.hpp:117-121, src/hdf5_utils.hpp:730-738).  The reference does this through its C++ header class
 * (src/sqeazy_header.hpp); there is no C symbol for it there.
 *   Header_Pipeline: copies the NUL-terminated "pipename" of the header at src into out.  *outlength in: capacity of out,
 *                    out: bytes needed (out == NULL: size query).
 *   Header_Build:    writes the header (with delimiter, without payload) the encoder would put in front of
 *                    `encoded_bytes` of payload for this pipeline / voxel size / shape.  Same length protocol. */
SQY_FUNCTION_PREFIX int SQYAMD_Header_Pipeline(const char* src, long srclength, char* out, long* outlength);
SQY_FUNCTION_PREFIX int SQYAMD_Header_Build(const char* pipeline, int sizeof_voxel, const long* shape, unsigned shape_size,
                                            long encoded_bytes, char* out, long* outlength);

/* "sqeazy_amd <version> (gfx950)" */
SQY_FUNCTION_PREFIX const char* SQYAMD_Version(void);

#endif /* SQEAZY_AMD_H_ */
