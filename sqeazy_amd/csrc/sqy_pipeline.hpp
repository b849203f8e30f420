// sqy_pipeline.hpp -- host-side mirror of the reference's pipeline object for the hot path:
// pipeline grammar, per-stage configuration strings, size bounds and the sqy header.
//
// Mirrors (paths relative to /root/reference/src/cpp/src):
//   string_parsers.hpp:285-471        pipeline_parser::to_pairs / minors (with <verbatim> protection)
//   dynamic_pipeline.hpp:137-226      from_string / can_be_built_from
//   dynamic_pipeline.hpp:476-503      name()
//   dynamic_pipeline.hpp:866-890      max_encoded_size
//   sqeazy_header.hpp:146-193,294-344 header pack / unpack
//   encoders/lz4.hpp:58-188           lz4 parameter logic, config string, chunking, size bound
//   sqeazy_algorithms.hpp:14-22       thread-count clamp
#ifndef SQY_PIPELINE_HPP_
#define SQY_PIPELINE_HPP_

#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace sqy {

typedef std::vector<std::pair<std::string, std::string>> pairs_t;

// split on `sep` outside <verbatim>...</verbatim>; returns {} for a malformed (unbalanced) string
std::vector<std::string> split_outside_verbatim(const std::string& s, const std::string& sep, bool* ok = nullptr);
pairs_t parse_pairs(const std::string& pipeline);
std::map<std::string, std::string> parse_minors(const std::string& cfg);

enum class StageKind { diff3x3x1, bitswap1, bitshuffle, frame_shuffle, raster_reorder, zcurve_reorder, tile_shuffle, quantiser, lz4, pass_through, unsupported };

struct Lz4Params {
    int accel = 1;
    uint32_t blocksize_kb = 256, framestep_kb = 256, n_chunks = 0;
    int block_id = 5;                       // LZ4F blockSizeID 4..7
    explicit Lz4Params(const std::string& cfg = "");
    std::string config() const;
    uint64_t block_bytes() const;
    uint64_t bytes_per_chunk(uint64_t nbytes) const;                       // lz4.hpp:146-156
    uint64_t max_encoded_size(uint64_t nbytes, unsigned nthreads) const;   // lz4.hpp:166-188
    static uint64_t compress_bound(uint64_t src, int block_id);            // LZ4F_compressBound, autoFlush = 0
};

// ---- block-linked LZ4 frames (lz4::encode_serial, encoders/lz4_utils.hpp:99-173) ----
// One entry per LZ4 block in stream order; same layout as sqy::Lz4Block (sqy_kernels.h), which the kernels read.
struct Lz4BlockPlan {
    uint64_t start;          // byte offset of the block in the stream
    uint32_t n;              // bytes
    uint32_t flags;          // bit 0: opens a frame (fresh LZ4 stream), bit 1: closes it
    int64_t low_in, low_dict;   // liblz4's lowLimit for matches that start inside the block / in the history in front of it
};
struct Lz4Plan {
    std::vector<Lz4BlockPlan> blocks;
    std::vector<uint32_t> frame_first;      // frame f = blocks [frame_first[f], frame_first[f+1])
    uint32_t max_block = 0;
    bool ok = true;                         // false: a case liblz4 would run in a mode the kernels do not model
};
// What liblz4 1.9.3's frame layer does with `total` bytes under sqeazy's preferences (block-linked, autoFlush 0, stableSrc 0):
// serial = true : ONE frame, LZ4F_compressUpdate every `step` bytes (encode_serial, nthreads == 1)
// serial = false: one frame per `step` bytes, each fed by a single update (encode_parallel -> encode_serial per chunk)
Lz4Plan lz4_plan_blocks(uint64_t total, uint64_t step, uint64_t block_bytes, bool serial);

struct Stage {
    std::string name;
    StageKind kind = StageKind::unsupported;
    std::map<std::string, std::string> cfg;   // parsed (k=v,...) payload; std::map => sorted like the reference's config_map
    Lz4Params lz4;
    std::string config() const;               // re-serialised configuration, as the reference's config()
    std::string full_name() const;            // name or name(config)
};

// which factory lists know a name (sqeazy_pipelines.hpp:31-77; optional ffmpeg/bitshuffle stages are not built)
bool known_head_filter(const std::string& n);
bool known_sink(const std::string& n);
bool known_tail_filter(const std::string& n);

struct Pipeline {
    std::vector<Stage> stages;
    int sink_index = -1;                      // index of the sink stage, -1 when the pipeline only filters
    unsigned nthreads = 1;                    // stage default (dynamic_stage.hpp:21-24)

    // the reference's validity rule over its full stage lists
    static bool reference_accepts(const std::string& s);
    // true when every stage is one this library implements (subset of the above)
    static bool supported(const std::string& s, int elem_size, std::string* why = nullptr);
    // elem_size > 0 fills the defaults that depend on the voxel type (raster_reorder: tile_size = 16 / sizeof(T))
    static Pipeline from_string(const std::string& s, int elem_size = 0);

    std::string name() const;
    uint64_t max_encoded_size(uint64_t nbytes, int elem_size) const;
    void set_n_threads(int n);
};

int clean_number_of_threads(int n);

// raster_reorder (encoders/raster_reorder_utils.hpp:36-367): false for the geometries whose result the reference leaves
// undefined (a remainder in some but not all dimensions; tile_size a proper multiple of the 16-byte SSE block on a
// remainder-free shape)
bool raster_geometry_defined(uint64_t Z, uint64_t Y, uint64_t X, uint64_t tile_size, int elem_size);

// zcurve_reorder (encoders/zcurve_reorder_utils.hpp): tile sizes 2..128 (powers of two) -- inside a tile the reference's
// morton_at_ct<log2(tile)> code is plain row-major, so the stage is "tiles of tile^3 (smaller at the high ends) in (z,y,x) tile
// order"; false where the reference runs past its buffers (other tile sizes; encode_full with a tile that does not divide the shape)
bool zcurve_geometry_defined(uint64_t Z, uint64_t Y, uint64_t X, uint64_t tile_size);
// tile_shuffle (encoders/tile_shuffle_utils.hpp:104-224, encode_full): only shapes that are whole multiples of the tile
bool tile_shuffle_geometry_defined(uint64_t Z, uint64_t Y, uint64_t X, uint64_t tile_size);
// metric = (T)(sequential float sum / voxels per tile), sorted ascending, slot i <- first tile whose metric equals sorted[i]
void tile_shuffle_order(const float* sums, size_t ntiles, size_t per_tile, int elem_size, uint64_t* decode_map, bool signed_char = false);
// bitshuffle: elements per block (bshuf_default_block_size for 0); 0 when the configured size is not a multiple of 8
uint64_t bitshuffle_block_elems(int elem_size, uint64_t block_size);

// ---- header ----
std::string header_pack(int elem_size, bool is_signed8, const std::vector<uint64_t>& shape, const std::string& pipename,
                        uint64_t payload_bytes);
void header_pack_parts(int elem_size, bool is_signed8, const std::vector<uint64_t>& shape, const std::string& pipename,
                       std::string* prefix, std::string* suffix);
struct HeaderInfo {
    bool valid = false;
    std::string pipename, type;
    std::vector<uint64_t> shape;
    uint64_t payload_bytes = 0;
    uint64_t size = 0;       // header bytes including the delimiter
    int elem_size() const;
};
HeaderInfo header_unpack(const char* begin, const char* end);

// quantiser `weighting_function` as quantiser_scheme::encode reads it (encoders/quantiser_scheme_impl.hpp:186-198, extract_ratio
// :25-47): "none" anywhere in the string -> weights 1; otherwise every run of digits is an integer (one: n/1, two: n/d, else
// 0/0) and "offset" anywhere selects offset_power_of (encoders/quantiser_weighters.hpp:20-95) instead of power_of (:98-160).
struct QuantiserWeighting {
    int mode = 0;            // 0 none, 1 power_of, 2 offset_power_of
    int num = 1, den = 1;    // exponent = float(num) / den
};
// false: the reference's exponent would be NaN or infinite (no "_" in the string, not one or two integers, denominator 0)
bool quantiser_parse_weighting(const std::string& text, QuantiserWeighting* out);
// quantiser LUTs from a 65536-bin histogram (encoders/quantiser_utils.hpp:386-418, :227-306, weights :317-322):
// lut_encode[65536] bytes, lut_decode[256] raw values.  IEEE binary32/64 in the reference's statement order.
void quantiser_build_luts(const uint32_t* histo, size_t nbins, unsigned char* lut_encode, uint16_t* lut_decode,
                          const QuantiserWeighting& weighting = QuantiserWeighting());
// quantiser::lut_to_file / lut_from_file (encoders/quantiser_utils.hpp:490-515): one decimal value per line
bool quantiser_lut_to_file(const std::string& path, const uint16_t* lut, size_t n);
bool quantiser_lut_from_file(const std::string& path, uint16_t* lut, size_t n);

// frame_shuffle ordering from the per-frame float sums (encoders/frame_shuffle_utils.hpp:126-166):
// metric = sum / per_frame, sorted ascending, slot i <- first frame whose metric equals sorted[i]
void frame_shuffle_order(const float* sums, size_t Z, size_t per_frame, uint64_t* decode_map);

std::string base64_encode(const unsigned char* src, size_t n);
std::vector<unsigned char> base64_decode(const std::string& s);
std::string to_verbatim(const void* data, size_t bytes);

uint32_t xxh32(const unsigned char* p, size_t len, uint32_t seed);

extern const char* const kVersion;
extern const char* const kHeadRef;
extern const int kVersionTriple[3];

} // namespace sqy
#endif
