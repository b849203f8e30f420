// sqy_pipeline.cpp -- see sqy_pipeline.hpp.  Pure host logic, no HIP.
#include "sqy_pipeline.hpp"

// IEEE evaluation in statement order for the quantiser / frame metric host code: no fused multiply-adds (hipcc defaults to
// -ffp-contract=fast for HIP sources; the declared parity target is the reference compiled without fast-math)
#pragma clang fp contract(off)

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <sys/stat.h>
#include <thread>

namespace sqy {

const char* const kVersion = "0.5.2";    // sqy.version: a build-time constant in the reference (sqeazy_header.hpp:172)
const char* const kHeadRef = "mi355x";   // sqy.headref: `git describe --always` of the reference build (:173)
const int kVersionTriple[3] = {0, 5, 2};

static const std::string kVerbOpen = "<verbatim>";
static const std::string kVerbClose = "</verbatim>";
static const std::string kHeaderEnd = "|01307#!";   // sqeazy_header.hpp:586

std::vector<std::string> split_outside_verbatim(const std::string& s, const std::string& sep, bool* ok)
{
    std::vector<std::string> out;
    if (ok) *ok = true;
    std::string cur;
    size_t i = 0;
    while (i < s.size()) {
        if (s.compare(i, kVerbOpen.size(), kVerbOpen) == 0) {
            const size_t j = s.find(kVerbClose, i);
            if (j == std::string::npos) {            // "found only 1 verbatim delimeter" (string_parsers.hpp:150-153)
                if (ok) *ok = false;
                return {};
            }
            cur.append(s, i, j + kVerbClose.size() - i);
            i = j + kVerbClose.size();
            continue;
        }
        if (s.compare(i, sep.size(), sep) == 0) {
            out.push_back(cur);
            cur.clear();
            i += sep.size();
            continue;
        }
        cur.push_back(s[i]);
        ++i;
    }
    out.push_back(cur);
    return out;
}

pairs_t parse_pairs(const std::string& pipeline)
{
    pairs_t value;
    if (pipeline.empty()) return value;
    for (const std::string& major : split_outside_verbatim(pipeline, "->")) {
        const size_t d = major.find('(');
        if (d == std::string::npos)
            value.emplace_back(major, "");
        else
            value.emplace_back(major.substr(0, d), major.size() > d + 1 ? major.substr(d + 1, major.size() - d - 2) : "");
    }
    return value;
}

std::map<std::string, std::string> parse_minors(const std::string& cfg)
{
    std::map<std::string, std::string> value;
    if (cfg.empty()) return value;
    for (const std::string& item : split_outside_verbatim(cfg, ",")) {
        const size_t d = item.find('=');
        if (d == std::string::npos) {
            value[item] = item;
        } else {
            value[item.substr(0, d)] = (d + 1 < item.size()) ? item.substr(d + 1) : item;
        }
    }
    return value;
}

// ---- lz4 parameters (encoders/lz4.hpp:58-114) ----
static uint32_t closest_blocksize_kb(uint32_t kb)
{
    static const uint32_t sizes[4] = {64, 256, 1024, 4096};   // lz4_utils.hpp:60-93
    const uint32_t* it = std::lower_bound(sizes, sizes + 4, kb);
    if (it == sizes + 4) return sizes[3];
    if (it == sizes) return sizes[0];
    const uint32_t lo = *(it - 1), hi = *it;
    const uint32_t middle = lo + (hi - lo) / 2;
    return kb >= middle ? hi : lo;
}

static float stof_or(const std::string& s, float dflt)
{
    char* end = nullptr;
    const float v = std::strtof(s.c_str(), &end);
    return (end == s.c_str()) ? dflt : v;
}

Lz4Params::Lz4Params(const std::string& cfg)
{
    const auto m = parse_minors(cfg);
    auto f = m.find("accel");
    if (f != m.end()) accel = (int)stof_or(f->second, 1.f);
    f = m.find("blocksize_kb");
    if (f != m.end()) blocksize_kb = (uint32_t)stof_or(f->second, 256.f);
    f = m.find("framestep_kb");
    if (f != m.end()) framestep_kb = (uint32_t)stof_or(f->second, 256.f);
    f = m.find("n_chunks_of_input");
    if (f != m.end()) n_chunks = (uint32_t)stof_or(f->second, 0.f);
    if (blocksize_kb == 0) blocksize_kb = 256;
    if (framestep_kb < blocksize_kb)
        framestep_kb = blocksize_kb;
    else
        framestep_kb = (uint32_t)(std::round(framestep_kb / float(blocksize_kb)) * blocksize_kb);
    if (n_chunks != 0) framestep_kb = 0;
    const uint32_t c = closest_blocksize_kb(blocksize_kb);
    block_id = c == 64 ? 4 : c == 256 ? 5 : c == 1024 ? 6 : 7;
}

std::string Lz4Params::config() const
{
    std::ostringstream msg;
    msg << "accel=" << accel << ",blocksize_kb=" << blocksize_kb << ",framestep_kb=" << framestep_kb
        << ",n_chunks_of_input=" << n_chunks;
    return msg.str();
}

uint64_t Lz4Params::block_bytes() const
{
    static const uint64_t b[8] = {0, 0, 0, 0, 64u << 10, 256u << 10, 1u << 20, 4u << 20};
    return b[block_id];
}

uint64_t Lz4Params::bytes_per_chunk(uint64_t nbytes) const
{
    uint64_t value = framestep_kb ? (uint64_t)framestep_kb << 10 : (n_chunks ? nbytes / n_chunks : nbytes);
    if (value >= nbytes || n_chunks >= nbytes) value = nbytes;
    return value;
}

uint64_t Lz4Params::compress_bound(uint64_t srcSize, int block_id)
{
    static const uint64_t b[8] = {0, 0, 0, 0, 64u << 10, 256u << 10, 1u << 20, 4u << 20};
    const uint64_t blockSize = b[block_id];
    const uint64_t maxSrcSize = srcSize + (blockSize - 1);        // autoFlush = 0: a full tmp buffer is assumed
    const uint64_t nbFullBlocks = maxSrcSize / blockSize;
    const uint64_t partial = maxSrcSize & (blockSize - 1);
    const uint64_t last = (srcSize == 0) ? partial : 0;
    const uint64_t nbBlocks = nbFullBlocks + (last > 0);
    return 4 * nbBlocks + blockSize * nbFullBlocks + last + 4;
}

uint64_t Lz4Params::max_encoded_size(uint64_t nbytes, unsigned nthreads) const
{
    const uint64_t per_chunk = bytes_per_chunk(nbytes);
    const uint64_t hdr_max = 19;                                   // LZ4F_HEADER_SIZE_MAX (liblz4 1.9.3)
    if (per_chunk >= nbytes) return hdr_max + compress_bound(per_chunk, block_id);
    if (nthreads == 0) nthreads = 1;
    const uint64_t nchunks = (nbytes + per_chunk - 1) / per_chunk;
    const uint64_t per_thread = (nchunks + nthreads - 1) / nthreads;
    return per_thread * (compress_bound(per_chunk, block_id) + hdr_max) * nthreads;
}

// ---- block-linked frames: liblz4 1.9.3's LZ4F buffer management, as far as it decides the bytes ----
// LZ4F_compressUpdate compresses whole blocks straight from the caller's buffer and parks a remainder in tmpBuff; after an
// update that compressed from the caller's buffer it copies the last 64 KiB of history into tmpBuff (LZ4F_localSaveDict,
// stableSrc = 0).  LZ4_compress_fast_continue then runs a block in PREFIX mode when it follows its dictionary in memory
// (lowLimit = start of the dictionary) and in EXTERNAL-DICTIONARY mode otherwise (lowLimit = block start for matches
// inside the block, dictionary start for matches in the history).  Both see the same logical history; lowLimit bounds the
// backward catch-up of a match and is what this model hands to the kernel per block.
namespace {
struct Lz4fModel {
    enum Space { kNone, kCaller, kTmp };
    // LZ4_stream_t: where the dictionary sits and how long it is, plus the index counter that triggers LZ4_renormDictT
    Space dict_space = kNone;
    uint64_t dict_addr = 0;
    uint32_t dict_size = 0;
    uint32_t current_offset = 0;
    // LZ4F_cctx
    uint64_t block_size, max_buffer;
    uint64_t tmp_in = 0, tmp_fill = 0, tmp_stream_pos = 0;
    Lz4Plan* plan;
    bool* ok;

    void compress_block(uint64_t stream_pos, uint64_t n, Space where, uint64_t addr)
    {
        if ((uint64_t)current_offset + n > 0x80000000ull) {          // LZ4_renormDictT: table rescaled (transparent), dictionary clipped
            current_offset = 64u << 10;
            if (dict_size > (64u << 10)) { dict_addr += dict_size - (64u << 10); dict_size = 64u << 10; }
        }
        bool follows = dict_space == where && dict_addr + dict_size == addr;
        if (dict_size >= 1 && dict_size <= 3 && !follows) { dict_size = 0; dict_space = where; dict_addr = addr; follows = true; }
        if (dict_size < (64u << 10) && dict_size < current_offset) *ok = false;      // liblz4's dictSmall variant: never reached with blocks >= 64 KiB
        Lz4BlockPlan b;
        b.start = stream_pos; b.n = (uint32_t)n; b.flags = 0;
        b.low_dict = (int64_t)stream_pos - (int64_t)dict_size;
        b.low_in = follows ? b.low_dict : (int64_t)stream_pos;
        plan->blocks.push_back(b);
        if (n > plan->max_block) plan->max_block = (uint32_t)n;
        current_offset += (uint32_t)n;
        if (follows) dict_size += (uint32_t)n;
        else { dict_space = where; dict_addr = addr; dict_size = (uint32_t)n; }
    }
    uint64_t save_dict()                                             // LZ4_saveDict(stream, tmpBuff, 64 KB)
    {
        const uint32_t d = dict_size < (64u << 10) ? dict_size : (64u << 10);
        dict_space = kTmp; dict_addr = 0; dict_size = d;
        return d;
    }
    void update(uint64_t pos, uint64_t size)                         // LZ4F_compressUpdate(ctx, .., src + pos, size, NULL)
    {
        uint64_t p = pos;
        const uint64_t end = pos + size;
        bool from_caller = false;
        if (tmp_fill > 0) {
            const uint64_t to_copy = block_size - tmp_fill;
            if (to_copy > size) { tmp_fill += size; p = end; }
            else {
                p += to_copy;
                compress_block(tmp_stream_pos, block_size, kTmp, tmp_in);
                tmp_in += block_size;
                tmp_fill = 0;
            }
        }
        while (end - p >= block_size) {
            from_caller = true;
            compress_block(p, block_size, kCaller, p);
            p += block_size;
        }
        if (from_caller) tmp_in = save_dict();
        if (tmp_in + block_size > max_buffer) tmp_in = save_dict();
        if (p < end) { tmp_stream_pos = p; tmp_fill = end - p; }
    }
    void finish() { if (tmp_fill > 0) compress_block(tmp_stream_pos, tmp_fill, kTmp, tmp_in); tmp_fill = 0; }   // LZ4F_compressEnd -> LZ4F_flush
};
} // namespace

Lz4Plan lz4_plan_blocks(uint64_t total, uint64_t step, uint64_t block_bytes, bool serial)
{
    Lz4Plan plan;
    if (total == 0 || step == 0 || block_bytes == 0) { plan.frame_first.push_back(0); return plan; }
    const uint64_t nframes = serial ? 1 : (total + step - 1) / step;
    for (uint64_t f = 0; f < nframes; ++f) {
        const uint64_t f_begin = serial ? 0 : f * step;
        const uint64_t f_end = serial ? total : std::min(total, f_begin + step);
        plan.frame_first.push_back((uint32_t)plan.blocks.size());
        Lz4fModel m;
        m.block_size = block_bytes; m.max_buffer = block_bytes + (128u << 10); m.plan = &plan; m.ok = &plan.ok;
        // positions handed to the model are relative to the frame's first byte (its own LZ4 stream); `start` is made absolute below
        const size_t first = plan.blocks.size();
        for (uint64_t pos = f_begin; pos < f_end; pos += step) m.update(pos - f_begin, std::min(step, f_end - pos));
        m.finish();
        for (size_t i = first; i < plan.blocks.size(); ++i) {
            plan.blocks[i].start += f_begin; plan.blocks[i].low_in += (int64_t)f_begin; plan.blocks[i].low_dict += (int64_t)f_begin;
        }
        if (plan.blocks.size() > first) { plan.blocks[first].flags |= 1u; plan.blocks.back().flags |= 2u; }
    }
    plan.frame_first.push_back((uint32_t)plan.blocks.size());
    return plan;
}

// ---- stages ----
static StageKind kind_of(const std::string& n)
{
    if (n == "diff3x3x1") return StageKind::diff3x3x1;
    if (n == "bitswap1") return StageKind::bitswap1;
    if (n == "frame_shuffle") return StageKind::frame_shuffle;
    if (n == "raster_reorder") return StageKind::raster_reorder;
    if (n == "zcurve_reorder") return StageKind::zcurve_reorder;
    if (n == "tile_shuffle") return StageKind::tile_shuffle;
    if (n == "bitshuffle") return StageKind::bitshuffle;
    if (n == "quantiser") return StageKind::quantiser;
    if (n == "lz4") return StageKind::lz4;
    if (n == "pass_through") return StageKind::pass_through;
    return StageKind::unsupported;
}

std::string Stage::config() const
{
    switch (kind) {
        case StageKind::bitswap1: return "num_bits_per_plane=1";                 // bitswap_scheme_impl.hpp:84-90
        case StageKind::diff3x3x1: return "";                                    // diff_scheme_impl.hpp:55-59
        case StageKind::lz4: return lz4.config();                                // lz4.hpp:132-141
        case StageKind::pass_through: return "";
        case StageKind::quantiser: {                                             // quantiser_scheme_impl.hpp:102-118
            std::string s;
            size_t count = 0;
            for (const auto& kv : cfg) {
                s += kv.first + "=" + kv.second;
                if (count++ < cfg.size() - 1) s += ",";
            }
            return s;
        }
        case StageKind::raster_reorder: {                                        // raster_reorder_scheme_impl.hpp:53-59
            auto t = cfg.find("tile_size");
            return "tile_size=" + std::to_string(t != cfg.end() ? std::atoi(t->second.c_str()) : 0);
        }
        case StageKind::zcurve_reorder: {                                        // zcurve_reorder_scheme_impl.hpp:68-74 (default tile 2)
            auto t = cfg.find("tile_size");
            return "tile_size=" + std::to_string(t != cfg.end() ? std::atoi(t->second.c_str()) : 2);
        }
        case StageKind::tile_shuffle: {                                          // tile_shuffle_scheme_impl.hpp:57-66 (default tile 32)
            auto t = cfg.find("tile_size");
            auto m = cfg.find("reorder_map");
            return "tile_size=" + std::to_string(t != cfg.end() ? std::atoi(t->second.c_str()) : 32) + ",reorder_map=" + (m != cfg.end() ? m->second : "");
        }
        case StageKind::bitshuffle: {                                            // bitshuffle_scheme_impl.hpp:75-81 (default block 0 = library default)
            auto b = cfg.find("block_size");
            return "block_size=" + std::to_string(b != cfg.end() ? std::atoi(b->second.c_str()) : 0);
        }
        case StageKind::frame_shuffle: {                                         // frame_shuffle_scheme_impl.hpp:58-66
            auto c = cfg.find("frame_chunk_size");
            auto m = cfg.find("reorder_map");
            const int chunk = c != cfg.end() ? std::atoi(c->second.c_str()) : 1;
            return "frame_chunk_size=" + std::to_string(chunk) + ",reorder_map=" + (m != cfg.end() ? m->second : "");
        }
        default: return "";
    }
}

std::string Stage::full_name() const
{
    const std::string c = config();
    return c.empty() ? name : name + "(" + c + ")";
}

static bool in_list(const std::string& n, const char* const* list, size_t cnt)
{
    for (size_t i = 0; i < cnt; ++i) if (n == list[i]) return true;
    return false;
}

bool known_head_filter(const std::string& n)
{
    // (bitshuffle: present when the reference is built with USE_BITSHUFFLE, its default: CMakeLists.txt:47, sqeazy_pipelines.hpp:35-37)
    static const char* const l[] = {"diff3x3x1", "bitswap1", "bitshuffle", "remove_background", "rmbkrd_neighbor5x5x5", "rmestbkrd",
                                    "raster_reorder", "tile_shuffle", "frame_shuffle", "zcurve_reorder"};
    return in_list(n, l, sizeof(l) / sizeof(l[0]));
}
bool known_sink(const std::string& n)
{
    static const char* const l[] = {"pass_through", "quantiser", "lz4"};
    return in_list(n, l, sizeof(l) / sizeof(l[0]));
}
bool known_tail_filter(const std::string& n)
{
    static const char* const l[] = {"diff3x3x1", "bitswap1", "bitshuffle", "lz4", "raster_reorder", "tile_shuffle", "frame_shuffle",
                                    "zcurve_reorder"};
    return in_list(n, l, sizeof(l) / sizeof(l[0]));
}

bool Pipeline::reference_accepts(const std::string& s)
{
    // dynamic_pipeline.hpp:177-226
    const pairs_t pairs = parse_pairs(s);
    bool ok = true;
    const std::vector<std::string> majors = s.empty() ? std::vector<std::string>() : split_outside_verbatim(s, "->", &ok);
    if (!ok || majors.size() != pairs.size()) return false;
    uint32_t found = 0;
    bool sink_matched = false;
    for (const auto& p : pairs) {
        if (!sink_matched && known_head_filter(p.first)) { found++; continue; }
        if (known_sink(p.first)) { found++; sink_matched = true; continue; }
        if (known_tail_filter(p.first)) found++;
    }
    bool value = found == pairs.size();
    size_t rebuild = 2 * (pairs.size() - 1);          // size_t arithmetic, wraps for the empty string like the reference
    for (const auto& p : pairs) {
        rebuild += p.first.size();
        if (!p.second.empty()) rebuild += 2 + p.second.size();
    }
    return value && rebuild == s.size();
}

Pipeline Pipeline::from_string(const std::string& s, int elem_size)
{
    // dynamic_pipeline.hpp:137-170: head filters until the first sink, then tail filters; unknown names are skipped
    Pipeline p;
    for (const auto& pr : parse_pairs(s)) {
        Stage st;
        st.name = pr.first;
        st.kind = kind_of(pr.first);
        st.cfg = parse_minors(pr.second);
        if (st.kind == StageKind::lz4) st.lz4 = Lz4Params(pr.second);
        if (st.kind == StageKind::raster_reorder && elem_size > 0 && !st.cfg.count("tile_size"))
            st.cfg["tile_size"] = std::to_string(16 / (p.sink_index >= 0 ? 1 : elem_size));   // raster_reorder_scheme_impl.hpp:23 (behind the sink: raster_reorder_scheme<char>)
        if (p.sink_index < 0) {
            if (known_head_filter(pr.first)) { p.stages.push_back(st); continue; }
            if (known_sink(pr.first)) { p.sink_index = (int)p.stages.size(); p.stages.push_back(st); }
        } else if (known_tail_filter(pr.first)) {
            p.stages.push_back(st);
        }
    }
    return p;
}

bool Pipeline::supported(const std::string& s, int elem_size, std::string* why)
{
    auto fail = [&](const std::string& m) { if (why) *why = m; return false; };
    if (!reference_accepts(s)) return fail("not a valid sqeazy pipeline");
    const Pipeline p = from_string(s);
    if (p.stages.empty()) return fail("empty pipeline");
    for (size_t i = 0; i < p.stages.size(); ++i) {
        const Stage& st = p.stages[i];
        switch (st.kind) {
            case StageKind::bitswap1: break;
            case StageKind::diff3x3x1:
                break;                                                           // (head filter, or tail filter on the sink's char output)
            case StageKind::frame_shuffle: {
                auto c = st.cfg.find("frame_chunk_size");
                // (N > 1: N frames per sort unit when N divides the frames -- checked against the shape at encode time)
                if (c != st.cfg.end() && std::atoi(c->second.c_str()) < 1)
                    return fail("frame_shuffle: frame_chunk_size must be positive");
                break;
            }
            case StageKind::zcurve_reorder:
                break;                                                           // (head filter, or tail filter on the sink's char output: round 5)
            case StageKind::tile_shuffle: {
                auto t = st.cfg.find("tile_size");
                if (t != st.cfg.end() && std::atoi(t->second.c_str()) <= 0) return fail("tile_shuffle: tile_size must be positive");
                break;
            }
            case StageKind::bitshuffle: {
                auto b = st.cfg.find("block_size");
                if (b != st.cfg.end() && (std::atoi(b->second.c_str()) < 0 || std::atoi(b->second.c_str()) % 8))
                    return fail("bitshuffle: block_size must be a non-negative multiple of 8");
                break;
            }
            case StageKind::raster_reorder: {
                auto t = st.cfg.find("tile_size");
                if (t != st.cfg.end() && std::atoi(t->second.c_str()) <= 0) return fail("raster_reorder: tile_size must be positive");
                break;
            }
            case StageKind::quantiser: {
                if (elem_size != 2) return fail("quantiser: only 16-bit input is implemented on MI355X");
                auto w = st.cfg.find("weighting_function");
                QuantiserWeighting qw;
                if (w != st.cfg.end() && !quantiser_parse_weighting(w->second, &qw))
                    return fail("quantiser: weighting_function=" + w->second + " gives the reference a NaN or infinite exponent; refused");
                break;
            }
            case StageKind::lz4:
                if (i + 1 != p.stages.size()) return fail("lz4 must be the last stage on MI355X");
                if (st.lz4.accel >= 3) return fail("lz4 accel >= 3 selects LZ4HC in liblz4; not implemented on MI355X");
                break;
            case StageKind::pass_through:
                break;
            default:
                return fail("stage '" + st.name + "' is not implemented on MI355X");
        }
    }
    return true;
}

std::string Pipeline::name() const
{
    std::string v;
    for (size_t i = 0; i < stages.size(); ++i) {
        if (i) v += "->";
        v += stages[i].full_name();
    }
    return v;
}

bool raster_geometry_defined(uint64_t Z, uint64_t Y, uint64_t X, uint64_t ts, int elem_size)
{
    if (ts == 0 || Z == 0 || Y == 0 || X == 0) return false;
    const int nrem = (Z % ts != 0) + (Y % ts != 0) + (X % ts != 0);
    if (nrem != 0 && nrem != 3) return false;                            // raster_reorder_utils.hpp:271-305: extent-0 tiles
    const uint64_t block = 16 / (uint64_t)elem_size;
    if (nrem == 0 && ts % block == 0 && ts != block) return false;       // :160-243: encode_full_simd overwrites the tile row's head
    return true;
}

int clean_number_of_threads(int n)
{
    static int max_threads = (int)std::thread::hardware_concurrency();
    if (n > max_threads || n <= 0) n = max_threads;
    return n;
}

void Pipeline::set_n_threads(int n) { nthreads = (unsigned)clean_number_of_threads(n); }

uint64_t Pipeline::max_encoded_size(uint64_t nbytes, int elem_size) const
{
    // header(incoming_t(), nbytes, name()): rank-1 shape {nbytes}, payload = nbytes*sizeof(T) (sqeazy_header.hpp:229-251)
    const std::string hdr = header_pack(elem_size, false, std::vector<uint64_t>(1, nbytes), name(), nbytes * (uint64_t)elem_size);
    uint64_t best = 0;
    for (const Stage& st : stages) {
        uint64_t v = nbytes;
        if (st.kind == StageKind::lz4) v = st.lz4.max_encoded_size(nbytes, nthreads);
        else if (st.kind == StageKind::quantiser) v = nbytes * (uint64_t)elem_size + 256u * (uint64_t)elem_size;
        best = std::max(best, v);
    }
    return 2 * hdr.size() + best;
}

// ---- header ----
static void json_escape(const std::string& s, std::string& out)
{
    // Boost.PropertyTree json writer escapes: " \ / and control characters
    static const char* hex = "0123456789ABCDEF";
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '/': out += "\\/"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default:
                if (c < 0x20) { out += "\\u00"; out += hex[c >> 4]; out += hex[c & 15]; }
                else out += (char)c;
        }
    }
}

// the header as  [pad] prefix | payload bytes in decimal | suffix  (pad: spaces in front up to a multiple of elem_size, which
// depends on the number of digits): what a kernel needs to write the header once the payload size is known on the device
void header_pack_parts(int elem_size, bool is_signed8, const std::vector<uint64_t>& shape, const std::string& pipename,
                       std::string* prefix, std::string* suffix)
{
    // (the byte count is the only number behind `"bytes": "`; the text is searched from the back, where nothing a caller supplies sits)
    const std::string probe = header_pack(1, is_signed8 && elem_size == 1, shape, pipename, 0);
    const std::string key = "\"bytes\": \"";
    const size_t at = probe.rfind(key);
    *prefix = probe.substr(0, at + key.size());
    *suffix = probe.substr(at + key.size() + 1);                                    // (behind the single digit "0")
    if (elem_size == 2) {
        // header_pack(1, ..) wrote the 8-bit type name; the 16-bit header differs in that word only
        const std::string t8 = "\"type\": \"uint8\"", t16 = "\"type\": \"uint16\"";
        const size_t tp = prefix->rfind(t8);
        if (tp != std::string::npos) prefix->replace(tp, t8.size(), t16);
    }
}

std::string header_pack(int elem_size, bool is_signed8, const std::vector<uint64_t>& shape, const std::string& pipename,
                        uint64_t payload_bytes)
{
    const char* type = elem_size == 2 ? "uint16" : (is_signed8 ? "int8" : "uint8");   // header_utils.hpp:19-34
    std::string j;
    j += "{\n    \"pipename\": \"";
    json_escape(pipename, j);
    j += "\",\n    \"raw\": {\n        \"type\": \"";
    j += type;
    j += "\",\n        \"rank\": \"" + std::to_string(shape.size()) + "\",\n        \"shape\": {\n";
    for (size_t i = 0; i < shape.size(); ++i) {
        j += "            \"dim\": \"" + std::to_string(shape[i]) + "\"";
        j += (i + 1 < shape.size()) ? ",\n" : "\n";
    }
    j += "        }\n    },\n    \"encoded\": {\n        \"bytes\": \"" + std::to_string(payload_bytes) + "\"\n    },\n";
    j += "    \"sqy\": {\n        \"version\": \"";
    j += kVersion;
    j += "\",\n        \"headref\": \"";
    j += kHeadRef;
    j += "\"\n    }\n}\n";
    j += kHeaderEnd;
    if (j.size() % (size_t)elem_size != 0) j.insert(0, (size_t)elem_size - j.size() % (size_t)elem_size, ' ');
    return j;
}

int HeaderInfo::elem_size() const
{
    if (type == "uint16" || type == "int16") return 2;
    if (type == "uint8" || type == "int8") return 1;
    if (type == "uint32" || type == "int32") return 4;
    if (type == "uint64" || type == "int64") return 8;
    return 0;
}

static bool read_json_string(const char*& p, const char* end, std::string& out)
{
    out.clear();
    if (p >= end || *p != '"') return false;
    ++p;
    while (p < end && *p != '"') {
        if (*p == '\\' && p + 1 < end) {
            ++p;
            switch (*p) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u':
                    if (p + 4 < end) { out += (char)std::strtol(std::string(p + 1, p + 5).c_str(), nullptr, 16); p += 4; }
                    break;
                default: out += *p;
            }
            ++p;
        } else {
            out += *p++;
        }
    }
    if (p >= end) return false;
    ++p;
    return true;
}

HeaderInfo header_unpack(const char* begin, const char* end)
{
    HeaderInfo h;
    const char* delim = std::search(begin, end, kHeaderEnd.begin(), kHeaderEnd.end());
    if (delim == end) return h;
    // sqeazy_header.hpp:520-531 (valid_header): balanced braces, more than one ':'
    if (std::count(begin, delim, '{') == 0 || std::count(begin, delim, '{') != std::count(begin, delim, '}')) return h;
    if (std::count(begin, delim, ':') <= 1) return h;
    const char* p = begin;
    std::string key, val;
    bool have_pipe = false, have_bytes = false;
    while (p < delim) {
        if (*p != '"') { ++p; continue; }
        if (!read_json_string(p, delim, key)) break;
        while (p < delim && (*p == ' ' || *p == '\t' || *p == '\n')) ++p;
        if (p >= delim || *p != ':') continue;
        ++p;
        while (p < delim && (*p == ' ' || *p == '\t' || *p == '\n')) ++p;
        if (p < delim && *p == '"') {
            if (!read_json_string(p, delim, val)) break;
            if (key == "pipename") { h.pipename = val; have_pipe = true; }
            else if (key == "type") h.type = val;
            else if (key == "dim") h.shape.push_back(std::strtoull(val.c_str(), nullptr, 10));
            else if (key == "bytes") { h.payload_bytes = std::strtoull(val.c_str(), nullptr, 10); have_bytes = true; }
        }
    }
    h.size = (uint64_t)(delim - begin) + kHeaderEnd.size();
    h.valid = have_pipe && have_bytes && !h.type.empty();
    return h;
}

// ---- quantiser LUT construction (host, float) ----
bool quantiser_parse_weighting(const std::string& text, QuantiserWeighting* out)
{
    QuantiserWeighting w;
    if (text.find("none") != std::string::npos) { *out = w; return true; }           // quantiser_scheme_impl.hpp:186
    if (text.rfind('_') == std::string::npos) return false;                          // extract_ratio :29-32 -> 0/0
    std::vector<long> ints;                                                          // regex_helpers.hpp:101-115, "[0-9]+"
    for (size_t i = 0; i < text.size();) {
        if (text[i] < '0' || text[i] > '9') { ++i; continue; }
        size_t j = i;
        long v = 0;
        while (j < text.size() && text[j] >= '0' && text[j] <= '9') {
            v = v * 10 + (text[j] - '0');
            if (v > INT32_MAX) return false;                                         // (std::stoi would throw)
            ++j;
        }
        ints.push_back(v);
        i = j;
    }
    if (ints.size() == 1) ints.push_back(1);
    if (ints.size() != 2 || ints[1] == 0) return false;
    w.mode = text.find("offset") != std::string::npos ? 2 : 1;                        // :189
    w.num = (int)ints[0];
    w.den = (int)ints[1];
    *out = w;
    return true;
}

bool quantiser_lut_to_file(const std::string& path, const uint16_t* lut, size_t n)
{
    FILE* f = std::fopen(path.c_str(), "w");
    if (!f) return false;
    for (size_t i = 0; i < n; ++i) std::fprintf(f, "%u\n", (unsigned)lut[i]);
    return std::fclose(f) == 0;
}

// The path may come out of a blob's header, i.e. from untrusted input (decode): only a REGULAR file of at most 64 KiB is opened (a
// FIFO or a device would block the call or feed it without end), and it has to hold exactly n values that fit 16 bits -- a short
// or garbled file is an error here, where the reference decodes with whatever its stream extraction left in the table.
bool quantiser_lut_from_file(const std::string& path, uint16_t* lut, size_t n)
{
    for (size_t i = 0; i < n; ++i) lut[i] = 0;
    struct stat sb;
    if (::stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size > (64 << 10)) return false;
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    unsigned long v = 0;
    size_t i = 0;
    bool ok = true;
    while (i < n && std::fscanf(f, "%lu", &v) == 1) {
        if (v > 65535ul) { ok = false; break; }
        lut[i++] = (uint16_t)v;
    }
    if (ok && i == n && std::fscanf(f, "%lu", &v) == 1) ok = false;          // more than n values
    std::fclose(f);
    return ok && i == n;
}

void quantiser_build_luts(const uint32_t* histo, size_t nbins, unsigned char* lut_encode, uint16_t* lut_decode,
                          const QuantiserWeighting& weighting)
{
    const size_t max_compressed = 256;                                     // quantiser<raw, char>::max_compressed_
    std::vector<float> importance(nbins);
    std::memset(lut_encode, 0, nbins);
    for (size_t i = 0; i < max_compressed; ++i) lut_decode[i] = 0;
    // computeWeights (quantiser_utils.hpp:317-322): the weights start out as 1.f (:85,104); power_of writes std::pow(i, e)
    // into every bin, offset_power_of std::pow(i - offset, e) from the first non-zero bin on (quantiser_weighters.hpp:40-84,
    // :121-136).  The index is an integer, so std::pow works in double and the result is narrowed to float.
    std::vector<float> weights(nbins, 1.f);
    if (weighting.mode != 0) {
        const float exponent = float(weighting.num) / weighting.den;
        size_t offset = 0;
        if (weighting.mode == 2) while (offset < nbins && !histo[offset]) ++offset;
        for (size_t i = offset; i < nbins; ++i) weights[i] = (float)std::pow((double)(long long)(i - offset), (double)exponent);
    }
    // computeImportance: importance = histo * weight
    for (size_t i = 0; i < nbins; ++i) importance[i] = histo[i] * weights[i];
    // std::accumulate(importance.begin(), importance.end(), 0.) -> double accumulator, assigned to float
    double total = 0.;
    for (size_t i = 0; i < nbins; ++i) total = total + importance[i];
    const float importanceSum = (float)total;
    if (!(importanceSum != 0)) return;
    uint32_t n_levels = 0;
    for (size_t i = 0; i < nbins; ++i) if (importance[i] != 0.f) ++n_levels;

    if (n_levels <= max_compressed) {
        // linear_mapping_quantisation
        uint32_t comp_idx = 0;
        for (uint32_t raw_idx = 0; raw_idx < nbins && comp_idx < max_compressed; ++raw_idx) {
            lut_encode[raw_idx] = (unsigned char)comp_idx;
            lut_decode[comp_idx] = (uint16_t)raw_idx;
            if (importance[raw_idx]) comp_idx++;
        }
        const uint16_t raw_max = nbins == 65536 ? 65535 : 255;
        if (comp_idx < max_compressed && comp_idx > 0 && lut_decode[comp_idx] == raw_max)
            for (size_t i = comp_idx; i < max_compressed; ++i) lut_decode[i] = lut_decode[comp_idx - 1];
        return;
    }
    // adaptive_lloyd_com
    size_t levels_available = max_compressed;
    float bucketSize = importanceSum / levels_available;
    float importanceIntegral = importance[0];
    float quantile_sum = importance[0];
    uint32_t comp_idx = 0;
    float weighted_mean_importance_in_bucket = 0 * importance[0];
    float index_weighted_mean_importance = 0;
    for (uint32_t raw_idx = 1; raw_idx < nbins; ++raw_idx) {
        if (quantile_sum >= bucketSize && (comp_idx < max_compressed - 1)) {
            lut_decode[comp_idx] = static_cast<uint16_t>(index_weighted_mean_importance);
            comp_idx++;
            levels_available--;
            quantile_sum = importance[raw_idx];
            weighted_mean_importance_in_bucket = raw_idx * importance[raw_idx];
            if (importanceIntegral < importanceSum) bucketSize = (importanceSum - importanceIntegral) / levels_available;
            if (quantile_sum != 0.) index_weighted_mean_importance = std::round(weighted_mean_importance_in_bucket / quantile_sum);
        } else {
            quantile_sum += importance[raw_idx];
            weighted_mean_importance_in_bucket += raw_idx * importance[raw_idx];
            if (quantile_sum != 0.) index_weighted_mean_importance = std::round(weighted_mean_importance_in_bucket / quantile_sum);
        }
        lut_encode[raw_idx] = static_cast<unsigned char>(comp_idx);
        importanceIntegral += importance[raw_idx];
    }
    lut_decode[comp_idx] = static_cast<uint16_t>(index_weighted_mean_importance);
}

bool zcurve_geometry_defined(uint64_t Z, uint64_t Y, uint64_t X, uint64_t ts)
{
    if (Z == 0 || Y == 0 || X == 0) return false;
    if (ts < 2 || ts > 128 || (ts & (ts - 1))) return false;                  // zcurve_reorder_utils.hpp:30-67: other sizes get 1-bit Morton codes that leave the tile
    auto flog2 = [](uint64_t v) { int l = 0; while (v >>= 1) ++l; return l; };
    const uint64_t common = (uint64_t)1 << std::min(flog2(Z), std::min(flog2(Y), flog2(X)));   // :69-81 common_power_of_2
    const bool has_remainder = (Z % common) || (Y % common) || (X % common);                   // :83-94, :125-131
    if (!has_remainder && ((Z % ts) || (Y % ts) || (X % ts))) return false;                    // encode_full (:141-186) assumes whole tiles
    return true;
}

bool tile_shuffle_geometry_defined(uint64_t Z, uint64_t Y, uint64_t X, uint64_t ts)
{
    return ts > 0 && Z >= ts && Y >= ts && X >= ts && Z % ts == 0 && Y % ts == 0 && X % ts == 0;
}

void tile_shuffle_order(const float* sums, size_t ntiles, size_t per_tile, int elem_size, uint64_t* decode_map, bool signed_char)
{
    // tile_shuffle_utils.hpp:176-219: std::vector<in_value_t> metric; metric[i] = sum / n_elements_per_tile (float -> voxel type).
    // signed_char: the tail filter form, tile_shuffle_scheme<char> -- the metric is a (signed) char and sorts as one
    std::vector<int32_t> metric(ntiles);
    for (size_t i = 0; i < ntiles; ++i) {
        const float m = sums[i] / per_tile;
        metric[i] = signed_char ? (int32_t)(int8_t)m : elem_size == 2 ? (int32_t)(uint16_t)m : (int32_t)(uint8_t)m;
    }
    std::vector<int32_t> sorted_metric = metric;
    std::sort(sorted_metric.begin(), sorted_metric.end());
    // std::find(metric, sorted[i]) for every i = the first tile with that metric
    std::map<int32_t, uint64_t> first;
    for (size_t i = ntiles; i-- > 0;) first[metric[i]] = i;
    for (size_t i = 0; i < ntiles; ++i) decode_map[i] = first[sorted_metric[i]];
}

uint64_t bitshuffle_block_elems(int elem_size, uint64_t block_size)
{
    if (block_size == 0) {
        // bshuf_default_block_size (kiyo-masui/bitshuffle, bitshuffle_core.c): "needs to be absolutely stable between versions"
        uint64_t b = 8192 / (uint64_t)elem_size;
        b = (b / 8) * 8;
        return b < 128 ? 128 : b;
    }
    return block_size % 8 ? 0 : block_size;
}

void frame_shuffle_order(const float* sums, size_t Z, size_t per_frame, uint64_t* decode_map)
{
    std::vector<float> metric(Z);
    for (size_t z = 0; z < Z; ++z) metric[z] = sums[z] / per_frame;          // float / size_t
    // The reference sorts the metrics and looks every sorted value up with std::find (frame_shuffle_utils.hpp:138-161): slot i gets
    // the FIRST frame whose metric equals the i-th smallest.  The same map in Z log Z instead of Z^2 compares (0.3 ms of host time per
    // call at 1024 frames): a stable argsort puts the frames of equal metric in index order, every slot of such a run takes the run's
    // first frame.
    std::vector<uint32_t> idx(Z);
    for (size_t z = 0; z < Z; ++z) idx[z] = (uint32_t)z;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return metric[a] < metric[b]; });
    uint32_t first = 0;
    for (size_t i = 0; i < Z; ++i) {
        if (i == 0 || !(metric[idx[i]] == metric[idx[i - 1]])) first = idx[i];
        decode_map[i] = first;
    }
}

// ---- base64 (base64.hpp:135-162: standard alphabet, '=' padded) ----
std::string base64_encode(const unsigned char* src, size_t n)
{
    static const char tbl[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string o;
    o.reserve(4 * ((n + 2) / 3));
    size_t i = 0;
    for (; i + 2 < n; i += 3) {
        const unsigned v = ((unsigned)src[i] << 16) | ((unsigned)src[i + 1] << 8) | src[i + 2];
        o += tbl[(v >> 18) & 63]; o += tbl[(v >> 12) & 63]; o += tbl[(v >> 6) & 63]; o += tbl[v & 63];
    }
    if (n - i == 1) {
        const unsigned v = (unsigned)src[i] << 16;
        o += tbl[(v >> 18) & 63]; o += tbl[(v >> 12) & 63]; o += "==";
    } else if (n - i == 2) {
        const unsigned v = ((unsigned)src[i] << 16) | ((unsigned)src[i + 1] << 8);
        o += tbl[(v >> 18) & 63]; o += tbl[(v >> 12) & 63]; o += tbl[(v >> 6) & 63]; o += '=';
    }
    return o;
}

std::vector<unsigned char> base64_decode(const std::string& s)
{
    std::vector<unsigned char> out;
    unsigned acc = 0;
    int bits = 0;
    for (char ch : s) {
        int v;
        if (ch >= 'A' && ch <= 'Z') v = ch - 'A';
        else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26;
        else if (ch >= '0' && ch <= '9') v = ch - '0' + 52;
        else if (ch == '+') v = 62;
        else if (ch == '/') v = 63;
        else continue;
        acc = (acc << 6) | (unsigned)v;
        bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((unsigned char)((acc >> bits) & 0xff)); }
    }
    return out;
}

std::string to_verbatim(const void* data, size_t bytes)
{
    if (!bytes) return "";
    return kVerbOpen + base64_encode(static_cast<const unsigned char*>(data), bytes) + kVerbClose;   // string_parsers.hpp:507-534
}

// ---- xxh32 (LZ4 frame descriptor checksum) ----
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t rd32(const unsigned char* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }

uint32_t xxh32(const unsigned char* p, size_t len, uint32_t seed)
{
    const uint32_t P1 = 2654435761U, P2 = 2246822519U, P3 = 3266489917U, P4 = 668265263U, P5 = 374761393U;
    const unsigned char* const end = p + len;
    uint32_t h;
    if (len >= 16) {
        const unsigned char* const limit = end - 16;
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

} // namespace sqy
