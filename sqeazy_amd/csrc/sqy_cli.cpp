// sqy -- command line front end over libsqeazy_amd's C-ABI (include/sqeazy_amd.h), modelled on the reference's
// `sqy` tool (/root/reference/src/cpp/src/sqy.cpp:183-330 and src/verbs/{compress,decompress,bench,compare}.hpp):
// same verbs and aliases, same option names, same .sqy files (a .sqy file IS the blob SQY_PipelineEncode_* returns).
//
//   sqy compress   [-p pipeline] [-o out | -e suffix] [-n nthreads] [-v] stack.tif ... | stack.raw -s ZxYxX -t uint16
//   sqy decompress [-o out | -e .tif|.raw] [-n nthreads] [-v] stack.sqy ...
//   sqy scan       stack.tif | stack.sqy ...
//   sqy compare    a.tif b.tif
//   sqy bench      [-p pipeline] [-r repetitions] [-c] [--noheader] [--comment text] stack.tif ...
//
// Differences, all forced by the hardware behind the library:
//   * -n/--nthreads defaults to 1 like the reference (src/sqy.cpp:190): ONE block-linked LZ4 frame -- bit-identical to the
//     reference's default output; its blocks are parsed block-parallel from verified table guesses (about a third of the chunked
//     layout's rate on microscopy stacks; data whose guesses fail, e.g. the sequence-heavy plane of a quantised stack, is
//     parsed in order and is slow: DESIGN.md section 3).  Pass -n 0 for the chunked layout (independent frames, the fast path);
//   * outputs other than .sqy (the reference can wrap the blob into .tif or .h5) are not written;
//   * TIFF input/output is the uncompressed 8/16-bit grayscale subset the reference itself writes (tiff_utils.hpp:
//     286-310), read and written here without libtiff; .raw needs --shape and --dtype.
// Uses nothing but the exported C symbols -- it doubles as the example of a third-party caller of the drop-in.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/sqeazy_amd.h"

namespace {

struct Stack {
    std::vector<long> shape;          // {z, y, x}
    int bytes_per_voxel = 0;
    std::vector<char> data;
    size_t voxels() const { size_t n = 1; for (long s : shape) n *= (size_t)s; return n; }
};

// ------------------------------------------------------------------------------------------------------------------
// minimal TIFF stack reader/writer: classic and BigTIFF, either byte order, one sample per pixel, 8 or 16 bits,
// uncompressed strips; one IFD per frame (what libtiff-based writers including the reference emit), or ImageJ's
// single-IFD "images=N" contiguous layout for stacks beyond 4 GiB
// ------------------------------------------------------------------------------------------------------------------
struct TiffReader {
    std::ifstream f;
    bool big = false, swap = false;
    std::string err;

    template <typename T> T rd(uint64_t off)
    {
        T v = 0;
        f.seekg((std::streamoff)off);
        f.read(reinterpret_cast<char*>(&v), sizeof(T));
        if (!f) { err = "truncated file"; return 0; }
        if (swap) { T r = 0; for (size_t i = 0; i < sizeof(T); ++i) r |= ((v >> (8 * i)) & 0xff) << (8 * (sizeof(T) - 1 - i)); v = r; }
        return v;
    }
    struct Entry { uint16_t tag, type; uint64_t count, value_off; };   // value_off: file offset of the value bytes
    static size_t type_size(uint16_t t)
    {
        switch (t) { case 1: case 2: case 6: case 7: return 1; case 3: case 8: return 2; case 4: case 9: case 11: case 13: return 4;
                     case 5: case 10: case 12: case 16: case 17: case 18: return 8; default: return 0; }
    }
    uint64_t value(const Entry& e, uint64_t i)
    {
        const uint64_t o = e.value_off + i * type_size(e.type);
        if (o < e.value_off) { err = "value offset wraps"; return 0; }
        switch (type_size(e.type)) { case 1: return rd<uint8_t>(o); case 2: return rd<uint16_t>(o); case 4: return rd<uint32_t>(o); case 8: return rd<uint64_t>(o); }
        return 0;
    }
    // reads one IFD, returns the offset of the next one
    uint64_t read_ifd(uint64_t off, std::map<uint16_t, Entry>& tags)
    {
        const uint64_t n = big ? rd<uint64_t>(off) : rd<uint16_t>(off);
        uint64_t p = off + (big ? 8 : 2);
        const uint64_t esz = big ? 20 : 12, inl = big ? 8 : 4;
        if (n > 4096) { err = "implausible IFD"; return 0; }
        for (uint64_t i = 0; i < n && err.empty(); ++i, p += esz) {
            Entry e;
            e.tag = rd<uint16_t>(p); e.type = rd<uint16_t>(p + 2);
            e.count = big ? rd<uint64_t>(p + 4) : rd<uint32_t>(p + 4);
            const uint64_t vpos = p + (big ? 12 : 8);
            const uint64_t bytes = e.count * type_size(e.type);
            e.value_off = bytes <= inl ? vpos : (big ? rd<uint64_t>(vpos) : rd<uint32_t>(vpos));
            tags[e.tag] = e;
        }
        return big ? rd<uint64_t>(p) : rd<uint32_t>(p);
    }
    bool load(const std::string& path, Stack& out, bool header_only = false)
    {
        f.open(path, std::ios::binary);
        if (!f) { err = "unable to open"; return false; }
        // the file is untrusted input: everything read from it is checked against the file's size (the strips are uncompressed, so
        // a stack can never be larger than the file that holds it), IFD chains may not revisit an offset
        f.seekg(0, std::ios::end);
        const uint64_t file_size = (uint64_t)std::max<std::streamoff>(f.tellg(), 0);
        f.seekg(0);
        char bo[2] = {0, 0};
        f.read(bo, 2);
        if (!f) { err = "not a TIFF file"; return false; }
        const uint16_t probe = 1;
        const bool host_le = *reinterpret_cast<const uint8_t*>(&probe) == 1;
        if (bo[0] == 'I' && bo[1] == 'I') swap = !host_le; else if (bo[0] == 'M' && bo[1] == 'M') swap = host_le; else { err = "not a TIFF file"; return false; }
        const uint16_t magic = rd<uint16_t>(2);
        if (magic == 43) big = true; else if (magic != 42) { err = "not a TIFF file"; return false; }
        uint64_t ifd = big ? rd<uint64_t>(8) : rd<uint32_t>(4);
        uint64_t w = 0, h = 0, bits = 0, frames = 0;
        std::vector<std::pair<uint64_t, uint64_t>> strips;     // (offset, bytes) in frame order
        uint64_t imagej_images = 0;
        std::set<uint64_t> seen_ifds;
        while (ifd && err.empty()) {
            if (ifd >= file_size || !seen_ifds.insert(ifd).second) { err = "IFD chain leaves the file or loops"; break; }
            std::map<uint16_t, Entry> t;
            const uint64_t next = read_ifd(ifd, t);
            if (!err.empty()) break;
            if (!t.count(256) || !t.count(257) || !t.count(273)) { err = "missing size/strip tags"; break; }
            const uint64_t fw = value(t[256], 0), fh = value(t[257], 0);
            const uint64_t fbits = t.count(258) ? value(t[258], 0) : 1;
            const uint64_t comp = t.count(259) ? value(t[259], 0) : 1;
            const uint64_t spp = t.count(277) ? value(t[277], 0) : 1;
            if (comp != 1) { err = "compressed TIFF strips are not supported (write the stack uncompressed)"; break; }
            if (spp != 1 || (fbits != 8 && fbits != 16)) { err = "only 8/16-bit single-sample stacks are supported"; break; }
            if (frames == 0) { w = fw; h = fh; bits = fbits; }
            else if (fw != w || fh != h || fbits != bits) { err = "frames of different shape/type"; break; }
            if (fw == 0 || fh == 0 || fw > file_size || fh > file_size || fw * fh > file_size) { err = "frame larger than the file"; break; }
            const uint64_t nstrips = t[273].count;
            if (nstrips == 0 || nstrips > fh || (t.count(279) && t[279].count < nstrips)) { err = "implausible strip table"; break; }
            uint64_t have = 0;
            for (uint64_t s = 0; s < nstrips && err.empty(); ++s) {
                const uint64_t so = value(t[273], s);
                uint64_t sb = t.count(279) ? value(t[279], s) : fw * fh * (fbits / 8);
                if (so > file_size || sb > file_size - so) { err = "strip outside the file"; break; }
                strips.emplace_back(so, sb);
                have += sb;
            }
            if (!err.empty()) break;
            if (t.count(270) && frames == 0 && t[270].count <= (1u << 20) && t[270].value_off <= file_size &&
                t[270].count <= file_size - t[270].value_off) {   // ImageJ: "ImageJ=...\nimages=N\n..."
                std::string d((size_t)t[270].count, '\0');
                f.seekg((std::streamoff)t[270].value_off); f.read(&d[0], (std::streamsize)d.size());
                const size_t k = d.find("images=");
                if (d.compare(0, 6, "ImageJ") == 0 && k != std::string::npos) imagej_images = std::strtoull(d.c_str() + k + 7, nullptr, 10);
            }
            (void)have;
            ++frames;
            ifd = next;
        }
        if (!err.empty()) return false;
        if (frames == 0) { err = "no image in file"; return false; }
        const uint64_t frame_bytes = w * h * (bits / 8);
        if (frames == 1 && imagej_images > 1) {                  // contiguous hyperstack behind the first strip
            const uint64_t first = strips.front().first;
            if (frame_bytes == 0 || imagej_images > (file_size - first) / frame_bytes) { err = "ImageJ stack larger than the file"; return false; }
            frames = imagej_images;
            strips.assign(1, std::make_pair(first, frames * frame_bytes));
        }
        if (frame_bytes == 0 || frames > file_size / frame_bytes) { err = "stack larger than the file"; return false; }
        out.shape = {(long)frames, (long)h, (long)w};
        out.bytes_per_voxel = (int)(bits / 8);
        if (header_only) return true;
        out.data.resize(frames * frame_bytes);
        uint64_t pos = 0;
        for (auto& s : strips) {
            uint64_t nb = s.second;
            if (pos + nb > out.data.size()) nb = out.data.size() - pos;
            f.seekg((std::streamoff)s.first);
            f.read(out.data.data() + pos, (std::streamsize)nb);
            if (!f) { err = "truncated strip"; return false; }
            pos += nb;
        }
        if (pos != out.data.size()) { err = "strip sizes do not add up to the frames"; return false; }
        if (swap && bits == 16) for (size_t i = 0; i + 1 < out.data.size(); i += 2) std::swap(out.data[i], out.data[i + 1]);
        return true;
    }
};

// one uncompressed strip per frame, little endian; BigTIFF when offsets would not fit 32 bits.  Tags as the reference's
// writer sets them (tiff_utils.hpp:286-310).
bool write_tiff(const std::string& path, const Stack& s)
{
    if (s.shape.size() != 3) return false;
    const uint64_t frames = (uint64_t)s.shape[0], h = (uint64_t)s.shape[1], w = (uint64_t)s.shape[2];
    const uint64_t frame_bytes = w * h * (uint64_t)s.bytes_per_voxel;
    const bool big = frames * (frame_bytes + 256) + 64 > 0xfff00000ull;
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    auto put = [&](uint64_t v, int n) { f.write(reinterpret_cast<const char*>(&v), n); };   // little-endian host assumed (x86-64)
    struct T { uint16_t tag, type; uint64_t count, value; };
    const uint64_t nent = 12, esz = big ? 20 : 12;
    const uint64_t ifd_bytes = (big ? 8 : 2) + nent * esz + (big ? 8 : 4);
    const uint64_t hdr = big ? 16 : 8;
    f.write("II", 2);
    if (big) { put(43, 2); put(8, 2); put(0, 2); put(hdr, 8); } else { put(42, 2); put(hdr, 4); }
    // layout: header | per frame: IFD, pixel data
    uint64_t off = hdr;
    for (uint64_t z = 0; z < frames; ++z) {
        const uint64_t data_off = off + ifd_bytes;
        const uint64_t next = (z + 1 < frames) ? data_off + frame_bytes + ((data_off + frame_bytes) & 1) : 0;
        const T tags[nent] = {{254, 4, 1, 2}, {256, 4, 1, w}, {257, 4, 1, h}, {258, 3, 1, (uint64_t)s.bytes_per_voxel * 8}, {259, 3, 1, 1}, {262, 3, 1, 1},
                              {273, (uint16_t)(big ? 16 : 4), 1, data_off}, {277, 3, 1, 1}, {278, 4, 1, h}, {279, (uint16_t)(big ? 16 : 4), 1, frame_bytes},
                              {297, 3, 2, z | (frames << 16)}, {339, 3, 1, 1}};
        put(nent, big ? 8 : 2);
        for (const T& t : tags) { put(t.tag, 2); put(t.type, 2); put(t.count, big ? 8 : 4); put(t.value, big ? 8 : 4); }
        put(next, big ? 8 : 4);
        f.write(s.data.data() + z * frame_bytes, (std::streamsize)frame_bytes);
        if ((data_off + frame_bytes) & 1) f.put('\0');
        off = next;
    }
    return (bool)f;
}

bool read_file(const std::string& path, std::vector<char>& out)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const std::streamoff n = f.tellg();
    out.resize((size_t)n);
    f.seekg(0);
    f.read(out.data(), n);
    return (bool)f;
}

std::string ext_of(const std::string& p)
{
    const size_t s = p.find_last_of('/'), d = p.find_last_of('.');
    if (d == std::string::npos || (s != std::string::npos && d < s)) return "";
    return p.substr(d);
}
std::string with_ext(const std::string& p, const std::string& e)
{
    const std::string x = ext_of(p);
    const std::string stem = p.substr(0, p.size() - x.size());
    return stem + e;                                              // suffix without a period is appended to the stem, like the reference
}

struct Options {
    std::string verb;
    std::vector<std::string> files;
    std::string pipeline = "bitswap1->lz4", output_name, output_suffix, comment, shape, dtype = "uint16";
    int nthreads = 1, repetitions = 10;
    bool verbose = false, help = false, as_csv = false, noheader = false;
};

bool parse_shape(const std::string& s, std::vector<long>& shape)
{
    shape.clear();
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, 'x')) { if (item.empty()) return false; shape.push_back(std::atol(item.c_str())); }
    return shape.size() == 3 && shape[0] > 0 && shape[1] > 0 && shape[2] > 0;
}

bool load_stack(const std::string& path, const Options& o, Stack& st, std::string& err)
{
    const std::string e = ext_of(path);
    if (e == ".tif" || e == ".tiff" || e == ".TIF" || e == ".TIFF") {
        TiffReader r;
        if (!r.load(path, st)) { err = r.err; return false; }
        return true;
    }
    if (e == ".raw") {
        if (!parse_shape(o.shape, st.shape)) { err = ".raw input needs --shape ZxYxX"; return false; }
        st.bytes_per_voxel = (o.dtype == "uint8" || o.dtype == "8") ? 1 : 2;
        if (!read_file(path, st.data)) { err = "unable to open"; return false; }
        if (st.data.size() != st.voxels() * (size_t)st.bytes_per_voxel) { err = "file size does not match --shape/--dtype"; return false; }
        return true;
    }
    err = "unknown input format (expected .tif or .raw)";
    return false;
}

int encode_stack(const Options& o, Stack& st, std::vector<char>& blob, long& written)
{
    long len = (long)o.pipeline.size();
    int rc = st.bytes_per_voxel == 2 ? SQY_Pipeline_Max_Compressed_Length_3D_UI16(o.pipeline.c_str(), st.shape.data(), 3, &len)
                                     : SQY_Pipeline_Max_Compressed_Length_3D_UI8(o.pipeline.c_str(), st.shape.data(), 3, &len);
    if (rc) return rc;
    if ((long)blob.size() != len) blob.assign((size_t)len, 0);
    written = 0;
    return st.bytes_per_voxel == 2 ? SQY_PipelineEncode_UI16(o.pipeline.c_str(), st.data.data(), st.shape.data(), 3, blob.data(), &written, o.nthreads)
                                   : SQY_PipelineEncode_UI8(o.pipeline.c_str(), st.data.data(), st.shape.data(), 3, blob.data(), &written, o.nthreads);
}

std::string output_for(const Options& o, const std::string& in, const std::string& default_ext)
{
    if (!o.output_name.empty() && o.files.size() == 1) return o.output_name;
    return with_ext(in, o.output_suffix.empty() ? default_ext : o.output_suffix);
}

int do_compress(const Options& o)
{
    const bool ok16 = SQY_Pipeline_Possible_UI16(o.pipeline.c_str()), ok8 = SQY_Pipeline_Possible_UI8(o.pipeline.c_str());
    if (!ok16 && !ok8) { std::cerr << "[SQY]\tunable to build pipeline from " << o.pipeline << "\nDoing nothing.\n"; return 1; }
    if (o.files.size() > 1 && !o.output_name.empty()) std::cout << "[SQY]\tmultiple input files detected, ignoring --output_name flag\n";
    int ret = 0;
    std::vector<char> blob;
    for (const std::string& file : o.files) {
        Stack st; std::string err;
        if (!load_stack(file, o, st, err)) { std::cerr << "[SQY]\tunable to open " << file << " (" << err << ")\t skipping it\n"; ret = 1; continue; }
        const std::string out = output_for(o, file, ".sqy");
        if (ext_of(out) != ".sqy") { std::cerr << "[SQY]\toutput format " << ext_of(out) << " is not written by this build (native .sqy only)\n"; ret = 1; continue; }
        if (!(st.bytes_per_voxel == 2 ? ok16 : ok8)) { std::cerr << "[SQY]\tpipeline " << o.pipeline << " cannot be applied to " << st.bytes_per_voxel * 8 << "-bit data\n"; ret = 1; continue; }
        long written = 0;
        if (encode_stack(o, st, blob, written)) { std::cerr << "[SQY]\tnative compression failed! Nothing to write to disk...\n"; ret = 1; continue; }
        std::ofstream f(out, std::ios::binary);
        if (!f) { std::cerr << "[SQY]\tunable to open " << out << " as output file. Skipping it!\n"; ret = 1; continue; }
        f.write(blob.data(), written);
        if (o.verbose) std::cout << "[SQY]\t" << file << " -> " << out << " " << st.data.size() << " -> " << written << " bytes (ratio "
                                 << (written ? (double)st.data.size() / (double)written : 0.0) << ")\n";
    }
    return ret;
}

int decode_blob(const std::vector<char>& blob, int nthreads, Stack& st, std::string& pipeline_info)
{
    long v = (long)blob.size();
    if (SQY_Decompressed_NDims(blob.data(), &v)) return 1;
    const long rank = v;
    if (rank < 1 || rank > 8) return 1;
    std::vector<long> shape((size_t)rank, 0);
    shape[0] = (long)blob.size();
    if (SQY_Decompressed_Shape(blob.data(), shape.data())) return 1;
    v = (long)blob.size();
    if (SQY_Decompressed_Sizeof(blob.data(), &v)) return 1;
    st.bytes_per_voxel = (int)v;
    v = (long)blob.size();
    if (SQY_Decompressed_Length(blob.data(), &v)) return 1;
    st.shape = shape;
    st.data.assign((size_t)v, 0);
    (void)pipeline_info;
    return st.bytes_per_voxel == 2 ? SQY_Decode_UI16(blob.data(), (long)blob.size(), st.data.data(), nthreads)
                                   : SQY_Decode_UI8(blob.data(), (long)blob.size(), st.data.data(), nthreads);
}

int do_decompress(const Options& o)
{
    int ret = 0;
    for (const std::string& file : o.files) {
        std::vector<char> blob;
        if (!read_file(file, blob)) { std::cerr << "[SQY]\tunable to open " << file << "\t skipping it\n"; ret = 1; continue; }
        Stack st; std::string info;
        if (decode_blob(blob, o.nthreads, st, info)) { std::cerr << "[SQY]\tdecompression of " << file << " failed\n"; ret = 1; continue; }
        const std::string out = output_for(o, file, ".tif");
        bool ok = false;
        if (ext_of(out) == ".raw") { std::ofstream f(out, std::ios::binary); f.write(st.data.data(), (std::streamsize)st.data.size()); ok = (bool)f; }
        else if (st.shape.size() == 3) ok = write_tiff(out, st);
        else std::cerr << "[SQY]\tonly rank-3 stacks can be written as TIFF (use -e .raw)\n";
        if (!ok) { std::cerr << "[SQY]\tunable to write " << out << "\n"; ret = 1; continue; }
        if (o.verbose) std::cout << "[SQY]\t" << file << " -> " << out << " " << blob.size() << " -> " << st.data.size() << " bytes\n";
    }
    return ret;
}

int do_scan(const Options& o)
{
    int ret = 0;
    for (const std::string& file : o.files) {
        if (ext_of(file) == ".sqy") {
            std::vector<char> blob;
            if (!read_file(file, blob)) { std::cerr << "[SQY]\tunable to open " << file << "\n"; ret = 1; continue; }
            long hs = (long)blob.size();
            if (SQY_Header_Size(blob.data(), &hs)) { std::cerr << "[SQY]\t" << file << " has no sqy header\n"; ret = 1; continue; }
            std::cout << file << ":\n" << std::string(blob.data(), (size_t)hs) << "\n"
                      << "payload bytes: " << blob.size() - (size_t)hs << "\n";
            continue;
        }
        Stack st; std::string err;
        if (!load_stack(file, o, st, err)) { std::cerr << "[SQY]\tunable to open " << file << " (" << err << ")\n"; ret = 1; continue; }
        uint64_t lo = ~0ull, hi = 0; long double sum = 0;
        const size_t n = st.voxels();
        for (size_t i = 0; i < n; ++i) {
            const uint64_t v = st.bytes_per_voxel == 2 ? reinterpret_cast<const uint16_t*>(st.data.data())[i] : (uint8_t)st.data[i];
            lo = v < lo ? v : lo; hi = v > hi ? v : hi; sum += (long double)v;
        }
        std::cout << "filename,shape,bits,min,max,mean\n" << file << "," << st.shape[2] << "x" << st.shape[1] << "x" << st.shape[0] << ","
                  << st.bytes_per_voxel * 8 << "," << lo << "," << hi << "," << (double)(sum / (long double)(n ? n : 1)) << "\n";
    }
    return ret;
}

int do_compare(const Options& o)
{
    if (o.files.size() != 2) { std::cerr << "[SQY]\tcompare needs exactly two stacks\n"; return 1; }
    Stack a, b; std::string err;
    if (!load_stack(o.files[0], o, a, err)) { std::cerr << "[SQY]\tunable to open " << o.files[0] << " (" << err << ")\n"; return 1; }
    if (!load_stack(o.files[1], o, b, err)) { std::cerr << "[SQY]\tunable to open " << o.files[1] << " (" << err << ")\n"; return 1; }
    if (a.shape != b.shape || a.bytes_per_voxel != b.bytes_per_voxel) { std::cout << "stacks differ in shape or type\n"; return 1; }
    size_t ndiff = 0; long double sq = 0;
    const size_t n = a.voxels();
    for (size_t i = 0; i < n; ++i) {
        const long va = a.bytes_per_voxel == 2 ? reinterpret_cast<const uint16_t*>(a.data.data())[i] : (uint8_t)a.data[i];
        const long vb = b.bytes_per_voxel == 2 ? reinterpret_cast<const uint16_t*>(b.data.data())[i] : (uint8_t)b.data[i];
        if (va != vb) { ++ndiff; sq += (long double)(va - vb) * (long double)(va - vb); }
    }
    std::cout << (ndiff ? "stacks differ" : "stacks are equal") << ": " << ndiff << " of " << n << " voxels, mse " << (double)(sq / (long double)(n ? n : 1)) << "\n";
    return ndiff ? 1 : 0;
}

int do_bench(const Options& o)
{
    // columns of the reference's bench verb (verbs/bench.hpp:82-128)
    const std::string d = o.as_csv ? "," : " ";
    if (!o.noheader) std::cout << "id" << d << "shape" << d << "time_mus" << d << "final_bytes" << d << "ingest_bw_mbps" << d << "sizeof_pixel" << d
                               << "n_elements" << d << "filename" << d << "comment\n";
    int ret = 0;
    std::vector<char> blob;
    for (const std::string& file : o.files) {
        Stack st; std::string err;
        if (!load_stack(file, o, st, err)) { std::cerr << "[SQY]\tunable to open " << file << " (" << err << ")\t skipping it\n"; ret = 1; continue; }
        std::string comment = o.comment;
        if (comment.empty()) { std::ostringstream c; c << o.pipeline << "|" << o.nthreads << "threads|" << (long)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count(); comment = c.str(); }
        for (int i = 0; i < o.repetitions; ++i) {
            long written = 0;
            const auto t0 = std::chrono::high_resolution_clock::now();
            const int rc = encode_stack(o, st, blob, written);
            const auto t1 = std::chrono::high_resolution_clock::now();
            if (rc) { std::cerr << "[SQY]\tnative benchmark of compression at iteration " << i << " failed! Exiting.\n"; return 1; }
            const double mus = std::chrono::duration<double, std::micro>(t1 - t0).count();
            std::cout << i << d << st.shape[2] << "x" << st.shape[1] << "x" << st.shape[0] << d << (long)mus << d << written << d
                      << (double)st.data.size() / (1024.0 * 1024.0) / (mus * 1e-6) << d << st.bytes_per_voxel << d << st.voxels() << d
                      << (o.as_csv ? "\"" : "") << file << (o.as_csv ? "\"" : "") << d << (o.as_csv ? "\"" : "") << comment << (o.as_csv ? "\"" : "") << "\n";
        }
    }
    return ret;
}

void usage(const char* me)
{
    std::cout << "usage: " << me << " <-h|optional> <verb> <files|..>\n\n"
              << "available verbs (their description and aliases):\n"
              << "  compress    compress a tiff/raw stack to native sqy format            (compress|enc|encode|comp)\n"
              << "  decompress  decompress a .sqy file to tiff or raw                     (decompress|dec|decode|rec)\n"
              << "  scan        print the header of a .sqy file / statistics of a stack   (scan|info)\n"
              << "  compare     compare two stacks and see if they are equal              (compare|cmp)\n"
              << "  bench       benchmark the compression to native sqy format            (ben|bench)\n\n"
              << "options:\n"
              << "  -p, --pipeline <str>       compression pipeline (default bitswap1->lz4); stages: diff3x3x1, bitswap1, frame_shuffle, raster_reorder, quantiser, lz4\n"
              << "  -o, --output_name <file>   output file (single input only)\n"
              << "  -e, --output_suffix <ext>  output extension (compress: .sqy; decompress: .tif or .raw)\n"
              << "  -n, --nthreads <n>         as in the reference (default 1 = one block-linked LZ4 frame, serial); 0 = all = chunked layout, fast\n"
              << "  -s, --shape ZxYxX          shape of .raw input;   -t, --dtype uint8|uint16\n"
              << "  -r, --repetitions <n>      bench: repetitions (default 10);  -c, --as-csv;  --noheader;  --comment <str>\n"
              << "  -v, --verbose              -h, --help              --version\n";
}

}  // namespace

static int run(int argc, char** argv);
int main(int argc, char** argv)
{
    // (malformed input files must end in a message and an exit code, never in std::terminate)
    try { return run(argc, argv); }
    catch (const std::exception& e) { std::cerr << "[SQY]\t" << e.what() << "\n"; return 1; }
    catch (...) { std::cerr << "[SQY]\tunexpected error\n"; return 1; }
}
static int run(int argc, char** argv)
{
    Options o;
    std::vector<std::string> args(argv + 1, argv + argc);
    auto need = [&](size_t& i) -> std::string { if (i + 1 >= args.size()) { std::cerr << "[SQY]\toption " << args[i] << " needs a value\n"; std::exit(1); } return args[++i]; };
    for (size_t i = 0; i < args.size(); ++i) {
        const std::string& a = args[i];
        if (a == "-h" || a == "--help") o.help = true;
        else if (a == "--version") { int v[3] = {0, 0, 0}; SQY_Version_Triple(v); std::cout << v[0] << "." << v[1] << "." << v[2] << " (" << SQYAMD_Version() << ")\n"; return 0; }
        else if (a == "-v" || a == "--verbose") o.verbose = true;
        else if (a == "-p" || a == "--pipeline") o.pipeline = need(i);
        else if (a == "-o" || a == "--output_name") o.output_name = need(i);
        else if (a == "-e" || a == "--output_suffix") o.output_suffix = need(i);
        else if (a == "-n" || a == "--nthreads") o.nthreads = std::atoi(need(i).c_str());
        else if (a == "-s" || a == "--shape") o.shape = need(i);
        else if (a == "-t" || a == "--dtype") o.dtype = need(i);
        else if (a == "-r" || a == "--repetitions") o.repetitions = std::atoi(need(i).c_str());
        else if (a == "-c" || a == "--as-csv") o.as_csv = true;
        else if (a == "--noheader") o.noheader = true;
        else if (a == "--comment") o.comment = need(i);
        else if (a == "-d" || a == "--dataset_name") (void)need(i);      // HDF5 only: accepted and ignored
        else if (!a.empty() && a[0] == '-') { std::cerr << "[SQY]\tunknown option " << a << "\n"; return 1; }
        else if (o.verb.empty()) o.verb = a;
        else o.files.push_back(a);
    }
    static const std::map<std::string, std::string> alias = {
        {"compress", "compress"}, {"enc", "compress"}, {"encode", "compress"}, {"comp", "compress"},
        {"decompress", "decompress"}, {"dec", "decompress"}, {"decode", "decompress"}, {"rec", "decompress"},
        {"scan", "scan"}, {"info", "scan"}, {"compare", "compare"}, {"cmp", "compare"}, {"bench", "bench"}, {"ben", "bench"}, {"help", "help"}};
    const auto it = alias.find(o.verb);
    if (o.help || o.verb.empty() || it == alias.end() || it->second == "help") {
        usage(argc ? argv[0] : "sqy");
        if (!o.verb.empty() && it == alias.end()) { std::cerr << "[SQY]\tunknown verb " << o.verb << "\n"; return 1; }
        return o.help || (it != alias.end() && it->second == "help") ? 0 : 1;
    }
    if (o.files.empty()) { std::cerr << "[SQY]\tno input files given\n"; return 1; }
    const std::string& verb = it->second;
    if (verb == "compress") return do_compress(o);
    if (verb == "decompress") return do_decompress(o);
    if (verb == "scan") return do_scan(o);
    if (verb == "compare") return do_compare(o);
    return do_bench(o);
}
