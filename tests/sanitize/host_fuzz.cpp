// Host side of libsqeazy_amd under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only, no HIP): the pipeline grammar, the
// configuration strings, the sqy header (pack / unpack of untrusted bytes), base64, the LZ4 block planner, the quantiser's host LUTs
// and file readers, the frame / tile ordering -- everything sqy_pipeline.cpp holds -- driven with valid inputs, systematic
// truncations and seeded random mutations.  Built and run by tests/test_host_sanitizers.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all host_fuzz.cpp ../../sqeazy_amd/csrc/sqy_pipeline.cpp
// Exit code 0 and no sanitizer report = pass.  ("race detection / sanitizers".)
#include "../../sqeazy_amd/csrc/sqy_pipeline.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

using namespace sqy;

static unsigned long g_checks = 0;
#define CHECK(c) do { ++g_checks; if (!(c)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(2); } } while (0)

static std::string mutate(const std::string& s, std::mt19937& rng)
{
    std::string t = s;
    const char alphabet[] = "->(),=<>/verbatim_0123456789abcxyz \t\"\\{}[]:;";
    const int ops = 1 + (int)(rng() % 4);
    for (int i = 0; i < ops; ++i) {
        const unsigned op = rng() % 5;
        const size_t at = t.empty() ? 0 : rng() % (t.size() + 1);
        if (op == 0 && !t.empty()) t.erase(at % t.size(), 1 + rng() % 3);
        else if (op == 1) t.insert(at, 1, alphabet[rng() % (sizeof(alphabet) - 1)]);
        else if (op == 2 && !t.empty()) t[at % t.size()] = (char)(rng() & 0xff);
        else if (op == 3) t.insert(at, t.substr(0, rng() % (t.size() + 1)));
        else if (!t.empty()) t.resize(rng() % (t.size() + 1));
    }
    return t;
}

static void pipelines(std::mt19937& rng)
{
    const char* good[] = {
        "bitswap1->lz4", "lz4", "diff3x3x1->bitswap1->lz4", "frame_shuffle->lz4", "quantiser->bitswap1->lz4", "quantiser->lz4",
        "bitswap1->lz4(accel=1,blocksize_kb=64,framestep_kb=64)", "lz4(n_chunks_of_input=3)", "raster_reorder(tile_size=8)->lz4",
        "zcurve_reorder(tile_size=4)->bitswap1->lz4", "tile_shuffle(tile_size=16)->lz4", "bitshuffle(block_size=4096)->lz4",
        "pass_through->bitswap1->lz4", "quantiser(weighting_function=power_of_1_2)->lz4", "frame_shuffle(frame_chunk_size=4)->lz4",
        "quantiser(decode_lut_string=<verbatim>AAAA</verbatim>)->lz4", "rmbkrd_neighbor5x5x5->lz4", "remove_background->bitswap1->lz4",
        "", "->", "lz4->lz4", "bitswap1->", "(", "lz4(", "lz4()", "lz4(=)", "lz4(a=)", "lz4(=b)", "<verbatim>", "</verbatim>", "a<verbatim>b->c",
    };
    for (const char* g : good) {
        for (int round = 0; round < 400; ++round) {
            const std::string s = round == 0 ? std::string(g) : mutate(g, rng);
            bool ok = false;
            (void)split_outside_verbatim(s, "->", &ok);
            const pairs_t pr = parse_pairs(s);
            for (const auto& p : pr) (void)parse_minors(p.second);
            (void)Pipeline::reference_accepts(s);
            for (int elem = 1; elem <= 2; ++elem) {
                std::string why;
                if (Pipeline::supported(s, elem, &why)) {
                    Pipeline p = Pipeline::from_string(s, elem);
                    const std::string name = p.name();
                    CHECK(Pipeline::supported(name, elem));                       // a pipeline's own name parses again
                    // ... to the same pipeline, unless a value is empty: "key=" parses to the value "key=" (string_parsers.hpp:455-459, kept)
                    if (name.find("=,") == std::string::npos && name.find("=)") == std::string::npos)
                        CHECK(Pipeline::from_string(name, elem).name() == name);
                    else
                        (void)Pipeline::from_string(name, elem).name();
                    p.set_n_threads((int)(rng() % 70) - 3);
                    for (uint64_t nbytes : {0ull, 1ull, 12345ull, 1ull << 20, (1ull << 31) - 2, 1ull << 33})
                        (void)p.max_encoded_size(nbytes, elem);
                    for (const Stage& st : p.stages) { (void)st.config(); (void)st.full_name(); }
                }
            }
        }
    }
}

static void headers(std::mt19937& rng)
{
    const std::vector<std::vector<uint64_t>> shapes = {{1}, {7, 3}, {512, 1024, 1024}, {1, 1, 5}, {2, 3, 4, 5}, {}, {0, 1, 2}, {1ull << 40, 3, 3}};
    const char* names[] = {"bitswap1->lz4", "lz4(accel=1,blocksize_kb=256,framestep_kb=256,n_chunks_of_input=0)", "",
                           "quantiser(decode_lut_string=<verbatim>AAECAwQF\"\\\n</verbatim>)->lz4", "x"};
    for (const auto& shp : shapes)
        for (const char* nm : names)
            for (int elem = 1; elem <= 2; ++elem) {
                const uint64_t payload = rng() % (1u << 30);
                const std::string h = header_pack(elem, elem == 1 && (rng() & 1), shp, nm, payload);
                HeaderInfo hi = header_unpack(h.data(), h.data() + h.size());
                if (hi.valid) {
                    CHECK(hi.size == h.size());
                    CHECK(hi.payload_bytes == payload);
                    CHECK(hi.shape == shp);
                    (void)hi.elem_size();
                }
                // every prefix, and mutations: untrusted bytes
                for (size_t cut = 0; cut <= h.size(); cut += (h.size() > 200 ? 7 : 1)) (void)header_unpack(h.data(), h.data() + cut);
                for (int round = 0; round < 200; ++round) {
                    std::string m = mutate(h, rng);
                    HeaderInfo x = header_unpack(m.data(), m.data() + m.size());
                    if (x.valid) { (void)x.elem_size(); (void)Pipeline::supported(x.pipename, x.elem_size() > 0 ? x.elem_size() : 1); }
                }
            }
    {
        // The one escaping vector the reference holds (tests/test_header_tag_impl.cpp:87, "property_tag_cant_do_this"): what Boost's
        // JSON writer made of a quantiser LUT of raw bytes 00 40 00 80 00 and of the '/' of the closing verbatim tag -- NUL as
        // \u0000, '/' as \/, a byte >= 0x80 and '@' as they are.  The raw pipename goes in, the reference's 71 bytes must come out.
        const std::string raw = std::string("quantiser(decode_lut_string=<verbatim>") + std::string("\0@\0\200\0", 5) + "</verbatim>)";
        const std::string want("quantiser(decode_lut_string=<verbatim>\\u0000@\\u0000\200\\u0000<\\/verbatim>)", 71);
        const std::string h = header_pack(2, false, {1, 2, 3}, raw, 100);
        CHECK(h.find("\"pipename\": \"" + want + "\",\n") != std::string::npos);
        HeaderInfo hi = header_unpack(h.data(), h.data() + h.size());
        CHECK(hi.valid && hi.pipename == raw && hi.payload_bytes == 100);
        // .. and the reader takes the writer's other form of the same text (an unescaped '/') as well
        std::string h2 = h;
        const size_t at = h2.find("<\\/verbatim>");
        CHECK(at != std::string::npos);
        h2.erase(at + 1, 1);
        if (h2.size() % 2) h2.insert(0, 1, ' ');
        HeaderInfo hj = header_unpack(h2.data(), h2.data() + h2.size());
        CHECK(hj.valid && hj.pipename == raw);
    }
    std::vector<char> noise(4096);
    for (int round = 0; round < 300; ++round) {
        for (char& c : noise) c = (char)(rng() & 0xff);
        const size_t n = rng() % noise.size();
        (void)header_unpack(noise.data(), noise.data() + n);
    }
    (void)header_unpack(nullptr, nullptr);
}

static void base64(std::mt19937& rng)
{
    for (int round = 0; round < 2000; ++round) {
        std::vector<unsigned char> raw(rng() % 300);
        for (auto& c : raw) c = (unsigned char)(rng() & 0xff);
        const std::string e = base64_encode(raw.data(), raw.size());
        CHECK(base64_decode(e) == raw);
        (void)base64_decode(mutate(e, rng));
        const std::string v = to_verbatim(raw.data(), raw.size());
        CHECK(raw.empty() ? v.empty() : v.find("<verbatim>") == 0);
    }
}

static void lz4_plans(std::mt19937& rng)
{
    const uint64_t blocks[] = {64u << 10, 256u << 10, 1u << 20, 4u << 20};
    for (int round = 0; round < 3000; ++round) {
        const uint64_t bb = blocks[rng() % 4];
        const uint64_t total = (rng() % 8 == 0) ? rng() % 100 : (uint64_t)rng() % (40u << 20);
        uint64_t step = (rng() % 3 == 0) ? bb : 1 + (uint64_t)rng() % (8u << 20);
        const bool serial = rng() & 1;
        const Lz4Plan p = lz4_plan_blocks(total, step, bb, serial);
        if (!p.ok || total == 0) continue;
        uint64_t at = 0;
        for (const Lz4BlockPlan& b : p.blocks) {
            CHECK(b.start == at);
            CHECK(b.n > 0 && b.n <= bb && b.n <= p.max_block);
            CHECK(b.low_in <= (int64_t)b.start && b.low_dict <= b.low_in + (int64_t)bb + 65536);
            at += b.n;
        }
        CHECK(at == total);
        CHECK(!p.frame_first.empty() && p.frame_first.front() == 0 && p.frame_first.back() == p.blocks.size());
        for (size_t f = 0; f + 1 < p.frame_first.size(); ++f) {
            CHECK(p.frame_first[f] < p.frame_first[f + 1]);
            CHECK(p.blocks[p.frame_first[f]].flags & 1u);
            CHECK(p.blocks[p.frame_first[f + 1] - 1].flags & 2u);
        }
    }
    for (const char* cfg : {"", "accel=1", "accel=-3", "accel=99", "blocksize_kb=0", "blocksize_kb=99999999", "framestep_kb=0", "n_chunks_of_input=4294967295",
                            "blocksize_kb=-1", "framestep_kb=abc", "accel=", "=", ",,,"}) {
        Lz4Params p(cfg);
        (void)p.config(); (void)p.block_bytes();
        for (uint64_t n : {0ull, 1ull, 262144ull, 262145ull, 1ull << 31, 1ull << 40}) { (void)p.bytes_per_chunk(n); (void)p.max_encoded_size(n, (unsigned)(rng() % 9)); }
    }
}

static void quantiser(std::mt19937& rng)
{
    const char* wf[] = {"none", "power_of_1_2", "offset_power_of_3_2", "power_of_2", "power_of", "power_of_1_0", "power_of_1_2_3", "offset", "",
                        "power_of_99999999999999999999_1", "power_of_-1_2", "nonepower_of_1_1"};
    std::vector<uint32_t> histo(65536);
    std::vector<unsigned char> enc(65536);
    uint16_t dec[256];
    for (const char* w : wf) {
        QuantiserWeighting q;
        const bool ok = quantiser_parse_weighting(w, &q);
        for (int kind = 0; kind < 6; ++kind) {
            std::fill(histo.begin(), histo.end(), 0u);
            if (kind == 1) histo[rng() % 65536] = 1u << 30;
            else if (kind == 2) for (auto& h : histo) h = rng() % 1000;
            else if (kind == 3) for (int i = 0; i < 200; ++i) histo[rng() % 65536] += rng() % 100000;
            else if (kind == 4) for (int i = 0; i < 300; ++i) histo[i * 7] = 0xffffffffu;
            else if (kind == 5) histo[65535] = 1, histo[0] = 1;
            if (ok) quantiser_build_luts(histo.data(), histo.size(), enc.data(), dec, q);
        }
    }
    // LUT files: well-formed, short, garbled, missing
    const std::string dir = std::getenv("SQY_SAN_TMP") ? std::getenv("SQY_SAN_TMP") : "/tmp";
    const std::string path = dir + "/sqy_san_lut.txt";
    uint16_t lut[256], back[256];
    for (int i = 0; i < 256; ++i) lut[i] = (uint16_t)(i * 257);
    CHECK(quantiser_lut_to_file(path, lut, 256));
    CHECK(quantiser_lut_from_file(path, back, 256));
    CHECK(std::memcmp(lut, back, sizeof(lut)) == 0);
    for (const char* body : {"", "1\n2\n3\n", "abc\ndef\n", "99999999999999999999\n-5\n", "1 2 3 4 5 6 7 8 9", "\0\0\0\0"}) {
        FILE* f = std::fopen(path.c_str(), "wb");
        CHECK(f != nullptr);
        std::fwrite(body, 1, std::strlen(body), f);
        std::fclose(f);
        (void)quantiser_lut_from_file(path, back, 256);
    }
    std::remove(path.c_str());
    (void)quantiser_lut_from_file(dir + "/does/not/exist", back, 256);
    (void)quantiser_lut_to_file(dir + "/does/not/exist/x", lut, 256);
}

static void orderings(std::mt19937& rng)
{
    for (int round = 0; round < 300; ++round) {
        const size_t Z = 1 + rng() % 200;
        std::vector<float> sums(Z);
        for (auto& s : sums) s = (rng() % 4 == 0) ? 0.0f : (float)(rng() % 100000) * (rng() % 2 ? 1.0f : 0.25f);
        std::vector<uint64_t> map(Z, ~0ull);
        frame_shuffle_order(sums.data(), Z, 1 + rng() % 5000, map.data());
        for (uint64_t m : map) CHECK(m < Z);
        for (int elem = 1; elem <= 2; ++elem) {
            std::fill(map.begin(), map.end(), ~0ull);
            tile_shuffle_order(sums.data(), Z, 1 + rng() % 5000, elem, map.data());
            for (uint64_t m : map) CHECK(m < Z);
        }
    }
    for (uint64_t Z : {1ull, 2ull, 16ull, 17ull, 100ull})
        for (uint64_t Y : {1ull, 16ull, 33ull})
            for (uint64_t X : {1ull, 8ull, 16ull, 100ull})
                for (uint64_t ts : {0ull, 1ull, 2ull, 3ull, 8ull, 16ull, 64ull, 128ull, 256ull, ~0ull}) {
                    for (int elem = 1; elem <= 2; ++elem) (void)raster_geometry_defined(Z, Y, X, ts, elem);
                    (void)zcurve_geometry_defined(Z, Y, X, ts);
                    (void)tile_shuffle_geometry_defined(Z, Y, X, ts);
                }
    for (uint64_t bs : {0ull, 1ull, 7ull, 8ull, 4096ull, ~0ull}) { (void)bitshuffle_block_elems(1, bs); (void)bitshuffle_block_elems(2, bs); }
    for (int n : {-5, 0, 1, 2, 1000000}) (void)clean_number_of_threads(n);
    std::vector<unsigned char> buf(1000);
    for (auto& c : buf) c = (unsigned char)(rng() & 0xff);
    for (size_t n = 0; n <= buf.size(); n += 13) (void)xxh32(buf.data(), n, (uint32_t)rng());
    // xxh32 known answers (the LZ4 frame descriptor's header checksum byte is taken from it)
    CHECK(xxh32(nullptr, 0, 0) == 0x02CC5D05u);
}

int main(int argc, char** argv)
{
    std::mt19937 rng(argc > 1 ? (unsigned)std::atoi(argv[1]) : 20261004u);
    pipelines(rng);
    headers(rng);
    base64(rng);
    lz4_plans(rng);
    quantiser(rng);
    orderings(rng);
    std::printf("host_fuzz ok: %lu checks\n", g_checks);
    return 0;
}
