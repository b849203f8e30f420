// sqy_capi.cpp -- the C-ABI of libsqeazy_amd.so (include/sqeazy_amd.h) and the stage sequencing on the GPU.
//
// Mirrors src/cpp/src/sqeazy.cpp:16-335 (entry points) and dynamic_pipeline.hpp:560-690 (encode:
// header, head filters, sink, tail filters, header rewrite), with every stage a HIP kernel launch on
// device-resident ping-pong buffers instead of an OpenMP loop over host memory.
#include "../../include/sqeazy_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sqy_kernels.h"
#include "sqy_pipeline.hpp"

namespace {


using sqy::Pipeline;
using sqy::Stage;
using sqy::StageKind;

#define SQY_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            std::fprintf(stderr, "[sqeazy]\t HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)

// ---- run-time options ------------------------------------------------------------------------------
// Measurement / test switches.  The environment is read ONCE, when the library is loaded (getenv in the middle of a call races with
// setenv from other host threads); afterwards they change only through SQYAMD_Set_Option (atomics).  None of them changes a byte of
// any result.
long env_flag(const char* name) { const char* v = std::getenv(name); return v && *v && std::strcmp(v, "0") != 0 ? 1 : 0; }
long env_number(const char* name, long dflt, long lo, long hi)
{
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    char* end = nullptr;
    const long long x = std::strtoll(v, &end, 10);
    if (*end != '\0' || x < lo || x > hi) {
        std::fprintf(stderr, "[sqeazy]\t %s=%s is not a number in [%ld, %ld]: ignored\n", name, v, lo, hi);
        return dflt;
    }
    return (long)x;
}
constexpr long kWarmupMax = 1l << 30;
struct Options {
    std::atomic<long> transpose_chain;                  // the bit-plane transposes of calls in flight on LIBRARY-OWNED streams run one after the other
    std::atomic<long> transpose_chain_caller_streams;   // .. on streams the callers bring as well (opt-in: couples those streams, see the bitswap1 stage)
    std::atomic<long> block_parallel;                   // block-linked frames: block-parallel encode and decode (0: the one-wavefront walk)
    std::atomic<long> block_parallel_warmup;            // bytes parsed in front of a block to guess its table
    std::atomic<long> block_parallel_stats;             // print the blocks that failed the table check
    std::atomic<long> tail_scan;                        // serial-layout decode: the walk over the tails as a scan
    std::atomic<long> decode_two_waves;                 // chunked-layout decode: two wavefronts per frame (one parses, one copies)
    std::atomic<long> noise_digest;                     // frames in place: the transpose leaves the noise digest, the parse proves noise chunks empty from it
    std::atomic<long> transpose_blocks_per_cu;          // frames in place: workgroups of the transposer's grid per CU
    std::atomic<long> stored_tail_index;                // decode, chunked layout: the stored frames at the stream's end are found where they must start, not by the scan
    Options()
        : transpose_chain(env_flag("SQY_NO_TRANSPOSE_CHAIN") ? 0 : 1), transpose_chain_caller_streams(env_flag("SQY_TRANSPOSE_CHAIN_CALLER_STREAMS")),
          block_parallel(env_flag("SQY_NO_BLOCK_PARALLEL") ? 0 : 1), block_parallel_warmup(env_number("SQY_BLOCK_PARALLEL_WARMUP", 65536, 0, kWarmupMax)),
          block_parallel_stats(env_flag("SQY_BLOCK_PARALLEL_STATS")), tail_scan(env_flag("SQY_NO_TAIL_SCAN") ? 0 : 1),
          decode_two_waves(env_flag("SQY_NO_DECODE_TWO_WAVES") ? 0 : 1), noise_digest(env_flag("SQY_NO_NOISE_DIGEST") ? 0 : 1),
          transpose_blocks_per_cu(env_number("SQY_TRANSPOSE_BLOCKS_PER_CU", 32, 1, 64)), stored_tail_index(env_flag("SQY_NO_STORED_TAIL_INDEX") ? 0 : 1) { sqy::set_bitswap1_blocks_per_cu(transpose_blocks_per_cu.load()); }
    std::atomic<long>* find(const char* name)
    {
        if (!name) return nullptr;
        if (!std::strcmp(name, "transpose_chain")) return &transpose_chain;
        if (!std::strcmp(name, "transpose_chain_caller_streams")) return &transpose_chain_caller_streams;
        if (!std::strcmp(name, "block_parallel")) return &block_parallel;
        if (!std::strcmp(name, "block_parallel_warmup")) return &block_parallel_warmup;
        if (!std::strcmp(name, "block_parallel_stats")) return &block_parallel_stats;
        if (!std::strcmp(name, "tail_scan")) return &tail_scan;
        if (!std::strcmp(name, "decode_two_waves")) return &decode_two_waves;
        if (!std::strcmp(name, "noise_digest")) return &noise_digest;
        if (!std::strcmp(name, "transpose_blocks_per_cu")) return &transpose_blocks_per_cu;
        if (!std::strcmp(name, "stored_tail_index")) return &stored_tail_index;
        return nullptr;
    }
};
Options g_opt;

// ---- per-kernel timing -------------------------------------------------------------------------
struct ProfEntry { std::string name; double ms = 0; long launches = 0; };
struct PendingEvent { const char* name; hipEvent_t a, b; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<ProfEntry> g_prof;

// Timing events are pooled (created once, reused by every later call): creating and destroying two events per kernel cost the
// calls of a profiled run tens of microseconds of host time each.
std::mutex g_evpool_mu;
std::vector<hipEvent_t> g_evpool[16];           // per device (an event belongs to the device it was created on)
int ev_dev()
{
    int d = 0;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 16) ? d : -1;
}
hipEvent_t ev_take()
{
    const int d = ev_dev();
    if (d >= 0) {
        std::lock_guard<std::mutex> lock(g_evpool_mu);
        if (!g_evpool[d].empty()) { hipEvent_t e = g_evpool[d].back(); g_evpool[d].pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
void ev_give(hipEvent_t e)
{
    if (!e) return;
    const int d = ev_dev();
    if (d < 0) { hipEventDestroy(e); return; }
    std::lock_guard<std::mutex> lock(g_evpool_mu);
    g_evpool[d].push_back(e);
}

struct ProfScope {
    hipStream_t s;
    PendingEvent ev{};
    std::vector<PendingEvent>* sink;
    bool on;
    ProfScope(const char* name, hipStream_t stream, std::vector<PendingEvent>* pending) : s(stream), sink(pending), on(g_prof_on.load())
    {
        if (!on) return;
        ev.name = name;
        ev.a = ev_take(); ev.b = ev_take();
        if (!ev.a || !ev.b) { ev_give(ev.a); ev_give(ev.b); on = false; return; }
        hipEventRecord(ev.a, s);
    }
    ~ProfScope()
    {
        if (!on) return;
        hipEventRecord(ev.b, s);
        sink->push_back(ev);
    }
};

void prof_collect(std::vector<PendingEvent>& pending)
{
    for (PendingEvent& p : pending) {
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            std::lock_guard<std::mutex> lock(g_prof_mu);
            size_t i = 0;
            for (; i < g_prof.size(); ++i) if (g_prof[i].name == p.name) break;
            if (i == g_prof.size()) g_prof.push_back(ProfEntry{p.name, 0, 0});
            g_prof[i].ms += ms;
            g_prof[i].launches += 1;
        }
        ev_give(p.a);
        ev_give(p.b);
    }
    pending.clear();
}

// ---- HBM workspace (grow-only; one per leased context) ------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // quiet: an OPTIONAL buffer (the caller has a path that needs none and says which) -- no message from here
    int ensure(size_t bytes, bool quiet = false)
    {
        if (bytes <= cap) return 0;
        if (p) { hipFree(p); p = nullptr; cap = 0; }
        const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        if (hipMalloc(&p, want) != hipSuccess) {
            if (!quiet) std::fprintf(stderr, "[sqeazy]\t unable to allocate %zu bytes of HBM workspace\n", want);
            (void)hipGetLastError();          // (not left behind for the launch checks of a caller that carries on without this buffer)
            p = nullptr;
            return 1;
        }
        cap = want;
        return 0;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
};

struct Workspace {
    DevBuf ping, pong, lz4_scratch, csize, frame_off, io_src, io_dst, small, plan, dedupe;
    DevBuf spec;              // block-linked frames parsed block-parallel: per block the table it started from and the one it left, the walk lists
    DevBuf diff_side;         // diff3x3x1 in front of a 16-bit bitswap1: the columns the stage can touch (outside the ping/pong rotation)
    DevBuf digest;            // frames in place: the noise digest the transpose leaves for the LZ4 parse (19 KB per 256 KiB chunk)
    void* pinned = nullptr;   // 4 KiB of pinned host memory for small read-backs
    void release_buffers()
    {
        ping.release(); pong.release(); lz4_scratch.release(); csize.release(); frame_off.release();
        io_src.release(); io_dst.release(); small.release(); plan.release(); dedupe.release(); diff_side.release(); spec.release(); digest.release();
    }
};

// A context = one HBM workspace + one private stream.  Concurrent C-ABI calls (the reference is re-entrant:
// every call builds its own pipeline object, src/sqeazy.cpp:123) each lease their own context, so two host
// threads encoding different volumes overlap on the GPU instead of queueing behind a lock.
// Host <-> HBM transfers of the reference-protocol entry points (caller memory is pageable).  A plain hipMemcpy from
// pageable memory runs at 5-6 GB/s; here kLanes host threads each own two pinned staging buffers and a copy stream:
// memcpy user -> pinned slice k while the DMA of slice k-1 is in flight, slices dealt round-robin to the lanes.
struct Stager {
    static constexpr int kLanes = 4;
    static constexpr size_t kSlice = 8u << 20;
    char* pin[kLanes][2] = {};
    hipStream_t st[kLanes] = {};
    hipEvent_t ev[kLanes][2] = {};
    bool ready = false;

    bool init()
    {
        if (ready) return true;
        for (int l = 0; l < kLanes; ++l) {
            if (hipStreamCreateWithFlags(&st[l], hipStreamNonBlocking) != hipSuccess) return false;
            for (int b = 0; b < 2; ++b) {
                if (hipHostMalloc((void**)&pin[l][b], kSlice, hipHostMallocDefault) != hipSuccess) return false;
                if (hipEventCreateWithFlags(&ev[l][b], hipEventDisableTiming) != hipSuccess) return false;
            }
        }
        ready = true;
        return true;
    }
    // to_device: dev <- host;  else host <- dev.  Blocks until the bytes have arrived.  `after` (optional) is a stream
    // whose work must be complete before device memory is read (D2H of freshly computed data).
    bool copy(void* dev, void* host, size_t bytes, bool to_device, int device_id)
    {
        if (bytes == 0) return true;
        if (!init()) return false;
        const size_t nslices = (bytes + kSlice - 1) / kSlice;
        std::atomic<bool> ok(true);
        auto lane_fn = [&](int l) {
            if (hipSetDevice(device_id) != hipSuccess) { ok = false; return; }
            int b = 0;
            size_t pending_off[2] = {0, 0}, pending_len[2] = {0, 0};
            for (size_t k = (size_t)l; k < nslices && ok; k += kLanes, b ^= 1) {
                const size_t off = k * kSlice, len = std::min(kSlice, bytes - off);
                // the buffer's previous transfer must be over before it is reused
                if (hipEventSynchronize(ev[l][b]) != hipSuccess) { ok = false; break; }
                if (to_device) {
                    std::memcpy(pin[l][b], static_cast<char*>(host) + off, len);
                    if (hipMemcpyAsync(static_cast<char*>(dev) + off, pin[l][b], len, hipMemcpyHostToDevice, st[l]) != hipSuccess) { ok = false; break; }
                } else {
                    if (pending_len[b]) std::memcpy(static_cast<char*>(host) + pending_off[b], pin[l][b], pending_len[b]);
                    if (hipMemcpyAsync(pin[l][b], static_cast<char*>(dev) + off, len, hipMemcpyDeviceToHost, st[l]) != hipSuccess) { ok = false; break; }
                    pending_off[b] = off; pending_len[b] = len;
                }
                if (hipEventRecord(ev[l][b], st[l]) != hipSuccess) { ok = false; break; }
            }
            if (hipStreamSynchronize(st[l]) != hipSuccess) ok = false;
            if (!to_device && ok)
                for (int bb = 0; bb < 2; ++bb)
                    if (pending_len[bb]) std::memcpy(static_cast<char*>(host) + pending_off[bb], pin[l][bb], pending_len[bb]);
        };
        const int lanes = (int)std::min<size_t>(kLanes, nslices);
        std::vector<std::thread> th;
        for (int l = 1; l < lanes; ++l) th.emplace_back(lane_fn, l);
        lane_fn(0);
        for (auto& t : th) t.join();
        return ok;
    }
};

struct Context {
    Workspace ws;
    // Streams are created on first use, not with the context: HIP deals streams to a handful of hardware queues in creation
    // order, and a caller that brings its own streams (one per host thread) should not find two of them behind the same queue
    // because this library created streams of its own in between -- their kernels would then never overlap (bench, three
    // callers: 750 instead of 1120 GB/s in about every other process).
    hipStream_t stream = nullptr;       // used when the caller brings no stream (host-pointer entry points)
    hipStream_t side = nullptr;         // decode: stored frames are copied here while the compressed ones are decoded
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t t_done = nullptr;        // recorded behind this call's bit-plane transpose (the transposes of calls in flight run one after the other)
    hipStream_t own_stream()
    {
        if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) stream = nullptr;
        return stream;
    }
    bool ensure_side()
    {
        if (side && fork && join) return true;
        if (!side && hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { side = nullptr; return false; }
        if (!fork && hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) { fork = nullptr; return false; }
        if (!join && hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) { join = nullptr; return false; }
        return true;
    }
    std::vector<PendingEvent> pending;
    Stager stager;
    bool busy = false;
};

constexpr int kMaxDev = 16;
constexpr size_t kMaxCtxPerDev = 8;
// the chain of the bit-plane transposes of the calls in flight on one device (see the bitswap1 stage): the event behind the last
// transpose launched, and when that was
std::mutex g_tchain_mu[kMaxDev];
hipEvent_t g_tchain_last[kMaxDev] = {};
std::chrono::steady_clock::time_point g_tchain_when[kMaxDev];
std::mutex g_pool_mu;
std::condition_variable g_pool_cv;
std::vector<std::unique_ptr<Context>> g_pool[kMaxDev];

struct ContextLease {
    Context* ctx = nullptr;
    ContextLease()
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return;
        std::unique_lock<std::mutex> lock(g_pool_mu);
        for (;;) {
            for (auto& c : g_pool[dev]) if (!c->busy) { ctx = c.get(); break; }
            if (ctx) break;
            if (g_pool[dev].size() < kMaxCtxPerDev) {
                std::unique_ptr<Context> c(new Context());
                if (hipHostMalloc(&c->ws.pinned, 4096, hipHostMallocDefault) != hipSuccess) return;
                ctx = c.get();
                g_pool[dev].push_back(std::move(c));
                break;
            }
            g_pool_cv.wait(lock);
        }
        ctx->busy = true;
    }
    ~ContextLease()
    {
        if (!ctx) return;
        {
            std::lock_guard<std::mutex> lock(g_pool_mu);
            ctx->busy = false;
        }
        g_pool_cv.notify_one();
    }
    ContextLease(const ContextLease&) = delete;
    ContextLease& operator=(const ContextLease&) = delete;
};

// Every exit of an encode / decode -- the early error returns included -- leaves the stream idle before the context goes
// back to the pool: kernels and async copies still in flight would otherwise read host vectors that are being destroyed
// and HBM buffers the next call (on another stream) may reuse or free.  Timing events that nobody harvested are dropped.
struct DrainOnExit {
    hipStream_t s;
    std::vector<PendingEvent>* pending;
    hipStream_t side = nullptr;         // the context's side stream (decode)
    ~DrainOnExit()
    {
        (void)hipStreamSynchronize(s);
        if (side) (void)hipStreamSynchronize(side);
        if (!pending->empty()) {
            if (g_prof_on.load()) prof_collect(*pending);
            else { for (PendingEvent& p : *pending) { ev_give(p.a); ev_give(p.b); } pending->clear(); }
        }
    }
};

bool device_present()
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

// ---- encode --------------------------------------------------------------------------------------
// The body of dynamic_pipeline::encode (dynamic_pipeline.hpp:560-616) on device buffers.
// dstoffset == nullptr: the blob starts at d_dst.  Otherwise it may start anywhere inside [d_dst, d_dst + dst_capacity) and
// *dstoffset says where ("frames in place": a 16-bit bitswap1 in front of lz4 writes the plane stream straight into d_dst as
// the bodies of the LZ4 frames it will become; the stored frames that end the payload -- the noise planes, 98 % of the bytes of
// a microscopy stack -- then never move, only the compressed frames in front are gathered up against them).
// Where every `every`-th LZ4 frame of the payload starts (chunked layout): what a caller needs to re-order byte ranges of slab
// blobs into one blob (single-blob mode of the multi-GPU path) without walking the frames itself.
struct FrameQuery {
    int every = 0;              // 0: not asked for
    long* offsets = nullptr;    // out, relative to the blob start: frames 0, every, 2 every, ..; then the blob length
    int max_entries = 0;
    int count = 0;              // out: frames listed (the end entry comes on top)
};

int encode_on_device(Context& cx, const char* pipeline_c, const void* d_src, const long* shape, unsigned rank, int elem_size,
                     void* d_dst, uint64_t dst_capacity, long* dstlength, int nthreads, hipStream_t stream, long* dstoffset = nullptr,
                     FrameQuery* fq = nullptr)
{
    if (!pipeline_c || !d_src || !shape || !d_dst || !dstlength) return 1;
    if (dstoffset) *dstoffset = 0;
    const std::string pipeline(pipeline_c);
    std::string why;
    if (!Pipeline::supported(pipeline, elem_size, &why)) {
        if (Pipeline::reference_accepts(pipeline))
            std::fprintf(stderr, "[sqeazy]\t pipeline %s: %s\n", pipeline.c_str(), why.c_str());
        return 1;
    }
    Pipeline pipe = Pipeline::from_string(pipeline, elem_size);
    if (pipe.stages.empty()) {
        std::fprintf(stderr, "[sqeazy]\t received %spipeline of size 0, cannot encode buffer\n", pipe.name().c_str());
        return 1;
    }
    pipe.set_n_threads(nthreads);

    std::vector<uint64_t> dims(shape, shape + rank);
    uint64_t len = 1;
    for (uint64_t d : dims) {
        if ((long)d <= 0) { std::fprintf(stderr, "[sqeazy]\t non-positive extent in shape\n"); return 1; }
        len *= d;
        if (len >= ((uint64_t)1 << 31)) {
            // the reference multiplies the extents into an `int` (dynamic_pipeline.hpp:565): one call is < 2^31 voxels
            std::fprintf(stderr, "[sqeazy]\t %llu+ voxels in one call overflow the reference's int voxel count; encode z-slabs\n",
                         (unsigned long long)len);
            return 1;
        }
    }
    const uint64_t raw_bytes = len * (uint64_t)elem_size;

    Workspace* ws = &cx.ws;
    std::vector<PendingEvent>* pend = &cx.pending;
    DrainOnExit drain{stream, pend, cx.side};

    // ---- walk the stages ----
    const uint8_t* cur = static_cast<const uint8_t*>(d_src);
    int cur_elem = elem_size;            // bytes per element of the stream between stages
    uint64_t cur_len = len;              // elements
    bool use_ping = true;
    auto next_buf = [&](size_t bytes) -> uint8_t* {
        DevBuf& b = use_ping ? ws->ping : ws->pong;
        use_ping = !use_ping;
        if (b.ensure(bytes)) return nullptr;
        return static_cast<uint8_t*>(b.p);
    };

    // the shape a 3-D stage sees: the volume's -- or, behind a sink that did not write one byte per voxel, {1, 1, bytes}
    // (dynamic_pipeline.hpp:658-666: the tail chain's "sinked_shape")
    auto stage_shape = [&](size_t si, uint64_t& Z, uint64_t& Y, uint64_t& X) {
        const bool tail = pipe.sink_index >= 0 && (int)si > pipe.sink_index;
        const bool flat = tail && cur_len * (uint64_t)cur_elem != len;
        Z = flat ? 1 : dims[0]; Y = flat ? 1 : dims[1]; X = flat ? cur_len : dims[2];
    };

    uint64_t payload_bytes = 0;
    bool payload_is_lz4 = false;
    const sqy::Lz4Params* lz4p = nullptr;
    uint64_t lz4_total = 0, lz4_nchunks = 0, lz4_chunk = 0, lz4_stride = 0;
    const uint64_t* lz4_frame_map = nullptr;     // frame_shuffle directly in front of lz4: frames are read through the map
    uint64_t lz4_frame_bytes = 0;
    const uint32_t* lz4_piece_hash = nullptr;    // left by a 16-bit bitswap1 directly in front of lz4: hashes of the 1 KiB pieces of the plane stream
    const uint32_t* lz4_dup_of = nullptr;        // chunks that are byte-identical to an earlier chunk share its frame
    const sqy::Lz4Block* lz4_blocks = nullptr;   // block-linked frames (nthreads == 1, or chunks of several LZ4 blocks): the block list in HBM
    // frames in place: chunk k of the plane stream sits at d_dst + inplace_t0 + 11 + k * lz4_in_stride
    bool lz4_inplace = false;
    uint64_t lz4_in_stride = 0, inplace_t0 = 0;
    // diff3x3x1 directly in front of a 16-bit bitswap1: only the columns the stage can touch are computed (compact side buffer)
    const uint16_t* bsw_side = nullptr;
    uint32_t bsw_side_w = 0, bsw_side_X = 0;
    uint64_t* lz4_tail_info = nullptr;
    sqy::Lz4DedupeArgs lz4_dedupe_args;          // frames in place: the duplicate decision per chunk is made inside the parse kernel
    bool fused_dedupe = false;
    bool dedupe_cleared = false;                 // the duplicate search's table and the dense list's counter were zeroed in front of the transpose
    bool inplace_done = false;                   // frames in place, finished on the device: where the blob is
    uint64_t inplace_blob_at = 0, inplace_blob_bytes = 0, inplace_hdr_bytes = 0;
    uint64_t* lz4_holes = nullptr;               // frames in place: which 1 KiB pieces of the plane stream the transpose left unwritten (all zero)
    uint32_t* lz4_digest = nullptr;              // frames in place: the noise digest (sqy_kernels.h: launch_bitswap1_u16), lz4_digest_stride words per chunk
    uint32_t lz4_digest_stride = 0;
    static_assert(sizeof(sqy::Lz4Block) == sizeof(sqy::Lz4BlockPlan) && sizeof(sqy::Lz4Block) == 32, "plan entries are read by the kernels as they are");

    size_t skip_stage = ~(size_t)0;             // a stage that the stage in front of it has already done (quantiser + bitswap1 in one pass)
    for (size_t si = 0; si < pipe.stages.size(); ++si) {
        if (si == skip_stage) continue;

        Stage& st = pipe.stages[si];
        switch (st.kind) {
            case StageKind::bitswap1: {
                // lz4 right behind: leave piece hashes for its duplicate-chunk detection (bit planes of small values repeat)
                uint32_t* ph = nullptr;
                uint64_t gap_chunk = 0;
                if (cur_elem == 2 && si + 1 < pipe.stages.size() && pipe.stages[si + 1].kind == StageKind::lz4) {
                    const uint64_t words = sqy::bitswap1_piece_hash_words(cur, cur, cur_len);         // (0 unless whole tiles, 16-byte aligned input)
                    const uint64_t total = cur_len * 2;
                    const uint64_t chunk = pipe.stages[si + 1].lz4.bytes_per_chunk(total);
                    const bool chunked = chunk <= pipe.stages[si + 1].lz4.block_bytes() && !(pipe.nthreads == 1 && total > chunk);
                    if (words && chunked && chunk % 1024 == 0 && total > chunk) {
                        const uint64_t nch = (total + chunk - 1) / chunk;
                        const uint64_t ph_bytes = (words * 4 + 63) & ~(uint64_t)63;
                        if (ws->dedupe.ensure(ph_bytes + sqy::lz4_dedupe_work_bytes(nch) + ((nch * 4 + 7) & ~(uint64_t)7) + sqy::lz4_holes_map_bytes(nch, (uint32_t)chunk))) return 1;
                        ph = static_cast<uint32_t*>(ws->dedupe.p);
                        lz4_piece_hash = ph;
                        // frames in place: lz4 is the last stage, chunks a power of two, the caller takes the blob where it ends up,
                        // and the destination holds frame headers in front of and end marks behind every chunk
                        if (dstoffset && si + 2 == pipe.stages.size() && (chunk & (chunk - 1)) == 0) {
                            // room in front for the sqy header (its length depends on the payload size: take the longest)
                            const uint64_t hdr_max = sqy::header_pack(elem_size, false, dims, pipe.name(), (uint64_t)INT_MAX).size() + 2;
                            uint64_t t0 = hdr_max;
                            while ((reinterpret_cast<uintptr_t>(d_dst) + t0 + 11) & 15) ++t0;          // body of chunk 0 on a 16-byte boundary
                            if (t0 + nch * (chunk + 15) <= dst_capacity) {
                                gap_chunk = chunk;
                                inplace_t0 = t0;
                                lz4_in_stride = chunk + 15;
                                lz4_inplace = true;
                                // (one small kernel in front of the transpose instead of three fill dispatches between the kernels behind it)
                                if (ws->plan.ensure((nch + 1) * sizeof(uint32_t))) return 1;
                                SQY_HIP(sqy::launch_lz4_dedupe_clear(static_cast<uint8_t*>(ws->dedupe.p) + ph_bytes, nch, static_cast<uint32_t*>(ws->plan.p), stream));
                                dedupe_cleared = true;
                                // the noise digest (round 6): every plane segment a whole number of chunks, liblz4's plain search behind it
                                const uint32_t dstride = sqy::lz4_noise_digest_stride((uint32_t)chunk);
                                if (g_opt.noise_digest.load() && dstride && (cur_len / 8) % chunk == 0 && pipe.stages[si + 1].lz4.accel >= 0 &&
                                    !ws->digest.ensure(nch * (uint64_t)dstride * sizeof(uint32_t), true)) {
                                    lz4_digest = static_cast<uint32_t*>(ws->digest.p);
                                    lz4_digest_stride = dstride;
                                }
                            }
                        }
                    }
                }
                uint8_t* out = gap_chunk ? static_cast<uint8_t*>(d_dst) + inplace_t0 + 11 : next_buf(cur_len * cur_elem);
                if (!out) return 1;
                if (!gap_chunk && ph && (reinterpret_cast<uintptr_t>(out) & 15)) { ph = nullptr; lz4_piece_hash = nullptr; }
                // The bit-plane transposes of the calls in flight on one device run one after the other (round 4): a stream waits for the
                // transpose of the call in front before it starts its own.  Two HBM-bound kernels side by side each run at half speed
                // and end together; chained, the first call's parse starts a whole transpose earlier (bench, four calls in flight:
                // +3 %; also chaining the duplicate search behind it: -12 %, measured and not kept).  Only a transpose launched within
                // the last few milliseconds is waited for.  The chain is an edge between streams: by default only streams this library
                // owns (the host-pointer entry points, the Slabs workers) are chained -- a stream the CALLER brings may carry work this
                // library knows nothing about (a backlog, a host function that waits for another of the caller's threads), and a hidden
                // wait on it would couple calls that are documented as independent (round-4 advice).  A caller whose streams carry
                // nothing but these calls opts in: SQYAMD_Set_Option("transpose_chain_caller_streams", 1) (bench.py does, and says so).
                // "transpose_chain" = 0 (or SQY_NO_TRANSPOSE_CHAIN=1 when the library is loaded) switches the chain off altogether.
                const bool owned = stream != nullptr && stream == cx.stream;
                int devid = 0;
                const bool chain = g_opt.transpose_chain.load() && (owned || g_opt.transpose_chain_caller_streams.load()) && gap_chunk &&
                                   hipGetDevice(&devid) == hipSuccess && devid >= 0 && devid < kMaxDev;
                std::unique_lock<std::mutex> tlock;
                if (chain) {
                    if (!cx.t_done && hipEventCreateWithFlags(&cx.t_done, hipEventDisableTiming) != hipSuccess) return 1;
                    tlock = std::unique_lock<std::mutex>(g_tchain_mu[devid]);
                    const auto now = std::chrono::steady_clock::now();
                    if (g_tchain_last[devid] && g_tchain_last[devid] != cx.t_done && now - g_tchain_when[devid] < std::chrono::milliseconds(5))
                        SQY_HIP(hipStreamWaitEvent(stream, g_tchain_last[devid], 0));
                    g_tchain_when[devid] = now;
                }
                {
                ProfScope ps(cur_elem == 2 ? "bitswap1_u16" : "bitswap1_u8", stream, pend);
                if (cur_elem == 2)
                    SQY_HIP(sqy::launch_bitswap1_u16(reinterpret_cast<const uint16_t*>(cur), reinterpret_cast<uint16_t*>(out), cur_len, stream, ph,
                                                     (uint32_t)gap_chunk, bsw_side, bsw_side_w, bsw_side_X, gap_chunk ? lz4_digest : nullptr,
                                                     lz4_digest_stride));
                else
                    SQY_HIP(sqy::launch_bitswap1_u8(cur, out, cur_len, stream));
                }
                if (chain) {
                    SQY_HIP(hipEventRecord(cx.t_done, stream));
                    g_tchain_last[devid] = cx.t_done;
                    tlock.unlock();
                }
                bsw_side = nullptr; bsw_side_w = 0; bsw_side_X = 0;       // (consumed: a later bitswap1 of the pipeline reads its plain input)
                cur = out;
                break;
            }
            case StageKind::raster_reorder: {
                if (dims.size() != 3) {
                    std::fprintf(stderr, "[sqeazy::detail::reorder::encode] received non-3D shape which is currently unsupported!\n");
                    return 1;
                }
                const uint64_t ts = (uint64_t)std::atoi(st.cfg["tile_size"].c_str());
                // (round 5) as a tail filter the stream is the sink's `char` output: the volume's shape when that is one byte per voxel,
                // else {1, 1, bytes} (dynamic_pipeline.hpp:658-666; sqeazy_pipelines.hpp:64-77 lists the stage for the tail chain)
                uint64_t Z, Y, X;
                stage_shape(si, Z, Y, X);
                if (!sqy::raster_geometry_defined(Z, Y, X, ts, cur_elem)) {
                    std::fprintf(stderr, "[sqeazy]\t raster_reorder: the reference's result is undefined for shape %llux%llux%llu at tile_size=%llu "
                                         "(remainder in some dimensions only, or a tile wider than one 16-byte block); refused\n",
                                 (unsigned long long)Z, (unsigned long long)Y, (unsigned long long)X, (unsigned long long)ts);
                    return 1;
                }
                uint8_t* out = next_buf(cur_len * cur_elem);
                if (!out) return 1;
                ProfScope ps("raster_reorder", stream, pend);
                SQY_HIP(sqy::launch_raster_reorder(cur, out, Z, Y, X, ts, cur_elem, false, stream));
                cur = out;
                break;
            }
            case StageKind::pass_through: {
                // pass_through_scheme_impl.hpp:66-79: the sink that only re-types the stream to bytes
                cur_len *= (uint64_t)cur_elem;
                cur_elem = 1;
                break;
            }
            case StageKind::zcurve_reorder: {
                if (dims.size() != 3) {
                    std::fprintf(stderr, "[sqeazy::detail::zcurve::encode] received non-3D shape which is currently unsupported!\n");
                    return 1;
                }
                auto t = st.cfg.find("tile_size");
                const uint64_t ts = t != st.cfg.end() ? (uint64_t)std::atoi(t->second.c_str()) : 2;
                uint64_t Z, Y, X;
                stage_shape(si, Z, Y, X);                                  // (tail filter: the sink's char stream, see raster_reorder)
                if (!sqy::zcurve_geometry_defined(Z, Y, X, ts)) {
                    std::fprintf(stderr, "[sqeazy]\t zcurve_reorder: the reference's result is undefined for shape %llux%llux%llu at tile_size=%llu "
                                         "(tile sizes other than 2..128 powers of two, or a tile that does not divide a power-of-two shape); refused\n",
                                 (unsigned long long)Z, (unsigned long long)Y, (unsigned long long)X, (unsigned long long)ts);
                    return 1;
                }
                uint8_t* out = next_buf(cur_len * cur_elem);
                if (!out) return 1;
                ProfScope ps("zcurve_reorder", stream, pend);
                // (inside a tile the reference's morton_at_ct<log2(tile)> code is row-major: the tiled raster kernel is the stage)
                SQY_HIP(sqy::launch_raster_reorder(cur, out, Z, Y, X, ts, cur_elem, false, stream));
                cur = out;
                break;
            }
            case StageKind::bitshuffle: {
                auto b = st.cfg.find("block_size");
                const uint64_t be = sqy::bitshuffle_block_elems(cur_elem, b != st.cfg.end() ? (uint64_t)std::atoi(b->second.c_str()) : 0);
                if (!be) { std::fprintf(stderr, "[sqeazy]\t bitshuffle: block_size must be a multiple of 8\n"); return 1; }
                uint8_t* out = next_buf(cur_len * cur_elem);
                if (!out) return 1;
                ProfScope ps("bitshuffle", stream, pend);
                SQY_HIP(sqy::launch_bitshuffle(cur, out, cur_len, cur_elem, be, false, stream));
                cur = out;
                break;
            }
            case StageKind::tile_shuffle: {
                if (dims.size() != 3) {
                    std::fprintf(stderr, "[sqeazy::detail::tile_shuffle::encode] received non-3D shape which is currently unsupported!\n");
                    return 1;
                }
                auto t = st.cfg.find("tile_size");
                const uint64_t ts = t != st.cfg.end() ? (uint64_t)std::atoi(t->second.c_str()) : 32;
                // (tail filter: tile_shuffle_scheme<char> on the sink's stream -- the tile sums add SIGNED bytes and the metric is a char)
                const bool tail = pipe.sink_index >= 0 && (int)si > pipe.sink_index;
                uint64_t Z, Y, X;
                stage_shape(si, Z, Y, X);
                if (!sqy::tile_shuffle_geometry_defined(Z, Y, X, ts)) {
                    std::fprintf(stderr, "[sqeazy]\t tile_shuffle: shape %llux%llux%llu is not a whole multiple of tile_size=%llu; the reference's remainder "
                                         "path (P^2 median over tiles read past their end, thread-timing dependent map) is not reproduced; refused\n",
                                 (unsigned long long)Z, (unsigned long long)Y, (unsigned long long)X, (unsigned long long)ts);
                    return 1;
                }
                const uint64_t per_tile = ts * ts * ts, ntiles = cur_len / per_tile, tile_bytes = per_tile * (uint64_t)cur_elem;
                // 1. tiles made contiguous (tile-major copy), 2. their sequential binary32 sums, 3. order on the host, 4. tiles appended in that order
                uint8_t* tiled = next_buf(cur_len * cur_elem);
                if (!tiled) return 1;
                {
                    ProfScope ps("tile_gather", stream, pend);
                    SQY_HIP(sqy::launch_raster_reorder(cur, tiled, Z, Y, X, ts, cur_elem, false, stream));
                }
                if (ws->small.ensure(std::max<uint64_t>(ntiles * 16, 4096))) return 1;
                float* d_sums = static_cast<float*>(ws->small.p);
                uint64_t* d_map = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(ws->small.p) + ((ntiles * 4 + 15) & ~(uint64_t)15));
                {
                    const uint64_t fm_bytes = sqy::frame_metric_scratch_bytes(ntiles, per_tile, cur_elem);
                    if (ws->lz4_scratch.ensure(std::max<uint64_t>(fm_bytes, 16))) return 1;
                    ProfScope ps("tile_metric", stream, pend);
                    SQY_HIP(sqy::launch_frame_metric(tiled, ntiles, per_tile, cur_elem, d_sums, stream, ws->lz4_scratch.p, fm_bytes, tail));
                }
                std::vector<float> sums(ntiles);
                std::vector<uint64_t> map(ntiles);
                SQY_HIP(hipMemcpyAsync(sums.data(), d_sums, ntiles * sizeof(float), hipMemcpyDeviceToHost, stream));
                SQY_HIP(hipStreamSynchronize(stream));
                sqy::tile_shuffle_order(sums.data(), ntiles, per_tile, cur_elem, map.data(), tail);
                SQY_HIP(hipMemcpyAsync(d_map, map.data(), ntiles * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
                uint8_t* out = next_buf(cur_len * cur_elem);
                if (!out) return 1;
                {
                    ProfScope ps("tile_shuffle", stream, pend);
                    SQY_HIP(sqy::launch_frame_gather(tiled, out, ntiles, tile_bytes, d_map, stream));
                }
                SQY_HIP(hipStreamSynchronize(stream));                     // `map` (host) is read by the async copy above
                st.cfg["reorder_map"] = sqy::to_verbatim(map.data(), ntiles * sizeof(uint64_t));   // tile_shuffle_scheme_impl.hpp:88
                cur = out;
                break;
            }
            case StageKind::diff3x3x1: {
                if (dims.size() != 3) {
                    // diff_scheme_impl.hpp:84-87 returns the output pointer unmoved -> the chain throws
                    // (dynamic_stage_chain.hpp:313-317); no exception may cross this ABI
                    std::fprintf(stderr, "[diff_scheme] unable to process input data that is not 3D\n");
                    return 1;
                }
                // as a tail filter the stream is the sink's `char` output: the volume's shape when that is one byte per voxel, else
                // {1, 1, bytes} (dynamic_pipeline.hpp:658-666), which the stage cannot take
                const bool tail = pipe.sink_index >= 0 && (int)si > pipe.sink_index;
                const bool flat = tail && cur_len * (uint64_t)cur_elem != len;
                const uint64_t Z = flat ? 1 : dims[0], Y = flat ? 1 : dims[1], X = flat ? cur_len : dims[2];
                if ((int64_t)(X - 1) * (int64_t)(Y - 2) <= 1 || Y < 3 || X < 2) {
                    std::fprintf(stderr, "[sqeazy]\t diff3x3x1: shape %llux%llux%llu reads out of bounds in the reference; refused\n",
                                 (unsigned long long)Z, (unsigned long long)Y, (unsigned long long)X);
                    return 1;
                }
                if (cur_elem == 1 && (Z > 127 || Y > 127 || X > 127)) {
                    std::fprintf(stderr, "[sqeazy]\t diff3x3x1 on 8-bit voxels: extents > 127 overflow the reference's char coordinates; refused\n");
                    return 1;
                }
                {
                    const uint32_t sw = (!tail && si + 1 < pipe.stages.size() && pipe.stages[si + 1].kind == StageKind::bitswap1 &&
                                         (reinterpret_cast<uintptr_t>(cur) & 15) == 0)
                                            ? sqy::diff3x3x1_side_width(Z, Y, X, cur_elem) : 0;
                    if (sw) {
                        // a buffer of its own: `cur` stays where it is, so the transpose's output (the next buffer of the
                        // ping/pong rotation) can never be the buffer `cur` lives in
                        if (ws->diff_side.ensure(Z * Y * (uint64_t)sw * 2)) return 1;
                        uint8_t* side = static_cast<uint8_t*>(ws->diff_side.p);
                        ProfScope ps("diff3x3x1", stream, pend);
                        SQY_HIP(sqy::launch_diff3x3x1_side(reinterpret_cast<const uint16_t*>(cur), reinterpret_cast<uint16_t*>(side), Z, Y, X, sw, stream));
                        bsw_side = reinterpret_cast<const uint16_t*>(side);
                        bsw_side_w = sw;
                        bsw_side_X = (uint32_t)X;
                        break;                                              // (`cur` stays the stage's input: the transpose reads both)
                    }
                }
                uint8_t* out = next_buf(cur_len * cur_elem);
                if (!out) return 1;
                ProfScope ps("diff3x3x1", stream, pend);
                SQY_HIP(sqy::launch_diff3x3x1(cur, out, Z, Y, X, cur_elem, stream, tail));
                cur = out;
                break;
            }
            case StageKind::frame_shuffle: {
                if (dims.size() != 3) {
                    std::fprintf(stderr, "[sqeazy::detail::frame_shuffle::encode] received non-3D shape which is currently unsupported!\n");
                    return 1;
                }
                // (tail filter: signed bytes; ONE frame {1, 1, bytes} when the sink did not write one byte per voxel)
                const bool tail = pipe.sink_index >= 0 && (int)si > pipe.sink_index;
                const bool flat = tail && cur_len * (uint64_t)cur_elem != len;
                // frame_chunk_size = N: N consecutive frames are one sort unit (frame_shuffle_utils.hpp:105-133, encode_full) -- the stage
                // on Z / N "frames" of N * Y * X voxels.  Z % N != 0 takes the reference's encode_with_remainder (Boost's P^2 median
                // estimate as the metric, :193-260): not reproduced
                uint64_t fcs = 1;
                {
                    auto c = st.cfg.find("frame_chunk_size");
                    if (c != st.cfg.end()) fcs = (uint64_t)std::max(std::atoi(c->second.c_str()), 0);
                }
                const uint64_t Z0 = flat ? 1 : dims[0];
                if (fcs == 0 || Z0 % fcs != 0) {
                    std::fprintf(stderr, "[sqeazy]\t frame_shuffle: %llu frames are no whole multiple of frame_chunk_size=%llu; the reference's remainder path "
                                         "(a P^2 median estimate as the metric) is not reproduced; refused\n", (unsigned long long)Z0, (unsigned long long)fcs);
                    return 1;
                }
                const uint64_t Z = Z0 / fcs, per_frame = (flat ? cur_len : dims[1] * dims[2]) * fcs;
                if (ws->small.ensure(std::max<uint64_t>(Z * 16, 4096))) return 1;
                float* d_sums = static_cast<float*>(ws->small.p);
                uint64_t* d_map = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(ws->small.p) + ((Z * 4 + 15) & ~(uint64_t)15));
                {
                    const uint64_t fm_bytes = sqy::frame_metric_scratch_bytes(Z, per_frame, cur_elem);
                    if (ws->lz4_scratch.ensure(std::max<uint64_t>(fm_bytes, 16))) return 1;      // free until the sink runs
                    ProfScope ps("frame_metric", stream, pend);
                    SQY_HIP(sqy::launch_frame_metric(cur, Z, per_frame, cur_elem, d_sums, stream, ws->lz4_scratch.p, fm_bytes, tail));
                }
                std::vector<float> sums(Z);
                std::vector<uint64_t> map(Z);
                SQY_HIP(hipMemcpyAsync(sums.data(), d_sums, Z * sizeof(float), hipMemcpyDeviceToHost, stream));
                SQY_HIP(hipStreamSynchronize(stream));
                sqy::frame_shuffle_order(sums.data(), Z, per_frame, map.data());
                SQY_HIP(hipMemcpyAsync(d_map, map.data(), Z * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
                // when lz4 follows immediately and its chunks tile the frames, the permuted copy is never materialised:
                // the LZ4 kernels read frame map[f] where the stream has frame f
                const uint64_t frame_bytes = per_frame * (uint64_t)cur_elem;
                bool fused = false;
                if (si + 1 < pipe.stages.size() && pipe.stages[si + 1].kind == StageKind::lz4 && frame_bytes) {
                    const uint64_t total = cur_len * (uint64_t)cur_elem;
                    const uint64_t chunk = pipe.stages[si + 1].lz4.bytes_per_chunk(total);
                    fused = chunk && frame_bytes % chunk == 0 && chunk <= pipe.stages[si + 1].lz4.block_bytes() &&
                            !(pipe.nthreads == 1 && total > chunk);               // (block-linked frames read a gathered copy)
                }
                if (fused) {
                    lz4_frame_map = d_map;
                    lz4_frame_bytes = frame_bytes;
                    SQY_HIP(hipStreamSynchronize(stream));                 // `map` (host) is read by the async copy above
                } else {
                    uint8_t* out = next_buf(cur_len * cur_elem);
                    if (!out) return 1;
                    {
                        ProfScope ps("frame_gather", stream, pend);
                        SQY_HIP(sqy::launch_frame_gather(cur, out, Z, frame_bytes, d_map, stream));
                    }
                    SQY_HIP(hipStreamSynchronize(stream));                 // `map` (host) is read by the async copy above
                    cur = out;
                }
                st.cfg["frame_chunk_size"] = std::to_string(fcs);
                st.cfg["reorder_map"] = sqy::to_verbatim(map.data(), Z * sizeof(uint64_t));   // frame_shuffle_scheme_impl.hpp:86-90
                break;
            }
            case StageKind::quantiser: {
                // quantiser_scheme<uint16_t,char>::encode (quantiser_scheme_impl.hpp:176-226)
                if (ws->small.ensure(65536 * sizeof(uint32_t) + 65536)) return 1;
                uint32_t* d_histo = static_cast<uint32_t*>(ws->small.p);
                uint8_t* d_lut = static_cast<uint8_t*>(ws->small.p) + 65536 * sizeof(uint32_t);
                {
                    ProfScope ps("histogram_u16", stream, pend);
                    SQY_HIP(sqy::launch_histogram_u16(reinterpret_cast<const uint16_t*>(cur), cur_len, d_histo, stream));
                }
                std::vector<uint32_t> histo(65536);
                std::vector<unsigned char> lut_encode(65536);
                uint16_t lut_decode[256];
                SQY_HIP(hipMemcpyAsync(histo.data(), d_histo, 65536 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                SQY_HIP(hipStreamSynchronize(stream));
                sqy::QuantiserWeighting qw;
                {
                    auto wf = st.cfg.find("weighting_function");
                    if (wf != st.cfg.end() && !sqy::quantiser_parse_weighting(wf->second, &qw)) return 1;    // (refused by supported() already)
                }
                sqy::quantiser_build_luts(histo.data(), 65536, lut_encode.data(), lut_decode, qw);
                SQY_HIP(hipMemcpyAsync(d_lut, lut_encode.data(), 65536, hipMemcpyHostToDevice, stream));
                uint8_t* out = next_buf(cur_len);
                if (!out) return 1;
                // (round 5) bitswap1 right behind the sink: look-up and 8-bit bit-plane transpose in one pass, the stage behind is done too
                const bool fuse_bitswap = si + 1 < pipe.stages.size() && pipe.stages[si + 1].kind == StageKind::bitswap1 &&
                                          (reinterpret_cast<uintptr_t>(cur) & 15) == 0;
                if (fuse_bitswap) {
                    ProfScope ps("quantiser_bitswap1_u8", stream, pend);
                    SQY_HIP(sqy::launch_quantiser_apply_bitswap1_u8(reinterpret_cast<const uint16_t*>(cur), out, cur_len, d_lut, stream));
                    skip_stage = si + 1;
                } else {
                    ProfScope ps("quantiser_apply", stream, pend);
                    SQY_HIP(sqy::launch_quantiser_apply_u16(reinterpret_cast<const uint16_t*>(cur), out, cur_len, d_lut, stream));
                }
                SQY_HIP(hipStreamSynchronize(stream));                     // lut_encode (host) is read by the async copy above
                {
                    // quantiser_scheme_impl.hpp:200-204: the decode LUT goes to the file the caller named, else into the header
                    auto lp = st.cfg.find("decode_lut_path");
                    if (lp != st.cfg.end()) {
                        if (!sqy::quantiser_lut_to_file(lp->second, lut_decode, 256)) {
                            // (the reference does not notice and returns a blob nobody can decode; here the encode fails)
                            std::fprintf(stderr, "[sqeazy]\t quantiser: unable to write the decode LUT to %s\n", lp->second.c_str());
                            return 1;
                        }
                    } else
                        st.cfg["decode_lut_string"] = sqy::to_verbatim(lut_decode, sizeof(lut_decode));
                }
                cur = out;
                cur_elem = 1;                                              // sink output is `char`
                break;
            }
            case StageKind::lz4: {
                lz4p = &st.lz4;
                // liblz4's acceleration: LZ4F turns a negative compression level -k into acceleration k + 1 (lz4frame.c, LZ4F_compressBlock),
                // LZ4_compress_fast_continue caps it at 65537 (lz4.c, LZ4_ACCELERATION_MAX)
                const uint32_t lz4_accel = st.lz4.accel < 0 ? (uint32_t)std::min<int64_t>(1 - (int64_t)st.lz4.accel, 65537) : 1u;
                lz4_total = cur_len * (uint64_t)cur_elem;
                lz4_chunk = lz4_total ? st.lz4.bytes_per_chunk(lz4_total) : 1;
                lz4_nchunks = lz4_total ? (lz4_total + lz4_chunk - 1) / lz4_chunk : 0;
                const bool serial = pipe.nthreads == 1 && lz4_nchunks > 1;          // lz4.hpp:227-234: one block-linked frame
                if (!serial && lz4_chunk <= st.lz4.block_bytes()) {
                    // chunked layout, one LZ4 block per frame: every chunk is independent
                    lz4_stride = (lz4_chunk + 15) & ~(uint64_t)15;
                    if (ws->lz4_scratch.ensure(std::max<uint64_t>(lz4_nchunks * lz4_stride, 16))) return 1;
                    if (ws->csize.ensure(std::max<uint64_t>(lz4_nchunks, 1) * sizeof(uint32_t))) return 1;
                    if (ws->frame_off.ensure((lz4_nchunks + 1 + 4) * sizeof(uint64_t))) return 1;
                    if (lz4_inplace) lz4_tail_info = static_cast<uint64_t*>(ws->frame_off.p) + lz4_nchunks + 1;
                    if (lz4_piece_hash && si > 0 && pipe.stages[si - 1].kind == StageKind::bitswap1) {
                        const uint64_t words = sqy::bitswap1_piece_hash_words(cur, cur, cur_len);     // (same count as when they were made)
                        const uint64_t ph_bytes = (words * 4 + 63) & ~(uint64_t)63;
                        uint8_t* base = static_cast<uint8_t*>(ws->dedupe.p) + ph_bytes;
                        uint32_t* d_dup = reinterpret_cast<uint32_t*>(base + sqy::lz4_dedupe_work_bytes(lz4_nchunks));
                        if (lz4_inplace) lz4_holes = reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(d_dup) + ((lz4_nchunks * 4 + 7) & ~(uint64_t)7));
                        ProfScope ps("lz4_dedupe", stream, pend);
                        // frames in place (acceleration 1): only the key table is built here, the decision per chunk (byte compare, hole fill)
                        // is the first thing the chunk's parse wavefront does (lz4_chunk_dedupe)
                        fused_dedupe = lz4_inplace && lz4_accel == 1;
                        SQY_HIP(sqy::launch_lz4_dedupe(cur, lz4_total, (uint32_t)lz4_chunk, lz4_piece_hash, base, d_dup, stream, lz4_in_stride, lz4_holes,
                                                       dedupe_cleared, fused_dedupe ? &lz4_dedupe_args : nullptr));
                        lz4_dup_of = d_dup;
                        if (fused_dedupe && lz4_digest) { lz4_dedupe_args.digest = lz4_digest; lz4_dedupe_args.digest_stride = lz4_digest_stride; }
                    }
                    if (ws->plan.ensure((lz4_nchunks + 1) * sizeof(uint32_t))) return 1;
                    uint32_t* d_redo = static_cast<uint32_t*>(ws->plan.p);       // chunks the first pass leaves to the dense batches
                    {
                        ProfScope ps("lz4_chunks", stream, pend);
                        SQY_HIP(sqy::launch_lz4_chunks(cur, lz4_total, (uint32_t)lz4_chunk, static_cast<uint8_t*>(ws->lz4_scratch.p), lz4_stride,
                                                       static_cast<uint32_t*>(ws->csize.p), lz4_nchunks, stream, lz4_frame_map, lz4_frame_bytes, d_redo,
                                                       fused_dedupe ? nullptr : lz4_dup_of, lz4_in_stride, lz4_accel, dedupe_cleared,
                                                       fused_dedupe ? &lz4_dedupe_args : nullptr));
                    }
                    auto dense_pass = [&](uint32_t n_redo) -> int {
                        ProfScope ps("lz4_chunks_dense", stream, pend);
                        SQY_HIP(sqy::launch_lz4_chunks_dense(cur, lz4_total, (uint32_t)lz4_chunk, static_cast<uint8_t*>(ws->lz4_scratch.p), lz4_stride,
                                                             static_cast<uint32_t*>(ws->csize.p), d_redo, n_redo, stream, lz4_frame_map, lz4_frame_bytes,
                                                             lz4_in_stride));
                        return 0;
                    };
                    std::string hdr_prefix, hdr_suffix;
                    if (lz4_inplace && !(fq && fq->every > 0)) sqy::header_pack_parts(elem_size, false, dims, pipe.name(), &hdr_prefix, &hdr_suffix);
                    if (lz4_inplace && !hdr_prefix.empty() && hdr_prefix.size() + hdr_suffix.size() <= sqy::kLz4InplaceHeaderTextMax) {
                        // Frames in place, ONE host round trip per call (round 4): frame scan + tail marks, the stored chunks in front of
                        // the tail put aside, the gather and the sqy header are all queued behind the parse right away and take what
                        // they need (where the stored tail begins, the payload size) from device memory; what the host has to know
                        // comes back through pinned memory with the one synchronisation.  Only when the parse left chunks to the
                        // dense pass (streams of short sequences: seldom on microscopy stacks) do these kernels return untouched --
                        // they look at the list's counter -- and run again behind the dense pass.
                        const unsigned char fd[2] = {0x40, (unsigned char)(st.lz4.block_id << 4)};
                        const uint32_t hc = (sqy::xxh32(fd, 2, 0) >> 8) & 0xff;
                        uint8_t* outb = static_cast<uint8_t*>(d_dst);
                        volatile uint64_t* record = static_cast<volatile uint64_t*>(ws->pinned);
                        // (round 6) scan, tail marks, gather and header are ONE kernel (lz4_inplace_tail_fused_kernel: no workgroup waits for
                        // another; with calls in flight the five launches it replaces were 0.3 ms of a call's 2.4).  Only when stored chunks
                        // sit in front of the stored tail -- their bodies lie where gathered frames go -- does it hand back (status 4) to the
                        // separate kernels, which put those chunks aside first.
                        const bool fused_tail = lz4_nchunks <= 65536;         // (a workgroup of the fused kernel owns at most 64 chunks)
                        auto tail_separate = [&](const uint32_t* guard, bool scan_too) -> int {
                            record[0] = 0;
                            if (scan_too) {
                                ProfScope ps("lz4_frame_scan", stream, pend);
                                SQY_HIP(sqy::launch_lz4_frame_scan(static_cast<uint32_t*>(ws->csize.p), lz4_nchunks, lz4_total, (uint32_t)lz4_chunk,
                                                                   static_cast<uint64_t*>(ws->frame_off.p), stream, nullptr, lz4_dup_of, lz4_tail_info, guard,
                                                                   outb + inplace_t0 + 11, lz4_in_stride, fd[1], hc));
                            }
                            ProfScope ps("lz4_frame_gather", stream, pend);
                            SQY_HIP(sqy::launch_lz4_inplace_tail(outb, inplace_t0, lz4_in_stride, lz4_total, (uint32_t)lz4_chunk, lz4_nchunks,
                                                                 static_cast<uint8_t*>(ws->lz4_scratch.p), lz4_stride, static_cast<uint32_t*>(ws->csize.p),
                                                                 static_cast<uint64_t*>(ws->frame_off.p), lz4_dup_of, lz4_tail_info, fd[1], hc,
                                                                 hdr_prefix.data(), (uint32_t)hdr_prefix.size(), hdr_suffix.data(), (uint32_t)hdr_suffix.size(),
                                                                 (uint32_t)elem_size, guard, const_cast<uint64_t*>(record), stream));
                            return 0;
                        };
                        auto tail = [&](const uint32_t* guard) -> int {
                            if (!fused_tail) return tail_separate(guard, true);
                            record[0] = 0;
                            ProfScope ps("lz4_inplace_tail", stream, pend);
                            SQY_HIP(sqy::launch_lz4_inplace_tail_fused(outb, inplace_t0, lz4_in_stride, lz4_total, (uint32_t)lz4_chunk, lz4_nchunks,
                                                                       static_cast<uint8_t*>(ws->lz4_scratch.p), lz4_stride, static_cast<uint32_t*>(ws->csize.p),
                                                                       static_cast<uint64_t*>(ws->frame_off.p), lz4_dup_of, lz4_tail_info, fd[1], hc,
                                                                       hdr_prefix.data(), (uint32_t)hdr_prefix.size(), hdr_suffix.data(),
                                                                       (uint32_t)hdr_suffix.size(), (uint32_t)elem_size, guard, const_cast<uint64_t*>(record), stream));
                            return 0;
                        };
                        if (tail(d_redo)) return 1;
                        SQY_HIP(hipStreamSynchronize(stream));
                        if (record[0] == 2) {
                            if (dense_pass((uint32_t)record[6])) return 1;
                            if (tail(nullptr)) return 1;
                            SQY_HIP(hipStreamSynchronize(stream));
                        }
                        if (record[0] == 4) {                                  // stored chunks in front of the stored tail: put aside first
                            if (tail_separate(nullptr, false)) return 1;
                            SQY_HIP(hipStreamSynchronize(stream));
                        }
                        if (record[0] != 1) {
                            std::fprintf(stderr, "[sqeazy]\t internal error: frames in place did not finish (status %llu)\n", (unsigned long long)record[0]);
                            return 1;
                        }
                        inplace_done = true;
                        inplace_blob_at = record[1]; inplace_blob_bytes = record[2]; payload_bytes = record[3];
                        inplace_hdr_bytes = inplace_blob_bytes - payload_bytes;
                        payload_is_lz4 = true;
                        break;
                    }
                    SQY_HIP(hipMemcpyAsync(ws->pinned, d_redo, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                    SQY_HIP(hipStreamSynchronize(stream));
                    const uint32_t n_redo = *static_cast<uint32_t*>(ws->pinned);
                    if (n_redo && dense_pass(n_redo)) return 1;
                } else if (lz4_total) {
                    // block-linked frames: the serial layout (nthreads == 1) or chunks that span several LZ4 blocks.  The table
                    // of a frame is carried from block to block (lz4_utils.hpp:99-173): one wavefront walks each frame -- or, below, every
                    // block is parsed at once from a guess of that table that is checked afterwards
                    const sqy::Lz4Plan plan = sqy::lz4_plan_blocks(lz4_total, lz4_chunk, st.lz4.block_bytes(), serial);
                    if (!plan.ok || plan.blocks.empty()) {
                        std::fprintf(stderr, "[sqeazy]\t lz4: block layout not available on MI355X\n");
                        return 1;
                    }
                    const uint64_t nblocks = plan.blocks.size(), nframes = plan.frame_first.size() - 1;
                    const uint64_t blocks_bytes = nblocks * sizeof(sqy::Lz4Block), first_bytes = (nframes + 1) * sizeof(uint32_t);
                    if (ws->plan.ensure(blocks_bytes + first_bytes)) return 1;
                    sqy::Lz4Block* d_blocks = static_cast<sqy::Lz4Block*>(ws->plan.p);
                    uint32_t* d_first = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(ws->plan.p) + blocks_bytes);
                    SQY_HIP(hipMemcpyAsync(d_blocks, plan.blocks.data(), blocks_bytes, hipMemcpyHostToDevice, stream));
                    SQY_HIP(hipMemcpyAsync(d_first, plan.frame_first.data(), first_bytes, hipMemcpyHostToDevice, stream));
                    lz4_stride = ((uint64_t)plan.max_block + 15) & ~(uint64_t)15;
                    if (ws->lz4_scratch.ensure(std::max<uint64_t>(nblocks * lz4_stride, 16))) return 1;
                    if (ws->csize.ensure(nblocks * sizeof(uint32_t))) return 1;
                    if (ws->frame_off.ensure((nblocks + 1) * sizeof(uint64_t))) return 1;
                    // Few long frames (the serial layout above all): block-parallel.  Every block is parsed by its own wavefront from
                    // a table rebuilt by parsing the >= 64 KiB in front of it, the tables are checked against what the block in
                    // front really left, and what fails the check is parsed again in order (sqy_kernels.h: Lz4SpecArgs).  Twice the
                    // parse work on thousands of wavefronts instead of one: worth it when the frame walks would leave the chip empty.
                    uint64_t longest = 0;
                    for (uint64_t f = 0; f < nframes; ++f) longest = std::max<uint64_t>(longest, plan.frame_first[f + 1] - plan.frame_first[f]);
                    // measurement / test knobs: SQY_NO_BLOCK_PARALLEL (the frame walk of rounds 2-3), SQY_BLOCK_PARALLEL_WARMUP = bytes
                    // of warm-up in front of a block (default and liblz4's reach: 64 KiB; less makes the guess fail more often --
                    // the result stays exact, the blocks that fail are parsed again)
                    const bool spec_off = g_opt.block_parallel.load() == 0;
                    const uint64_t warmup = (uint64_t)g_opt.block_parallel_warmup.load();
                    const uint64_t list_bytes = nblocks * sizeof(uint32_t);
                    // (without room for the tables -- 32 KiB per block -- the walk, which needs none)
                    const bool spec_wanted = !spec_off && longest >= 3 && nframes < 1024;
                    const bool spec_room = spec_wanted && !ws->spec.ensure(nblocks * sqy::kLz4SpecTableWords * sizeof(uint32_t) + 3 * list_bytes, true);
                    if (spec_wanted && !spec_room) {
                        // (round-4 advice) said once, not per call: the result is the same, the rate is not
                        static std::atomic<bool> told{false};
                        if (!told.exchange(true))
                            std::fprintf(stderr, "[sqeazy]\t lz4: no HBM for the block-parallel parse's tables (%llu MiB): block-linked frames are walked by one "
                                                 "wavefront each (same bytes, hundreds of times slower on long frames)\n",
                                         (unsigned long long)((nblocks * sqy::kLz4SpecTableWords * sizeof(uint32_t)) >> 20));
                    }
                    if (spec_room) {
                        std::vector<uint32_t> wfirst(nblocks), wlast(nblocks), ok(nblocks);
                        for (uint64_t f = 0; f < nframes; ++f)
                            for (uint32_t k = plan.frame_first[f]; k < plan.frame_first[f + 1]; ++k) {
                                uint32_t j = k;
                                uint64_t have = 0;
                                while (j > plan.frame_first[f] && have < warmup) { --j; have += plan.blocks[j].n; }
                                wfirst[k] = j; wlast[k] = (uint32_t)k;
                            }
                        sqy::Lz4SpecArgs sa;
                        sa.tables = static_cast<uint32_t*>(ws->spec.p);
                        uint32_t* d_wfirst = sa.tables + nblocks * sqy::kLz4SpecTableWords;
                        uint32_t* d_wlast = d_wfirst + nblocks;
                        uint32_t* d_ok = d_wlast + nblocks;
                        sa.wave_first = d_wfirst; sa.wave_last = d_wlast; sa.mode = 1;
                        SQY_HIP(hipMemcpyAsync(d_wfirst, wfirst.data(), list_bytes, hipMemcpyHostToDevice, stream));
                        SQY_HIP(hipMemcpyAsync(d_wlast, wlast.data(), list_bytes, hipMemcpyHostToDevice, stream));
                        {
                            ProfScope ps("lz4_linked_blocks", stream, pend);
                            SQY_HIP(sqy::launch_lz4_linked_spec(cur, d_blocks, sa, nblocks, plan.max_block, static_cast<uint8_t*>(ws->lz4_scratch.p),
                                                                lz4_stride, static_cast<uint32_t*>(ws->csize.p), stream, lz4_accel));
                        }
                        for (uint64_t round = 0;; ++round) {
                            {
                                ProfScope ps("lz4_linked_verify", stream, pend);
                                SQY_HIP(sqy::launch_lz4_linked_verify(d_blocks, nblocks, sa.tables, plan.max_block, d_ok, stream));
                            }
                            SQY_HIP(hipMemcpyAsync(ok.data(), d_ok, list_bytes, hipMemcpyDeviceToHost, stream));
                            SQY_HIP(hipStreamSynchronize(stream));
                            // runs of blocks that did not start from the true table: one wavefront each, in order, from the table in front
                            // (round-5 advice) a run is parsed by ONE wavefront, block after block: at most kRunMax blocks of it per launch (the
                            // rest keep failing the check and are taken by the next rounds, each from the table the last one left) -- the top
                            // plane of a quantised stack fails as one run of 511 blocks, seconds of work: sixteen launches of a fraction of a
                            // second instead of one kernel that runs for seconds; and the caller is told, once, what layout to ask for.
                            constexpr uint64_t kRunMax = 32;
                            uint64_t nruns = 0, longest_run = 0;
                            for (uint64_t k = 0; k < nblocks; ++k) {
                                if (ok[k]) continue;
                                uint64_t e = k;
                                while (e + 1 < nblocks && !ok[e + 1] && !(plan.blocks[e + 1].flags & 1u)) ++e;
                                longest_run = std::max(longest_run, e - k + 1);
                                wfirst[nruns] = (uint32_t)k; wlast[nruns] = (uint32_t)std::min(e, k + kRunMax - 1); ++nruns;
                                k = e;
                            }
                            if (longest_run > 4 * kRunMax) {
                                static std::atomic<bool> told{false};
                                if (!told.exchange(true))
                                    std::fprintf(stderr, "[sqeazy]\t lz4: %llu blocks in a row of this block-linked frame (nthreads = 1) can only be parsed one after "
                                                         "the other -- a stream of short sequences, whose table no guess reproduces -- by one wavefront, "
                                                         "~10 ms per block.  The chunked layout (nthreads = 0 or > 1: independent frames, same decoder) "
                                                         "takes milliseconds for the same data.\n", (unsigned long long)longest_run);
                            }
                            if (g_opt.block_parallel_stats.load()) {
                                uint64_t nbad = 0;
                                for (uint64_t r = 0; r < nruns; ++r) nbad += wlast[r] - wfirst[r] + 1;
                                std::fprintf(stderr, "[sqeazy]\t lz4 block-parallel: round %llu, %llu of %llu blocks to parse again in %llu runs",
                                             (unsigned long long)round, (unsigned long long)nbad, (unsigned long long)nblocks, (unsigned long long)nruns);
                                for (uint64_t r = 0; r < nruns && r < 24; ++r) std::fprintf(stderr, "%s%u..%u", r ? ", " : ": blocks ", wfirst[r], wlast[r]);
                                std::fprintf(stderr, "\n");
                            }
                            if (nruns == 0) break;
                            if (round > nblocks + 8) {
                                std::fprintf(stderr, "[sqeazy]\t lz4: the block-parallel parse did not settle\n");
                                return 1;
                            }
                            SQY_HIP(hipMemcpyAsync(d_wfirst, wfirst.data(), nruns * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                            SQY_HIP(hipMemcpyAsync(d_wlast, wlast.data(), nruns * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                            sa.mode = 2;
                            ProfScope ps("lz4_linked_redo", stream, pend);
                            SQY_HIP(sqy::launch_lz4_linked_spec(cur, d_blocks, sa, nruns, plan.max_block, static_cast<uint8_t*>(ws->lz4_scratch.p),
                                                                lz4_stride, static_cast<uint32_t*>(ws->csize.p), stream, lz4_accel));
                            SQY_HIP(hipStreamSynchronize(stream));         // (wfirst / wlast are reused by the next round)
                        }
                    } else {
                        ProfScope ps("lz4_linked", stream, pend);
                        SQY_HIP(sqy::launch_lz4_linked(cur, d_blocks, d_first, nframes, plan.max_block, static_cast<uint8_t*>(ws->lz4_scratch.p),
                                                       lz4_stride, static_cast<uint32_t*>(ws->csize.p), stream, lz4_accel));
                    }
                    SQY_HIP(hipStreamSynchronize(stream));                 // `plan` (host) is read by the async copies above
                    lz4_blocks = d_blocks;
                    lz4_nchunks = nblocks;                                  // scan and gather work per block from here on
                    lz4_chunk = plan.max_block;
                }
                {
                    ProfScope ps("lz4_frame_scan", stream, pend);
                    SQY_HIP(sqy::launch_lz4_frame_scan(static_cast<uint32_t*>(ws->csize.p), lz4_nchunks, lz4_total, (uint32_t)lz4_chunk,
                                                       static_cast<uint64_t*>(ws->frame_off.p), stream, lz4_blocks, lz4_dup_of, lz4_tail_info));
                }
                payload_is_lz4 = true;
                break;
            }
            default:
                std::fprintf(stderr, "[sqeazy]\t stage %s is not implemented on MI355X\n", st.name.c_str());
                return 1;
        }
    }

    if (inplace_done) {
        if (payload_bytes > (uint64_t)INT_MAX) {
            std::fprintf(stderr, "[sqeazy]\t lz4: %llu payload bytes overflow the reference's int byte count\n", (unsigned long long)payload_bytes);
            return 1;
        }
        if (g_prof_on.load()) prof_collect(cx.pending);
        *dstoffset = (long)inplace_blob_at;
        *dstlength = (long)inplace_blob_bytes;
        (void)inplace_hdr_bytes;
        return 0;
    }
    // ---- payload size ----
    uint64_t tail_j = 0, tail_head_bytes = 0, tail_raw_head = 0;
    if (payload_is_lz4) {
        if (lz4_nchunks == 0) {
            payload_bytes = 7 + 4;                     // empty input: frame header + end mark
        } else if (lz4_inplace) {
            SQY_HIP(hipMemcpyAsync(ws->pinned, lz4_tail_info, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
            SQY_HIP(hipStreamSynchronize(stream));
            const uint64_t* ti = static_cast<const uint64_t*>(ws->pinned);
            tail_j = ti[0]; tail_head_bytes = ti[1]; tail_raw_head = ti[2]; payload_bytes = ti[3];
        } else {
            SQY_HIP(hipMemcpyAsync(ws->pinned, static_cast<uint64_t*>(ws->frame_off.p) + lz4_nchunks, sizeof(uint64_t),
                                   hipMemcpyDeviceToHost, stream));
            SQY_HIP(hipStreamSynchronize(stream));
            payload_bytes = *static_cast<uint64_t*>(ws->pinned);
        }
        if (payload_bytes > (uint64_t)INT_MAX) {
            // encode_parallel sums the chunk sizes into an `int` and rejects the result (lz4_utils.hpp:264-273)
            std::fprintf(stderr, "[sqeazy]\t lz4: %llu payload bytes overflow the reference's int byte count\n",
                         (unsigned long long)payload_bytes);
            return 1;
        }
    } else {
        payload_bytes = cur_len * (uint64_t)cur_elem;
    }

    // ---- header (written after encoding, as the reference rewrites it: dynamic_pipeline.hpp:599-612) ----
    const std::string hdr = sqy::header_pack(elem_size, false, dims, pipe.name(), payload_bytes);
    if (fq && fq->every > 0) {
        if (!payload_is_lz4 || lz4_blocks || !fq->offsets) { std::fprintf(stderr, "[sqeazy]\t frame offsets: the payload is not one LZ4 frame per chunk\n"); return 1; }
        const uint64_t cnt = (lz4_nchunks + (uint64_t)fq->every - 1) / (uint64_t)fq->every;
        if (cnt + 1 > (uint64_t)std::max(fq->max_entries, 0)) { std::fprintf(stderr, "[sqeazy]\t frame offsets: %llu entries do not fit\n", (unsigned long long)(cnt + 1)); return 1; }
        std::vector<uint64_t> fo(cnt + 1, 0);
        if (cnt)
            SQY_HIP(hipMemcpy2DAsync(fo.data(), sizeof(uint64_t), ws->frame_off.p, (size_t)fq->every * sizeof(uint64_t), sizeof(uint64_t), cnt,
                                     hipMemcpyDeviceToHost, stream));
        SQY_HIP(hipStreamSynchronize(stream));
        for (uint64_t i = 0; i < cnt; ++i) fq->offsets[i] = (long)(fo[i] + hdr.size());
        fq->offsets[cnt] = (long)(hdr.size() + payload_bytes);
        fq->count = (int)cnt;
    }
    const uint64_t blob_bytes = hdr.size() + payload_bytes;
    if (blob_bytes > dst_capacity) {
        std::fprintf(stderr, "[sqeazy]\t destination buffer too small (%llu > %llu bytes)\n", (unsigned long long)blob_bytes,
                     (unsigned long long)dst_capacity);
        return 1;
    }
    uint8_t* out = static_cast<uint8_t*>(d_dst);
    if (lz4_inplace) {
        // the run of stored chunks j.. that ends the payload stays where the bit-plane transpose put it; frames 0..j-1 are
        // gathered so that they end where frame j begins, the header goes in front of them
        const uint64_t frame_j = inplace_t0 + tail_j * lz4_in_stride;
        if (tail_head_bytes + hdr.size() > frame_j) { std::fprintf(stderr, "[sqeazy]\t internal error: frames in place overlap the header\n"); return 1; }
        const uint64_t payload_at = frame_j - tail_head_bytes, blob_at = payload_at - hdr.size();
        const unsigned char fd[2] = {0x40, (unsigned char)(lz4p->block_id << 4)};
        const uint32_t hc = (sqy::xxh32(fd, 2, 0) >> 8) & 0xff;
        uint8_t* body0 = out + inplace_t0 + 11;
        {
            ProfScope ps("lz4_tail_marks", stream, pend);
            SQY_HIP(sqy::launch_lz4_tail_marks(body0, lz4_in_stride, lz4_total, (uint32_t)lz4_chunk, lz4_nchunks, fd[1], hc, lz4_tail_info, stream));
        }
        if (tail_raw_head) {
            ProfScope ps("lz4_stash_raw", stream, pend);
            SQY_HIP(sqy::launch_lz4_stash_raw(body0, lz4_in_stride, lz4_total, (uint32_t)lz4_chunk, static_cast<uint8_t*>(ws->lz4_scratch.p), lz4_stride,
                                              static_cast<uint32_t*>(ws->csize.p), lz4_dup_of, tail_j, stream));
        }
        if (tail_j) {
            ProfScope ps("lz4_frame_gather", stream, pend);
            SQY_HIP(sqy::launch_lz4_frame_gather(body0, lz4_total, (uint32_t)lz4_chunk, static_cast<uint8_t*>(ws->lz4_scratch.p), lz4_stride,
                                                 static_cast<uint32_t*>(ws->csize.p), static_cast<uint64_t*>(ws->frame_off.p), out + payload_at, fd[1], hc,
                                                 tail_j, stream, nullptr, 0, nullptr, lz4_dup_of, lz4_in_stride, tail_raw_head != 0));
        }
        SQY_HIP(hipMemcpyAsync(out + blob_at, hdr.data(), hdr.size(), hipMemcpyHostToDevice, stream));
        SQY_HIP(hipStreamSynchronize(stream));
        if (g_prof_on.load()) prof_collect(cx.pending);
        *dstoffset = (long)blob_at;
        *dstlength = (long)blob_bytes;
        return 0;
    }
    SQY_HIP(hipMemcpyAsync(out, hdr.data(), hdr.size(), hipMemcpyHostToDevice, stream));
    if (payload_is_lz4) {
        const unsigned char fd[2] = {0x40, (unsigned char)(lz4p->block_id << 4)};
        const uint32_t hc = (sqy::xxh32(fd, 2, 0) >> 8) & 0xff;
        if (lz4_nchunks == 0) {
            const unsigned char empty[11] = {0x04, 0x22, 0x4D, 0x18, fd[0], fd[1], (unsigned char)hc, 0, 0, 0, 0};
            SQY_HIP(hipMemcpyAsync(out + hdr.size(), empty, sizeof(empty), hipMemcpyHostToDevice, stream));
        } else {
            ProfScope ps("lz4_frame_gather", stream, pend);
            SQY_HIP(sqy::launch_lz4_frame_gather(cur, lz4_total, (uint32_t)lz4_chunk, static_cast<uint8_t*>(ws->lz4_scratch.p), lz4_stride,
                                                 static_cast<uint32_t*>(ws->csize.p), static_cast<uint64_t*>(ws->frame_off.p),
                                                 out + hdr.size(), fd[1], hc, lz4_nchunks, stream, lz4_frame_map, lz4_frame_bytes, lz4_blocks, lz4_dup_of));
        }
    } else {
        ProfScope ps("payload_copy", stream, pend);
        SQY_HIP(hipMemcpyAsync(out + hdr.size(), cur, payload_bytes, hipMemcpyDeviceToDevice, stream));
    }
    SQY_HIP(hipStreamSynchronize(stream));
    if (g_prof_on.load()) prof_collect(cx.pending);
    *dstlength = (long)blob_bytes;
    (void)raw_bytes;
    return 0;
}

// dst_capacity < 0: the caller followed the reference protocol and allocated SQY_Pipeline_Max_Compressed_Length bytes
int encode_from_host(const char* pipeline, const char* src, long* shape, unsigned rank, int elem_size, char* dst,
                     long* dstlength, int nthreads, long dst_capacity = -1)
{
    if (!pipeline || !src || !shape || !dst || !dstlength) return 1;
    {
        std::string why;
        if (!Pipeline::supported(pipeline, elem_size, &why)) {
            if (Pipeline::reference_accepts(pipeline)) std::fprintf(stderr, "[sqeazy]\t pipeline %s: %s\n", pipeline, why.c_str());
            return 1;   // sqeazy.cpp:81-82,118-119: invalid pipeline -> 1 before touching any buffer
        }
    }
    if (!device_present()) { std::fprintf(stderr, "[sqeazy]\t no MI355X (HIP device) visible: sqeazy_amd has no CPU path\n"); return 1; }
    ContextLease lease;
    if (!lease.ctx) { std::fprintf(stderr, "[sqeazy]\t no usable HIP device\n"); return 1; }
    Workspace* ws = &lease.ctx->ws;
    hipStream_t stream = lease.ctx->own_stream();
    if (!stream) { std::fprintf(stderr, "[sqeazy]\t no HIP stream\n"); return 1; }
    uint64_t len = 1;
    for (unsigned i = 0; i < rank; ++i) {
        if (shape[i] <= 0) return 1;
        len *= (uint64_t)shape[i];
        if (len >= ((uint64_t)1 << 31)) break;
    }
    if (len >= ((uint64_t)1 << 31)) {
        std::fprintf(stderr, "[sqeazy]\t 2^31 or more voxels in one call overflow the reference's int voxel count; encode z-slabs\n");
        return 1;
    }
    const uint64_t raw = len * (uint64_t)elem_size;
    Pipeline pipe = Pipeline::from_string(pipeline, elem_size);
    pipe.set_n_threads(nthreads);
    // What the caller was told to allocate: SQY_Pipeline_Max_Compressed_Length_* evaluates the bound on a fresh
    // pipeline (n_threads = 1, sqeazy.cpp:144-231).  The reference itself writes past that for pipelines whose
    // header grows while encoding (frame_shuffle's reorder_map on stacks of many small frames); here the
    // documented "error 1 - destination buffer is not large enough" (inc/sqeazy.h:105) is returned instead.
    const uint64_t bound = dst_capacity >= 0 ? (uint64_t)dst_capacity : Pipeline::from_string(pipeline, elem_size).max_encoded_size(raw, elem_size);
    if (ws->io_src.ensure(std::max<uint64_t>(raw, 16)) || ws->io_dst.ensure(std::max<uint64_t>(bound, 16))) return 1;
    int dev_id = 0;
    SQY_HIP(hipGetDevice(&dev_id));
    if (!lease.ctx->stager.copy(ws->io_src.p, const_cast<char*>(src), raw, true, dev_id)) { std::fprintf(stderr, "[sqeazy]\t host to device transfer failed\n"); return 1; }
    long out_len = 0, out_at = 0;
    const int rc = encode_on_device(*lease.ctx, pipeline, ws->io_src.p, shape, rank, elem_size, ws->io_dst.p, bound, &out_len, nthreads, stream, &out_at);
    if (rc) return rc;                  // (returns with the blob complete: the stream has been synchronised)
    if (!lease.ctx->stager.copy(static_cast<char*>(ws->io_dst.p) + out_at, dst, (size_t)out_len, false, dev_id)) { std::fprintf(stderr, "[sqeazy]\t device to host transfer failed\n"); return 1; }
    *dstlength = out_len;
    return 0;
}

// ---- decode --------------------------------------------------------------------------------------
// dynamic_pipeline::decode (dynamic_pipeline.hpp:740-846): tail filters^-1, sink^-1, head filters^-1 in reverse.
// the quantiser's decode LUT (256 x u16, base64 in the header) into ws->small; synchronous: the host copy does not outlive the call
int quantiser_lut_to_device(const sqy::Stage& st, Workspace* ws, hipStream_t stream)
{
    std::vector<unsigned char> lut;
    auto lp = st.cfg.find("decode_lut_path");
    if (lp != st.cfg.end()) {
        // quantiser_scheme_impl.hpp:83-85: a path in the configuration wins over a LUT string
        lut.resize(512);
        if (!sqy::quantiser_lut_from_file(lp->second, reinterpret_cast<uint16_t*>(lut.data()), 256)) {
            std::fprintf(stderr, "lut from %s cannot be loaded, decoding skipped\n", lp->second.c_str());               // quantiser_utils.hpp:559-562
            return 1;
        }
    } else {
        auto it = st.cfg.find("decode_lut_string");
        if (it == st.cfg.end() || it->second.size() < 21) { std::fprintf(stderr, "[sqeazy]\t quantiser: no decode_lut_string in the header\n"); return 1; }
        const std::string b64 = it->second.substr(10, it->second.size() - 21);   // strip <verbatim> ... </verbatim>
        lut = sqy::base64_decode(b64);
    }
    if (lut.size() != 512) { std::fprintf(stderr, "[sqeazy]\t quantiser: malformed decode LUT\n"); return 1; }
    if (ws->small.ensure(4096)) return 1;
    SQY_HIP(hipMemcpy(ws->small.p, lut.data(), 512, hipMemcpyHostToDevice));
    (void)stream;
    return 0;
}

// The header of a blob is untrusted input: rank 1..16, every extent positive and below 2^31, fewer than 2^31 voxels (what one
// encode call can have produced), header and payload inside the blob.  *raw_bytes = decoded size.
bool header_shape_ok(const sqy::HeaderInfo& h, uint64_t srclen, uint64_t* raw_bytes)
{
    if (h.shape.empty() || h.shape.size() > 16) { std::fprintf(stderr, "[sqeazy]\t decode: header with rank %zu\n", h.shape.size()); return false; }
    uint64_t n = 1;
    for (uint64_t d : h.shape) {
        if (d == 0 || d >= ((uint64_t)1 << 31)) { std::fprintf(stderr, "[sqeazy]\t decode: header with an extent of %llu\n", (unsigned long long)d); return false; }
        n *= d;
        if (n >= ((uint64_t)1 << 31)) { std::fprintf(stderr, "[sqeazy]\t decode: header claims 2^31 or more voxels\n"); return false; }
    }
    const int elem = h.elem_size();
    if (elem != 1 && elem != 2) { std::fprintf(stderr, "[sqeazy]\t decode: blob holds %s voxels\n", h.type.c_str()); return false; }
    if (h.size > srclen || h.payload_bytes > srclen - h.size) { std::fprintf(stderr, "[sqeazy]\t decode: blob truncated\n"); return false; }
    *raw_bytes = n * (uint64_t)elem;
    return true;
}

int decode_on_device(Context& cx, const void* d_src_v, uint64_t srclen, void* d_dst, uint64_t dst_capacity, int want_elem, hipStream_t stream)
{
    if (!d_src_v || !d_dst) return 1;
    const uint8_t* d_src = static_cast<const uint8_t*>(d_src_v);
    Workspace* ws = &cx.ws;
    std::vector<PendingEvent>* pend = &cx.pending;
    DrainOnExit drain{stream, pend, cx.side};

    // header: fetch a prefix of the blob, grow until the delimiter is inside
    std::vector<char> head;
    sqy::HeaderInfo h;
    for (uint64_t want = 1 << 16;; want *= 16) {
        const uint64_t take = std::min<uint64_t>(want, srclen);
        head.resize(take);
        SQY_HIP(hipMemcpyAsync(head.data(), d_src, take, hipMemcpyDeviceToHost, stream));
        SQY_HIP(hipStreamSynchronize(stream));
        h = sqy::header_unpack(head.data(), head.data() + take);
        if (h.valid || take == srclen) break;
    }
    if (!h.valid) { std::fprintf(stderr, "[sqeazy]\t unable to find a sqy header in the blob\n"); return 1; }
    const int elem = h.elem_size();
    if (elem != want_elem) { std::fprintf(stderr, "[sqeazy]\t blob holds %s voxels\n", h.type.c_str()); return 1; }
    std::string why;
    if (!Pipeline::supported(h.pipename, elem, &why)) {
        std::fprintf(stderr, "[sqeazy]\t%s cannot be build with this version of sqeazy (%s)\n", h.pipename.c_str(), why.c_str());
        return 1;
    }
    Pipeline pipe = Pipeline::from_string(h.pipename);
    // the header is untrusted input: every extent positive, the voxel count below 2^31 (what one encode call can have
    // produced), no wrap-around anywhere
    uint64_t raw_bytes = 0;
    if (!header_shape_ok(h, srclen, &raw_bytes)) return 1;
    const uint64_t n = raw_bytes / (uint64_t)elem;
    if (raw_bytes > dst_capacity) {
        std::fprintf(stderr, "[sqeazy]\t decode: buffer too small or blob truncated\n");
        return 1;
    }
    // composite return codes of dynamic_pipeline::detail_decode (dynamic_pipeline.hpp:795-846): a failing tail filter
    // returns its code, a failing sink code + 10, a failing head filter code + 100
    const int sink_index = pipe.sink_index;
    auto stage_error = [&](size_t si) -> int {
        if (sink_index >= 0 && (int)si > sink_index) return 1;
        if (sink_index >= 0 && (int)si == sink_index) return 1 + 10;
        return 1 + 100;
    };
    // element size of the stream in front of every stage (the quantiser sink turns it into bytes)
    // (the quantiser maps every voxel to one byte; pass_through re-types the voxels: elem times as many one-byte elements)
    std::vector<int> elem_before(pipe.stages.size());
    std::vector<uint64_t> count_before(pipe.stages.size());
    {
        int e = elem;
        uint64_t c = n;
        for (size_t i = 0; i < pipe.stages.size(); ++i) {
            elem_before[i] = e;
            count_before[i] = c;
            if (pipe.stages[i].kind == StageKind::quantiser) e = 1;
            if (pipe.stages[i].kind == StageKind::pass_through) { c *= (uint64_t)e; e = 1; }
        }
    }
    const uint8_t* cur = d_src + h.size;
    uint64_t cur_bytes = h.payload_bytes;
    bool use_ping = true;
    bool diff_in_place = false;            // the bit-plane inverse wrote into the volume itself; the diff3x3x1 inverse works there
    const uint32_t* lz4_flag = nullptr;    // the LZ4 decoder's error flag, read when the call ends
    int lz4_flag_stage = 0;
    auto out_buf = [&](size_t stage_index, uint64_t bytes) -> uint8_t* {
        if (stage_index == 0) return static_cast<uint8_t*>(d_dst);               // the first stage's inverse produces the volume
        DevBuf& b = use_ping ? ws->ping : ws->pong;
        use_ping = !use_ping;
        if (b.ensure(std::max<uint64_t>(bytes, 16))) return nullptr;
        return static_cast<uint8_t*>(b.p);
    };

    // the shape a 3-D stage saw on the encoder's side (h.shape.size() == 3 checked by the caller): the volume's, or {1, 1, bytes} behind
    // a sink that did not write one byte per voxel (dynamic_pipeline.hpp:658-666)
    auto stage_shape = [&](size_t si, uint64_t n_in, uint64_t& Z, uint64_t& Y, uint64_t& X) {
        const bool flat = sink_index >= 0 && (int)si > sink_index && n_in != n;
        Z = flat ? 1 : h.shape[0]; Y = flat ? 1 : h.shape[1]; X = flat ? n_in : h.shape[2];
    };

    // frame_shuffle's inverse: the reorder map out of the header, checked and sent to the device (ws->small).  Used by the stage itself and
    // by the LZ4 stage behind it, which decodes its frames straight to their places when it can (round 5).
    std::vector<unsigned char> fs_map;                                          // (alive until the call's last synchronisation)
    std::vector<uint64_t> fs_unnamed;                                           // places no map entry names (maps that are no permutation)
    // zeros where nobody writes: the places the map does not name (round 6: not the whole volume -- the C4 stack's map leaves a few of its
    // 1024 places out, and clearing 1 GiB for them was 0.25 of the decode's 0.85 ms)
    auto zero_unnamed_places = [&](uint8_t* out, uint64_t place_bytes, uint64_t bytes) -> int {
        // (a memset's launch costs about what 25 MB of it cost the memory)
        if (fs_unnamed.size() * (place_bytes + (25ull << 20)) >= bytes) { SQY_HIP(hipMemsetAsync(out, 0, bytes, stream)); return 0; }
        for (uint64_t v : fs_unnamed) SQY_HIP(hipMemsetAsync(out + v * place_bytes, 0, place_bytes, stream));
        return 0;
    };
    auto frame_shuffle_prepare = [&](size_t fi, uint64_t& Z, uint64_t& frame_bytes_dec, bool& permutation) -> int {
        const Stage& fs = pipe.stages[fi];
        const uint64_t fs_n = count_before[fi], fs_bytes = fs_n * (uint64_t)elem_before[fi];
        if (h.shape.size() != 3) return 1;
        auto it = fs.cfg.find("reorder_map");
        // (as a tail filter behind a sink that did not write one byte per voxel the stream is ONE frame: {1, 1, bytes})
        const bool one_frame = sink_index >= 0 && (int)fi > sink_index && fs_n != n;
        uint64_t fcs = 1;
        {
            auto c = fs.cfg.find("frame_chunk_size");
            if (c != fs.cfg.end()) fcs = (uint64_t)std::max(std::atoi(c->second.c_str()), 0);
        }
        const uint64_t Z0 = one_frame ? 1 : h.shape[0];
        if (fcs == 0 || Z0 % fcs != 0) { std::fprintf(stderr, "[sqeazy]\t frame_shuffle: frame_chunk_size does not divide the frames\n"); return 1; }
        Z = Z0 / fcs;
        frame_bytes_dec = (one_frame ? fs_bytes : h.shape[1] * h.shape[2] * (uint64_t)elem_before[fi]) * fcs;
        if (it == fs.cfg.end() || it->second.size() < 21) { std::fprintf(stderr, "[sqeazy]\t frame_shuffle: no reorder_map in the header\n"); return 1; }
        fs_map = sqy::base64_decode(it->second.substr(10, it->second.size() - 21));
        if (fs_map.size() != Z * 8) { std::fprintf(stderr, "[sqeazy]\t frame_shuffle: malformed reorder_map\n"); return 1; }
        std::vector<bool> targeted(Z, false);
        permutation = true;
        for (uint64_t i = 0; i < Z; ++i) {
            uint64_t v; std::memcpy(&v, fs_map.data() + 8 * i, 8);
            if (v >= Z) { std::fprintf(stderr, "[sqeazy]\t frame_shuffle: reorder_map out of range\n"); return 1; }
            if (targeted[v]) permutation = false;
            targeted[v] = true;
        }
        fs_unnamed.clear();
        if (!permutation) for (uint64_t v = 0; v < Z; ++v) if (!targeted[v]) fs_unnamed.push_back(v);
        if (!permutation) {
            // A map that names a place twice (frames of equal metric on the encoder's side: same bytes -- or a crafted blob: not): the
            // reference's decode walks the frames in order, the LAST one named for a place stays (frame_shuffle_utils.hpp:337-344).
            // Same here, whatever order the scatter's workgroups run in: the earlier ones are struck from the device's copy of the map.
            std::vector<uint64_t> last(Z, ~0ull);
            for (uint64_t i = 0; i < Z; ++i) { uint64_t v; std::memcpy(&v, fs_map.data() + 8 * i, 8); last[v] = i; }
            for (uint64_t i = 0; i < Z; ++i) {
                uint64_t v; std::memcpy(&v, fs_map.data() + 8 * i, 8);
                if (last[v] != i) { const uint64_t none = ~0ull; std::memcpy(fs_map.data() + 8 * i, &none, 8); }
            }
        }
        if (ws->small.ensure(std::max<uint64_t>(Z * 8, 4096))) return 1;
        SQY_HIP(hipMemcpyAsync(ws->small.p, fs_map.data(), Z * 8, hipMemcpyHostToDevice, stream));
        SQY_HIP(hipStreamSynchronize(stream));                                   // (pageable source: gone from the host's side before anything can return)
        return 0;
    };

    for (size_t si = pipe.stages.size(); si-- > 0;) {
        const Stage& st = pipe.stages[si];
        const int e_in = elem_before[si];                                       // element size on the ENCODER's input side of this stage
        const uint64_t n_in = count_before[si];                                 // elements on that side
        const uint64_t stage_in_bytes = n_in * (uint64_t)e_in;                  // bytes the inverse has to produce
        switch (st.kind) {
            case StageKind::lz4: {
                const uint64_t total = stage_in_bytes;
                const uint64_t chunk = total ? st.lz4.bytes_per_chunk(total) : 1;
                const uint64_t block_bytes = st.lz4.block_bytes();
                const uint64_t nchunks = total ? (total + chunk - 1) / chunk : 0;
                const uint64_t max_blocks = std::max<uint64_t>(nchunks * ((chunk + block_bytes - 1) / block_bytes), total / block_bytes + 1) + 16;
                // block list, frame starts, and a table of frame-start candidates (16 B x >= 8 slots per expected frame)
                const uint64_t idx_bytes = (max_blocks * 16 + (max_blocks + 2) * 4 + 64 + 15) & ~15ull;
                const uint64_t cand_bytes = sqy::lz4_frame_rank_scratch_bytes(nchunks);
                if (ws->lz4_scratch.ensure(idx_bytes + cand_bytes)) return 1;
                uint8_t* blk = static_cast<uint8_t*>(ws->lz4_scratch.p);
                uint32_t* frame_first = reinterpret_cast<uint32_t*>(blk + max_blocks * 16);
                void* cand = blk + idx_bytes;
                if (ws->csize.ensure(64)) return 1;
                uint32_t* counts = static_cast<uint32_t*>(ws->csize.p);          // [0..3] index result, [4] decode error flag
                SQY_HIP(hipMemsetAsync(counts, 0, 64, stream));
                uint32_t hc[8] = {0, 0, 100, 0, 0, 0, 0, 0};
                if (nchunks > 1) {
                    // chunked layout expected: rank the frame list in parallel.  The frames at the stream's end that are stored blocks of
                    // the chunk size are found where they must start, not by the scan (hc[6] of them); should the ranking give up with
                    // such a tail, the whole stream is scanned before the walk below is tried.
                    for (int with_tail = g_opt.stored_tail_index.load() ? 1 : 0; with_tail >= 0; --with_tail) {
                        {
                            ProfScope ps("lz4_frame_rank", stream, pend);
                            SQY_HIP(sqy::launch_lz4_frame_rank(cur, cur_bytes, blk, frame_first, max_blocks, counts, nchunks, cand, stream,
                                                               with_tail ? chunk : 0, with_tail ? total - (nchunks - 1) * chunk : 0));
                        }
                        SQY_HIP(hipMemcpyAsync(hc, counts, sizeof(hc), hipMemcpyDeviceToHost, stream));
                        SQY_HIP(hipStreamSynchronize(stream));
                        if (hc[2] != 100 || hc[6] == 0) break;
                        SQY_HIP(hipMemsetAsync(counts, 0, 64, stream));
                    }
                }
                if (hc[2] == 100) {
                    // one frame, the serial block-linked layout, or anything the parallel ranking does not cover
                    {
                        ProfScope ps("lz4_frame_index", stream, pend);
                        SQY_HIP(sqy::launch_lz4_frame_index(cur, cur_bytes, blk, frame_first, max_blocks, counts, stream));
                    }
                    SQY_HIP(hipMemcpyAsync(hc, counts, sizeof(hc), hipMemcpyDeviceToHost, stream));
                    SQY_HIP(hipStreamSynchronize(stream));
                }
                if (hc[2]) { std::fprintf(stderr, "[sqy::lz4] corrupt LZ4 frame stream (code %u)\n", hc[2]); return stage_error(si); }
                const uint32_t nframes = hc[0];
                if (nframes > 1 && nframes != nchunks) {
                    std::fprintf(stderr, "[sqy::lz4] %u frames where %llu chunks were expected\n", nframes, (unsigned long long)nchunks);
                    return stage_error(si);
                }
                if (nframes == 0 && total > 0) {
                    std::fprintf(stderr, "[sqy::lz4] no LZ4 frame in the payload, %llu bytes expected\n", (unsigned long long)total);
                    return stage_error(si);
                }
                // frame_shuffle right in front (on the encoder's side), the chunked layout, every chunk inside one of its frames: the frames
                // are decoded straight to where the shuffle's inverse would move them (round 5: one pass over the volume less -- the C4
                // config's decode 1.49 -> 1.1 ms)
                const uint64_t* remap = nullptr;
                uint64_t remap_bytes = 0;
                bool remap_zero = false;
                if (si >= 1 && pipe.stages[si - 1].kind == StageKind::frame_shuffle && nframes == nchunks && nframes > 1 && total % chunk == 0 &&
                    count_before[si - 1] * (uint64_t)elem_before[si - 1] == total && h.shape.size() == 3) {
                    uint64_t Z = 0, fb = 0;
                    bool permutation = true;
                    if (const int rc = frame_shuffle_prepare(si - 1, Z, fb, permutation)) return rc;
                    // (round-5 advice) a map that names a place twice -- frames of equal metric on the encoder's side, or a crafted blob --:
                    // several LZ4 frames must not decode into one place at once (the ring kernels read matches that reach behind their
                    // ring back from there).  The device's copy of such a map has every frame but the last one named for a place struck
                    // (frame_shuffle_prepare): struck frames are not decoded, the places nobody names are zeroed first.
                    if (fb && fb % chunk == 0 && Z * fb == total) {
                        remap = static_cast<const uint64_t*>(ws->small.p);
                        remap_bytes = fb;
                        remap_zero = !permutation;
                    }
                }
                uint8_t* out = out_buf(remap ? si - 1 : si, total);
                if (!out) return 1;
                if (remap_zero) { if (const int rc = zero_unnamed_places(out, remap_bytes, total)) return rc; }   // (frames nobody names come out as zeros, as behind the stage's own inverse)
                uint32_t bad = 0;
                bool decoded = false;
                // ONE block-linked frame (nthreads = 1 on the encoder's side): every block at once with the history as an unknown, the
                // references resolved afterwards (sqy_kernels.hip: lz4_blocks_decode_sym_kernel).  A stream that is not a frame of
                // full blocks, or is damaged, raises the flag: the one-wavefront walk below then decides, as in rounds 2-3.
                const bool par_wanted = nframes == 1 && g_opt.block_parallel.load() && sqy::lz4_linked_decode_parallel_possible(hc[1], total, block_bytes);
                const bool par_room = par_wanted && !ws->spec.ensure(((total * sizeof(uint16_t) + 255) & ~(uint64_t)255) + sqy::lz4_linked_decode_scan_scratch_bytes(hc[1]), true);
                if (par_wanted && !par_room) {
                    // (round-4 advice) the references need 2 bytes per decoded byte; without them the walk below decodes the frame -- said once
                    static std::atomic<bool> told{false};
                    if (!told.exchange(true))
                        std::fprintf(stderr, "[sqeazy]\t lz4: no HBM for the block-parallel decode's references (%llu MiB): the block-linked frame is decoded "
                                             "by one wavefront (same bytes, hundreds of times slower)\n", (unsigned long long)((total * sizeof(uint16_t)) >> 20));
                }
                if (par_room) {
                    // (no room for the references: the walk needs none)
                    hipError_t le;
                    {
                        ProfScope ps("lz4_linked_decode", stream, pend);
                        uint8_t* scan = static_cast<uint8_t*>(ws->spec.p) + ((total * sizeof(uint16_t) + 255) & ~(uint64_t)255);
                        le = sqy::launch_lz4_linked_decode_parallel(cur, blk, hc[1], out, static_cast<uint16_t*>(ws->spec.p), total, block_bytes,
                                                                    counts + 4, stream, g_opt.tail_scan.load() ? scan : nullptr);
                    }
                    if (le != hipSuccess) (void)hipGetLastError();                    // (e.g. no 128 KiB of LDS for the tails: the walk below)
                    SQY_HIP(hipMemcpyAsync(&bad, counts + 4, sizeof(bad), hipMemcpyDeviceToHost, stream));
                    SQY_HIP(hipStreamSynchronize(stream));
                    decoded = le == hipSuccess && bad == 0;
                    if (!decoded) SQY_HIP(hipMemsetAsync(counts + 4, 0, sizeof(uint32_t), stream));
                }
                if (!decoded) {
                    {
                        const bool side_ok = cx.ensure_side();           // (without it the copy simply follows on the same stream)
                        ProfScope ps("lz4_frames_decode", stream, pend);
                        SQY_HIP(sqy::launch_lz4_frames_decode(cur, blk, frame_first, nframes, out, total, chunk, block_bytes, hc[3], counts + 4, stream, side_ok ? cx.side : nullptr, cx.fork, cx.join,
                                                              remap, remap_bytes, g_opt.decode_two_waves.load() && hc[1] == nframes));
                    }
                    // (the decoder's verdict is read at the END of the call, with the call's last synchronisation: the stages in between are
                    // plain data movement and stay inside their buffers whatever the bytes are -- one host round trip less per decode)
                    lz4_flag = counts + 4;
                    lz4_flag_stage = (int)si;
                }
                cur = out; cur_bytes = total;
                if (remap) si -= 1;                                        // the frame_shuffle stage is done as well
                break;
            }
            case StageKind::bitswap1: {
                // quantiser right in front (on the encoder's side): the inverse transpose and the quantiser's look-up in one pass
                if (e_in == 1 && si >= 1 && pipe.stages[si - 1].kind == StageKind::quantiser && count_before[si - 1] == n_in &&
                    elem_before[si - 1] == 2) {
                    if (quantiser_lut_to_device(pipe.stages[si - 1], ws, stream)) return 1;
                    uint8_t* out16 = out_buf(si - 1, n_in * 2);
                    if (!out16) return 1;
                    if (sqy::bitswap1_u8_decode_lut_possible(cur, out16, n_in)) {
                        {
                            ProfScope ps("bitswap1_quantiser_decode", stream, pend);
                            SQY_HIP(sqy::launch_bitswap1_u8_decode_lut(cur, reinterpret_cast<uint16_t*>(out16), n_in,
                                                                       static_cast<const uint16_t*>(ws->small.p), stream));
                        }
                        cur = out16; cur_bytes = n_in * 2;
                        si -= 1;                                           // the quantiser stage is done as well
                        break;
                    }
                    // (odd sizes: the two stages one after the other; out16 is the quantiser's output buffer below)
                    if (si - 1 != 0) use_ping = !use_ping;                 // hand the buffer back to the quantiser stage
                }
                // diff3x3x1 as the pipeline's first stage (16-bit, the usual geometry): its inverse can only change the leading columns of
                // a row, so the planes are transposed straight into the volume and the inverse works there (round 4; before: into a
                // work buffer, from which the inverse copied every untouched column -- 0.75 ms of a 2 GiB slab's 1.1)
                uint8_t* out = nullptr;
                if (si == 1 && pipe.stages[0].kind == StageKind::diff3x3x1 && e_in == 2 && h.shape.size() == 3 && n_in == n &&
                    (reinterpret_cast<uintptr_t>(d_dst) & 15) == 0 && sqy::diff3x3x1_decode_chain_columns(h.shape[0], h.shape[1], h.shape[2], 2)) {
                    out = static_cast<uint8_t*>(d_dst);
                    diff_in_place = true;
                } else
                    out = out_buf(si, stage_in_bytes);
                if (!out) return 1;
                ProfScope ps("bitswap1_decode", stream, pend);
                SQY_HIP(sqy::launch_bitswap1_decode(cur, out, n_in, e_in, stream));
                cur = out; cur_bytes = stage_in_bytes;
                break;
            }
            case StageKind::raster_reorder: {
                if (h.shape.size() != 3) return 1;
                auto t = st.cfg.find("tile_size");
                const uint64_t ts = t != st.cfg.end() ? (uint64_t)std::atoi(t->second.c_str()) : 0;
                uint64_t Z, Y, X;
                stage_shape(si, n_in, Z, Y, X);                              // (tail filter: the sink's char stream)
                if (!sqy::raster_geometry_defined(Z, Y, X, ts, e_in)) {
                    std::fprintf(stderr, "[sqeazy]\t raster_reorder: tile_size %llu does not fit the shape\n", (unsigned long long)ts);
                    return stage_error(si);
                }
                uint8_t* out = out_buf(si, stage_in_bytes);
                if (!out) return 1;
                ProfScope ps("raster_reorder_decode", stream, pend);
                SQY_HIP(sqy::launch_raster_reorder(cur, out, Z, Y, X, ts, e_in, true, stream));
                cur = out; cur_bytes = stage_in_bytes;
                break;
            }
            case StageKind::pass_through:
                break;                                                          // pass_through_scheme_impl.hpp:81-95: bytes are the voxels
            case StageKind::zcurve_reorder: {
                if (h.shape.size() != 3) return stage_error(si);
                auto t = st.cfg.find("tile_size");
                const uint64_t ts = t != st.cfg.end() ? (uint64_t)std::atoi(t->second.c_str()) : 2;
                uint64_t Z, Y, X;
                stage_shape(si, n_in, Z, Y, X);
                if (!sqy::zcurve_geometry_defined(Z, Y, X, ts)) {
                    std::fprintf(stderr, "[sqeazy]\t zcurve_reorder: tile_size %llu does not fit the shape\n", (unsigned long long)ts);
                    return stage_error(si);
                }
                uint8_t* out = out_buf(si, stage_in_bytes);
                if (!out) return 1;
                ProfScope ps("zcurve_reorder_decode", stream, pend);
                SQY_HIP(sqy::launch_raster_reorder(cur, out, Z, Y, X, ts, e_in, true, stream));
                cur = out; cur_bytes = stage_in_bytes;
                break;
            }
            case StageKind::bitshuffle: {
                const int e_here = (sink_index >= 0 && (int)si > sink_index) ? 1 : e_in;      // tail filters work on the sink's bytes
                auto b = st.cfg.find("block_size");
                const uint64_t be = sqy::bitshuffle_block_elems(e_here, b != st.cfg.end() ? (uint64_t)std::atoi(b->second.c_str()) : 0);
                if (!be) return stage_error(si);
                uint8_t* out = out_buf(si, stage_in_bytes);
                if (!out) return 1;
                ProfScope ps("bitshuffle_decode", stream, pend);
                SQY_HIP(sqy::launch_bitshuffle(cur, out, stage_in_bytes / (uint64_t)e_here, e_here, be, true, stream));
                cur = out; cur_bytes = stage_in_bytes;
                break;
            }
            case StageKind::tile_shuffle: {
                if (h.shape.size() != 3) return stage_error(si);
                auto t = st.cfg.find("tile_size");
                const uint64_t ts = t != st.cfg.end() ? (uint64_t)std::atoi(t->second.c_str()) : 32;
                uint64_t Z, Y, X;
                stage_shape(si, n_in, Z, Y, X);
                if (!sqy::tile_shuffle_geometry_defined(Z, Y, X, ts)) {
                    std::fprintf(stderr, "[sqeazy]\t tile_shuffle: tile_size %llu does not divide the shape\n", (unsigned long long)ts);
                    return stage_error(si);
                }
                auto it = st.cfg.find("reorder_map");
                const uint64_t per_tile = ts * ts * ts, ntiles = n_in / per_tile, tile_bytes = per_tile * (uint64_t)e_in;
                if (it == st.cfg.end() || it->second.size() < 21) { std::fprintf(stderr, "[sqeazy]\t tile_shuffle: no reorder_map in the header\n"); return stage_error(si); }
                const std::vector<unsigned char> mapb = sqy::base64_decode(it->second.substr(10, it->second.size() - 21));
                if (mapb.size() != ntiles * 8) { std::fprintf(stderr, "[sqeazy]\t tile_shuffle: malformed reorder_map\n"); return stage_error(si); }
                // tile_shuffle_utils.hpp:473-482: encoded tile i goes to slot map[i], a later i wins, unnamed slots stay zero:
                // as a gather, slot t takes the LAST i that names it
                std::vector<uint64_t> src_of(ntiles, ~0ull);
                for (uint64_t i = 0; i < ntiles; ++i) {
                    uint64_t v; std::memcpy(&v, mapb.data() + 8 * i, 8);
                    if (v >= ntiles) { std::fprintf(stderr, "[sqeazy]\t tile_shuffle: reorder_map out of range\n"); return stage_error(si); }
                    src_of[v] = i;
                }
                if (ws->small.ensure(std::max<uint64_t>(ntiles * 8, 4096))) return 1;
                SQY_HIP(hipMemcpyAsync(ws->small.p, src_of.data(), ntiles * 8, hipMemcpyHostToDevice, stream));
                DevBuf& tb = use_ping ? ws->ping : ws->pong;                    // tile-major intermediate
                use_ping = !use_ping;
                if (tb.ensure(std::max<uint64_t>(stage_in_bytes, 16))) return 1;
                {
                    ProfScope ps("tile_unshuffle", stream, pend);
                    SQY_HIP(sqy::launch_frame_gather(cur, tb.p, ntiles, tile_bytes, static_cast<const uint64_t*>(ws->small.p), stream));
                }
                SQY_HIP(hipStreamSynchronize(stream));                          // src_of (host) is read by the async copy above
                uint8_t* out = out_buf(si, stage_in_bytes);
                if (!out) return 1;
                {
                    ProfScope ps("tile_scatter", stream, pend);
                    SQY_HIP(sqy::launch_raster_reorder(tb.p, out, Z, Y, X, ts, e_in, true, stream));
                }
                cur = out; cur_bytes = stage_in_bytes;
                break;
            }
            case StageKind::diff3x3x1: {
                if (h.shape.size() != 3) return 1;
                // as a tail filter (behind the sink) the stream is `char` and has the volume's shape only when the sink wrote one
                // byte per voxel (dynamic_pipeline.hpp:658-666); anything else the encoder refused
                const bool tail = sink_index >= 0 && (int)si > sink_index;
                if (tail && (n_in != n || e_in != 1)) return stage_error(si);
                uint8_t* out = out_buf(si, stage_in_bytes);
                if (!out) return 1;
                if (ws->lz4_scratch.ensure(sqy::diff3x3x1_decode_scratch_bytes(h.shape[2]))) return 1;
                void* left_tmp = nullptr;
                if (si == 0 && diff_in_place && cur == out) {              // the bit-plane inverse wrote the volume's own memory (above)
                    DevBuf& tb = use_ping ? ws->ping : ws->pong;
                    use_ping = !use_ping;
                    if (tb.ensure(std::max<uint64_t>(stage_in_bytes, 16))) return 1;
                    left_tmp = tb.p;
                }
                ProfScope ps("diff3x3x1_decode", stream, pend);
                const bool side_ok = cx.ensure_side();
                SQY_HIP(sqy::launch_diff3x3x1_decode(cur, out, h.shape[0], h.shape[1], h.shape[2], e_in, ws->lz4_scratch.p, stream, tail,
                                                     side_ok ? cx.side : nullptr, cx.fork, cx.join, left_tmp));
                cur = out; cur_bytes = stage_in_bytes;
                break;
            }
            case StageKind::quantiser: {
                if (quantiser_lut_to_device(st, ws, stream)) return 1;
                uint8_t* out = out_buf(si, stage_in_bytes);
                if (!out) return 1;
                {
                    ProfScope ps("quantiser_decode", stream, pend);
                    SQY_HIP(sqy::launch_quantiser_decode(cur, reinterpret_cast<uint16_t*>(out), n, static_cast<const uint16_t*>(ws->small.p), stream));
                }
                SQY_HIP(hipStreamSynchronize(stream));
                cur = out; cur_bytes = stage_in_bytes;
                break;
            }
            case StageKind::frame_shuffle: {
                uint64_t Z = 0, frame_bytes_dec = 0;
                bool permutation = true;
                if (const int rc = frame_shuffle_prepare(si, Z, frame_bytes_dec, permutation)) return rc;
                uint8_t* out = out_buf(si, stage_in_bytes);
                if (!out) return 1;
                // Frames with equal metrics share ONE source frame in the encoder (std::find, frame_shuffle_utils.hpp:158-161): the map then
                // names a frame twice and others not at all.  The reference's decode leaves the frames nobody names as the output
                // buffer had them (frame_shuffle_utils.hpp:337-344); here they come out as zeros, not as whatever the
                // workspace held.
                if (!permutation) { if (const int rc = zero_unnamed_places(out, frame_bytes_dec, stage_in_bytes)) return rc; }
                {
                    ProfScope ps("frame_scatter", stream, pend);
                    SQY_HIP(sqy::launch_frame_scatter(cur, out, Z, frame_bytes_dec, static_cast<const uint64_t*>(ws->small.p), stream));
                }
                cur = out; cur_bytes = stage_in_bytes;
                break;
            }
            default:
                return 1;
        }
    }
    if (cur != d_dst) SQY_HIP(hipMemcpyAsync(d_dst, cur, raw_bytes, hipMemcpyDeviceToDevice, stream));
    if (lz4_flag) SQY_HIP(hipMemcpyAsync(ws->pinned, lz4_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    SQY_HIP(hipStreamSynchronize(stream));
    if (g_prof_on.load()) prof_collect(cx.pending);
    if (lz4_flag && *static_cast<const uint32_t*>(ws->pinned)) {
        std::fprintf(stderr, "[sqy::lz4] corrupt LZ4 block, or a frame that does not decode to its share of the volume\n");
        return stage_error((size_t)lz4_flag_stage);
    }
    return 0;
}

int decode_from_host(const char* src, long srclength, char* dst, int elem_size)
{
    if (!src || !dst || srclength <= 0) return 1;
    const sqy::HeaderInfo h = sqy::header_unpack(src, src + srclength);
    if (!h.valid) { std::fprintf(stderr, "[sqeazy]\t unable to find a sqy header in the blob\n"); return 1; }
    if (!device_present()) { std::fprintf(stderr, "[sqeazy]\t no MI355X (HIP device) visible: sqeazy_amd has no CPU path\n"); return 1; }
    ContextLease lease;
    if (!lease.ctx) { std::fprintf(stderr, "[sqeazy]\t no usable HIP device\n"); return 1; }
    Workspace* ws = &lease.ctx->ws;
    hipStream_t stream = lease.ctx->own_stream();
    if (!stream) { std::fprintf(stderr, "[sqeazy]\t no HIP stream\n"); return 1; }
    uint64_t raw = 0;
    if (!header_shape_ok(h, (uint64_t)srclength, &raw)) return 1;          // untrusted input: before anything is allocated or uploaded
    if (ws->io_src.ensure(std::max<uint64_t>((uint64_t)srclength, 16)) || ws->io_dst.ensure(std::max<uint64_t>(raw, 16))) return 1;
    int dev_id = 0;
    SQY_HIP(hipGetDevice(&dev_id));
    if (!lease.ctx->stager.copy(ws->io_src.p, const_cast<char*>(src), (size_t)srclength, true, dev_id)) { std::fprintf(stderr, "[sqeazy]\t host to device transfer failed\n"); return 1; }
    const int rc = decode_on_device(*lease.ctx, ws->io_src.p, (uint64_t)srclength, ws->io_dst.p, raw, elem_size, stream);
    if (rc) return rc;
    SQY_HIP(hipStreamSynchronize(stream));
    if (!lease.ctx->stager.copy(ws->io_dst.p, dst, raw, false, dev_id)) { std::fprintf(stderr, "[sqeazy]\t device to host transfer failed\n"); return 1; }
    return 0;
}

int max_compressed_length(const char* pipeline, long pipeline_length, long* length, int elem_size, uint64_t raw_bytes)
{
    if (!pipeline || !length || pipeline_length < 0) return 1;
    const std::string s(pipeline, pipeline + pipeline_length);
    if (!Pipeline::supported(s, elem_size)) return 1;
    const Pipeline p = Pipeline::from_string(s, elem_size);
    if (p.stages.empty()) {
        std::fprintf(stderr, "[sqeazy]\t received %spipeline of size 0, cannot compite Max_Compressed_Length\n", p.name().c_str());
        return 1;
    }
    *length = (long)p.max_encoded_size(raw_bytes, elem_size);
    return 0;
}

// No C++ exception may cross the C-ABI (std::bad_alloc from a std::string / std::vector, anything a parser throws):
// every entry point runs inside one of these and turns an exception into the reference's generic error code.
template <class F>
int guarded(F&& f) noexcept
{
    try { return f(); }
    catch (const std::exception& e) { std::fprintf(stderr, "[sqeazy]\t %s\n", e.what()); return 1; }
    catch (...) { std::fprintf(stderr, "[sqeazy]\t unknown exception\n"); return 1; }
}
template <class F>
bool guarded_bool(F&& f) noexcept
{
    try { return f(); } catch (...) { return false; }
}

} // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

int SQY_Header_Size(const char* src, long* length)
{
    return guarded([&]() -> int {
    if (!src || !length) return 1;
    const sqy::HeaderInfo h = sqy::header_unpack(src, src + *length);
    *length = h.valid ? (long)h.size : 0;
    return 0;
    });
}

int SQY_Decompressed_NDims(const char* src, long* num)
{
    return guarded([&]() -> int {
    if (!src || !num) return 1;
    const sqy::HeaderInfo h = sqy::header_unpack(src, src + *num);
    *num = (long)h.shape.size();
    return 0;
    });
}

int SQY_Decompressed_Shape(const char* src, long* shape)
{
    return guarded([&]() -> int {
    if (!src || !shape) return 1;
    const sqy::HeaderInfo h = sqy::header_unpack(src, src + shape[0]);
    for (size_t i = 0; i < h.shape.size(); ++i) shape[i] = (long)h.shape[i];
    return 0;
    });
}

int SQY_Decompressed_Sizeof(const char* src, long* Sizeof)
{
    return guarded([&]() -> int {
    if (!src || !Sizeof) return 1;
    const sqy::HeaderInfo h = sqy::header_unpack(src, src + *Sizeof);
    *Sizeof = h.valid ? h.elem_size() : 0;
    return 0;
    });
}

int SQY_Decompressed_Length(const char* data, long* length)
{
    return guarded([&]() -> int {
    if (!data || !length) return 1;
    const sqy::HeaderInfo h = sqy::header_unpack(data, data + *length);
    uint64_t n = 1;
    for (uint64_t d : h.shape) n *= d;
    *length = h.valid ? (long)(n * (uint64_t)h.elem_size()) : 0;
    return 0;
    });
}

int SQY_Version_Triple(int* version)
{
    return guarded([&]() -> int {
    if (!version) return 1;
    version[0] = sqy::kVersionTriple[0];
    version[1] = sqy::kVersionTriple[1];
    version[2] = sqy::kVersionTriple[2];
    return 0;
    });
}

int SQY_PipelineEncode_UI8(const char* pipeline, const char* src, long* shape, unsigned shape_size, char* dst, long* dstlength, int nthreads)
{
    return guarded([&]() -> int {
    return encode_from_host(pipeline, src, shape, shape_size, 1, dst, dstlength, nthreads);
    });
}

int SQY_PipelineEncode_UI16(const char* pipeline, const char* src, long* shape, unsigned shape_size, char* dst, long* dstlength, int nthreads)
{
    return guarded([&]() -> int {
    return encode_from_host(pipeline, src, shape, shape_size, 2, dst, dstlength, nthreads);
    });
}

int SQY_Pipeline_Max_Compressed_Length_UI8(const char* pipeline, long pipeline_length, long* length)
{
    return guarded([&]() -> int {
    return length ? max_compressed_length(pipeline, pipeline_length, length, 1, (uint64_t)*length) : 1;
    });
}

int SQY_Pipeline_Max_Compressed_Length_UI16(const char* pipeline, long pipeline_length, long* length)
{
    return guarded([&]() -> int {
    return length ? max_compressed_length(pipeline, pipeline_length, length, 2, (uint64_t)*length) : 1;
    });
}

static int max_len_3d(const char* pipeline, long* shape, unsigned shape_size, long* length, int elem)
{
    if (!shape || !length) return 1;
    long n = 1;
    for (unsigned i = 0; i < shape_size; ++i) n *= shape[i];   // std::accumulate(..., 1, multiplies<long>) (sqeazy.cpp:195)
    return max_compressed_length(pipeline, *length, length, elem, (uint64_t)n * (uint64_t)elem);
}

int SQY_Pipeline_Max_Compressed_Length_3D_UI8(const char* pipeline, long* shape, unsigned shape_size, long* length)
{
    return guarded([&]() -> int {
    return max_len_3d(pipeline, shape, shape_size, length, 1);
    });
}

int SQY_Pipeline_Max_Compressed_Length_3D_UI16(const char* pipeline, long* shape, unsigned shape_size, long* length)
{
    return guarded([&]() -> int {
    return max_len_3d(pipeline, shape, shape_size, length, 2);
    });
}

bool SQY_Pipeline_Possible_UI16(const char* s) { return guarded_bool([&]() -> bool { return s && Pipeline::supported(s, 2); }); }
bool SQY_Pipeline_Possible_UI8(const char* s) { return guarded_bool([&]() -> bool { return s && Pipeline::supported(s, 1); }); }
bool SQY_Pipeline_Possible(const char* s, int sizeofpixel)
{
    return guarded_bool([&]() -> bool {
    if (!s) return false;
    if (sizeofpixel == 2) return Pipeline::supported(s, 2);
    if (sizeofpixel == 1) return Pipeline::supported(s, 1);
    return false;
    });
}

int SQY_Decode_UI16(const char* src, long srclength, char* dst, int nthreads)
{
    return guarded([&]() -> int {
    (void)nthreads;   // the layout is read from the blob; the GPU decodes every frame in parallel
    return decode_from_host(src, srclength, dst, 2);
    });
}

int SQY_Decode_UI8(const char* src, long srclength, char* dst, int nthreads)
{
    return guarded([&]() -> int {
    (void)nthreads;
    return decode_from_host(src, srclength, dst, 1);
    });
}

int SQYAMD_PipelineEncode_UI16_Device(const char* pipeline, const void* d_src, const long* shape, unsigned shape_size, void* d_dst,
                                      long dst_capacity, long* dstlength, int nthreads, void* hip_stream)
{
    return guarded([&]() -> int {
    ContextLease lease;
    if (!lease.ctx) { std::fprintf(stderr, "[sqeazy]\t no usable HIP device\n"); return 1; }
    return encode_on_device(*lease.ctx, pipeline, d_src, shape, shape_size, 2, d_dst, (uint64_t)std::max(dst_capacity, 0l), dstlength, nthreads,
                            static_cast<hipStream_t>(hip_stream));
    });
}

int SQYAMD_PipelineEncode_UI8_Device(const char* pipeline, const void* d_src, const long* shape, unsigned shape_size, void* d_dst,
                                     long dst_capacity, long* dstlength, int nthreads, void* hip_stream)
{
    return guarded([&]() -> int {
    ContextLease lease;
    if (!lease.ctx) { std::fprintf(stderr, "[sqeazy]\t no usable HIP device\n"); return 1; }
    return encode_on_device(*lease.ctx, pipeline, d_src, shape, shape_size, 1, d_dst, (uint64_t)std::max(dst_capacity, 0l), dstlength, nthreads,
                            static_cast<hipStream_t>(hip_stream));
    });
}

int SQYAMD_PipelineEncode_UI16_DeviceAt(const char* pipeline, const void* d_src, const long* shape, unsigned shape_size, void* d_dst,
                                        long dst_capacity, long* dstoffset, long* dstlength, int nthreads, void* hip_stream)
{
    return guarded([&]() -> int {
    if (!dstoffset) return 1;
    ContextLease lease;
    if (!lease.ctx) { std::fprintf(stderr, "[sqeazy]\t no usable HIP device\n"); return 1; }
    return encode_on_device(*lease.ctx, pipeline, d_src, shape, shape_size, 2, d_dst, (uint64_t)std::max(dst_capacity, 0l), dstlength, nthreads,
                            static_cast<hipStream_t>(hip_stream), dstoffset);
    });
}

int SQYAMD_PipelineEncode_UI16_DeviceAt_Frames(const char* pipeline, const void* d_src, const long* shape, unsigned shape_size, void* d_dst,
                                               long dst_capacity, long* dstoffset, long* dstlength, int nthreads, void* hip_stream, int every,
                                               long* frame_offsets, int max_entries, int* count)
{
    return guarded([&]() -> int {
    if (!dstoffset || every <= 0 || !frame_offsets || !count) return 1;
    ContextLease lease;
    if (!lease.ctx) { std::fprintf(stderr, "[sqeazy]\t no usable HIP device\n"); return 1; }
    FrameQuery fq;
    fq.every = every; fq.offsets = frame_offsets; fq.max_entries = max_entries;
    const int rc = encode_on_device(*lease.ctx, pipeline, d_src, shape, shape_size, 2, d_dst, (uint64_t)std::max(dst_capacity, 0l), dstlength, nthreads,
                                    static_cast<hipStream_t>(hip_stream), dstoffset, &fq);
    *count = fq.count;
    return rc;
    });
}

int SQYAMD_PipelineEncode_UI8_DeviceAt_Frames(const char* pipeline, const void* d_src, const long* shape, unsigned shape_size, void* d_dst,
                                              long dst_capacity, long* dstoffset, long* dstlength, int nthreads, void* hip_stream, int every,
                                              long* frame_offsets, int max_entries, int* count)
{
    return guarded([&]() -> int {
    if (!dstoffset || every <= 0 || !frame_offsets || !count) return 1;
    ContextLease lease;
    if (!lease.ctx) { std::fprintf(stderr, "[sqeazy]\t no usable HIP device\n"); return 1; }
    FrameQuery fq;
    fq.every = every; fq.offsets = frame_offsets; fq.max_entries = max_entries;
    const int rc = encode_on_device(*lease.ctx, pipeline, d_src, shape, shape_size, 1, d_dst, (uint64_t)std::max(dst_capacity, 0l), dstlength, nthreads,
                                    static_cast<hipStream_t>(hip_stream), dstoffset, &fq);
    *count = fq.count;
    return rc;
    });
}

int SQYAMD_PipelineEncode_UI8_DeviceAt(const char* pipeline, const void* d_src, const long* shape, unsigned shape_size, void* d_dst,
                                       long dst_capacity, long* dstoffset, long* dstlength, int nthreads, void* hip_stream)
{
    return guarded([&]() -> int {
    if (!dstoffset) return 1;
    ContextLease lease;
    if (!lease.ctx) { std::fprintf(stderr, "[sqeazy]\t no usable HIP device\n"); return 1; }
    return encode_on_device(*lease.ctx, pipeline, d_src, shape, shape_size, 1, d_dst, (uint64_t)std::max(dst_capacity, 0l), dstlength, nthreads,
                            static_cast<hipStream_t>(hip_stream), dstoffset);
    });
}

// A volume as `nslabs` independent z-slab blobs, `inflight` slab calls at a time on library-owned streams (one host thread, context
// and stream per call in flight -- what three caller threads would do, available to a plain C caller with one call).
static int encode_slabs(const char* pipeline, const void* d_src, const long* shape, unsigned rank, int elem_size, int nslabs, void* d_dst,
                        long slab_capacity, long* offsets, long* lengths, int nthreads, int inflight)
{
    if (!pipeline || !d_src || !shape || !d_dst || !offsets || !lengths || rank == 0 || nslabs <= 0 || slab_capacity <= 0) return 1;
    for (unsigned i = 0; i < rank; ++i) if (shape[i] <= 0) return 1;
    if ((long)nslabs > shape[0]) { std::fprintf(stderr, "[sqeazy]\t more slabs (%d) than frames (%ld)\n", nslabs, shape[0]); return 1; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { std::fprintf(stderr, "[sqeazy]\t no usable HIP device\n"); return 1; }
    uint64_t per_frame = (uint64_t)elem_size;
    for (unsigned i = 1; i < rank; ++i) per_frame *= (uint64_t)shape[i];
    const long base = shape[0] / nslabs, rem = shape[0] % nslabs;           // the first Z % nslabs slabs get one frame more
    if (inflight <= 0) inflight = 3;
    if (inflight > nslabs) inflight = nslabs;
    if (inflight > (int)kMaxCtxPerDev) inflight = (int)kMaxCtxPerDev;
    std::atomic<int> first_error(0);
    for (int i = 0; i < nslabs; ++i) { offsets[i] = 0; lengths[i] = 0; }      // (defined whatever happens below)
    // The slab calls run on streams of the library's own (non-blocking ones: nothing orders them behind the default stream by itself), the
    // caller passes none -- so what the caller has queued on the DEFAULT stream up to now (the kernel that makes d_src, a fill of d_dst: a
    // framework's allocations and copies usually live there) is put in front of them here: an event on the default stream, waited for by
    // every worker's stream.  Found by the full-size slab test of round 6, whose fill of d_dst overtook the first slabs' transposes and
    // was parsed in their place.  Work on OTHER streams of the caller has to be complete when the call is made (include/sqeazy_amd.h).
    hipEvent_t front = nullptr;
    if (hipEventCreateWithFlags(&front, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); front = nullptr; }
    struct EventGuard { hipEvent_t& e; ~EventGuard() { if (e) (void)hipEventDestroy(e); } } front_guard{front};
    if (!front || hipEventRecord(front, nullptr) != hipSuccess) {            // (no event to be had: wait here for everything instead)
        (void)hipGetLastError();
        if (front) { (void)hipEventDestroy(front); front = nullptr; }
        if (hipDeviceSynchronize() != hipSuccess) return 1;
    }
    auto work = [&](int t) {
        if (hipSetDevice(dev) != hipSuccess) { int z = 0; first_error.compare_exchange_strong(z, 1); return; }
        bool ordered = front == nullptr;
        for (int i = t; i < nslabs && first_error.load() == 0; i += inflight) {
            const long z0 = (long)i * base + std::min<long>(i, rem), nz = base + (i < rem ? 1 : 0);
            std::vector<long> shp(shape, shape + rank);
            shp[0] = nz;
            long at = 0, len = 0;
            int rc;
            {
                ContextLease lease;
                if (!lease.ctx) rc = 1;
                else {
                    hipStream_t stream = lease.ctx->own_stream();
                    // (every lease may be another context with another stream: each call waits; an event that has completed costs nothing)
                    if (stream && !ordered && hipStreamWaitEvent(stream, front, 0) != hipSuccess) { (void)hipGetLastError(); stream = nullptr; }
                    rc = stream ? encode_on_device(*lease.ctx, pipeline, static_cast<const char*>(d_src) + (uint64_t)z0 * per_frame, shp.data(), rank,
                                                   elem_size, static_cast<char*>(d_dst) + (uint64_t)i * (uint64_t)slab_capacity,
                                                   (uint64_t)slab_capacity, &len, nthreads, stream, &at)
                                : 1;
                }
            }
            if (rc) { int z = 0; first_error.compare_exchange_strong(z, rc); return; }
            offsets[i] = (long)((uint64_t)i * (uint64_t)slab_capacity) + at;
            lengths[i] = len;
        }
    };
    // no exception leaves a worker (a thrown std::bad_alloc etc. becomes the call's error code), and the threads are joined on every
    // way out of this function -- they hold references to its locals (round-3 advice)
    auto worker = [&](int t) {
        try { work(t); }
        catch (const std::exception& e) { std::fprintf(stderr, "[sqeazy]\t slab worker: %s\n", e.what()); int z = 0; first_error.compare_exchange_strong(z, 1); }
        catch (...) { int z = 0; first_error.compare_exchange_strong(z, 1); }
    };
    std::vector<std::thread> th;
    struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{th};
    try {
        th.reserve((size_t)inflight);
        for (int t = 1; t < inflight; ++t) th.emplace_back(worker, t);
    } catch (...) {                                                            // (no thread to be had: the call fails, the threads that did start are joined)
        int z = 0; first_error.compare_exchange_strong(z, 1);
    }
    worker(0);
    for (auto& x : th) x.join();
    return first_error.load();
}

int SQYAMD_PipelineEncode_Slabs_UI16_Device(const char* pipeline, const void* d_src, const long* shape, unsigned shape_size, int nslabs,
                                            void* d_dst, long slab_capacity, long* offsets, long* lengths, int nthreads, int inflight)
{
    return guarded([&]() -> int {
    return encode_slabs(pipeline, d_src, shape, shape_size, 2, nslabs, d_dst, slab_capacity, offsets, lengths, nthreads, inflight);
    });
}

int SQYAMD_PipelineEncode_Slabs_UI8_Device(const char* pipeline, const void* d_src, const long* shape, unsigned shape_size, int nslabs,
                                           void* d_dst, long slab_capacity, long* offsets, long* lengths, int nthreads, int inflight)
{
    return guarded([&]() -> int {
    return encode_slabs(pipeline, d_src, shape, shape_size, 1, nslabs, d_dst, slab_capacity, offsets, lengths, nthreads, inflight);
    });
}

int SQYAMD_PipelineEncode_UI16_Cap(const char* pipeline, const char* src, long* shape, unsigned shape_size, char* dst,
                                   long dst_capacity, long* dstlength, int nthreads)
{
    return guarded([&]() -> int {
    return encode_from_host(pipeline, src, shape, shape_size, 2, dst, dstlength, nthreads, std::max(dst_capacity, 0l));
    });
}

int SQYAMD_PipelineEncode_UI8_Cap(const char* pipeline, const char* src, long* shape, unsigned shape_size, char* dst,
                                  long dst_capacity, long* dstlength, int nthreads)
{
    return guarded([&]() -> int {
    return encode_from_host(pipeline, src, shape, shape_size, 1, dst, dstlength, nthreads, std::max(dst_capacity, 0l));
    });
}

int SQYAMD_Decode_UI16_Device(const void* d_src, long srclength, void* d_dst, long dst_capacity, void* hip_stream)
{
    return guarded([&]() -> int {
    ContextLease lease;
    if (!lease.ctx) { std::fprintf(stderr, "[sqeazy]\t no usable HIP device\n"); return 1; }
    return decode_on_device(*lease.ctx, d_src, (uint64_t)std::max(srclength, 0l), d_dst, (uint64_t)std::max(dst_capacity, 0l), 2, static_cast<hipStream_t>(hip_stream));
    });
}

int SQYAMD_Decode_UI8_Device(const void* d_src, long srclength, void* d_dst, long dst_capacity, void* hip_stream)
{
    return guarded([&]() -> int {
    ContextLease lease;
    if (!lease.ctx) { std::fprintf(stderr, "[sqeazy]\t no usable HIP device\n"); return 1; }
    return decode_on_device(*lease.ctx, d_src, (uint64_t)std::max(srclength, 0l), d_dst, (uint64_t)std::max(dst_capacity, 0l), 1, static_cast<hipStream_t>(hip_stream));
    });
}

void SQYAMD_Profile_Enable(int enable)
{
    g_prof_on.store(enable != 0);
}

void SQYAMD_Profile_Reset(void)
{
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof.clear();
}

const char* SQYAMD_Profile_Get(int i, double* total_ms, long* launches)
{
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (i < 0 || (size_t)i >= g_prof.size()) return nullptr;
    if (total_ms) *total_ms = g_prof[i].ms;
    if (launches) *launches = g_prof[i].launches;
    return g_prof[i].name.c_str();
}

int SQYAMD_Set_Option(const char* name, long value)
{
    std::atomic<long>* o = g_opt.find(name);
    if (!o) return 1;
    if (o == &g_opt.block_parallel_warmup) { if (value < 0 || value > kWarmupMax) return 1; }
    else if (o == &g_opt.transpose_blocks_per_cu) { if (value < 1 || value > 64) return 1; sqy::set_bitswap1_blocks_per_cu(value); }
    else if (value != 0 && value != 1) return 1;
    o->store(value);
    return 0;
}

long SQYAMD_Get_Option(const char* name)
{
    std::atomic<long>* o = g_opt.find(name);
    return o ? o->load() : -1;
}

void SQYAMD_Release_Workspace(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    for (auto& c : g_pool[dev]) if (!c->busy) c->ws.release_buffers();
}

int SQYAMD_Header_Pipeline(const char* src, long srclength, char* out, long* outlength)
{
    return guarded([&]() -> int {
    if (!src || !outlength || srclength <= 0) return 1;
    const sqy::HeaderInfo h = sqy::header_unpack(src, src + srclength);
    if (!h.valid) return 1;
    const long need = (long)h.pipename.size() + 1;
    const long have = out ? *outlength : 0;
    *outlength = need;
    if (!out) return 0;                          // size query
    if (have < need) return 1;
    std::memcpy(out, h.pipename.c_str(), (size_t)need);
    return 0;
    });
}

int SQYAMD_Header_Build(const char* pipeline, int sizeof_voxel, const long* shape, unsigned shape_size, long encoded_bytes,
                        char* out, long* outlength)
{
    return guarded([&]() -> int {
    if (!pipeline || !shape || !outlength || shape_size == 0 || (sizeof_voxel != 1 && sizeof_voxel != 2) || encoded_bytes < 0) return 1;
    try {
        if (!sqy::Pipeline::supported(pipeline, sizeof_voxel)) return 1;
        const sqy::Pipeline p = sqy::Pipeline::from_string(pipeline, sizeof_voxel);
        std::vector<uint64_t> shp(shape, shape + shape_size);
        // what one encode call can have produced: < 2^31 voxels, at most INT_MAX payload bytes (decode refuses anything else)
        uint64_t nvox = 1;
        for (uint64_t v : shp) {
            if ((long)v <= 0 || v >= ((uint64_t)1 << 31)) return 1;
            nvox *= v;
            if (nvox >= ((uint64_t)1 << 31)) return 1;
        }
        if (encoded_bytes > (long)INT_MAX) return 1;
        const std::string hdr = sqy::header_pack(sizeof_voxel, false, shp, p.name(), (uint64_t)encoded_bytes);
        const long need = (long)hdr.size();
        const long have = out ? *outlength : 0;
        *outlength = need;
        if (!out) return 0;                      // size query
        if (have < need) return 1;
        std::memcpy(out, hdr.data(), hdr.size());
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "[sqeazy]\t %s\n", e.what());
        return 1;
    }
    });
}

const char* SQYAMD_Version(void) { return "sqeazy_amd 0.1.0 (gfx950, sqy header 0.5.2)"; }

} // extern "C"
