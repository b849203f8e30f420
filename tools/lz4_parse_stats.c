// tools/lz4_parse_stats.c -- CPU model of liblz4 1.9.3's fast parse (byU32, hashLog 12, acceleration 1, capacity n - 1) with counters
// for how the GPU kernel's paths would be exercised on a 256 KiB chunk: sequences, where their match was found (probe index since
// the last match), how many 64-probe batches found nothing and at which step, match lengths, candidate distances.
//   gcc -O2 tools/lz4_parse_stats.c -o /tmp/lz4_parse_stats && /tmp/lz4_parse_stats chunk.bin [chunk2.bin ...]
// Not part of the product, not part of the oracle (round 6: what the noise planes and plane 8 of the bench stack cost and why).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t hash5(const uint8_t* p) { return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> 52); }

typedef struct {
    long seqs, probes, hit_lean, hit_u15_63, hit_later, batches_nohit_lt960, batches_nohit_ge960, batches_hit_lt960, batches_hit_ge960;
    long ml_lt12, ml_lt64, ml_lt1024, ml_ge1024, far8k, tooFar, zero_lit, out;
    long hit_first64_after_reset; long co_at; long shorts_max; long def_at[5];
} stats_t;

static int parse(const uint8_t* src, int n, stats_t* s)
{
    static uint32_t table[4096];
    memset(table, 0, sizeof table);
    memset(s, 0, sizeof *s);
    if (n < 13) return 0;
    const uint8_t* ip = src; const uint8_t* anchor = src;
    const uint8_t* const end = src + n; const uint8_t* const mflimitPlusOne = end - 12 + 1; const uint8_t* const matchlimit = end - 5;
    long op = 0; const long olimit = n - 1;
    table[hash5(ip)] = 0; ip++;
    uint32_t forwardH = hash5(ip);
    for (;;) {
        const uint8_t* match; const uint8_t* forwardIp = ip; int step = 1; int searchMatchNb = 1 << 6; long u = 1;   // u: unified probe index (0 = test-next-position)
        long u_batch_start = 0;
        do {
            const uint32_t h = forwardH; const uint32_t cur = (uint32_t)(forwardIp - src); const uint32_t mi = table[h];
            ip = forwardIp; forwardIp += step; step = searchMatchNb++ >> 6;
            if (forwardIp > mflimitPlusOne) goto last;
            forwardH = hash5(forwardIp); table[h] = cur; s->probes++;
            match = src + mi;
            if (mi + 65535 < cur) { s->tooFar++; u++; continue; }
            if (rd32(match) == rd32(ip)) break;
            u++;
        } while (1);
        (void)u_batch_start;
        // where was it found: u counts probes of this search (1-based here; the GPU's lean loop covers unified indices 0..14)
        if (u <= 14) s->hit_lean++; else if (u <= 63) s->hit_u15_63++; else s->hit_later++;
        {   // batches of 64 probes this search went through without a hit (the GPU's generic / no-hit batches): unified index U = 0, 64, 128 ..
            const long nb = u / 64;                                   // full batches in front of the one that hit
            for (long b = 0; b < nb; ++b) { if (64 * b >= 960) s->batches_nohit_ge960++; else s->batches_nohit_lt960++; }
            if (u > 14) { if (64 * nb >= 960) s->batches_hit_ge960++; else s->batches_hit_lt960++; }
        }
        while (ip > anchor && match > src && ip[-1] == match[-1]) { ip--; match--; }
        {
            const long lit = ip - anchor;
            if (op + lit + (2 + 1 + 5) + lit / 255 > olimit) { s->out = op; return 0; }
            if (!s->co_at && lit >= 256 && (anchor - src) >= 32768 && op >= (anchor - src) + 256) s->co_at = (anchor - src) + 1;   // the count-only rule of round 6 (measured, not kept)
            { static const long cp[5] = {16384, 32768, 65536, 131072, 196608}; for (int i = 0; i < 5; ++i) if (!s->def_at[i] && (anchor - src) >= cp[i]) s->def_at[i] = op - (anchor - src) + 1000000; }
            op += 1 + lit + (lit >= 15 ? (lit - 15) / 255 + 1 : 0);
        }
    next_match:
        {
            if ((ip - match) > 8192 - 16) s->far8k++;
            const uint8_t* p = ip + 4; const uint8_t* m = match + 4;
            while (p < matchlimit && *p == *m) { p++; m++; }
            const long ml = p - (ip + 4);
            ip += ml + 4;
            if (op + (1 + 5) + (ml + 240) / 255 > olimit) { s->out = op; return 0; }
            op += 2 + (ml >= 15 ? (ml - 15) / 255 + 1 : 0);
            s->seqs++;
            if (ml + 4 < 16) s->ml_lt12++; else if (ml + 4 < 64) s->ml_lt64++; else if (ml + 4 < 1024) s->ml_lt1024++; else s->ml_ge1024++;
        }
        anchor = ip;
        if (ip >= mflimitPlusOne) break;
        table[hash5(ip - 2)] = (uint32_t)(ip - 2 - src);
        {
            const uint32_t h = hash5(ip); const uint32_t cur = (uint32_t)(ip - src); const uint32_t mi = table[h];
            table[h] = cur; match = src + mi; s->probes++;
            if (mi + 65535 >= cur && rd32(match) == rd32(ip)) { op++; s->zero_lit++; s->hit_lean++; goto next_match; }
        }
        forwardH = hash5(++ip);
    }
last:
    {
        const long lastRun = end - anchor;
        if (op + lastRun + 1 + (lastRun + 255 - 15) / 255 > olimit) { s->out = op; return 0; }
        op += 1 + lastRun + (lastRun >= 15 ? (lastRun - 15) / 255 + 1 : 0);
    }
    s->out = op;
    return (int)op;
}

int main(int argc, char** argv)
{
    for (int a = 1; a < argc; ++a) {
        FILE* f = fopen(argv[a], "rb");
        if (!f) { perror(argv[a]); return 1; }
        static uint8_t buf[1 << 22];
        const int n = (int)fread(buf, 1, sizeof buf, f);
        fclose(f);
        for (int off = 0; off + 262144 <= n || off == 0; off += 262144) {
            const int len = n - off < 262144 ? n - off : 262144;
            stats_t s;
            const int c = parse(buf + off, len, &s);
            printf("%s +%d: csize %d (%s) seqs %ld probes %ld | hits: lean(u<15) %ld (zero-lit %ld) u15..63 %ld later %ld | no-hit batches U<960 %ld U>=960 %ld, hit batches U<960 %ld U>=960 %ld | "
                   "ml <16 %ld <64 %ld <1024 %ld >=1024 %ld | offset>8K %ld tooFar %ld | count-only from %ld | output - input at 16K %ld 32K %ld 64K %ld 128K %ld 192K %ld end %ld\n",
                   argv[a], off, c, c ? "compressed" : "stored", s.seqs, s.probes, s.hit_lean, s.zero_lit, s.hit_u15_63, s.hit_later, s.batches_nohit_lt960,
                   s.batches_nohit_ge960, s.batches_hit_lt960, s.batches_hit_ge960, s.ml_lt12, s.ml_lt64, s.ml_lt1024, s.ml_ge1024, s.far8k, s.tooFar, s.co_at, s.def_at[0] - 1000000, s.def_at[1] - 1000000, s.def_at[2] - 1000000, s.def_at[3] - 1000000, s.def_at[4] - 1000000, s.out - (long)len);
        }
    }
    return 0;
}
