/* CPU experiment (analysis tool, not product, not oracle): does a liblz4 greedy parse that is started in the middle of a
 * 256 KiB chunk -- from an empty table, some warm-up in front of a segment border -- converge to the true parse of the chunk?
 *
 *   gcc -O2 -o /tmp/segconv tools/segment_convergence.c && /tmp/segconv plane.bin [chunk=262144] [seg=65536] [warm=65536]
 *
 * For every chunk of the file: the true parse (fresh table, position 0), then for every segment border s = k*seg a speculative
 * parse from s - warm.  Reported per border: the first anchor >= s both parses share, and from there on how many of the
 * speculative sequences of [s, s + seg) equal the true ones (same anchor, match start, offset, length), how many runs of
 * differing sequences there are and where the last differing sequence ends.  A segment "converges" when nothing differs
 * behind the first shared anchor.  The parse is liblz4 1.9.3's LZ4_compress_generic (byU32, acceleration 1) as restated in
 * SURVEY.md Appendix B. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t anchor, mstart, off, mlen; } seq_t;

static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t h5(const uint8_t* p) { return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> 52); }

/* parse src[0..n) as liblz4 would from `start` on with the table given (positions relative to src); sequences -> out */
static size_t parse(const uint8_t* src, uint32_t n, uint32_t start, uint32_t* table, seq_t* out, size_t cap, uint32_t stop_at)
{
    size_t ns = 0;
    const uint8_t* ip = src + start;
    const uint8_t* anchor = ip;
    const uint8_t* iend = src + n;
    const uint8_t* mfl = iend - 12 + 1;
    const uint8_t* mlimit = iend - 5;
    if (n - start < 13) return 0;
    table[h5(ip)] = start;
    ip++;
    uint32_t fh = h5(ip);
    for (;;) {
        const uint8_t* match;
        {
            const uint8_t* fip = ip;
            int step = 1, nb = 1 << 6;
            for (;;) {
                uint32_t h = fh, cur = (uint32_t)(fip - src), mi = table[h];
                ip = fip;
                fip += step;
                step = nb++ >> 6;
                if (fip > mfl) return ns;
                match = src + mi;
                fh = h5(fip);
                table[h] = cur;
                if (mi + 65535 < cur) continue;
                if (rd32(match) == rd32(ip)) break;
            }
        }
        while (ip > anchor && match > src && ip[-1] == match[-1]) { ip--; match--; }
        uint32_t a = (uint32_t)(anchor - src);
        for (;;) {
            const uint8_t* pi = ip + 4;
            const uint8_t* pm = match + 4;
            while (pi < mlimit && *pi == *pm) { pi++; pm++; }
            if (ns < cap) { out[ns].anchor = a; out[ns].mstart = (uint32_t)(ip - src); out[ns].off = (uint32_t)(ip - match); out[ns].mlen = (uint32_t)(pi - ip); }
            ns++;
            ip = pi;
            anchor = ip;
            if (ip >= mfl) return ns;
            table[h5(ip - 2)] = (uint32_t)(ip - 2 - src);
            if ((uint32_t)(ip - src) == stop_at) return ns;      /* the state "at anchor stop_at": ip-2 entered, ip not yet probed */
            uint32_t h = h5(ip), cur = (uint32_t)(ip - src), mi = table[h];
            match = src + mi;
            table[h] = cur;
            if (mi + 65535 >= cur && rd32(match) == rd32(ip)) { a = cur; continue; }
            break;
        }
        fh = h5(++ip);
    }
}

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    uint32_t chunk = argc > 2 ? atoi(argv[2]) : 262144, seg = argc > 3 ? atoi(argv[3]) : 65536, warm = argc > 4 ? atoi(argv[4]) : 65536;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    long total = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* buf = malloc(total + 16);
    if (fread(buf, 1, total, f) != (size_t)total) return 1;
    fclose(f);
    seq_t* T = malloc(sizeof(seq_t) * chunk);
    seq_t* S = malloc(sizeof(seq_t) * chunk);
    int32_t* at = malloc(sizeof(int32_t) * (chunk + 1));
    static uint32_t table[4096];
    long n_seg = 0, n_conv = 0, n_nosync = 0, n_hand = 0, n_hand_ok = 0, n_hand_table = 0, n_hand_anchor = 0;
    double sum_sync = 0, sum_bad = 0, sum_seqs = 0, sum_runs = 0;
    for (long c0 = 0; c0 + chunk <= total; c0 += chunk) {
        const uint8_t* src = buf + c0;
        memset(table, 0, sizeof table);
        size_t nt = parse(src, chunk, 0, table, T, chunk, 0xffffffffu);
        memset(at, -1, sizeof(int32_t) * (chunk + 1));
        for (size_t i = 0; i < nt; ++i) at[T[i].anchor] = (int32_t)i;
        for (uint32_t s = seg; s < chunk; s += seg) {
            uint32_t st = s > warm ? s - warm : 0;
            memset(table, 0, sizeof table);
            size_t nsp = parse(src, chunk, st, table, S, chunk, 0xffffffffu);
            /* first shared anchor >= s */
            size_t i = 0;
            while (i < nsp && (S[i].anchor < s || at[S[i].anchor] < 0)) ++i;
            uint32_t e = s + seg;
            n_seg++;
            if (i == nsp || S[i].anchor >= e) { n_nosync++; printf("chunk %ld seg %u: no shared anchor\n", c0 / chunk, s / seg); continue; }
            uint32_t sync = S[i].anchor;
            {   /* at which shared anchor are the two tables equivalent (equal on every entry still in reach)? */
                static uint32_t ta[4096], tb[4096];
                size_t k = i; int tries = 0; int found = -1; uint32_t where = 0; int ndiff0 = -1;
                for (; k < nsp && S[k].anchor < e && tries < 400; ++k) {
                    if (at[S[k].anchor] < 0 || S[k].anchor == 0) continue;
                    uint32_t A = S[k].anchor;
                    memset(ta, 0, sizeof ta); memset(tb, 0, sizeof tb);
                    parse(src, chunk, 0, ta, NULL, 0, A);
                    parse(src, chunk, st, tb, NULL, 0, A);
                    int nd = 0;
                    for (int h = 0; h < 4096; ++h) {
                        int ra = ta[h] + 65535u >= A && ta[h] != 0, rb = tb[h] + 65535u >= A && tb[h] != 0;   /* (position 0: "empty" and a real candidate alike, both tables started from zeros) */
                        if (ra != rb || (ra && ta[h] != tb[h])) nd++;
                    }
                    if (tries == 0) ndiff0 = nd;
                    tries++;
                    if (!nd) { found = tries; where = A; break; }
                }
                if (found > 0) printf("   tables equivalent at shared anchor #%d (+%u); %d buckets differ at the first\n", found, where - s, ndiff0);
                else printf("   tables never equivalent in %d shared anchors; %d buckets differ at the first\n", tries, ndiff0);
            }
            {   /* the hand-over as a kernel would do it without looking for a SHARED anchor: the wavefront in front stops at ITS first anchor at
                   or behind the border, the one behind starts counting at ITS first anchor at or behind the border -- the same anchor? */
                size_t a = 0; while (a < nt && T[a].anchor < s) ++a;
                size_t b = 0; while (b < nsp && S[b].anchor < s) ++b;
                n_hand++;
                if (a < nt && b < nsp && T[a].anchor == S[b].anchor) {
                    static uint32_t ta[4096], tb[4096];
                    const uint32_t A = T[a].anchor;
                    memset(ta, 0, sizeof ta); memset(tb, 0, sizeof tb);
                    parse(src, chunk, 0, ta, NULL, 0, A);
                    parse(src, chunk, st, tb, NULL, 0, A);
                    int nd = 0;
                    for (int h = 0; h < 4096; ++h) {
                        int ra = ta[h] + 65535u >= A && ta[h] != 0, rb = tb[h] + 65535u >= A && tb[h] != 0;
                        if (ra != rb || (ra && ta[h] != tb[h])) nd++;
                    }
                    if (!nd) n_hand_ok++; else n_hand_table++;
                } else n_hand_anchor++;
            }
            long bad = 0, seqs = 0, runs = 0, inrun = 0;
            uint32_t lastbad = 0;
            for (; i < nsp && S[i].anchor < e; ++i) {
                int32_t j = at[S[i].anchor];
                int same = j >= 0 && T[j].mstart == S[i].mstart && T[j].off == S[i].off && T[j].mlen == S[i].mlen;
                seqs++;
                if (!same) { bad++; lastbad = S[i].mstart + S[i].mlen; if (!inrun) runs++; inrun = 1; } else inrun = 0;
            }
            printf("chunk %ld seg %u: sync at +%u, %ld sequences, %ld differ in %ld runs, last difference ends at +%d\n", c0 / chunk, s / seg,
                   sync - s, seqs, bad, runs, bad ? (int)(lastbad - s) : -1);
            if (!bad) n_conv++;
            sum_sync += sync - s; sum_bad += bad; sum_seqs += seqs; sum_runs += runs;
        }
        printf("chunk %ld: %zu true sequences\n", c0 / chunk, nt);
    }
    printf("== %ld segments, %ld converge outright, %ld never share an anchor; mean sync +%.0f B; %.2f %% of the sequences differ, %.1f runs per segment\n",
           n_seg, n_conv, n_nosync, n_seg ? sum_sync / n_seg : 0, sum_seqs ? 100.0 * sum_bad / sum_seqs : 0, n_seg ? sum_runs / n_seg : 0);
    printf("== hand-over at each parse's OWN first anchor at or behind the border: %ld borders, %ld equal anchor + equivalent tables, %ld equal anchor but tables differ, %ld different anchors\n",
           n_hand, n_hand_ok, n_hand_table, n_hand_anchor);
    return 0;
}
