/* CPU model of the DENSE batches of lz4_chunks_kernel (sqy_kernels.hip) -- analysis / design tool, not product, not oracle.
 *
 *   gcc -O2 -o /tmp/dwm tools/dense_walk_model.c && /tmp/dwm tools/_c5plane.bin [chunk=262144]
 *
 * A dense batch lets 64 lanes probe positions P .. P+63 against the hash table as it stands and then WALKS the batch: sequence after
 * sequence, first event lane at or behind the cursor -> its verdict -> cursor behind the match.  The model runs, for every batch,
 *   (1) the walk as the kernel of round 4 does it (one scalar step per sequence, lanes that share a bucket with an earlier lane of the
 *       batch resolved by looking for the latest mate that has entered the table), and
 *   (2) the walk of round 5: every lane i works out, in parallel, what the parse does when its cursor stands at i -- the lane fq of the
 *       probe that matches, the verdict word, the next cursor E[i] -- as far as that follows from i alone; the serial part is then only
 *       the chain 0 -> E[0] -> E[E[0]] .. (one lane read per sequence).  Where the answer depends on more history than the cursor
 *       (a mate four or more lanes in front of the cursor) the lane says SLOW and the chain hands that one sequence to (1)'s step,
 * checks that both give the same sequences, and that the whole chunk's sequence list equals the plain liblz4 parse (SURVEY.md
 * Appendix B).  It counts what decides the design: sequences per batch, how many steps are plain / share a bucket / go SLOW. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef W
#define W 64            /* positions per batch: 64 (one per lane) or 128 (two per lane, -DW=128) */
#endif
typedef unsigned __int128 mask_t;
static inline int ctzm(mask_t m) { return (uint64_t)m ? __builtin_ctzll((uint64_t)m) : 64 + __builtin_ctzll((uint64_t)(m >> 64)); }
static inline int topm(mask_t m) { return (uint64_t)(m >> 64) ? 127 - __builtin_clzll((uint64_t)(m >> 64)) : 63 - __builtin_clzll((uint64_t)m); }
static inline int popm(mask_t m) { return __builtin_popcountll((uint64_t)m) + __builtin_popcountll((uint64_t)(m >> 64)); }
#define ONE ((mask_t)1)

typedef struct { uint32_t anchor, mstart, off, mlen; } seq_t;
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t h5(const uint8_t* p) { return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> 52); }

/* ---- plain reference parse (whole chunk) ---- */
static size_t parse_ref(const uint8_t* src, uint32_t n, seq_t* out)
{
    static uint32_t table[4096];
    memset(table, 0, sizeof table);
    size_t ns = 0;
    const uint8_t *ip = src, *anchor = src, *iend = src + n, *mfl = iend - 12 + 1, *mlimit = iend - 5;
    table[h5(ip)] = 0; ip++;
    uint32_t fh = h5(ip);
    for (;;) {
        const uint8_t* match;
        { const uint8_t* fip = ip; int step = 1, nb = 1 << 6;
          for (;;) { uint32_t h = fh, cur = (uint32_t)(fip - src), mi = table[h]; ip = fip; fip += step; step = nb++ >> 6;
                     if (fip > mfl) return ns; match = src + mi; fh = h5(fip); table[h] = cur;
                     if (mi + 65535 < cur) continue; if (rd32(match) == rd32(ip)) break; } }
        while (ip > anchor && match > src && ip[-1] == match[-1]) { ip--; match--; }
        uint32_t a = (uint32_t)(anchor - src);
        for (;;) {
            const uint8_t *pi = ip + 4, *pm = match + 4;
            while (pi < mlimit && *pi == *pm) { pi++; pm++; }
            out[ns].anchor = a; out[ns].mstart = (uint32_t)(ip - src); out[ns].off = (uint32_t)(ip - match); out[ns].mlen = (uint32_t)(pi - ip); ns++;
            ip = pi; anchor = ip;
            if (ip >= mfl) return ns;
            table[h5(ip - 2)] = (uint32_t)(ip - 2 - src);
            uint32_t h = h5(ip), cur = (uint32_t)(ip - src), mi = table[h];
            match = src + mi; table[h] = cur;
            if (mi + 65535 >= cur && rd32(match) == rd32(ip)) { a = cur; continue; }
            break;
        }
        fh = h5(++ip);
    }
}

/* ---- the batched parse ---- */
typedef struct {
    const uint8_t* src; uint32_t n; uint32_t table[4096];
    uint32_t P;            /* cursor = anchor: a match has just ended here (or the chunk's first probe) */
    int put2;              /* P - 2 still has to enter the table */
    seq_t* out; size_t ns;
    int done;
} st_t;

/* one sequence by the book, from the cursor state; returns 0 at the end of the chunk */
static int step_scalar(st_t* s, int first_of_chunk)
{
    const uint8_t* src = s->src;
    const uint8_t *iend = src + s->n, *mfl = iend - 12 + 1, *mlimit = iend - 5;
    const uint8_t *anchor = src + s->P, *ip = anchor, *match;
    uint32_t a = s->P;
    int have = 0;
    if (first_of_chunk) { s->table[h5(ip)] = 0; ip++; }
    else {
        if (ip >= mfl) { s->done = 1; return 0; }
        if (s->put2) { s->table[h5(ip - 2)] = (uint32_t)(ip - 2 - src); s->put2 = 0; }
        uint32_t h = h5(ip), cur = (uint32_t)(ip - src), mi = s->table[h];
        match = src + mi; s->table[h] = cur;
        if (mi + 65535 >= cur && rd32(match) == rd32(ip)) have = 1; else ip++;
    }
    if (!have) {
        uint32_t fh = h5(ip);
        const uint8_t* fip = ip; int step = 1, nb = 1 << 6;
        for (;;) { uint32_t h = fh, cur = (uint32_t)(fip - src), mi = s->table[h]; ip = fip; fip += step; step = nb++ >> 6;
                   if (fip > mfl) { s->done = 1; return 0; } match = src + mi; fh = h5(fip); s->table[h] = cur;
                   if (mi + 65535 < cur) continue; if (rd32(match) == rd32(ip)) break; }
        while (ip > anchor && match > src && ip[-1] == match[-1]) { ip--; match--; }
    }
    const uint8_t *pi = ip + 4, *pm = match + 4;
    while (pi < mlimit && *pi == *pm) { pi++; pm++; }
    s->out[s->ns].anchor = a; s->out[s->ns].mstart = (uint32_t)(ip - src); s->out[s->ns].off = (uint32_t)(ip - match); s->out[s->ns].mlen = (uint32_t)(pi - ip); s->ns++;
    s->P = (uint32_t)(pi - src); s->put2 = 1;
    if (pi >= mfl) { s->done = 1; return 0; }
    return 1;
}

typedef struct { int nseq, cur, keep; int fq[W / 4]; uint32_t inf[W / 4]; } walk_t;
/* verdict word of lane `lane` against candidate position cand: forward bytes (0..16) | equal bytes in front (0..4) << 5 | literal limit << 8 | offset << 16 */
static uint32_t verdict(const uint8_t* src, uint32_t pos, uint32_t cand)
{
    uint32_t d = 0; while (d < 16 && src[pos + d] == src[cand + d]) d++;
    uint32_t bk = 0; while (bk < 4 && cand >= bk + 1 && src[pos - 1 - bk] == src[cand - 1 - bk]) bk++;
    if (cand < 4 && bk == cand) bk = bk; /* (positions in front of the chunk do not exist: the kernel's shifted read gives no match there) */
    uint32_t maxlit1 = (d == 16 || cand < 16) ? 0 : (bk == 4 ? 5 : 15);
    return d | (bk << 5) | (maxlit1 << 8) | ((pos - cand) << 16);
}

static long g_batches, g_seqs, g_plain, g_dupsteps, g_dup_mate, g_slow, g_fast_dup, g_passed_dup, g_chain_steps, g_scalar_steps, g_end_lit;
static long g_case[8], g_rank[8], g_groups, g_duplanes;
static long g_x_dup, g_x_self2, g_x_table, g_x_slow, g_x_lit_self1, g_x_lit_self2, g_x_lit_other, g_x_lit_front;

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    uint32_t chunk = argc > 2 ? (uint32_t)atoi(argv[2]) : 262144;
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); long total = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* buf = malloc(total + 64); if (fread(buf, 1, total, f) != (size_t)total) return 1; fclose(f);
    seq_t* R = malloc(sizeof(seq_t) * chunk); seq_t* B = malloc(sizeof(seq_t) * chunk);
    for (long c0 = 0; c0 + chunk <= total; c0 += chunk) {
        const uint8_t* src = buf + c0;
        size_t nr = parse_ref(src, chunk, R);
        st_t* s = calloc(1, sizeof(st_t)); s->src = src; s->n = chunk; s->out = B;
        const uint32_t matchlimit = chunk - 5;
        step_scalar(s, 1);
        while (!s->done) {
            const uint32_t P = s->P;
            if (!(P >= 4 && P + W + 48 <= matchlimit)) { g_scalar_steps++; step_scalar(s, 0); continue; }
            /* ---- the vector part: every lane judges its own table candidate ---- */
            if (s->put2) { s->table[h5(src + P - 2)] = P - 2; s->put2 = 0; }
            uint32_t h[W], infoT[W]; mask_t M = 0, D = 0;
            mask_t mates[W];
            for (int l = 0; l < W; ++l) {
                h[l] = h5(src + P + l);
                mates[l] = 0;
                for (int k = 0; k < l; ++k) if (h[k] == h[l]) mates[l] |= ONE << k;
                if (mates[l]) D |= ONE << l;
                uint32_t old = s->table[h[l]];
                int near = (P + l - old) <= 65535 && rd32(src + old) == rd32(src + P + l);
                infoT[l] = verdict(src, P + l, old);
                if (near && (infoT[l] & 31) >= 4) M |= ONE << l;
            }
            { mask_t todo = D; while (todo) { int c = ctzm(todo); mask_t g = 0; for (int k = 0; k < W; ++k) if (h[k] == h[c]) g |= ONE << k; todo &= ~g; g_groups++; }
              g_duplanes += popm(D); }
            /* ---- (1) the walk of round 4 ---- */
            walk_t w1; memset(&w1, 0, sizeof w1); w1.keep = 1;
            {
                mask_t evm = M | D; int cur = 0, nseq = 0; mask_t inside = 0;   /* inside: lanes strictly inside recorded matches, minus ip-2 */
                for (;;) {
                    if (cur >= W || nseq >= W / 4) break;
                    mask_t ev = evm >> cur; if (!ev) break;
                    int fq = cur + ctzm(ev);
                    uint32_t inf = infoT[fq]; int is_hit = (int)((M >> fq) & 1);
                    if ((int)((D >> fq) & 1)) {
                        g_dupsteps++;
                        mask_t m = mates[fq] & ~inside;
                        if (m) { int qm = topm(m); inf = verdict(src, P + fq, P + qm); is_hit = (inf & 31) >= 4; g_dup_mate++;
                                 { int rank = popm(mates[fq] >> qm); g_rank[rank < 6 ? rank : 6]++; }
                                 /* the kernel's P + qm < 16 rule is part of verdict() (cand < 16) */ }
                        if (!is_hit) { evm &= ~(ONE << fq); continue; }
                    } else g_plain++;
                    if ((uint32_t)(fq - cur) >= ((inf >> 8) & 15)) { w1.keep = 0; break; }
                    w1.fq[nseq] = fq; w1.inf[nseq] = inf; nseq++;
                    int e = fq + (inf & 31);
                    for (int c = fq + 1; c < e && c < W; ++c) if (c != e - 2) inside |= ONE << c;
                    cur = e;
                }
                w1.nseq = nseq; w1.cur = cur;
            }
            /* ---- statistics for the "event at the cursor" rule: lane i is an event lane, shares its bucket with earlier lanes; the cursor stands
               at i itself (no literals).  i-1 and i-3 lie inside the match that just ended, i-2 is its ip-2: if i-2 is a mate it is the
               candidate (the verdict against pos-2 comes out of the lane's own bytes); if the bucket's FIRST lane is i-3 or later and i-2 is
               no mate, no mate has entered the table: the table entry is the candidate.  Else: history decides. ---- */
            {
                mask_t evm = M | D; int cur = 0, nseq = 0; mask_t inside = 0;
                for (;;) {
                    if (cur >= W || nseq >= W / 4) break;
                    mask_t ev = evm >> cur; if (!ev) break;
                    int fq = cur + ctzm(ev);
                    uint32_t inf = infoT[fq]; int is_hit = (int)((M >> fq) & 1);
                    if ((int)((D >> fq) & 1)) {
                        g_x_dup++;
                        int first = ctzm(mates[fq]);
                        if (fq == cur && cur >= 2) {
                            if ((int)((mates[fq] >> (cur - 2)) & 1)) g_x_self2++;
                            else if (first >= cur - 3) g_x_table++;
                            else g_x_slow++;
                        } else if (fq == cur) g_x_slow++;      /* cur < 2: cannot happen for a dup lane but for lane 1 */
                        else {
                            /* event behind the cursor (literals in between): nearest mate at or behind the cursor? */
                            int m1 = topm(mates[fq]);
                            if (m1 >= cur) { if (m1 == fq - 1) g_x_lit_self1++; else if (m1 == fq - 2) g_x_lit_self2++; else g_x_lit_other++; }
                            else g_x_lit_front++;
                        }
                        mask_t m = mates[fq] & ~inside;
                        if (m) { int qm = topm(m); inf = verdict(src, P + fq, P + qm); is_hit = (inf & 31) >= 4; }
                        if (!is_hit) { evm &= ~(ONE << fq); continue; }
                    }
                    if ((uint32_t)(fq - cur) >= ((inf >> 8) & 15)) break;
                    nseq++;
                    int e = fq + (inf & 31);
                    for (int c = fq + 1; c < e && c < W; ++c) if (c != e - 2) inside |= ONE << c;
                    cur = e;
                }
            }
            /* ---- (2) round 5: per-cursor answers, then the chain ---- */
            walk_t w2; memset(&w2, 0, sizeof w2); w2.keep = 1;
            {
                /* status per cursor lane i: 0 = a sequence (fq, inf, E), 1 = no event lane left (batch over), 2 = literal limit (batch ends, leave the
                   dense batches), 3 = SLOW (depends on lanes more than 3 in front of the cursor) */
                int st[W], Efq[W], Enext[W]; uint32_t Einf[W];
                for (int i = 0; i < W; ++i) {
                    mask_t evm = M | D; int e_from = i; st[i] = 1; Efq[i] = 0; Einf[i] = 0; Enext[i] = W;
                    for (int tries = 0; ; ++tries) {
                        mask_t ev = e_from < W ? evm >> e_from : 0; if (!ev) { st[i] = 1; break; }
                        int e = e_from + ctzm(ev);
                        uint32_t inf = infoT[e]; int is_hit = (int)((M >> e) & 1);
                        if ((int)((D >> e) & 1)) {
                            /* nearest mates first: a mate at or behind the cursor has entered the table; i-2 has (ip-2); i-1, i-3 have not (inside the
                               match that ended at i: it is at least 4 long); anything further in front of the cursor: history decides -> SLOW.
                               The batch's first cursor (i == 0) has no lanes in front of it. */
                            mask_t m = mates[e]; int cand = -1, slow = 0;
                            while (m) {
                                int q = topm(m); m &= ~(ONE << q);
                                if (q >= i || q == i - 2) { cand = q; break; }
                                if (q == i - 1 || q == i - 3) continue;
                                slow = 1; break;
                            }
                            if (slow) { st[i] = 3; break; }
                            if (cand >= 0) { inf = verdict(src, P + e, P + cand); is_hit = (inf & 31) >= 4; }
                            if (!is_hit) { e_from = e + 1; if (tries >= 3) { st[i] = 3; break; } continue; }   /* (a bounded number of passed lanes in the vector form) */
                        }
                        if ((uint32_t)(e - i) >= ((inf >> 8) & 15)) { st[i] = 2; Efq[i] = e; break; }
                        st[i] = 0; Efq[i] = e; Einf[i] = inf; Enext[i] = e + (inf & 31);
                        break;
                    }
                }
                /* the chain */
                int cur = 0, nseq = 0; mask_t inside = 0, evm = M | D;
                for (;;) {
                    if (cur >= W || nseq >= W / 4) break;
                    int status = st[cur];
                    if (status == 1) break;
                    if (status == 2) { w2.keep = 0; break; }
                    int fq; uint32_t inf;
                    if (status == 0) { fq = Efq[cur]; inf = Einf[cur]; g_chain_steps++; }
                    else {
                        /* SLOW: this one sequence by the step of (1) (needs `inside`, kept up to date below) */
                        g_slow++;
                        int found = 0; fq = 0; inf = 0;
                        mask_t evs = evm;
                        for (;;) {
                            mask_t ev = evs >> cur; if (!ev) break;
                            int e = cur + ctzm(ev);
                            uint32_t x = infoT[e]; int is_hit = (int)((M >> e) & 1);
                            if ((int)((int)((D >> e) & 1))) { mask_t m = mates[e] & ~inside;
                                                if (m) { int qm = topm(m); x = verdict(src, P + e, P + qm); is_hit = (x & 31) >= 4;
                                                         int rank = popm(mates[e] >> qm); g_case[rank < 6 ? rank : 6]++; }
                                                else g_case[0]++; }
                            else g_case[7]++;
                            if (!is_hit) { evs &= ~(ONE << e); continue; }
                            fq = e; inf = x; found = 1; break;
                        }
                        if (!found) break;
                        if ((uint32_t)(fq - cur) >= ((inf >> 8) & 15)) { w2.keep = 0; break; }
                    }
                    w2.fq[nseq] = fq; w2.inf[nseq] = inf; nseq++;
                    int e = fq + (inf & 31);
                    for (int c = fq + 1; c < e && c < W; ++c) if (c != e - 2) inside |= ONE << c;
                    cur = e;
                }
                w2.nseq = nseq; w2.cur = cur;
            }
            if (w1.nseq != w2.nseq || w1.cur != w2.cur || w1.keep != w2.keep || memcmp(w1.fq, w2.fq, sizeof(int) * w1.nseq) || memcmp(w1.inf, w2.inf, 4 * w1.nseq)) {
                printf("MISMATCH between the two walks at chunk %ld P %u: nseq %d/%d cur %d/%d keep %d/%d\n", c0 / chunk, P, w1.nseq, w2.nseq, w1.cur, w2.cur, w1.keep, w2.keep);
                return 2;
            }
            g_batches++;
            if (w1.nseq == 0) { g_scalar_steps++; step_scalar(s, 0); continue; }
            g_seqs += w1.nseq;
            /* the sequences, the table, the cursor */
            mask_t inside = 0; int anc = 0;
            for (int k = 0; k < w1.nseq; ++k) {
                int fq = w1.fq[k]; uint32_t inf = w1.inf[k]; int d = inf & 31, bk = (inf >> 5) & 7, lit0 = fq - anc;
                int back = bk < lit0 ? bk : lit0;
                s->out[s->ns].anchor = P + anc; s->out[s->ns].mstart = P + fq - back; s->out[s->ns].off = inf >> 16; s->out[s->ns].mlen = d + back; s->ns++;
                int e = fq + d;
                for (int c = fq + 1; c < e && c < W; ++c) if (c != e - 2) inside |= ONE << c;
                anc = e;
            }
            for (int l = 0; l < W && l < w1.cur; ++l) if (!(int)((inside >> l) & 1)) { uint32_t pos = P + l; if (s->table[h[l]] < pos) s->table[h[l]] = pos; }
            s->P = P + w1.cur; s->put2 = 1;
            if (!w1.keep) { g_end_lit++; g_scalar_steps++; step_scalar(s, 0); }
        }
        if (s->ns != nr || memcmp(B, R, sizeof(seq_t) * nr)) {
            size_t k = 0; while (k < nr && k < s->ns && !memcmp(&B[k], &R[k], sizeof(seq_t))) ++k;
            printf("chunk %ld: batched parse DIFFERS from the reference at sequence %zu of %zu / %zu (anchor %u vs %u)\n", c0 / chunk, k, s->ns, nr, B[k].anchor, R[k].anchor);
            return 3;
        }
        printf("chunk %ld: %zu sequences, equal to the reference parse\n", c0 / chunk, nr);
        free(s);
    }
    printf("batches %ld, sequences in batches %ld (%.2f per batch), scalar steps %ld, batches ended by a literal limit %ld\n", g_batches, g_seqs, g_batches ? (double)g_seqs / g_batches : 0, g_scalar_steps, g_end_lit);
    printf("round-4 walk: plain steps %ld, same-bucket steps %ld (%.1f %%), of those with a mate in the table %ld\n", g_plain, g_dupsteps, 100.0 * g_dupsteps / (g_plain + g_dupsteps + 1e-9), g_dup_mate);
    printf("  buckets with two or more lanes per batch %.2f, lanes with an earlier mate per batch %.2f; same-bucket steps by the rank of the mate that was the candidate: nearest %ld, 2nd %ld, 3rd %ld, 4th %ld, 5th %ld, further %ld\n",
           (double)g_groups / g_batches, (double)g_duplanes / g_batches, g_rank[1], g_rank[2], g_rank[3], g_rank[4], g_rank[5], g_rank[6]);
    printf("same-bucket event lanes %ld: AT the cursor -> candidate is lane-2 (own bytes) %ld, -> the table entry (bucket starts at lane-3 or later) %ld, history decides %ld; "
           "BEHIND the cursor -> nearest mate is lane-1 %ld, lane-2 %ld, another lane of the literal run %ld, a lane in front of the cursor %ld\n",
           g_x_dup, g_x_self2, g_x_table, g_x_slow, g_x_lit_self1, g_x_lit_self2, g_x_lit_other, g_x_lit_front);
    printf("round-5 walk: chain steps %ld, SLOW sequences %ld (%.2f per batch)\n", g_chain_steps, g_slow, g_batches ? (double)g_slow / g_batches : 0);
    printf("  SLOW events by what the candidate turned out to be: table %ld, nearest mate %ld, 2nd %ld, 3rd %ld, 4th %ld, 5th %ld, further %ld; plain lanes met on the way %ld\n",
           g_case[0], g_case[1], g_case[2], g_case[3], g_case[4], g_case[5], g_case[6], g_case[7]);
    return 0;
}
