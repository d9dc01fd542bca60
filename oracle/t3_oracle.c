/*
 * oracle/t3_oracle.c — TEST INFRASTRUCTURE (see t3_oracle.h).
 *
 * A scalar restatement, stage by stage and pass by pass, of what the reference computes on the
 * Word27 path.  OLD:n = /root/reference/old/include/ternary_image_codec_v6_min.hpp line n;
 * TPACK:n = /root/reference/include/ternary_packing.hpp line n.  It keeps the reference's pass
 * structure (materialise trits, regroup, interleave, split, encode, scramble, frame) so that it can
 * also stand as the "port" CPU baseline.  Pinned: tests/test_oracle_vs_ref.py (against
 * oracle/_ref, the unmodified reference compiled in place) and tests/golden/ (vectors captured
 * from oracle/_ref by tests/golden/make_golden.py).
 *
 * T3_MODE_FIXED is NOT reference behaviour: it is the build's self-consistent "v6c" variant
 * (DESIGN.md §fixed).  Its restatement here is pinned by round-trip properties only.
 */
#include "t3_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small growable byte vector (the reference leans on std::vector push_back)                    */
typedef struct { uint8_t* p; uint64_t n, cap; } bvec;
static void bv_reserve(bvec* v, uint64_t c) { if (c > v->cap) { v->p = (uint8_t*)realloc(v->p, c ? c : 1); v->cap = c; } }
static void bv_push(bvec* v, uint8_t x) { if (v->n == v->cap) bv_reserve(v, v->cap ? v->cap * 2 : 64); v->p[v->n++] = x; }
static void bv_free(bvec* v) { free(v->p); v->p = NULL; v->n = v->cap = 0; }

/* ---- trits <-> symbol (OLD:24-31) ---------------------------------------------------------- */
static inline uint8_t sym_of(uint8_t a, uint8_t b, uint8_t c) { return (uint8_t)(a + 3 * b + 9 * c); }
static inline void trits_of(uint8_t s, uint8_t* t) { t[0] = s % 3; t[1] = (s / 3) % 3; t[2] = (s / 9) % 3; }

/* ---- GF(27) = GF(3)[x]/(x^3+2x+1) (OLD:383-413) --------------------------------------------- */
uint8_t t3o_gf_add(uint8_t a, uint8_t b) {
    uint8_t x[3], y[3]; trits_of(a, x); trits_of(b, y);
    return sym_of((x[0] + y[0]) % 3, (x[1] + y[1]) % 3, (x[2] + y[2]) % 3);
}
uint8_t t3o_gf_sub(uint8_t a, uint8_t b) {
    uint8_t x[3], y[3]; trits_of(a, x); trits_of(b, y);
    return sym_of((3 + x[0] - y[0]) % 3, (3 + x[1] - y[1]) % 3, (3 + x[2] - y[2]) % 3);
}
uint8_t t3o_gf_mul(uint8_t a, uint8_t b) {
    if (!a || !b) return 0;
    uint8_t x[3], y[3]; trits_of(a, x); trits_of(b, y);
    int c[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c[i + j] += x[i] * y[j];
    /* x^3 = x + 2, x^4 = x^2 + 2x (mod 3)  (OLD:408-411) */
    c[0] += 2 * c[3]; c[1] += c[3]; c[1] += 2 * c[4]; c[2] += c[4];
    return sym_of(c[0] % 3, c[1] % 3, c[2] % 3);
}

typedef struct { uint8_t exp[78]; int16_t log[27]; uint8_t mul[729]; uint8_t inv[27]; uint8_t prim; int ready; } gf_t;
static gf_t G;
static void gf_init(void) { /* GF27Context::init OLD:436-466 */
    if (G.ready) return;
    uint8_t prim = 0;
    for (uint8_t c = 2; c < 27 && !prim; ++c) { /* order_of OLD:425-435 */
        uint8_t x = 1; int ord = -1;
        for (int i = 1; i <= 26; ++i) { x = t3o_gf_mul(x, c); if (x == 1) { ord = i; break; } }
        if (ord == 26) prim = c;
    }
    if (!prim) prim = 3;
    G.prim = prim;
    for (int i = 0; i < 27; ++i) G.log[i] = -1;
    G.exp[0] = 1; G.log[1] = 0;
    for (int i = 1; i < 26; ++i) { G.exp[i] = t3o_gf_mul(G.exp[i - 1], prim); G.log[G.exp[i]] = (int16_t)i; }
    for (int i = 26; i < 78; ++i) G.exp[i] = G.exp[i - 26];
    for (int a = 0; a < 27; ++a) for (int b = 0; b < 27; ++b) G.mul[a * 27 + b] = t3o_gf_mul((uint8_t)a, (uint8_t)b);
    G.inv[0] = 0;
    for (int a = 1; a < 27; ++a) G.inv[a] = G.exp[(26 - G.log[a]) % 26];
    G.ready = 1;
}
static inline uint8_t gmul(uint8_t a, uint8_t b) { return G.mul[a * 27 + b]; }
static inline uint8_t apow(int e) { return G.exp[((e % 26) + 26) % 26]; } /* pow_alpha OLD:479-482 */

void t3o_gf_tables(uint8_t* exp78, int16_t* log27, uint8_t* mul729, uint8_t* inv27, uint8_t* prim) {
    gf_init();
    memcpy(exp78, G.exp, 78); memcpy(log27, G.log, sizeof G.log); memcpy(mul729, G.mul, 729); memcpy(inv27, G.inv, 27);
    *prim = G.prim;
}

/* ---- RS(26,k) (OLD:490-663) ---------------------------------------------------------------- */
static int rs_k_ok(int k) { return k == 24 || k == 22 || k == 20 || k == 18; }

int t3o_rs_generator(int k, uint8_t* g) { /* build_gen OLD:501-516: g = prod (x - alpha^i), g[j] = coef of x^j */
    if (!rs_k_ok(k)) return -1;
    gf_init();
    int r = 26 - k, len = 1; uint8_t cur[10] = {1}, nxt[10];
    for (int i = 1; i <= r; ++i) {
        memset(nxt, 0, sizeof nxt);
        uint8_t root = apow(i);
        for (int j = 0; j < len; ++j) { nxt[j] = t3o_gf_sub(nxt[j], gmul(cur[j], root)); nxt[j + 1] = t3o_gf_add(nxt[j + 1], cur[j]); }
        ++len; memcpy(cur, nxt, sizeof cur);
    }
    memcpy(g, cur, (size_t)len);
    return len;
}

/* encode_block OLD:517-535 — the reference's own recurrence (NOT a valid RS codeword). */
static void rs_encode_compat(int k, const uint8_t* g, const uint8_t* data, uint8_t* out) {
    int r = 26 - k; uint8_t T[26];
    memset(T, 0, sizeof T); memcpy(T, data, (size_t)k);
    for (int i = 0; i < k; ++i) {
        uint8_t coef = T[i];
        if (!coef) continue;
        for (int j = 0; j <= r; ++j) T[i + j] = t3o_gf_sub(T[i + j], gmul(g[j], coef));
    }
    memcpy(out, data, (size_t)k); memcpy(out + k, T + k, (size_t)r);
}

/* FIXED: parity p with c = data || p and c(alpha^j) = 0, j = 1..r, c(x) = sum c_i x^i — the
 * convention decode_block's syndromes use (OLD:551-561).  Solved by Gauss-Jordan over GF(27). */
static void rs_encode_fixed(int k, const uint8_t* data, uint8_t* out) {
    int r = 26 - k; uint8_t A[8][9];
    for (int j = 0; j < r; ++j) {
        for (int m = 0; m < r; ++m) A[j][m] = apow((j + 1) * (k + m));
        uint8_t s = 0;
        for (int i = 0; i < k; ++i) s = t3o_gf_add(s, gmul(data[i], apow((j + 1) * i)));
        A[j][r] = t3o_gf_sub(0, s);
    }
    for (int c = 0; c < r; ++c) {
        int piv = c; while (piv < r && !A[piv][c]) ++piv;
        if (piv != c) for (int m = 0; m <= r; ++m) { uint8_t t = A[c][m]; A[c][m] = A[piv][m]; A[piv][m] = t; }
        uint8_t iv = G.inv[A[c][c]];
        for (int m = 0; m <= r; ++m) A[c][m] = gmul(A[c][m], iv);
        for (int j = 0; j < r; ++j) if (j != c && A[j][c]) {
            uint8_t f = A[j][c];
            for (int m = 0; m <= r; ++m) A[j][m] = t3o_gf_sub(A[j][m], gmul(f, A[c][m]));
        }
    }
    memcpy(out, data, (size_t)k);
    for (int m = 0; m < r; ++m) out[k + m] = A[m][r];
}

int t3o_rs_encode_blocks(int k, int mode, const uint8_t* data, uint64_t n_blocks, uint8_t* code26) {
    uint8_t g[10];
    if (t3o_rs_generator(k, g) < 0) return -1;
    for (uint64_t b = 0; b < n_blocks; ++b) {
        if (mode == T3_MODE_FIXED) rs_encode_fixed(k, data + b * k, code26 + b * 26);
        else rs_encode_compat(k, g, data + b * k, code26 + b * 26);
    }
    return 0;
}
int t3o_rs_parity_matrix(int k, int mode, uint8_t* P) {
    if (!rs_k_ok(k)) return -1;
    int r = 26 - k; uint8_t e[24], c[26];
    for (int i = 0; i < k; ++i) {
        memset(e, 0, sizeof e); e[i] = 1;
        t3o_rs_encode_blocks(k, mode, e, 1, c);
        memcpy(P + i * r, c + k, (size_t)r);
    }
    return 0;
}

/* decode_block OLD:546-662.  forney_sub = 0 reproduces the reference (adds the magnitude, OLD:658);
 * forney_sub = 1 is the FIXED decoder (subtracts it, and rejects when #roots != deg sigma). */
static int rs_decode_block(int k, uint8_t* c, uint8_t* out_k, int forney_sub) {
    const int n = 26, r = n - k, t = r / 2;
    uint8_t S[8]; int all0 = 1;
    for (int j = 0; j < r; ++j) { /* OLD:551-561 */
        uint8_t acc = 0;
        for (int i = 0; i < n; ++i) acc = t3o_gf_add(acc, gmul(c[i], apow(((j + 1) * i) % 26)));
        S[j] = acc; if (acc) all0 = 0;
    }
    if (all0) { memcpy(out_k, c, (size_t)k); return 1; } /* OLD:562-566 */
    /* Berlekamp–Massey OLD:567-605, polynomials as (coef array, length) like the vectors there */
    uint8_t sg[20] = {1}, B[20] = {1}; int ns = 1, nB = 1, L = 0, m = 1;
    for (int nS = 0; nS < r; ++nS) {
        uint8_t delta = S[nS];
        for (int i = 1; i <= L; ++i) if (i < ns) delta = t3o_gf_add(delta, gmul(sg[i], S[nS - i]));
        if (delta) {
            uint8_t T[20]; int nT = ns; memcpy(T, sg, sizeof T);
            uint8_t sh[40]; int nsh = m + nB; memset(sh, 0, sizeof sh);
            for (int i = 0; i < nB; ++i) sh[m + i] = gmul(delta, B[i]);
            int nd = ns > nsh ? ns : nsh; uint8_t nw[40];
            for (int i = 0; i < nd; ++i) nw[i] = t3o_gf_sub(i < ns ? sg[i] : 0, i < nsh ? sh[i] : 0);
            memcpy(sg, nw, (size_t)(nd < 20 ? nd : 20)); ns = nd;
            if (2 * L <= nS) {
                uint8_t iv = G.inv[delta];
                nB = nT; for (int i = 0; i < nT; ++i) B[i] = gmul(T[i], iv);
                L = nS + 1 - L; m = 1;
            } else m += 1;
        } else m += 1;
    }
    /* Omega = S(x) sigma(x) mod x^r  OLD:606-610 */
    uint8_t Om[32]; int nOm = (r + 1) + ns - 1; memset(Om, 0, sizeof Om);
    for (int i = 0; i < r; ++i) for (int j = 0; j < ns; ++j) Om[i + j] = t3o_gf_add(Om[i + j], gmul(S[i], sg[j]));
    if (nOm > r) nOm = r;
    /* Chien OLD:611-624 */
    int pos[26], np = 0;
    for (int i = 0; i < n; ++i) {
        uint8_t x = apow((-i) % 26), acc = 0;
        for (int d = ns - 1; d >= 0; --d) acc = t3o_gf_add(gmul(acc, x), sg[d]);
        if (!acc) pos[np++] = i;
    }
    if (np > t) return 0;
    if (forney_sub) { int deg = ns - 1; while (deg > 0 && !sg[deg]) --deg; if (np != deg) return 0; }
    /* formal derivative in characteristic 3  OLD:625-641 */
    uint8_t dp[20]; int ndp = ns > 1 ? ns - 1 : 1; memset(dp, 0, sizeof dp);
    for (int i = 1; i < ns; ++i) {
        int im = i % 3;
        dp[i - 1] = im == 0 ? 0 : (im == 1 ? sg[i] : t3o_gf_add(sg[i], sg[i]));
    }
    for (int e = 0; e < np; ++e) { /* Forney OLD:642-659 */
        uint8_t Xi = apow((-pos[e]) % 26), num = 0, den = 0;
        for (int d = nOm - 1; d >= 0; --d) num = t3o_gf_add(gmul(num, Xi), Om[d]);
        for (int d = ndp - 1; d >= 0; --d) den = t3o_gf_add(gmul(den, Xi), dp[d]);
        if (!den) return 0;
        uint8_t mag = gmul(t3o_gf_sub(0, num), G.inv[den]);
        c[pos[e]] = forney_sub ? t3o_gf_sub(c[pos[e]], mag) : t3o_gf_add(c[pos[e]], mag);
    }
    memcpy(out_k, c, (size_t)k);
    return 1;
}
int t3o_rs_decode_blocks(int k, int mode, uint8_t* code26, uint64_t n_blocks, uint8_t* data_k, uint8_t* ok) {
    if (!rs_k_ok(k)) return -1;
    gf_init();
    for (uint64_t b = 0; b < n_blocks; ++b) ok[b] = (uint8_t)rs_decode_block(k, code26 + b * 26, data_k + b * k, mode == T3_MODE_FIXED);
    return 0;
}

/* ---- RAW packer (OLD:675-747) ---------------------------------------------------------------- */
typedef struct { uint16_t Y; int16_t Cb, Cr; } px_t;
static void put_digits(uint32_t v, int w, uint8_t* T, int s) { for (int i = 0; i < w; ++i) { T[s + i] = (uint8_t)(v % 3); v /= 3; } } /* i2tr */
static uint32_t get_digits(const uint8_t* T, int w, int s) { uint32_t v = 0, p = 1; for (int i = 0; i < w; ++i) { v += p * T[s + i]; p *= 3; } return v; } /* tr2i */
static void pack2(const px_t* a, const px_t* b, uint8_t* w9) { /* pack_two_pixels OLD:693-705 */
    uint8_t T[27]; memset(T, 0, sizeof T);
    put_digits(a->Y, 5, T, 0); put_digits((uint32_t)(a->Cb + 40), 4, T, 5); put_digits((uint32_t)(a->Cr + 40), 4, T, 9);
    put_digits(b->Y, 5, T, 13); put_digits((uint32_t)(b->Cb + 40), 4, T, 18); put_digits((uint32_t)(b->Cr + 40), 4, T, 22);
    for (int s = 0; s < 9; ++s) w9[s] = sym_of(T[3 * s], T[3 * s + 1], T[3 * s + 2]);
}
int t3o_pack_pixels(const void* px6, uint64_t n_px, void* words9) { /* OLD:723-734; odd count pads a zero pixel */
    const px_t* px = (const px_t*)px6; uint8_t* w = (uint8_t*)words9; px_t z = {0, 0, 0};
    for (uint64_t i = 0; i < n_px; i += 2) pack2(&px[i], i + 1 < n_px ? &px[i + 1] : &z, w + (i / 2) * 9);
    return 0;
}
int t3o_unpack_words(const void* words9, uint64_t n_words, void* px6) { /* OLD:706-722, 735-747 */
    const uint8_t* w = (const uint8_t*)words9; px_t* px = (px_t*)px6;
    for (uint64_t i = 0; i < n_words; ++i) {
        uint8_t T[27];
        for (int s = 0; s < 9; ++s) trits_of(w[i * 9 + s], T + 3 * s);
        px_t a, b;
        a.Y = (uint16_t)get_digits(T, 5, 0);  a.Cb = (int16_t)((int16_t)get_digits(T, 4, 5) - 40);  a.Cr = (int16_t)((int16_t)get_digits(T, 4, 9) - 40);
        b.Y = (uint16_t)get_digits(T, 5, 13); b.Cb = (int16_t)((int16_t)get_digits(T, 4, 18) - 40); b.Cr = (int16_t)((int16_t)get_digits(T, 4, 22) - 40);
        px[2 * i] = a; px[2 * i + 1] = b;
    }
    return 0;
}

/* ---- regroup 26 trits/word -> 3-trit symbols with carry (OLD:1051-1082) ------------------------ */
static void regroup_into(const uint8_t* w, uint64_t n_words, bvec* sy) {
    uint8_t carry[3] = {0, 0, 0}; int clen = 0;
    for (uint64_t wi = 0; wi < n_words; ++wi) {
        uint8_t T[27];
        for (int s = 0; s < 9; ++s) trits_of(w[wi * 9 + s], T + 3 * s);
        int i = 0;
        if (clen > 0) {
            while (clen < 3 && i < 26) carry[clen++] = T[i++];
            if (clen == 3) { bv_push(sy, sym_of(carry[0], carry[1], carry[2])); clen = 0; }
        }
        for (; i + 2 < 26; i += 3) bv_push(sy, sym_of(T[i], T[i + 1], T[i + 2]));
        for (; i < 26; ++i) carry[clen++] = T[i];
    }
    if (clen > 0) { while (clen < 3) carry[clen++] = 0; bv_push(sy, sym_of(carry[0], carry[1], carry[2])); }
}
uint64_t t3o_regroup(const uint8_t* words9, uint64_t n_words, uint8_t* out) {
    bvec sy = {0}; regroup_into(words9, n_words, &sy);
    memcpy(out, sy.p, sy.n); uint64_t n = sy.n; bv_free(&sy); return n;
}

/* ---- 2-D boustrophedon (OLD:750-813) -------------------------------------------------------------- */
void t3o_interleave2d(uint8_t* s, uint64_t n, uint16_t w, uint16_t h, int inverse) {
    if (!w || !h) return;
    uint64_t A = (uint64_t)w * h; uint8_t* out = (uint8_t*)malloc(n ? n : 1);
    for (uint64_t base = 0; base < n; base += A) {
        uint64_t take = n - base < A ? n - base : A, q = 0;
        for (uint32_t r = 0; r < h; ++r) {
            if (r % 2 == 0) {
                for (uint32_t c = 0; c < w && (uint64_t)r * w + c < take; ++c) {
                    uint64_t idx = (uint64_t)r * w + c;
                    if (inverse) out[base + idx] = s[base + q++]; else out[base + q++] = s[base + idx];
                }
            } else {
                for (int c = (int)w - 1; c >= 0; --c) {
                    uint64_t idx = (uint64_t)r * w + (uint32_t)c;
                    if (idx < take) { if (inverse) out[base + idx] = s[base + q++]; else out[base + q++] = s[base + idx]; }
                }
            }
        }
    }
    memcpy(s, out, n); free(out);
}

/* ---- scrambler (OLD:77-94): one chain over the stream, state stepped BEFORE use ---------------------- */
void t3o_scramble(uint8_t* s, uint64_t n, uint32_t a, uint32_t b, uint32_t s0, int inverse) {
    uint32_t st = s0 % 3;
    for (uint64_t i = 0; i < n; ++i) {
        st = (a * st + b) % 3; /* uint32 wrap-around, OLD:83 */
        uint8_t t[3]; trits_of(s[i], t);
        for (int c = 0; c < 3; ++c) t[c] = (uint8_t)(inverse ? (3 + t[c] - st % 3) % 3 : (t[c] + st) % 3);
        s[i] = sym_of(t[0], t[1], t[2]);
    }
}
uint8_t t3o_beacon_symbol(uint8_t profile, uint16_t frame_seq_mod, uint8_t health) { /* OLD:107-113 */
    return (uint8_t)((profile + 5 * (uint8_t)(frame_seq_mod % 5) + 15 * (uint8_t)(health % 3)) % 27);
}

/* ---- ternary CRC-12 + 27-symbol header (OLD:176-380) ---------------------------------------------------- */
void t3o_crc12(const uint8_t* msg, uint64_t n, uint8_t* out) { /* LFSR taps 0,3,4,7 (OLD:183-200), 12 flush steps */
    uint8_t r[12]; memset(r, 0, sizeof r);
    for (uint64_t i = 0; i < n + 12; ++i) {
        uint8_t in = i < n ? msg[i] : 0, fb = (uint8_t)((in + r[11]) % 3), nx[12];
        nx[0] = fb;
        for (int j = 1; j < 12; ++j) nx[j] = r[j - 1];
        nx[3] = (uint8_t)((nx[3] + fb) % 3); nx[4] = (uint8_t)((nx[4] + fb) % 3); nx[7] = (uint8_t)((nx[7] + fb) % 3);
        memcpy(r, nx, sizeof r);
    }
    memcpy(out, r, 12);
}
static int is_crc_slot(int i) { return i == 20 || i == 21 || i == 22 || i == 26; }
static void header_crc(const uint8_t* s27, uint8_t* r12) {
    uint8_t tr[81]; int n = 0;
    for (int i = 0; i < 27; ++i) if (!is_crc_slot(i)) { trits_of(s27[i], tr + n); n += 3; }
    t3o_crc12(tr, (uint64_t)n, r12);
}
static uint8_t subword_code(uint8_t sub) { switch (sub) { case 24: return 1; case 21: return 2; case 18: return 3; case 15: return 4; default: return 0; } }
void t3o_header_pack(const t3_cfg* c, uint32_t frame_seq, uint32_t band_map_hash, uint8_t* p) { /* OLD:208-289 */
    const uint32_t magic = 0x0A2, version = 1;
    memset(p, 0, 27);
    p[0] = magic % 27; p[1] = (magic / 27) % 27; p[2] = version % 27; p[3] = (uint8_t)(c->profile % 27);
    for (int g = 0; g < 3; ++g) { uint32_t u = 0; for (int i = 3 * g; i < 3 * g + 3; ++i) u = u * 3 + c->band_profile[i] % 3; p[4 + g] = (uint8_t)(u % 27); }
    p[7] = c->tile_w % 27; p[8] = c->tile_h % 27;
    p[9] = c->seed_a % 27; p[10] = c->seed_b % 27; p[11] = c->seed_s0 % 27;
    p[12] = (uint8_t)((subword_code(c->subword) + 9 * (c->centered ? 1 : 0)) % 27);
    p[13] = band_map_hash % 27; p[14] = (band_map_hash / 27) % 27; p[15] = (band_map_hash / 729) % 27;
    p[16] = (uint8_t)(c->coset % 3);
    p[17] = frame_seq % 27; p[18] = (frame_seq / 27) % 27; p[19] = (frame_seq / 729) % 27;
    p[23] = c->beacon_enabled ? 1 : 0; p[24] = c->beacon_band_slot % 27;
    p[25] = (uint8_t)(c->beacon_words_period < 26 ? c->beacon_words_period : 26);
    uint8_t r[12]; header_crc(p, r);
    p[20] = sym_of(r[0], r[1], r[2]); p[21] = sym_of(r[3], r[4], r[5]); p[22] = sym_of(r[6], r[7], r[8]); p[26] = sym_of(r[9], r[10], r[11]);
}
int t3o_header_check(const uint8_t* p) { /* OLD:290-316 */
    uint8_t r[12], h[12]; header_crc(p, r);
    trits_of(p[20], h); trits_of(p[21], h + 3); trits_of(p[22], h + 6); trits_of(p[26], h + 9);
    return memcmp(r, h, 12) == 0;
}
void t3o_header_unpack(const uint8_t* p, t3_cfg* o, uint32_t* frame_seq, uint32_t* band_map_hash, uint16_t* magic, uint8_t* version) { /* OLD:317-379 */
    static const uint8_t subs[5] = {27, 24, 21, 18, 15};
    uint8_t q[27]; for (int i = 0; i < 27; ++i) q[i] = p[i] % 27;
    *magic = (uint16_t)(q[0] + 27 * q[1]); *version = q[2];
    o->profile = q[3] % 5;
    for (int g = 0; g < 3; ++g) { uint32_t v = q[4 + g]; for (int i = 0; i < 3; ++i) { o->band_profile[3 * g + i] = v % 3; v /= 3; } }
    o->tile_w = q[7]; o->tile_h = q[8]; o->seed_a = q[9]; o->seed_b = q[10]; o->seed_s0 = q[11];
    uint8_t sub = q[12] % 9; o->subword = sub < 5 ? subs[sub] : 27; o->centered = ((q[12] / 9) % 3) != 0;
    *band_map_hash = q[13] + 27u * q[14] + 729u * q[15];
    o->coset = q[16] % 3;
    *frame_seq = q[17] + 27u * q[18] + 729u * q[19];
    o->beacon_enabled = q[23] != 0; o->beacon_band_slot = q[24] % 9; o->beacon_words_period = q[25];
}

/* ---- frame level -------------------------------------------------------------------------------------- */
static int band_k(const t3_cfg* c, int b) { static const int ks[4] = {24, 22, 20, 18}; return ks[c->band_profile[b] % 4]; } /* OLD:1089-1100 */
static int want_2d(const t3_cfg* c) { return c->profile == T3_P5_RS26_22_2D && c->tile_w && c->tile_h; }                  /* OLD:1083 */

/* FIXED-mode extension symbols: what the 27-symbol header loses to its %27 fields. */
static void fixed_ext(const t3_cfg* c, uint64_t n_raw, uint8_t* e) {
    uint32_t v; uint64_t q; int i;
    v = c->tile_w / 27u; for (i = 0; i < 3; ++i) { e[i] = v % 27; v /= 27; }
    v = c->tile_h / 27u; for (i = 0; i < 3; ++i) { e[3 + i] = v % 27; v /= 27; }
    v = c->beacon_words_period; for (i = 0; i < 7; ++i) { e[6 + i] = v % 27; v /= 27; }
    q = n_raw; for (i = 0; i < 10; ++i) { e[13 + i] = (uint8_t)(q % 27); q /= 27; }
    for (i = 0; i < 3; ++i) { uint8_t f = 0; for (int j = 0; j < 3; ++j) if (c->band_profile[3 * i + j] % 4 == 3) f |= (uint8_t)(1 << j); e[23 + i] = f; }
    e[26] = (uint8_t)(((c->seed_a * 0u + c->seed_b) % 3) + 3 * ((c->seed_a * 1u + c->seed_b) % 3) + 9 * ((c->seed_a * 2u + c->seed_b) % 3));
}
static void fixed_canon(const t3_cfg* c, t3_cfg* o) { *o = *c; for (int b = 0; b < 9; ++b) o->band_profile[b] = c->band_profile[b] % 4; }

static int header_syms_of(const t3_cfg* c) { return c->mode == T3_MODE_FIXED ? 90 : 52; }
static void make_header(const t3_cfg* c, uint64_t n_raw, uint8_t* out) {
    uint8_t hp[27], A[18], B[18], g[10];
    t3o_rs_generator(18, g);
    if (c->mode == T3_MODE_FIXED) {
        t3_cfg cc; fixed_canon(c, &cc);
        uint8_t ext[27], C[18]; t3o_header_pack(&cc, 0, 0, hp); fixed_ext(&cc, n_raw, ext);
        memcpy(A, hp, 18); memcpy(B, hp + 18, 9); memcpy(B + 9, ext, 9); memcpy(C, ext + 9, 18);
        rs_encode_fixed(18, A, out); rs_encode_fixed(18, B, out + 26); rs_encode_fixed(18, C, out + 52);
        memset(out + 78, 0, 12); /* pad to 10 whole words: the body then starts word- and 2-byte-aligned */
    } else { /* OLD:1142-1162 */
        t3o_header_pack(c, 0, 0, hp);
        memcpy(A, hp, 18); memcpy(B, hp + 18, 9); memset(B + 9, 0, 9);
        rs_encode_compat(18, g, A, out); rs_encode_compat(18, g, B, out + 26);
    }
}

static int encode_syms(bvec* sy, const t3_cfg* c, uint64_t n_raw, void* out9, uint64_t cap, uint64_t* n_out) {
    const int fixed = c->mode == T3_MODE_FIXED;
    gf_init();
    if (want_2d(c)) t3o_interleave2d(sy->p, sy->n, c->tile_w, c->tile_h, 0);
    bvec bands[9]; memset(bands, 0, sizeof bands);
    for (uint64_t i = 0; i < sy->n; ++i) bv_push(&bands[i % 9], sy->p[i]); /* OLD:1087-1088 */
    bvec body = {0};
    for (int b = 0; b < 9; ++b) { /* OLD:1102-1115 */
        int k = band_k(c, b); uint8_t g[10], kb[26], nb[26]; t3o_rs_generator(k, g);
        uint64_t j = 0;
        for (; j + (uint64_t)k <= bands[b].n; j += (uint64_t)k) {
            if (fixed) rs_encode_fixed(k, bands[b].p + j, nb); else rs_encode_compat(k, g, bands[b].p + j, nb);
            for (int i = 0; i < 26; ++i) bv_push(&body, nb[i]);
        }
        if (fixed && j < bands[b].n) { /* v6c: zero-padded last block instead of dropping it */
            memset(kb, 0, sizeof kb); memcpy(kb, bands[b].p + j, bands[b].n - j);
            rs_encode_fixed(k, kb, nb);
            for (int i = 0; i < 26; ++i) bv_push(&body, nb[i]);
        }
        bv_free(&bands[b]);
    }
    t3o_scramble(body.p, body.n, c->seed_a, c->seed_b, c->seed_s0, 0); /* OLD:1116-1117 */
    if (c->beacon_enabled && c->beacon_words_period > 0) { /* OLD:1118-1141 */
        bvec s2 = {0}; uint64_t word_idx = 0, q = 0;
        uint8_t bs = t3o_beacon_symbol(c->profile, (uint16_t)(c->superframe_words % 5), 0);
        while (q < body.n) {
            int ins = (word_idx % c->beacon_words_period) == 0;
            for (int slot = 0; slot < 9; ++slot) {
                if (ins && slot == c->beacon_band_slot) bv_push(&s2, bs);
                else bv_push(&s2, q < body.n ? body.p[q++] : 0);
            }
            ++word_idx;
        }
        bv_free(&body); body = s2;
    }
    int hs = header_syms_of(c); uint8_t hdr[96]; make_header(c, n_raw, hdr);
    uint64_t total = (uint64_t)hs + body.n, words = (total + 8) / 9; /* OLD:1164 */
    *n_out = words;
    if (words > cap) { bv_free(&body); return T3_E_CAPACITY; }
    uint8_t* o = (uint8_t*)out9; memset(o, 0, words * 9);
    memcpy(o, hdr, (size_t)hs); if (body.n) memcpy(o + hs, body.p, body.n);
    bv_free(&body);
    return T3_OK;
}

int t3o_encode_profile(const void* raw9, uint64_t n_raw, const t3_cfg* c, void* out9, uint64_t cap, uint64_t* n_out) {
    if (c->mode == T3_MODE_FIXED && c->profile != T3_RAW_MODE) { /* v6c limits */
        if (c->profile > T3_P5_RS26_22_2D || n_raw >= 205891132094649ull /* 27^10 */) return T3_E_ARG;
        if (c->beacon_enabled && c->beacon_words_period > 0 && c->beacon_band_slot >= 9) return T3_E_ARG;
    }
    if (c->profile == T3_RAW_MODE) { /* OLD:1046-1050 */
        *n_out = n_raw; if (n_raw > cap) return T3_E_CAPACITY;
        memcpy(out9, raw9, n_raw * 9); return T3_OK;
    }
    bvec sy = {0}; bv_reserve(&sy, n_raw * 9 + 1);
    regroup_into((const uint8_t*)raw9, n_raw, &sy);
    int rc = encode_syms(&sy, c, n_raw, out9, cap, n_out);
    bv_free(&sy);
    return rc;
}
uint64_t t3o_encoded_words(uint64_t n_raw, const t3_cfg* c) {
    if (c->profile == T3_RAW_MODE) return n_raw;
    uint64_t nsym = (26 * n_raw + 2) / 3, body = 0;
    for (int b = 0; b < 9; ++b) {
        uint64_t len = nsym > (uint64_t)b ? (nsym - b + 8) / 9 : 0, k = (uint64_t)band_k(c, b);
        body += 26 * (c->mode == T3_MODE_FIXED ? (len + k - 1) / k : len / k);
    }
    if (c->beacon_enabled && c->beacon_words_period > 0) {
        uint64_t P = c->beacon_words_period, w = 0, room = 0;
        while (room < body) { room += (c->beacon_band_slot < 9 && w % P == 0) ? 8 : 9; ++w; }
        body = 9 * w;
    }
    return ((uint64_t)header_syms_of(c) + body + 8) / 9;
}
int t3o_encode_frame(const void* px6, uint64_t n_px, const t3_cfg* c, void* out9, uint64_t cap, uint64_t* n_out) {
    uint64_t nw = (n_px + 1) / 2; uint8_t* raw = (uint8_t*)malloc(nw * 9 + 1);
    t3o_pack_pixels(px6, n_px, raw);
    int rc = t3o_encode_profile(raw, nw, c, out9, cap, n_out);
    free(raw); return rc;
}

/* COMPAT decode = decode_profile_to_raw OLD:995-1041 with its own framing (6 header words, slot->band
 * demap, band-serial output).  Returns T3_OK for `true`, T3_E_HEADER / T3_E_RS for `false`. */
static int decode_compat(const uint8_t* in, uint64_t n_in, t3_cfg* seen, bvec* use) {
    if (n_in < 6) return T3_E_HEADER; /* OLD:920 */
    uint8_t A[26], B[26], a18[18], b18[18], hp[27];
    for (int i = 0; i < 26; ++i) { A[i] = in[i] % 27; B[i] = in[26 + i] % 27; } /* symbols >= 27 are UB in the reference */
    if (!rs_decode_block(18, A, a18, 0)) return T3_E_HEADER;
    if (!rs_decode_block(18, B, b18, 0)) return T3_E_HEADER;
    memcpy(hp, a18, 18); memcpy(hp + 18, b18, 9);
    if (!t3o_header_check(hp)) return T3_E_HEADER;
    uint32_t fs, bh; uint16_t mg; uint8_t ver; uint8_t keep_mode = seen->mode; uint32_t keep_sf = seen->superframe_words;
    t3o_header_unpack(hp, seen, &fs, &bh, &mg, &ver); /* OLD:1006-1013 */
    seen->mode = keep_mode; seen->superframe_words = keep_sf;
    uint64_t nbw = n_in - 6, nbs = nbw * 9; uint8_t* body = (uint8_t*)malloc(nbs ? nbs : 1);
    memcpy(body, in + 54, nbs);
    t3o_scramble(body, nbs, seen->seed_a, seen->seed_b, seen->seed_s0, 1); /* OLD:938-947 */
    bvec bands[9]; memset(bands, 0, sizeof bands);
    int skip = seen->beacon_enabled && seen->beacon_words_period > 0;
    for (uint64_t wi = 0; wi < nbw; ++wi) /* OLD:953-961 */
        for (int slot = 0; slot < 9; ++slot) {
            if (skip && (wi % seen->beacon_words_period) == 0 && slot == seen->beacon_band_slot) continue;
            bv_push(&bands[slot], body[wi * 9 + slot]);
        }
    free(body);
    int rc = T3_OK;
    for (int b = 0; b < 9 && rc == T3_OK; ++b) { /* OLD:963-991 */
        int k = band_k(seen, b); uint8_t nb[26], kb[26];
        for (uint64_t j = 0; j + 26 <= bands[b].n; j += 26) {
            memcpy(nb, bands[b].p + j, 26);
            if (!rs_decode_block(k, nb, kb, 0)) { rc = T3_E_RS; break; }
            for (int i = 0; i < k; ++i) bv_push(use, kb[i]);
        }
    }
    for (int b = 0; b < 9; ++b) bv_free(&bands[b]);
    if (rc != T3_OK) return rc;
    if (seen->profile == T3_P5_RS26_22_2D && seen->tile_w && seen->tile_h) t3o_interleave2d(use->p, use->n, seen->tile_w, seen->tile_h, 1);
    return T3_OK;
}

/* FIXED decode: the inverse of encode_syms(mode=FIXED). */
static int decode_fixed(const uint8_t* in, uint64_t n_in, t3_cfg* seen, bvec* use, uint64_t* n_raw_out) {
    if (n_in < 10) return T3_E_HEADER;
    uint8_t blk[3][26], dat[3][18], hp[27], ext[27];
    for (int q = 0; q < 3; ++q) { for (int i = 0; i < 26; ++i) blk[q][i] = in[26 * q + i] % 27; if (!rs_decode_block(18, blk[q], dat[q], 1)) return T3_E_HEADER; }
    memcpy(hp, dat[0], 18); memcpy(hp + 18, dat[1], 9); memcpy(ext, dat[1] + 9, 9); memcpy(ext + 9, dat[2], 18);
    if (!t3o_header_check(hp)) return T3_E_HEADER;
    uint32_t fs, bh; uint16_t mg; uint8_t ver; uint32_t keep_sf = seen->superframe_words;
    t3_cfg c = *seen; t3o_header_unpack(hp, &c, &fs, &bh, &mg, &ver);
    c.mode = T3_MODE_FIXED; c.superframe_words = keep_sf;
    /* the header packs band triples MSD-first but unpacks them LSD-first (OLD:222-224 vs 327-340): v6c undoes that */
    for (int g3 = 0; g3 < 3; ++g3) { uint8_t t = c.band_profile[3 * g3]; c.band_profile[3 * g3] = c.band_profile[3 * g3 + 2]; c.band_profile[3 * g3 + 2] = t; }
    uint32_t v = 0; uint64_t q = 0; int i;
    for (v = 0, i = 2; i >= 0; --i) v = v * 27 + ext[i];
    c.tile_w = (uint16_t)(c.tile_w + 27 * v);
    for (v = 0, i = 2; i >= 0; --i) v = v * 27 + ext[3 + i];
    c.tile_h = (uint16_t)(c.tile_h + 27 * v);
    for (q = 0, i = 6; i >= 0; --i) q = q * 27 + ext[6 + i];
    c.beacon_words_period = (uint32_t)q;
    for (q = 0, i = 9; i >= 0; --i) q = q * 27 + ext[13 + i];
    uint64_t n_raw = q;
    for (i = 0; i < 9; ++i) if (ext[23 + i / 3] >> (i % 3) & 1) c.band_profile[i] = 3;
    uint8_t next[3] = {(uint8_t)(ext[26] % 3), (uint8_t)((ext[26] / 3) % 3), (uint8_t)((ext[26] / 9) % 3)};
    *seen = c; *n_raw_out = n_raw;
    if (t3o_encoded_words(n_raw, &c) > n_in) return T3_E_HEADER;
    /* undo beacon framing */
    uint64_t nsym = (26 * n_raw + 2) / 3, body_syms = 0, blocks[9], len[9];
    for (int b = 0; b < 9; ++b) { len[b] = nsym > (uint64_t)b ? (nsym - b + 8) / 9 : 0; uint64_t k = (uint64_t)band_k(&c, b); blocks[b] = (len[b] + k - 1) / k; body_syms += 26 * blocks[b]; }
    uint8_t* body = (uint8_t*)malloc(body_syms ? body_syms : 1); const uint8_t* fr = in + 90;
    if (c.beacon_enabled && c.beacon_words_period > 0) {
        uint64_t w = 0, got = 0;
        while (got < body_syms) { for (int slot = 0; slot < 9 && got < body_syms; ++slot) { if (w % c.beacon_words_period == 0 && slot == c.beacon_band_slot) continue; body[got++] = fr[w * 9 + slot]; } ++w; }
    } else memcpy(body, fr, body_syms);
    /* descramble with the transmitted next-state table */
    { uint32_t st = c.seed_s0 % 3;
      for (uint64_t j = 0; j < body_syms; ++j) { st = next[st]; uint8_t t[3]; trits_of(body[j], t); for (int d = 0; d < 3; ++d) t[d] = (uint8_t)((3 + t[d] - st) % 3); body[j] = sym_of(t[0], t[1], t[2]); } }
    /* RS per band, then re-merge the i%9 round robin */
    bv_reserve(use, nsym + 1); use->n = nsym; memset(use->p, 0, nsym + 1);
    uint64_t off = 0; int rc = T3_OK;
    for (int b = 0; b < 9 && rc == T3_OK; ++b) {
        int k = band_k(&c, b); uint8_t kb[26];
        for (uint64_t m = 0; m < blocks[b]; ++m) {
            if (!rs_decode_block(k, body + off + 26 * m, kb, 1)) { rc = T3_E_RS; break; }
            for (int p = 0; p < k; ++p) { uint64_t j = m * (uint64_t)k + (uint64_t)p; if (j < len[b]) use->p[9 * j + (uint64_t)b] = kb[p]; }
        }
        off += 26 * blocks[b];
    }
    free(body);
    if (rc != T3_OK) return rc;
    if (want_2d(&c)) t3o_interleave2d(use->p, use->n, c.tile_w, c.tile_h, 1);
    return T3_OK;
}

int t3o_decode_profile(const void* in9, uint64_t n_in, t3_cfg* seen, void* out9, uint64_t cap, uint64_t* n_out) {
    *n_out = 0;
    gf_init();
    if (seen->profile == T3_RAW_MODE) { /* OLD:998-1002 */
        *n_out = n_in; if (n_in > cap) return T3_E_CAPACITY;
        memcpy(out9, in9, n_in * 9); return T3_OK;
    }
    bvec use = {0}; uint64_t n_raw = 0; int rc;
    if (seen->mode == T3_MODE_FIXED) rc = decode_fixed((const uint8_t*)in9, n_in, seen, &use, &n_raw);
    else rc = decode_compat((const uint8_t*)in9, n_in, seen, &use);
    if (rc != T3_OK) { bv_free(&use); return rc; }
    /* symbols -> trits -> 26 trits per word, T[26] = 0 (OLD:1022-1040) */
    uint64_t ntr = use.n * 3, words = ntr / 26;
    if (seen->mode == T3_MODE_FIXED) words = n_raw;
    *n_out = words;
    if (words > cap) { bv_free(&use); return T3_E_CAPACITY; }
    uint8_t* tr = (uint8_t*)malloc(ntr + 27);
    for (uint64_t i = 0; i < use.n; ++i) trits_of(use.p[i], tr + 3 * i);
    uint8_t* o = (uint8_t*)out9;
    for (uint64_t w = 0; w < words; ++w) {
        uint8_t T[27]; memcpy(T, tr + 26 * w, 26); T[26] = 0;
        for (int s = 0; s < 9; ++s) o[w * 9 + s] = sym_of(T[3 * s], T[3 * s + 1], T[3 * s + 2]);
    }
    free(tr); bv_free(&use);
    return T3_OK;
}
int t3o_decode_frame(const void* in9, uint64_t n_in, t3_cfg* seen, void* px6, uint64_t cap_px, uint64_t* n_px) {
    uint64_t cap = n_in + 16, nw = 0; uint8_t* raw = (uint8_t*)malloc(cap * 9);
    int rc = t3o_decode_profile(in9, n_in, seen, raw, cap, &nw);
    *n_px = 2 * nw;
    if (rc == T3_OK) { if (2 * nw > cap_px) rc = T3_E_CAPACITY; else t3o_unpack_words(raw, nw, px6); }
    free(raw); return rc;
}

/* ---- subword helpers (OLD:816-859) and wire helpers (TPACK:18-65) ----------------------------------------- */
uint64_t t3o_extract_subword_stream(const void* words9, uint64_t n_words, int N, uint8_t* out) {
    const uint8_t* w = (const uint8_t*)words9; uint64_t n = 0;
    for (uint64_t i = 0; i < n_words; ++i) { uint8_t T[27]; for (int s = 0; s < 9; ++s) trits_of(w[i * 9 + s], T + 3 * s); for (int j = 0; j < N; ++j) out[n++] = T[j]; }
    return n;
}
uint64_t t3o_build_words_from_subword_stream(const uint8_t* in, uint64_t n, int N, uint8_t fill, void* words9) {
    uint8_t* w = (uint8_t*)words9; uint64_t idx = 0, nw = 0;
    while (idx < n) { /* OLD:845-859: buf zero-filled up to N, `fill` beyond N */
        uint8_t T[27]; int take = (int)(n - idx < (uint64_t)N ? n - idx : (uint64_t)N);
        for (int i = 0; i < 27; ++i) T[i] = i < take ? in[idx + i] : (i < N ? 0 : fill);
        for (int s = 0; s < 9; ++s) w[nw * 9 + s] = sym_of(T[3 * s], T[3 * s + 1], T[3 * s + 2]);
        ++nw; idx += (uint64_t)take;
    }
    return nw;
}
uint64_t t3o_ut_to_base243(const uint8_t* in, uint64_t n, uint8_t* out) { /* TPACK:28-38 */
    uint32_t total = (uint32_t)n; memcpy(out, &total, 4); uint64_t o = 4;
    for (uint64_t i = 0; i < n; i += 5) { int v = 0, p = 1; for (int j = 0; j < 5; ++j) { v += p * (i + j < n ? in[i + j] : 0); p *= 3; } out[o++] = (uint8_t)v; }
    return o;
}
int64_t t3o_base243_to_ut(const uint8_t* in, uint64_t n, uint8_t* out) { /* TPACK:40-50 */
    if (n < 4) return -1;
    uint32_t total; memcpy(&total, in, 4); uint64_t got = 0;
    for (uint64_t idx = 4; idx < n && got < total; ++idx) { int v = in[idx]; for (int j = 0; j < 5; ++j) { if (got < total) out[got++] = (uint8_t)(v % 3); v /= 3; } }
    return got == total ? (int64_t)got : -1;
}
void t3o_words_to_bytes(const void* words9, uint64_t n_words, uint8_t* out) { const uint8_t* w = (const uint8_t*)words9; for (uint64_t i = 0; i < n_words * 9; ++i) out[i] = w[i] % 27; } /* TPACK:53-58 */
uint64_t t3o_bytes_to_words(const uint8_t* in, uint64_t n, void* words9) { if (n % 9) return 0; uint8_t* w = (uint8_t*)words9; for (uint64_t i = 0; i < n; ++i) w[i] = in[i] % 27; return n / 9; } /* TPACK:60-65 */

/* ---- checkers' utilities ------------------------------------------------------------------------------------ */
uint64_t t3o_fnv1a64(const void* d, uint64_t n) { const uint8_t* p = (const uint8_t*)d; uint64_t h = 1469598103934665603ull; for (uint64_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; } return h; }
uint32_t t3o_crc32(const void* d, uint64_t n) { /* poly 0xEDB88320, as the containers use (io_t3p_t3v.cpp:18-33) */
    static uint32_t T[256]; static int init = 0;
    if (!init) { for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int j = 0; j < 8; ++j) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1); T[i] = c; } init = 1; }
    const uint8_t* p = (const uint8_t*)d; uint32_t c = 0xFFFFFFFFu;
    for (uint64_t i = 0; i < n; ++i) c = T[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
uint32_t t3o_sym_sum(const void* d, uint64_t n) { const uint8_t* p = (const uint8_t*)d; uint32_t s = 0; for (uint64_t i = 0; i < n; ++i) s += p[i]; return s; }
void t3o_lcg_pixels(void* px6, uint64_t n_px, uint32_t seed) { /* SURVEY §8d / BASELINE.md §3 */
    px_t* px = (px_t*)px6; uint32_t s = seed;
    for (uint64_t i = 0; i < n_px; ++i) {
        s = s * 1664525u + 1013904223u; px[i].Y = (uint16_t)((s >> 8) % 243);
        s = s * 1664525u + 1013904223u; px[i].Cb = (int16_t)((int)((s >> 8) % 81) - 40);
        s = s * 1664525u + 1013904223u; px[i].Cr = (int16_t)((int)((s >> 8) % 81) - 40);
    }
}
/* ---- row f1: RGB8 <-> quantised YCbCr (old/include/io_image.hpp:47-90).  float products and sums are rounded one by
 * one, left to right, as the reference's expressions evaluate on x86-64 SSE (no fused multiply-add: the Makefile passes
 * -ffp-contract=off); lround = round half away from zero; quantisation in double. ---- */
#include <math.h>
static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
void t3o_rgb_to_quant(const uint8_t* rgb, uint64_t n_px, void* px6) {
    uint8_t* o = (uint8_t*)px6;
    for (uint64_t i = 0; i < n_px; ++i) {
        const float r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        const float y = 0.299f * r + 0.587f * g + 0.114f * b;                       /* io_image.hpp:50 */
        const float cb = -0.168736f * r - 0.331264f * g + 0.5f * b + 128.0f;        /* :51 */
        const float cr = 0.5f * r - 0.418688f * g - 0.081312f * b + 128.0f;         /* :52 */
        const int Y = clampi((int)lroundf(y), 0, 255), Cb = clampi((int)lroundf(cb), 0, 255), Cr = clampi((int)lroundf(cr), 0, 255);
        const uint16_t Yq = (uint16_t)clampi((int)lround(Y * (242.0 / 255.0)), 0, 242);   /* :72 */
        const int16_t Cbq = (int16_t)clampi((int)lround((Cb - 128) * (40.0 / 128.0)), -40, 40);
        const int16_t Crq = (int16_t)clampi((int)lround((Cr - 128) * (40.0 / 128.0)), -40, 40);
        memcpy(o + 6 * i, &Yq, 2); memcpy(o + 6 * i + 2, &Cbq, 2); memcpy(o + 6 * i + 4, &Crq, 2);
    }
}
void t3o_quant_to_rgb(const void* px6, uint64_t n_px, uint8_t* rgb) {
    const uint8_t* p = (const uint8_t*)px6;
    for (uint64_t i = 0; i < n_px; ++i) {
        uint16_t Yq; int16_t Cbq, Crq; memcpy(&Yq, p + 6 * i, 2); memcpy(&Cbq, p + 6 * i + 2, 2); memcpy(&Crq, p + 6 * i + 4, 2);
        const uint8_t Y = (uint8_t)clampi((int)lround(Yq * (255.0 / 242.0)), 0, 255);      /* io_image.hpp:81 */
        const uint8_t Cb = (uint8_t)clampi((int)lround(128 + Cbq * (128.0 / 40.0)), 0, 255);
        const uint8_t Cr = (uint8_t)clampi((int)lround(128 + Crq * (128.0 / 40.0)), 0, 255);
        const float y = Y, cb = Cb - 128.0f, cr = Cr - 128.0f;                            /* :59 */
        const float r = y + 1.402f * cr, g = y - 0.344136f * cb - 0.714136f * cr, b = y + 1.772f * cb;
        rgb[3 * i] = (uint8_t)clampi((int)lroundf(r), 0, 255); rgb[3 * i + 1] = (uint8_t)clampi((int)lroundf(g), 0, 255); rgb[3 * i + 2] = (uint8_t)clampi((int)lroundf(b), 0, 255);
    }
}

void t3o_lcg_rgb(uint8_t* rgb, uint64_t n_px, uint32_t seed) { uint32_t s = seed; for (uint64_t i = 0; i < 3 * n_px; ++i) { s = s * 1664525u + 1013904223u; rgb[i] = (uint8_t)((s >> 8) % 256); } }

/* counter hash shared with the device injector (csrc/t3_kernels.hip: inject_errors_kernel) */
static uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
void t3o_inject_errors(void* words9, uint64_t first_sym, uint64_t n_blocks, uint32_t seed, int max_err) {
    uint8_t* s = (uint8_t*)words9 + first_sym;
    for (uint64_t b = 0; b < n_blocks; ++b) {
        uint32_t h = mix32(seed ^ mix32((uint32_t)b * 0x9e3779b9u + (uint32_t)(b >> 32)));
        int e = (int)(h % (uint32_t)(max_err + 1)); uint32_t used = 0;
        for (int q = 0; q < e; ++q) {
            h = mix32(h + 0x632be5abu);
            uint32_t pos = h % 26; while (used >> pos & 1) pos = (pos + 1) % 26; used |= 1u << pos;
            uint32_t trit = (h >> 8) % 3, delta = 1 + ((h >> 12) & 1);
            uint8_t t[3]; trits_of(s[26 * b + pos], t); t[trit] = (uint8_t)((t[trit] + delta) % 3);
            s[26 * b + pos] = sym_of(t[0], t[1], t[2]);
        }
    }
}

/* ---- centring blits (row f3), old/include/io_image.hpp:125-140 and :215-235 ----
 * PARITY UNPINNED like the rest of io_image.hpp (the header does not compile in this image): restated from the text.
 * blit: zeroed canvas, source rows copied at (x0, y0) = (max(0,(cw-sw)/2), max(0,(ch-sh)/2)), rows past the canvas dropped.
 * Requires sw <= cw (the reference overruns the canvas row otherwise). */
void t3o_blit_center_rgb(const uint8_t* src, int sw, int sh, uint8_t* dst, int cw, int ch) {
    memset(dst, 0, (size_t)cw * (size_t)ch * 3);
    const int x0 = (cw - sw) / 2 > 0 ? (cw - sw) / 2 : 0, y0 = (ch - sh) / 2 > 0 ? (ch - sh) / 2 : 0;
    for (int y = 0; y < sh; ++y) {
        if (y + y0 < 0 || y + y0 >= ch) continue;
        memcpy(dst + (size_t)(y + y0) * cw * 3 + (size_t)x0 * 3, src + (size_t)y * sw * 3, (size_t)sw * 3);
    }
}
/* extract: the sw x sh window at (x0, y0) of a fw x fh frame of 6-byte pixels; rows below the frame are zero pixels.  sw <= fw. */
void t3o_extract_center_q(const void* full_px6, int fw, int fh, void* sub_px6, int sw, int sh) {
    const uint8_t* f = (const uint8_t*)full_px6; uint8_t* o = (uint8_t*)sub_px6;
    const int x0 = (fw - sw) / 2 > 0 ? (fw - sw) / 2 : 0, y0 = (fh - sh) / 2 > 0 ? (fh - sh) / 2 : 0;
    for (int y = 0; y < sh; ++y) {
        const int fy = y + y0;
        if (fy < 0 || fy >= fh) { memset(o + (size_t)y * sw * 6, 0, (size_t)sw * 6); continue; }
        memcpy(o + (size_t)y * sw * 6, f + ((size_t)fy * fw + (size_t)x0) * 6, (size_t)sw * 6);
    }
}
