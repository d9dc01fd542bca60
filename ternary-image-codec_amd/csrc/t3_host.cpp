// t3_host.cpp — see t3_host.hpp.  Host-only; compiled into libt3hip.so next to the HIP code.
#include "t3_host.hpp"

#include <string.h>

namespace t3 {

// ---- GF(27) = GF(3)[x] / (x^3 + 2x + 1) -----------------------------------------------------------
namespace {
struct V3 { int t[3]; };
V3 to_v(int s) { return V3{{s % 3, (s / 3) % 3, (s / 9) % 3}}; }
int to_s(const V3& v) { return v.t[0] + 3 * v.t[1] + 9 * v.t[2]; }
int mul_poly(int a, int b) {   // schoolbook product, then x^3 -> x + 2, x^4 -> x^2 + 2x  (gf27_mul_poly OLD:402-413)
    V3 x = to_v(a), y = to_v(b);
    int c[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c[i + j] += x.t[i] * y.t[j];
    c[2] += c[4]; c[1] += 2 * c[4];
    c[1] += c[3]; c[0] += 2 * c[3];
    return (c[0] % 3) + 3 * (c[1] % 3) + 9 * (c[2] % 3);
}
Field make_field() {
    Field f; memset(&f, 0, sizeof f);
    for (int a = 0; a < 27; ++a) {
        V3 x = to_v(a);
        for (int b = 0; b < 27; ++b) {
            V3 y = to_v(b);
            f.t.add[a * 27 + b] = (uint8_t)to_s(V3{{(x.t[0] + y.t[0]) % 3, (x.t[1] + y.t[1]) % 3, (x.t[2] + y.t[2]) % 3}});
            f.t.mul[a * 27 + b] = (uint8_t)mul_poly(a, b);
        }
        f.t.neg[a] = (uint8_t)to_s(V3{{(3 - x.t[0]) % 3, (3 - x.t[1]) % 3, (3 - x.t[2]) % 3}});
    }
    // smallest element >= 2 of multiplicative order 26 (GF27Context::init OLD:438-447) -> 3 (= "x")
    int prim = 0;
    for (int c = 2; c < 27 && !prim; ++c) {
        int x = 1, ord = 0;
        for (int i = 1; i <= 26; ++i) { x = f.t.mul[x * 27 + c]; if (x == 1) { ord = i; break; } }
        if (ord == 26) prim = c;
    }
    f.prim = (uint8_t)(prim ? prim : 3);
    for (int i = 0; i < 27; ++i) f.log[i] = -1;
    int x = 1;
    for (int i = 0; i < 26; ++i) { f.t.exp[i] = (uint8_t)x; f.log[x] = (int16_t)i; x = f.t.mul[x * 27 + f.prim]; }
    for (int i = 0; i < 78; ++i) f.exp78[i] = f.t.exp[i % 26];
    f.t.inv[0] = 0;
    for (int a = 1; a < 27; ++a) f.t.inv[a] = f.t.exp[(26 - f.log[a]) % 26];
    return f;
}
}  // namespace

const Field& field() { static const Field f = make_field(); return f; }

bool valid_k(int k) { return k == 24 || k == 22 || k == 20 || k == 18; }
int k_of_band_profile(uint8_t bp) { static const int ks[4] = {24, 22, 20, 18}; return ks[bp % 4]; }

void rs_generator(int k, uint8_t* g) {
    const Field& F = field(); RsView v = rs_view(F.t);
    const int r = 26 - k;
    uint8_t cur[10] = {1}, nxt[10]; int len = 1;
    for (int i = 1; i <= r; ++i) {            // multiply by (x - alpha^i); coefficient index = degree
        memset(nxt, 0, sizeof nxt);
        const uint8_t root = v.P(i);
        for (int j = 0; j < len; ++j) { nxt[j] = v.S(nxt[j], v.M(cur[j], root)); nxt[j + 1] = v.A(nxt[j + 1], cur[j]); }
        memcpy(cur, nxt, sizeof cur); ++len;
    }
    memcpy(g, cur, (size_t)(r + 1));
}

static void parity_compat(int k, const uint8_t* g, const uint8_t* d, uint8_t* par) {
    // The reference's in-place recurrence T[i+j] -= g[j]*T[i] for i<k, j<=r (OLD:522-531). T[i] itself is
    // never read again after step i, so only the r-symbol window ahead of i matters.
    RsView v = rs_view(field().t);
    const int r = 26 - k; uint8_t T[27];
    memset(T, 0, sizeof T); memcpy(T, d, (size_t)k);
    for (int i = 0; i < k; ++i) {
        const uint8_t coef = T[i];
        if (!coef) continue;
        for (int j = 0; j <= r; ++j) T[i + j] = v.S(T[i + j], v.M(g[j], coef));
    }
    memcpy(par, T + k, (size_t)r);
}

static void parity_fixed(int k, const uint8_t* d, uint8_t* par) {
    // solve sum_m p_m alpha^{j(k+m)} = - sum_i d_i alpha^{j i}, j = 1..r  (roots as OLD:551-561 evaluates them)
    const Field& F = field(); RsView v = rs_view(F.t);
    const int r = 26 - k; uint8_t A[8][9];
    for (int j = 0; j < r; ++j) {
        for (int m = 0; m < r; ++m) A[j][m] = v.P((j + 1) * (k + m));
        uint8_t s = 0;
        for (int i = 0; i < k; ++i) s = v.A(s, v.M(d[i], v.P((j + 1) * i)));
        A[j][r] = F.t.neg[s];
    }
    for (int c = 0; c < r; ++c) {
        int piv = c; while (piv < r && !A[piv][c]) ++piv;
        if (piv != c) for (int m = 0; m <= r; ++m) { uint8_t t = A[c][m]; A[c][m] = A[piv][m]; A[piv][m] = t; }
        const uint8_t iv = F.t.inv[A[c][c]];
        for (int m = 0; m <= r; ++m) A[c][m] = v.M(A[c][m], iv);
        for (int j = 0; j < r; ++j) if (j != c && A[j][c]) {
            const uint8_t f = A[j][c];
            for (int m = 0; m <= r; ++m) A[j][m] = v.S(A[j][m], v.M(f, A[c][m]));
        }
    }
    for (int m = 0; m < r; ++m) par[m] = A[m][r];
}

void rs_parity_matrix(int k, int mode, uint8_t* P) {
    const int r = 26 - k; uint8_t g[10], e[24];
    rs_generator(k, g);
    for (int i = 0; i < k; ++i) {
        memset(e, 0, sizeof e); e[i] = 1;
        if (mode == T3_MODE_FIXED) parity_fixed(k, e, P + i * r); else parity_compat(k, g, e, P + i * r);
    }
}

bool rs_decode_host(int k, uint8_t* c, bool fixed) {
    RsView v = rs_view(field().t);
    switch (26 - k) {
        case 2: return rs_decode_block<2>(v, c, fixed);
        case 4: return rs_decode_block<4>(v, c, fixed);
        case 6: return rs_decode_block<6>(v, c, fixed);
        case 8: return rs_decode_block<8>(v, c, fixed);
    }
    return false;
}

// ---- encode LUT ------------------------------------------------------------------------------------
LutGeom lut_geom(int k) {
    LutGeom g; g.k = k; g.r = 26 - k;
    switch (g.r) {
        case 2: g.n_dw = 3; g.var_dw = 2; g.var_off = 256; g.slab_bytes = 1024; break;
        case 4: g.n_dw = 3; g.var_dw = 2; g.var_off = 256; g.slab_bytes = 1024; break;
        case 6: g.n_dw = 4; g.var_dw = 2; g.var_off = 256; g.slab_bytes = 1024; break;
        default: g.n_dw = 5; g.var_dw = 4; g.var_off = 512; g.slab_bytes = 1280; break;
    }
    g.total_bytes = k * g.slab_bytes;
    return g;
}

static uint8_t add13(uint8_t d, int s) { return field().t.add[d * 27 + 13 * s]; }   // + (s,s,s) trit-wise

void build_encode_lut(int k, int mode, std::vector<uint32_t>& image) {
    const Field& F = field();
    const LutGeom G = lut_geom(k); const int r = G.r, nmain = r < 5 ? r : 5;
    std::vector<uint8_t> P((size_t)k * r); rs_parity_matrix(k, mode, P.data());
    image.assign((size_t)G.total_bytes / 4, 0u);
    for (int p = 0; p < k; ++p) {
        uint32_t* slab = image.data() + (size_t)p * G.slab_bytes / 4;
        for (int d = 0; d < 27; ++d) {
            uint32_t dw[6] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < r; ++j) {
                const int c = F.t.mul[d * 27 + P[p * r + j]];
                const int tr[3] = {c % 3, (c / 3) % 3, c / 9};
                if (j < nmain) for (int q = 0; q < 3; ++q) dw[q] |= (uint32_t)tr[q] << (6 * j);
                else for (int q = 0; q < 3; ++q) { const int f = 3 * (j - 5) + q; dw[3 + f / 5] |= (uint32_t)tr[q] << (6 * (f % 5)); }
            }
            slab[2 * d] = dw[0]; slab[2 * d + 1] = dw[1];                       // table A: {dw0,dw1}
            if (r == 8) { slab[64 + 2 * d] = dw[2]; slab[64 + 2 * d + 1] = dw[3]; }  // table B (r=8): {dw2,dw3}
            for (int s = 0; s < 3; ++s) {                                        // variant tables V_s, s = scrambler state
                uint32_t* v = slab + G.var_off / 4 + 64 * s + 2 * d;
                const uint32_t img = (uint32_t)add13((uint8_t)d, s) << 24;       // scrambled image of d in the top byte
                if (r == 6) { v[0] = dw[2]; v[1] = dw[3] | img; }
                else { v[0] = dw[G.var_dw] | img; v[1] = 0; }
            }
        }
    }
}

void build_mfma_encode(int k, int mode, std::vector<uint32_t>& afrag, std::vector<uint32_t>& lds_img) {
    const Field& F = field();
    const int r = 26 - k, H = r / 2;
    std::vector<uint8_t> P((size_t)k * r); rs_parity_matrix(k, mode, P.data());
    afrag.assign(3 * 64 * 4, 0u);
    for (int s = 0; s < 3; ++s) for (int l = 0; l < 64; ++l) for (int d = 0; d < 4; ++d) {
        const int m = l & 31, kh = l >> 5, hh = (m >> 2) & 1, i = (m & 3) + 4 * (m >> 3);
        const int p = 8 * s + 4 * kh + d;
        if (i >= 3 * H) continue;
        const int j = hh * H + i / 3, t = i % 3;
        uint32_t dw = 0;
        if (p < k) {
            for (int tt = 0; tt < 3; ++tt) {
                const int e = tt == 0 ? 1 : tt == 1 ? 3 : 9;                  // the symbol whose only non-zero trit is tt
                const int c = F.t.mul[e * 27 + P[p * r + j]];
                const int tr[3] = {c % 3, (c / 3) % 3, c / 9};
                dw |= (uint32_t)(tr[t] == 2 ? 0xFFu : (uint32_t)tr[t]) << (8 * tt);   // signed: 2 == -1
            }
        } else if (r >= 4 && (p == mfma_scr_pos(k) || p == mfma_scr_pos(k) + 1)) {
            for (int bb = 0; bb < 4; ++bb) if (4 * (p - mfma_scr_pos(k)) + bb == j) dw |= 1u << (8 * bb);
        }
        afrag[(size_t)(s * 64 + l) * 4 + d] = dw;
    }
    lds_img.assign(kMfmaLdsBytes / 4, 0u);
    auto sb = [](int t) { return (uint32_t)(t == 2 ? 0xFF : t); };
    for (int v = 0; v < 3; ++v) for (int d = 0; d < 27; ++d) {
        const uint32_t dw = sb(d % 3) | sb((d / 3) % 3) << 8 | sb(d / 9) << 16 | (uint32_t)add13((uint8_t)d, v) << 24;
        for (int c = 0; c < 32; ++c) lds_img[(size_t)v * kMfmaTState / 4 + (size_t)d * 32 + c] = dw;
    }
    uint8_t* mt = (uint8_t*)lds_img.data() + kMfmaModOff;
    for (int t = 0; t < 3; ++t) for (int x = 0; x < 128; ++x) mt[128 * t + x] = (uint8_t)((((x - 64) % 3 + 3) % 3) * (t == 0 ? 1 : t == 1 ? 3 : 9));
}

void mfma_scrambler_table(int k, const ScrCycle& sc, uint32_t out[12]) {
    const int r = 26 - k;
    for (int c = 0; c < 6; ++c) {
        uint32_t dw[2] = {0, 0};
        for (int j = 0; j < r && j < 8; ++j) dw[j / 4] |= (uint32_t)sc.cyc[(c + j) % 6] << (8 * (j % 4));
        out[2 * c] = dw[0]; out[2 * c + 1] = dw[1];
    }
}

int syndrome_lut_slab(int k) { return 26 - k == 8 ? 768 : 512; }
void build_syndrome_lut(int k, std::vector<uint32_t>& image) {
    const Field& F = field(); RsView v = rs_view(F.t);
    const int r = 26 - k, nmain = r < 5 ? r : 5, slab = syndrome_lut_slab(k);
    image.assign((size_t)26 * slab / 4, 0u);
    for (int i = 0; i < 26; ++i) {
        uint32_t* sl = image.data() + (size_t)i * slab / 4;
        for (int c = 0; c < 27; ++c) {
            uint32_t dw[6] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < r; ++j) {
                const int y = v.M((uint8_t)c, v.P((j + 1) * i));
                const int tr[3] = {y % 3, (y / 3) % 3, y / 9};
                if (j < nmain) for (int q = 0; q < 3; ++q) dw[q] |= (uint32_t)tr[q] << (6 * j);
                else for (int q = 0; q < 3; ++q) { const int f = 3 * (j - 5) + q; dw[3 + f / 5] |= (uint32_t)tr[q] << (6 * (f % 5)); }
            }
            for (int q = 0; q < (r == 8 ? 5 : 4); ++q) sl[32 * q + c] = dw[q];    // table q of the position: 27 dwords (128-byte pitch)
        }
    }
}

void build_mfma_syndrome(int k, std::vector<uint32_t>& afrag) {
    const Field& F = field(); RsView v = rs_view(F.t);
    const int r = 26 - k, H = r / 2;
    afrag.assign(4 * 64 * 4, 0u);
    for (int s = 0; s < 4; ++s) for (int l = 0; l < 64; ++l) for (int d = 0; d < 4; ++d) {
        const int m = l & 31, kh = l >> 5, hh = (m >> 2) & 1, i = (m & 3) + 4 * (m >> 3);
        const int q = 4 * s + d;
        if (q >= 13 || i >= 3 * H) continue;
        const int p = 13 * kh + q, j = hh * H + i / 3, t = i % 3;
        const uint8_t g = v.P((j + 1) * p);
        uint32_t dw = 0;
        for (int tt = 0; tt < 3; ++tt) {
            const int e = tt == 0 ? 1 : tt == 1 ? 3 : 9;                      // the symbol whose only non-zero trit is tt
            const int c = F.t.mul[e * 27 + g];
            const int tr[3] = {c % 3, (c / 3) % 3, c / 9};
            dw |= (uint32_t)(tr[t] == 2 ? 0xFFu : (uint32_t)tr[t]) << (tt == 2 ? 24 : 8 * tt);
        }
        afrag[(size_t)(s * 64 + l) * 4 + d] = dw;
    }
}
void build_syndrome_T(std::vector<uint32_t>& img, int copies) {
    const Field& F = field();
    img.assign((size_t)3 * 27 * copies, 0u);
    auto sb = [](int t) { return (uint32_t)(t == 2 ? 0xFF : t); };
    for (int st = 0; st < 3; ++st) for (int c = 0; c < 27; ++c) {
        const int d = F.t.add[c * 27 + F.t.neg[13 * st]];                      // c - (st,st,st) trit-wise (descramble_symbol OLD:88-94)
        const uint32_t dw = sb(d % 3) | sb((d / 3) % 3) << 8 | (uint32_t)d << 16 | sb(d / 9) << 24;
        for (int b = 0; b < copies; ++b) img[((size_t)st * 27 + c) * copies + b] = dw;
    }
}
void build_fx2_small(uint8_t out[kFx2SmallBytes]) {
    const Field& F = field();
    memset(out, 0, kFx2SmallBytes);
    for (int x = 0; x < 27; ++x) {
        out[kFx2LG + x] = x ? (uint8_t)F.log[x] : 0xFF;
        out[kFx2INV + x] = F.t.inv[x]; out[kFx2NEG + x] = F.t.neg[x]; out[kFx2NINV + x] = F.t.neg[F.t.inv[x]];
    }
    for (int i = 0; i < 26; ++i) out[kFx2EX + i] = F.t.exp[i];
}
void build_fx2_mod(uint8_t out[kFx2ModBytes]) {
    for (int t = 0; t < 3; ++t) for (int x = 0; x < 160; ++x) out[160 * t + x] = (uint8_t)((((x - 78) % 3 + 3) % 3) * (t == 0 ? 1 : t == 1 ? 3 : 9));   // index = trit sum + 64 + 14
}

// ---- scrambler ----------------------------------------------------------------------------------------
ScrCycle scrambler_cycle_from_next(const uint8_t next[3], uint32_t s0) {
    ScrCycle c; memset(&c, 0, sizeof c);
    for (int i = 0; i < 3; ++i) c.next[i] = next[i];
    uint32_t st = s0 % 3; uint8_t seq[8];
    for (int i = 0; i < 8; ++i) { st = next[st]; seq[i] = (uint8_t)st; }   // state is stepped before use (OLD:83)
    c.pre[0] = seq[0]; c.pre[1] = seq[1];
    // a map on 3 points has pre-period <= 2 and period in {1,2,3}; from index 2 on the sequence is 6-periodic
    for (int i = 0; i < 6; ++i) c.cyc[i] = seq[2 + i];
    uint32_t f = 0; for (int i = 0; i < 6; ++i) f |= (uint32_t)c.cyc[i] << (2 * i);
    c.cyc24 = f | f << 12;
    return c;
}
ScrCycle scrambler_cycle(uint32_t a, uint32_t b, uint32_t s0) {
    uint8_t next[3];
    for (uint32_t s = 0; s < 3; ++s) next[s] = (uint8_t)((a * s + b) % 3u);   // uint32 wrap-around as in OLD:83
    return scrambler_cycle_from_next(next, s0);
}

// ---- ternary CRC-12 and the 27-symbol header ---------------------------------------------------------------
void crc12(const uint8_t* msg, int n, uint8_t out[12]) {
    // register r[0..11], feedback fb = in + r[11]; r <- shift, with fb added into taps 3, 4, 7 and loaded into 0
    // (OLD:183-200); the message is followed by 12 zero trits (OLD:202).
    uint8_t r[12]; memset(r, 0, sizeof r);
    for (int i = 0; i < n + 12; ++i) {
        const int fb = ((i < n ? msg[i] : 0) + r[11]) % 3;
        for (int j = 11; j > 0; --j) r[j] = r[j - 1];
        r[0] = (uint8_t)fb;
        r[3] = (uint8_t)((r[3] + fb) % 3); r[4] = (uint8_t)((r[4] + fb) % 3); r[7] = (uint8_t)((r[7] + fb) % 3);
    }
    memcpy(out, r, 12);
}
static const int kCrcSlots[4] = {20, 21, 22, 26};
static void header_crc_of(const uint8_t s[27], uint8_t r[12]) {
    uint8_t tr[81]; int n = 0;
    for (int i = 0; i < 27; ++i) {
        if (i == 20 || i == 21 || i == 22 || i == 26) continue;
        tr[n++] = s[i] % 3; tr[n++] = (s[i] / 3) % 3; tr[n++] = (s[i] / 9) % 3;
    }
    crc12(tr, n, r);
}
static uint8_t subword_code(uint8_t sub) { return sub == 24 ? 1 : sub == 21 ? 2 : sub == 18 ? 3 : sub == 15 ? 4 : 0; }

void header_pack(const t3_cfg& c, uint32_t frame_seq, uint32_t band_map_hash, uint8_t s[27]) {
    memset(s, 0, 27);
    const uint32_t magic = 0x0A2;                       // SuperframeHeader defaults OLD:157-158
    s[0] = magic % 27; s[1] = (magic / 27) % 27; s[2] = 1; s[3] = c.profile % 27;
    for (int g = 0; g < 3; ++g) {                       // three bands per symbol, most significant first (OLD:219-228)
        uint32_t u = 0;
        for (int i = 0; i < 3; ++i) u = u * 3 + c.band_profile[3 * g + i] % 3;
        s[4 + g] = (uint8_t)u;
    }
    s[7] = c.tile_w % 27; s[8] = c.tile_h % 27;
    s[9] = c.seed_a % 27; s[10] = c.seed_b % 27; s[11] = c.seed_s0 % 27;
    s[12] = (uint8_t)((subword_code(c.subword) + (c.centered ? 9 : 0)) % 27);
    s[13] = band_map_hash % 27; s[14] = (band_map_hash / 27) % 27; s[15] = (band_map_hash / 729) % 27;
    s[16] = c.coset % 3;
    s[17] = frame_seq % 27; s[18] = (frame_seq / 27) % 27; s[19] = (frame_seq / 729) % 27;
    s[23] = c.beacon_enabled ? 1 : 0; s[24] = c.beacon_band_slot % 27;
    s[25] = (uint8_t)(c.beacon_words_period < 26 ? c.beacon_words_period : 26);   // clamped, not reduced (OLD:267)
    uint8_t r[12]; header_crc_of(s, r);
    for (int q = 0; q < 4; ++q) s[kCrcSlots[q]] = (uint8_t)(r[3 * q] + 3 * r[3 * q + 1] + 9 * r[3 * q + 2]);
}
bool header_check(const uint8_t s[27]) {
    uint8_t r[12]; header_crc_of(s, r);
    for (int q = 0; q < 4; ++q) {
        const uint8_t v = s[kCrcSlots[q]];
        if (r[3 * q] != v % 3 || r[3 * q + 1] != (v / 3) % 3 || r[3 * q + 2] != (v / 9) % 3) return false;
    }
    return true;
}
void header_unpack(const uint8_t in[27], t3_cfg& o, uint32_t* frame_seq, uint32_t* band_map_hash) {
    static const uint8_t subs[5] = {27, 24, 21, 18, 15};
    uint8_t s[27]; for (int i = 0; i < 27; ++i) s[i] = in[i] % 27;
    o.profile = s[3] % 5;
    for (int g = 0; g < 3; ++g) { uint32_t v = s[4 + g]; for (int i = 0; i < 3; ++i) { o.band_profile[3 * g + i] = v % 3; v /= 3; } }   // LSD first (OLD:327-340)
    o.tile_w = s[7]; o.tile_h = s[8]; o.seed_a = s[9]; o.seed_b = s[10]; o.seed_s0 = s[11];
    const uint8_t sub = s[12] % 9;
    o.subword = sub < 5 ? subs[sub] : 27; o.centered = ((s[12] / 9) % 3) != 0;
    if (band_map_hash) *band_map_hash = s[13] + 27u * s[14] + 729u * s[15];
    o.coset = s[16] % 3;
    if (frame_seq) *frame_seq = s[17] + 27u * s[18] + 729u * s[19];
    o.beacon_enabled = s[23] != 0; o.beacon_band_slot = s[24] % 9; o.beacon_words_period = s[25];
}
uint8_t beacon_symbol(uint8_t profile, uint16_t fsm, uint8_t health) {
    return (uint8_t)((profile + 5 * (fsm % 5) + 15 * (health % 3)) % 27);
}

// FIXED-mode extension symbols: what the 27-symbol header loses to its %27 fields
static void fixed_ext(const t3_cfg& c, uint64_t n_raw, uint8_t e[27]) {
    uint32_t v = c.tile_w / 27u; for (int i = 0; i < 3; ++i) { e[i] = v % 27; v /= 27; }
    v = c.tile_h / 27u;          for (int i = 0; i < 3; ++i) { e[3 + i] = v % 27; v /= 27; }
    v = c.beacon_words_period;   for (int i = 0; i < 7; ++i) { e[6 + i] = v % 27; v /= 27; }
    uint64_t q = n_raw;          for (int i = 0; i < 10; ++i) { e[13 + i] = (uint8_t)(q % 27); q /= 27; }
    for (int g = 0; g < 3; ++g) { uint8_t f = 0; for (int j = 0; j < 3; ++j) if (c.band_profile[3 * g + j] % 4 == 3) f |= (uint8_t)(1 << j); e[23 + g] = f; }
    const ScrCycle sc = scrambler_cycle(c.seed_a, c.seed_b, c.seed_s0);
    e[26] = (uint8_t)(sc.next[0] + 3 * sc.next[1] + 9 * sc.next[2]);
}

int header_encode(const t3_cfg& cin, uint64_t n_raw, uint8_t* out) {
    uint8_t hp[27], g[10], blk[18];
    if (cin.mode == T3_MODE_FIXED) {
        t3_cfg c = cin; for (int b = 0; b < 9; ++b) c.band_profile[b] %= 4;
        uint8_t ext[27]; header_pack(c, 0, 0, hp); fixed_ext(c, n_raw, ext);
        memcpy(blk, hp, 18);                               memcpy(out, blk, 18);      parity_fixed(18, blk, out + 18);
        memcpy(blk, hp + 18, 9); memcpy(blk + 9, ext, 9);  memcpy(out + 26, blk, 18); parity_fixed(18, blk, out + 44);
        memcpy(blk, ext + 9, 18);                          memcpy(out + 52, blk, 18); parity_fixed(18, blk, out + 70);
        memset(out + 78, 0, 12);                           // 10 whole words: the body starts word- and 2-byte-aligned
        return 90;
    }
    header_pack(cin, 0, 0, hp);                            // frame_seq / band_map_hash are never set by the encoder (OLD:1142-1150)
    rs_generator(18, g);
    memcpy(blk, hp, 18);                                   memcpy(out, blk, 18);      parity_compat(18, g, blk, out + 18);
    memcpy(blk, hp + 18, 9); memset(blk + 9, 0, 9);        memcpy(out + 26, blk, 18); parity_compat(18, g, blk, out + 44);
    return 52;
}

int header_parse(const uint8_t* w, uint64_t n_words, int mode, t3_cfg& seen, uint64_t* n_raw, uint8_t next[3]) {
    const bool fixed = mode == T3_MODE_FIXED;
    const int nblk = fixed ? 3 : 2;
    if (n_words < (uint64_t)(fixed ? 10 : 6)) return T3_E_HEADER;        // OLD:920
    uint8_t blk[3][26];
    for (int q = 0; q < nblk; ++q) {
        for (int i = 0; i < 26; ++i) blk[q][i] = w[26 * q + i] % 27;     // symbols >= 27 are UB in the reference; reduced here
        if (!rs_decode_host(18, blk[q], fixed)) return T3_E_HEADER;      // OLD:929-930
    }
    uint8_t hp[27]; memcpy(hp, blk[0], 18); memcpy(hp + 18, blk[1], 9);
    if (!header_check(hp)) return T3_E_HEADER;                            // OLD:934
    const uint8_t keep_mode = seen.mode; const uint32_t keep_sf = seen.superframe_words;
    header_unpack(hp, seen, nullptr, nullptr);                            // OLD:1006-1013
    seen.mode = keep_mode; seen.superframe_words = keep_sf;
    if (!fixed) {
        const ScrCycle sc = scrambler_cycle(seen.seed_a, seen.seed_b, seen.seed_s0);
        if (next) memcpy(next, sc.next, 3);
        if (n_raw) *n_raw = 0;
        return T3_OK;
    }
    // the 27-symbol header packs each band triple most-significant-first but unpacks it least-significant-first
    // (OLD:222-224 vs 327-340); v6c keeps the reference's pack and undoes the reversal here
    for (int g3 = 0; g3 < 3; ++g3) { const uint8_t t = seen.band_profile[3 * g3]; seen.band_profile[3 * g3] = seen.band_profile[3 * g3 + 2]; seen.band_profile[3 * g3 + 2] = t; }
    uint8_t e[27]; memcpy(e, blk[1] + 9, 9); memcpy(e + 9, blk[2], 18);
    uint32_t v = 0; uint64_t q = 0;
    for (int i = 2; i >= 0; --i) v = v * 27 + e[i];
    seen.tile_w = (uint16_t)(seen.tile_w + 27 * v);
    v = 0; for (int i = 2; i >= 0; --i) v = v * 27 + e[3 + i];
    seen.tile_h = (uint16_t)(seen.tile_h + 27 * v);
    for (int i = 6; i >= 0; --i) q = q * 27 + e[6 + i];
    seen.beacon_words_period = (uint32_t)q;
    q = 0; for (int i = 9; i >= 0; --i) q = q * 27 + e[13 + i];
    if (n_raw) *n_raw = q;
    for (int i = 0; i < 9; ++i) if ((e[23 + i / 3] >> (i % 3)) & 1) seen.band_profile[i] = 3;
    if (next) { next[0] = e[26] % 3; next[1] = (e[26] / 3) % 3; next[2] = (e[26] / 9) % 3; }
    seen.mode = T3_MODE_FIXED;
    return T3_OK;
}

// ---- geometry --------------------------------------------------------------------------------------------------
bool want_interleave(const t3_cfg& c) { return c.profile == T3_P5_RS26_22_2D && c.tile_w && c.tile_h; }

int plan(uint64_t W, const t3_cfg& c, t3_layout& L) {
    memset(&L, 0, sizeof L);
    L.n_raw_words = W;
    if (c.profile == T3_RAW_MODE) { L.out_words = W; L.out_syms = 9 * W; return T3_OK; }
    const bool fixed = c.mode == T3_MODE_FIXED;
    if (c.mode > T3_MODE_FIXED) return T3_E_ARG;
    if (W > (1ull << 31) * 3 / 26) return T3_E_ARG;                   // stream symbol indices are 31-bit on the device
    L.beacon_on = (c.beacon_enabled && c.beacon_words_period > 0) ? 1 : 0;
    if (fixed) {
        if (c.profile > T3_P5_RS26_22_2D) return T3_E_ARG;
        if (L.beacon_on && c.beacon_band_slot >= 9) return T3_E_ARG;
    }
    L.n_sym = (26 * W + 2) / 3;                                         // OLD:1051-1082 (last symbol zero-padded)
    L.interleave2d = want_interleave(c) ? 1 : 0;
    L.header_syms = fixed ? 90 : 52;
    uint64_t off = 0;
    for (int b = 0; b < 9; ++b) {
        const uint64_t k = (uint64_t)k_of_band_profile(c.band_profile[b]);
        L.band_k[b] = (uint8_t)k;
        L.band_len[b] = L.n_sym > (uint64_t)b ? (L.n_sym - b + 8) / 9 : 0;   // bands[i%9] (OLD:1088)
        L.band_blocks[b] = fixed ? (L.band_len[b] + k - 1) / k : L.band_len[b] / k;   // tail dropped in COMPAT (OLD:1107)
        L.band_body_off[b] = off;
        off += 26 * L.band_blocks[b];
    }
    L.body_syms = off;
    L.body_syms_framed = off;
    if (L.beacon_on) {
        // OLD:1123-1139: words are emitted while body symbols remain; word w carries a beacon in slot band_slot
        // iff w % period == 0 (never, if band_slot >= 9, but the body is still laid out in whole words).
        const uint64_t P = c.beacon_words_period; const bool hit = c.beacon_band_slot < 9;
        uint64_t lo = 0, hi = off / 8 + 2;               // smallest n with 9n - beacons(n) >= off
        while (lo < hi) {
            const uint64_t n = (lo + hi) / 2, room = 9 * n - (hit ? (n + P - 1) / P : 0);
            if (room >= off) hi = n; else lo = n + 1;
        }
        L.body_syms_framed = 9 * lo;
    }
    L.out_syms = L.header_syms + L.body_syms_framed;
    L.out_words = (L.out_syms + 8) / 9;                                  // OLD:1164
    return T3_OK;
}

void plan_decode_compat(uint64_t n_in, const t3_cfg& s, DecLayoutCompat& D) {
    memset(&D, 0, sizeof D);
    D.body_words = n_in >= 6 ? n_in - 6 : 0;
    const bool skip = s.beacon_enabled && s.beacon_words_period > 0;    // OLD:952
    uint64_t off = 0;
    for (int b = 0; b < 9; ++b) {
        uint64_t n = D.body_words;
        if (skip && b == s.beacon_band_slot) n -= (D.body_words + s.beacon_words_period - 1) / s.beacon_words_period;
        D.band_syms[b] = n; D.band_blocks[b] = n / 26;                   // OLD:983
        D.band_k[b] = (uint8_t)k_of_band_profile(s.band_profile[b]);
        D.use_off[b] = off; off += D.band_blocks[b] * D.band_k[b];
    }
    D.use_syms = off;
    D.out_words = off * 3 / 26;                                          // OLD:1030
}

FastDiv fastdiv(uint32_t d) {
    FastDiv f; f.d = d; f.mul = 0; f.sh = 0;
    if (d <= 1) return f;
    uint32_t s = 0; while ((1ull << (s + 1)) < d) ++s;                   // s = ceil(log2 d) - 1
    f.sh = s;
    f.mul = (uint32_t)((((unsigned __int128)1 << (32 + s)) + d - 1) / d);  // exact for n < 2^31
    return f;
}

}  // namespace t3
