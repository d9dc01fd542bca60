// t3_rs_core.h — GF(27) Reed–Solomon RS(26,k) block decoder, shared by the host (superframe header,
// 2-3 blocks per frame) and the device (K4: one lane per flagged block).  Replaces
// RSCodec::decode_block (OLD:546-662).  Written over fixed-size coefficient arrays whose *lengths* follow
// the reference's std::vector sizes step for step, because the reference's results on garbage input
// (miscorrections, early `false`) depend on them and COMPAT mode must reproduce those bit for bit.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define T3_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define T3_HD inline
#endif

namespace t3 {

// Tables the decoder needs, laid out so one pointer serves host memory, __constant__ or LDS.
struct RsTables {
    uint8_t mul[729];   // a*27+b -> a.b              (GF27Context::mul OLD:473-476)
    uint8_t add[729];   // a*27+b -> a+b (trit-wise)  (gf27_add OLD:383-388)
    uint8_t neg[27];    // -a
    uint8_t inv[27];    // a^-1, inv[0]=0            (OLD:459-465)
    uint8_t exp[26];    // alpha^i                    (OLD:450-456)
    uint8_t pad_[3];
};

struct RsView {
    const uint8_t* mul; const uint8_t* add; const uint8_t* neg; const uint8_t* inv; const uint8_t* exp;
    T3_HD uint8_t M(uint8_t a, uint8_t b) const { return mul[a * 27 + b]; }
    T3_HD uint8_t A(uint8_t a, uint8_t b) const { return add[a * 27 + b]; }
    T3_HD uint8_t S(uint8_t a, uint8_t b) const { return add[a * 27 + neg[b]]; }
    T3_HD uint8_t P(int e) const { e %= 26; if (e < 0) e += 26; return exp[e]; }   // pow_alpha OLD:479-482
};

// Syndromes S_j = sum_i c_i alpha^{(j+1) i}, j = 0..R-1 (OLD:549-561). Returns true when all are zero.
template <int R>
T3_HD bool rs_syndromes(const RsView& g, const uint8_t* c, uint8_t* S) {
    bool zero = true;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        uint8_t acc = 0;
#pragma unroll
        for (int i = 0; i < 26; ++i) acc = g.A(acc, g.M(c[i], g.exp[((j + 1) * i) % 26]));   // exponent is a compile-time constant
        S[j] = acc;
        zero = zero && acc == 0;
    }
    return zero;
}

// Everything after the syndromes: Berlekamp–Massey (OLD:567-605), Omega (OLD:606-610), Chien (OLD:611-624),
// formal derivative in characteristic 3 (OLD:625-641), Forney (OLD:642-659).
// fixed=false: reference behaviour, magnitude ADDED (OLD:658).  fixed=true: magnitude subtracted and the
// block is rejected unless #roots == deg(sigma).  Corrects c in place; returns decode_block's bool.
template <int R>
T3_HD bool rs_correct(const RsView& g, uint8_t* c, const uint8_t* S, bool fixed) {
    constexpr int T = R / 2, NP = R + 2;          // polynomial lengths never exceed R+1
    uint8_t sg[NP], B[NP], tmp[NP];
    for (int i = 0; i < NP; ++i) { sg[i] = 0; B[i] = 0; }
    sg[0] = 1; B[0] = 1;
    int ns = 1, nB = 1, L = 0, m = 1;
    for (int n = 0; n < R; ++n) {
        uint8_t d = S[n];
        for (int i = 1; i <= L; ++i) if (i < ns) d = g.A(d, g.M(sg[i], S[n - i]));
        if (d != 0) {
            const int nsh = m + nB, nd = ns > nsh ? ns : nsh, nT = ns;
            for (int i = 0; i < NP; ++i) tmp[i] = sg[i];
            for (int i = 0; i < NP; ++i) {
                if (i >= nd) break;
                const uint8_t a = i < ns ? sg[i] : 0;
                const uint8_t b = (i >= m && i < nsh) ? g.M(d, B[i - m]) : 0;
                sg[i] = g.S(a, b);
            }
            ns = nd;
            if (2 * L <= n) {
                const uint8_t iv = g.inv[d];
                for (int i = 0; i < NP; ++i) B[i] = i < nT ? g.M(tmp[i], iv) : 0;
                nB = nT; L = n + 1 - L; m = 1;
            } else m += 1;
        } else m += 1;
    }
    uint8_t Om[R];                                  // (S(x) sigma(x)) mod x^R
    for (int q = 0; q < R; ++q) {
        uint8_t acc = 0;
        for (int j = 0; j < ns; ++j) if (j <= q) acc = g.A(acc, g.M(S[q - j], sg[j]));
        Om[q] = acc;
    }
    // Horner over the trailing zero coefficients of the reference's vector contributes nothing: start at the true degree
    int deg = ns - 1; while (deg > 0 && sg[deg] == 0) --deg;
    int pos[T + 1], np = 0;
    for (int i = 0; i < 26; ++i) {
        const uint8_t x = g.exp[i == 0 ? 0 : 26 - i];            // alpha^{-i}
        uint8_t acc = sg[deg];
        for (int q = deg - 1; q >= 0; --q) acc = g.A(g.M(acc, x), sg[q]);
        if (acc == 0) { if (np < T + 1) pos[np] = i; ++np; }
    }
    if (np > T) return false;
    if (fixed && np != deg) return false;
    uint8_t dp[NP]; const int ndp = ns > 1 ? ns - 1 : 1;
    for (int i = 0; i < NP; ++i) dp[i] = 0;
    for (int i = 1; i < ns; ++i) { const int im = i % 3; dp[i - 1] = im == 0 ? 0 : (im == 1 ? sg[i] : g.A(sg[i], sg[i])); }
    int ddeg = ndp - 1; while (ddeg > 0 && dp[ddeg] == 0) --ddeg;
    for (int e = 0; e < np; ++e) {
        const uint8_t xi = g.exp[pos[e] == 0 ? 0 : 26 - pos[e]];
        uint8_t num = 0, den = dp[ddeg];
        for (int q = R - 1; q >= 0; --q) num = g.A(g.M(num, xi), Om[q]);
        for (int q = ddeg - 1; q >= 0; --q) den = g.A(g.M(den, xi), dp[q]);
        if (den == 0) return false;
        const uint8_t mag = g.M(g.neg[num], g.inv[den]);
        c[pos[e]] = fixed ? g.S(c[pos[e]], mag) : g.A(c[pos[e]], mag);
    }
    return true;
}

template <int R>
T3_HD bool rs_decode_block(const RsView& g, uint8_t* c, bool fixed) {
    uint8_t S[R];
    if (rs_syndromes<R>(g, c, S)) return true;
    return rs_correct<R>(g, c, S, fixed);
}

}  // namespace t3
