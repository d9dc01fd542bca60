// t3_decode.hip — decode-side gfx950 kernels (generic, framing-agnostic versions).
//
//   K3+K4 dec_gather_rs_kernel    one lane per RS block: gather 26 symbols through the framing's index map,
//                                 descramble, syndromes, (BM/Chien/Forney when non-zero), emit k data symbols
//                                 (demap_and_rsdecode_bands_from_words OLD:948-993 + descramble OLD:938-947)
//   K5'   dec_emit_kernel         [de-interleave OLD:781-813 ->] symbols -> 26-trit words / pixels (OLD:1022-1040, 735-747)
//         rs_decode_blocks_kernel block-level RSCodec::decode_block (OLD:546-662)
//         inject_errors_kernel    seeded trit-error injector for the recovery test
//         crc_chunks_kernel / frame_record_kernel   CRC-32 + frame index record (multi-GPU exchange payload)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/t3hip.h"
#include "t3_decode.h"

namespace t3 {

__device__ __forceinline__ uint32_t fdiv2(uint32_t n, const DevDiv& d) { return d.d <= 1 ? n : (__umulhi(n, d.mul) >> d.sh); }

__device__ __forceinline__ uint32_t scr_state(uint64_t i, uint32_t cyc24, uint32_t pre0, uint32_t pre1) {
    if (i < 2) return i == 0 ? pre0 : pre1;
    return (cyc24 >> (2u * (uint32_t)((i - 2) % 6))) & 3u;
}
__device__ __forceinline__ uint8_t sub_trits(uint32_t s, uint32_t st) {   // descramble_symbol OLD:88-94
    const uint32_t t0 = (3u + s % 3u - st) % 3u, t1 = (3u + (s / 3u) % 3u - st) % 3u, t2 = (3u + (s / 9u) % 3u - st) % 3u;
    return (uint8_t)(t0 + 3u * t1 + 9u * t2);
}

template <int R>
__device__ __noinline__ bool decode_one(const RsView& v, uint8_t* c, bool fixed) { return rs_decode_block<R>(v, c, fixed); }

__global__ __launch_bounds__(256) void dec_gather_rs_kernel(const DecArgs a) {
    __shared__ RsTables sT;
    for (int i = threadIdx.x; i < (int)sizeof(RsTables); i += blockDim.x) ((uint8_t*)&sT)[i] = ((const uint8_t*)a.tab)[i];
    __syncthreads();
    const RsView v{sT.mul, sT.add, sT.neg, sT.inv, sT.exp};
    for (uint64_t item = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; item < a.total_blocks; item += (uint64_t)gridDim.x * blockDim.x) {
        int b = 0;
#pragma unroll
        for (int q = 1; q < 9; ++q) if (item >= a.band_first[q]) b = q;
        const uint64_t m = item - a.band_first[b];
        const uint32_t k = a.band_k[b];
        uint8_t c[26];
        for (int i = 0; i < 26; ++i) {
            uint64_t pos, si;                                             // byte position after the header; scrambler index
            const uint64_t j = 26 * m + i;
            if (!a.fixed) {
                // reference decoder: band b = slot b of every body word, minus the beacon words of the beacon slot (OLD:953-961)
                uint64_t wi = j;
                if (a.beacon_on && (uint32_t)b == a.slot) wi = j + j / (a.period - 1u) + 1u;   // period > 1 here (else the band is empty)
                pos = 9 * wi + b; si = pos;
            } else {
                // v6c: band-serial body, beacon slots interleaved afterwards
                const uint64_t q = a.band_off[b] + j; si = q; pos = q;
                if (a.beacon_on) {
                    const uint64_t per = 9ull * a.period - 1u, t = q / per, rem = q - t * per;
                    if (rem < 8) pos = 9 * (t * a.period) + (rem < a.slot ? rem : rem + 1);
                    else { const uint64_t r2 = rem - 8; pos = 9 * (t * a.period + 1 + r2 / 9) + r2 % 9; }
                }
            }
            const uint32_t s = a.in[a.hdr_syms + pos];
            c[i] = sub_trits(s, scr_state(si, a.cyc24, a.pre0, a.pre1));
        }
        bool ok;
        switch (k) {
            case 24: ok = decode_one<2>(v, c, a.fixed != 0); break;
            case 22: ok = decode_one<4>(v, c, a.fixed != 0); break;
            case 20: ok = decode_one<6>(v, c, a.fixed != 0); break;
            default: ok = decode_one<8>(v, c, a.fixed != 0); break;
        }
        if (!ok) { atomicAdd(a.fail, 1u); continue; }
        if (!a.fixed) { uint8_t* o = a.use + a.band_off[b] + m * k; for (uint32_t p = 0; p < k; ++p) o[p] = c[p]; }
        else for (uint32_t p = 0; p < k; ++p) { const uint64_t s = 9 * (m * k + p) + b; if (s < a.n_sym) a.use[s] = c[p]; }
    }
}

__device__ __forceinline__ uint32_t il_perm_n(uint32_t u, uint32_t n, const EmitArgs& a) {
    const uint32_t chunk = fdiv2(u, a.div_A), base = chunk * a.il_A, rem = u - base;
    const uint32_t take = min(a.il_A, n - base);
    const uint32_t r = fdiv2(rem, a.div_w), c = rem - r * a.il_w;
    const uint32_t rowlen = min(a.il_w, take - r * a.il_w);
    return base + r * a.il_w + ((r & 1u) ? rowlen - 1u - c : c);
}

__global__ __launch_bounds__(256) void dec_emit_kernel(const EmitArgs a) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < a.n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t T0 = 26 * w, u0 = T0 / 3; const uint32_t o = (uint32_t)(T0 - 3 * u0);
        uint32_t tr[30];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const uint64_t u = u0 + i; uint32_t s = 0;
            if (u < a.use_syms) s = a.use[a.il_on ? il_perm_n((uint32_t)u, (uint32_t)a.use_syms, a) : u];
            tr[3 * i] = s % 3u; tr[3 * i + 1] = (s / 3u) % 3u; tr[3 * i + 2] = s / 9u;
        }
        uint32_t sy[9];
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            uint32_t t[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { const int q = 3 * s + j; t[j] = q < 26 ? tr[o + q] : 0u; }   // T[26] = 0 (OLD:1034)
            sy[s] = t[0] + 3u * t[1] + 9u * t[2];
        }
        if (!a.to_pixels) { uint8_t* p = (uint8_t*)a.out + 9 * w; for (int s = 0; s < 9; ++s) p[s] = (uint8_t)sy[s]; }
        else {
            uint32_t h[6];                                             // unpack_two_pixels OLD:706-722
            h[0] = sy[0] + 27u * (sy[1] % 9u); h[1] = (uint32_t)((int)(sy[1] / 9u + 3u * sy[2]) - 40) & 0xFFFFu;
            h[2] = (uint32_t)((int)(sy[3] + 27u * (sy[4] % 3u)) - 40) & 0xFFFFu;
            h[3] = sy[4] / 3u + 9u * sy[5]; h[4] = (uint32_t)((int)(sy[6] + 27u * (sy[7] % 3u)) - 40) & 0xFFFFu;
            h[5] = (uint32_t)((int)(sy[7] / 3u + 9u * (sy[8] % 9u)) - 40) & 0xFFFFu;
            uint32_t* p = (uint32_t*)((uint8_t*)a.out + 12 * w);       // 12-byte records: three aligned dwords per lane
            p[0] = h[0] | h[1] << 16; p[1] = h[2] | h[3] << 16; p[2] = h[4] | h[5] << 16;
        }
    }
}

__global__ __launch_bounds__(256) void rs_decode_blocks_kernel(uint8_t* code, uint64_t n_blocks, int k, int fixed, const RsTables* tab, uint8_t* data, uint8_t* okv) {
    __shared__ RsTables sT;
    for (int i = threadIdx.x; i < (int)sizeof(RsTables); i += blockDim.x) ((uint8_t*)&sT)[i] = ((const uint8_t*)tab)[i];
    __syncthreads();
    const RsView v{sT.mul, sT.add, sT.neg, sT.inv, sT.exp};
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= n_blocks) return;
    uint8_t c[26];
    for (int i = 0; i < 26; ++i) c[i] = code[blk * 26 + i] % 27u;
    bool ok;
    switch (k) {
        case 24: ok = decode_one<2>(v, c, fixed != 0); break;
        case 22: ok = decode_one<4>(v, c, fixed != 0); break;
        case 20: ok = decode_one<6>(v, c, fixed != 0); break;
        default: ok = decode_one<8>(v, c, fixed != 0); break;
    }
    for (int i = 0; i < 26; ++i) code[blk * 26 + i] = c[i];          // corrected (or partially modified) in place, like inout_n
    if (ok) for (int i = 0; i < k; ++i) data[blk * k + i] = c[i];    // out_k only on success (OLD:564,660)
    okv[blk] = ok ? 1 : 0;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void inject_errors_kernel(uint8_t* syms, uint64_t n_blocks, uint32_t seed, int max_err) {
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h = mix32(seed ^ mix32((uint32_t)b * 0x9e3779b9u + (uint32_t)(b >> 32)));
        const int e = (int)(h % (uint32_t)(max_err + 1)); uint32_t used = 0;
        for (int q = 0; q < e; ++q) {
            h = mix32(h + 0x632be5abu);
            uint32_t pos = h % 26u; while (used >> pos & 1u) pos = (pos + 1u) % 26u; used |= 1u << pos;
            const uint32_t trit = (h >> 8) % 3u, delta = 1u + ((h >> 12) & 1u);
            const uint32_t s = syms[26 * b + pos]; uint32_t t[3] = {s % 3u, (s / 3u) % 3u, (s / 9u) % 3u};
            t[trit] = (t[trit] + delta) % 3u;
            syms[26 * b + pos] = (uint8_t)(t[0] + 3u * t[1] + 9u * t[2]);
        }
    }
}

// ---- CRC-32 (poly 0xEDB88320, io_t3p_t3v.cpp:18-33) over the payload, in parallel -----------------------------
// Register update is GF(2)-linear: R(x, A||B) = Z_{|B|} R(x, A) ^ R(0, B).  Each lane takes a chunk, computes
// R(0, chunk), moves it to the end of the stream with the "append zero bytes" operators and XORs it in.
__device__ __forceinline__ uint32_t gf2_apply(const uint32_t* col, uint32_t x) {
    uint32_t y = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) y ^= (x >> i & 1u) ? col[i] : 0u;
    return y;
}

// append `n` zero bytes to register x: one operator per set bit of n
__device__ __forceinline__ uint32_t crc_shift(const uint32_t* zpow, uint32_t x, uint64_t n) {
    for (int j = 0; n; ++j, n >>= 1) if (n & 1u) x = gf2_apply(zpow + 32 * j, x);
    return x;
}

// One lane = one chunk, taken as two halves in lockstep (two independent register chains), a 32-bit word per step and
// chain by slicing-by-4: four independent table reads instead of four dependent ones.  The four 256-entry tables sit in
// LDS in four copies each (copy = lane & 3).  Measured alternatives (8K frame, 187 MB): one dependent byte-table chain per
// lane 147 us; this kernel 127 us; byte table in 32 per-bank copies with four chains 175 us; a coalesced row sweep with
// advance tables 193 us (profiles/r01/notes.md).
__device__ __forceinline__ uint32_t crc_word(const uint32_t* tb, uint32_t cp, uint32_t r, uint32_t w) {
    r ^= w;
    return tb[((3u * 256u + (r & 0xFFu)) << 2) + cp] ^ tb[((2u * 256u + ((r >> 8) & 0xFFu)) << 2) + cp] ^
           tb[((1u * 256u + ((r >> 16) & 0xFFu)) << 2) + cp] ^ tb[((r >> 24) << 2) + cp];
}
__global__ __launch_bounds__(256) void crc_chunks_kernel(const CrcArgs a) {
    __shared__ uint32_t tb[4 * 256 * 4]; __shared__ uint32_t zp[kCrcPows * 32];
    {   // T_0[e] = the byte table; T_{j+1}[e] = T_j[e] advanced by one more zero byte (eight bit steps)
        uint32_t c = threadIdx.x;
        for (int j = 0; j < 4; ++j) {
            for (int i = 0; i < 8; ++i) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            for (int cp = 0; cp < 4; ++cp) tb[((j * 256 + threadIdx.x) << 2) + cp] = c;
        }
    }
    for (int i = threadIdx.x; i < kCrcPows * 32; i += blockDim.x) zp[i] = a.zpow[i];
    __syncthreads();
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x, cp = threadIdx.x & 3u;
    uint32_t sum = 0, part = 0;
    if (ch < a.n_chunks) {
        const uint64_t beg = (uint64_t)ch * a.chunk_bytes, end = min(beg + a.chunk_bytes, a.n_bytes);
        const uint64_t mid = min(beg + (uint64_t)(a.chunk_bytes / 32u) * 16u, end);   // halves start 16-byte aligned
        uint32_t rA = 0, rB = 0;
        uint64_t i = beg, j = mid;
        if (((uintptr_t)a.data & 15u) == 0) {
            for (; i + 16 <= mid && j + 16 <= end; i += 16, j += 16) {
                const uint4 qa = *(const uint4*)(a.data + i), qb = *(const uint4*)(a.data + j);
                const uint32_t wa[4] = {qa.x, qa.y, qa.z, qa.w}, wb[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    sum = __builtin_amdgcn_sad_u8(wa[k], 0u, sum); sum = __builtin_amdgcn_sad_u8(wb[k], 0u, sum);
                    rA = crc_word(tb, cp, rA, wa[k]); rB = crc_word(tb, cp, rB, wb[k]);
                }
            }
        }
        for (; i < mid; ++i) { const uint32_t v = a.data[i]; sum += v; rA = tb[((rA ^ v) & 0xFFu) << 2] ^ (rA >> 8); }
        for (; j < end; ++j) { const uint32_t v = a.data[j]; sum += v; rB = tb[((rB ^ v) & 0xFFu) << 2] ^ (rB >> 8); }
        const uint32_t r = crc_shift(zp, rA, end - mid) ^ rB;
        part = crc_shift(zp, r, a.n_bytes - end);                    // move it to the end of the stream
    }
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_down(sum, o); part ^= __shfl_down(part, o); }
    if ((threadIdx.x & 63) == 0) { if (part) atomicXor(a.chunk_crc, part); if (sum) atomicAdd(a.sym_sum, sum); }
}

// header symbols of the frame against the ones the previous frame parsed to (speculative decode, t3_api_decode.cpp)
// One workgroup; writes both verdict words: [0] = header differs, [1] = 0 (the body kernels count uncorrectable blocks into it)
__global__ void hdr_compare_kernel(const uint8_t* in, const HdrExpect expect, uint32_t n, uint32_t* verdict) {
    __shared__ uint8_t ex[96];
    if (threadIdx.x < 96) ex[threadIdx.x] = expect.b[threadIdx.x < 96 ? threadIdx.x : 0];     // (kernel arguments are not indexed dynamically: staged)
    __syncthreads();
    const int diff = __syncthreads_or(threadIdx.x < n && in[threadIdx.x] != ex[threadIdx.x]);
    if (threadIdx.x == 0) { verdict[0] = diff ? 1u : 0u; verdict[1] = 0u; }
}

// One wave.  acc[0] / acc[1]: XOR of the chunk remainders moved to the end of the stream / symbol sum, as the CRC kernels left them;
// lead = the leading 0xFFFFFFFF carried through n_bytes zero bytes (host: square-and-multiply on the operator -- on the device that is a
// chain of dependent global loads, 9 us); tail: the last tail_len < 2048 bytes of the stream that no matrix-core round covered (null: none).
// The tail is taken right-aligned on a grid of 64 pieces of 32 bytes (leading zeros do not move a zero register), bit-serially per lane,
// and the pieces are joined by a butterfly of "append 32 * 2^l zero bytes" operators -- it sits at the end of the stream, so its
// remainder needs no further shift.
__global__ __launch_bounds__(64) void frame_record_kernel(const uint32_t* acc, uint32_t lead, const uint8_t* tail, uint32_t tail_len, const uint32_t* zpow,
                                                          const uint8_t* words, uint64_t n_words, uint64_t frame_idx, uint32_t profile, uint32_t mode, void* recv,
                                                          const uint32_t* partials, uint32_t n_partials) {   // n_partials != 0: the CRC kernel left one (xor, sum) per workgroup instead of acc[0..1]
    t3_frame_record* rec = (t3_frame_record*)recv;
    const uint32_t lane = threadIdx.x;
    uint32_t r = 0, sum = 0;
    __shared__ uint32_t zp[6 * 32];                                                       // the butterfly's six operators, fetched in one pass (level by
    if (tail && tail_len) {                                                               // level from global memory the kernel took 15 us)
        for (uint32_t i = lane; i < 6u * 32u; i += 64u) zp[i] = zpow[32u * 5u + i];
        __syncthreads();
    }
    if (tail && tail_len) {
        const int32_t lo = (int32_t)tail_len - 32 * (int32_t)(64u - lane);
        // the lane's 32 bytes first, all loads in flight together (one after the other, each behind the previous byte's eight register
        // steps, they were 32 memory latencies in a row: 6 of the kernel's 9 us); bytes in front of the tail read as zero
        uint32_t bytes[32];
#pragma unroll
        for (int32_t i = 0; i < 32; ++i) bytes[i] = lo + i >= 0 ? (uint32_t)tail[lo + i] : 0u;
#pragma unroll
        for (int32_t i = 0; i < 32; ++i) {
            const uint32_t v = bytes[i]; sum += v; r ^= v;                                // (leading zero bytes leave a zero register zero)
#pragma unroll
            for (int q = 0; q < 8; ++q) r = (r & 1u) ? (0xEDB88320u ^ (r >> 1)) : (r >> 1);
        }
#pragma unroll
        for (uint32_t l = 0; l < 6; ++l) {
            const uint32_t other = __shfl_xor(r, 1 << l), up = (lane >> l) & 1u;
            r = gf2_apply(zp + 32u * l, up ? other : r) ^ (up ? r : other);
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    }
    uint32_t ax = 0, as = 0;
    if (n_partials) {
        for (uint32_t i = lane; i < n_partials; i += 64u) { const uint2 v = *(const uint2*)(partials + 2u * i); ax ^= v.x; as += v.y; }
        for (int o = 32; o > 0; o >>= 1) { ax ^= __shfl_down(ax, o); as += __shfl_down(as, o); }
    } else if (lane == 0) { ax = acc[0]; as = acc[1]; }
    if (lane == 0) {
        rec->frame_idx = frame_idx; rec->n_words = n_words; rec->byte_offset = 0;
        rec->crc32 = (lead ^ ax ^ r) ^ 0xFFFFFFFFu; rec->sym_sum = as + sum;              // final inversion
        rec->profile = (uint8_t)profile; rec->mode = (uint8_t)mode;
        for (int i = 0; i < 8; ++i) rec->pad_[i] = 0;
    }
    if (lane < 54) rec->header_syms[lane] = lane < 9 * n_words ? words[lane] : 0;
}

}  // namespace t3
