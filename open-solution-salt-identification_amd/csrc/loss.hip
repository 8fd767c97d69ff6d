// loss.hip — Lovasz hinge and BCE+Dice on the GPU.
//
// Lovasz hinge (lovasz_losses.py:81-115,21-33 via models.py:326-328): the reference loops over the B
// images in Python and issues ~10 small torch kernels per image (sort, gather, cumsum x2, ...).  Here
// ONE launch does the whole batch: one 1024-thread workgroup per image runs a stable LSD radix sort
// (4 x 8-bit digits) of the hinge errors with a (flat index, label) payload through a ping-pong
// workspace that stays L2-resident (P = 2*H*W = 32768 keys = 256 KB per image), then a single fused
// pass does the label scan, the Jaccard gradient g_k, the dot with elu(errors) and the scatter of
// d loss / d logit back to NCHW order.  Ties keep flat-index order (stable sort), so results are
// deterministic; the loss value itself is tie-order invariant.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int LT = 1024;                 // threads per image
constexpr int LW = LT / 64;              // waves

__device__ __forceinline__ unsigned desc_key(float e) {          // ascending uint order == descending float order
    unsigned u = __float_as_uint(e);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~u;
}
__device__ __forceinline__ float key_to_float(unsigned k) {
    unsigned u = ~k;
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

__global__ __launch_bounds__(LT) void lovasz_kernel(salt_lovasz_args a) {
    __shared__ unsigned hist[256];
    __shared__ unsigned base[256];
    __shared__ unsigned wcount[LW][256];
    __shared__ float red[LW];
    __shared__ unsigned scan_w[LW];
    __shared__ unsigned carry_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P;
    const float* z = a.logits + (int64_t)b * P;
    const float* y = a.target + (int64_t)b * P;
    unsigned* k0 = a.ws_keys + (int64_t)b * P;
    unsigned* v0 = a.ws_vals + (int64_t)b * P;
    unsigned* k1 = a.ws_keys + ((int64_t)a.B + b) * P;
    unsigned* v1 = a.ws_vals + ((int64_t)a.B + b) * P;

    // ---- keys + total positives
    float gsum = 0.f;
    for (int i = tid; i < P; i += LT) {
        const float lab = y[i] > 0.5f ? 1.f : 0.f;              // target.long() of a {0.,1.} mask
        const float e = 1.f - z[i] * (2.f * lab - 1.f);
        k0[i] = desc_key(e);
        v0[i] = ((unsigned)i << 1) | (lab > 0.5f ? 1u : 0u);
        gsum += lab;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gsum += __shfl_xor(gsum, o);
    if (lane == 0) red[wave] = gsum;
    __syncthreads();
    float G = 0.f;
    for (int w = 0; w < LW; ++w) G += red[w];
    __syncthreads();

    // ---- stable LSD radix sort, 8 bits per pass
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int pass = 0; pass < 4; ++pass) {
        const unsigned* ki = (pass & 1) ? k1 : k0; const unsigned* vi = (pass & 1) ? v1 : v0;
        unsigned* ko = (pass & 1) ? k0 : k1; unsigned* vo = (pass & 1) ? v0 : v1;
        const int shift = pass * 8;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < P; i += LT) atomicAdd(&hist[(ki[i] >> shift) & 255u], 1u);
        __syncthreads();
        if (tid < 64) {                                           // exclusive scan of 256 counters by one wave
            unsigned c[4], s = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { c[j] = hist[tid * 4 + j]; s += c[j]; }
            unsigned incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            unsigned run = incl - s;
#pragma unroll
            for (int j = 0; j < 4; ++j) { base[tid * 4 + j] = run; run += c[j]; }
        }
        __syncthreads();
        for (int c0 = 0; c0 < P; c0 += LT) {
            for (int i = tid; i < LW * 256; i += LT) (&wcount[0][0])[i] = 0;
            __syncthreads();
            const int i = c0 + tid;
            const bool ok = i < P;
            unsigned key = 0, val = 0, d = 256;
            if (ok) { key = ki[i]; val = vi[i]; d = (key >> shift) & 255u; }
            // lanes of this wave holding the same digit
            unsigned long long m = __ballot(ok);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const unsigned long long bm = __ballot((d >> bit) & 1u);
                m &= ((d >> bit) & 1u) ? bm : ~bm;
            }
            const unsigned rank = (unsigned)__popcll(m & lt_mask);
            if (ok && rank == 0) wcount[wave][d] = (unsigned)__popcll(m);
            __syncthreads();
            if (tid < 256) {                                      // digit tid: prefix over waves, advance base
                unsigned run = base[tid];
                for (int w = 0; w < LW; ++w) { const unsigned c = wcount[w][tid]; wcount[w][tid] = run; run += c; }
                base[tid] = run;
            }
            __syncthreads();
            if (ok) { const unsigned dst = wcount[wave][d] + rank; ko[dst] = key; vo[dst] = val; }
            __syncthreads();
        }
    }
    // after 4 passes the sorted sequence is back in (k0, v0)

    // ---- fused scan + Jaccard gradient + dot + scatter
    if (tid == 0) carry_s = 0;
    __syncthreads();
    float lsum = 0.f;
    const float gscale = a.loss_scale / (float)a.B;
    float* dz = a.dlogits ? a.dlogits + (int64_t)b * P : nullptr;
    for (int c0 = 0; c0 < P; c0 += LT) {
        const int i = c0 + tid;
        const bool ok = i < P;
        unsigned val = ok ? v0[i] : 0u;
        const unsigned lab = val & 1u;
        unsigned incl = ok ? lab : 0u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) scan_w[wave] = incl;
        __syncthreads();
        unsigned woff = carry_s;
        for (int w = 0; w < wave; ++w) woff += scan_w[w];
        const unsigned c_k = woff + incl;                         // inclusive count of positives up to rank k
        __syncthreads();
        if (tid == LT - 1) carry_s = c_k;
        if (ok) {
            const float e = key_to_float(k0[i]);
            const float kf = (float)(i + 1), ck = (float)c_k, ckm = (float)(c_k - lab);
            // exactly the reference's fp32 sequence (lovasz_losses.py:27-32): one correctly rounded division, one
            // subtraction from 1, one first difference; no fma contraction (the difference cancels ~3 digits)
            const float jk = __fsub_rn(1.f, __fdiv_rn(G - ck, G + (kf - ck)));
            float jm = 0.f;
            if (i > 0) jm = __fsub_rn(1.f, __fdiv_rn(G - ckm, G + ((kf - 1.f) - ckm)));
            const float g = (i > 0) ? __fsub_rn(jk, jm) : jk;
            const float el = e > 0.f ? e : expm1f(e);
            lsum += el * g;
            if (dz) {
                const float d = e > 0.f ? 1.f : __expf(e);
                const float s = lab ? 1.f : -1.f;
                dz[val >> 1] = -s * d * g * gscale;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < LW; ++w) t += red[w];
        if (P == 0) t = 0.f;
        a.loss_per_image[b] = t;
    }
}

// Prefetching variant (the one that is launched): same chunk order as lovasz_kernel (chunk c = elements c*1024 + tid, so the
// result is identical), but
//   * the (key, payload) pairs of the next group of 4 chunks are loaded while the current group is ranked and scattered (the
//     ping-pong workspace is L2-resident; the per-chunk critical path becomes LDS-only), and
//   * the digit histogram of pass p+1 is accumulated while pass p scatters (a histogram does not depend on element order),
//     which removes the four separate counting sweeps.
constexpr int LG = 4;                    // chunks per prefetch group
__global__ __launch_bounds__(LT) void lovasz_pf_kernel(salt_lovasz_args a) {
    __shared__ unsigned hist[2][256];
    __shared__ unsigned base[256];
    __shared__ unsigned wcount[LW][256];
    __shared__ float red[LW];
    __shared__ unsigned scan_w[LW];
    __shared__ unsigned carry_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P;
    const float* z = a.logits + (int64_t)b * P;
    const float* y = a.target + (int64_t)b * P;
    unsigned* k0 = a.ws_keys + (int64_t)b * P;
    unsigned* v0 = a.ws_vals + (int64_t)b * P;
    unsigned* k1 = a.ws_keys + ((int64_t)a.B + b) * P;
    unsigned* v1 = a.ws_vals + ((int64_t)a.B + b) * P;
    const int nchunks = (P + LT - 1) / LT, ngroups = (nchunks + LG - 1) / LG;

    // ---- keys + total positives + histogram of the first digit
    if (tid < 256) { hist[0][tid] = 0; hist[1][tid] = 0; }
    __syncthreads();
    float gsum = 0.f;
    for (int i = tid; i < P; i += LT) {
        const float lab = y[i] > 0.5f ? 1.f : 0.f;              // target.long() of a {0.,1.} mask
        const float e = 1.f - z[i] * (2.f * lab - 1.f);
        const unsigned key = desc_key(e);
        k0[i] = key;
        v0[i] = ((unsigned)i << 1) | (lab > 0.5f ? 1u : 0u);
        atomicAdd(&hist[0][key & 255u], 1u);
        gsum += lab;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gsum += __shfl_xor(gsum, o);
    if (lane == 0) red[wave] = gsum;
    __syncthreads();
    float G = 0.f;
    for (int w = 0; w < LW; ++w) G += red[w];
    __syncthreads();

    // ---- stable LSD radix sort, 8 bits per pass
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int pass = 0; pass < 4; ++pass) {
        const unsigned* ki = (pass & 1) ? k1 : k0; const unsigned* vi = (pass & 1) ? v1 : v0;
        unsigned* ko = (pass & 1) ? k0 : k1; unsigned* vo = (pass & 1) ? v0 : v1;
        const int shift = pass * 8;
        unsigned* hcur = hist[pass & 1];
        unsigned* hnext = hist[(pass + 1) & 1];
        unsigned ck[LG], cv[LG], nk[LG], nv[LG];
#pragma unroll
        for (int u = 0; u < LG; ++u) { const int i = u * LT + tid; ck[u] = i < P ? ki[i] : 0u; cv[u] = i < P ? vi[i] : 0u; }
        if (tid < 64) {                                           // exclusive scan of 256 counters by one wave
            unsigned c[4], s = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { c[j] = hcur[tid * 4 + j]; s += c[j]; }
            unsigned incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            unsigned run = incl - s;
#pragma unroll
            for (int j = 0; j < 4; ++j) { base[tid * 4 + j] = run; run += c[j]; }
        }
        __syncthreads();
        if (tid < 256) hcur[tid] = 0;                             // becomes the histogram of pass+2 (written from pass+1 on)
        for (int g = 0; g < ngroups; ++g) {
            const int cn = (g + 1) * LG;
#pragma unroll
            for (int u = 0; u < LG; ++u) { const int i = (cn + u) * LT + tid; nk[u] = i < P ? ki[i] : 0u; nv[u] = i < P ? vi[i] : 0u; }
#pragma unroll
            for (int u = 0; u < LG; ++u) {
                const int c0 = (g * LG + u) * LT;
                if (c0 >= P) break;                               // uniform
                for (int i = tid; i < LW * 256; i += LT) (&wcount[0][0])[i] = 0;
                __syncthreads();
                const bool ok = c0 + tid < P;
                const unsigned key = ck[u];
                const unsigned d = ok ? ((key >> shift) & 255u) : 256u;
                unsigned long long m = __ballot(ok);              // lanes of this wave holding the same digit
#pragma unroll
                for (int bit = 0; bit < 8; ++bit) {
                    const unsigned long long bm = __ballot((d >> bit) & 1u);
                    m &= ((d >> bit) & 1u) ? bm : ~bm;
                }
                const unsigned rank = (unsigned)__popcll(m & lt_mask);
                if (ok && rank == 0) wcount[wave][d] = (unsigned)__popcll(m);
                __syncthreads();
                if (tid < 256) {                                  // digit tid: prefix over waves, advance base
                    unsigned run = base[tid];
                    for (int w = 0; w < LW; ++w) { const unsigned c = wcount[w][tid]; wcount[w][tid] = run; run += c; }
                    base[tid] = run;
                }
                __syncthreads();
                if (ok) {
                    const unsigned dst = wcount[wave][d] + rank;
                    ko[dst] = key; vo[dst] = cv[u];
                    if (pass < 3) atomicAdd(&hnext[(key >> (shift + 8)) & 255u], 1u);
                }
                __syncthreads();
            }
#pragma unroll
            for (int u = 0; u < LG; ++u) { ck[u] = nk[u]; cv[u] = nv[u]; }
        }
        __syncthreads();
    }
    // after 4 passes the sorted sequence is back in (k0, v0)

    // ---- fused scan + Jaccard gradient + dot + scatter
    if (tid == 0) carry_s = 0;
    __syncthreads();
    float lsum = 0.f;
    const float gscale = a.loss_scale / (float)a.B;
    float* dz = a.dlogits ? a.dlogits + (int64_t)b * P : nullptr;
    unsigned ck[LG], cv[LG], nk[LG], nv[LG];
#pragma unroll
    for (int u = 0; u < LG; ++u) { const int i = u * LT + tid; ck[u] = i < P ? k0[i] : 0u; cv[u] = i < P ? v0[i] : 0u; }
    for (int g = 0; g < ngroups; ++g) {
        const int cn = (g + 1) * LG;
#pragma unroll
        for (int u = 0; u < LG; ++u) { const int i = (cn + u) * LT + tid; nk[u] = i < P ? k0[i] : 0u; nv[u] = i < P ? v0[i] : 0u; }
#pragma unroll
        for (int u = 0; u < LG; ++u) {
            const int c0 = (g * LG + u) * LT;
            if (c0 >= P) break;
            const int i = c0 + tid;
            const bool ok = i < P;
            const unsigned val = ok ? cv[u] : 0u;
            const unsigned lab = val & 1u;
            unsigned incl = ok ? lab : 0u;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            if (lane == 63) scan_w[wave] = incl;
            __syncthreads();
            unsigned woff = carry_s;
            for (int w = 0; w < wave; ++w) woff += scan_w[w];
            const unsigned c_k = woff + incl;                     // inclusive count of positives up to rank k
            __syncthreads();
            if (tid == LT - 1) carry_s = c_k;
            if (ok) {
                const float e = key_to_float(ck[u]);
                const float kf = (float)(i + 1), ckf = (float)c_k, ckm = (float)(c_k - lab);
                // exactly the reference's fp32 sequence (lovasz_losses.py:27-32): one correctly rounded division, one
                // subtraction from 1, one first difference; no fma contraction (the difference cancels ~3 digits)
                const float jk = __fsub_rn(1.f, __fdiv_rn(G - ckf, G + (kf - ckf)));
                float jm = 0.f;
                if (i > 0) jm = __fsub_rn(1.f, __fdiv_rn(G - ckm, G + ((kf - 1.f) - ckm)));
                const float gk = (i > 0) ? __fsub_rn(jk, jm) : jk;
                const float el = e > 0.f ? e : expm1f(e);
                lsum += el * gk;
                if (dz) {
                    const float d = e > 0.f ? 1.f : __expf(e);
                    const float sg = lab ? 1.f : -1.f;
                    dz[val >> 1] = -sg * d * gk * gscale;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < LG; ++u) { ck[u] = nk[u]; cv[u] = nv[u]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < LW; ++w) t += red[w];
        if (P == 0) t = 0.f;
        a.loss_per_image[b] = t;
    }
}

// ---------------------------------------------------------------- Lovasz hinge, split over several workgroups per image
// The one-workgroup-per-image kernel above is a chain of ~500 barriers on 32 of the 256 CUs (265 us for B = 32).  The split form
// gives every image S segments of SL = 2048 consecutive positions, one 256-thread workgroup each, and runs the SAME stable LSD
// radix sort as 1 + 4 launches - the launch boundary is the only cross-workgroup synchronisation:
//   keys    : keys / payload of the segment, histogram of digit 0 per segment              -> H0[b][seg][256]
//   pass p  : digit base of the segment = (keys of all segments with a smaller digit) + (same digit in earlier segments), from
//             H_p; stable ranking chunk by chunk exactly as above (4 waves instead of 16); scatter; the histogram of digit p+1 is
//             collected per DESTINATION segment in LDS and flushed with integer atomics into H_(p+1) (three rotating buffers: the
//             one read in the previous pass is cleared for the next); the last pass counts the positives per destination segment
//   scan    : label scan with the carry of the earlier segments, Jaccard gradient, dot, scatter; per-segment loss partials that
//             lovasz_mean_split_kernel adds in fixed order.
// Integer atomics only: results do not depend on scheduling.  Positions, ties and the fp32 operation sequence of g_k are those of
// lovasz_kernel, so gradients are bit-identical to it; the loss differs in the last bits (different summation tree).
constexpr int ST = 256, SWV = ST / 64, SL = 2048, SCH = SL / ST, SMAXSEG = 64;
struct LovaszSplit { int S; unsigned* hist; unsigned* pos; float* part; };      // hist [3][B][S][256], pos [B][S], part [B][S]

__global__ __launch_bounds__(ST) void lovasz_keys_kernel(salt_lovasz_args a, LovaszSplit sp) {
    __shared__ unsigned h[256];
    const int b = blockIdx.x / sp.S, seg = blockIdx.x % sp.S, tid = threadIdx.x;
    const int P = a.P, i0 = seg * SL, i1 = min(i0 + SL, P);
    const float* z = a.logits + (int64_t)b * P;
    const float* y = a.target + (int64_t)b * P;
    unsigned* k0 = a.ws_keys + (int64_t)b * P;
    unsigned* v0 = a.ws_vals + (int64_t)b * P;
    h[tid] = 0;
    __syncthreads();
    for (int i = i0 + tid; i < i1; i += ST) {
        const float lab = y[i] > 0.5f ? 1.f : 0.f;
        const float e = 1.f - z[i] * (2.f * lab - 1.f);
        const unsigned key = desc_key(e);
        k0[i] = key;
        v0[i] = ((unsigned)i << 1) | (lab > 0.5f ? 1u : 0u);
        atomicAdd(&h[key & 255u], 1u);
    }
    __syncthreads();
    const int64_t slot = ((int64_t)b * sp.S + seg) * 256 + tid, hb = (int64_t)a.B * sp.S * 256;
    sp.hist[slot] = h[tid];
    sp.hist[hb + slot] = 0;                                       // buffer 1 collects digit 1 during pass 0
    if (tid == 0) sp.pos[b * sp.S + seg] = 0;
}

__global__ __launch_bounds__(ST) void lovasz_pass_kernel(salt_lovasz_args a, LovaszSplit sp, int pass) {
    extern __shared__ unsigned nh[];                              // [S][256] next digit per destination segment (+ [S] positives)
    __shared__ unsigned base[256];
    __shared__ unsigned wcount2[2][SWV][256];                     // double buffered: no barrier between a chunk's scatter and the next chunk's reset
    __shared__ unsigned scan_w[SWV];
    const int b = blockIdx.x / sp.S, seg = blockIdx.x % sp.S, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P, S = sp.S, i0 = seg * SL;
    const unsigned* ki = a.ws_keys + ((int64_t)((pass & 1) ? a.B : 0) + b) * P;
    const unsigned* vi = a.ws_vals + ((int64_t)((pass & 1) ? a.B : 0) + b) * P;
    unsigned* ko = a.ws_keys + ((int64_t)((pass & 1) ? 0 : a.B) + b) * P;
    unsigned* vo = a.ws_vals + ((int64_t)((pass & 1) ? 0 : a.B) + b) * P;
    const int64_t hb = (int64_t)a.B * S * 256;
    const unsigned* Hc = sp.hist + (pass % 3) * hb + (int64_t)b * S * 256;
    unsigned* Hn = sp.hist + ((pass + 1) % 3) * hb + (int64_t)b * S * 256;
    unsigned* Hz = sp.hist + ((pass + 2) % 3) * hb + (int64_t)b * S * 256;
    const int shift = pass * 8;
    // the segment's keys: all SCH chunks in registers before the first barrier
    unsigned ck[SCH], cv[SCH];
#pragma unroll
    for (int u = 0; u < SCH; ++u) { const int i = i0 + u * ST + tid; ck[u] = i < P ? ki[i] : 0u; cv[u] = i < P ? vi[i] : 0u; }
    // ---- digit bases of this segment
    {
        unsigned tot = 0, before = 0;
        for (int s2 = 0; s2 < S; ++s2) { const unsigned c = Hc[s2 * 256 + tid]; tot += c; if (s2 < seg) before += c; }
        unsigned incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) scan_w[wave] = incl;
        for (int e = tid; e < S * 256 + S; e += ST) nh[e] = 0;
        __syncthreads();
        unsigned woff = 0;
        for (int w = 0; w < wave; ++w) woff += scan_w[w];
        base[tid] = woff + incl - tot + before;
        Hz[seg * 256 + tid] = 0;                                  // read in the previous pass, collects digit pass+2 in the next
    }
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int u = 0; u < SCH; ++u) {
        const int c0 = i0 + u * ST;
        if (c0 >= P) break;                                       // uniform
        unsigned (*wcount)[256] = wcount2[u & 1];
#pragma unroll
        for (int w = 0; w < SWV; ++w) wcount[w][tid] = 0;
        __syncthreads();                                          // also publishes base[] / nh[] on the first round
        const bool ok = c0 + tid < P;
        const unsigned key = ck[u];
        const unsigned d = ok ? ((key >> shift) & 255u) : 256u;
        unsigned long long m = __ballot(ok);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long bm = __ballot((d >> bit) & 1u);
            m &= ((d >> bit) & 1u) ? bm : ~bm;
        }
        const unsigned rank = (unsigned)__popcll(m & lt_mask);
        if (ok && rank == 0) wcount[wave][d] = (unsigned)__popcll(m);
        __syncthreads();
        {
            unsigned run = base[tid];
#pragma unroll
            for (int w = 0; w < SWV; ++w) { const unsigned c = wcount[w][tid]; wcount[w][tid] = run; run += c; }
            base[tid] = run;
        }
        __syncthreads();
        if (ok) {
            const unsigned dst = wcount[wave][d] + rank;
            ko[dst] = key; vo[dst] = cv[u];
            const unsigned ds = dst / SL;
            if (pass < 3) atomicAdd(&nh[ds * 256 + ((key >> (shift + 8)) & 255u)], 1u);
            else if (cv[u] & 1u) atomicAdd(&nh[S * 256 + ds], 1u);
        }
    }
    __syncthreads();
    if (pass < 3) {
        for (int e = tid; e < S * 256; e += ST) { const unsigned c = nh[e]; if (c) atomicAdd(&Hn[e], c); }
    } else if (tid < S) {
        const unsigned c = nh[S * 256 + tid];
        if (c) atomicAdd(&sp.pos[b * S + tid], c);
    }
}

__global__ __launch_bounds__(ST) void lovasz_scan_kernel(salt_lovasz_args a, LovaszSplit sp) {
    __shared__ unsigned scan_w[SWV];
    __shared__ float red[SWV];
    const int b = blockIdx.x / sp.S, seg = blockIdx.x % sp.S, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P, S = sp.S, i0 = seg * SL;
    const unsigned* k0 = a.ws_keys + (int64_t)b * P;              // after 4 passes the sorted sequence is back in buffer 0
    const unsigned* v0 = a.ws_vals + (int64_t)b * P;
    unsigned ck[SCH], cv[SCH];
#pragma unroll
    for (int u = 0; u < SCH; ++u) { const int i = i0 + u * ST + tid; ck[u] = i < P ? k0[i] : 0u; cv[u] = i < P ? v0[i] : 0u; }
    unsigned gtot = 0, carry = 0;
    for (int s2 = 0; s2 < S; ++s2) { const unsigned c = sp.pos[b * S + s2]; gtot += c; if (s2 < seg) carry += c; }
    const float G = (float)gtot;
    const float gscale = a.loss_scale / (float)a.B;
    float* dz = a.dlogits ? a.dlogits + (int64_t)b * P : nullptr;
    float lsum = 0.f;
#pragma unroll
    for (int u = 0; u < SCH; ++u) {
        const int i = i0 + u * ST + tid;
        if (i0 + u * ST >= P) break;
        const bool ok = i < P;
        const unsigned val = ok ? cv[u] : 0u;
        const unsigned lab = val & 1u;
        unsigned incl = ok ? lab : 0u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        __syncthreads();                                          // scan_w of the previous chunk has been read
        if (lane == 63) scan_w[wave] = incl;
        __syncthreads();
        unsigned woff = carry, ctot = 0;
        for (int w = 0; w < SWV; ++w) { if (w < wave) woff += scan_w[w]; ctot += scan_w[w]; }
        const unsigned c_k = woff + incl;
        carry += ctot;
        if (ok) {
            const float e = key_to_float(ck[u]);
            const float kf = (float)(i + 1), ckf = (float)c_k, ckm = (float)(c_k - lab);
            // the reference's fp32 sequence (lovasz_losses.py:27-32), as in lovasz_kernel
            const float jk = __fsub_rn(1.f, __fdiv_rn(G - ckf, G + (kf - ckf)));
            float jm = 0.f;
            if (i > 0) jm = __fsub_rn(1.f, __fdiv_rn(G - ckm, G + ((kf - 1.f) - ckm)));
            const float gk = (i > 0) ? __fsub_rn(jk, jm) : jk;
            const float el = e > 0.f ? e : expm1f(e);
            lsum += el * gk;
            if (dz) {
                const float d = e > 0.f ? 1.f : __expf(e);
                const float sg = lab ? 1.f : -1.f;
                dz[val >> 1] = -sg * d * gk * gscale;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    __syncthreads();
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < SWV; ++w) t += red[w];
        sp.part[b * S + seg] = t;
    }
}

__global__ void lovasz_mean_split_kernel(const float* part, int B, int S, float scale, float* per_image, float* out) {
    __shared__ float sm[1024];
    const int tid = threadIdx.x;
    for (int b = tid; b < B; b += blockDim.x) {
        float t = 0.f;
        for (int s2 = 0; s2 < S; ++s2) t += part[b * S + s2];
        per_image[b] = t;
        if (b < 1024) sm[b] = t;
    }
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int b = 0; b < B; ++b) t += (b < 1024) ? sm[b] : per_image[b];
        out[0] = t * scale / (float)B;
    }
}

__global__ void mean_kernel(const float* v, int n, float scale, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += v[i];
        out[0] = s * scale / (float)n;
    }
}

// ---------------------------------------------------------------- BCE + Dice
__global__ __launch_bounds__(256) void bce_dice_partial_kernel(const float* z, const float* t, int HW, int ppp, int per, float* partials) {
    __shared__ float sm4[4];
    const int plane = blockIdx.x / ppp, part = blockIdx.x % ppp;
    const int i0 = part * per, i1 = min(i0 + per, HW);
    float s_pt = 0.f, s_p = 0.f, s_t = 0.f, s_b = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float zz = z[(int64_t)plane * HW + i], tt = t[(int64_t)plane * HW + i];
        const float p = 1.f / (1.f + expf(-zz));
        s_pt += p * tt; s_p += p; s_t += tt;
        s_b += fmaxf(zz, 0.f) - zz * tt + log1pf(expf(-fabsf(zz)));
    }
    const float r0 = block_sum_256(s_pt, sm4), r1 = block_sum_256(s_p, sm4), r2 = block_sum_256(s_t, sm4), r3 = block_sum_256(s_b, sm4);
    if (threadIdx.x == 0) { float* o = partials + (int64_t)blockIdx.x * 4; o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; }
}

__global__ void bce_dice_finalize_kernel(const float* partials, int B, int C, int ppp, int HW, float dw, float bw, float scale, float* sums, float* loss) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double bce = 0.0, dice = 0.0;
    for (int c = 0; c < C; ++c) {
        double pt = 0.0, p = 0.0, t = 0.0;
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < ppp; ++k) {
                const float* o = partials + ((int64_t)(b * C + c) * ppp + k) * 4;
                pt += o[0]; p += o[1]; t += o[2]; bce += o[3];
            }
        sums[c * 3 + 0] = (float)pt; sums[c * 3 + 1] = (float)p; sums[c * 3 + 2] = (float)t;
        dice += 1.0 - 2.0 * pt / (p + t + 1e-7);
    }
    sums[3 * C] = (float)bce;
    loss[0] = (float)((dw * dice / C + bw * bce / ((double)B * C * HW)) * scale);
}

__global__ void bce_dice_grad_kernel(const float* z, const float* t, int B, int C, int HW, const float* sums, float dw, float bw, float scale, float* dz) {
    const int64_t n = (int64_t)B * C * HW;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        const int c = (int)((i / HW) % C);
        const float pt = sums[c * 3], S = sums[c * 3 + 1] + sums[c * 3 + 2] + 1e-7f;
        const float zz = z[i], tt = t[i];
        const float p = 1.f / (1.f + expf(-zz));
        const float dD = -2.f * (tt * S - pt) / (S * S);
        dz[i] = scale * (dw / (float)C * dD * p * (1.f - p) + bw / (float)n * (p - tt));
    }
}

int bce_ppp(int HW, int* per) {
    int ppp = cdiv(HW, 4096);
    if (ppp > 16) ppp = 16;
    if (ppp < 1) ppp = 1;
    const int pp = cdiv(HW, ppp);
    if (per) *per = pp;
    return cdiv(HW, pp);
}

// ---------------------------------------------------------------- Adam (+L2), flat buffer
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                            const float* __restrict__ hyper) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2 = hyper[6], gs = hyper[7];
    const float step_size = lr / bc1, rs = 1.f / sqrtf(bc2);
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += gridDim.x * 256LL) {
        f32x4 pp = reinterpret_cast<const f32x4*>(p)[i], mm = reinterpret_cast<const f32x4*>(m)[i], vv = reinterpret_cast<const f32x4*>(v)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
        adam4(pp, gg, mm, vv, b1, b2, eps, wd, gs, step_size, rs);             // (common.h: shared with adam_pack_kernel - same bits)
        reinterpret_cast<f32x4*>(p)[i] = pp; reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    for (int64_t i = (n4 << 2) + blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        const float gr = g[i] * gs + wd * p[i];
        const float m1 = b1 * m[i] + (1.f - b1) * gr, v1 = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = m1; v[i] = v1;
        p[i] -= step_size * m1 / (sqrtf(v1) * rs + eps);
    }
}

__global__ void adam_tick_kernel(float* hyper, int64_t* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int64_t t = step[0] + 1;
        step[0] = t;
        hyper[5] = (float)(1.0 - pow((double)hyper[1], (double)t));
        hyper[6] = (float)(1.0 - pow((double)hyper[2], (double)t));
    }
}

// ---------------------------------------------------------------- TTA: sigmoid -> inverse flip -> mean
struct TtaKP { const float* logits; float* prob; int V, B, C, H, W; int ud[16], lr[16], rq[16]; int method; };
__global__ void tta_mean_kernel(TtaKP p) {
    const int64_t hw = (int64_t)p.H * p.W, n = (int64_t)p.B * p.C * hw;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        const int x = (int)(i % p.W); int64_t r = i / p.W; const int y = (int)(r % p.H); const int64_t plane = r / p.H;
        float s = p.method == 1 ? -1.f : (p.method == 2 ? 2.f : 0.f);
        for (int v = 0; v < p.V; ++v) {
            // out = flipud?(fliplr?(rot90(pred, -k))): undo the flips on the output coordinate, then read through the rotation
            // (np.rot90(m, q)[i, j] for q = 1, 2, 3: m[j, n-1-i], m[n-1-i, n-1-j], m[n-1-j, i]; rq = (-k) mod 4, square maps)
            const int yf = p.ud[v] ? p.H - 1 - y : y, xf = p.lr[v] ? p.W - 1 - x : x;
            int yy = yf, xx = xf;
            const int q = p.rq[v];
            if (q == 1) { yy = xf; xx = p.W - 1 - yf; }
            else if (q == 2) { yy = p.H - 1 - yf; xx = p.W - 1 - xf; }
            else if (q == 3) { yy = p.H - 1 - xf; xx = yf; }
            const float zz = p.logits[((int64_t)v * p.B * p.C + plane) * hw + (int64_t)yy * p.W + xx];
            const float pr = 1.f / (1.f + expf(-zz));
            if (p.method == 0) s += pr;
            else if (p.method == 1) s = fmaxf(s, pr);
            else if (p.method == 2) s = fminf(s, pr);
            else s += logf(pr);
        }
        p.prob[i] = p.method == 0 ? s / (float)p.V : (p.method == 3 ? expf(s / (float)p.V) : s);
    }
}
__global__ void flip_kernel(salt_flip_args a) {
    const int64_t hw = (int64_t)a.H * a.W, n = (int64_t)a.B * a.C * hw;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        const int x = (int)(i % a.W); int64_t r = i / a.W; const int y = (int)(r % a.H); const int64_t plane = r / a.H;
        const int yy = a.flip_ud ? a.H - 1 - y : y, xx = a.flip_lr ? a.W - 1 - x : x;
        a.y[i] = a.x[plane * hw + (int64_t)yy * a.W + xx];
    }
}

// ---------------------------------------------------------------- inference epilogue: crop + threshold, metric counts
__global__ void crop_threshold_kernel(salt_crop_threshold_args a) {
    const int64_t n = (int64_t)a.B * a.h * a.w;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        const int x = (int)(i % a.w); int64_t r = i / a.w; const int y = (int)(r % a.h); const int b = (int)(r / a.h);
        const float p = a.prob[(((int64_t)b * a.C + a.cls) * a.H + a.top + y) * a.W + a.left + x];
        a.mask[i] = p > a.threshold ? 1 : 0;
    }
}

struct IouKP { salt_iou_sweep_args a; double th[SALT_MAX_THRESHOLDS]; };
__global__ __launch_bounds__(256) void iou_sweep_kernel(IouKP k) {
    __shared__ int s_inter[SALT_MAX_THRESHOLDS], s_pred[SALT_MAX_THRESHOLDS], s_gt;
    const salt_iou_sweep_args& a = k.a;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < SALT_MAX_THRESHOLDS) { s_inter[tid] = 0; s_pred[tid] = 0; }
    if (tid == 0) s_gt = 0;
    __syncthreads();
    int ci[SALT_MAX_THRESHOLDS], cp[SALT_MAX_THRESHOLDS], cg = 0;
#pragma unroll
    for (int t = 0; t < SALT_MAX_THRESHOLDS; ++t) { ci[t] = 0; cp[t] = 0; }
    const int n = a.h * a.w;
    for (int i = tid; i < n; i += 256) {
        const int y = i / a.w, x = i - y * a.w;
        const double p = (double)a.prob[(((int64_t)b * a.C + a.cls) * a.H + a.top + y) * a.W + a.left + x];
        const int g = a.gt[(int64_t)b * n + i] ? 1 : 0;
        cg += g;
#pragma unroll
        for (int t = 0; t < SALT_MAX_THRESHOLDS; ++t) {
            const int pr = (t < a.T && p > k.th[t]) ? 1 : 0;
            cp[t] += pr; ci[t] += pr & g;
        }
    }
#pragma unroll
    for (int t = 0; t < SALT_MAX_THRESHOLDS; ++t) {
        if (t < a.T) {                                   // uniform
            int vi = ci[t], vp = cp[t];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { vi += __shfl_xor(vi, o); vp += __shfl_xor(vp, o); }
            if ((tid & 63) == 0) { atomicAdd(&s_inter[t], vi); atomicAdd(&s_pred[t], vp); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cg += __shfl_xor(cg, o);
    if ((tid & 63) == 0) atomicAdd(&s_gt, cg);
    __syncthreads();
    if (tid < a.T) { a.inter[b * a.T + tid] = s_inter[tid]; a.pred[b * a.T + tid] = s_pred[tid]; }
    if (tid == 0) a.gt_count[b] = s_gt;
}

}  // namespace

extern "C" int64_t salt_lovasz_split_words(int P) {
    const int S = (P + SL - 1) / SL;
    if (P < 2 * SL || S > SMAXSEG) return 0;
    return (int64_t)3 * S * 256 + 2 * S;
}

extern "C" int salt_lovasz_hinge(const salt_lovasz_args* a, void* stream) {
    if (!a || !a->logits || !a->target || a->B < 1 || a->P < 0 || !a->ws_keys || !a->ws_vals || !a->loss_per_image || !a->loss)
        SALT_FAIL(SALT_E_BADARG, "lovasz: bad args");
    static const bool plain = getenv("SALT_LOVASZ_PLAIN") != nullptr;       // A/B switch: the non-prefetching kernel
    static const bool nosplit = getenv("SALT_LOVASZ_NOSPLIT") != nullptr;   // A/B switch: one workgroup per image
    const int S = (a->P + SL - 1) / SL;
    if (a->ws_split && !plain && !nosplit && a->P >= 2 * SL && S <= SMAXSEG) {
        hipStream_t st = (hipStream_t)stream;
        LovaszSplit sp{S, a->ws_split, a->ws_split + (int64_t)3 * a->B * S * 256, reinterpret_cast<float*>(a->ws_split + (int64_t)3 * a->B * S * 256 + (int64_t)a->B * S)};
        const dim3 grid(a->B * S);
        hipLaunchKernelGGL(lovasz_keys_kernel, grid, dim3(ST), 0, st, *a, sp);
        const size_t lds = (size_t)(S * 256 + S) * sizeof(unsigned);
        for (int pass = 0; pass < 4; ++pass) hipLaunchKernelGGL(lovasz_pass_kernel, grid, dim3(ST), lds, st, *a, sp, pass);
        hipLaunchKernelGGL(lovasz_scan_kernel, grid, dim3(ST), 0, st, *a, sp);
        SALT_CHECK_LAUNCH();
        hipLaunchKernelGGL(lovasz_mean_split_kernel, dim3(1), dim3(256), 0, st, sp.part, a->B, S, a->loss_scale, a->loss_per_image, a->loss);
        SALT_CHECK_LAUNCH();
        return SALT_OK;
    }
    if (plain) hipLaunchKernelGGL(lovasz_kernel, dim3(a->B), dim3(LT), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(lovasz_pf_kernel, dim3(a->B), dim3(LT), 0, (hipStream_t)stream, *a);
    SALT_CHECK_LAUNCH();
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->loss_per_image, a->B, a->loss_scale, a->loss);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_bce_dice_parts(const salt_bce_dice_args* a) {
    if (!a || a->B < 1 || a->C < 1 || a->HW < 1) return -1;
    return a->B * a->C * bce_ppp(a->HW, nullptr);
}

extern "C" int salt_bce_dice(const salt_bce_dice_args* a, void* stream) {
    if (!a || !a->logits || !a->target || a->B < 1 || a->C < 1 || a->HW < 1 || !a->partials || !a->sums || !a->loss) SALT_FAIL(SALT_E_BADARG, "bce_dice: bad args");
    int per = 0;
    const int ppp = bce_ppp(a->HW, &per);
    if (a->nparts != a->B * a->C * ppp) SALT_FAIL(SALT_E_BADARG, "bce_dice: nparts %d, expected %d", a->nparts, a->B * a->C * ppp);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_dice_partial_kernel, dim3(a->nparts), dim3(256), 0, st, a->logits, a->target, a->HW, ppp, per, a->partials);
    SALT_CHECK_LAUNCH();
    hipLaunchKernelGGL(bce_dice_finalize_kernel, dim3(1), dim3(64), 0, st, a->partials, a->B, a->C, ppp, a->HW, a->dice_weight, a->bce_weight, a->loss_scale, a->sums, a->loss);
    SALT_CHECK_LAUNCH();
    if (a->dlogits) {
        const int64_t n = (int64_t)a->B * a->C * a->HW;
        const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(bce_dice_grad_kernel, dim3(blocks), dim3(256), 0, st, a->logits, a->target, a->B, a->C, a->HW, a->sums, a->dice_weight, a->bce_weight, a->loss_scale, a->dlogits);
        SALT_CHECK_LAUNCH();
    }
    return SALT_OK;
}

extern "C" int salt_adam(const salt_adam_args* a, void* stream) {
    if (!a || !a->param || !a->grad || !a->exp_avg || !a->exp_avg_sq || !a->hyper || a->n < 0) SALT_FAIL(SALT_E_BADARG, "adam: bad args");
    if (a->n == 0) return SALT_OK;
    if ((reinterpret_cast<uintptr_t>(a->param) | reinterpret_cast<uintptr_t>(a->grad) | reinterpret_cast<uintptr_t>(a->exp_avg) | reinterpret_cast<uintptr_t>(a->exp_avg_sq)) & 15)
        SALT_FAIL(SALT_E_BADARG, "adam: buffers must be 16-byte aligned");
    const int64_t n4 = (a->n + 3) / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a->param, a->grad, a->exp_avg, a->exp_avg_sq, a->n, a->hyper);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_adam_tick(const salt_adam_tick_args* a, void* stream) {
    if (!a || !a->hyper || !a->step) SALT_FAIL(SALT_E_BADARG, "adam_tick: bad args");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->hyper, a->step);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_tta_mean(const salt_tta_mean_args* a, void* stream) {
    if (!a || !a->logits || !a->prob || a->V < 1 || a->V > 16 || !a->flip_ud || !a->flip_lr) SALT_FAIL(SALT_E_BADARG, "tta_mean: bad args");
    TtaKP p;
    p.logits = a->logits; p.prob = a->prob; p.V = a->V; p.B = a->B; p.C = a->C; p.H = a->H; p.W = a->W;
    if (a->method < 0 || a->method > 3) SALT_FAIL(SALT_E_BADARG, "tta_mean: method %d (0 mean, 1 max, 2 min, 3 gmean)", a->method);
    p.method = a->method;
    for (int v = 0; v < a->V; ++v) {
        p.ud[v] = a->flip_ud[v]; p.lr[v] = a->flip_lr[v];
        const int k = a->rot ? ((a->rot[v] % 4) + 4) % 4 : 0;
        if (k && a->H != a->W) SALT_FAIL(SALT_E_BADARG, "tta_mean: rotated variants need square maps");
        p.rq[v] = (4 - k) % 4;
    }
    const int64_t n = (int64_t)a->B * a->C * a->H * a->W;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(tta_mean_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_flip(const salt_flip_args* a, void* stream) {
    if (!a || !a->x || !a->y) SALT_FAIL(SALT_E_BADARG, "flip: bad args");
    const int64_t n = (int64_t)a->B * a->C * a->H * a->W;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(flip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

static bool crop_ok(int B, int C, int H, int W, int cls, int top, int left, int h, int w) {
    return B >= 1 && C >= 1 && cls >= 0 && cls < C && h >= 1 && w >= 1 && top >= 0 && left >= 0 && top + h <= H && left + w <= W;
}

extern "C" int salt_crop_threshold(const salt_crop_threshold_args* a, void* stream) {
    if (!a || !a->prob || !a->mask || !crop_ok(a->B, a->C, a->H, a->W, a->cls, a->top, a->left, a->h, a->w)) SALT_FAIL(SALT_E_BADARG, "crop_threshold: bad args");
    const int64_t n = (int64_t)a->B * a->h * a->w;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(crop_threshold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_iou_sweep(const salt_iou_sweep_args* a, void* stream) {
    if (!a || !a->prob || !a->gt || !a->thresholds || !a->inter || !a->pred || !a->gt_count || a->T < 1 || a->T > SALT_MAX_THRESHOLDS ||
        !crop_ok(a->B, a->C, a->H, a->W, a->cls, a->top, a->left, a->h, a->w)) SALT_FAIL(SALT_E_BADARG, "iou_sweep: bad args");
    IouKP k;
    k.a = *a;
    for (int t = 0; t < SALT_MAX_THRESHOLDS; ++t) k.th[t] = t < a->T ? a->thresholds[t] : 2.0;
    hipLaunchKernelGGL(iou_sweep_kernel, dim3(a->B), dim3(256), 0, (hipStream_t)stream, k);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}
