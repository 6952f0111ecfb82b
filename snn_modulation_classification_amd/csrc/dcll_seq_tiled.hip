// dcll_seq_tiled.hip — k_lif_seq_c32t, the fused all-T layer kernel for large planes (second translation unit of
// libdcll_hip.so).  Built with -fno-slp-vectorize on top of the common flags: the SLP vectoriser pairs the scalar
// trace updates into v_pk_* forms whose 64-bit register pairs push the kernel ~70 VGPRs over the 256 it has
// (measured: 76 spilled registers with, 0-3 without).
#include "dcll_internal.h"

// ------------------------------------------------------------------------------------------------------------
// k_lif_seq_c32t — the same 32 -> 32 channel 7x7 layer over all T steps on a LARGE plane (H % 8 == 0, W % 32 == 0;
// the reference's default 128x128 I/Q plane, test_radio_ml.py:52): one workgroup per (sample, 8 x 32 pixel tile).
// Same machine as k_lif_seq_c32 (8 waves x 4 input channels, weights in registers, systolic accumulator hand-off,
// one barrier per stage, epilogue split by register quad), with these differences:
//   - an MFMA tile is ONE image row of 32 pixels (lane&31 = column), so every spike word is still one ballot half
//     and the B-fragment reads of a half-wave are 32 consecutive floats (conflict free);
//   - the workgroup keeps the eps1 traces of its tile PLUS the 3-pixel halo (14 rows x 38 columns per channel,
//     row stride TRW, channel stride TCH = 32 mod 64 banks) and recomputes the halo redundantly (2.08x trace work,
//     <2 % of the stage; the halo values are the same fp32 ops on the same inputs as the owning tile computes, so the
//     result stays bit-exact).  Pixels outside the plane stay 0 = the convolution's zero padding;
//   - only wave w ever reads its 4 channels of the image, so ONE image suffices: region rows are advanced to the next
//     step in place as soon as the wave's own chain no longer needs them (row r is last read by tile r):
//       end of stage m=2: rows 0-2 -> step t+1      end of stage m=0: rows 9-11  -> step t  (first read by tile 3)
//       end of stage m=5: rows 3-5 -> step t+1      end of stage m=1: rows 12-13 -> step t  (first read by tile 6)
//       end of stage m=7: rows 6-8 -> step t+1
//     with the input spike words fetched at the start of the stage and consumed after the MFMA chain.
// eps0 of the region lives in registers (5 row groups x 2 slots x 4 channels).
// ------------------------------------------------------------------------------------------------------------
constexpr int TRW = 38, TCH = 544, TIMG = 32 * TCH;
constexpr int TGROUP = 3 * TRW;         // floats per row group (3 rows; the 5th group has 2)

template <bool REFRACTORY, int OUT>     // OUT bit0: pv, bit1: v
__global__ __launch_bounds__(512) void k_lif_seq_c32t(const uint32_t *__restrict__ spk_in, const float *__restrict__ W,
                                                       const float *__restrict__ bias, const float *__restrict__ tau4,
                                                       float *__restrict__ eps0_g, float *__restrict__ eps1_g,
                                                       float *__restrict__ arp_g, uint32_t *__restrict__ spk_out,
                                                       float *__restrict__ pv_out, float *__restrict__ v_out, int T,
                                                       int B, int H, int Wd, float alpharp, float wrp)
{
    __shared__ __attribute__((aligned(16))) float lds[TIMG + NWAVE * 2 * SLOT_FLOATS + 32];
    float *slots = lds + TIMG;
    float *sbias = slots + NWAVE * 2 * SLOT_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = w & 3, wpar = w >> 2;
    const int wpr = Wd >> 5;                                  // spike words per image row = tiles per row
    const int ntile = (H >> 3) * wpr;
    const long b = blockIdx.x / ntile;
    const int tile = blockIdx.x % ntile;
    const int y0 = (tile / wpr) * 8, tx = tile % wpr, x0 = tx * 32;
    const long HW = (long)H * Wd;
    const long words = HW >> 5;

    for (int i = tid; i < TIMG; i += 512) lds[i] = 0.0f;
    if (tid < 32) sbias[tid] = bias[tid];

    // trace element (group k, slot s) of a lane: region element idx = lane + 64 s of the group's 3 (2) rows, i.e.
    // row 3k + idx / 38, column idx % 38; its LDS offset inside a channel image is simply 114 k + idx.
    int erow[2], egx[2], eoff[2];
    bool colok[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int idx = lane + 64 * s;
        erow[s] = idx / TRW;
        egx[s] = x0 - 3 + (idx % TRW);
        colok[s] = (unsigned)egx[s] < (unsigned)Wd;
        eoff[s] = colok[s] ? erow[s] * wpr + (egx[s] >> 5) : 0;      // word offset inside the group's rows
    }
    // global position of element (k, s): row gy = y0 - 3 + 3k + erow[s], column egx[s]; valid if inside the plane and
    // inside the group (idx < 114, or < 76 for the last group)
    auto in_group = [&](int k, int s) -> bool { return lane + 64 * s < (k < 4 ? TGROUP : 2 * TRW); };
    auto elem_ok = [&](int k, int s) -> bool {
        return in_group(k, s) && colok[s] && (unsigned)(y0 - 3 + 3 * k + erow[s]) < (unsigned)H;
    };
    const uint32_t *in_b = spk_in + (b * 32 + 4 * w) * words;
    const long in_step = (long)B * 32 * words;
    // spike words of (step ts, group k): xw[s][c]; wave-uniform base + one 32-bit lane offset per slot
    auto fetch = [&](int k, int ts, uint32_t (&xw)[2][4]) {
        const uint32_t *base = in_b + (long)ts * in_step + (long)(y0 - 3 + 3 * k) * wpr;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool ok = elem_ok(k, s);
#pragma unroll
            for (int c = 0; c < 4; ++c) xw[s][c] = ok ? (base + c * words)[eoff[s]] : 0u;
        }
    };
    float e0[5][2][4];
    __syncthreads();        // image zeroed

    // prologue: state of the region from HBM (0 outside the plane), advanced to step 0 -> image
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        uint32_t xw[2][4];
        fetch(k, 0, xw);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool ok = elem_ok(k, s);
            const long goff = (long)(y0 - 3 + 3 * k + erow[s]) * Wd + egx[s];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ta = tau4[0 * 32 + 4 * w + c], tm = tau4[1 * 32 + 4 * w + c];
                const float tas = tau4[2 * 32 + 4 * w + c], ts = tau4[3 * 32 + 4 * w + c];
                const long gidx = (b * 32 + 4 * w + c) * HW + goff;
                float e0v = ok ? eps0_g[gidx] : 0.0f, e1 = ok ? eps1_g[gidx] : 0.0f;
                const float xin = (float)((xw[s][c] >> (egx[s] & 31)) & 1u);
                trace_update(xin, ta, tm, tas, ts, e0v, e1);
                e0[k][s][c] = e0v;
                if (in_group(k, s)) lds[(4 * w + c) * TCH + k * TGROUP + lane + 64 * s] = e1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // one group at a time: keeps the prologue's register peak low
    }
    // weight fragments (after the state prologue, so that its loads are not in flight on top of these 98 registers)
    float wf[2][49];
#pragma unroll
    for (int cp = 0; cp < 2; ++cp)
#pragma unroll
        for (int k = 0; k < 49; ++k) wf[cp][k] = W[((long)j * 32 + 4 * w + 2 * cp + h) * 49 + k];

    // refractory trace of my epilogue share: tiles (rows) m = 2k + wpar, quad wq: channel rr + 8 wq + 4h, column j
    float arp[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            arp[k][rr] = REFRACTORY ? arp_g[(b * 32 + rr + 8 * wq + 4 * h) * HW + (long)(y0 + 2 * k + wpar) * Wd + x0 + j]
                                    : 0.0f;
    const int bbase = (4 * w + h) * TCH + j;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) asm volatile("" ::"v"(arp[k][rr]));
#pragma unroll
    for (int cp = 0; cp < 2; ++cp)
#pragma unroll
        for (int k = 0; k < 49; ++k) asm volatile("" ::"v"(wf[cp][k]));
    __syncthreads();

    // advance group K of my 4 channels by one step with the fetched spike words
    auto advance = [&](auto KC, const uint32_t (&xw)[2][4]) {
        constexpr int K = decltype(KC)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float ta = tau4[0 * 32 + 4 * w + c], tm = tau4[1 * 32 + 4 * w + c];
            const float tas = tau4[2 * 32 + 4 * w + c], ts = tau4[3 * 32 + 4 * w + c];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (in_group(K, s)) {
                    float *p = lds + (4 * w + c) * TCH + K * TGROUP + lane + 64 * s;
                    float e1 = *p, e0v = e0[K][s][c];
                    // (opaque to the SLP vectoriser: paired v_pk_* forms of these updates cost ~70 spilled registers)
                    asm volatile("" : "+v"(e1), "+v"(e0v));
                    const float xin = (float)((xw[s][c] >> (egx[s] & 31)) & 1u);
                    trace_update(xin, ta, tm, tas, ts, e0v, e1);
                    asm volatile("" : "+v"(e1), "+v"(e0v));
                    e0[K][s][c] = e0v;
                    *p = e1;
                }
            }
        }
    };

    const int nstage = 8 * T + 9;
    for (int g = 0; g < nstage; ++g) {
        // ---- (1) epilogue share: quad wq of tile qe = g - 8 ----
        const int qe = g - 8;
        if (qe >= 0 && qe < 8 * T && ((qe & 1) == wpar)) {
            __builtin_amdgcn_s_setprio(1);
            const int te = qe >> 3, me = qe & 7;
            const f32x4 v4 = *((const f32x4 *)(slots + (7 * 2 + ((g - 1) & 1)) * SLOT_FLOATS) + wq * 64 + lane);
            const long obase = ((long)te * B + b) * 32 + 8 * wq + 4 * h;      // + rr = channel
            const long oelem = obase * HW + (long)(y0 + me) * Wd + x0 + j;    // + rr*HW
            float *pvp = pv_out + oelem, *vp = v_out + oelem;
            auto quad = [&](float (&ar)[4]) {
                uint32_t myword = 0;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    float v = v4[rr];
                    bool s;
                    if (REFRACTORY) v = refractory(v4[rr], ar[rr], alpharp, wrp, s);
                    else s = v > 0.0f;
                    unsigned long long mk = __ballot(s);
                    uint32_t mine = h ? (uint32_t)(mk >> 32) : (uint32_t)mk;
                    myword = (j == rr) ? mine : myword;
                    if (OUT & 1) pvp[rr * HW] = sigmoidf_dev(v);
                    if (OUT & 2) vp[rr * HW] = v;
                }
                if (spk_out && j < 4) spk_out[(obase + j) * words + (long)(y0 + me) * wpr + tx] = myword;
            };
            switch (me >> 1) {
            case 0: quad(arp[0]); break;
            case 1: quad(arp[1]); break;
            case 2: quad(arp[2]); break;
            default: quad(arp[3]); break;
            }
            __builtin_amdgcn_s_setprio(0);
        }
        const int q = g - w;
        if (q >= 0 && q < 8 * T) {
            const int m = q & 7, t = q >> 3;
            // ---- (2a) spike words of the row group this stage advances (consumed after the chain) ----
            //   m = 2, 5, 7 -> groups 0, 1, 2 to step t+1;   m = 0, 1 -> groups 3, 4 to step t (t >= 1)
            int tg = -1, tstep = 0;
            if (m == 2) { tg = 0; tstep = t + 1; }
            else if (m == 5) { tg = 1; tstep = t + 1; }
            else if (m == 7) { tg = 2; tstep = t + 1; }
            else if (m == 0) { tg = 3; tstep = t; }
            else if (m == 1) { tg = 4; tstep = t; }
            if (tstep >= T || tstep < 1) tg = -1;
            uint32_t xw[2][4];
            if (tg >= 0) fetch(tg, tstep, xw);
            // ---- (3) my K-slice of the chain ----
            f32x16 acc;
            if (w == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = sbias[(r & 3) + 8 * (r >> 2) + 4 * h];
            } else {
                const f32x4 *sp = (const f32x4 *)(slots + ((w - 1) * 2 + ((g - 1) & 1)) * SLOT_FLOATS) + lane;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 v4 = sp[c * 64];
                    acc[4 * c + 0] = v4[0]; acc[4 * c + 1] = v4[1]; acc[4 * c + 2] = v4[2]; acc[4 * c + 3] = v4[3];
                }
            }
            const int i0 = bbase + m * TRW;
            auto tapval = [&](int cp, int off) -> float { return lds[i0 + cp * 2 * TCH + off]; };
            float bq[2][7];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) bq[0][kx] = tapval(0, kx);
#pragma unroll
            for (int r = 0; r < 14; ++r) {
                if (r + 1 < 14) {
                    const int cpn = (r + 1) / 7, kyn = (r + 1) % 7;
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) bq[(r + 1) & 1][kx] = tapval(cpn, kyn * TRW + kx);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[r / 7][(r % 7) * 7 + kx], bq[r & 1][kx], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 *dp = (f32x4 *)(slots + (w * 2 + (g & 1)) * SLOT_FLOATS) + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 v4 = {acc[4 * c + 0], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
                dp[c * 64] = v4;
            }
            // ---- (2b) advance the row group (its rows are no longer read by this step's remaining tiles) ----
            switch (tg) {       // wave-uniform: keeps e0[][][] statically indexed (registers)
            case 0: advance(std::integral_constant<int, 0>{}, xw); break;
            case 1: advance(std::integral_constant<int, 1>{}, xw); break;
            case 2: advance(std::integral_constant<int, 2>{}, xw); break;
            case 3: advance(std::integral_constant<int, 3>{}, xw); break;
            case 4: advance(std::integral_constant<int, 4>{}, xw); break;
            default: break;
            }
        }
        __syncthreads();
    }

    // state back to HBM: the interior of the region (rows 3..10, columns 3..34)
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int idx = lane + 64 * s, rr = 3 * k + erow[s], cc = idx % TRW;
            if (in_group(k, s) && rr >= 3 && rr < 11 && cc >= 3 && cc < 35) {
                const long goff = (long)(y0 - 3 + rr) * Wd + egx[s];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const long gidx = (b * 32 + 4 * w + c) * HW + goff;
                    eps0_g[gidx] = e0[k][s][c];
                    eps1_g[gidx] = lds[(4 * w + c) * TCH + k * TGROUP + idx];
                }
            }
        }
    if (REFRACTORY) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                arp_g[(b * 32 + rr + 8 * wq + 4 * h) * HW + (long)(y0 + 2 * k + wpar) * Wd + x0 + j] = arp[k][rr];
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_lif_seq_c1t — the first layer (c_in = 1, one input spike per step as a cell index or raw IQ) on a large plane:
// one 256-thread workgroup per (sample, 8 x 32 pixel tile), all T steps.  Same MFMA form as k_lif_seq_c1 (49 taps
// padded per kernel row to 4 k-pairs with a ZERO weight on the pad tap, weight-stationary in 28 VGPRs); the tile's
// trace region (14 x 38 with the 3-pixel halo, recomputed redundantly like in k_lif_seq_c32t) lives in registers
// (eps0, eps1: 3 elements per thread) with eps1 mirrored into an LDS plane for the B fragments.
// ------------------------------------------------------------------------------------------------------------
constexpr int C1T_MAXT = 4096;
constexpr int C1T_REGION = 14 * TRW;        // 532

template <bool REFRACTORY>
__global__ __launch_bounds__(256) void k_lif_seq_c1t(int c_out, const int32_t *__restrict__ cells,
                                                      const float *__restrict__ iq, const float *__restrict__ thr_i,
                                                      const float *__restrict__ thr_q, int L, int t0,
                                                      const float *__restrict__ W, const float *__restrict__ bias,
                                                      const float *__restrict__ tau4, float *__restrict__ eps0_g,
                                                      float *__restrict__ eps1_g, float *__restrict__ arp_g,
                                                      uint32_t *__restrict__ spk_out, float *__restrict__ pv_out,
                                                      float *__restrict__ v_out, int T, int B, int H, int Wd,
                                                      float alpharp, float wrp)
{
    __shared__ float plane[C1T_REGION + 8];     // + the pad tap's read past the last row (zero weight, finite data)
    __shared__ float sbias[32];
    __shared__ int scell[C1T_MAXT];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpr = Wd >> 5;
    const int ntile = (H >> 3) * wpr;
    const long b = blockIdx.x / ntile;
    const int tile = blockIdx.x % ntile;
    const int y0 = (tile / wpr) * 8, tx = tile % wpr, x0 = tx * 32;
    const long HW = (long)H * Wd;
    const long words = HW >> 5;
    const float alpha = tau4[0], tau_m = tau4[1], alphas = tau4[2], tau_s = tau4[3];
    for (int i = tid; i < C1T_REGION + 8; i += 256) plane[i] = 0.0f;
    if (iq) {
        // quantisation of iq2spiketrain (data/utils.py:60-82) in threshold form, as k_iq_encode: cell = row * W + col
        for (int t = tid; t < T; t += 256) {
            const float vi = iq[(b * 2 + 0) * L + t0 + t], vq = iq[(b * 2 + 1) * L + t0 + t];
            int ci = 0, cq = 0;
            for (int k = 0; k < Wd - 1; ++k) ci += vi >= thr_i[k];
            for (int k = 0; k < H - 1; ++k) cq += vq >= thr_q[k];
            scell[t] = cq * Wd + ci;
        }
    }
    // my (up to) 3 region elements: idx = tid + 256 s -> region row idx / 38, column idx % 38; gpix = its pixel
    // index in the plane, or -1 outside the plane / region (then it stays 0 = zero padding)
    int gpix[3];
    float e0[3], e1[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int idx = tid + 256 * s, rr = idx / TRW, cc = idx % TRW;
        const int gy = y0 - 3 + rr, gx = x0 - 3 + cc;
        const bool ok = idx < C1T_REGION && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)Wd;
        gpix[s] = ok ? gy * Wd + gx : -1;
        e0[s] = ok ? eps0_g[b * HW + gpix[s]] : 0.0f;
        e1[s] = ok ? eps1_g[b * HW + gpix[s]] : 0.0f;
    }
    float wf[7][4];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kx = 2 * i + h;
            wf[ky][i] = (kx < 7 && j < c_out) ? W[j * 49 + ky * 7 + kx] : 0.0f;
        }
    // refractory trace of my two tiles (rows y0 + 2w + tl): arp[tl][r] <-> channel (r&3)+8(r>>2)+4h, column x0 + j
    float arp[2][16];
    if (tid < 32) sbias[tid] = tid < c_out ? bias[tid] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
            arp[tl][r] = (REFRACTORY && co < c_out)
                             ? arp_g[(b * c_out + co) * HW + (long)(y0 + 2 * w + tl) * Wd + x0 + j] : 0.0f;
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int cell = iq ? scell[t] : cells[(long)t * B + b];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            trace_update(cell == gpix[s] ? 1.0f : 0.0f, alpha, tau_m, alphas, tau_s, e0[s], e1[s]);
            if (tid + 256 * s < C1T_REGION) plane[tid + 256 * s] = e1[s];
        }
        __syncthreads();
        const long obase = ((long)t * B + b) * c_out;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int m = 2 * w + tl;
            const float *bp = plane + m * TRW + j + h;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = sbias[(r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[ky][i], bp[ky * TRW + 2 * i], acc, 0, 0, 0);
            uint32_t myword = 0;
            const long opix = (long)(y0 + m) * Wd + x0 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[r];
                bool s;
                if (REFRACTORY) v = refractory(acc[r], arp[tl][r], alpharp, wrp, s);
                else s = v > 0.0f;
                unsigned long long mk = __ballot(s);
                uint32_t mine = h ? (uint32_t)(mk >> 32) : (uint32_t)mk;
                myword = (j == r) ? mine : myword;
                if (co < c_out) {
                    if (pv_out) pv_out[(obase + co) * HW + opix] = sigmoidf_dev(v);
                    if (v_out) v_out[(obase + co) * HW + opix] = v;
                }
            }
            const int cow = (j & 3) + 8 * (j >> 2) + 4 * h;
            if (spk_out && j < 16 && cow < c_out) spk_out[(obase + cow) * words + (long)(y0 + m) * wpr + tx] = myword;
        }
        __syncthreads();
    }
    // state back: the interior of the region
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int idx = tid + 256 * s, rr = idx / TRW, cc = idx % TRW;
        if (gpix[s] >= 0 && rr >= 3 && rr < 11 && cc >= 3 && cc < 35) {
            eps0_g[b * HW + gpix[s]] = e0[s];
            eps1_g[b * HW + gpix[s]] = e1[s];
        }
    }
    if (REFRACTORY) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
                if (co < c_out) arp_g[(b * c_out + co) * HW + (long)(y0 + 2 * w + tl) * Wd + x0 + j] = arp[tl][r];
        }
    }
}

int dcll_launch_seq_c1t(const dcll_conv_desc *d, const int32_t *cells, const float *iq, const float *thr_i,
                        const float *thr_q, int L, int t0, const float *W, const float *b, const float *tau4,
                        float *eps0, float *eps1, float *arp, uint32_t *spk_out, float *pv_out, float *v_out, int T,
                        int B, hipStream_t st)
{
    if (iq && T > C1T_MAXT) return fail(DCLL_ERR_UNSUPPORTED, "fused IQ encoder: T exceeds 4096 steps");
    const long nwg = (long)B * (d->h / 8) * (d->w / 32);
    if (nwg > 0x7fffffffL) return fail(DCLL_ERR_INVALID, "sequence kernel: batch x tiles exceeds the grid limit");
    if (d->refractory)
        hipLaunchKernelGGL(k_lif_seq_c1t<true>, dim3((unsigned)nwg), dim3(256), 0, st, d->c_out, cells, iq, thr_i,
                           thr_q, L, t0, W, b, tau4, eps0, eps1, arp, spk_out, pv_out, v_out, T, B, d->h, d->w,
                           d->alpharp, d->wrp);
    else
        hipLaunchKernelGGL(k_lif_seq_c1t<false>, dim3((unsigned)nwg), dim3(256), 0, st, d->c_out, cells, iq, thr_i,
                           thr_q, L, t0, W, b, tau4, eps0, eps1, arp, spk_out, pv_out, v_out, T, B, d->h, d->w,
                           d->alpharp, d->wrp);
    HIP_CHECK_LAUNCH("k_lif_seq_c1t");
    return DCLL_OK;
}

int dcll_launch_seq_c32t(const dcll_conv_desc *d, const uint32_t *spk_in, const float *W, const float *b,
                         const float *tau4, float *eps0, float *eps1, float *arp, uint32_t *spk_out, float *pv_out,
                         float *v_out, int32_t T, int32_t B, hipStream_t st)
{
    const int out = (pv_out ? 1 : 0) | (v_out ? 2 : 0);
    const long nwg = (long)B * (d->h / 8) * (d->w / 32);
    if (nwg > 0x7fffffffL) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_sequence: batch x tiles exceeds the grid limit");
#define DCLL_LAUNCH_C32T(R, O)                                                                                          \
    hipLaunchKernelGGL((k_lif_seq_c32t<R, O>), dim3((unsigned)nwg), dim3(512), 0, st, spk_in, W, b, tau4, eps0, eps1,   \
                       arp, spk_out, pv_out, v_out, T, B, d->h, d->w, d->alpharp, d->wrp)
    if (d->refractory) {
        switch (out) {
        case 0: DCLL_LAUNCH_C32T(true, 0); break;
        case 1: DCLL_LAUNCH_C32T(true, 1); break;
        case 2: DCLL_LAUNCH_C32T(true, 2); break;
        default: DCLL_LAUNCH_C32T(true, 3); break;
        }
    } else {
        switch (out) {
        case 0: DCLL_LAUNCH_C32T(false, 0); break;
        case 1: DCLL_LAUNCH_C32T(false, 1); break;
        case 2: DCLL_LAUNCH_C32T(false, 2); break;
        default: DCLL_LAUNCH_C32T(false, 3); break;
        }
    }
#undef DCLL_LAUNCH_C32T
    HIP_CHECK_LAUNCH("k_lif_seq_c32t");
    return DCLL_OK;
}
