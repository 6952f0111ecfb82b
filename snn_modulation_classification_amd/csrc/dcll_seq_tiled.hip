// dcll_seq_tiled.hip — k_lif_seq_c32t, the fused all-T layer kernel for large planes (second translation unit of
// libdcll_hip.so).  Built with -fno-slp-vectorize on top of the common flags: the SLP vectoriser pairs the scalar
// trace updates into v_pk_* forms whose 64-bit register pairs push the kernel ~70 VGPRs over the 256 it has
// (measured: 76 spilled registers with, 0-3 without).
#include "dcll_internal.h"

// ------------------------------------------------------------------------------------------------------------
// k_lif_seq_c32t — the 32 -> 32 channel 7x7 layer over all T steps on a LARGE plane (H % 8 == 0, W % 32 == 0; the
// reference's default 128x128 I/Q plane, test_radio_ml.py:52): one workgroup per (sample, 8 x 32 pixel tile).
// The machine is k_lif_seq_c32d's (dcll_hip.hip): 8 waves x 4 input channels, weights in registers, two tiles per
// wave and stage as independent accumulator chains handed from wave to wave through single-buffered LDS slots, two
// barriers per stage, all non-MFMA work of a stage done by every wave between them (while the matrix pipe is empty),
// stage loop unrolled by four.  What differs:
//   - an MFMA tile is ONE image row of 32 pixels (lane&31 = column): a spike word is still one ballot half, a
//     half-wave's B reads are 32 consecutive floats (conflict free), and the pair (2p, 2p+1) shares 8 LDS rows per
//     channel pair (tap row ky of the second tile = tap row ky+1 of the first): 112 ds_read dwords per stage;
//   - the workgroup keeps the eps1 traces of its tile PLUS the 3-pixel halo (14 rows x 38 columns per channel, row
//     stride TRW, channel stride TCH = 32 mod 64 banks) and recomputes the halo redundantly (2.08x trace work; the
//     halo values are the same fp32 ops on the same inputs as the owning tile computes, so the result stays
//     bit-exact).  Pixels outside the plane stay 0 = the convolution's zero padding;
//   - only the wave that owns 4 input channels ever reads them, so ONE image suffices: region row r (last read by
//     tile min(r,7), first needed again by tile max(0,r-6)) is advanced in place between those two reads — in the
//     non-MFMA phase of the stage with pair index p:
//       p = 0: rows 5-7 -> step t     p = 1: rows 8-10 -> step t     p = 2: rows 11-13 -> step t
//       p = 3: rows 0-4 -> step t+1
//   - input spikes: one global load per lane fetches the (row, channel, word) triples of the next stage's row group
//     while the chains run; the lanes of a trace element pick their word with ds_bpermute.
// eps0 of the region lives in registers (9 element slots x 4 channels).
// ------------------------------------------------------------------------------------------------------------
constexpr int TRW = 38, TCH = 544, TIMG = 32 * TCH;

// row group advanced in the stage with pair index p: first region row, number of rows
struct tgroup { int row0, nrows; };
__device__ constexpr tgroup TG[4] = {{5, 3}, {8, 3}, {11, 3}, {0, 5}};

template <bool REFRACTORY, int OUT>     // OUT bit0: pv, bit1: v
__global__ __launch_bounds__(512) void k_lif_seq_c32t(const uint32_t *__restrict__ spk_in, const dcll_wsrc W,
                                                       const float *__restrict__ bias, const float *__restrict__ tau4,
                                                       const float *__restrict__ eps0_in,
                                                       const float *__restrict__ eps1_in, float *__restrict__ eps0_g,
                                                       float *__restrict__ eps1_g, float *__restrict__ arp_g,
                                                       uint32_t *__restrict__ spk_out, float *__restrict__ pv_out,
                                                       float *__restrict__ v_out, int T, int B, int H, int Wd,
                                                       float alpharp, float wrp)
{
    __shared__ __attribute__((aligned(16))) float lds[TIMG + (NWAVE * 2 + 1) * SLOT_FLOATS];
    float *slots = lds + TIMG;                  // [wave][tile of the pair][16 x 64]
    float *sbias = slots + NWAVE * 2 * SLOT_FLOATS;     // the bias as a slot-shaped tile: wave 0's chain input
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // chain position: channels 4w..4w+3
    const int wq = w & 3, wpar = w >> 2;                          // my epilogue share: quad wq of the pair's tile wpar
    const int wpr = Wd >> 5;                                  // spike words per image row = tiles per row
    const int ntile = (H >> 3) * wpr;
    const long b = blockIdx.x / ntile;
    const int tile = blockIdx.x % ntile;
    const int y0 = (tile / wpr) * 8, tx = tile % wpr, x0 = tx * 32;
    const long HW = (long)H * Wd;
    const long words = HW >> 5;

    for (int i = tid; i < TIMG; i += 512) lds[i] = 0.0f;
    // slot layout: float4 c of lane l = accumulator registers 4c..4c+3 = channels (r&3) + 8c + 4(l>>5)
    for (int i = tid; i < SLOT_FLOATS; i += 512) sbias[i] = bias[(i & 3) + 8 * (i >> 8) + 4 * ((i >> 7) & 1)];

    // trace element slot s of a lane inside a row group: element idx = lane + 64 s of the group's rows, i.e. group row
    // idx / 38, column idx % 38; its LDS offset inside a channel image is row0 * 38 + idx.
    auto erow = [&](int s) -> int { return (lane + 64 * s) / TRW; };
    auto ecol = [&](int s) -> int { return (lane + 64 * s) % TRW; };
    // spike-word fetch role of a lane: (group row fr, channel fc, word fw - 1 relative to the tile's own word tx)
    const int fr = lane / 12, fc = (lane % 12) / 3, fw = lane % 3;
    const bool fwok = (unsigned)(tx - 1 + fw) < (unsigned)wpr;
    const int foff = fwok ? fr * wpr + fc * (int)words + fw : 0;        // < 2^31: 32 channels of the plane fit an int
    const uint32_t *in_b = spk_in + (b * 32 + 4 * w) * words;
    const long in_step = (long)B * 32 * words;
    auto fetch = [&](int gi, int ts) -> uint32_t {       // gi, ts wave-uniform
        const int gy = y0 - 3 + TG[gi].row0 + fr;
        const bool ok = fr < TG[gi].nrows && fwok && (unsigned)gy < (unsigned)H;
        const uint32_t *base = in_b + (long)ts * in_step + (long)(y0 - 3 + TG[gi].row0) * wpr + (tx - 1);
        return ok ? base[foff] : 0u;
    };
    // input bit of element (slot s, channel c) out of the fetched words: source lane erow*12 + 3c + word select.
    // pb / sh: per-slot byte address of the source lane for c = 0 and the bit position; kept to 6 registers (the
    // asm keeps the compiler from hoisting all 12 (s, c) addresses out of the stage loop).
    int pb[3], sh[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        pb[s] = (erow(s) * 12 + (((ecol(s) - 3) >> 5) + 1)) * 4;
        sh[s] = (ecol(s) - 3) & 31;
    }
    // x * tau_s for x in {0, 1} without an int->float conversion: sign-extend the spike bit to a 0 / ~0 mask and AND
    // it onto tau_s (exact: the product is tau_s or +0.0)
    auto spike_times = [&](uint32_t word, int s, int c, float ts) -> float {
        int a = pb[s];
        asm volatile("" : "+v"(a));
        const int wv = __builtin_amdgcn_ds_bpermute(a + 12 * c, (int)word);
        const int mask = __builtin_amdgcn_sbfe(wv, sh[s], 1);
        return __int_as_float(mask & __float_as_int(ts));
    };
    float e0[4][3][4];
    __syncthreads();        // image zeroed

    // prologue: state of the region from HBM (0 outside the plane), advanced to step 0 -> image
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
        const uint32_t word = fetch(gi, 0);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (64 * s >= TG[gi].nrows * TRW) continue;
            const int idx = lane + 64 * s;
            const int gy = y0 - 3 + TG[gi].row0 + erow(s), gx = x0 - 3 + ecol(s);
            const bool ing = idx < TG[gi].nrows * TRW;
            const bool ok = ing && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)Wd;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ta = tau4[0 * 32 + 4 * w + c], tm = tau4[1 * 32 + 4 * w + c];
                const float tas = tau4[2 * 32 + 4 * w + c], ts = tau4[3 * 32 + 4 * w + c];
                const long gidx = (b * 32 + 4 * w + c) * HW + (long)gy * Wd + gx;
                float e0v = ok ? eps0_in[gidx] : 0.0f, e1 = ok ? eps1_in[gidx] : 0.0f;
                const float bb = tas * e0v;
                e0v = spike_times(word, s, c, ts) + bb;
                const float cc = ta * e1, dd = e0v * tm;
                e1 = cc + dd;
                e0[gi][s][c] = e0v;
                if (ing) lds[(4 * w + c) * TCH + TG[gi].row0 * TRW + idx] = e1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // one group at a time: keeps the prologue's register peak low
    }
    // weight fragments (after the state prologue, so that its loads are not in flight on top of these 98 registers)
    float wf[2][49];
    load_wf_c32(W, j, w, h, wf);

    // refractory trace of my epilogue share: tiles (rows) m = 2k + wpar (k = pair index), quad wq
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

    // advance row group GI of my 4 channels by one step with the fetched spike words.  Branch-free and batched: all
    // ds_bpermutes and image reads go out first, then the arithmetic, then the writes — one LDS round trip per phase.
    // Lanes past the end of the group are pointed at the 12 pad floats behind their channel's 14 x 38 image.
    auto advance = [&](auto GC, uint32_t word) {
        constexpr int GI = decltype(GC)::value;
        constexpr int NS = (TG[GI].nrows * TRW + 63) / 64;
        int off[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s)
            off[s] = (lane + 64 * s < TG[GI].nrows * TRW) ? TG[GI].row0 * TRW + lane + 64 * s : 14 * TRW;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {        // two channels at a time: 12 instead of 24 transient registers
            float xts[NS][2], e1[NS][2];        // x * tau_s, eps1
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int c = 2 * ch + k;
                    xts[s][k] = spike_times(word, s, c, tau4[3 * 32 + 4 * w + c]);
                    e1[s][k] = lds[(4 * w + c) * TCH + off[s]];
                }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = 2 * ch + k;
                const float ta = tau4[0 * 32 + 4 * w + c], tm = tau4[1 * 32 + 4 * w + c], tas = tau4[2 * 32 + 4 * w + c];
#pragma unroll
                for (int s = 0; s < NS; ++s) {      // dcll/pytorch_libdcll.py:493-494, every op rounded separately
                    const float bb = tas * e0[GI][s][c];
                    e0[GI][s][c] = xts[s][k] + bb;
                    const float cc = ta * e1[s][k];
                    const float dd = e0[GI][s][c] * tm;
                    e1[s][k] = cc + dd;
                }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int k = 0; k < 2; ++k) lds[(4 * w + 2 * ch + k) * TCH + off[s]] = e1[s][k];
        }
    };
    // spike words of the row group stage g advances (0 if it advances none): groups 0-2 -> step t (t >= 1),
    // group 3 -> step t+1 (t+1 < T)
    auto fetch_for_stage = [&](const int g) -> uint32_t {
        const int q = g - w;
        if (q < 0 || q >= 4 * T) return 0u;
        const int p = q & 3, t = q >> 2;
        if (p < 3) return t >= 1 ? fetch(p, t) : 0u;
        return t + 1 < T ? fetch(3, t + 1) : 0u;
    };
    uint32_t fword = fetch_for_stage(0);

    // one stage; U = g & 3 at compile time
    auto stage = [&](const int g, auto UC) {
        constexpr int U = decltype(UC)::value;
        const int q = g - w;
        const bool active = q >= 0 && q < 4 * T;
        const int p = q & 3, t = q >> 2;
        // ---- (0) epilogue share's input: quad wq of tile wpar of the pair qe = g - 8 (pair index U) that wave 7 finished last stage
        const int qe = g - 8;
        const bool epi = qe >= 0 && qe < 4 * T;
        f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
        if (epi) v4 = *((const f32x4 *)(slots + (7 * 2 + wpar) * SLOT_FLOATS) + wq * 64 + lane);
        // ---- (2) trace rows of this stage (their rows are not read between tile 2p-1 and tile 2p) ----
        if (active) {
            switch (p) {        // wave-uniform: keeps e0[][][] statically indexed (registers)
            case 0: if (t >= 1) advance(std::integral_constant<int, 0>{}, fword); break;
            case 1: if (t >= 1) advance(std::integral_constant<int, 1>{}, fword); break;
            case 2: if (t >= 1) advance(std::integral_constant<int, 2>{}, fword); break;
            default: if (t + 1 < T) advance(std::integral_constant<int, 3>{}, fword); break;
            }
        }
        // ---- (1) epilogue share ----
        if (epi) {
            const int te = qe >> 2, me = 2 * U + wpar;
            // stores as wave-uniform base (per value) + 32-bit lane offset: no 64-bit address VALU
            const long ubase = (((long)te * B + b) * 32 + 8 * wq) * HW;        // channel 8 wq of this step and sample
            float *pvb = pv_out + ubase, *vb = v_out + ubase;
            const unsigned loff = 4 * h * (unsigned)HW + (unsigned)((y0 + me) * Wd + x0 + j);   // + rr*HW
            uint32_t myword = 0;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float v = v4[rr];
                bool s;
                if (REFRACTORY) v = refractory(v4[rr], arp[U][rr], alpharp, wrp, s);
                else s = v > 0.0f;
                const unsigned long long mk = __ballot(s);
                const uint32_t mine = h ? (uint32_t)(mk >> 32) : (uint32_t)mk;
                myword = (j == rr) ? mine : myword;
                if (OUT & 1) (pvb + rr * HW)[loff] = sigmoidf_dev(v);
                if (OUT & 2) (vb + rr * HW)[loff] = v;
            }
            if (spk_out && j < 4)
                (spk_out + (ubase >> 5) + (long)(y0 + me) * wpr + tx)[(unsigned)(4 * h + j) * (unsigned)words] = myword;
        }
        //   chain inputs out of the slots (written in the previous stage) — read last in the phase: 32 registers that
        //   would otherwise be live across the epilogue and the trace rows (14 spilled VGPRs)
        f32x16 accA, accB;
        if (active) {
            // wave 0 starts both chains from the bias tile, wave w > 0 from the two tiles wave w-1 left: one code path
            // (a branch here costs 32 v_mov per stage to merge the accumulator tuples)
            const float *inA = (w == 0) ? sbias : slots + ((w - 1) * 2) * SLOT_FLOATS;
            const float *inB = (w == 0) ? sbias : slots + ((w - 1) * 2 + 1) * SLOT_FLOATS;
            const f32x4 *spa = (const f32x4 *)inA + lane, *spb = (const f32x4 *)inB + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 va = spa[c * 64], vb = spb[c * 64];
                accA[4 * c + 0] = va[0]; accA[4 * c + 1] = va[1]; accA[4 * c + 2] = va[2]; accA[4 * c + 3] = va[3];
                accB[4 * c + 0] = vb[0]; accB[4 * c + 1] = vb[1]; accB[4 * c + 2] = vb[2]; accB[4 * c + 3] = vb[3];
            }
        }
        // every slot read of this stage has completed before any wave writes its slots again
        lds_barrier();
        // spike words of the NEXT stage's row group: they land while the chains run
        fword = fetch_for_stage(g + 1);
        // ---- (3) my K-slice of both chains ----
        if (active) {
            // LDS rows 0..7 below the pair's first image row, per channel pair: row rho is tap row ky = rho of tile A
            // (rho <= 6) and tap row ky = rho - 1 of tile B (rho >= 1); next row fetched before the MFMAs of this one.
            const int i0 = bbase + 2 * p * TRW;
            // opaque 32-bit LDS bases — per channel pair one for rows 0..3 and one for rows 4..7 (8 rows x 38 floats do not
            // fit ds_read2_b32's 255-dword reach from one): every read of the chains is base + immediate (k_lif_seq_c32d:
            // left to the compiler, 28 address instructions per 196 MFMAs, on the pipe the MFMAs execute on)
            lds_cfloat *ib[2][2] = {{(lds_cfloat *)(lds + i0), (lds_cfloat *)(lds + i0 + 4 * TRW)},
                                    {(lds_cfloat *)(lds + i0 + 2 * TCH), (lds_cfloat *)(lds + i0 + 2 * TCH + 4 * TRW)}};
            asm volatile("" : "+v"(ib[0][0]), "+v"(ib[0][1]), "+v"(ib[1][0]), "+v"(ib[1][1]));
            float bq[2][7];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) bq[0][kx] = ib[0][0][kx];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cp = r / 8, rho = r % 8;
                if (r + 1 < 16) {
                    const int cpn = (r + 1) / 8, rhon = (r + 1) % 8;
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) bq[(r + 1) & 1][kx] = ib[cpn][rhon >> 2][(rhon & 3) * TRW + kx];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    if (rho <= 6)
                        accA = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cp][rho * 7 + kx], bq[r & 1][kx], accA, 0, 0, 0);
                    if (rho >= 1)
                        accB = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cp][(rho - 1) * 7 + kx], bq[r & 1][kx], accB, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (r == 14) {          // tile A is complete (its last tap row was rho = 6 of the second channel pair)
                    f32x4 *dpa = (f32x4 *)(slots + (w * 2) * SLOT_FLOATS) + lane;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        dpa[c * 64] = f32x4{accA[4 * c + 0], accA[4 * c + 1], accA[4 * c + 2], accA[4 * c + 3]};
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            f32x4 *dp = (f32x4 *)(slots + (w * 2 + 1) * SLOT_FLOATS) + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                dp[c * 64] = f32x4{accB[4 * c + 0], accB[4 * c + 1], accB[4 * c + 2], accB[4 * c + 3]};
        }
        // stage barrier: only the LDS traffic has to be complete, not the pv / spike stores of the epilogue
        lds_barrier();
    };

    const int nstage = 4 * T + 8;       // a multiple of 4
    for (int g = 0; g < nstage; g += 4) {
        stage(g + 0, std::integral_constant<int, 0>{});
        stage(g + 1, std::integral_constant<int, 1>{});
        stage(g + 2, std::integral_constant<int, 2>{});
        stage(g + 3, std::integral_constant<int, 3>{});
    }

    // state back to HBM: the interior of the region (rows 3..10, columns 3..34)
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (64 * s >= TG[gi].nrows * TRW) continue;
            int l2 = lane;
            asm volatile("" : "+v"(l2));            // recomputed here: not worth registers across the stage loop
            const int idx = l2 + 64 * s, rr = TG[gi].row0 + idx / TRW, cc = idx % TRW;
            if (idx < TG[gi].nrows * TRW && rr >= 3 && rr < 11 && cc >= 3 && cc < 35) {
                const long goff = (long)(y0 - 3 + rr) * Wd + x0 - 3 + cc;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const long gidx = (b * 32 + 4 * w + c) * HW + goff;
                    eps0_g[gidx] = e0[gi][s][c];
                    eps1_g[gidx] = lds[(4 * w + c) * TCH + TG[gi].row0 * TRW + idx];
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

// FAST: c_out == 32 and exactly the outputs spk_out + pv_out (see k_lif_seq_c1); 2: pv_out receives v (pv_presigmoid).
template <bool REFRACTORY, int FAST = 0>
__global__ __launch_bounds__(256) void k_lif_seq_c1t(int c_out, const int32_t *__restrict__ cells,
                                                      const float *__restrict__ iq, const float *__restrict__ thr_i,
                                                      const float *__restrict__ thr_q, const dcll_iq_tail tail, int L, int t0,
                                                      const dcll_wsrc W, const float *__restrict__ bias,
                                                      const float *__restrict__ tau4, const float *__restrict__ eps0_in,
                                                      const float *__restrict__ eps1_in, float *__restrict__ eps0_g,
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
        const float *ti, *tq;
        iq_tables(thr_i, thr_q, tail, b, ti, tq);
        for (int t = tid; t < T; t += 256) {
            const float vi = iq[(b * 2 + 0) * L + t0 + t], vq = iq[(b * 2 + 1) * L + t0 + t];
            int ci = 0, cq = 0;
            for (int k = 0; k < Wd - 1; ++k) ci += vi >= ti[k];
            for (int k = 0; k < H - 1; ++k) cq += vq >= tq[k];
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
        e0[s] = ok ? eps0_in[b * HW + gpix[s]] : 0.0f;
        e1[s] = ok ? eps1_in[b * HW + gpix[s]] : 0.0f;
    }
    float wf[7][4];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kx = 2 * i + h;
            wf[ky][i] = (kx < 7 && j < c_out) ? W.at(j * 49 + ky * 7 + kx, j) : 0.0f;
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
        lds_barrier();      // LDS-only: does not wait for this step's pv stores
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
            int myword = 0;
            const long opix = (long)(y0 + m) * Wd + x0 + j;
            float *pvb = pv_out + obase * HW;                       // wave-uniform base of this step's pv planes
            const unsigned loff = 4 * h * (unsigned)HW + (unsigned)opix;   // 32 planes fit 32 bits (launcher check)
            static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[r];
                bool s;
                if (REFRACTORY) v = refractory(acc[r], arp[tl][r], alpharp, wrp, s);
                else s = v > 0.0f;
                const unsigned long long mk = __ballot(s);
                // the two spike words of register r parked in lanes r / 32 + r (see k_lif_seq_c1 for the wait states)
                asm("s_nop 1\n\tv_writelane_b32 %0, %1, %3\n\ts_nop 1\n\tv_writelane_b32 %0, %2, %4"
                    : "+v"(myword) : "s"((uint32_t)mk), "s"((uint32_t)(mk >> 32)), "n"(r), "n"(32 + r));
                if (FAST) {
                    (pvb + ((r & 3) + 8 * (r >> 2)) * HW)[loff] = FAST == 2 ? v : sigmoidf_dev(v);
                } else if (co < c_out) {
                    if (pv_out) pv_out[(obase + co) * HW + opix] = sigmoidf_dev(v);
                    if (v_out) v_out[(obase + co) * HW + opix] = v;
                }
            });
            const int cow = (j & 3) + 8 * (j >> 2) + 4 * h;
            if (FAST) {
                if (j < 16) (spk_out + obase * words + (long)(y0 + m) * wpr + tx)[(unsigned)cow * (unsigned)words] = (uint32_t)myword;
            } else if (spk_out && j < 16 && cow < c_out) {
                spk_out[(obase + cow) * words + (long)(y0 + m) * wpr + tx] = (uint32_t)myword;
            }
        }
        lds_barrier();      // LDS-only: does not wait for this step's pv stores
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

// The workgroups of one launch read their tile's initial traces PLUS a 3-pixel halo that belongs to neighbouring tiles
// and write their interior back at the end; nothing orders the workgroups of a grid, so a late workgroup could read a
// neighbour's already advanced state as its initial halo.  Race-free by construction: the initial state is snapshot
// into caller scratch (stream-ordered device copies) and every read goes to the snapshot, every write to the state.
static int snapshot_state(const float *eps0, const float *eps1, float *scratch, long n, hipStream_t st, const char *who)
{
    if (!scratch) return fail(DCLL_ERR_INVALID, "planes larger than 16x16 need state_scratch (2*B*c_in*h*w floats)", who);
    if (hipMemcpyAsync(scratch, eps0, n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(scratch + n, eps1, n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        (void)hipGetLastError();
        return fail(DCLL_ERR_LAUNCH, "state snapshot copy failed", who);
    }
    return DCLL_OK;
}

int dcll_launch_seq_c1t(const dcll_conv_desc *d, const int32_t *cells, const float *iq, const float *thr_i,
                        const float *thr_q, dcll_iq_tail tail, int L, int t0, dcll_wsrc W, const float *b, const float *tau4,
                        float *eps0, float *eps1, float *arp, uint32_t *spk_out, float *pv_out, float *v_out,
                        float *state_scratch, int T, int B, hipStream_t st, bool presig)
{
    if (iq && T > C1T_MAXT) return fail(DCLL_ERR_UNSUPPORTED, "fused IQ encoder: T exceeds 4096 steps");
    const long nwg = (long)B * (d->h / 8) * (d->w / 32);
    if (nwg > 0x7fffffffL) return fail(DCLL_ERR_INVALID, "sequence kernel: batch x tiles exceeds the grid limit");
    const long nstate = (long)B * d->h * d->w;
    int rc = snapshot_state(eps0, eps1, state_scratch, nstate, st, "dcll_conv_lif_sequence_cells/_iq");
    if (rc) return rc;
    const float *eps0_in = state_scratch, *eps1_in = state_scratch + nstate;
    const bool fastpath = d->c_out == 32 && spk_out && pv_out && !v_out && (long)d->h * d->w * 32 < (1L << 30);
    if (presig && !fastpath) { v_out = pv_out; pv_out = nullptr; }        // (the caller made sure only one of them is wanted)
#define DCLL_LAUNCH_C1T(R, F)                                                                                           \
    hipLaunchKernelGGL((k_lif_seq_c1t<R, F>), dim3((unsigned)nwg), dim3(256), 0, st, d->c_out, cells, iq, thr_i, thr_q,  \
                       tail, L, t0, W, b, tau4, eps0_in, eps1_in, eps0, eps1, arp, spk_out, pv_out, v_out, T, B, d->h, d->w,         \
                       d->alpharp, d->wrp)
    if (d->refractory) {
        if (fastpath && presig) DCLL_LAUNCH_C1T(true, 2);
        else if (fastpath) DCLL_LAUNCH_C1T(true, 1);
        else DCLL_LAUNCH_C1T(true, 0);
    } else {
        if (fastpath && presig) DCLL_LAUNCH_C1T(false, 2);
        else if (fastpath) DCLL_LAUNCH_C1T(false, 1);
        else DCLL_LAUNCH_C1T(false, 0);
    }
#undef DCLL_LAUNCH_C1T
    HIP_CHECK_LAUNCH("k_lif_seq_c1t");
    return DCLL_OK;
}

int dcll_launch_seq_c32t(const dcll_conv_desc *d, const uint32_t *spk_in, dcll_wsrc W, const float *b,
                         const float *tau4, float *eps0, float *eps1, float *arp, uint32_t *spk_out, float *pv_out,
                         float *v_out, float *state_scratch, int32_t T, int32_t B, hipStream_t st)
{
    const int out = (pv_out ? 1 : 0) | (v_out ? 2 : 0);
    const long nwg = (long)B * (d->h / 8) * (d->w / 32);
    if (nwg > 0x7fffffffL) return fail(DCLL_ERR_INVALID, "dcll_conv_lif_sequence: batch x tiles exceeds the grid limit");
    if ((long)d->h * d->w >= (1L << 26)) return fail(DCLL_ERR_UNSUPPORTED, "dcll_conv_lif_sequence: plane larger than 2^26 pixels");
    const long nstate = (long)B * 32 * d->h * d->w;
    int rc = snapshot_state(eps0, eps1, state_scratch, nstate, st, "dcll_conv_lif_sequence");
    if (rc) return rc;
    const float *eps0_in = state_scratch, *eps1_in = state_scratch + nstate;
#define DCLL_LAUNCH_C32T(R, O)                                                                                          \
    hipLaunchKernelGGL((k_lif_seq_c32t<R, O>), dim3((unsigned)nwg), dim3(512), 0, st, spk_in, W, b, tau4, eps0_in,      \
                       eps1_in, eps0, eps1, arp, spk_out, pv_out, v_out, T, B, d->h, d->w, d->alpharp, d->wrp)
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
