// dcll_seq_w3.hip — k_lif_seq_w3: the fused all-T layer kernel for the geometry of networks/radio_ml_conv_ref.yaml
// (BASELINE config 5): 64 output channels, kernel (1,3), padding (0,1), max-pool (1,2) — k_lif_seq_w3 for the six 64 -> 64
// layers (bit-packed spikes in, fp32-MFMA chains), k_lif_seq_w3f for the first layer (c_in = 1, input = one spike per step as
// a cell index: a register-only streaming kernel, below) — fifth translation unit of libdcll_hip.so.
//
// What the geometry gives: the kernel height is 1, so the rows of the plane are independent in every layer, and pooling
// pairs are neighbours in the row-major flattening.  A layer's input is therefore handled as a stream of 32-pixel TILES
// of the flattened (H x W) plane of each sample — a tile is 32/W rows when the layer has become narrow (W = 16 ... 2
// after the poolings), so the MFMA lanes stay full in all seven layers.  One 512-thread workgroup owns W3_NT = 8
// consecutive tiles (256 pixels: whole rows, of one sample or — narrow layers — of several) for all T steps:
//   - eps0 of its 64 x 256 trace elements in registers (thread = (input channel = lane, tile = wave): the 32 pixels of
//     ONE input spike word), eps1 in a pixel-major LDS image (layout below) with one shared zero position between rows
//     (the conv's horizontal padding);
//   - weights stationary in registers: wave (mt, g) holds the A fragments of output-channel tile mt (32 of the 64
//     channels) for the whole chain K = 64 x 3 = 192 — 96 VGPRs — and runs the complete pinned fmaf chain
//     (cp, kx, h: ci = 2cp + h) of its two pixel tiles 2g, 2g+1, one after the other: no hand-off between waves (unlike
//     k_lif_seq_c32d, whose K = 1568 had to be split over the waves);
//   - lane <-> pixel map of a tile: lanes 0..15 hold the EVEN pixels, lanes 16..31 the ODD ones, so a pooling pair sits
//     in lanes j and j + 16 = the same lane of two neighbouring 16-lane ROWS.  Two accumulator registers (channels c, c+1)
//     are pooled by ONE v_permlane16_swap_b32 (gfx950: swaps the odd rows of one register with the even rows of the
//     other) + ONE v_max_f32: afterwards rows 0 / 2 hold the pooled map of channel c (lane halves h = 0 / 1), rows 1 / 3
//     that of channel c+1 — 64 DISTINCT pooled values in one register, one store instruction, no LDS round trip (the
//     first form used a ds_swizzle per value and stored every pooled value from both lanes of its pair).  The pooled
//     spike is (pooled v > 0) — max(s_a, s_b) exactly — so one v_cmp on that register yields the four 16-bit pooled
//     half-words of the two channels; the half-words of a wave's two tiles are one 32-bit word of the next layer's input;
//   - per step: trace update (all threads) | barrier | 2 x 96 MFMAs + epilogue per wave | barrier; wide layers keep two
//     images (by step parity) and write step t+1's traces right behind the chains of step t: one barrier per step.
// The conv weights may arrive as int8 + one fp32 scale per output channel (dcll_wsrc): converted once, in the prologue.
// Outputs are the POOLED maps: spk_out (T,B,64,H*W/64) packed, pv_out (T,B,64,H,W/2); v_out (T,B,64,H,W) un-pooled.
// Same pinned arithmetic as every other path (include/dcll_hip.h): bit-identical to the per-step kernels / the C oracle.
#include "dcll_internal.h"

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// Stores of the epilogue as buffer stores (tile_rsrc, dcll_internal.h): a 128-bit descriptor per tile and step (base = the
// tile's first output row, in SGPRs) + 32-bit scalar offset (channel plane, walks by scalar adds) + 32-bit lane byte offset —
// no 64-bit address per store (left to the compiler, the flat form cost two scalar adds per store or, in some
// instantiations, a 64-bit VECTOR sum: v_mad_i64_i32 + v_lshl_add_u64).  Offsets stay below 2^31: 64 planes of < 2^24 pixels.
constexpr int W3_NT = 8;
#ifndef W3_EPG
#define W3_EPG 2                    // register pairs per epilogue group
#endif
#ifndef W3_EPI
#define W3_EPI 1                    // epilogue form: 0 pair by pair, 1 two pairs in lockstep (experiments/ablate_w3)
#endif
#ifndef W3_ORDER
#define W3_ORDER 0                  // 0: chain A, epilogue A, chain B, epilogue B; 1: both chains, then both epilogues
#endif
#ifndef W3_TRG
#define W3_TRG 8                    // trace elements per group (read - update - write)
#endif
// eps1 image in LDS, PIXEL-major: element (channel ci, padded pixel position q) at q * PST + ci, PST = 66 (c_in = 64) —
//   - q = p + (p >> log2 W) + 1 for pixel p of the workgroup's 256: one shared zero position between rows = the conv's
//     horizontal padding; W3_NPOS positions: 256 + 256 / W + 2 <= 386 (W = 2), 266 when W >= 32;
//   - the 64 lanes of a trace access are the 64 channels of one pixel: consecutive floats, conflict-free;
//   - a B fragment (lane = pixel, k lanes = channel pair 2cp + h, tap kx) is at lane base + kx * PST + 2 cp: all 96 k-steps of
//     a chain lie within 2 * 66 + 62 = 194 floats of ONE per-lane base — inside the 255-dword reach of ds_read2_b32's 8-bit
//     offsets, so the chain has no address arithmetic at all (channel-major images needed a new base per channel pair:
//     ~70 v_add_u32 per wave and step on the vector pipe the MFMAs run on); PST = 66 = 2 mod 64 puts the 32 pixels x 2
//     channels of a fragment read on 64 different banks.
constexpr int W3_NPOS = 386;

// experiments/ablate_w3 -DW3_STAMPS: s_memtime per wave and phase of workgroup 0, summed over the steps (not in the product)
#ifdef W3_STAMPS
__device__ unsigned long long w3_stamps[8][8];
// (summed in LDS with a return-less ds_add: a global read-modify-write per stamp costs more than the phases it measures)
#define W3_STAMP(ph)                                                                                                    \
    do {                                                                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                   \
        if (lane == 0) __hip_atomic_fetch_add(&lstamps[w][ph], now_ - stamp_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        stamp_ = now_;                                                                                                  \
    } while (0)
#else
#define W3_STAMP(ph)
#endif

// LW: log2 of the plane width W, as a template parameter — 5 stands for every W >= 32 (WIDE: a thread's 32 trace pixels lie
// in one row), 4 .. 1 are W = 16 .. 2, where pixel i of a thread sits i >> LW rows further on: with LW a compile-time
// constant that row term is part of the immediate offset of every trace access (round 3; as a runtime value it cost an
// address add per access — 64 VALU instructions per wave and step on the narrow layers).  The launcher maps the runtime width.
// OUT bit0: pooled pv, bit1: un-pooled v, bit2 (with bit0): the pooled map is written BEFORE the sigmoid (dcll_layer_opts
// pv_presigmoid: max-pooled v; the readout applies the sigmoid) — two quarter-rate transcendentals and two more VALU
// instructions per value off the pipe this kernel shares with its MFMAs
// FULL: every workgroup has its 8 tiles (B * HW / 32 is a multiple of 8 — always so when HW >= 256): the validity of a tile
// is a compile-time constant, and the ~100 scalar instructions per wave and step that predicate the stores on it (exec
// masks, branches around every store group) are gone from the time loop.
template <int CIN, bool REFRACTORY, int OUT, int LW, bool FULL>
__global__ __launch_bounds__(512, 2) void k_lif_seq_w3(const uint32_t *__restrict__ spk_in,
                                                     const dcll_wsrc W, const float *__restrict__ bias,
                                                     const float *__restrict__ tau4, float *__restrict__ eps0_g,
                                                     float *__restrict__ eps1_g, float *__restrict__ arp_g,
                                                     uint32_t *__restrict__ spk_out, float *__restrict__ pv_out,
                                                     float *__restrict__ v_out, int T, int B, int HW, int logW,
                                                     float alpharp, float wrp)
{
    static_assert(CIN == 64, "the 64 -> 64 layers; the first layer (c_in = 1) is k_lif_seq_w3f");
    constexpr int NK = 96;                              // MFMA k-steps of a chain
    constexpr int NE = 32;                              // trace elements per owning thread
    // DB (wide 64-channel layers = 3/4 of the network's work): a channel image needs only 256 + 256/32 + 1 floats, so TWO
    // images fit (2 x 74.8 KB), double-buffered by step parity: the traces of step t+1 are written into the other image
    // right after a wave's own chains of step t — one barrier per step instead of two, and the waves of a SIMD drift apart
    // (one in its trace / epilogue phase while the other issues MFMAs).
    constexpr bool WIDE = LW >= 5;
    constexpr bool DB = WIDE;
    constexpr int PST = 66;                             // floats per padded pixel position
    constexpr int IMG = (DB ? 266 : W3_NPOS) * PST + 8;
    __shared__ __attribute__((aligned(16))) float img[(DB ? 2 : 1) * IMG];
    __shared__ float sbias[64];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, jj = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = w & 1, g = w >> 1;
    const int NTS = HW >> 5;                             // tiles per sample
    const long ntot = (long)B * NTS;
    const long G0 = (long)blockIdx.x * W3_NT;            // first tile of this workgroup

    for (int i = tid; i < (DB ? 2 : 1) * IMG; i += 512) img[i] = 0.0f;
    if (tid < 64) sbias[tid] = bias[tid];

    // ---- trace ownership -------------------------------------------------------------------------------------------
    // thread (ci = lane, wq = wave) owns the 32 pixels of tile wq of channel ci (= one input word): a wave is one tile of all
    // 64 channels — validity, sample and tile index are wave-uniform, LDS accesses conflict-free
    const int ci_t = lane, wq = w;
    const long Gt = G0 + wq;
    const bool tvalid = FULL || Gt < ntot;
    const long bt = tvalid ? Gt / NTS : 0;
    const int mtile = tvalid ? (int)(Gt % NTS) : 0;
    const float ta = tau4[0 * CIN + ci_t], tm = tau4[1 * CIN + ci_t], tas = tau4[2 * CIN + ci_t], ts = tau4[3 * CIN + ci_t];
    float e0[NE];
    // LDS offset of my first pixel: block pixel p -> (p + (p >> logW) + 1) * PST + ci
    // (for my 32 consecutive pixels 32 wq + i the row term splits into a per-thread part and a wave-uniform one:
    //  (32 wq + i) >> logW == ((32 wq) >> logW) + (i >> logW), W a power of two)
    const int p0 = 32 * wq;
    const int loff0 = (p0 + 1 + (p0 >> logW)) * PST + ci_t;
    const long sbase = (bt * CIN + ci_t) * HW + 32L * mtile;             // my first state element
    __syncthreads();                                     // image zeroed
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        e0[i] = tvalid ? eps0_g[sbase + i] : 0.0f;
        if (tvalid) img[loff0 + (i + (WIDE ? 0 : (i >> LW))) * PST] = eps1_g[sbase + i];
    }

    // ---- weights of my output-channel tile, stationary: A[co = 32 mt + jj][k] ----------------------------------------
    float wf[NK];
    // k lanes = input-channel pair: step s = cp * 3 + kx -> W[co][2 cp + h][kx]
    if (W.q) {              // (one wave-uniform branch on the weight format around the whole load)
        const float sc = W.scale[32 * mt + jj];
#pragma unroll
        for (int s = 0; s < NK; ++s) wf[s] = (float)W.q[((long)(32 * mt + jj) * CIN + 2 * (s / 3) + h) * 3 + s % 3] * sc;
    } else {
#pragma unroll
        for (int s = 0; s < NK; ++s) wf[s] = W.f[((long)(32 * mt + jj) * CIN + 2 * (s / 3) + h) * 3 + s % 3];
    }
    // ---- my two pixel tiles (independent chains) ----------------------------------------------------------------------
    const int perm = jj < 16 ? 2 * jj : 2 * (jj - 16) + 1;          // lane -> pixel of the tile (even | odd)
    const long GA = G0 + 2 * g;
    const bool validA = FULL || GA < ntot, validB = FULL || GA + 1 < ntot;
    // (wave-uniform by construction; the 64-bit division leaves them in vector registers — moved to SGPRs here, so that
    //  every address term derived from them in the time loop is scalar)
    const long bA = uniform_long(validA ? GA / NTS : 0), bB = uniform_long(validB ? (GA + 1) / NTS : 0);
    const int mA = __builtin_amdgcn_readfirstlane(validA ? (int)(GA % NTS) : 0);
    const int mB = __builtin_amdgcn_readfirstlane(validB ? (int)((GA + 1) % NTS) : 0);
    const int pA = 64 * g + perm, pB = pA + 32;
    // B-fragment lane base: channel h of the pair
    const int baseA = h + (pA + (pA >> logW)) * PST, baseB = h + (pB + (pB >> logW)) * PST;
    float arpA[16], arpB[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
        arpA[r] = (REFRACTORY && validA) ? arp_g[(bA * 64 + co) * HW + 32L * mA + perm] : 0.0f;
        arpB[r] = (REFRACTORY && validB) ? arp_g[(bB * 64 + co) * HW + 32L * mB + perm] : 0.0f;
    }
    const int HW2 = HW >> 1, NW2 = NTS >> 1;             // pooled pixels / pooled words per channel plane
    // input of step 0
    uint32_t word = 0;
    if (tvalid) word = spk_in[(bt * CIN + ci_t) * NTS + mtile];
    const long in_step = (long)B * CIN * NTS;

    // traces of one step (dcll/pytorch_libdcll.py:493-494, every op rounded separately): eps1 read from image `src`,
    // written to image `dst` (float offsets; the same image when single-buffered); eight elements at a time (all 32 in
    // flight would hold 32 more registers on top of weights + accumulators + states)
    auto trace_step = [&](const uint32_t wd, const int src, const int dst) {
        int lb = loff0;
        const f32x2 ta2 = {ta, ta}, tm2 = {tm, tm}, tas2 = {tas, tas};
        // opaque per step: keeps the 32 element addresses (and their 32 wave-uniform row terms) out of loop-invariant
        // registers — they are two instructions each to recompute
        asm volatile("" : "+v"(lb));
#pragma unroll
        for (int i0 = 0; i0 < NE; i0 += W3_TRG) {
            float e1[W3_TRG];
#pragma unroll
            for (int i = i0; i < i0 + W3_TRG && i < NE; ++i) e1[i - i0] = img[src + lb + (i + (WIDE ? 0 : (i >> LW))) * PST];
            // x * tau_s for x in {0, 1} = the sign-extended input bit AND tau_s (v_bfe_i32 + v_and_b32: exact), then
            // the two trace lines of dcll/pytorch_libdcll.py:493-494, every op rounded separately — on PAIRS of
            // elements: v_pk_mul_f32 / v_pk_add_f32 are the same IEEE operations at two per lane and issue
#pragma unroll
            for (int i = i0; i < i0 + W3_TRG && i < NE; i += 2) {
                const f32x2 a = {__int_as_float(__builtin_amdgcn_sbfe((int)wd, i, 1) & __float_as_int(ts)),
                                 __int_as_float(__builtin_amdgcn_sbfe((int)wd, i + 1, 1) & __float_as_int(ts))};
                f32x2 p0 = {e0[i], e0[i + 1]}, p1 = {e1[i - i0], e1[i - i0 + 1]};
                const f32x2 bb = tas2 * p0;
                p0 = a + bb;
                const f32x2 cc = ta2 * p1;
                const f32x2 dd = p0 * tm2;
                p1 = cc + dd;
                e0[i] = p0[0], e0[i + 1] = p0[1];
                e1[i - i0] = p1[0], e1[i - i0 + 1] = p1[1];
            }
#pragma unroll
            for (int i = i0; i < i0 + W3_TRG && i < NE; ++i)
                if (tvalid) img[dst + lb + (i + (WIDE ? 0 : (i >> LW))) * PST] = e1[i - i0];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (DB) {               // step 0's traces in place in image 0, then the loop runs one barrier per step
        trace_step(word, 0, 0);
        lds_barrier();
    }
#ifdef W3_STAMPS
    __shared__ unsigned long long lstamps[8][8];
    if (tid < 64) lstamps[tid >> 3][tid & 7] = 0;
    __syncthreads();
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
    for (int t = 0; t < T; ++t) {
        const int cur = DB ? (t & 1) * IMG : 0;             // image the chains of this step read
        if (!DB) {
            trace_step(word, 0, 0);
            W3_STAMP(4);
            lds_barrier();
            W3_STAMP(6);
        }
        // next step's input: lands during the chains
        if (t + 1 < T && tvalid) word = spk_in[(long)(t + 1) * in_step + (bt * CIN + ci_t) * NTS + mtile];
        // ---- (2) + (3) per tile: the chain in the pinned order (cp, kx, h), then its epilogue.  One tile after the other
        //      (two interleaved chains + a joint epilogue need 32 accumulator registers and twice the temporaries: with
        //      96 weight and 64 state registers that spilled ~160 VGPRs) ----
        int vwA = 0, vwB = 0;
        auto chain = [&](const int base, f32x16 &acc, const int st0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = sbias[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h];
            // the 6 B fragments of channel pairs cp + 2, cp + 3 are fetched before the MFMAs of pairs cp, cp + 1
            float bq[2][6];
#pragma unroll
            for (int q = 0; q < 6; ++q) bq[0][q] = img[cur + base + (q / 3) * 2 + (q % 3) * PST];
#pragma unroll
            for (int c2 = 0; c2 < 16; ++c2) {
                if (c2 + 1 < 16) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) bq[(c2 + 1) & 1][q] = img[cur + base + (2 * (c2 + 1) + q / 3) * 2 + (q % 3) * PST];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[c2 * 6 + q], bq[c2 & 1][q], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            W3_STAMP(st0);
        };
        auto epilogue = [&](f32x16 &acc, float (&arp)[16], const bool valid, const long bb, const int mm, int &vw, const int st0) {
            // epilogue: refractory trace, threshold, (1,2) max-pool, sigmoid, pooled spike half-words — two accumulator
            // registers (channels cr, cr + 1) at a time.  Stores as wave-uniform base (per pair) + 32-bit lane offset.
            const long row = ((long)t * B + bb) * 64 + 32 * mt;
            // lane BYTE offsets.  Un-pooled v: value r of lane (h, jj) is channel cr(r) + 4 h, pixel perm(jj).  Pooled map
            // after the row swap: lane row rw = lane >> 4 holds channel cr + (rw & 1) + 4 (rw >> 1), pooled pixel lane & 15.
            const int rw = lane >> 4;
            unsigned lp = 4u * ((4 * (rw >> 1) + (rw & 1)) * HW2 + 16 * mm + (lane & 15)), lv = 4u * (4 * h * HW + 32 * mm + perm);
            // (lane offsets opaque per step: otherwise the per-value store addresses derived from them are hoisted out of the
            //  time loop into registers).  The scalar bases walk from channel to channel: cr = 0 2 8 10 16 18 24 26 — steps of
            //  2 and 6 channel planes, two scalar adds per store.
            int d2p = 8 * HW2, d6p = 24 * HW2, d1v = 4 * HW, d5v = 20 * HW, pp = 0, vp = 0;
            asm volatile("" : "+v"(lp), "+v"(lv), "+s"(d2p), "+s"(d6p), "+s"(d1v), "+s"(d5v), "+s"(pp), "+s"(vp));
            const auto prs = tile_rsrc(pv_out + row * HW2), vrs = tile_rsrc(v_out + row * HW);
#if W3_EPI == 1
            // Second form of the epilogue (round 5): on gfx950 a packed-fp32 result cannot be consumed by the NEXT vector
            // instruction (one wait state: the compiler fills it with s_nop 0), and the first read of an accumulator waits 15
            // states behind the chain's last MFMA.  Written pair by pair the update was a serial chain — 134 wait states per
            // wave and step sat in s_nop.  Here (i) alpharp * arp of all 16 values comes first, in place (independent of the
            // accumulators: 8 instructions inside the last MFMA's shadow), (ii) two register pairs advance in lockstep, so
            // every packed instruction has an independent neighbour.
            if (REFRACTORY) {
                const f32x2 al2 = {alpharp, alpharp};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 a2 = al2 * f32x2{arp[r], arp[r + 1]};
                    arp[r] = a2[0], arp[r + 1] = a2[1];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            static_for<0, 4>([&](auto gc) {
                constexpr int r0 = 4 * decltype(gc)::value;
                const f32x2 wrp2 = {wrp, wrp}, big = {0x1p127f, 0x1p127f};
                f32x2 v0 = {acc[r0], acc[r0 + 1]}, v1 = {acc[r0 + 2], acc[r0 + 3]};
                if (REFRACTORY) {
                    const f32x2 a0 = {arp[r0], arp[r0 + 1]}, a1 = {arp[r0 + 2], arp[r0 + 3]};
                    v0 = v0 + a0, v1 = v1 + a1;
                    f32x2 s0, s1;
                    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(s0) : "v"(v0), "s"(big));
                    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(s1) : "v"(v1), "s"(big));
                    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(s0) : "v"(s0), "s"(big));
                    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(s1) : "v"(s1), "s"(big));
                    const f32x2 n0 = __builtin_elementwise_fma(-s0, wrp2, a0), n1 = __builtin_elementwise_fma(-s1, wrp2, a1);
                    arp[r0] = n0[0], arp[r0 + 1] = n0[1], arp[r0 + 2] = n1[0], arp[r0 + 3] = n1[1];
                }
                float vx0 = v0[0], vy0 = v0[1], vx1 = v1[0], vy1 = v1[1];
                // pinned at the group: the new arp is not read before the next step (see the first form)
                asm volatile("" : "+v"(arp[r0]), "+v"(arp[r0 + 1]), "+v"(arp[r0 + 2]), "+v"(arp[r0 + 3]), "+v"(vx0), "+v"(vy0), "+v"(vx1), "+v"(vy1));
                if ((OUT & 2) && valid) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vx0), vrs, lv, vp, 0);
                    vp += d1v;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vy0), vrs, lv, vp, 0);
                    vp += d1v;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vx1), vrs, lv, vp, 0);
                    vp += d1v;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vy1), vrs, lv, vp, 0);
                    vp += d5v;
                }
                const u32x2 sw0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(vx0), __float_as_uint(vy0), false, false);
                const u32x2 sw1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(vx1), __float_as_uint(vy1), false, false);
                float pm0, pm1;
                asm("v_max_f32 %0, %1, %2" : "=v"(pm0) : "v"(__uint_as_float(sw0[0])), "v"(__uint_as_float(sw0[1])));
                asm("v_max_f32 %0, %1, %2" : "=v"(pm1) : "v"(__uint_as_float(sw1[0])), "v"(__uint_as_float(sw1[1])));
                const unsigned long long mk0 = __ballot(pm0 > 0.0f), mk1 = __ballot(pm1 > 0.0f);
                int &vwr = vw;
                asm("s_nop 1\n\tv_writelane_b32 %0, %1, %5\n\tv_writelane_b32 %0, %2, %6\n\t"
                    "v_writelane_b32 %0, %3, %7\n\tv_writelane_b32 %0, %4, %8"
                    : "+v"(vwr) : "s"((uint32_t)mk0), "s"((uint32_t)(mk0 >> 32)), "s"((uint32_t)mk1), "s"((uint32_t)(mk1 >> 32)),
                      "n"(r0 / 2), "n"(32 + r0 / 2), "n"(r0 / 2 + 1), "n"(33 + r0 / 2));
                if ((OUT & 1) && valid) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((OUT & 4) ? pm0 : sigmoidf_dev(pm0)), prs, lp, pp, 0);
                    pp += d2p;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((OUT & 4) ? pm1 : sigmoidf_dev(pm1)), prs, lp, pp, 0);
                    pp += d6p;
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#else
            static_for<0, 16 / (2 * W3_EPG)>([&](auto gc) {             // W3_EPG pairs at a time: keeps the temporaries few
                constexpr int r0 = 2 * W3_EPG * decltype(gc)::value;
                float pm[W3_EPG];
                unsigned long long mk[W3_EPG];
                int &vwr = vw;          // (named here: the asm operand of the inner generic lambda alone does not capture it)
                static_for<0, W3_EPG>([&](auto kc) {
                    constexpr int k = decltype(kc)::value, r = r0 + 2 * k, cr = (r & 3) + 8 * (r >> 2);
                    float vx = acc[r], vy = acc[r + 1];
                    if (REFRACTORY) {       // refractory() of dcll_internal.h on the register pair (packed fp32: same IEEE ops)
                        const f32x2 al2 = {alpharp, alpharp}, wrp2 = {wrp, wrp};
                        const f32x2 a2 = al2 * f32x2{arp[r], arp[r + 1]};
                        const f32x2 v2 = f32x2{acc[r], acc[r + 1]} + a2;
                        vx = v2[0], vy = v2[1];
                        // arp' = a - s * wrp: s = (v > 0) as 0.0 / 1.0 from two packed clamps (spike01_pk), s * wrp exact, so
                        // the fused form rounds once, like the reference's subtraction (:503): 5 packed instructions per pair
                        // (before: 3 + 2 v_cmp + 2 v_cndmask)
                        const f32x2 n2 = __builtin_elementwise_fma(-spike01_pk(v2), wrp2, a2);
                        arp[r] = n2[0], arp[r + 1] = n2[1];
                        // pinned here: the new arp is not read before the next step, and the compiler would otherwise sink
                        // all 32 updates (cmp, select, subtract) of both tiles to the end of the loop body — v and a of
                        // every value live across the second chain and the trace phase (64 registers)
                        asm volatile("" : "+v"(arp[r]), "+v"(arp[r + 1]), "+v"(vx), "+v"(vy));
                    }
                    if ((OUT & 2) && valid) {           // channels cr, cr + 1; then on to the next pair (cr + 2 | cr + 6)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vx), vrs, lv, vp, 0);
                        vp += d1v;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vy), vrs, lv, vp, 0);
                        vp += (r & 2) ? d5v : d1v;
                    }
                    // rows (16 lanes): vx = [even px | odd px | even px | odd px] of channels cr (h = 0), cr + 4 (h = 1);
                    // after the swap sw[0] = [vx.row0, vy.row0, vx.row2, vy.row2], sw[1] = [vx.row1, vy.row1, vx.row3, vy.row3]:
                    // their maximum is the pooled v of channel cr in rows 0 / 2 and of channel cr + 1 in rows 1 / 3
                    // (v is never NaN: plain v_max_f32, no canonicalisation)
                    const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(vx), __float_as_uint(vy), false, false);
                    asm("v_max_f32 %0, %1, %2" : "=v"(pm[k]) : "v"(__uint_as_float(sw[0])), "v"(__uint_as_float(sw[1])));
                    // pooled spike = max(s_a, s_b) = (pooled v > 0), exactly: bits 0..15 channel cr, 16..31 channel cr + 1
                    // (h = 0), 32..63 the same for h = 1 -> parked in lanes k' and 32 + k' of vw (k' = pair index)
                    mk[k] = __ballot(pm[k] > 0.0f);
                    if (W3_EPG != 2 && W3_EPG != 4) {
                        int &vwi = vwr;
                        // (wait states as the compiler places them around its own v_writelane, see k_lif_seq_c1)
                        asm("s_nop 1\n\tv_writelane_b32 %0, %1, %3\n\ts_nop 1\n\tv_writelane_b32 %0, %2, %4"
                            : "+v"(vwi) : "s"((uint32_t)mk[k]), "s"((uint32_t)(mk[k] >> 32)), "n"(r / 2), "n"(32 + r / 2));
                    }
                });
                if (W3_EPG == 2) {
                    // the four masks of the group in one block: s_nop 1 covers the two wait states a v_writelane needs behind
                    // the v_cmp that wrote its SGPR, whichever of the two compares the compiler placed last
                    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %5\n\tv_writelane_b32 %0, %2, %6\n\t"
                        "v_writelane_b32 %0, %3, %7\n\tv_writelane_b32 %0, %4, %8"
                        : "+v"(vwr) : "s"((uint32_t)mk[0]), "s"((uint32_t)(mk[0] >> 32)), "s"((uint32_t)mk[W3_EPG - 1]),
                          "s"((uint32_t)(mk[W3_EPG - 1] >> 32)), "n"(r0 / 2), "n"(32 + r0 / 2), "n"(r0 / 2 + 1), "n"(33 + r0 / 2));
                }
                if (W3_EPG == 4) {
                    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %9\n\tv_writelane_b32 %0, %2, %10\n\t"
                        "v_writelane_b32 %0, %3, %11\n\tv_writelane_b32 %0, %4, %12\n\t"
                        "v_writelane_b32 %0, %5, %13\n\tv_writelane_b32 %0, %6, %14\n\t"
                        "v_writelane_b32 %0, %7, %15\n\tv_writelane_b32 %0, %8, %16"
                        : "+v"(vwr) : "s"((uint32_t)mk[0]), "s"((uint32_t)(mk[0] >> 32)), "s"((uint32_t)mk[1 % W3_EPG]),
                          "s"((uint32_t)(mk[1 % W3_EPG] >> 32)), "s"((uint32_t)mk[2 % W3_EPG]), "s"((uint32_t)(mk[2 % W3_EPG] >> 32)),
                          "s"((uint32_t)mk[3 % W3_EPG]), "s"((uint32_t)(mk[3 % W3_EPG] >> 32)),
                          "n"(r0 / 2), "n"(32 + r0 / 2), "n"(r0 / 2 + 1), "n"(33 + r0 / 2), "n"(r0 / 2 + 2), "n"(34 + r0 / 2),
                          "n"(r0 / 2 + 3), "n"(35 + r0 / 2));
                }
                if ((OUT & 1) && valid) {
                    static_for<0, W3_EPG>([&](auto kc) {
                        constexpr int k = decltype(kc)::value, r = r0 + 2 * k, cr = (r & 3) + 8 * (r >> 2);
                        // sigmoid is monotone: sigmoid(max-pool(v)) == max-pool(sigmoid(v)) (up to its last ulp; pv is not
                        // bit-pinned) — half as many transcendentals; (OUT & 4): the readout applies it (pv_presigmoid)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((OUT & 4) ? pm[k] : sigmoidf_dev(pm[k])), prs, lp, pp, 0);
                        pp += (r & 2) ? d6p : d2p;
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#endif
            W3_STAMP(st0);
        };
        f32x16 accA, accB;
#if W3_ORDER == 0
        chain(baseA, accA, 0);
        epilogue(accA, arpA, validA, bA, mA, vwA, 1);
        chain(baseB, accB, 2);
        epilogue(accB, arpB, validB, bB, mB, vwB, 3);
#else
        // both chains first: a wave running a dependent MFMA chain owns its SIMD (the other wave gets one instruction per
        // MFMA), so a wave whose second chain waits for its own first epilogue leaves the matrix pipe idle
        chain(baseA, accA, 0);
        chain(baseB, accB, 2);
        epilogue(accA, arpA, validA, bA, mA, vwA, 1);
        epilogue(accB, arpB, validB, bB, mB, vwB, 3);
#endif
        // lanes q / 32 + q (q < 8) hold, for the channel pair cr(2q), cr(2q) + 1 (+ 4 h), the pooled half-words of my two
        // tiles as (channel cr | channel cr + 1 << 16): regrouped into the two 32-bit words (tile A | tile B << 16) of the
        // next layer's input
        if (spk_out && jj < 8 && validA) {
            const long row = ((long)t * B + bA) * 64 + 32 * mt;
            uint32_t *wp = spk_out + row * NW2 + (unsigned)(mA >> 1);
            const unsigned cq = (unsigned)(((2 * jj) & 3) + 8 * ((2 * jj) >> 2) + 4 * h);
            wp[cq * (unsigned)NW2] = ((uint32_t)vwA & 0xffffu) | ((uint32_t)vwB << 16);
            wp[(cq + 1) * (unsigned)NW2] = ((uint32_t)vwA >> 16) | ((uint32_t)vwB & 0xffff0000u);
        }
        if (DB && t + 1 < T) trace_step(word, cur, cur ^ IMG);         // step t+1's traces into the other image
        W3_STAMP(4);
        lds_barrier();
        W3_STAMP(5);
    }
#ifdef W3_STAMPS
    __syncthreads();
    if (blockIdx.x == 0 && tid < 64) w3_stamps[tid >> 3][tid & 7] += lstamps[tid >> 3][tid & 7];
#endif
    const int fin = DB ? ((T - 1) & 1) * IMG : 0;           // image holding eps1 of the last step
    // ---- state back to HBM ----
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        if (tvalid) {
            eps0_g[sbase + i] = e0[i];
            eps1_g[sbase + i] = img[fin + loff0 + (i + (WIDE ? 0 : (i >> LW))) * PST];
        }
    }
    if (REFRACTORY) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (validA) arp_g[(bA * 64 + co) * HW + 32L * mA + perm] = arpA[r];
            if (validB) arp_g[(bB * 64 + co) * HW + 32L * mB + perm] = arpB[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lif_seq_w3f — the FIRST layer of that network (c_in = 1 -> 64 channels, input = one spike per step as a cell index) as a
// register-only streaming kernel (round 5).  With one input channel the layer is not matrix work — three fmaf per output —
// but a 137 GB write stream of pooled membrane values at batch 4096, and the MFMA form above (k_lif_seq_w3<1>: 8 tiles per
// 512-thread workgroup, an LDS image, two barriers per step, a 16-value-per-lane epilogue behind every 2-MFMA "chain")
// spent its step in dependent epilogue code at two waves per SIMD: 4.6 TB/s.  Here nothing is shared between waves:
//   - a WAVE owns 128 consecutive pixels of the flattened plane of one sample (lane l: pixels 2l, 2l+1 = one pooling pair)
//     and NCH of the 64 output channels, for all T steps;
//   - the single input channel's traces are a function of the cell index alone, so every lane advances the traces of ITS
//     FOUR pixels 2l-1 .. 2l+2 itself (the two neighbours redundantly: 10 packed instructions per step against ~12 per
//     channel) — no LDS image, no cross-lane traffic, no barrier; a neighbour beyond the row end is the conv's zero padding;
//   - weights of the NCH channels in SGPRs (wave-uniform), bias and the 2 x NCH refractory traces in VGPRs;
//   - per channel: the pinned chain bias -> kx = 0, 1, 2 as three v_pk_fma_f32 on the pixel pair, the refractory update on
//     the pair (packed, same IEEE operations as refractory() of dcll_internal.h), the (1,2) max-pool INSIDE the lane (one
//     v_max_f32), the pooled spike = (pooled v > 0) as a v_cmp whose 64-bit lane mask IS the two packed output words of this
//     (channel, segment), and one 256-byte store of pooled pv (two whole lines per instruction);
//   - four or five waves per SIMD with independent instruction streams hide each other's latencies.
// Same arithmetic, same outputs and state layout as k_lif_seq_w3<1> (bit-identical un-pooled v, spikes, state).
constexpr int W3F_NCH = 8, W3F_WPB = 4;      // channels per wave, waves per workgroup (measured: below)
template <bool REFRACTORY, int OUT, int NCH>
__global__ __launch_bounds__(64 * W3F_WPB) void k_lif_seq_w3f(const int32_t *__restrict__ cells, const dcll_wsrc W,
                                                      const float *__restrict__ bias, const float *__restrict__ tau4,
                                                      const float *__restrict__ eps0_g, const float *__restrict__ eps1_g,
                                                      float *__restrict__ arp_g, uint32_t *__restrict__ spk_out,
                                                      float *__restrict__ pv_out, float *__restrict__ v_out, int T, int B, int HW,
                                                      int logW, long nitems, float alpharp, float wrp)
{
    constexpr int NCG = 64 / NCH;
    const int lane = threadIdx.x & 63;
    // XCD-aware order (workgroup ids go round-robin over the 8 XCDs): the workgroups of one sample run on ONE XCD, one after
    // the other — a sample-step's 256 KB of pooled map leave through one L2, close together in time
    long bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const long item = uniform_long(bid * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (item >= nitems) return;                          // (whole waves; the kernel has no barrier)
    const int NSEG = HW >> 7;                            // 128-pixel segments per sample plane
    const int sg = __builtin_amdgcn_readfirstlane((int)(item % NSEG));
    const int cg = __builtin_amdgcn_readfirstlane((int)((item / NSEG) % NCG));
    const long b = uniform_long(item / NSEG / NCG);
    const int ch0 = cg * NCH;

    // ---- my four pixels: L = 2l-1 (left neighbour), A = 2l, B = 2l+1 (the pooling pair I own), R = 2l+2 --------------
    const int pA = 128 * sg + 2 * lane, Wm = (1 << logW) - 1;
    const bool vL = (pA & Wm) != 0, vR = ((pA + 2) & Wm) != 0;          // inside my row (else: zero padding)
    // cell index that sets the input spike of each of them (a padding position never spikes)
    const int qL = vL ? pA - 1 : -7, qA = pA, qB = pA + 1, qR = vR ? pA + 2 : -7;
    const float ta = tau4[0], tm = tau4[1], tas = tau4[2], ts = tau4[3];
    const long sb = b * HW + pA;
    f32x2 e0LA = {vL ? eps0_g[sb - 1] : 0.0f, eps0_g[sb]}, e0BR = {eps0_g[sb + 1], vR ? eps0_g[sb + 2] : 0.0f};
    f32x2 e1LA = {vL ? eps1_g[sb - 1] : 0.0f, eps1_g[sb]}, e1BR = {eps1_g[sb + 1], vR ? eps1_g[sb + 2] : 0.0f};

    // ---- weights of my channels: wave-uniform, converted once (dcll_wsrc::at), kept in SGPRs; bias in VGPRs -----------
    float w0[NCH], w1[NCH], w2[NCH], bv[NCH];
    f32x2 arp[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int co = ch0 + c;
        w0[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(W.at(co * 3 + 0, co))));
        w1[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(W.at(co * 3 + 1, co))));
        w2[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(W.at(co * 3 + 2, co))));
        bv[c] = bias[co];
        arp[c] = REFRACTORY ? *(const f32x2 *)(arp_g + (b * 64 + co) * HW + pA) : f32x2{0.0f, 0.0f};
    }
    const int HW2 = HW >> 1, NW2 = HW >> 6;              // pooled pixels / pooled words per channel plane
    int cell = cells[b];
    const f32x2 ta2 = {ta, ta}, tm2 = {tm, tm}, tas2 = {tas, tas}, al2 = {alpharp, alpharp}, wrp2 = {wrp, wrp};

    for (int t = 0; t < T; ++t) {
        // traces of the step (dcll/pytorch_libdcll.py:493-494, every op rounded separately; x * tau_s for x in {0, 1})
        {
            const f32x2 xLA = {cell == qL ? ts : 0.0f, cell == qA ? ts : 0.0f}, xBR = {cell == qB ? ts : 0.0f, cell == qR ? ts : 0.0f};
            e0LA = xLA + tas2 * e0LA;
            e0BR = xBR + tas2 * e0BR;
            e1LA = ta2 * e1LA + e0LA * tm2;
            e1BR = ta2 * e1BR + e0BR * tm2;
        }
        if (t + 1 < T) cell = cells[(long)(t + 1) * B + b];             // (scalar load; lands during the channel loop)
        const f32x2 e1AB = {e1LA[1], e1BR[0]};
        const long row = ((long)t * B + b) * 64 + ch0;
        const auto prs = tile_rsrc(pv_out + row * HW2 + 64 * sg), vrs = tile_rsrc(v_out + row * HW + 128 * sg);
        uint32_t wlo = 0, whi = 0;
        static_for<0, NCH / 2>([&](auto cc) {
            unsigned long long mk[2];
            static_for<0, 2>([&](auto kc) {
                constexpr int k = decltype(kc)::value, c = 2 * decltype(cc)::value + k;
                // the pinned chain of the pixel pair: bias, then kx = 0, 1, 2 (oracle/dcll_oracle.c: fmaf(eps1, w, acc))
                f32x2 acc = {bv[c], bv[c]};
                acc = __builtin_elementwise_fma(e1LA, f32x2{w0[c], w0[c]}, acc);
                acc = __builtin_elementwise_fma(e1AB, f32x2{w1[c], w1[c]}, acc);
                acc = __builtin_elementwise_fma(e1BR, f32x2{w2[c], w2[c]}, acc);
                f32x2 v2 = acc;
                if (REFRACTORY) {                                       // refractory() of dcll_internal.h on the pair
                    const f32x2 a2 = al2 * arp[c];
                    v2 = acc + a2;
                    arp[c] = __builtin_elementwise_fma(-spike01_pk(v2), wrp2, a2);
                }
                if (OUT & 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v2), vrs, 8 * lane, 4 * c * HW, 0);
                const float pm = __builtin_fmaxf(v2[0], v2[1]);         // (1,2) max-pool of v; v is never NaN
                // pooled spike = max(s_a, s_b) = (pooled v > 0) exactly; bit l of the ballot = pooled pixel 64 sg + l
                mk[k] = __ballot(pm > 0.0f);
                if (OUT & 1)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((OUT & 4) ? pm : sigmoidf_dev(pm)), prs, 4 * lane, 4 * c * HW2, 0);
            });
            // the two words of each channel parked in lane c of (wlo, whi); s_nop 1 = the two wait states a v_writelane needs
            // behind the v_cmp that wrote its SGPR, whichever of the two compares the compiler placed last (as in k_lif_seq_w3)
            constexpr int c0 = 2 * decltype(cc)::value;
            uint32_t &wl = wlo, &wh = whi;      // (named here: an asm operand alone does not capture in a generic lambda)
            asm("s_nop 1\n\tv_writelane_b32 %0, %2, %6\n\tv_writelane_b32 %1, %3, %6\n\t"
                "v_writelane_b32 %0, %4, %7\n\tv_writelane_b32 %1, %5, %7"
                : "+v"(wl), "+v"(wh) : "s"((uint32_t)mk[0]), "s"((uint32_t)(mk[0] >> 32)), "s"((uint32_t)mk[1]),
                  "s"((uint32_t)(mk[1] >> 32)), "n"(c0), "n"(c0 + 1));
        });
        if (spk_out && lane < NCH) {                                    // lane c: the two words of channel ch0 + c
            uint32_t *wp = spk_out + (row + lane) * NW2 + 2 * sg;
            *(u32x2 *)wp = u32x2{wlo, whi};
        }
    }
    // ---- state back to HBM: arp by its owner.  The input traces are NOT written here: every wave of a sample (all channel
    // groups, and the neighbouring segments through their halo pixels) reads them at its start, the grid is far larger than
    // residency and workgroup start order is not defined — a wave that started after an in-place writer had retired would
    // take end-of-sequence traces as its initial state.  k_w3f_traces_advance, launched behind this kernel, advances them.
    if (REFRACTORY) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) *(f32x2 *)(arp_g + (b * 64 + ch0 + c) * HW + pA) = arp[c];
    }
}

// The first layer's input traces after T steps, in place: thread = one pixel pair of one sample; reads and writes ITS two
// elements only, and runs behind k_lif_seq_w3f on the same stream (which only reads them) — no launch ever reads trace
// memory another wave of the same launch writes.  The same packed IEEE operations, in the same order, as the trace block
// of k_lif_seq_w3f, so the state equals what the per-step kernels leave, bit for bit.  Cost: B * HW / 2 threads x T steps
// of four packed instructions — under 1 % of k_lif_seq_w3f's time.
__global__ __launch_bounds__(256) void k_w3f_traces_advance(const int32_t *__restrict__ cells, const float *__restrict__ tau4,
                                                             float *__restrict__ eps0_g, float *__restrict__ eps1_g, int T,
                                                             int B, int HW, long npair)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npair) return;
    const int hp = HW >> 1;                              // pairs per sample plane: a multiple of 64, so b is wave-uniform
    const long b = uniform_long(i / hp);
    const int qA = 2 * (int)(i - b * hp);
    const float ta = tau4[0], tm = tau4[1], tas = tau4[2], ts = tau4[3];
    const f32x2 ta2 = {ta, ta}, tm2 = {tm, tm}, tas2 = {tas, tas};
    f32x2 e0 = *(const f32x2 *)(eps0_g + 2 * i), e1 = *(const f32x2 *)(eps1_g + 2 * i);
    int cell = cells[b];
    for (int t = 0; t < T; ++t) {
        const f32x2 x = {cell == qA ? ts : 0.0f, cell == qA + 1 ? ts : 0.0f};
        e0 = x + tas2 * e0;
        e1 = ta2 * e1 + e0 * tm2;
        if (t + 1 < T) cell = cells[(long)(t + 1) * B + b];
    }
    *(f32x2 *)(eps0_g + 2 * i) = e0;
    *(f32x2 *)(eps1_g + 2 * i) = e1;
}

// geometry served: c_in 1 or 64, c_out 64, kernel (1,3), padding (0,1), pooling (1,2), w a power of two <= 256,
// h * w a multiple of 32 (64 when packed spikes are wanted), time constants per input channel
bool dcll_seq_w3_geometry(const dcll_conv_desc *d)
{
    const bool pow2 = d->w >= 2 && d->w <= 256 && (d->w & (d->w - 1)) == 0;
    return d->stride == 1 && d->dilation == 1 && d->groups == 1 &&
           (d->c_in == 1 || d->c_in == 64) && d->c_out == 64 && d->kh == 1 && d->kw == 3 && d->pad_h == 0 && d->pad_w == 1 &&
           d->pool_h == 1 && d->pool_w == 2 && pow2 && ((long)d->h * d->w) % 32 == 0 && (256 % d->w == 0 || d->w == 256);
}

int dcll_launch_seq_w3(const dcll_conv_desc *d, const uint32_t *spk_in, const int32_t *cells, dcll_wsrc W, const float *b,
                       const float *tau4, float *eps0, float *eps1, float *arp, uint32_t *spk_out, float *pv_out,
                       float *v_out, bool presigmoid, int32_t T, int32_t B, hipStream_t st)
{
    const long HW = (long)d->h * d->w;
    if (HW >= (1L << 24)) return fail(DCLL_ERR_UNSUPPORTED, "sequence kernel (1,3): plane larger than 2^24 pixels");
    if (spk_out && HW % 64 != 0)
        return fail(DCLL_ERR_UNSUPPORTED, "sequence kernel (1,3): packed pooled spikes need h * w % 64 == 0");
    if (d->c_in == 1 && HW % 128 != 0)
        return fail(DCLL_ERR_UNSUPPORTED, "sequence kernel (1,3), first layer: h * w must be a multiple of 128");
    int logW = 0;
    while ((1 << logW) < d->w) ++logW;
    const long ntile = (long)B * (HW / 32);
    const long nwg = (ntile + W3_NT - 1) / W3_NT;
    if (nwg > 0x7fffffffL) return fail(DCLL_ERR_INVALID, "sequence kernel (1,3): batch x tiles exceeds the grid limit");
    const int out = (pv_out ? 1 : 0) | (v_out ? 2 : 0) | ((pv_out && presigmoid) ? 4 : 0);
    if (d->c_in == 1) {     // first layer: k_lif_seq_w3f, one wave per (sample, 128-pixel segment, group of W3F_NCH channels)
        // (its state, un-pooled v and spike words move as 8-byte pairs)
        if ((((uintptr_t)eps0 | (uintptr_t)eps1 | (uintptr_t)arp | (uintptr_t)spk_out | (uintptr_t)v_out) & 7) != 0)
            return fail(DCLL_ERR_INVALID, "sequence kernel (1,3), first layer: state / spike / v pointers must be 8-byte aligned");
        const long nitems = (long)B * (HW / 128) * (64 / W3F_NCH);
        const long nblk = (nitems + W3F_WPB - 1) / W3F_WPB;
        if (nblk > 0x7fffffffL) return fail(DCLL_ERR_INVALID, "sequence kernel (1,3): batch x segments exceeds the grid limit");
#define DCLL_LAUNCH_W3F(R, O)                                                                                           \
    hipLaunchKernelGGL((k_lif_seq_w3f<R, O, W3F_NCH>), dim3((unsigned)nblk), dim3(64 * W3F_WPB), 0, st, cells, W, b, tau4, \
                       eps0, eps1, arp, spk_out, pv_out, v_out, T, B, (int)HW, logW, nitems, d->alpharp, d->wrp)
#define DCLL_LAUNCH_W3FO(R)                                                                                             \
    switch (out) {                                                                                                      \
    case 0: DCLL_LAUNCH_W3F(R, 0); break;                                                                               \
    case 1: DCLL_LAUNCH_W3F(R, 1); break;                                                                               \
    case 2: DCLL_LAUNCH_W3F(R, 2); break;                                                                               \
    case 3: DCLL_LAUNCH_W3F(R, 3); break;                                                                               \
    case 5: DCLL_LAUNCH_W3F(R, 5); break;                                                                               \
    default: DCLL_LAUNCH_W3F(R, 7); break;                                                                              \
    }
        if (d->refractory) { DCLL_LAUNCH_W3FO(true) } else { DCLL_LAUNCH_W3FO(false) }
#undef DCLL_LAUNCH_W3FO
#undef DCLL_LAUNCH_W3F
        HIP_CHECK_LAUNCH("k_lif_seq_w3f");
        const long npair = (long)B * (HW / 2);
        hipLaunchKernelGGL(k_w3f_traces_advance, dim3((unsigned)((npair + 255) / 256)), dim3(256), 0, st, cells, tau4, eps0, eps1,
                           T, B, (int)HW, npair);
        HIP_CHECK_LAUNCH("k_w3f_traces_advance");
        return DCLL_OK;
    }
#define DCLL_LAUNCH_W3W(R, O, WD)                                                                                       \
    do {                                                                                                                \
        if (ntile % W3_NT == 0)                                                                                         \
            hipLaunchKernelGGL((k_lif_seq_w3<64, R, O, WD, true>), dim3((unsigned)nwg), dim3(512), 0, st, spk_in, W, b, tau4, eps0, \
                               eps1, arp, spk_out, pv_out, v_out, T, B, (int)HW, logW, d->alpharp, d->wrp);             \
        else                                                                                                            \
            hipLaunchKernelGGL((k_lif_seq_w3<64, R, O, WD, false>), dim3((unsigned)nwg), dim3(512), 0, st, spk_in, W, b, tau4, eps0, \
                               eps1, arp, spk_out, pv_out, v_out, T, B, (int)HW, logW, d->alpharp, d->wrp);             \
    } while (0)
#define DCLL_LAUNCH_W3(R, O)                                                                                            \
    do {                                                                                                                \
        switch (logW) {                                                                                                 \
        case 1: DCLL_LAUNCH_W3W(R, O, 1); break;                                                                        \
        case 2: DCLL_LAUNCH_W3W(R, O, 2); break;                                                                        \
        case 3: DCLL_LAUNCH_W3W(R, O, 3); break;                                                                        \
        case 4: DCLL_LAUNCH_W3W(R, O, 4); break;                                                                        \
        default: DCLL_LAUNCH_W3W(R, O, 5); break;                                                                       \
        }                                                                                                               \
    } while (0)
#define DCLL_LAUNCH_W3O(R)                                                                                              \
    switch (out) {                                                                                                      \
    case 0: DCLL_LAUNCH_W3(R, 0); break;                                                                                \
    case 1: DCLL_LAUNCH_W3(R, 1); break;                                                                                \
    case 2: DCLL_LAUNCH_W3(R, 2); break;                                                                                \
    case 3: DCLL_LAUNCH_W3(R, 3); break;                                                                                \
    case 5: DCLL_LAUNCH_W3(R, 5); break;                                                                                \
    default: DCLL_LAUNCH_W3(R, 7); break;                                                                               \
    }
    if (d->refractory) { DCLL_LAUNCH_W3O(true) } else { DCLL_LAUNCH_W3O(false) }
#undef DCLL_LAUNCH_W3O
#undef DCLL_LAUNCH_W3
#undef DCLL_LAUNCH_W3W
    HIP_CHECK_LAUNCH("k_lif_seq_w3");
    return DCLL_OK;
}
