// dcll_readout.hip — k_readout_direct, the readout GEMM of the whole-sequence path (third translation unit of
// libdcll_hip.so):   out[r, n] = sum_k pv[r, k] * Wt[n, k] + bias[n]      (i2o / output_, dcll/pytorch_libdcll.py:602-606)
// for rows = T*B (hundreds of thousands), K = c_out*h*w = 8192 on the 16x16 plane, N = 24 readout rows (48 on the output
// layer, where i2o and output_ share ONE pass over pv).
//
// The operation is a stream over pv (32 KB per row, 17.2 GB per layer at B = 4096) against a 0.8 / 1.5 MB matrix that
// stays in L2: HBM-bound if the matrix pipe keeps up.  v_mfma_f32_16x16x4_f32 tiles make 48 readout rows 3 x 16 with no
// padding (the 32-column tiles of k_readout_v4<2> pad 48 to 64: 25 % wasted matrix time, 0.42 of the HBM roofline).
//
// No LDS at all: both MFMA operands are loaded straight from global memory in fragment layout.  The k index inside a
// 32-float chunk may be permuted freely as long as A and B agree, so lane (i = lane & 15, kq = lane >> 4) loads the 8
// consecutive floats k0 + 8 kq .. + 7 of its row (two dwordx4; the 4 kq-lanes of a row cover one 128-byte line) and
// MFMA e of the chunk contracts k in {k0 + 8 kq + e}.  A wave owns RT x 16 rows and all NT x 16 readout rows; latency is
// hidden by occupancy (<= 80 VGPRs: 6 waves per SIMD), or — WPE = 8: <= 64 VGPRs, no LDS — the kernel fits beside a
// resident k_lif_seq_c32d workgroup (which leaves 64 VGPRs per SIMD lane, 6 wave slots and no LDS) and fills the gaps
// of its matrix pipe from a second stream (networks/__init__.py, test_sequence(overlap_readout=True)).
// Summation order: per output one chain over the chunks in k order, inside a chunk e = 0..7, inside an MFMA kq = 0..3
// (not bit-pinned, like every readout: |err| <= 1e-4); independent of rows / launch splitting.
#include "dcll_internal.h"

typedef __attribute__((address_space(1))) const float gfloat;
typedef __attribute__((address_space(1))) const f32x4 gf32x4;

// F4 = float4 loads per fragment and chunk: 2 -> 32-float chunks (the 4 kq-lanes of a row cover a 128-byte line), 1 ->
// 16-float chunks (half the registers: the co-resident form).  The fragments of chunk c+1 are requested before the MFMAs of
// chunk c are issued (explicit register double buffer, order pinned with sched_barrier): a wave always has one chunk of
// loads in flight under RT*NT*4*F4 MFMAs (1536 cycles for <2,3,2>).
template <int RT, int NT, int F4, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
void k_readout_direct(const float *__restrict__ pv, const float *__restrict__ Wt, const float *__restrict__ bias,
                      float *__restrict__ out, long rows, int K, int N)
{
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long row0 = ((long)blockIdx.x * 4 + wave) * (16 * RT);
    if (row0 >= rows) return;                       // no barriers in this kernel: a wave may leave alone
    // addressing: wave-uniform base (SGPRs, advanced per chunk) + one 32-bit lane offset per fragment row
    const float *abase = pv + row0 * K;
    unsigned aoff[RT], boff[NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        long r = rt * 16 + i;
        if (row0 + r >= rows) r = rows - 1 - row0;  // clamped rows are computed and not stored
        aoff[rt] = (unsigned)(r * K) + 4 * F4 * kq;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int n = nt * 16 + i;
        if (n >= N) n = N - 1;                      // idem for padded readout rows
        boff[nt] = (unsigned)n * (unsigned)K + 4 * F4 * kq;
    }
    f32x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nchunk = K / (16 * F4);               // a multiple of 2 (K % 32 == 0 for F4 = 1, K % 64 == 0 for F4 = 2)
    f32x4 a[2][RT][F4], b[2][NT][F4];
    auto fetch = [&](int buf, int c) {
        // the chunk's bases stay in SGPRs (opaque to the optimiser, which would otherwise fold them into 64-bit
        // per-lane addresses: 2 VGPRs + a 64-bit add per fragment row and chunk)
        gfloat *ab = (gfloat *)abase + 16 * F4 * c, *wb = (gfloat *)Wt + 16 * F4 * c;      // global address space kept
        asm volatile("" : "+s"(ab), "+s"(wb));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int f = 0; f < F4; ++f)                // pv is read exactly once: streamed (nontemporal)
                a[buf][rt][f] = __builtin_nontemporal_load((gf32x4 *)(ab + aoff[rt]) + f);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int f = 0; f < F4; ++f) b[buf][nt][f] = ((gf32x4 *)(wb + boff[nt]))[f];
    };
    auto mfmas = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4 * F4; ++e)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][rt][e >> 2][e & 3], b[buf][nt][e >> 2][e & 3],
                                                                       acc[rt][nt], 0, 0, 0);
    };
    fetch(0, 0);
    for (int c = 0; c < nchunk; c += 2) {
        fetch(1, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < nchunk) fetch(0, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // D layout of the 16x16 tile: column (readout row n) = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + i;
        if (n < N) {
            const float bn = bias ? bias[n] : 0.0f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long r = row0 + rt * 16 + 4 * kq + q;
                    if (r < rows) out[r * N + n] = acc[rt][nt][q] + bn;
                }
        }
    }
}

// One row tile per wave, 16-float chunks: <= 64 VGPRs, no LDS.  Requires K % 64 == 0, N <= 48, 16-byte aligned pv / Wt
// rows (the caller checked).
int dcll_launch_readout_direct(const float *pv, const float *Wt, const float *bias, float *out, long rows, int K, int N,
                               hipStream_t st)
{
    const unsigned g1 = (unsigned)((rows + 63) / 64);
    if (N <= 16) hipLaunchKernelGGL((k_readout_direct<1, 1, 1, 8>), dim3(g1), dim3(256), 0, st, pv, Wt, bias, out, rows, K, N);
    else if (N <= 32) hipLaunchKernelGGL((k_readout_direct<1, 2, 1, 8>), dim3(g1), dim3(256), 0, st, pv, Wt, bias, out, rows, K, N);
    else hipLaunchKernelGGL((k_readout_direct<1, 3, 1, 8>), dim3(g1), dim3(256), 0, st, pv, Wt, bias, out, rows, K, N);
    HIP_CHECK_LAUNCH("k_readout_direct");
    return DCLL_OK;
}

// ------------------------------------------------------------------------------------------------------------
// k_readout_t16 — the standalone readout GEMM on 16x16x4 tiles: 128 rows x (NT x 16) readout rows per workgroup, K in
// chunks of 32 staged through LDS (the fragment-layout problem of k_readout_direct does not arise: global loads are
// plain coalesced float4 rows).  48 stacked readout rows are 3 x 16 — no padding (k_readout_v4<2> pads them to 64 and
// is matrix-bound: 5.3 ms at B = 4096 vs 3.5 ms for 24 rows).
// LDS rows have stride 36 floats: 16-byte aligned, so staging is one ds_write_b128 per float4 and a fragment is one
// ds_read_b128 — lane (i = lane & 15, kq = lane >> 4) reads k = 16 g + 4 kq .. + 3 of row i, its operands of the 4 MFMAs
// of k-group g (the k order inside a group is permuted identically for A and B) — and 36 = 4 mod 32 makes the eight
// lanes of a b128 phase cover the 32 banks exactly once.
// ------------------------------------------------------------------------------------------------------------
constexpr int T16_ROWS = 128, T16_KC = 32, T16_LD = 36;

// kslice > 0 (few rows: per-step calls, rows = batch): workgroup (x, y) handles only columns [y * kslice, (y+1) * kslice)
// of K and writes its partial tile, without the bias, to out + y * rows * N; k_readout_sum adds the slices in order.
// SIG (dcll_readout_act, DCLL_ACT_SIGMOID): the staged pv values are v (dcll_layer_opts pv_presigmoid) and the sigmoid is
// applied here, once per value, between the global load and the LDS write — the kernel is HBM-bound and has the vector
// slots that the layer kernels (which share their vector pipe with the fp32 MFMAs) do not.
// (the body as a device function of the workgroup's coordinates: k_readout_t16 calls it with its block index, k_readout_t16m —
//  several readouts in ONE launch — with the coordinates of the item its blockIdx.z selects)
template <int NT, bool SIG>
__device__ __forceinline__ void readout_t16_body(const float *__restrict__ pv, const float *__restrict__ Wt,
                                                  const float *__restrict__ bias, float *__restrict__ out, const long rows,
                                                  const int K, const int N, const int kslice, const unsigned bx, const unsigned by)
{
    constexpr int NBF = (NT * 16 * 8 + 255) / 256;          // float4 loads of the weight chunk per thread
    __shared__ __attribute__((aligned(16))) float sA[T16_ROWS * T16_LD];
    __shared__ __attribute__((aligned(16))) float sB[NBF * 32 * T16_LD];     // (rows >= NT * 16: staged zeros nobody reads)
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = (long)bx * T16_ROWS;
    const int kc = (tid & 7) * 4, rsub = tid >> 3;          // 8 threads x float4 = one 32-float K-chunk of a row
    f32x4 acc[2][NT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Operand fetch as BUFFER loads (round 4): descriptor = this workgroup's 128 rows of pv (resp. the readout matrix), its
    // record count = the rows that exist, lane offset = (row, 4 floats of the chunk), scalar offset = the chunk — a row past
    // the end is out of range and reads as zeros.  (Flat loads: a 64-bit address and an exec-mask branch per load — 21
    // vector instructions + 6 branches per 32-float chunk on the pipe the fp32 MFMAs of the chunk execute on.)
    const long nra = rows - row0 < T16_ROWS ? rows - row0 : T16_ROWS;
    const auto ars = __builtin_amdgcn_make_buffer_rsrc((void *)(pv + row0 * K), 0, (int)(nra * K * 4), 0x00020000);
    const auto brs = __builtin_amdgcn_make_buffer_rsrc((void *)Wt, 0, (N < NT * 16 ? N : NT * 16) * K * 4, 0x00020000);
    unsigned avo[4], bvo[NBF];
#pragma unroll
    for (int q = 0; q < 4; ++q) avo[q] = 4u * (unsigned)((rsub + 32 * q) * K + kc);
#pragma unroll
    for (int q = 0; q < NBF; ++q) bvo[q] = 4u * (unsigned)((rsub + 32 * q) * K + kc);
    f32x4 ra[4], rb[NBF];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ra[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avo[q], 4u * k0, 2));
#pragma unroll
        for (int q = 0; q < NBF; ++q) rb[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, bvo[q], 4u * k0, 0));
    };
    const int kbeg = kslice > 0 ? by * kslice : 0, kend = kslice > 0 ? kbeg + kslice : K;
    if (kslice > 0) { out += (long)by * rows * N; bias = nullptr; }
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += T16_KC) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (SIG) {      // (rows past the end stage sigmoid(0): they are computed and never stored)
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[q][e] = sigmoidf_dev(ra[q][e]);
            }
            *(f32x4 *)(sA + (rsub + 32 * q) * T16_LD + kc) = ra[q];
        }
#pragma unroll
        for (int q = 0; q < NBF; ++q) *(f32x4 *)(sB + (rsub + 32 * q) * T16_LD + kc) = rb[q];
        __syncthreads();
        if (k0 + T16_KC < kend) fetch(k0 + T16_KC);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 a[2], b[NT];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) a[rt] = *(const f32x4 *)(sA + (wave * 32 + rt * 16 + i) * T16_LD + 16 * g + 4 * kq);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = *(const f32x4 *)(sB + (nt * 16 + i) * T16_LD + 16 * g + 4 * kq);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][e], b[nt][e], acc[rt][nt], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout of a 16x16 tile: column (readout row n) = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + i;
        if (n < N) {
            const float bn = bias ? bias[n] : 0.0f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long r = row0 + wave * 32 + rt * 16 + 4 * kq + q;
                    if (r < rows) out[r * N + n] = acc[rt][nt][q] + bn;
                }
        }
    }
}

template <int NT, bool SIG>
__global__ __launch_bounds__(256) void k_readout_t16(const float *__restrict__ pv, const float *__restrict__ Wt,
                                                      const float *__restrict__ bias, float *__restrict__ out,
                                                      long rows, int K, int N, int kslice)
{
    readout_t16_body<NT, SIG>(pv, Wt, bias, out, rows, K, N, kslice, blockIdx.x, blockIdx.y);
}
// k_readout_t16m — the split-K passes of SEVERAL per-step readouts (the slices of one network timestep: rows = batch) in one
// launch: blockIdx.z selects the item, whose own grid (row tiles x K slices) is a corner of the launch grid.  Each of these
// passes alone is one workgroup generation of 256 workgroups whose time is its latency (8.8 us for 16.8 MB at 512 rows, the
// same at 128): three of them side by side cost little more than one.  Arithmetic per item = k_readout_t16<NT> on it.
constexpr int T16M_MAX = 8;
struct readout_t16_items {
    const float *pv[T16M_MAX];
    const float *Wt[T16M_MAX];
    float *out[T16M_MAX];
    long rows[T16M_MAX];
    int K[T16M_MAX], N[T16M_MAX], kslice[T16M_MAX];
};
template <int NT>
__global__ __launch_bounds__(256) void k_readout_t16m(const readout_t16_items it)
{
    const int z = blockIdx.z;
    const long rows = it.rows[z];
    const int K = it.K[z], ks = it.kslice[z];
    if ((long)blockIdx.x * T16_ROWS >= rows || (int)blockIdx.y * ks >= K) return;        // (whole workgroups: before any barrier)
    // an item narrower than the launch's widest runs the instantiation of its own width (wave-uniform branch; its column
    // tiles, its LDS staging — not the widest item's with spare tiles computed on zeros)
    const int N = it.N[z];
    if (NT >= 3 && N <= 32)
        readout_t16_body<(NT >= 3 ? 2 : NT), false>(it.pv[z], it.Wt[z], nullptr, it.out[z], rows, K, N, ks, blockIdx.x, blockIdx.y);
    else
        readout_t16_body<NT, false>(it.pv[z], it.Wt[z], nullptr, it.out[z], rows, K, N, ks, blockIdx.x, blockIdx.y);
}
// Requires K % 32 == 0, N <= 64, 16-byte aligned pv / Wt rows (the caller checked).  kslice > 0: split-K launch (K %
// kslice == 0, kslice % 32 == 0), `out` = the partial tiles (K / kslice) x rows x N.
int dcll_launch_readout_t16(const float *pv, const float *Wt, const float *bias, float *out, long rows, int K, int N,
                            int kslice, hipStream_t st, int act)
{
    if (K >= (1 << 22)) return fail(DCLL_ERR_UNSUPPORTED, "readout GEMM: K above 4 M features per row");
    const dim3 g((unsigned)((rows + T16_ROWS - 1) / T16_ROWS), kslice > 0 ? K / kslice : 1);
#define DCLL_T16(NT_)                                                                                                   \
    do {                                                                                                                \
        if (act == DCLL_ACT_SIGMOID)                                                                                    \
            hipLaunchKernelGGL((k_readout_t16<NT_, true>), g, dim3(256), 0, st, pv, Wt, bias, out, rows, K, N, kslice);  \
        else                                                                                                            \
            hipLaunchKernelGGL((k_readout_t16<NT_, false>), g, dim3(256), 0, st, pv, Wt, bias, out, rows, K, N, kslice); \
    } while (0)
    if (N <= 16) DCLL_T16(1);
    else if (N <= 32) DCLL_T16(2);
    else if (N <= 48) DCLL_T16(3);
    else DCLL_T16(4);
#undef DCLL_T16
    HIP_CHECK_LAUNCH("k_readout_t16");
    return DCLL_OK;
}
// Split-K passes of n <= T16M_MAX readouts in one launch (dcll_step_readouts_multi): item i writes its partial tiles
// (K[i] / kslice[i]) x rows[i] x N[i] to out[i].  Same requirements per item as dcll_launch_readout_t16 with kslice > 0.
int dcll_launch_readout_t16_multi(const float *const *pv, const float *const *Wt, float *const *out, const long *rows,
                                  const int *K, const int *N, const int *kslice, int n, hipStream_t st)
{
    if (n < 1 || n > T16M_MAX) return fail(DCLL_ERR_INVALID, "readout GEMM (multi): 1 .. 8 items");
    readout_t16_items it;
    unsigned gx = 1, gy = 1;
    int nmax = 0;
    for (int i = 0; i < n; ++i) {
        if (K[i] >= (1 << 22)) return fail(DCLL_ERR_UNSUPPORTED, "readout GEMM: K above 4 M features per row");
        if (kslice[i] < 1 || K[i] % kslice[i] != 0) return fail(DCLL_ERR_INVALID, "readout GEMM (multi): split-K items only");
        it.pv[i] = pv[i], it.Wt[i] = Wt[i], it.out[i] = out[i], it.rows[i] = rows[i], it.K[i] = K[i], it.N[i] = N[i];
        it.kslice[i] = kslice[i];
        gx = max(gx, (unsigned)((rows[i] + T16_ROWS - 1) / T16_ROWS));
        gy = max(gy, (unsigned)(K[i] / kslice[i]));
        nmax = max(nmax, N[i]);
    }
    for (int i = n; i < T16M_MAX; ++i) it.pv[i] = it.Wt[i] = nullptr, it.out[i] = nullptr, it.rows[i] = 0, it.K[i] = it.N[i] = 0, it.kslice[i] = 1;
    const dim3 g(gx, gy, (unsigned)n);
    // one kernel for the launch, instantiated for the widest item (narrower items branch to their own width inside it)
    if (nmax <= 16) hipLaunchKernelGGL(k_readout_t16m<1>, g, dim3(256), 0, st, it);
    else if (nmax <= 32) hipLaunchKernelGGL(k_readout_t16m<2>, g, dim3(256), 0, st, it);
    else if (nmax <= 48) hipLaunchKernelGGL(k_readout_t16m<3>, g, dim3(256), 0, st, it);
    else hipLaunchKernelGGL(k_readout_t16m<4>, g, dim3(256), 0, st, it);
    HIP_CHECK_LAUNCH("k_readout_t16m");
    return DCLL_OK;
}
