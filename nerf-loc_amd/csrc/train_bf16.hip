// bf16 training kernels: input-gradient chain and weight gradients on v_mfma_f32_32x32x16_bf16 (float32
// accumulation, float32 master weights and gradients).  What autograd computes in loss.backward() (run_nerf.py:785)
// for NeRF.forward (run_nerf_helpers.py:96-119), with bf16-rounded operands.
//
//   forward            mlp_bf16_kernel<.., TAPE> (mlp.hip): the inference kernel, which also stores every layer input
//                      as the packed operand quads it already holds (train_bf16.h)
//   input gradients    mlp_dgrad_bf16_kernel: the same register-resident layer loop on the transposed weights; the
//                      epilogue multiplies the packed gradient by the ReLU bits the forward left (one word per
//                      tile and lane) and stores it in the same piece layout
//   weight gradients   wgrad16_kernel: dW[out, in] = sum_m G[m, out] X[m, in] with the sample as the MFMA k index.
//                      Tapes are sample-on-lane; the k index must be in the register.  Pieces are copied to LDS as
//                      they are (LDS-DMA) and read back with ds_read_b64_tr_b16, which hands each lane 4 samples of
//                      one feature.  HBM-bound: 1 KiB per sample and layer against 131 kFLOP.
#include "mlp_bf16.h"
#include "train_bf16.h"
#include "wgrad.h"

namespace nerf {

using namespace tape16;

// ------------------------------------------------------------------------------------------
// input-gradient chain
// ------------------------------------------------------------------------------------------
struct EpiDgrad {
    static constexpr bool ZERO_BIAS = true;
    unsigned bits[72];           // ReLU bit words (train_bf16.h) of every masked tile of the pass, loaded at its start:
                                 // no vector-memory load (and so no wait on the tape stores) inside the layer loop
    unsigned ones;               // 0x00010001
    __device__ __forceinline__ void prefetch(const EpiTile&, int) {}
    // Bit 8 ct + 4 half + qd (+ 16) says whether the lower (upper) unit of this packed dword was active in the
    // forward pass; the 16-bit product keeps or clears each half of the packed gradient.  Written as an instruction:
    // left to the compiler the multiply becomes compare / select chains.
    __device__ __forceinline__ unsigned post(unsigned v, const EpiTile& t, int, int half, int ct, int qd) const {
        if (!t.has_act) return v;
        const unsigned m = (bits[t.idx] >> (8 * ct + 4 * half + qd)) & ones;
        unsigned r;
        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(v), "v"(m));
        return r;
    }
    __device__ __forceinline__ void store(const u32x4& v, const EpiTile& t, int half, int ct) const {
        if (t.has_out) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(t.out + (half * 2 + ct) * 1024));
    }
};

struct Dgrad16Args {
    const unsigned char* packed;   // transposed bf16 weight stream (make_backward_desc(.., bf16))
    int n_stages;
    long long M;
    const float* d_raw;            // [M, OUT_CH]
    const unsigned char* tape;
    unsigned char* gtape;
};

template <bool VD, int OUT_CH>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_dgrad_bf16_kernel(Dgrad16Args a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    Stream st;
    {
        MlpArgs ma{};
        ma.packed = a.packed; ma.n_stages = a.n_stages; ma.n_bias_tiles = 0;
        pipe_start(st.p, ma, lds);
    }
    st.slot0 = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const long long n_tiles = (a.M + 255) / 256;
    const unsigned Mu = (unsigned)a.M;
    constexpr int NFRAGS = VD ? FRAGS_BWD_BF16_VD : FRAGS_BWD_BF16_NOVD;
    constexpr int NPAD = pad_to_stage(NFRAGS);
#pragma unroll
    for (int k = 0; k < PF; ++k) st.fetch(k);

    // output gradients of the samples, fetched one pass ahead; samples beyond M get zeros, which the chain (no
    // biases) carries through every layer: their rows of the gradient tape are zero
    float dr_next[NCT][OUT_CH];
    auto fetch_draw = [&](long long tile, float (&dr)[NCT][OUT_CH]) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = (unsigned)tile * 256u + wave * 64 + ct * 32 + c;
            const float* p = a.d_raw + (size_t)(m < Mu ? m : Mu - 1) * OUT_CH;
#pragma unroll
            for (int k = 0; k < OUT_CH; ++k) dr[ct][k] = m < Mu ? p[k] : 0.0f;
        }
    };
    if ((long long)blockIdx.x < n_tiles) fetch_draw(blockIdx.x, dr_next);

    EpiDgrad epi;
    epi.ones = 0x00010001u;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        float dr[NCT][OUT_CH];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int k = 0; k < OUT_CH; ++k) dr[ct][k] = dr_next[ct][k];
        if (tile + gridDim.x < n_tiles) fetch_draw(tile + gridDim.x, dr_next);

        const unsigned long long blk = (unsigned long long)(tile * 4 + wave);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)blk), bhi = __builtin_amdgcn_readfirstlane((unsigned)(blk >> 32));
        const unsigned long long blk_u = ((unsigned long long)bhi << 32) | blo;
        {   // ReLU bit words of the pass: 32 bytes per lane and layer; the layer used first is loaded last, so the one
            // wait in front of its first use covers them all
            const unsigned char* bw = a.tape + blk_u * (total(VD) * PIECE_BYTES) + lane * 32;
#pragma unroll
            for (int i = 0; i < (VD ? 9 : 8); ++i) {
                const u32x4* p = reinterpret_cast<const u32x4*>(bw + bits(VD, i) * PIECE_BYTES);
                const u32x4 v0 = __builtin_nontemporal_load(p);
#pragma unroll
                for (int k = 0; k < 4; ++k) epi.bits[8 * i + k] = v0[k];
                if (i < 8) {                                   // views_linears.0 has four tiles
                    const u32x4 v1 = __builtin_nontemporal_load(p + 1);
#pragma unroll
                    for (int k = 0; k < 4; ++k) epi.bits[8 * i + 4 + k] = v1[k];
                }
            }
        }
        unsigned char* gw = a.gtape + blk_u * (gtotal(VD) * PIECE_BYTES) + lane * 16;

        // head gradients as one-k-step operands: slot (h, j) <-> row 8 (j >> 2) + 4 h + (j & 3); stored padded to
        // two pieces for the heads' weight gradients
        u32x4 g0[1][NCT], g1[1][NCT];
        const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            if (VD) {
                g0[0][ct] = zero4; g1[0][ct] = zero4;
                if (h == 0) {
                    g0[0][ct][0] = pack_pair(dr[ct][0], dr[ct][1], false);
                    g0[0][ct][1] = pack_pair(dr[ct][2], 0.0f, false);
                    g1[0][ct][0] = pack_pair(dr[ct][3], 0.0f, false);
                }
                *reinterpret_cast<u32x4*>(gw + DRGB * PIECE_BYTES + ct * 1024) = g0[0][ct];
                *reinterpret_cast<u32x4*>(gw + (DRGB + 1) * PIECE_BYTES + ct * 1024) = zero4;
                *reinterpret_cast<u32x4*>(gw + DSIG * PIECE_BYTES + ct * 1024) = g1[0][ct];
                *reinterpret_cast<u32x4*>(gw + (DSIG + 1) * PIECE_BYTES + ct * 1024) = zero4;
            } else {
                g0[0][ct] = zero4; g1[0][ct] = zero4;
                if (h == 0) {
                    g0[0][ct][0] = pack_pair(dr[ct][0], dr[ct][1], false);
                    g0[0][ct][1] = pack_pair(dr[ct][2], dr[ct][3], false);
                } else if (OUT_CH > 4) {
                    g0[0][ct][0] = pack_pair(dr[ct][OUT_CH > 4 ? 4 : 0], 0.0f, false);
                }
                *reinterpret_cast<u32x4*>(gw + DOUT * PIECE_BYTES + ct * 1024) = g0[0][ct];
                *reinterpret_cast<u32x4*>(gw + (DOUT + 1) * PIECE_BYTES + ct * 1024) = zero4;
            }
        }

        u32x4 A[16][NCT], B[16][NCT];
        f32x16 accp[2][NCT];
        f32x16 bias = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        EpiTile carry{nullptr, nullptr, false, false, 0};
        int pos = 0, bt = 0;
#define NERF_D(NT, KPE, KH, KVW, hin, pdst, hasp, hout, layer, masked, out_piece)                                  \
    layer_bf16<NT, KPE, KH, KVW>(st, pos, bt, 1, h, bias, g0, hin, g1, accp, pdst, false, hasp, hout, false, epi, carry, \
                                 nullptr, masked, gw + (out_piece) * PIECE_BYTES, true, 8 * (layer))
        if (VD) {
            NERF_D(4, 1, 0, 0, A, A, false, B, 8, true, DZV);             // rgb_linear^T: d(rgb) -> d(view hidden)
            NERF_D(8, 0, 8, 0, B, B + 6, true, A, 0, false, DFEAT);        // views_linears.0^T (feature columns)
            NERF_D(8, 0, 16, 1, A, A + 14, true, B, 7, true, dz(7)); // feature_linear^T + alpha_linear^T
        } else {
            NERF_D(8, 1, 0, 0, A, A, false, B, 7, true, dz(7));     // output_linear^T
        }
        NERF_D(8, 0, 16, 0, B, B + 14, true, A, 6, true, dz(6));    // pts_linears.7^T
        NERF_D(8, 0, 16, 0, A, A + 14, true, B, 5, true, dz(5));    // .6
        NERF_D(8, 0, 16, 0, B, B + 14, true, A, 4, true, dz(4));    // .5 (hidden columns)
        NERF_D(8, 0, 16, 0, A, A + 14, true, B, 3, true, dz(3));    // .4
        NERF_D(8, 0, 16, 0, B, B + 14, true, A, 2, true, dz(2));    // .3
        NERF_D(8, 0, 16, 0, A, A + 14, true, B, 1, true, dz(1));    // .2
        NERF_D(8, 0, 16, 0, B, B + 14, true, A, 0, true, dz(0));    // .1
#undef NERF_D
        // the last tile has no successor to hide its epilogue under
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) epi_slice(accp[(bt - 1) & 1], A + 14, false, sl, epi, carry, (bt - 1) & 1);
        // zero fragments that pad the stream to a whole stage
#pragma unroll
        for (int g = NFRAGS; g < NPAD; ++g) {
            st.tick(g);
            st.fetch(g + PF);
        }
        st.end_pass(NPAD);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// weight gradients
// ------------------------------------------------------------------------------------------
struct Wg16Args {
    const unsigned char* G; unsigned g_blk, g_off;   // tape, bytes per (T, w) block, byte offset of the layer's first piece
    const unsigned char* X; unsigned x_blk, x_off;
    int n_chunks;                // 32-sample chunks (one column tile of one wave): padded samples / 32
    int per_wg;                  // chunks per workgroup
    float* part;                 // [grid][256][32 CT]
    float* bpart;                // [grid][256] column sums of G, or nullptr
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// One 1 KiB piece half (the chunk's column tile) -> LDS, lane-linear.  Streamed (nt): the tapes are read once; without
// the hint the launches that miss the last-level cache take 158 instead of 135 us.
__device__ __forceinline__ void dma_1k(unsigned long long gsrc /*uniform*/, unsigned lds_dst /*uniform*/, unsigned lane16) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane16), "s"(lds_dst), "s"(gsrc)
        : "memory");
}

// PG pieces of G (16 output features each), 2 CT pieces of X.  RT 32-row tiles per wave; waves beyond the rows idle
// in the MFMA part but take their share of the copies.
template <int PG, int CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void wgrad16_kernel(Wg16Args a) {
    constexpr int RT = PG >= 16 ? 2 : 1;
    constexpr int NW = PG / (2 * RT);                 // waves with rows
    constexpr int NP = PG + 2 * CT;                   // pieces per stage
    constexpr int NLD = (NP + 3) / 4;                 // copies per wave and stage
    constexpr int SB = NP * 1024;                     // stage bytes
    constexpr int NST = 4;                            // stages in LDS: 3 in flight ahead of the one being read
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * SB + 4 * 1024];   // + a dump KiB per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)lds;
    const int q0 = blockIdx.x * a.per_wg;
    const int q1 = q0 + a.per_wg < a.n_chunks ? q0 + a.per_wg : a.n_chunks;
    const int n = q1 - q0;

    // LDS image of a piece: the quad of lane (h, c) sits in slot 32 h + (c ^ 4 h ^ 8 (s & 1)) (s = piece index), which
    // makes the transposed reads below conflict-free.  LDS-DMA writes lane-linear, so the permutation is applied to
    // the source address: two lane offsets, for even and for odd pieces.
    const unsigned src_even = ((lane & 32) + ((lane & 31) ^ ((lane >> 5) * 4))) * 16;
    const unsigned src_odd = src_even ^ (8 * 16);
    auto issue = [&](int j) {
        const int q = q0 + j < q1 ? q0 + j : q1 - 1;
        const unsigned slot = (unsigned)j % NST;
        const unsigned long long gb = (unsigned long long)(q >> 1) * a.g_blk + a.g_off + (q & 1) * 1024;
        const unsigned long long xb = (unsigned long long)(q >> 1) * a.x_blk + a.x_off + (q & 1) * 1024;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = wave + 4 * k;               // piece of the stage; beyond NP: a copy into the dump area
            const bool real = i < NP;
            const int ii = real ? i : NP - 1;
            const unsigned long long src = ii < PG ? reinterpret_cast<unsigned long long>(a.G) + gb + (unsigned long long)ii * PIECE_BYTES
                                                   : reinterpret_cast<unsigned long long>(a.X) + xb + (unsigned long long)(ii - PG) * PIECE_BYTES;
            const unsigned dst = real ? lds0 + slot * SB + ii * 1024 : lds0 + NST * SB + wave * 1024;
            const unsigned slo = __builtin_amdgcn_readfirstlane((unsigned)src), shi = __builtin_amdgcn_readfirstlane((unsigned)(src >> 32));
            // PG is even, so the parity of a piece within the stage is the parity within its layer
            dma_1k(((unsigned long long)shi << 32) | slo, __builtin_amdgcn_readfirstlane(dst), (ii & 1) ? src_odd : src_even);
        }
    };

    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][ct][e] = 0.0f;
    f32x16 bacc[RT];             // column sums of G (bias gradients): G^T times a matrix of ones
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) bacc[i][e] = 0.0f;
    const u32x4 ones4 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    // ds_read_b64_tr_b16: in every 16-lane group, lane 4 q + p addresses row q (a sample), columns 4 p .. 4 p + 3
    // (4 consecutive features = 8 bytes of a lane quad), and lane i receives column i of the four rows.  Group g
    // covers features 16 (g & 1) .. + 15 of a 32-feature tile (piece 2 ft + (g & 1)) for the MFMA k half g >> 1.
    // Feature 4 p + e of a piece sits in lane half p & 1, quad half p >> 1, element e.
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3, hh = g >> 1;
    // sample c = 16 kk + 8 hh + 4 rd + qq of lane half pp & 1 in piece 2 ft + (g & 1): slot 32 (pp & 1) + (c ^ swizzle)
    const unsigned lane_rd0 = (g & 1) * 1024 + ((pp & 1) * 32 + ((8 * hh + qq) ^ (4 * (pp & 1)) ^ (8 * (g & 1)))) * 16 + (pp >> 1) * 8;
    const unsigned lane_rd1 = lane_rd0 ^ (4 * 16);     // rd = 1: sample + 4
    auto tr_read = [&](unsigned base, int off) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(size_t)(base + off));
    };
    const bool rows = wave < NW;

    issue(0);
    issue(1);
    issue(2);
    for (int i = 0; i < n; ++i) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory"); // this wave's copies of stage i have landed
        __builtin_amdgcn_s_barrier();                                  // everyone's have, and stage i - 1 is read out
        asm volatile("" ::: "memory");
        issue(i + 3);
        if (rows) {
            const unsigned sbase = lds0 + ((unsigned)i % NST) * SB + lane_rd0;
            const unsigned sbase1 = lds0 + ((unsigned)i % NST) * SB + lane_rd1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {                            // 16 samples per MFMA
                // wide X (10 tiles: 320 + 32 accumulators) is taken in two halves to keep the operand registers few
                constexpr int CG = CT > 8 ? (CT + 1) / 2 : CT;
                bf16x8 ga[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int ft = wave * RT + rt;
                    const s16x4 lo = tr_read(sbase, ft * 2048 + kk * 256), hi = tr_read(sbase1, ft * 2048 + kk * 256);
                    const short __attribute__((ext_vector_type(8))) v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    ga[rt] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
                for (int c0 = 0; c0 < CT; c0 += CG) {
                    bf16x8 xb[CG];
#pragma unroll
                    for (int j = 0; j < CG; ++j) {
                        const int ct = c0 + j < CT ? c0 + j : CT - 1;
                        const s16x4 lo = tr_read(sbase, PG * 1024 + ct * 2048 + kk * 256), hi = tr_read(sbase1, PG * 1024 + ct * 2048 + kk * 256);
                        const short __attribute__((ext_vector_type(8))) v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        xb[j] = __builtin_bit_cast(bf16x8, v);
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int j = 0; j < CG; ++j)
                            if (c0 + j < CT)
                                acc[rt][c0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[rt], xb[j], acc[rt][c0 + j], 0, 0, 0);
                    if (CT > 8) __builtin_amdgcn_sched_barrier(0);      // keep the halves apart
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    bacc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[rt], __builtin_bit_cast(bf16x8, ones4), bacc[rt], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead copies must not outlive the workgroup
    if (!rows) return;
    const int r = lane & 31, h = lane >> 5;
    float* part = a.part + (size_t)blockIdx.x * 256 * (32 * CT);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row0 = 32 * (wave * RT + rt);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                part[(size_t)row * (32 * CT) + 32 * ct + r] = acc[rt][ct][e];
            }
        if (a.bpart && r == 0) {        // every column of the ones product holds the sums: take column 0
#pragma unroll
            for (int e = 0; e < 16; ++e)
                a.bpart[(size_t)blockIdx.x * 256 + row0 + (e & 3) + 8 * (e >> 2) + 4 * h] = bacc[rt][e];
        }
    }
}

// First phase: G = `pg` pieces from piece g_piece of the gradient tape, X = 2 ct pieces from piece x_piece of the
// forward tape; partial tiles [parts][256][32 ct] (and bias partials) go to ws.  Returns the number of parts.
static int launch_wgrad16(const unsigned char* gtape, bool vd, int g_piece, int pg, const unsigned char* tape, int x_piece,
                          int ct, long long m_pad, float* ws, bool want_bias, hipStream_t st, int* parts_out) {
    const int n_chunks = (int)(m_pad / 32);
    int parts = n_chunks / 8;                         // at least 8 chunks (256 samples) per workgroup
    if (parts > WG_MAX_PARTS) parts = WG_MAX_PARTS;
    if (parts > cu_count()) parts = cu_count();
    if (parts < 1) parts = 1;
    const int per_wg = (n_chunks + parts - 1) / parts;
    parts = (n_chunks + per_wg - 1) / per_wg;
    NERF_REQUIRE(32 * ct <= WG_PART_COLS, "weight gradient (bf16): %d column tiles exceed the workspace", ct);
    Wg16Args a{gtape, (unsigned)(gtotal(vd) * PIECE_BYTES), (unsigned)(g_piece * PIECE_BYTES),
               tape, (unsigned)(total(vd) * PIECE_BYTES), (unsigned)(x_piece * PIECE_BYTES),
               n_chunks, per_wg, ws, want_bias ? ws + WG_BIAS_OFFSET : nullptr};
    const dim3 grid((unsigned)parts), block(256);
    const int key = pg * 16 + ct;
    switch (key) {
        case 16 * 16 + 8: hipLaunchKernelGGL((wgrad16_kernel<16, 8>), grid, block, 0, st, a); break;
        case 16 * 16 + 10: hipLaunchKernelGGL((wgrad16_kernel<16, 10>), grid, block, 0, st, a); break;
        case 16 * 16 + 2: hipLaunchKernelGGL((wgrad16_kernel<16, 2>), grid, block, 0, st, a); break;
        case 8 * 16 + 9: hipLaunchKernelGGL((wgrad16_kernel<8, 9>), grid, block, 0, st, a); break;
        case 2 * 16 + 8: hipLaunchKernelGGL((wgrad16_kernel<2, 8>), grid, block, 0, st, a); break;
        case 2 * 16 + 4: hipLaunchKernelGGL((wgrad16_kernel<2, 4>), grid, block, 0, st, a); break;
        default: NERF_REQUIRE(false, "weight gradient (bf16): unsupported shape %d pieces x %d tiles", pg, ct);
    }
    NERF_LAUNCH_CHECK();
    *parts_out = parts;
    return NERF_OK;
}

size_t tape16_bytes(long long M, bool vd) { return (size_t)(pad256(M) / 64) * total(vd) * PIECE_BYTES; }
size_t gtape16_bytes(long long M, bool vd) { return (size_t)(pad256(M) / 64) * gtotal(vd) * PIECE_BYTES; }

int pack_backward_bf16(const float* const* tensors_host, bool vd, int output_ch, void* packed_out, hipStream_t st) {
    PackArgs a;
    a.net = make_backward_desc(vd, output_ch, true);
    a.bf16 = 1;
    a.with_scales = 0;
    a.total_frags = total_frags(a.net, 1);
    const int want = vd ? FRAGS_BWD_BF16_VD : FRAGS_BWD_BF16_NOVD;
    a.padded_frags = pad_to_stage(want);
    NERF_REQUIRE(a.total_frags == want, "nerf_mlp_pack_backward: internal layout mismatch");
    const int n_t = vd ? 24 : 20;
    for (int i = 0; i < 24; ++i) a.t[i] = i < n_t ? tensors_host[i] : nullptr;
    for (int i = 0; i < a.net.n_layers; ++i) {
        NERF_REQUIRE(a.t[a.net.layer[i].w_idx], "nerf_mlp_pack_backward: weight tensor %d is null", a.net.layer[i].w_idx);
        if (a.net.layer[i].extra_frags) NERF_REQUIRE(a.t[a.net.layer[i].w2_idx], "nerf_mlp_pack_backward: weight tensor %d is null", a.net.layer[i].w2_idx);
    }
    a.out = static_cast<unsigned char*>(packed_out);
    return launch_pack(a, st);
}

int backward_bf16(const void* packed_bwd, const void* packed_fwd, bool vd, int output_ch, const float* d_raw, long long M,
                  const void* tape_v, void* gtape_v, float* ws, float* const* g, hipStream_t st) {
    const unsigned char* tape = static_cast<const unsigned char*>(tape_v);
    unsigned char* gtape = static_cast<unsigned char*>(gtape_v);
    Dgrad16Args d;
    d.packed = static_cast<const unsigned char*>(packed_bwd);
    d.n_stages = pad_to_stage(vd ? FRAGS_BWD_BF16_VD : FRAGS_BWD_BF16_NOVD) / STAGE_FRAGS;
    d.M = M; d.d_raw = d_raw; d.tape = tape; d.gtape = gtape;
    const long long n_tiles = (M + 255) / 256;
    const dim3 grid((unsigned)(n_tiles < cu_count() ? n_tiles : cu_count())), block(MLP_THREADS);
    if (vd) hipLaunchKernelGGL((mlp_dgrad_bf16_kernel<true, 4>), grid, block, 0, st, d);
    else if (output_ch == 4) hipLaunchKernelGGL((mlp_dgrad_bf16_kernel<false, 4>), grid, block, 0, st, d);
    else hipLaunchKernelGGL((mlp_dgrad_bf16_kernel<false, 5>), grid, block, 0, st, d);
    NERF_LAUNCH_CHECK();

    // output scales of the forward stream: the tape holds scale * activation (mlp_layout.h)
    const float* sc = reinterpret_cast<const float*>(static_cast<const unsigned char*>(packed_fwd) +
                                                     (size_t)(vd ? FRAGS_BF16_VD : FRAGS_BF16_NOVD) * FRAG_BYTES +
                                                     (size_t)(vd ? BIAS_TILES_VD : BIAS_TILES_NOVD) * BIAS_TILE_BYTES);
    const long long mp = pad256(M);
    int rc;
#define NERF_TRY(x) do { rc = (x); if (rc != NERF_OK) return rc; } while (0)
    // (kernel, then one reduction per run of columns of the parameter tensor; maps 3 / 4 put the encoded-input
    //  columns of the bf16 operand order back into reference order)
    int parts = 0;
    NERF_TRY(launch_wgrad16(gtape, vd, dz(0), 16, tape, PE, 2, mp, ws, true, st, &parts));
    NERF_TRY(launch_wgrad_reduce(ws, parts, 64, 0, 256, 64, g[0], 63, 0, 3, nullptr, g[1], st));
    for (int i = 1; i < 8; ++i) {
        if (i == 5) {   // pts_linears.5 reads [encoded points, h4]: one pass over [h4, points] on the tape
            NERF_TRY(launch_wgrad16(gtape, vd, dz(5), 16, tape, h(vd, 4), 10, mp, ws, true, st, &parts));
            NERF_TRY(launch_wgrad_reduce(ws, parts, 320, 0, 256, 256, g[10], 319, 63, 0, sc + 4, g[11], st));
            NERF_TRY(launch_wgrad_reduce(ws, parts, 320, 256, 256, 64, g[10], 319, 0, 3, nullptr, nullptr, st));
            continue;
        }
        NERF_TRY(launch_wgrad16(gtape, vd, dz(i), 16, tape, h(vd, i - 1), 8, mp, ws, true, st, &parts));
        NERF_TRY(launch_wgrad_reduce(ws, parts, 256, 0, 256, 256, g[2 * i], 256, 0, 0, sc + (i - 1), g[2 * i + 1], st));
    }
    if (vd) {
        NERF_TRY(launch_wgrad16(gtape, vd, DFEAT, 16, tape, h(vd, 7), 8, mp, ws, true, st, &parts));
        NERF_TRY(launch_wgrad_reduce(ws, parts, 256, 0, 256, 256, g[18], 256, 0, 0, sc + 7, g[19], st));
        // views_linears.0 reads [feature, encoded view]: contiguous on the tape
        NERF_TRY(launch_wgrad16(gtape, vd, DZV, 8, tape, FEAT, 9, mp, ws, true, st, &parts));
        NERF_TRY(launch_wgrad_reduce(ws, parts, 288, 0, 128, 256, g[16], 283, 0, 0, sc + 8, g[17], st));
        NERF_TRY(launch_wgrad_reduce(ws, parts, 288, 256, 128, 32, g[16], 283, 256, 4, nullptr, nullptr, st));
        NERF_TRY(launch_wgrad16(gtape, vd, DSIG, 2, tape, h(vd, 7), 8, mp, ws, true, st, &parts));
        NERF_TRY(launch_wgrad_reduce(ws, parts, 256, 0, 1, 256, g[20], 256, 0, 0, sc + 7, g[21], st));
        NERF_TRY(launch_wgrad16(gtape, vd, DRGB, 2, tape, HV, 4, mp, ws, true, st, &parts));
        NERF_TRY(launch_wgrad_reduce(ws, parts, 128, 0, 3, 128, g[22], 128, 0, 0, sc + 9, g[23], st));
    } else {
        NERF_TRY(launch_wgrad16(gtape, vd, DOUT, 2, tape, h(vd, 7), 8, mp, ws, true, st, &parts));
        NERF_TRY(launch_wgrad_reduce(ws, parts, 256, 0, output_ch, 256, g[18], 256, 0, 0, sc + 7, g[19], st));
    }
#undef NERF_TRY
    return NERF_OK;
}

}  // namespace nerf
