// bf16 INFERENCE kernel on v_mfma_f32_16x16x32_bf16 (operand conventions: mlp_layout.h).
//
// Same structure as the 32x32x16 layer loop (mlp_bf16.h), which the training kernels keep: a wave owns 64 samples
// (four 16-column tiles) and carries them through all layers in registers; weight fragments (1 KiB: 16 output rows x
// 32 k-slots) stream through the LDS ring and each feeds four MFMAs; the four accumulator registers a lane gets from
// a 16-row output tile are, after ReLU + v_cvt_pk_bf16_f32, half of its B operand for one k-step of the next layer.
// Why the other shape: the chip runs this kernel at its power limit, and at equal FLOP per cycle the 16x16x32
// instruction holds a higher clock than 32x32x16 (measured on bare loops with the A operand re-read from LDS:
// 1897 against 1670 TFLOP/s).
#pragma once
#include "mlp_bf16.h"

namespace nerf {
namespace m16 {

// 16-column tiles per wave.  Five, not four: every weight fragment read from LDS, every LDS-DMA byte, every wait and
// bias read is shared by one more MFMA (the per-fragment overhead is what separates this kernel from a bare MFMA loop),
// and 80 samples' activations (2 x 160 registers) + encodings + accumulators still fit the 512 registers of a lane.
constexpr int NCT = 5;
constexpr int WSAMP = NCT * 16;              // samples per wave and pass
constexpr int TILE16 = 4 * WSAMP;            // samples per workgroup and pass
constexpr int PF16 = 2;                      // LDS read-ahead in fragments (a fragment feeds NCT MFMAs = 80 cycles); one
                                             // more would not fit the registers
static_assert(PF16 < QD, "fragment queue depth");
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4v bias16(const unsigned char* lds_ring, int tile, int g) {
    // accumulator register r of lane group g holds output row 4 g + r of the tile
    return *reinterpret_cast<const f32x4v*>(lds_ring - BIAS_PAD + tile * BIAS16_TILE_BYTES + g * 16);
}

// ---- encodings for the lane's k-slots ---------------------------------------------------------------------------
// v_sin / v_cos take revolutions; v_fract does the range reduction.  One transcendental pair per two octaves, the
// second octave by the double-angle identities (error doubles once: far below the bf16 rounding of the result).
__device__ __forceinline__ void sincos2(float rev, float* v) {
    const float r = __builtin_amdgcn_fractf(rev);
    const float s = __builtin_amdgcn_sinf(r), c = __builtin_amdgcn_cosf(r);
    v[0] = s;
    v[1] = c;
    v[2] = 2.0f * s * c;
    v[3] = __builtin_fmaf(-2.0f * s, s, 1.0f);
}

__device__ __forceinline__ void encode_point16(float px, float py, float pz, int g, float (&v)[16]) {
    constexpr float INV_2PI = 0.15915494309189535f;
    const float sc = (g == 0 ? 1.0f : g == 1 ? 4.0f : g == 2 ? 16.0f : 64.0f) * INV_2PI;   // octaves 2 g, 2 g + 1
    sincos2(px * sc, &v[0]);
    sincos2(py * sc, &v[4]);
    sincos2(pz * sc, &v[8]);
    const float pc = g == 0 ? px : (g == 1 ? py : pz);                                     // octaves 8, 9 of coordinate g
    sincos2(pc * (256.0f * INV_2PI), &v[12]);
    if (g == 3) { v[12] = px; v[13] = py; v[14] = pz; v[15] = 0.0f; }
}

__device__ __forceinline__ void encode_view16(float dx, float dy, float dz, int g, float (&v)[8]) {
    constexpr float INV_2PI = 0.15915494309189535f;
    const float dc = g == 0 ? dx : (g == 1 ? dy : dz);
    sincos2(dc * INV_2PI, &v[0]);
    sincos2(dc * (4.0f * INV_2PI), &v[4]);
    if (g == 3) {
        v[0] = dx; v[1] = dy; v[2] = dz;
#pragma unroll
        for (int j = 3; j < 8; ++j) v[j] = 0.0f;
    }
}

template <bool VD>
__device__ __forceinline__ void encode_raw16(const RawIn& r, int g, float (&pe)[16], float (&vw)[8]) {
    // pts = rays_o + rays_d * z (run_nerf.py:387): product and sum rounded separately
    encode_point16(fadd(r.o[0], fmul(r.d[0], r.z)), fadd(r.o[1], fmul(r.d[1], r.z)), fadd(r.o[2], fmul(r.d[2], r.z)), g, pe);
    if (VD) encode_view16(r.v[0], r.v[1], r.v[2], g, vw);
}

// ---- raw inputs through LDS ------------------------------------------------------------------------------------------
// The inputs of the NEXT pass are requested by LDS-DMA into a per-wave staging block [field 10][sample 64] (fields:
// origin 3, direction 3, view direction 3, depth) and read back with ds_read at the top of that pass.  Loads into
// registers would do, but the compiler cannot see the LDS-DMA of the weight ring in its vmcnt bookkeeping and guards
// their first use with waits that drain the whole ring and the newest requests (4 k of 102 k cycles per pass).
// Done by hand the count is exact: NIN vector-memory instructions per pass, in every mode, which the next four
// stage hand-overs of the ring add to what may stay in flight (Pipe::advance_extra).
constexpr int IN_FIELDS = 10;
constexpr int IN_FS = WSAMP * 4;                           // bytes of one field of one wave
constexpr int IN_WAVE_BYTES = IN_FIELDS * IN_FS;
constexpr int IN_STAGE_BYTES = 4 * IN_WAVE_BYTES;
constexpr int LDS16_BYTES = LDS_BYTES + IN_STAGE_BYTES;
static_assert(LDS16_BYTES <= 160 * 1024, "LDS of a compute unit");
template <bool VD> constexpr int nin() { return 2 * (VD ? 10 : 7); }      // two lane groups: samples 0..63 and 64..79

#define NERF_IN_FIRST(va) "s_mov_b32 m0, %[base]\n\ts_nop 0\n\tglobal_load_lds_dword " va ", off\n\t"
#define NERF_IN_NEXT(va, inc, off) "s_add_u32 m0, m0, %[" inc "]\n\ts_nop 0\n\tglobal_load_lds_dword " va ", off offset:" #off "\n\t"
#define NERF_IN_OPEN "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %[keep], m0\n\ts_mov_b64 %[ex], exec\n\ts_mov_b64 exec, %[mask]\n\t"
#define NERF_IN_CLOSE "s_mov_b64 exec, %[ex]\n\ts_mov_b32 m0, %[keep]"
// One lane group (the lanes of `mask`) requests the fields of ITS samples: field j of lane l lands at base + IN_FS j +
// 4 l.  The LDS address of a load is M0 + instruction offset + 4 l and the same offset is added to the memory address,
// hence the uneven M0 steps: FS - 4 between consecutive floats of a row (c1), FS - 12 across near / far (c2),
// FS + 40 / FS + 8 / FS back to a pointer's first float (c3, c5, c0), 4 FS + 20 / 4 FS over unused fields (c4, c6).
template <bool VD>
__device__ __forceinline__ void request_group(const MlpArgs& a, unsigned m, unsigned base, unsigned long long mask) {
    const unsigned ray = sample_ray(a, m);
    unsigned keep;
    unsigned long long ex;
#define NERF_IN_CONSTS [c0] "n"(IN_FS), [c1] "n"(IN_FS - 4), [c2] "n"(IN_FS - 12), [c3] "n"(IN_FS + 40), [c4] "n"(4 * IN_FS + 20), \
                       [c5] "n"(IN_FS + 8), [c6] "n"(4 * IN_FS)
    if (a.mode == NERF_IN_RAYS) {
        const unsigned long long row = reinterpret_cast<unsigned long long>(a.in0) + (unsigned long long)ray * (unsigned)a.ray_cols * 4ull;
        const unsigned long long zz = reinterpret_cast<unsigned long long>(a.in1) + (unsigned long long)m * 4ull;
        if (VD)
            asm volatile(NERF_IN_OPEN
                         NERF_IN_FIRST("%[row]") NERF_IN_NEXT("%[row]", "c1", 4) NERF_IN_NEXT("%[row]", "c1", 8)
                         NERF_IN_NEXT("%[row]", "c1", 12) NERF_IN_NEXT("%[row]", "c1", 16) NERF_IN_NEXT("%[row]", "c1", 20)
                         NERF_IN_NEXT("%[row]", "c2", 32) NERF_IN_NEXT("%[row]", "c1", 36) NERF_IN_NEXT("%[row]", "c1", 40)
                         NERF_IN_NEXT("%[zz]", "c3", 0)
                         NERF_IN_CLOSE
                         : [keep] "=&s"(keep), [ex] "=&s"(ex)
                         : [row] "v"(row), [zz] "v"(zz), [base] "s"(base), [mask] "s"(mask), NERF_IN_CONSTS : "memory", "scc");
        else
            asm volatile(NERF_IN_OPEN
                         NERF_IN_FIRST("%[row]") NERF_IN_NEXT("%[row]", "c1", 4) NERF_IN_NEXT("%[row]", "c1", 8)
                         NERF_IN_NEXT("%[row]", "c1", 12) NERF_IN_NEXT("%[row]", "c1", 16) NERF_IN_NEXT("%[row]", "c1", 20)
                         NERF_IN_NEXT("%[zz]", "c4", 0)
                         NERF_IN_CLOSE
                         : [keep] "=&s"(keep), [ex] "=&s"(ex)
                         : [row] "v"(row), [zz] "v"(zz), [base] "s"(base), [mask] "s"(mask), NERF_IN_CONSTS : "memory", "scc");
    } else {
        // NERF_IN_POINTS: origin = point, direction and depth are zero (not read back); their slots are filled by
        // repeats of the first load so that the instruction count is the same in both modes
        const unsigned long long pt = reinterpret_cast<unsigned long long>(a.in0) + (unsigned long long)m * 12ull;
        const unsigned long long vd = VD ? reinterpret_cast<unsigned long long>(a.in1) + (unsigned long long)ray * 12ull : pt;
        if (VD)
            asm volatile(NERF_IN_OPEN
                         NERF_IN_FIRST("%[pt]") NERF_IN_NEXT("%[pt]", "c1", 4) NERF_IN_NEXT("%[pt]", "c1", 8)
                         NERF_IN_NEXT("%[pt]", "c5", 0) NERF_IN_NEXT("%[pt]", "c0", 0) NERF_IN_NEXT("%[pt]", "c0", 0)
                         NERF_IN_NEXT("%[vd]", "c0", 0) NERF_IN_NEXT("%[vd]", "c1", 4) NERF_IN_NEXT("%[vd]", "c1", 8)
                         NERF_IN_NEXT("%[pt]", "c5", 0)
                         NERF_IN_CLOSE
                         : [keep] "=&s"(keep), [ex] "=&s"(ex)
                         : [pt] "v"(pt), [vd] "v"(vd), [base] "s"(base), [mask] "s"(mask), NERF_IN_CONSTS : "memory", "scc");
        else
            asm volatile(NERF_IN_OPEN
                         NERF_IN_FIRST("%[pt]") NERF_IN_NEXT("%[pt]", "c1", 4) NERF_IN_NEXT("%[pt]", "c1", 8)
                         NERF_IN_NEXT("%[pt]", "c5", 0) NERF_IN_NEXT("%[pt]", "c0", 0) NERF_IN_NEXT("%[pt]", "c0", 0)
                         NERF_IN_NEXT("%[pt]", "c6", 0)
                         NERF_IN_CLOSE
                         : [keep] "=&s"(keep), [ex] "=&s"(ex)
                         : [pt] "v"(pt), [base] "s"(base), [mask] "s"(mask), NERF_IN_CONSTS : "memory", "scc");
    }
#undef NERF_IN_CONSTS
}
#undef NERF_IN_FIRST
#undef NERF_IN_NEXT
#undef NERF_IN_OPEN
#undef NERF_IN_CLOSE

// The wave's WSAMP samples starting at `first` (clamped to the last valid sample `last`): lanes 0..63 take samples
// 0..63, then lanes 0..15 samples 64..79.  nin<VD>() vector-memory instructions, always.
template <bool VD>
__device__ __forceinline__ void request_inputs(const MlpArgs& a, unsigned first, unsigned last, int lane,
                                               unsigned wave_block_lds_addr) {
    // sample numbers stay below 2^31 (checked by the host), so unsigned arithmetic cannot wrap; a `first` beyond the
    // end (the request after the last pass) only has to produce valid addresses
    const unsigned base = __builtin_amdgcn_readfirstlane(wave_block_lds_addr);
    const unsigned m0 = min(first, last) + lane, m1 = min(first, last) + 64 + (lane & 15);
    request_group<VD>(a, min(m0, last), base, ~0ull);
    request_group<VD>(a, min(m1, last), base + 256u, 0xffffull);
}

// the staged inputs of sample s (0 .. WSAMP - 1) of the wave
template <bool VD>
__device__ __forceinline__ void read_inputs(const float* block, int s, bool rays_mode, RawIn& r) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.o[k] = block[k * WSAMP + s];
        const float d = block[(3 + k) * WSAMP + s];
        r.d[k] = rays_mode ? d : 0.0f;
        r.v[k] = VD ? block[(6 + k) * WSAMP + s] : 0.0f;
    }
    const float z = block[9 * WSAMP + s];
    r.z = rays_mode ? z : 0.0f;
}

// NERF_IN_FEATURES: the caller supplies the encodings; pick this lane's k-slots out of the row.
template <bool VD>
__device__ __forceinline__ void load_features16(const MlpArgs& a, unsigned m, int g, float (&pe)[16], float (&vw)[8]) {
    const int ch = VD ? 90 : 63;
    const float* f = a.in0 + (size_t)m * ch;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i0 = pe16_slot_feature(0, q), i1 = pe16_slot_feature(1, q), i2 = pe16_slot_feature(2, q),
                  i3 = pe16_slot_feature(3, q);
        const int idx = g == 0 ? i0 : (g == 1 ? i1 : (g == 2 ? i2 : i3));
        pe[q] = idx >= 0 ? f[idx >= 0 ? idx : 0] : 0.0f;
    }
    if (VD) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i0 = view16_slot_feature(0, q), i1 = view16_slot_feature(1, q), i2 = view16_slot_feature(2, q),
                      i3 = view16_slot_feature(3, q);
            const int idx = g == 0 ? i0 : (g == 1 ? i1 : (g == 2 ? i2 : i3));
            vw[q] = idx >= 0 ? f[63 + (idx >= 0 ? idx : 0)] : 0.0f;
        }
    }
}

// ---- layer loop ---------------------------------------------------------------------------------------------------
// What is still to be done with the accumulators of the tile that ran last: convert them into dwords pdw, pdw + 1 of
// the operand quads dst[0][ct] (nothing if !live).
struct Pending {
    u32x4 (*dst)[NCT];
    int pdw;
    bool relu, live;
};

// slice sl (0 .. 2 NCT - 1) of a finished tile's epilogue: two accumulator registers -> one packed operand dword
__device__ __forceinline__ void epi_slice16(const f32x4v (&accp)[NCT], const Pending& p, int sl) {
    const int ct = sl >> 1, half = sl & 1;
    p.dst[0][ct][p.pdw + half] = pack_pair(accp[ct][2 * half], accp[ct][2 * half + 1], p.relu);
}

// One 16-row output tile: k-steps [pe KPE][hidden KH][view KVW] accumulate into acc while the epilogue of the
// previous tile (accumulators accp) is issued in the MFMA gaps of k-steps 1..EPI_N.  `early`: this tile reads, at
// its last hidden k-step, what that epilogue writes (first tile of a layer), so the epilogue must be done two
// k-steps before the end.
// `staged`: raw-input requests of this pass (NIN of them, issued before position 0) may still be in flight; the first
// four stage hand-overs of the pass allow for them.
template <int KPE, int KH, int KVW, bool early, int NIN>
__device__ __forceinline__ void tile16(Stream& st, int pos, int g, bool staged, int bias_next_idx, f32x4v& bias,
                                       const u32x4 (*pe)[NCT], const u32x4 (*hin)[NCT], const u32x4 (*vw)[NCT],
                                       f32x4v (&acc)[NCT], const f32x4v (&accp)[NCT], const Pending& prev) {
    constexpr int KS = KPE + KH + KVW;
    constexpr int EPI_N = early ? (KS - 2 < 6 ? KS - 2 : 6) : KS - 1;     // 0: no k-step to hide it under
    f32x4v bias_next = bias;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = pos + s;
        if ((k & (STAGE_FRAGS - 1)) == STAGE_FRAGS / 2) {
            if (k < 4 * STAGE_FRAGS && staged) st.p.template advance_extra<NIN>();
            else st.p.advance();
        }
        __builtin_amdgcn_sched_barrier(0);
        st.fetch(k + PF16);
        if (s == KS / 2) bias_next = bias16(st.p.lds, bias_next_idx, g);
        __builtin_amdgcn_sched_barrier(0);
        const u32x4* b = s < KPE ? pe[s] : (s < KPE + KH ? hin[s - KPE] : vw[s - KPE - KH]);
        const bf16x8 w = __builtin_bit_cast(bf16x8, st.q[k & (QD - 1)]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, b[ct]), s == 0 ? bias : acc[ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (prev.live && EPI_N >= 1 && s >= 1 && s <= EPI_N) {
                // gaps 0 .. NCT EPI_N - 1 of k-steps 1..EPI_N share the 2 NCT slices evenly
                const int gap = NCT * (s - 1) + ct, ngap = NCT * (EPI_N >= 1 ? EPI_N : 1);
                const int lo = (2 * NCT * gap + ngap - 1) / ngap, hi = (2 * NCT * (gap + 1) + ngap - 1) / ngap;
#pragma unroll
                for (int sl = lo; sl < (hi < 2 * NCT ? hi : 2 * NCT); ++sl) epi_slice16(accp, prev, sl);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (prev.live && EPI_N < 1) {
#pragma unroll
        for (int sl = 0; sl < 2 * NCT; ++sl) epi_slice16(accp, prev, sl);
    }
    bias = bias_next;
}

// NT tiles of one layer.  Tile number bt of the pass accumulates into accs[bt & 1]; tile t of the layer goes to dwords
// 2 (t & 1), 2 (t & 1) + 1 of the quads hout[t >> 1][ct].  `pend` describes the tile that ran before this layer on
// entry and this layer's last tile on return.
template <int NT, int KPE, int KH, int KVW, int NIN>
__device__ __forceinline__ void layer16(Stream& st, int& pos, int& bt, int n_bias_tiles, int g, bool staged, f32x4v& bias,
                                        const u32x4 (*pe)[NCT], const u32x4 (*hin)[NCT], const u32x4 (*vw)[NCT],
                                        f32x4v (&accs)[2][NCT], Pending& pend, u32x4 (*hout)[NCT], bool relu, bool convert) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nb = (bt + 1) % n_bias_tiles;
        if (t == 0) tile16<KPE, KH, KVW, true, NIN>(st, pos, g, staged, nb, bias, pe, hin, vw, accs[bt & 1], accs[(bt & 1) ^ 1], pend);
        else tile16<KPE, KH, KVW, false, NIN>(st, pos, g, staged, nb, bias, pe, hin, vw, accs[bt & 1], accs[(bt & 1) ^ 1], pend);
        pend = Pending{hout + (t >> 1), 2 * (t & 1), relu, convert};
        pos += KPE + KH + KVW;
        bt += 1;
    }
}

template <bool VD, int OUT_CH>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp16_bf16_kernel(MlpArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS16_BYTES];
    const TileWalk tw = tile_walk(a, (a.M + TILE16 - 1) / TILE16);
    Stream st;
    pipe_start(st.p, a, lds);
    st.slot0 = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    constexpr int NB = VD ? BIAS_TILES16_VD : BIAS_TILES16_NOVD;
    constexpr int NFRAGS = VD ? FRAGS16_VD : FRAGS16_NOVD;
    constexpr int NPAD = VD ? FRAGS_BF16_VD : FRAGS_BF16_NOVD;      // stream length with the zero padding

#pragma unroll
    for (int k = 0; k < PF16; ++k) st.fetch(k);
    f32x4v bias = bias16(st.p.lds, 0, g);

    // raw inputs are requested one whole pass ahead (one wave per SIMD: nothing else hides the latency)
    const bool feat_mode = a.mode == NERF_IN_FEATURES, rays_mode = a.mode == NERF_IN_RAYS;
    const bool staged = !feat_mode;
    const unsigned s_end = (unsigned)tw.sample_end;      // one past the last sample of this network (< 2^31)
    const unsigned s_last = s_end - 1;
    const float* in_block = reinterpret_cast<const float*>(lds + LDS_BYTES + wave * IN_WAVE_BYTES);
    const unsigned in_block_addr = (unsigned)(size_t)(lds_ptr_t)(lds + LDS_BYTES) + wave * IN_WAVE_BYTES;
    if (staged) {
        request_inputs<VD>(a, (unsigned)tw.sample0 + (unsigned)tw.first * TILE16 + wave * WSAMP, s_last, lane, in_block_addr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // once per launch: the first pass has nothing to hide it under
    }

#ifdef NERF_EXP_STAMP
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, acc_t[6] = {0, 0, 0, 0, 0, 0};
#endif
    for (long long t = tw.first; t < tw.end; t += tw.stride) {
        const unsigned pass0 = (unsigned)tw.sample0 + (unsigned)t * TILE16 + wave * WSAMP;
        NERF_STAMP(t0)
        u32x4 pe[2][NCT], vw[1][NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = pass0 + ct * 16 + c;
            float pv[16], vv[8];
            if (feat_mode) {
                load_features16<VD>(a, m < s_end ? m : s_last, g, pv, vv);
            } else {
                RawIn cur;
                read_inputs<VD>(in_block, ct * 16 + c, rays_mode, cur);
                encode_raw16<VD>(cur, g, pv, vv);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) pe[ks][ct][j] = pack_pair(pv[ks * 8 + 2 * j], pv[ks * 8 + 2 * j + 1], false);
            if (VD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) vw[0][ct][j] = pack_pair(vv[2 * j], vv[2 * j + 1], false);
            }
        }
        if (staged) {
            // always issued, so that the count the ring's waits allow for is exact: past the end, the last sample again
            // (the last pass of a 2^31-sample launch: pass0 + stride TILE16 < 2^31 + 256 x 320, no wrap)
            request_inputs<VD>(a, pass0 + (unsigned)tw.stride * TILE16, s_last, lane, in_block_addr);
        }

        NERF_STAMP(t1)
        u32x4 A[8][NCT], B[8][NCT];     // the 256 features of a layer's input / output as 8 k-step operands
        f32x4v accs[2][NCT];
        Pending pend{A, 0, false, false};
        int pos = 0, bt = 0;
#define NERF_L16(NT, KPE, KH, KVW, hin, hout, relu, conv) \
    layer16<NT, KPE, KH, KVW, nin<VD>()>(st, pos, bt, NB, g, staged, bias, pe, hin, vw, accs, pend, hout, relu, conv)
        NERF_L16(16, 2, 0, 0, A, A, true, true);      // L0: pe -> A
        NERF_L16(16, 0, 8, 0, A, B, true, true);      // L1
        NERF_L16(16, 0, 8, 0, B, A, true, true);      // L2
        NERF_L16(16, 0, 8, 0, A, B, true, true);      // L3
        NERF_L16(16, 0, 8, 0, B, A, true, true);      // L4
        NERF_STAMP(t2)
        NERF_L16(16, 2, 8, 0, A, B, true, true);      // L5: [pe, h]
        NERF_L16(16, 0, 8, 0, B, A, true, true);      // L6
        NERF_L16(16, 0, 8, 0, A, B, true, true);      // L7 -> B

        NERF_STAMP(t3)
        float o0[NCT], o1[NCT], o2[NCT], o3[NCT], o4[NCT];
        if (VD) {
            NERF_L16(16, 0, 8, 0, B, A, false, true);     // feature -> A (no relu)
            NERF_L16(1, 0, 8, 0, B, A, false, false);     // alpha: row 3 of its tile, kept as float
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) o3[ct] = accs[(bt - 1) & 1][ct][3];
            NERF_L16(8, 0, 8, 1, A, B, true, true);       // views: [feature, view] -> B[0..3]
            NERF_L16(1, 0, 4, 0, B, B, false, false);     // rgb: rows 0..2
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x4v& r = accs[(bt - 1) & 1][ct];
                o0[ct] = r[0]; o1[ct] = r[1]; o2[ct] = r[2]; o4[ct] = 0.f;
            }
        } else {
            NERF_L16(1, 0, 8, 0, B, B, false, false);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x4v& r = accs[(bt - 1) & 1][ct];
                o0[ct] = r[0]; o1[ct] = r[1]; o2[ct] = r[2]; o3[ct] = r[3];
                o4[ct] = r[0];   // row 4 sits in register 0 of lane group 1
            }
        }
#undef NERF_L16
        // The zero padding at the end of the stream: nothing to compute, but the ring keeps turning.  Its one mid-stage
        // position makes the next pass's first stage visible; then the first PF16 fragments of that pass are queued.
        // (Written out: left as a loop, hipcc does not unroll it and spills the ring state to LDS.)
        constexpr int KMID = (NFRAGS & ~(STAGE_FRAGS - 1)) + STAGE_FRAGS / 2;
        static_assert(KMID >= NFRAGS && KMID < NPAD, "the padding covers the last stage's mid-point");
        st.tick(KMID);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < PF16; ++k) st.fetch(NPAD + k);
        __builtin_amdgcn_sched_barrier(0);
        st.end_pass(NPAD);
        NERF_STAMP(t4)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const unsigned m = pass0 + ct * 16 + c;
            if (m < s_end) {
                if (OUT_CH == 4) {
                    if (g == 0) *reinterpret_cast<float4*>(a.out + (size_t)m * 4) = make_float4(o0[ct], o1[ct], o2[ct], o3[ct]);
                } else {
                    float* o = a.out + (size_t)m * OUT_CH;
                    if (g == 0) { o[0] = o0[ct]; o[1] = o1[ct]; o[2] = o2[ct]; o[3] = o3[ct]; }
                    else if (g == 1) { o[4] = o4[ct]; }
                }
            }
        }
#ifdef NERF_EXP_STAMP
        NERF_STAMP(t5)
        acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[2] += t3 - t2; acc_t[3] += t4 - t3; acc_t[4] += t5 - t4; acc_t[5] += 1;
#endif
    }
#ifdef NERF_EXP_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 256)
        for (int i = 0; i < 6; ++i) g_stamps[blockIdx.x * 8 + i] = acc_t[i];
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

}  // namespace m16
}  // namespace nerf
