// Device-side building blocks shared by the inference kernels (mlp.hip) and the training kernels
// (train.hip): the LDS weight ring, the input encoders and the exact-float32 layer loop.
// See mlp_layout.h for the operand conventions.
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace nerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NSTAGE = 8;                   // LDS ring slots of STAGE_BYTES
#ifndef NERF_PREFETCH
#define NERF_PREFETCH 4
#endif
constexpr int PREFETCH = NERF_PREFETCH;     // stages in flight beyond the one being made ready (<= NSTAGE-3)
constexpr int RING_BYTES = NSTAGE * STAGE_BYTES;
// LDS map: [bias tiles | padding to BIAS_PAD][ring].  The bias block comes first so that its ds_read
// addresses fit the 16-bit instruction offset (no per-tile address arithmetic).
constexpr int BIAS_PAD = 10240;
constexpr int LDS_BYTES = BIAS_PAD + RING_BYTES;
static_assert(BIAS_TILES_VD * BIAS_TILE_BYTES <= BIAS_PAD, "bias block");
constexpr int MLP_THREADS = 256;
static_assert(PREFETCH <= NSTAGE - 3, "a slot may be refilled only two stages after its last read");
static_assert(PREFETCH * 4 <= 63, "vmcnt is a 6-bit counter");

#define NERF_STR2(x) #x
#define NERF_STR(x) NERF_STR2(x)
#if defined(NERF_EXP_HALFDMA)
#define NERF_VMCNT_KEEP "8"
#elif NERF_PREFETCH == 3
#define NERF_VMCNT_KEEP "12"
#elif NERF_PREFETCH == 4
#define NERF_VMCNT_KEEP "16"
#elif NERF_PREFETCH == 5
#define NERF_VMCNT_KEEP "20"
#else
#error "NERF_PREFETCH must be 3, 4 or 5"
#endif                                       // = PREFETCH stages x 4 LDS-DMA loads per wave and stage

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct MlpArgs {
    const unsigned char* packed;   // fragment stream, then the bias block
    int n_stages;                  // stages per network pass
    int n_bias_tiles;
    int mode;                      // NERF_IN_*
    const float* in0;
    const float* in1;
    int R, S, ray_cols;
    long long M;                   // samples
    float* out;
    unsigned char* tape;           // bf16 training forward only (train_bf16.h)
    // several independent networks in one launch (nerf_mlp_forward_objects): DEVICE table of n_obj packed streams,
    // network o owns tiles [o * tiles_per_obj, (o + 1) * tiles_per_obj); table == nullptr: the single stream `packed`
    const unsigned char* const* table;
    int n_obj;
    long long tiles_per_obj;
    long long samples_per_obj;     // bf16 inference kernel: a network's samples need not fill its last tile
    size_t table_off;              // byte offset of the stream this kernel reads inside each packed buffer
    // sample index / S without a division (mlp16.h): exact for indices < 2^31, set by set_sample_divisor()
    unsigned div_magic;
    int div_shift;
};

// n / d for n < 2^31 as (n * magic) >> (31 + l), l = ceil(log2 d), magic = floor(2^(31+l) / d) + 1 < 2^32: the error
// magic * d - 2^(31+l) is at most d, and n * d < 2^(31+l) for every n < 2^31, so the quotient is exact.
inline void set_sample_divisor(MlpArgs& a) {
    const unsigned d = (unsigned)a.S;
    int l = 0;
    while ((1ull << l) < d) ++l;
    a.div_magic = (unsigned)((1ull << (31 + l)) / d + 1ull);
    a.div_shift = l;
}
__device__ __forceinline__ unsigned sample_ray(const MlpArgs& a, unsigned m) {
    return a.div_shift == 0 ? m : (__umulhi(m, a.div_magic) >> (a.div_shift - 1));
}

// ------------------------------------------------------------------------------------------
// weight stream: LDS ring fed by LDS-DMA
// ------------------------------------------------------------------------------------------
struct Pipe {
    const unsigned char* src;        // packed fragment stream (wave-uniform)
    unsigned src_off;                // wave * 4 KiB + lane * 16
    unsigned char* lds;              // ring base
    unsigned lds_addr;               // the same as a 32-bit LDS byte address (for M0)
    unsigned wave_off;               // wave * 4 KiB: this wave's quarter of every stage
    unsigned lane16;                 // lane * 16
    int n_stages;
    int issue_src;                   // next source stage (wraps at n_stages)
    unsigned issue_slot;             // next ring slot to fill
    unsigned read_off;               // byte offset of the slot being read
    unsigned next_read_slot;

    __device__ __forceinline__ void issue() {
        // wave-uniform 64-bit source of the stage (scalar registers) + one constant 32-bit lane offset
        const unsigned long long g = reinterpret_cast<unsigned long long>(src) + (unsigned long long)issue_src * STAGE_BYTES;
        const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)g), ghi = __builtin_amdgcn_readfirstlane((unsigned)(g >> 32));
        const unsigned long long gs = ((unsigned long long)ghi << 32) | glo;
        // LDS-DMA: 4 x 1 KiB per wave.  The LDS address is M0 + lane*16 + the instruction offset.
        // Written as inline asm on purpose: once hipcc sees a global_load_lds it treats the
        // LDS counter as out of order and turns every counted "s_waitcnt lgkmcnt(N)" of the
        // fragment queue into lgkmcnt(0).  The loads are waited for by hand (advance()).
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr + issue_slot * STAGE_BYTES + wave_off);
        unsigned keep;
#ifdef NERF_EXP_NODMA     // timing experiment only (results are garbage): what the LDS-DMA issue costs the wave
        (void)keep; (void)m0v; (void)gs;
#else
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %3\n\t"
            "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
#ifndef NERF_EXP_HALFDMA    // timing experiment only (results are garbage): half the bytes and instructions
            "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
            "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
#endif
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(src_off), "s"(m0v), "s"(gs)
            : "memory");
#endif
        issue_src = (issue_src + 1 == n_stages) ? 0 : issue_src + 1;
        issue_slot = (issue_slot + 1) & (NSTAGE - 1);
    }
    // Issue one more stage, then make the oldest outstanding one visible to the workgroup.
    __device__ __forceinline__ void advance() {
        issue();
#ifdef NERF_EXP_LAXWAIT      // timing experiment only (results are garbage): twelve more operations may stay in flight
        asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(" NERF_VMCNT_KEEP ")" ::: "memory");
#endif
#ifndef NERF_EXP_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
    }
    // advance() while EXTRA younger vector-memory operations of this wave (issued after the stage that has to become
    // visible) are in flight as well: vmcnt counts in issue order, so they are added to what may stay outstanding
    template <int EXTRA>
    __device__ __forceinline__ void advance_extra() {
        static_assert(PREFETCH * 4 + EXTRA <= 63, "vmcnt is a 6-bit counter");
        issue();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PREFETCH * 4 + EXTRA) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void next_stage() {
        read_off = next_read_slot * STAGE_BYTES;
        next_read_slot = (next_read_slot + 1) & (NSTAGE - 1);
    }
    // Called before fragment `pos` (position in the pass) is consumed.  After full unrolling
    // `pos` is a constant and exactly one branch per 8 fragments survives.
    __device__ __forceinline__ void tick(int pos) {
        if ((pos & (STAGE_FRAGS - 1)) == 0) next_stage();
        if ((pos & (STAGE_FRAGS - 1)) == STAGE_FRAGS / 2) advance();
    }
    __device__ __forceinline__ const unsigned char* frag(int pos) const {
        return lds + read_off + (pos & (STAGE_FRAGS - 1)) * FRAG_BYTES + lane16;
    }
};

// Which tiles a workgroup takes.  One network: tiles b, b + G, ... of all n_tiles (G workgroups).  Several networks:
// workgroup b serves network b % n_obj for its whole life (so its weight ring never changes stream) and, as number
// b / n_obj of the workgroups of that network, every stride-th of that network's tiles.
struct TileWalk {
    long long base, first, end, stride;      // tile = base + t, t = first, first + stride, ... < end
    long long sample0, sample_end;           // the samples of the network this workgroup serves: [sample0, sample_end)
};
__device__ __forceinline__ TileWalk tile_walk(MlpArgs& a, long long n_tiles) {
    if (!a.table) return TileWalk{0, (long long)blockIdx.x, n_tiles, (long long)gridDim.x, 0, a.M};
    const int obj = (int)(blockIdx.x % (unsigned)a.n_obj);
    a.packed = a.table[obj] + a.table_off;
    return TileWalk{obj * a.tiles_per_obj, (long long)(blockIdx.x / (unsigned)a.n_obj), a.tiles_per_obj,
                    (long long)((gridDim.x - obj + a.n_obj - 1) / a.n_obj), obj * a.samples_per_obj,
                    (obj + 1) * a.samples_per_obj};
}

__device__ __forceinline__ void pipe_start(Pipe& p, const MlpArgs& a, unsigned char* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    p.src = a.packed;
    p.src_off = wave * 4096 + lane * 16;
    p.lds = lds + BIAS_PAD;
    p.lds_addr = (unsigned)(size_t)(lds_ptr_t)(lds + BIAS_PAD);
    p.wave_off = wave * 4096;
    p.lane16 = lane * 16;
    p.n_stages = a.n_stages;
    p.issue_src = 0;
    p.issue_slot = 0;
    p.next_read_slot = 0;
    p.read_off = 0;
    // biases: n_bias_tiles x 32 floats behind the ring
    const float* bsrc = reinterpret_cast<const float*>(a.packed + (size_t)a.n_stages * STAGE_BYTES);
    float* bdst = reinterpret_cast<float*>(lds);
    for (int i = threadIdx.x; i < a.n_bias_tiles * 32; i += MLP_THREADS) bdst[i] = bsrc[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PREFETCH; ++i) p.issue();
    p.advance();   // stage 0 ready
}

// ------------------------------------------------------------------------------------------
// encodings, computed by the lane for ITS k-slots (mlp_layout.h)
// ------------------------------------------------------------------------------------------
template <bool FAST>
__device__ __forceinline__ void sincos_any(float x, float& s, float& c) {
    if (FAST) {
        // v_sin/v_cos take revolutions and need |x| <= 256: reduce with v_fract first
        const float r = __builtin_amdgcn_fractf(x * 0.15915494309189535f);
        s = __builtin_amdgcn_sinf(r);
        c = __builtin_amdgcn_cosf(r);
    } else {
        sincosf(x, &s, &c);
    }
}

// FAST (bf16 mode): only the lowest frequency of each lane half goes through the transcendental unit; the higher
// octaves follow from the double-angle identities (3 multiply-adds instead of 2 quarter-rate ops).  The error doubles
// per octave: 4 octaves on top of ~1e-6 stay 100x below the bf16 rounding of the result.
template <bool FAST, int NF>
__device__ __forceinline__ void encode_octaves(const float (&p)[3], float* v) {
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) {
        float s, c;
        sincos_any<FAST>(p[c3], s, c);
        v[c3 * 2 + 0] = s;
        v[c3 * 2 + 1] = c;
#pragma unroll
        for (int fl = 1; fl < NF; ++fl) {
            if (FAST) {
                const float s2 = 2.0f * s * c;
                c = __builtin_fmaf(-2.0f * s, s, 1.0f);
                s = s2;
            } else {
                sincos_any<false>(p[c3] * (float)(1 << fl), s, c);   // exact: power-of-two scaling
            }
            v[(fl * 3 + c3) * 2 + 0] = s;
            v[(fl * 3 + c3) * 2 + 1] = c;
        }
    }
}

template <bool FAST>
__device__ __forceinline__ void encode_point(float px, float py, float pz, int h, float (&v)[32]) {
    const float sc = h ? 32.0f : 1.0f;   // upper lane half: frequencies 2^5 .. 2^9
    const float p[3] = {px * sc, py * sc, pz * sc};
    encode_octaves<FAST, 5>(p, &v[0]);
    v[30] = h ? pz : px;
    v[31] = h ? 0.0f : py;
}

template <bool FAST>
__device__ __forceinline__ void encode_view(float dx, float dy, float dz, int h, float (&v)[16]) {
    const float sc = h ? 4.0f : 1.0f;    // upper lane half: frequencies 2^2, 2^3
    const float p[3] = {dx * sc, dy * sc, dz * sc};
    encode_octaves<FAST, 2>(p, &v[0]);
    v[12] = h ? 0.0f : dx;
    v[13] = h ? 0.0f : dy;
    v[14] = h ? 0.0f : dz;
    v[15] = 0.0f;
}

// Raw inputs of one sample: what has to come from memory before the encodings can be computed.
// NERF_IN_POINTS is expressed as origin = point, direction = 0, z = 0 (o + 0*0 == o exactly).
struct RawIn {
    float o[3], d[3], v[3], z;
};

// Issue the loads for sample m (already clamped to < M).  Nothing is computed here, so the loads
// can be issued a whole network pass before their values are needed.
template <bool VD>
__device__ __forceinline__ void fetch_raw(const MlpArgs& a, unsigned m, RawIn& r) {
    const unsigned ray = m / (unsigned)a.S;
    if (a.mode == NERF_IN_RAYS) {
        const float* q = a.in0 + (size_t)ray * a.ray_cols;
#pragma unroll
        for (int k = 0; k < 3; ++k) { r.o[k] = q[k]; r.d[k] = q[3 + k]; }
        if (VD) {
#pragma unroll
            for (int k = 0; k < 3; ++k) r.v[k] = q[8 + k];
        }
        r.z = a.in1[m];
    } else {
        const float* q = a.in0 + (size_t)m * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) { r.o[k] = q[k]; r.d[k] = 0.0f; }
        if (VD) {
            const float* vq = a.in1 + (size_t)ray * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) r.v[k] = vq[k];
        }
        r.z = 0.0f;
    }
}

template <bool VD, bool FAST>
__device__ __forceinline__ void encode_raw(const RawIn& r, int h, float (&pe)[32], float (&vw)[16]) {
    // pts = rays_o + rays_d * z (run_nerf.py:387): product and sum rounded separately
    encode_point<FAST>(fadd(r.o[0], fmul(r.d[0], r.z)), fadd(r.o[1], fmul(r.d[1], r.z)),
                       fadd(r.o[2], fmul(r.d[2], r.z)), h, pe);
    if (VD) encode_view<FAST>(r.v[0], r.v[1], r.v[2], h, vw);
}

// NERF_IN_FEATURES: the caller supplies the encodings; pick this lane's k-slots out of the row.
template <bool VD>
__device__ __forceinline__ void load_features(const MlpArgs& a, unsigned m, int h, float (&pe)[32], float (&vw)[16]) {
    const int ch = VD ? 90 : 63;
    const float* f = a.in0 + (size_t)m * ch;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const int i0 = pe_slot_feature(0, q), i1 = pe_slot_feature(1, q);
        const int idx = h ? i1 : i0;
        pe[q] = idx >= 0 ? f[idx] : 0.0f;
    }
    if (VD) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i0 = view_slot_feature(0, q), i1 = view_slot_feature(1, q);
            const int idx = h ? i1 : i0;
            vw[q] = idx >= 0 ? f[63 + idx] : 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// shared by both precisions: bias tile -> accumulator layout
// ------------------------------------------------------------------------------------------
// `lds` is the ring base (Pipe::lds)
__device__ __forceinline__ f32x16 bias_tile(const unsigned char* lds, int tile, int h) {
    // accumulator register r holds output row (r & 3) + 8 (r >> 2) + 4 h
    const f32x4* b = reinterpret_cast<const f32x4*>(lds - BIAS_PAD + tile * BIAS_TILE_BYTES + h * 16);
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 q = b[g * 2];
        acc[4 * g + 0] = q[0]; acc[4 * g + 1] = q[1]; acc[4 * g + 2] = q[2]; acc[4 * g + 3] = q[3];
    }
    return acc;
}

// ------------------------------------------------------------------------------------------
// f32 kernel: exact float32 MFMA (k-ordered fmaf chain), 1 column tile (32 samples) per wave
// ------------------------------------------------------------------------------------------
template <int NF>
__device__ __forceinline__ void mma_run_f32(Pipe& p, int pos, f32x16& acc, const float* b) {
#pragma unroll
    for (int g = 0; g < NF; ++g) {
        p.tick(pos + g);
        const f32x4 w = *reinterpret_cast<const f32x4*>(p.frag(pos + g));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], b[4 * g + j], acc, 0, 0, 0);
    }
}

struct NoTileHook {
    __device__ __forceinline__ void operator()(int, float (&)[16]) const {}
};

// One linear layer for the wave's 32 samples.  hook(t, values) runs on every finished output tile
// (after the optional relu) and may modify it: the training kernels save / mask activations there.
template <int NT, bool PE, int HT /*hidden input tiles*/, bool VW, bool RELU, class Hook = NoTileHook>
__device__ __forceinline__ void layer_f32(Pipe& p, int& pos, int& btile, int h, const float (&pe)[32],
                                          const float (&hin)[8][16], const float (&vw)[16], float (&hout)[8][16],
                                          Hook hook = Hook()) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x16 acc = bias_tile(p.lds, btile + t, h);
        if (PE) { mma_run_f32<8>(p, pos, acc, &pe[0]); pos += 8; }
        if (HT) { mma_run_f32<HT * 4>(p, pos, acc, &hin[0][0]); pos += HT * 4; }
        if (VW) { mma_run_f32<4>(p, pos, acc, &vw[0]); pos += 4; }
#pragma unroll
        for (int e = 0; e < 16; ++e) hout[t][e] = RELU ? fmaxf(acc[e], 0.0f) : acc[e];
        hook(t, hout[t]);
    }
    btile += NT;
}

// ------------------------------------------------------------------------------------------
// host side: weight packer launch (kernel in mlp.hip) and device facts
// ------------------------------------------------------------------------------------------
struct PackArgs {
    NetDesc net;
    const float* t[24];
    int bf16;
    int with_scales;      // bf16 forward stream: output scales are computed and folded into weights and biases
    int total_frags;      // fragments the descriptor defines
    int padded_frags;     // >= total_frags; the tail is zero-filled (keeps the stream stage-aligned)
    unsigned char* out;
};
int launch_pack(const PackArgs& a, hipStream_t stream);

// compute units of the CURRENT device (a process may drive several GPUs: cached per device id)
static inline int cu_count() {
    constexpr int MAX_DEV = 64;
    static int n[MAX_DEV] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 256;
    if (n[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev] = v;
    }
    return n[dev];
}


}  // namespace nerf
