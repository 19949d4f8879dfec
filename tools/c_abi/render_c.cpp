// render_c: the whole coarse + fine render of one view through the C ABI of libnerf_hip.so, from a program that knows
// neither Python nor torch: what a C/C++ serving loop would do with include/nerf_hip.h (render -> render_rays of
// run_nerf.py:69-134, 314-424 as seven library calls on one HIP stream).
//
//   render_c <input file> <output file> direct|graph|onecall [timing repetitions]
//
// "onecall": ray generation, then nerf_render_rays: the other six calls as one.
//
// "graph": the seven calls are recorded once with hipStreamBeginCapture and replayed with hipGraphLaunch; no entry
// point on this path allocates or synchronises, so the sequence is capturable as it stands.  With repetitions > 0 the
// program also times that many replays against as many direct issues and prints one JSON line.
//
// Input file (little endian): int32 {magic 0x4652454e, H, W, S, N_importance, precision, white_bkgd}; float32 {fx, fy,
// cx, cy, near, far}; c2w[12]; t_vals[S]; u[N_importance]; then the 24 state_dict tensors of the coarse network and
// the 24 of the fine network (run_nerf_helpers.py:67-94, view directions on).  Output file: float32 rgb[N,3], disp[N],
// acc[N], rgb0[N,3], disp0[N], acc0[N], z_std[N] with N = H*W.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/nerf_hip.h"

#define HIP_OK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "render_c: %s: %s (line %d)\n", #expr, hipGetErrorString(e_), __LINE__); \
            return 2;                                                                            \
        }                                                                                        \
    } while (0)
#define NERF_OK_OR_DIE(expr)                                                                     \
    do {                                                                                         \
        int rc_ = (expr);                                                                        \
        if (rc_ != NERF_OK) {                                                                    \
            fprintf(stderr, "render_c: %s -> %d: %s (line %d)\n", #expr, rc_, nerf_last_error(), __LINE__); \
            return 3;                                                                            \
        }                                                                                        \
    } while (0)

// element counts of the 24 tensors in state_dict order
static const size_t TENSOR_NUMEL[24] = {
    256 * 63,  256, 256 * 256, 256, 256 * 256, 256, 256 * 256, 256, 256 * 256, 256, 256 * 319, 256, 256 * 256, 256,
    256 * 256, 256,                       // pts_linears.0 .. 7
    128 * 283, 128,                       // views_linears.0
    256 * 256, 256,                       // feature_linear
    256,       1,                         // alpha_linear
    3 * 128,   3};                        // rgb_linear

struct Network {
    float* tensors[24];
    void* packed;
};

static bool read_exact(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes; }

static int load_network(FILE* f, int precision, hipStream_t st, Network& n) {
    std::vector<float> host;
    for (int i = 0; i < 24; ++i) {
        host.resize(TENSOR_NUMEL[i]);
        if (!read_exact(f, host.data(), host.size() * 4)) {
            fprintf(stderr, "render_c: input file ends inside tensor %d\n", i);
            return 1;
        }
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&n.tensors[i]), host.size() * 4));
        HIP_OK(hipMemcpy(n.tensors[i], host.data(), host.size() * 4, hipMemcpyHostToDevice));
    }
    HIP_OK(hipMalloc(&n.packed, nerf_mlp_packed_bytes(precision, 1)));
    NERF_OK_OR_DIE(nerf_mlp_pack(n.tensors, 1, 4, precision, n.packed, st));
    return 0;
}

struct Frame {          // everything one render reads and writes, all device memory
    int H, W, S, NI, precision, white;
    float fx, fy, cx, cy, near, far;
    float *c2w, *t_vals, *u, *rays, *z, *raw, *weights, *z_all, *raw_all;
    float *rgb, *disp, *acc, *depth, *rgb0, *disp0, *acc0, *z_std;
    Network coarse, fine;
};

static bool g_onecall = false;

// render() + render_rays() of the reference as library calls, all asynchronous on `st`
static int enqueue(const Frame& f, hipStream_t st) {
    const int N = f.H * f.W, S = f.S, SA = f.S + f.NI;
    NERF_OK_OR_DIE(nerf_make_rays(N, f.H, f.W, nullptr, nullptr, f.c2w, nullptr, f.fx, f.fy, f.cx, f.cy, f.near, f.far,
                                  nullptr, nullptr, 1, 0, f.fx, 1.0f, f.rays, st));
    if (g_onecall) {
        nerf_render_cfg cfg{};
        cfg.n_samples = S; cfg.n_importance = f.NI; cfg.precision = f.precision; cfg.use_viewdirs = 1; cfg.output_ch = 4;
        cfg.ray_cols = 11; cfg.white_bkgd = f.white; cfg.t_vals = f.t_vals; cfg.u = f.u;
        nerf_render_buffers b{};
        b.rgb = f.rgb; b.disp = f.disp; b.acc = f.acc; b.depth = f.depth; b.rgb0 = f.rgb0; b.disp0 = f.disp0; b.acc0 = f.acc0;
        b.z_std = f.z_std; b.raw = f.raw_all; b.z_vals = f.z_all; b.z_coarse = f.z; b.raw_coarse = f.raw; b.weights = f.weights;
        NERF_OK_OR_DIE(nerf_render_rays(f.coarse.packed, f.fine.packed, f.rays, N, &cfg, &b, st));
        return 0;
    }
    NERF_OK_OR_DIE(nerf_z_vals(f.rays, 11, N, S, f.t_vals, nullptr, 0, f.z, st));
    NERF_OK_OR_DIE(nerf_mlp_forward(f.coarse.packed, f.precision, 1, 4, NERF_IN_RAYS, f.rays, f.z, N, S, 11, f.raw, st));
    NERF_OK_OR_DIE(nerf_raw2outputs(f.raw, 4, f.z, f.rays + 3, 11, nullptr, N, S, f.white, f.rgb0, f.disp0, f.acc0, f.depth,
                                    f.weights, st));
    NERF_OK_OR_DIE(nerf_sample_pdf(f.z, S, 1, S - 1, f.weights + 1, S, f.u, f.NI, 0, N, f.z, S, nullptr, nullptr, f.z_all,
                                   f.z_std, st));
    NERF_OK_OR_DIE(nerf_mlp_forward(f.fine.packed, f.precision, 1, 4, NERF_IN_RAYS, f.rays, f.z_all, N, SA, 11, f.raw_all, st));
    NERF_OK_OR_DIE(nerf_raw2outputs(f.raw_all, 4, f.z_all, f.rays + 3, 11, nullptr, N, SA, f.white, f.rgb, f.disp, f.acc,
                                    f.depth, nullptr, st));
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: render_c <input> <output> direct|graph|onecall [timing repetitions]\n");
        return 1;
    }
    const bool use_graph = strcmp(argv[3], "graph") == 0;
    g_onecall = strcmp(argv[3], "onecall") == 0;
    const int reps = argc > 4 ? atoi(argv[4]) : 0;
    if (nerf_hip_abi_version() != NERF_HIP_ABI_VERSION) {
        fprintf(stderr, "render_c: library ABI %d, header %d\n", nerf_hip_abi_version(), NERF_HIP_ABI_VERSION);
        return 1;
    }
    FILE* fi = fopen(argv[1], "rb");
    if (!fi) { perror(argv[1]); return 1; }
    int32_t hdr[7];
    float cam[6], c2w[12];
    if (!read_exact(fi, hdr, sizeof hdr) || hdr[0] != 0x4652454e || !read_exact(fi, cam, sizeof cam) ||
        !read_exact(fi, c2w, sizeof c2w)) {
        fprintf(stderr, "render_c: bad input header\n");
        return 1;
    }
    Frame f{};
    f.H = hdr[1]; f.W = hdr[2]; f.S = hdr[3]; f.NI = hdr[4]; f.precision = hdr[5]; f.white = hdr[6];
    f.fx = cam[0]; f.fy = cam[1]; f.cx = cam[2]; f.cy = cam[3]; f.near = cam[4]; f.far = cam[5];
    if (f.H < 1 || f.W < 1 || f.S < 2 || f.NI < 1 || (long long)f.H * f.W > (1 << 24)) {
        fprintf(stderr, "render_c: sizes out of range\n");
        return 1;
    }
    const size_t N = (size_t)f.H * f.W, S = f.S, SA = (size_t)f.S + f.NI;
    std::vector<float> tv(S), u(f.NI);
    if (!read_exact(fi, tv.data(), S * 4) || !read_exact(fi, u.data(), (size_t)f.NI * 4)) {
        fprintf(stderr, "render_c: input file ends inside the tables\n");
        return 1;
    }

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    if (int rc = load_network(fi, f.precision, st, f.coarse)) return rc;
    if (int rc = load_network(fi, f.precision, st, f.fine)) return rc;
    fclose(fi);

    struct { float** p; size_t n; } bufs[] = {
        {&f.c2w, 12}, {&f.t_vals, S}, {&f.u, (size_t)f.NI}, {&f.rays, N * 11}, {&f.z, N * S}, {&f.raw, N * S * 4},
        {&f.weights, N * S}, {&f.z_all, N * SA}, {&f.raw_all, N * SA * 4}, {&f.rgb, N * 3}, {&f.disp, N}, {&f.acc, N},
        {&f.depth, N}, {&f.rgb0, N * 3}, {&f.disp0, N}, {&f.acc0, N}, {&f.z_std, N}};
    for (auto& b : bufs) HIP_OK(hipMalloc(reinterpret_cast<void**>(b.p), b.n * 4));
    HIP_OK(hipMemcpy(f.c2w, c2w, sizeof c2w, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f.t_vals, tv.data(), S * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(f.u, u.data(), (size_t)f.NI * 4, hipMemcpyHostToDevice));

    // the first issue is always direct: it also makes the library's per-device look-ups (compute-unit count)
    if (int rc = enqueue(f, st)) return rc;
    HIP_OK(hipStreamSynchronize(st));

    hipGraphExec_t exec = nullptr;
    if (use_graph) {
        float** outs[] = {&f.rgb, &f.disp, &f.acc, &f.rgb0, &f.disp0, &f.acc0, &f.z_std};
        const size_t widths[] = {3, 1, 1, 3, 1, 1, 1};
        for (int i = 0; i < 7; ++i) HIP_OK(hipMemset(*outs[i], 0xff, N * widths[i] * 4));      // the replay must rewrite them
        hipGraph_t graph;
        HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        const int rc = enqueue(f, st);
        HIP_OK(hipStreamEndCapture(st, &graph));
        if (rc) return rc;
        HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIP_OK(hipGraphLaunch(exec, st));
        HIP_OK(hipStreamSynchronize(st));
    }

    if (reps > 0) {
        auto run = [&](bool graph_mode, double& ms) -> int {
            HIP_OK(hipStreamSynchronize(st));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) {
                if (graph_mode) HIP_OK(hipGraphLaunch(exec, st));
                else if (int rc = enqueue(f, st)) return rc;
            }
            HIP_OK(hipStreamSynchronize(st));
            ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
            return 0;
        };
        double ms_direct = 0.0, ms_graph = 0.0;
        if (int rc = run(false, ms_direct)) return rc;
        if (use_graph)
            if (int rc = run(true, ms_graph)) return rc;
        char graph_ms[32] = "null";
        if (use_graph) snprintf(graph_ms, sizeof graph_ms, "%.4f", ms_graph);
        printf("{\"rays\": %zu, \"precision\": %d, \"reps\": %d, \"ms_direct\": %.4f, \"ms_graph\": %s}\n", N, f.precision,
               reps, ms_direct, graph_ms);
    }

    FILE* fo = fopen(argv[2], "wb");
    if (!fo) { perror(argv[2]); return 1; }
    struct { const float* p; size_t n; } outs[] = {{f.rgb, N * 3}, {f.disp, N}, {f.acc, N}, {f.rgb0, N * 3},
                                                  {f.disp0, N}, {f.acc0, N}, {f.z_std, N}};
    std::vector<float> host;
    for (auto& o : outs) {
        host.resize(o.n);
        HIP_OK(hipMemcpy(host.data(), o.p, o.n * 4, hipMemcpyDeviceToHost));
        if (fwrite(host.data(), 4, o.n, fo) != o.n) { perror("write"); return 1; }
    }
    fclose(fo);
    return 0;
}
