// nerf_render_rays: render_rays of the reference (run_nerf.py:314-424, nerf_utils.py:248-342) as ONE library call.
// Host code only: it enqueues the kernels of the other entry points on the caller's stream, in the reference's order,
// into caller-owned buffers.  Nothing is allocated, nothing synchronises.
#include "common.h"

using namespace nerf;

extern "C" int nerf_render_rays(const void* packed_coarse, const void* packed_fine, const float* rays, int R,
                                const nerf_render_cfg* cfg, const nerf_render_buffers* buf, void* stream) {
    NERF_REQUIRE(cfg && buf, "nerf_render_rays: null configuration or buffer block");
    NERF_REQUIRE(R >= 0 && cfg->n_samples >= 1 && cfg->n_importance >= 0, "nerf_render_rays: bad sizes R=%d S=%d N_importance=%d",
                 R, cfg->n_samples, cfg->n_importance);
    NERF_REQUIRE(cfg->ray_cols >= (cfg->use_viewdirs ? 11 : 8), "nerf_render_rays: rays need %d columns, got %d",
                 cfg->use_viewdirs ? 11 : 8, cfg->ray_cols);
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(packed_coarse && rays && cfg->t_vals, "nerf_render_rays: null network, rays or t_vals");
    const int S = cfg->n_samples, NI = cfg->n_importance, C = cfg->output_ch, cols = cfg->ray_cols;
    const bool fine = NI > 0;
    NERF_REQUIRE(buf->rgb && buf->raw && buf->z_vals, "nerf_render_rays: rgb, raw and z_vals buffers are required");
    if (fine) {
        NERF_REQUIRE(cfg->u && buf->z_coarse && buf->raw_coarse && buf->weights,
                     "nerf_render_rays: importance sampling needs u and the z_coarse, raw_coarse and weights buffers");
        NERF_REQUIRE(S >= 3, "nerf_render_rays: importance sampling needs at least 3 coarse samples, got %d", S);
    }
    float* z0 = fine ? buf->z_coarse : buf->z_vals;
    float* raw0 = fine ? buf->raw_coarse : buf->raw;
    int rc;
    // z_vals, optional stratified jitter (run_nerf.py:363-385)
    if ((rc = nerf_z_vals(rays, cols, R, S, cfg->t_vals, cfg->t_rand, cfg->lindisp, z0, stream))) return rc;
    // coarse network on pts = o + d z, encodings fused (:387-389)
    if ((rc = nerf_mlp_forward(packed_coarse, cfg->precision, cfg->use_viewdirs, C, NERF_IN_RAYS, rays, z0, R, S, cols, raw0,
                               stream))) return rc;
    if (!fine)      // :391
        return nerf_raw2outputs(raw0, C, z0, rays + 3, cols, cfg->noise0, R, S, cfg->white_bkgd, buf->rgb, buf->disp, buf->acc,
                                buf->depth, buf->weights, stream);
    if ((rc = nerf_raw2outputs(raw0, C, z0, rays + 3, cols, cfg->noise0, R, S, cfg->white_bkgd, buf->rgb0, buf->disp0, buf->acc0,
                               nullptr, buf->weights, stream))) return rc;
    // mid-points, sample_pdf on weights[..., 1:-1], sort(cat), z_std (:393-401, 418)
    if ((rc = nerf_sample_pdf(z0, S, 1, S - 1, buf->weights + 1, S, cfg->u, NI, cfg->u_per_ray, R, z0, S, nullptr, nullptr,
                              buf->z_vals, buf->z_std, stream))) return rc;
    // fine network (or the coarse one again when there is none, :403) and the final compositing (:405-407)
    if ((rc = nerf_mlp_forward(packed_fine ? packed_fine : packed_coarse, cfg->precision, cfg->use_viewdirs, C, NERF_IN_RAYS, rays,
                               buf->z_vals, R, S + NI, cols, buf->raw, stream))) return rc;
    return nerf_raw2outputs(buf->raw, C, buf->z_vals, rays + 3, cols, cfg->noise1, R, S + NI, cfg->white_bkgd, buf->rgb, buf->disp,
                            buf->acc, buf->depth, nullptr, stream);
}
