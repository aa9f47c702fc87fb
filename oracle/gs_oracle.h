/*
 * gs_oracle.h -- CPU restatement of the arhik/GaussianSplat hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or
 * call it, and only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (Julia + CUDA.jl device kernels) ships no tests,
 * golden vectors or fixtures and cannot be executed in this pipeline (no Julia, no
 * NVIDIA GPU, no CPU code path).  This file follows the reference's written
 * arithmetic statement by statement (citations are /root/reference/src/<file>:<line>)
 * and is pinned instead by (1) an independent NumPy restatement (gs_oracle_np.py)
 * that must agree bit-for-bit on every fp32/integer output, (2) closed-form
 * known-answer tests and (3) fp64 autograd / finite differences for the adjoint.
 *
 * Numeric contract ("the spec", shared with the HIP path):
 *   - every fp32 operation is individually IEEE-rounded in the written order
 *     (compile with -ffp-contract=off, no fast-math);
 *   - Float64-literal promotions of the Julia source are reproduced (marked f64);
 *   - exp() is gso_expf below (range reduction + degree-6 polynomial, fp32, no fma);
 *   - Julia max/min NaN propagation is reproduced.
 */
#ifndef GS_ORACLE_H
#define GS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSO_ORDER_INDEX      0  /* literal reference: per-tile lists in gaussian-index order */
#define GSO_ORDER_DEPTH_DESC 1  /* sortperm(-tps[3,:]) order, forward.jl:103 (far -> near)   */
#define GSO_ORDER_DEPTH_ASC  2  /* near -> far (extension)                                    */

typedef struct {
    int32_t W, H;          /* image size, pixels                                  */
    float   T[16];         /* world->view, column-major (camera.jl:88-100)         */
    float   P[16];         /* view->clip, column-major  (camera.jl:102-111)        */
    float   fx, fy;
    float   near_, far_;
    float   eye[3];
    float   lookAt[3];
} gso_camera;

/* camera.jl:24-47 defaults + camera.jl:88-111 matrices */
void gso_camera_matrices(const float eye[3], const float lookAt[3], const float up[3],
                         float fx, float fy, float near_, float far_, int W, int H,
                         gso_camera *out);

float gso_expf(float x);

/* projection.jl:39-155, cov2d.jl:30-45, boundingbox.jl:4-36, splat.jl:175-193.
 * Any output pointer may be NULL.  Layouts are the reference's column-major
 * [component, gaussian]. */
void gso_preprocess(int64_t n, int sh_degree,
                    const float *means, const float *scales, const float *quats,
                    const float *opacities, const float *shs, const gso_camera *cam,
                    float *ts, float *tps, float *mu, float *cov3d, float *cov2d,
                    float *invcov, float *bbs, float *rgb, float *sig);

/* forward.jl:103 -- stable permutation (0-based ids), NaN last like isless. */
void gso_depth_order(int64_t n, const float *tps, int order, uint32_t *perm);
/* 32-bit radix key whose unsigned order equals the isless order used above. */
uint32_t gso_depth_key(float clipz, int order);

/* binning.jl:3-35 tile rectangle of one gaussian: 1-based inclusive, clipped to the
 * grid; returns 0 when the gaussian touches no tile. */
int gso_tile_rect(const float bb[4], int tile, int gx, int gy, int32_t rect[4]);

/* Sparse per-tile lists (binning.jl + forward.jl:137-141 + compact.jl semantics).
 * perm==NULL -> index order.  ranges: 2*gx*gy uint32 [start,end) with tile id
 * (ty-1)*gx+(tx-1).  ids (0-based gaussian ids) and keys (tile<<32 | depth key, or
 * tile<<32 | id in index order) are written when non-NULL, capacity cap; returns the
 * number of instances (even if > cap). */
int64_t gso_bin(int64_t n, const float *bbs, const float *tps, const uint32_t *perm,
                int order, int tile, int gx, int gy,
                uint32_t *ranges, uint32_t *ids, uint64_t *keys, int64_t cap);

/* Literal dense path (binning.jl hits -> inclusive scan over the gaussian axis ->
 * compact.jl) for small cases.  hitIdxs is gx*gy*maxBin uint32 (1-based ids, 0 =
 * empty slot), x fastest.  Returns maxHits. */
int64_t gso_bin_dense_literal(int64_t n, const float *bbs, int tile, int gx, int gy,
                              uint32_t *hitIdxs, int64_t maxBin);

/* splat.jl:195-269.  t_min == 0 is the literal reference (no early-out).  t_min > 0 (build
 * extension): a pixel stops taking splats at the first list position that is a multiple of
 * GSO_EARLY_BATCH (counted from the start of its tile's list) at which its T < t_min. */
#define GSO_EARLY_BATCH 64
void gso_composite_forward(const gso_camera *cam, int tile, int gx, int gy,
                           const uint32_t *ranges, const uint32_t *ids,
                           const float *mu, const float *invcov, const float *bbs,
                           const float *sig, const float *rgb, const float *tps,
                           float t_min, float *image, float *trans);

/* Derived adjoint of the forward above (the reference has no valid 3-D backward,
 * backward.jl:3-38 / splat.jl:271-396 are a stale 2-D kernel).  fp64 arithmetic with
 * the fp32 forward's discrete decisions (bbs, lists, order).  Gradients ACCUMULATE
 * (+=) into the d* arrays (reference contract, splat.jl:137-173). */
void gso_backward(int64_t n, int sh_degree,
                  const float *means, const float *scales, const float *quats,
                  const float *opacities, const float *shs, const gso_camera *cam,
                  int tile, int gx, int gy, const uint32_t *ranges, const uint32_t *ids,
                  const float *bbs, float t_min, const float *dC,
                  double *dmeans, double *dscales, double *dquats, double *dopac,
                  double *dshs,
                  double *g2d /* optional n*10: drgb3,dsig,dmu2,dinv4 */);

/* ---- 2-D image-fitting renderer (GAUSSIAN_2D) ---------------------------------------- */

/* The spec's sin/cos for cov2d.jl:6-8 (Cody-Waite by pi/2 + Cephes polynomials, fp32 mul/add only). */
void gso_sincosf(float x, float *sn, float *cs);

/* preprocess(::GaussianRenderer2D), forward.jl:9-33 = cov2d.jl:3-45 + boundingbox.jl:4-36 on the pixel position
 * (w*mx, h*my) of splat.jl:337-339.  means 2xN in [0,1]^2, scales 2xN (log), rots 1xN, opacities 1xN (raw),
 * colors 3xN.  Outputs as gso_preprocess (rgb = colors, sig = opacities). */
void gso_preprocess2d(int64_t n, const float *means, const float *scales, const float *rots,
                      const float *opacities, const float *colors, int W, int H,
                      float *mu, float *cov2d, float *invcov, float *bbs, float *rgb, float *sig);

/* fp64 adjoint of gso_preprocess2d + composite; gradients ACCUMULATE (SplatGrads2D, splat.jl:28-34). */
void gso_backward2d(int64_t n, const float *means, const float *scales, const float *rots,
                    const float *opacities, const float *colors, int W, int H,
                    int tile, int gx, int gy, const uint32_t *ranges, const uint32_t *ids,
                    float t_min, const float *dC,
                    double *dmeans, double *dscales, double *drots, double *dopac, double *dcolors,
                    double *g2d /* optional n*10 */);

int gso_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
