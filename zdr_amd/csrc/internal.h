// internal.h — structures shared by the host side (zdr_api.cpp) and the kernels (zdr_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sampler.h"
#include "scene.h"

#define ZDR_MAX_RECORDED_DEPTH 16     // prb.py:15 max_depth; vertex records kept per path in backward

// primary ring (integrators.h): parked camera-ray vertices per lane, and camera samples generated per refill
#ifndef ZDR_RING_CAP
#define ZDR_RING_CAP 16
#endif
#ifndef ZDR_RING_BATCH
#define ZDR_RING_BATCH 8
#endif

#define ZDR_MAX_PERSISTENT_BLOCKS 8192   // 256 CUs x 4 SIMDs x 8 waves: upper bound of the path kernels' persistent grid

// Wave-uniform launch configuration (kernel argument, lives in SGPRs).
struct RenderCfg {
    int32_t width, height;
    int32_t x0, y0, x1, y1;
    uint32_t sample_begin, sample_end;
    uint32_t chunk;                   // samples per wave: chunk c covers [begin + c*chunk, ...)
    int32_t nchunks, tiles_x, tiles_y;
    int32_t shard_index, shard_count, ntiles;   // interleaved tile shard: tile numbers index, index + count, ... ; ntiles = how many that is
    int32_t shard_skew;                         // row r of the tile grid is numbered starting at column r * skew (1 when sharded: diagonals; 0 otherwise)
    int32_t use_tent, max_depth, rr_depth;
    int32_t tex_h, tex_w;
    int32_t prb_mode;                 // backward, path: ZDR_PRB_* of zdr.h (expectation / detached: roulette factors and MIS weights held constant / literal: the weight of prb.py:162)
    int32_t cell_copies;              // backward: replicas of the staging-cell array (scene.h: few texels), >= 1
    float two_over_w, two_over_h, aspect;      // integrator.py:22-23
    float inv_spp;                             // 1 / spp as computed by IEEE division
    float alpha;                               // (sample_end - sample_begin) / spp
    float cam_o[3], cam_fwd[3], cam_right[3], cam_upp[3], cam_tan;   // camera.py:12-15
    int32_t debug_no_scatter;         // measurement builds only (-DZDR_MEASURE, env ZDR_DEBUG_NO_SCATTER; always 0 in the product): timing-only ablations — 1 gradients computed, not added; 2 atomics confined to 64 KiB; 3 no sweep; 4 / 5 the sweep loop cut after 2 / 1 iterations (the tail of long paths dropped: what any scheme that defers it could save at most); 6 queue and flush run, no atomic is issued
};

struct KernelIO {
    const float4 *material;           // (tex_h, tex_w) float4
    float4 *image;                    // (H, W) float4
    float4 *partial;                  // scratch when nchunks > 1: [chunk][tile of the shard][lane] float4 (one 1 KiB line per wave)
    const float4 *d_image;            // backward: cotangent
    float *d_material;                // backward: += gathered from the staging cells by k_cells_to_grad
    float *cells;                     // backward: (tex_h + 1) x (tex_w + 1) staging cells of 16 floats, zeroed per call
    unsigned long long *counters;     // stats variant: 8 counters
    const unsigned long long *tile_masks;   // brute-force accel: per 8x8 tile, the triangle pairs its camera rays can hit (k_tile_masks); null = all
    int32_t tile_masks_valid;               // the masks in the buffer already belong to this camera and shard: skip k_tile_masks
    unsigned int *work_counters;      // path integrator: 8 item counters, one per XCD, zeroed before the launch (fetch_item)
    float4 *ring;                     // path integrator: one FIFO of parked camera-ray vertices per persistent workgroup (integrators.h)
};

int zdr_launch_render(const DScene &S, const RenderCfg &R, const SamplerCfg &C, const KernelIO &io,
                      int integrator, int accel_is_bvh, int backward, int stats, hipStream_t stream);
int zdr_launch_zero(void *p, size_t bytes, hipStream_t stream);   // kernel zero-fill (graph-safe, see zdr_kernels.hip)
int zdr_launch_trace(const DScene &S, int accel_is_bvh, int any, const float *rays, uint32_t n,
                     int32_t *out_i, float *out_f, hipStream_t stream);
int zdr_launch_sampler_dump(const SamplerCfg &C, const int32_t *queries, uint32_t n, int32_t nvert,
                            int32_t rr_depth, float *out, int as_path_kernels, int *batched, hipStream_t stream);
int zdr_launch_path_dump(const DScene &S, const RenderCfg &R, const SamplerCfg &C, const KernelIO &io, int accel_is_bvh,
                         const int32_t *queries, uint32_t n, int32_t maxv, float *out, hipStream_t stream);
