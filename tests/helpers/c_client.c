/* A caller of libzdr_hip.so that is neither Python nor torch: plain C99, include/zdr.h, device memory from the HIP runtime's C API.
 * tests/test_gpu_c_client.py compiles it with gcc, hands it a scene file and compares what it writes with zdr_amd.Scene — the drop-in
 * boundary of DESIGN.md 1 ("plain pointers and sizes") exercised from the other side.
 *   c_client <scene.bin> <out.bin>
 * scene.bin: int32 head[16] = {nverts, ntris, ninst, W, H, spp, seed, tex_h, tex_w, integrator, use_tent, max_depth, rr_depth, accel, 0, 0},
 *            float camera[10] = {fov, origin, target, up}, verts8, tris, inst_tri_begin (ninst + 1), inst_xform (ninst x 16),
 *            inst_emission (ninst x 3), material (tex_h x tex_w x 4), cotangent (H x W x 4)
 * out.bin:   image (H x W x 4 float32), d_material (tex_h x tex_w x 4 float32; backward called with seed as Scene.render_backward does: + 1 inside) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "zdr.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define ZDR(x) do { int rc_ = (x); if (rc_ != ZDR_OK) { fprintf(stderr, "%s: %d %s\n", #x, rc_, zdr_last_error()); return 3; } } while (0)

static void *slurp(FILE *f, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short scene file\n"); exit(4); }
    return p;
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: c_client scene.bin out.bin\n"); return 1; }
    if (zdr_abi_version() != ZDR_ABI_VERSION) { fprintf(stderr, "header %d, library %d\n", ZDR_ABI_VERSION, zdr_abi_version()); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t *head = (int32_t *)slurp(f, 16 * sizeof(int32_t));
    const uint32_t nverts = (uint32_t)head[0], ntris = (uint32_t)head[1], ninst = (uint32_t)head[2];
    const int W = head[3], H = head[4], tex_h = head[7], tex_w = head[8];
    float *cam = (float *)slurp(f, 10 * sizeof(float));
    float *verts = (float *)slurp(f, (size_t)nverts * 8 * sizeof(float));
    int32_t *tris = (int32_t *)slurp(f, (size_t)ntris * 3 * sizeof(int32_t));
    int32_t *begin = (int32_t *)slurp(f, ((size_t)ninst + 1) * sizeof(int32_t));
    float *xform = (float *)slurp(f, (size_t)ninst * 16 * sizeof(float));
    float *emission = (float *)slurp(f, (size_t)ninst * 3 * sizeof(float));
    const size_t mat_bytes = (size_t)tex_h * tex_w * 4 * sizeof(float), img_bytes = (size_t)H * W * 4 * sizeof(float);
    float *material = (float *)slurp(f, mat_bytes);
    float *cotangent = (float *)slurp(f, img_bytes);
    fclose(f);

    zdr_scene *scene = NULL;
    ZDR(zdr_scene_create(verts, nverts, tris, ntris, begin, xform, emission, ninst, 0, head[13], &scene));
    zdr_scene_info_t info;
    ZDR(zdr_scene_info(scene, &info));
    printf("%s: %u triangles, %u instances, %u lights, accel %d, %llu bytes of HBM\n", zdr_version(), info.ntris, info.ninst, info.light_count,
           info.accel, (unsigned long long)info.device_bytes);

    zdr_render_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.integrator = head[9]; p.sampler = ZDR_SAMPLER_CMJ;
    p.width = W; p.height = H; p.spp = (uint32_t)head[5]; p.seed = (uint32_t)head[6];
    p.use_tent = head[10];
    p.x0 = 0; p.y0 = 0; p.x1 = W; p.y1 = H;
    p.sample_begin = 0; p.sample_end = p.spp;
    p.max_depth = head[11]; p.rr_depth = head[12];
    p.camera.fov = cam[0];
    memcpy(p.camera.origin, cam + 1, 3 * sizeof(float)); memcpy(p.camera.target, cam + 4, 3 * sizeof(float)); memcpy(p.camera.up, cam + 7, 3 * sizeof(float));
    p.tex_h = tex_h; p.tex_w = tex_w;
    p.tile_shard_index = 0; p.tile_shard_count = 1;
    p.prb_mode = ZDR_PRB_EXPECTATION;

    hipStream_t stream;
    HIP(hipStreamCreate(&stream));
    float *d_material, *d_image, *d_cot, *d_grad;
    HIP(hipMalloc((void **)&d_material, mat_bytes)); HIP(hipMalloc((void **)&d_grad, mat_bytes));
    HIP(hipMalloc((void **)&d_image, img_bytes)); HIP(hipMalloc((void **)&d_cot, img_bytes));
    HIP(hipMemcpyAsync(d_material, material, mat_bytes, hipMemcpyHostToDevice, stream));
    HIP(hipMemcpyAsync(d_cot, cotangent, img_bytes, hipMemcpyHostToDevice, stream));
    HIP(hipMemsetAsync(d_image, 0, img_bytes, stream));
    HIP(hipMemsetAsync(d_grad, 0, mat_bytes, stream));
    ZDR(zdr_render_forward(scene, &p, d_material, d_image, stream));
    p.seed += 1u;                                                   /* render.py:196: the backward pass replays seed + 1 */
    ZDR(zdr_render_backward(scene, &p, d_cot, d_material, d_grad, stream));
    ZDR(zdr_scene_check(scene, stream));                            /* synchronises; a tripped device watchdog is an error, not a picture */

    float *image = (float *)malloc(img_bytes), *grad = (float *)malloc(mat_bytes);
    HIP(hipMemcpy(image, d_image, img_bytes, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(grad, d_grad, mat_bytes, hipMemcpyDeviceToHost));
    FILE *o = fopen(argv[2], "wb");
    if (!o || fwrite(image, 1, img_bytes, o) != img_bytes || fwrite(grad, 1, mat_bytes, o) != mat_bytes || fclose(o) != 0) { perror(argv[2]); return 1; }

    /* a wrong struct_size must be refused before anything is launched */
    p.struct_size = sizeof p - 4;
    if (zdr_render_forward(scene, &p, d_material, d_image, stream) != ZDR_E_INVALID) { fprintf(stderr, "a short struct was accepted\n"); return 5; }
    printf("refused: %s\n", zdr_last_error());

    ZDR(zdr_scene_destroy(scene));
    HIP(hipFree(d_material)); HIP(hipFree(d_grad)); HIP(hipFree(d_image)); HIP(hipFree(d_cot));
    HIP(hipStreamDestroy(stream));
    printf("ok\n");
    return 0;
}
