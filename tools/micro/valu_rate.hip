// valu_rate.hip — measures the sustained VALU issue rate of one SIMD on gfx950 for the instruction
// classes the path-tracing kernels are made of, at 1..8 waves per SIMD.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, float a, float b, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) {          // v_fma_f32 with scalar operands
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b));
            } else if (KIND == 1) {   // v_pk_fma_f32: two floats per lane per instruction
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            } else if (KIND == 2) {   // v_mul_f32 e32 (VOP2) with an SGPR source
                asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
            } else if (KIND == 3) {   // v_cmp (VOPC -> vcc) + v_cndmask pairs
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %7, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : : "vcc");
            } else if (KIND == 4) {   // v_cmp_e64 into SGPR pairs + s_and (the mask chains of the triangle test)
                asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %2, %3\n s_and_b64 s[20:21], s[20:21], s[22:23]\n"
                             "v_cmp_lt_f32 s[22:23], %4, %5\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_cmp_lt_f32 vcc, %6, %7\n s_and_b64 vcc, s[20:21], vcc\n"
                             "v_cndmask_b32 %0, %0, %1, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : : "vcc", "s20", "s21", "s22", "s23");
            } else if (KIND == 5) {   // v_rcp_f32 (transcendental pipe)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (KIND == 7) {   // v_mul_lo_u32 (the CMJ hash is made of these)
                asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                             "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
            } else if (KIND == 8) {   // v_mul_u32_u24
                asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n"
                             "v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
            } else if (KIND == 9) {   // v_xor_b32 / v_lshrrev_b32 pairs
                asm volatile("v_xor_b32 %0, %0, %1\n v_lshrrev_b32 %1, 3, %1\n v_xor_b32 %2, %2, %3\n v_lshrrev_b32 %3, 3, %3\n"
                             "v_xor_b32 %4, %4, %5\n v_lshrrev_b32 %5, 3, %5\n v_xor_b32 %6, %6, %7\n v_lshrrev_b32 %7, 3, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (KIND == 10) {  // v_min3_f32 / v_med3
                asm volatile("v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_min3_f32 %2, %2, %3, %4\n v_min3_f32 %3, %3, %4, %5\n"
                             "v_min3_f32 %4, %4, %5, %6\n v_min3_f32 %5, %5, %6, %7\n v_min3_f32 %6, %6, %7, %0\n v_min3_f32 %7, %7, %0, %1\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (KIND == 11) {  // v_pk_mul_f32 + v_pk_add_f32 with an SGPR-pair operand and op_sel broadcast
                asm volatile("v_pk_mul_f32 %0, %4, %0 op_sel_hi:[1,0]\n v_pk_add_f32 %1, %4, %1 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3\n"
                             "v_pk_mul_f32 %0, %4, %0 op_sel_hi:[1,0]\n v_pk_add_f32 %1, %4, %1 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(pa));
            } else if (KIND == 12) {  // v_cmpx chain + v_mov under the narrowed exec + exec restore
                asm volatile("s_mov_b64 s[20:21], exec\n v_cmpx_lt_f32 vcc, %0, %1\n v_cmpx_lt_f32 vcc, %2, %3\n v_cmpx_lt_f32 vcc, %4, %5\n"
                             "v_mov_b32 %6, %7\n v_mov_b32 %0, %2\n s_mov_b64 exec, s[20:21]\n v_add_f32 %1, %1, %3\n v_add_f32 %5, %5, %3\n v_add_f32 %4, %4, %3\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : : "vcc", "s20", "s21");
            } else if (KIND == 13) {  // v_pk_mul_lo_u16: two 16-bit products per lane (the CMJ hash only needs the low bits of its state)
                asm volatile("v_pk_mul_lo_u16 %0, %0, %8\n v_pk_mul_lo_u16 %1, %1, %8\n v_pk_mul_lo_u16 %2, %2, %8\n v_pk_mul_lo_u16 %3, %3, %8\n"
                             "v_pk_mul_lo_u16 %4, %4, %8\n v_pk_mul_lo_u16 %5, %5, %8\n v_pk_mul_lo_u16 %6, %6, %8\n v_pk_mul_lo_u16 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
            } else if (KIND == 14) {  // v_pk_lshrrev_b16 / v_pk_add_u16 pairs
                asm volatile("v_pk_lshrrev_b16 %0, 3, %0\n v_pk_add_u16 %1, %1, %0\n v_pk_lshrrev_b16 %2, 3, %2\n v_pk_add_u16 %3, %3, %2\n"
                             "v_pk_lshrrev_b16 %4, 3, %4\n v_pk_add_u16 %5, %5, %4\n v_pk_lshrrev_b16 %6, 3, %6\n v_pk_add_u16 %7, %7, %6\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (KIND == 15) {  // v_perm_b32 (byte shuffle of two registers)
                asm volatile("v_perm_b32 %0, %0, %1, %8\n v_perm_b32 %1, %1, %2, %8\n v_perm_b32 %2, %2, %3, %8\n v_perm_b32 %3, %3, %4, %8\n"
                             "v_perm_b32 %4, %4, %5, %8\n v_perm_b32 %5, %5, %6, %8\n v_perm_b32 %6, %6, %7, %8\n v_perm_b32 %7, %7, %0, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
            } else if (KIND == 16) {  // v_and_b32 / v_bitop3_b32 pairs
                asm volatile("v_and_b32 %0, %0, %1\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x78\n v_and_b32 %2, %2, %3\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x78\n"
                             "v_and_b32 %4, %4, %5\n v_bitop3_b32 %5, %5, %6, %7 bitop3:0x78\n v_and_b32 %6, %6, %7\n v_bitop3_b32 %7, %7, %0, %1 bitop3:0x78\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (KIND == 17) {  // the hash's pattern: v_mul_lo_u32 feeding v_xor / v_and / v_lshrrev (dependent chain, one state)
                asm volatile("v_mul_lo_u32 %0, %0, %2\n v_and_b32 %1, 0xff, %0\n v_lshrrev_b32 %1, 2, %1\n v_xor_b32 %0, %0, %1\n"
                             "v_mul_lo_u32 %0, %0, %2\n v_and_b32 %1, 0xff, %0\n v_lshrrev_b32 %1, 2, %1\n v_xor_b32 %0, %0, %1\n"
                             : "+v"(x0), "+v"(x1) : "s"(a));
            } else if (KIND == 6) {   // dependent chain: one accumulator
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(x0) : "s"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND> static void run(const char *name, float *out, int nsimd, double mhz, int per_group) {
    printf("%-44s", name);
    for (int w : {1, 2, 3, 4, 6, 8}) {
        int iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<nsimd * w, 64>>>(out, 1.0001f, 0.5f, 10);
        hipEventRecord(e0);
        k<KIND><<<nsimd * w, 64>>>(out, 1.0001f, 0.5f, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr = (double)iters * 16 * per_group * w;          // wave-instructions per SIMD
        printf("  w=%d: %5.2f", w, ms * 1e-3 * mhz * 1e6 / instr); // cycles per wave-instruction per SIMD
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int nsimd = p.multiProcessorCount * 4; double mhz = p.clockRate / 1000.0;
    printf("%s: %d CUs, clock %.0f MHz; cycles per wave64 instruction per SIMD (at the nominal clock)\n", p.name, p.multiProcessorCount, mhz);
    float *out; hipMalloc(&out, (size_t)nsimd * 8 * 64 * 4);
    run<0>("v_fma_f32 (8 independent, sgpr operand)", out, nsimd, mhz, 8);
    run<6>("v_fma_f32 (dependent chain)", out, nsimd, mhz, 8);
    run<1>("v_pk_fma_f32 (4 independent)", out, nsimd, mhz, 8);
    run<2>("v_mul_f32 e32 (8 independent)", out, nsimd, mhz, 8);
    run<3>("v_cmp->vcc + v_cndmask", out, nsimd, mhz, 8);
    run<4>("4 v_cmp_e64 + 3 s_and + cndmask (8 instr)", out, nsimd, mhz, 8);
    run<5>("v_rcp_f32 (8 independent)", out, nsimd, mhz, 8);
    run<7>("v_mul_lo_u32", out, nsimd, mhz, 8);
    run<8>("v_mul_u32_u24", out, nsimd, mhz, 8);
    run<9>("v_xor_b32 / v_lshrrev_b32", out, nsimd, mhz, 8);
    run<10>("v_min3_f32", out, nsimd, mhz, 8);
    run<11>("v_pk_mul/add_f32, sgpr pair + op_sel", out, nsimd, mhz, 8);
    run<12>("3 v_cmpx + 2 v_mov + 3 v_add (8 VALU, 2 SALU)", out, nsimd, mhz, 8);
    run<13>("v_pk_mul_lo_u16", out, nsimd, mhz, 8);
    run<14>("v_pk_lshrrev_b16 / v_pk_add_u16", out, nsimd, mhz, 8);
    run<15>("v_perm_b32", out, nsimd, mhz, 8);
    run<16>("v_and_b32 / v_bitop3_b32", out, nsimd, mhz, 8);
    run<17>("hash chain: mul_lo, and, lshr, xor (dependent)", out, nsimd, mhz, 8);
    return 0;
}
