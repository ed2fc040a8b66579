// atomic_rate.hip — throughput of no-return global float atomics on gfx950 by access shape.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/atomic_rate.hip -o /tmp/atomic_rate && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// SHAPE 0: each instruction = 4 random 64-B cells x 16 floats (the kernel's staging-cell scatter)
// SHAPE 1: each instruction = 64 random cells x 1 float
// SHAPE 2: each instruction = 1 random 256-B block x 64 floats
// SHAPE 3: as 0, but only lanes 0..3 of each 16 active (4 floats per cell: what a direct (texel, channel) scatter of one corner would do)
template <int SHAPE>
__global__ __launch_bounds__(64) void k(float *buf, uint32_t ncells_mask, int iters) {
    const int lane = threadIdx.x;
    uint32_t seed = (blockIdx.x * 64u + 1u) * 2654435761u;
    for (int i = 0; i < iters; i++) {
        seed = hash32(seed + i);
        uint32_t cell;
        if (SHAPE == 0 || SHAPE >= 3) cell = hash32(seed + (lane >> 4)) & ncells_mask;
        else if (SHAPE == 1) cell = hash32(seed + lane) & ncells_mask;
        else cell = (hash32(seed) & ncells_mask) & ~3u;
        float *p = buf + 16 * (size_t)cell + ((SHAPE == 1) ? 0 : (SHAPE == 2 ? lane : (lane & 15)));
        if (SHAPE == 3 && (lane & 15) >= 4) continue;
        if (SHAPE == 4) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // as shape 0, workgroup scope
        else if (SHAPE == 5) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // as shape 0, wavefront scope
        else if (SHAPE == 6) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // as shape 0, system scope
        else unsafeAtomicAdd(p, 1.0f);
    }
}

template <int SHAPE> static void run(const char *name, float *buf, size_t ncells, int dwords_per_instr) {
    for (size_t cells : {ncells, (size_t)16384}) {
        int iters = 2000, blocks = 256 * 12;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<SHAPE><<<blocks, 64>>>(buf, (uint32_t)(cells - 1), 50);
        hipEventRecord(e0);
        k<SHAPE><<<blocks, 64>>>(buf, (uint32_t)(cells - 1), iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr = (double)blocks * iters;
        printf("%-46s %9zu cells: %7.2f G instr/s  %8.2f G dwords/s  (%.1f cycles per instr per CU)\n", name, cells, instr / ms * 1e-6,
               instr * dwords_per_instr / ms * 1e-6, ms * 1e-3 * 2.4e9 / (instr / 256));
    }
}

int main() {
    size_t ncells = 1u << 20;                       // 64 MiB of 64-byte cells (the 1024^2 texture's staging cells)
    float *buf; hipMalloc(&buf, ncells * 64); hipMemset(buf, 0, ncells * 64);
    run<0>("4 cells x 16 floats per instruction", buf, ncells, 64);
    run<3>("4 cells x 4 floats per instruction", buf, ncells, 16);
    run<1>("64 cells x 1 float per instruction", buf, ncells, 64);
    run<2>("1 block x 64 floats per instruction", buf, ncells, 64);
    run<4>("4 cells x 16 floats, WORKGROUP scope", buf, ncells, 64);
    run<5>("4 cells x 16 floats, WAVEFRONT scope", buf, ncells, 64);
    run<6>("4 cells x 16 floats, SYSTEM scope", buf, ncells, 64);
    return 0;
}
