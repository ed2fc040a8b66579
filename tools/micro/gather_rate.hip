// gather_rate.hip — what a divergent BVH walk costs the CU's vector-memory path on gfx950, and whether fetching each
// lane's 64-byte record with QUADS of lanes (4 lanes x 16 B = one line per record) and handing it over through LDS is
// cheaper than four per-lane dwordx4 loads (64 lines per instruction).  Dependent chains like a traversal: the record
// holds the index of the next record.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_rate.hip -o /tmp/gather_rate && /tmp/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: per-lane, 4 x dwordx4 from the lane's own record
// MODE 1: quad fetch + LDS hand-over (two halves of 32 rays, 2 KiB of LDS)
// MODE 2: per-lane, ONE dwordx4 per visit (16-byte records: the floor of a per-lane walk)
template <int MODE>
__global__ __launch_bounds__(64) void k(const float4 *__restrict__ rec, uint32_t mask, int visits, float *out) {
    __shared__ float4 xb[128];
    const int lane = threadIdx.x;
    uint32_t idx = hash32(blockIdx.x * 64u + lane + 1u) & mask;
    float acc = 0.0f;
    for (int v = 0; v < visits; v++) {
        float4 n0, n1, n2, n3;
        if (MODE == 0) { const float4 *p = rec + 4 * (size_t)idx; n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; }
        else if (MODE == 2) { n0 = rec[4 * (size_t)idx + 3]; n1 = n2 = n0; n3 = n0; }
        else if (MODE == 3) { const float4 *p = rec + 4 * (size_t)idx; n0 = p[2]; n3 = p[3]; n1 = n2 = n0; }                 // 2 x dwordx4
        else if (MODE == 4) { const float4 *p = rec + 4 * (size_t)idx; n0 = p[1]; n1 = p[2]; n3 = p[3]; n2 = n0; }            // 3 x dwordx4
        else if (MODE == 5) { const float2 *p = (const float2 *)(rec + 4 * (size_t)idx); float2 a = p[0], b = p[2], c = p[4], d = p[7];   // 4 x dwordx2
                              n0 = make_float4(a.x, a.y, 0, 0); n1 = make_float4(b.x, b.y, 0, 0); n2 = make_float4(c.x, c.y, 0, 0); n3 = make_float4(0, 0, d.x, d.y); }
        else if (MODE == 6) { const float *p = (const float *)(rec + 4 * (size_t)idx); n0 = make_float4(p[0], 0, 0, 0); n1 = make_float4(0, p[5], 0, 0); n2 = make_float4(0, 0, p[10], 0); n3 = make_float4(0, 0, 0, p[15]); }   // 4 x dword
        else {
#pragma unroll
            for (int half = 0; half < 2; half++) {
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const int src = 32 * half + 16 * kk + (lane >> 2);
                    const uint32_t j = (uint32_t)__shfl((int)idx, src, 64);
                    xb[64 * kk + lane] = rec[4 * (size_t)j + (lane & 3)];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                if ((lane >> 5) == half) { const float4 *q = xb + 4 * (lane & 31); n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3]; }
                __builtin_amdgcn_wave_barrier();
            }
        }
        acc += n0.x + n1.y + n2.z;
        idx = hash32(idx + 0x9e3779b9u * (uint32_t)(v + 1) + __float_as_uint(n3.w)) & mask;   // the next record: known only now (and not a walk of a fixed random map, whose orbits merge)
    }
    out[blockIdx.x * 64 + lane] = acc;
}

// Round 3: (a) what does a per-lane 4 x dwordx4 fetch cost when only some lanes are ACTIVE (a BVH walk runs at a third of its lanes:
// is the vector-memory path charged per instruction or per active lane?), and (b) what would the same record cost from LDS
// (a per-wave cache of the top levels of the tree: 21 nodes = 1,344 B, divergent 64-byte-aligned ds_read_b128 x 4).
template <int STRIDE>
__global__ __launch_bounds__(64) void k_partial(const float4 *__restrict__ rec, uint32_t mask, int visits, float *out) {
    const int lane = threadIdx.x;
    uint32_t idx = hash32(blockIdx.x * 64u + lane + 1u) & mask;
    float acc = 0.0f;
    const bool active = (lane % STRIDE) == 0;
    for (int v = 0; v < visits; v++) {
        float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0, n3 = n0;
        if (active) { const float4 *p = rec + 4 * (size_t)idx; n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; }
        acc += n0.x + n1.y + n2.z;
        idx = hash32(idx + 0x9e3779b9u * (uint32_t)(v + 1) + __float_as_uint(n3.w)) & mask;
    }
    out[blockIdx.x * 64 + lane] = acc;
}
template <int NODES>
__global__ __launch_bounds__(64) void k_lds(const float4 *__restrict__ rec, int visits, float *out) {
    __shared__ float4 top[4 * NODES];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4 * NODES; i += 64) top[i] = rec[i];
    __syncthreads();
    uint32_t idx = hash32(blockIdx.x * 64u + lane + 1u) % NODES;
    float acc = 0.0f;
    for (int v = 0; v < visits; v++) {
        const float4 *p = top + 4 * idx;
        float4 n0 = p[0], n1 = p[1], n2 = p[2], n3 = p[3];
        acc += n0.x + n1.y + n2.z;
        idx = hash32(idx + 0x9e3779b9u * (uint32_t)(v + 1) + __float_as_uint(n3.w)) % NODES;
    }
    out[blockIdx.x * 64 + lane] = acc;
}
template <int STRIDE> static void run_partial(const float4 *rec, size_t nrec, float *out, int waves_per_cu) {
    const int visits = 400, blocks = 256 * waves_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_partial<STRIDE><<<blocks, 64>>>(rec, (uint32_t)(nrec - 1), 20, out);
    hipEventRecord(e0);
    k_partial<STRIDE><<<blocks, 64>>>(rec, (uint32_t)(nrec - 1), visits, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_visits = (double)blocks * visits;
    printf("per-lane 4 x dwordx4, %2d of 64 lanes active      %8zu records (%5.1f MiB) %2d waves/CU: %6.1f cycles of the CU per wave-visit (%5.2f per active lane)\n", 64 / STRIDE, nrec,
           nrec * 64 / 1048576.0, waves_per_cu, ms * 1e-3 * 2.4e9 / (wave_visits / 256), ms * 1e-3 * 2.4e9 / (wave_visits / 256) / (64 / STRIDE));
}
template <int NODES> static void run_lds(const float4 *rec, float *out, int waves_per_cu) {
    const int visits = 400, blocks = 256 * waves_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_lds<NODES><<<blocks, 64>>>(rec, 20, out);
    hipEventRecord(e0);
    k_lds<NODES><<<blocks, 64>>>(rec, visits, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_visits = (double)blocks * visits;
    printf("from LDS, 4 x ds_read_b128 per lane, %3d records (%5d B per wave)        %2d waves/CU: %6.1f cycles of the CU per wave-visit\n", NODES, NODES * 64, waves_per_cu,
           ms * 1e-3 * 2.4e9 / (wave_visits / 256));
}

template <int MODE> static void run(const char *name, const float4 *rec, size_t nrec, float *out, int waves_per_cu) {
    const int visits = 400, blocks = 256 * waves_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(rec, (uint32_t)(nrec - 1), 20, out);
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(rec, (uint32_t)(nrec - 1), visits, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_visits = (double)blocks * visits;
    printf("%-44s %8zu records (%5.1f MiB) %2d waves/CU: %7.2f G lane-visits/s  %6.0f ns per wave-visit  %6.1f cycles of the CU per wave-visit\n", name, nrec, nrec * 64 / 1048576.0,
           waves_per_cu, wave_visits * 64 / ms * 1e-6, ms * 1e6 / visits, ms * 1e-3 * 2.4e9 / (wave_visits / 256));
}

int main() {
    for (size_t nrec : {(size_t)1 << 14, (size_t)1 << 17, (size_t)1 << 21}) {        // 1 MiB (L2), 8 MiB, 128 MiB (Infinity Cache)
        std::vector<float4> h(4 * nrec);
        for (size_t i = 0; i < nrec; i++) {
            uint32_t nxt = (uint32_t)((i * 2654435761ull + 12345) ^ (i >> 3)) * 2246822519u;
            for (int k = 0; k < 4; k++) h[4 * i + k] = make_float4(1e-9f * k, 1e-9f, 1e-9f, 0.0f);
            float f; memcpy(&f, &nxt, 4); h[4 * i + 3].w = f;
        }
        float4 *rec; hipMalloc(&rec, h.size() * 16); hipMemcpy(rec, h.data(), h.size() * 16, hipMemcpyHostToDevice);
        float *out; hipMalloc(&out, 256 * 32 * 64 * 4);
        for (int w : {20}) {
            run<0>("per-lane 4 x dwordx4", rec, nrec, out, w);
            run<1>("quad fetch + LDS hand-over", rec, nrec, out, w);
            run<2>("per-lane 1 x dwordx4", rec, nrec, out, w);
            run<3>("per-lane 2 x dwordx4", rec, nrec, out, w);
            run<4>("per-lane 3 x dwordx4", rec, nrec, out, w);
            run<5>("per-lane 4 x dwordx2", rec, nrec, out, w);
            run<6>("per-lane 4 x dword", rec, nrec, out, w);
            run_partial<1>(rec, nrec, out, w); run_partial<2>(rec, nrec, out, w); run_partial<4>(rec, nrec, out, w); run_partial<8>(rec, nrec, out, w);
        }
        if (nrec == ((size_t)1 << 14)) { run_lds<5>(rec, out, 20); run_lds<21>(rec, out, 20); run_lds<85>(rec, out, 16); }
        hipFree(rec); hipFree(out);
    }
    return 0;
}
