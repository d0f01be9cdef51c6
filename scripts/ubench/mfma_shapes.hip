// Microbenchmark (round 6): does the SHAPE of the fp16 MFMA matter at the socket's power cap?  The same f16x2 arithmetic --
// three products m h' + h m' + h h' per block and K step -- on a 64 x 64 wave tile, operands in registers (no LDS, no global
// traffic inside the loop), random operand bits, 8 waves per workgroup, one workgroup per CU, level-1 flush every 256 columns:
//   SHAPE 0   v_mfma_f32_32x32x16_f16: 2 x 2 blocks, 4 operand fragments per plane and K = 16   (gram_planes.hip today)
//   SHAPE 1   v_mfma_f32_16x16x32_f16: 4 x 4 blocks, 8 operand fragments per plane and K = 32
// 16x16x32 reads and writes HALF the accumulator registers per MAC and TWICE the operand registers.  Both issue the same
// MACs per iteration (64 x 64 x 32 x 3).  Reports ms, TF of MFMA work and the clock (cycles / time).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -pragma-unroll-threshold=1000000 mfma_shapes.hip -o mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// FEED 0: the operand fragments stay in their registers (the MFMAs alone); FEED 1: every K = 32 step re-reads its 16 fragments
// (1 KiB each per wave: the same bytes for both shapes) from a 64 KiB pool in LDS at a moving offset -- the operands toggle and
// the LDS reads draw their power, as in gram_planes.hip.
template <int SHAPE, int FEED>
__global__ __launch_bounds__(512, 1) void kern(const f16x8* __restrict__ in, float* __restrict__ out, int iters, long long* __restrict__ clk) {
    __shared__ f16x8 pool[4096];     // 64 KiB
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 512) pool[i] = in[i];
    __syncthreads();
    const int lane = tid & 63;
    const long long t0 = __builtin_readcyclecounter();
    float total = 0.0f;
    if constexpr (SHAPE == 0) {
        f16x8 ah[2][2], am[2][2], bh[2][2], bm[2][2];     // [k half of 32][block]
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                ah[k][b] = in[(tid * 16 + k * 8 + b * 4 + 0) & 8191];
                am[k][b] = in[(tid * 16 + k * 8 + b * 4 + 1) & 8191];
                bh[k][b] = in[(tid * 16 + k * 8 + b * 4 + 2) & 8191];
                bm[k][b] = in[(tid * 16 + k * 8 + b * 4 + 3) & 8191];
            }
        f32x16 acc[2][2], acc2[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][n][e] = acc2[m][n][e] = 0.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {              // 8 x 32 = 256 columns, then the flush
#pragma unroll
                for (int k = 0; k < 2; ++k) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[k][m], bh[k][n], acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k][m], bm[k][n], acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k][m], bh[k][n], acc[m][n], 0, 0, 0);
                }
                // (keep the operands opaque so that nothing is hoisted or folded)
                if constexpr (FEED == 1) {
                    const int base = ((it * 8 + s) * 16 * 64) & 4095;
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            ah[k][b] = pool[(base + ((k * 2 + b) * 4 + 0) * 64 + lane) & 4095];
                            am[k][b] = pool[(base + ((k * 2 + b) * 4 + 1) * 64 + lane) & 4095];
                            bh[k][b] = pool[(base + ((k * 2 + b) * 4 + 2) * 64 + lane) & 4095];
                            bm[k][b] = pool[(base + ((k * 2 + b) * 4 + 3) * 64 + lane) & 4095];
                        }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(ah[k][b]), "+v"(am[k][b]), "+v"(bh[k][b]), "+v"(bm[k][b]));
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        acc2[m][n][e] += acc[m][n][e];
                        acc[m][n][e] = 0.0f;
                    }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) total += acc2[m][n][e];
    } else {
        f16x8 ah[4], am[4], bh[4], bm[4];                   // [16-row block], K = 32 per fragment
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            ah[b] = in[(tid * 16 + b * 4 + 0) & 8191];
            am[b] = in[(tid * 16 + b * 4 + 1) & 8191];
            bh[b] = in[(tid * 16 + b * 4 + 2) & 8191];
            bm[b] = in[(tid * 16 + b * 4 + 3) & 8191];
        }
        f32x4 acc[4][4], acc2[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[m][n][e] = acc2[m][n][e] = 0.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bm[n], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                if constexpr (FEED == 1) {
                    const int base = ((it * 8 + s) * 16 * 64) & 4095;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        ah[b] = pool[(base + (b * 4 + 0) * 64 + lane) & 4095];
                        am[b] = pool[(base + (b * 4 + 1) * 64 + lane) & 4095];
                        bh[b] = pool[(base + (b * 4 + 2) * 64 + lane) & 4095];
                        bm[b] = pool[(base + (b * 4 + 3) * 64 + lane) & 4095];
                    }
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(ah[b]), "+v"(am[b]), "+v"(bh[b]), "+v"(bm[b]));
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc2[m][n][e] += acc[m][n][e];
                        acc[m][n][e] = 0.0f;
                    }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) total += acc2[m][n][e];
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + tid] = total;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 4000;
    const char* fill = argc > 2 ? argv[2] : "random";
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int wgs = prop.multiProcessorCount;
    std::vector<_Float16> host(8192 * 8);
    unsigned s = 12345u;
    for (auto& v : host) {
        s = s * 1664525u + 1013904223u;
        const float x = (static_cast<int>(s >> 8) % 65536 - 32768) / 32768.0f * 700.0f;     // full-range mantissas, |x| < 700
        v = static_cast<_Float16>(fill[0] == 'z' ? 0.0f : (fill[0] == 'o' ? 1.0f : x));
    }
    f16x8* in;
    float* out;
    long long* clk;
    CHECK(hipMalloc(&in, host.size() * sizeof(_Float16)));
    CHECK(hipMalloc(&out, static_cast<size_t>(wgs) * 512 * sizeof(float)));
    CHECK(hipMalloc(&clk, static_cast<size_t>(wgs) * sizeof(long long)));
    CHECK(hipMemcpy(in, host.data(), host.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep)
        for (int feed = 0; feed < 2; ++feed)
        for (int shape = 0; shape < 2; ++shape) {
            for (int warm = 0; warm < 2; ++warm) {
                if (shape == 0 && feed == 0) kern<0, 0><<<wgs, 512>>>(in, out, iters, clk);
                else if (shape == 1 && feed == 0) kern<1, 0><<<wgs, 512>>>(in, out, iters, clk);
                else if (shape == 0) kern<0, 1><<<wgs, 512>>>(in, out, iters, clk);
                else kern<1, 1><<<wgs, 512>>>(in, out, iters, clk);
                if (warm == 0) { CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(a)); }
            }
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, a, b));
            std::vector<long long> c(wgs);
            CHECK(hipMemcpy(c.data(), clk, wgs * sizeof(long long), hipMemcpyDeviceToHost));
            double cyc = 0;
            for (long long v : c) cyc += static_cast<double>(v);
            cyc /= wgs;
            const double flops = 2.0 * 64 * 64 * 32 * 3 * 8.0 * iters * 8 * wgs;     // per launch
            std::printf("%s data, %s, %s: %.3f ms, %.0f TF of fp16 MFMA (%.3f of 2.5 PF), %.0f cycles per workgroup = %.2f GHz\n", fill,
                        shape == 0 ? "32x32x16" : "16x16x32", feed == 0 ? "operands in registers" : "operands re-read from LDS", ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15, cyc,
                        cyc / (ms * 1e-3) / 1e9);
        }
    return 0;
}
