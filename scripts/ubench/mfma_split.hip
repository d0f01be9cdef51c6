// Microbenchmark: how fast can one CU run the bf16 x 3 split contraction out of LDS-resident fp32 tiles?
// No global traffic inside the loop: this isolates the {ds_read, split, MFMA} inner loop of gram.hip so that wave
// tile shapes, occupancies and instruction interleaves can be compared before the real kernel is rebuilt.
//   MODE 0  split everything, then all MFMAs (what gram.hip does today), compiler's own schedule
//   MODE 1  software pipelined: the NEXT step's fragments are read and split between this step's MFMAs
//           (sched_group_barrier pins 1 MFMA : ~5.5 VALU)
//   MODE 2  MFMAs only (planes computed once): the ceiling of the MFMA pipe at this power/clock
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -pragma-unroll-threshold=1000000 mfma_split.hip -o mfma_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const f32x4& lo4, const f32x4& hi4, bf16x8& h, bf16x8& m, bf16x8& l) {
    uint32_t xb[8], r1b[8], r2b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? lo4[e] : hi4[e - 4];
        xb[e] = __float_as_uint(x);
        const float r1 = x - __uint_as_float(xb[e] & 0xffff0000u);
        r1b[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(r1b[e] & 0xffff0000u);
        r2b[e] = __float_as_uint(r2);
    }
    u32x4 hp, mp, lp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hp[e] = __builtin_amdgcn_perm(xb[2 * e + 1], xb[2 * e], 0x07060302u);
        mp[e] = __builtin_amdgcn_perm(r1b[2 * e + 1], r1b[2 * e], 0x07060302u);
        lp[e] = __builtin_amdgcn_perm(r2b[2 * e + 1], r2b[2 * e], 0x07060302u);
    }
    h = __builtin_bit_cast(bf16x8, hp);
    m = __builtin_bit_cast(bf16x8, mp);
    l = __builtin_bit_cast(bf16x8, lp);
}

__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2); }

// 1 MFMA : V VALU, N times, with V spread so that the groups add up to KV VALU instructions
template <int I, int N, int KV>
struct Interleave {
    static __device__ __forceinline__ void emit() {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, (KV * (I + 1)) / N - (KV * I) / N, 0);
        Interleave<I + 1, N, KV>::emit();
    }
};
template <int N, int KV>
struct Interleave<N, N, KV> {
    static __device__ __forceinline__ void emit() {}
};

template <int MB, int NB, int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void kern(const float* __restrict__ in, float* __restrict__ out, int iters, long long* __restrict__ clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RA = MB * 64, RB = NB * 64;   // 2 x 2 waves
    for (int i = threadIdx.x; i < (RA + RB) * 32; i += 256) lds[i] = in[i];
    __syncthreads();
    const float* A = lds;
    const float* B = lds + RA * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int frag_row = lane & 31, frag_half = lane >> 5;

    const long long c0 = clock64(), w0 = wall_clock64();
    f32x16 acc[MB][NB], acc2[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[m][n][e] = 0.0f; acc2[m][n][e] = 0.0f; }

    bf16x8 ap[2][3][MB], bp[2][3][NB];
    f32x4 rawA[MB][2], rawB[NB][2];
    auto read_raw = [&](int j) __attribute__((always_inline)) {
        asm volatile("" ::: "memory");   // the tiles are loop invariant here: keep the reads inside the loop
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int row = wr * (MB * 32) + m * 32 + frag_row;
            rawA[m][0] = *reinterpret_cast<const f32x4*>(A + tile_off(row, 4 * j + 2 * frag_half));
            rawA[m][1] = *reinterpret_cast<const f32x4*>(A + tile_off(row, 4 * j + 2 * frag_half + 1));
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int row = wc * (NB * 32) + n * 32 + frag_row;
            rawB[n][0] = *reinterpret_cast<const f32x4*>(B + tile_off(row, 4 * j + 2 * frag_half));
            rawB[n][1] = *reinterpret_cast<const f32x4*>(B + tile_off(row, 4 * j + 2 * frag_half + 1));
        }
    };
    auto convert = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MB; ++m) split8(rawA[m][0], rawA[m][1], ap[set][0][m], ap[set][1][m], ap[set][2][m]);
#pragma unroll
        for (int n = 0; n < NB; ++n) split8(rawB[n][0], rawB[n][1], bp[set][0][n], bp[set][1][n], bp[set][2][n]);
    };
    auto mfmas = [&](int set) __attribute__((always_inline)) {
        constexpr int pa[6] = {2, 0, 1, 1, 0, 0};
        constexpr int pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[set][pa[t]][m], bp[set][pb[t]][n], acc[m][n], 0, 0, 0);
    };
    auto level1 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc2[m][n][e] += acc[m][n][e]; acc[m][n][e] = 0.0f; }
    };

    // 16 steps (256 columns) per level-0 chain, as in gram.hip's split mode
    if constexpr (MODE == 0 || MODE == 3 || MODE == 4) {
        if constexpr (MODE == 4) {
            // static priority: of the two waves that share a SIMD, the one in the odd hardware slot always wins
            const unsigned hw_id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);   // HW_ID[3:0] = wave slot
            if (hw_id & 1u) __builtin_amdgcn_s_setprio(1);
        }
        for (int it = 0; it < iters; it += 16) {
            for (int pair = 0; pair < 8; ++pair) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    read_raw(j);
                    convert(0);
                    if constexpr (MODE == 3) __builtin_amdgcn_s_setprio(1);
                    mfmas(0);
                    if constexpr (MODE == 3) __builtin_amdgcn_s_setprio(0);
                }
            }
            level1();
        }
    } else if constexpr (MODE == 1) {
        constexpr int kMfma = 6 * MB * NB;
        constexpr int kValu = (MB + NB) * 44;
        read_raw(0); convert(0);
        for (int it = 0; it < iters; it += 16) {
            for (int pair = 0; pair < 8; ++pair) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    read_raw(half ^ 1);
                    mfmas(half);
                    convert(half ^ 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MB + NB), 0);
                    Interleave<0, kMfma, kValu>::emit();
                }
            }
            level1();
        }
    } else {
        read_raw(0); convert(0);
        for (int it = 0; it < iters; it += 16) {
            for (int pair = 0; pair < 8; ++pair) {
                mfmas(0);
                mfmas(0);
            }
            level1();
        }
    }
    // keep everything alive
    float s = 0.0f;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc2[m][n][e] + acc[m][n][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = clock64() - c0;
        clk[1] = wall_clock64() - w0;
    }
}

template <int MB, int NB, int MODE, int OCC>
void run(const char* name, const float* in, float* out, int iters, long long* clk) {
    const int grid = 256 * OCC;
    const size_t lds = (MB + NB) * 64 * 32 * sizeof(float);
    auto fn = kern<MB, NB, MODE, OCC>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // pad LDS so that exactly OCC workgroups fit a CU
    const size_t pad = OCC == 1 ? 100 * 1024 : (OCC == 2 ? 64 * 1024 : lds);
    const size_t lds_bytes = lds > pad ? lds : pad;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds_bytes, 0, in, out, 64, clk);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds_bytes, 0, in, out, iters, clk);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double flops = double(grid) * 4 * iters * (6.0 * MB * NB) * 32768.0;
    long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const int wg_per_simd = OCC;   // 4 waves per workgroup, one per SIMD
    const double mfma_cycles = double(iters) * (6.0 * MB * NB) * 32.0 * wg_per_simd;
    printf("%-44s %8.3f ms  %7.1f TF bf16 (%4.1f%% of 2.5 PF)  shader clock %4.0f MHz, matrix pipe busy %4.1f%% of cycles\n", name,
           best, flops / best / 1e9, flops / best / 1e9 / 25.0, double(h[0]) / double(h[1]) * 100.0,
           100.0 * mfma_cycles / double(h[0]));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::vector<float> h(384 * 32 * 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = float((i * 2654435761u) % 2000u) / 1000.0f - 1.0f;
    float *in, *out;
    hipMalloc(&in, h.size() * 4);
    hipMalloc(&out, 256 * 4 * 256 * 4);
    if (argc > 2) for (auto& v : h) v = 0.0f;   // zero operands: least switching
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    long long* clk;
    hipMalloc(&clk, 16);
    run<2, 2, 2, 2>("2x2 blocks, 2 WG/CU, MFMA only", in, out, iters, clk);
    run<2, 2, 0, 2>("2x2 blocks, 2 WG/CU, split then MFMA", in, out, iters, clk);
    run<2, 2, 1, 2>("2x2 blocks, 2 WG/CU, pipelined", in, out, iters, clk);
    run<2, 2, 3, 2>("2x2 blocks, 2 WG/CU, setprio around MFMAs", in, out, iters, clk);
    run<2, 2, 4, 2>("2x2 blocks, 2 WG/CU, static prio by slot", in, out, iters, clk);
    run<4, 2, 2, 1>("4x2 blocks, 1 WG/CU, MFMA only", in, out, iters, clk);
    run<4, 2, 0, 1>("4x2 blocks, 1 WG/CU, split then MFMA", in, out, iters, clk);
    run<4, 2, 1, 1>("4x2 blocks, 1 WG/CU, pipelined", in, out, iters, clk);
    return 0;
}
