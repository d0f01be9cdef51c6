// Microbenchmark: issue cost of the counting idioms on gfx950.
//   A: v_cmp_le_f32 -> s_bcnt1 -> s_add      (1 VALU + 2 SALU per 64 values)
//   B: v_cmp_le_f32 -> v_addc_co_u32          (2 VALU per 64 values)
//   C: v_fma only (reference VALU rate)       D: v_min+v_max
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int R = 16;
template <int MODE>
__global__ __launch_bounds__(512, 4) void k(const float* in, int* out, int iters, float T0) {
    float x[R];
    for (int j = 0; j < R; ++j) x[j] = in[threadIdx.x + 512 * j];
    int cs = 0; int cv = 0; float T = T0; float acc = 0.f; float mn = 1e30f, mx = -1e30f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < R; ++j) cs += __popcll(__ballot(x[j] <= T));
            T += (cs & 1) ? 0.001f : 0.002f;
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < R; ++j) cv += (x[j] <= T) ? 1 : 0;
            T += 0.001f;
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < R; ++j) acc = __builtin_fmaf(x[j], T, acc);
            T += 0.001f;
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < R; ++j) { mn = __builtin_fminf(mn, x[j] + T); }
            T += 0.001f;
        } else if (MODE == 5) {  // A with 4 independent ballots in flight
#pragma unroll
            for (int j = 0; j < R; j += 4) {
                const unsigned long long m0 = __ballot(x[j] <= T), m1 = __ballot(x[j + 1] <= T),
                                         m2 = __ballot(x[j + 2] <= T), m3 = __ballot(x[j + 3] <= T);
                cs += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
            }
            T += (cs & 1) ? 0.001f : 0.002f;
        } else if (MODE == 6) {  // compare-free flag: sat((T - x) * H + 1), accumulated in fp32
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const float d = T - x[j];
                acc += __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(d, 1e30f, 1.0f), 0.0f), 1.0f);
            }
            T += 0.001f;
        } else if (MODE == 7) {  // the same on packed fp32 (two values per instruction)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 acc2 = {0.f, 0.f};
            const f2 T2 = {T, T}, H2 = {1e30f, 1e30f}, one2 = {1.f, 1.f};
#pragma unroll
            for (int j = 0; j < R; j += 2) {
                f2 x2 = {x[j], x[j + 1]}, d2, f;
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d2) : "v"(T2), "v"(x2));
                asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(f) : "v"(d2), "v"(H2), "v"(one2));
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(acc2) : "v"(acc2), "v"(f));
            }
            acc += acc2.x + acc2.y;
            T += 0.001f;
        } else if (MODE == 8) {  // integer sign-bit accumulate: d = T - x ; cnt -= (d >> 31)
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const float d = T - x[j];
                cv -= (__float_as_int(d) >> 31);
            }
            T += 0.001f;
        } else if (MODE == 4) {  // half A half B
#pragma unroll
            for (int j = 0; j < R; j += 2) { cs += __popcll(__ballot(x[j] <= T)); cv += (x[j + 1] <= T) ? 1 : 0; }
            T += (cs & 1) ? 0.001f : 0.002f;
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = cs + cv + __float_as_int(acc) + __float_as_int(mn) + __float_as_int(mx);
}
template <int MODE> void run(const char* name, const float* in, int* out, int instr_per_iter) {
    const int iters = 2000, blocks = 256 * 2 * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 512>>>(in, out, 10, 0.f);
    hipEventRecord(a);
    k<MODE><<<blocks, 512>>>(in, out, iters, 0.f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // waves per SIMD-slot: blocks*8 waves over 1024 SIMDs
    double waves_per_simd = blocks * 8.0 / 1024.0;
    double cyc = ms * 1e-3 * 2.4e9 / (waves_per_simd * iters * R);
    printf("%-28s %8.3f ms  -> %.2f cycles(@2.4GHz) per wave per 64-value step (%d instr)\n", name, ms, cyc, instr_per_iter);
}
int main() {
    float* in; int* out; hipMalloc(&in, 512 * R * 4); hipMalloc(&out, 256 * 8 * 512 * 4);
    std::vector<float> h(512 * R); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) / 977.f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("A cmp+s_bcnt+s_add", in, out, 3);
    run<1>("B cmp+v_addc", in, out, 2);
    run<2>("C v_fma", in, out, 1);
    run<3>("D v_add+v_min", in, out, 2);
    run<4>("E half A half B", in, out, 2);
    run<5>("F A, 4 ballots in flight", in, out, 3);
    run<6>("G sub+fma_clamp+add", in, out, 3);
    run<7>("H packed G (1.5/value)", in, out, 2);
    run<8>("I sub+ashr+sub", in, out, 3);
    return 0;
}
