// Probe (round 6): CU masks per stream.  (i) where do the workgroups of a masked stream land (XCC id, SE, CU of every workgroup)?
// (ii) what does a 1 : 1 read / write stream reach on k CUs per XCD?  (iii) an MFMA-bound kernel on the other CUs, alone and with
// the stream kernel running beside it.
// Build: hipcc --offload-arch=gfx950 -O3 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void where_kernel(unsigned* __restrict__ hist) {
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
        atomicAdd(&hist[((xcc & 7u) * 8 + se) * 32 + sh * 16 + cu], 1u);
    }
    __builtin_amdgcn_s_sleep(100);
}

// persistent-ish streaming kernel: every workgroup walks its share of the buffer; 64 B in flight per thread
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long long n4) {
    const long long stride = static_cast<long long>(gridDim.x) * 256 * 4;
    for (long long i = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 4; i + 3 < n4; i += stride) {
        const f32x4 a = in[i], b = in[i + 1], c = in[i + 2], d = in[i + 3];
        out[i] = a * 2.0f;
        out[i + 1] = b * 2.0f;
        out[i + 2] = c * 2.0f;
        out[i + 3] = d * 2.0f;
    }
}

__global__ __launch_bounds__(512, 1) void mfma_kernel(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    f16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(tid * 16 + i) & 8191];
        b[i] = in[(tid * 16 + 8 + i) & 8191];
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + t) & 7], b[(i * 3 + t) & 7], acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + tid] = s;
}

static float elapsed(hipEvent_t a, hipEvent_t b) { float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::printf("CUs %d\n", n_cu);
    const int words = (n_cu + 31) / 32;
    // mask variants: bit b set <=> CU b usable
    auto make = [&](auto pred) { std::vector<uint32_t> m(words, 0u); for (int b = 0; b < n_cu; ++b) if (pred(b)) m[b / 32] |= 1u << (b % 32); return m; };
    unsigned* hist;
    CHECK(hipMalloc(&hist, 8 * 8 * 32 * 4));
    auto show = [&](const char* name, hipStream_t s) {
        CHECK(hipMemsetAsync(hist, 0, 8 * 8 * 32 * 4, s));
        where_kernel<<<4096, 256, 0, s>>>(hist);
        CHECK(hipStreamSynchronize(s));
        std::vector<unsigned> h(8 * 8 * 32);
        CHECK(hipMemcpy(h.data(), hist, h.size() * 4, hipMemcpyDeviceToHost));
        std::printf("%s: workgroups per XCC:", name);
        int cus_used = 0;
        for (int x = 0; x < 8; ++x) { unsigned t = 0; int c = 0; for (int i = 0; i < 256; ++i) { t += h[x * 256 + i]; c += h[x * 256 + i] != 0; } std::printf(" %u (%d CUs)", t, c); cus_used += c; }
        std::printf("  -> %d CUs in all\n", cus_used);
    };
    hipStream_t s_all, s_small, s_big, s_first16, s_mod16;
    CHECK(hipStreamCreate(&s_all));
    // hypothesis 1: bit b -> XCC b % 8 (round-robin) ; hypothesis 2: bit b -> XCC b / 32
    auto m_small = make([&](int b) { return (b / 8) % 16 == 0; });          // bits 0..7, 128..135 (16 CUs)
    auto m_big = make([&](int b) { return (b / 8) % 16 != 0; });
    auto m_first16 = make([&](int b) { return b < 16; });
    auto m_mod16 = make([&](int b) { return b % 16 == 0; });
    CHECK(hipExtStreamCreateWithCUMask(&s_small, words, m_small.data()));
    CHECK(hipExtStreamCreateWithCUMask(&s_big, words, m_big.data()));
    CHECK(hipExtStreamCreateWithCUMask(&s_first16, words, m_first16.data()));
    CHECK(hipExtStreamCreateWithCUMask(&s_mod16, words, m_mod16.data()));
    show("no mask", s_all);
    show("bits 0-7 + 128-135", s_small);
    show("complement of that", s_big);
    show("bits 0-15", s_first16);
    show("bits 0, 16, 32, ...", s_mod16);

    const long long n4 = (8ll << 30) / 16;      // 8 GiB in, 8 GiB out
    f32x4 *in, *out;
    CHECK(hipMalloc(&in, n4 * 16));
    CHECK(hipMalloc(&out, n4 * 16));
    CHECK(hipMemset(in, 1, n4 * 16));
    f16x8* mi;
    float* mo;
    CHECK(hipMalloc(&mi, 8192 * 16));
    CHECK(hipMemset(mi, 0x3c, 8192 * 16));
    CHECK(hipMalloc(&mo, 4096 * 512 * 4));
    hipEvent_t e0, e1, e2, e3;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2)); CHECK(hipEventCreate(&e3));
    auto time_stream = [&](const char* name, hipStream_t s, int wgs) {
        stream_kernel<<<wgs, 256, 0, s>>>(in, out, n4);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipEventRecord(e0, s));
        stream_kernel<<<wgs, 256, 0, s>>>(in, out, n4);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        const float ms = elapsed(e0, e1);
        std::printf("stream kernel, %-22s %5d workgroups: %8.3f ms = %.2f TB/s (read + write)\n", name, wgs, ms, 2.0 * n4 * 16 / ms / 1e9);
    };
    time_stream("no mask", s_all, 8192);
    time_stream("no mask", s_all, 2048);
    time_stream("16-CU mask (0-7,128-135)", s_small, 128);
    time_stream("16-CU mask (0-7,128-135)", s_small, 256);
    time_stream("16-CU mask (0-7,128-135)", s_small, 1024);
    time_stream("16-CU mask (mod 16)", s_mod16, 256);
    time_stream("16-CU mask (mod 16)", s_mod16, 1024);
    time_stream("16-CU mask (first 16)", s_first16, 256);
    const int iters = 20000;
    auto time_mfma = [&](const char* name, hipStream_t s, int wgs, bool beside) {
        mfma_kernel<<<wgs, 512, 0, s>>>(mi, mo, 2000);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, s));
        mfma_kernel<<<wgs, 512, 0, s>>>(mi, mo, iters);
        CHECK(hipEventRecord(e1, s));
        if (beside) {
            CHECK(hipEventRecord(e2, s_small));
            stream_kernel<<<1024, 256, 0, s_small>>>(in, out, n4);
            CHECK(hipEventRecord(e3, s_small));
        }
        CHECK(hipDeviceSynchronize());
        const float ms = elapsed(e0, e1);
        std::printf("mfma kernel, %-28s %5d workgroups: %8.3f ms = %.0f TF", name, wgs, ms, 2.0 * wgs * 8 * 48.0 * iters * 8192 / ms / 1e9);
        if (beside) std::printf("   | the stream kernel beside it on the 16-CU stream: %.3f ms = %.2f TB/s", elapsed(e2, e3), 2.0 * n4 * 16 / elapsed(e2, e3) / 1e9);
        std::printf("\n");
    };
    time_mfma("no mask", s_all, 2048, false);
    time_mfma("240-CU mask", s_big, 1920, false);
    time_mfma("240-CU mask + stream beside", s_big, 1920, true);
    time_mfma("no mask + stream beside", s_all, 2048, true);
    time_mfma("no mask", s_all, 2048, false);
    return 0;
}
