// Probe (round 6): what shape of a plain copy kernel reaches the 6.3 TB/s (read + write) the copy engine's blit reaches?
// Variants: bytes in flight per thread (U x 16 B loaded before the first store), workgroups per launch, non-temporal loads / stores,
// the workgroup's span contiguous (each workgroup owns one block of the buffer) or interleaved (grid-stride).
// Build: hipcc --offload-arch=gfx950 -O3 copy_shapes.hip -o copy_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, int NT, bool BLOCKED>
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long long n) {
    // n = number of 16-byte elements; a "row" of work = 256 threads x U elements, consecutive threads on consecutive elements
    const long long rows = n / (256ll * U);
    long long r0, r1, step;
    if (BLOCKED) {
        const long long per = (rows + gridDim.x - 1) / gridDim.x;
        r0 = blockIdx.x * per;
        r1 = r0 + per < rows ? r0 + per : rows;
        step = 1;
    } else {
        r0 = blockIdx.x;
        r1 = rows;
        step = gridDim.x;
    }
    for (long long r = r0; r < r1; r += step) {
        const long long base = r * 256 * U + threadIdx.x;
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (NT & 1) ? __builtin_nontemporal_load(in + base + 256 * u) : in[base + 256 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT & 2) __builtin_nontemporal_store(v[u], out + base + 256 * u);
            else out[base + 256 * u] = v[u];
        }
    }
}

int main() {
    const long long bytes = 8ll << 30;
    const long long n = bytes / 16;
    f32x4 *in, *out;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMemset(in, 1, bytes));
    CHECK(hipMemset(out, 2, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto report = [&](const char* name, float ms) { std::printf("%-64s %8.3f ms = %.2f TB/s (read + write)\n", name, ms, 2.0 * bytes / ms / 1e9); };
    {
        CHECK(hipMemcpy(out, in, bytes, hipMemcpyDeviceToDevice));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 3; ++i) CHECK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0));
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        report("hipMemcpyAsync device to device", ms / 3);
    }
#define RUN(U, NT, BL, WGS)                                                                            \
    do {                                                                                               \
        copy_kernel<U, NT, BL><<<WGS, 256>>>(in, out, n);                                              \
        CHECK(hipDeviceSynchronize());                                                                 \
        CHECK(hipEventRecord(e0));                                                                     \
        for (int i = 0; i < 3; ++i) copy_kernel<U, NT, BL><<<WGS, 256>>>(in, out, n);                  \
        CHECK(hipEventRecord(e1));                                                                     \
        CHECK(hipDeviceSynchronize());                                                                 \
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));                                             \
        char name[128];                                                                                \
        std::snprintf(name, sizeof name, "U = %2d (%4d B in flight per thread) nt = %d %s %6d workgroups", U, U * 16, NT, BL ? "blocked    " : "grid-stride", WGS); \
        report(name, ms / 3);                                                                          \
    } while (0)
    RUN(1, 0, false, 8192); RUN(4, 0, false, 8192); RUN(8, 0, false, 8192); RUN(16, 0, false, 8192);
    RUN(4, 0, false, 2048); RUN(4, 0, false, 1024); RUN(4, 0, false, 512); RUN(8, 0, false, 2048); RUN(8, 0, false, 1024); RUN(16, 0, false, 2048); RUN(16, 0, false, 1024);
    RUN(4, 1, false, 2048); RUN(4, 2, false, 2048); RUN(4, 3, false, 2048); RUN(8, 3, false, 2048); RUN(16, 3, false, 2048);
    RUN(4, 0, true, 2048); RUN(8, 0, true, 2048); RUN(4, 0, true, 8192); RUN(4, 3, true, 2048); RUN(4, 0, true, 65536); RUN(4, 0, false, 65536);
    RUN(4, 0, false, 2040); RUN(4, 0, false, 4096); RUN(2, 0, false, 4096);
    return 0;
}
