// Cross-lane exchange primitives and the in-register bitonic network shared by the median-window kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace byz {
namespace lanes {

// ---- cross-lane exchange ------------------------------------------------------------------------
// value held by lane (lane ^ MASK); MASK is a compile-time constant after unrolling.
__device__ __forceinline__ float lane_xor(float v, int mask, int lane) {
    const int b = __float_as_int(v);
    int r;
    switch (mask) {
        case 1: r = __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
        case 2: r = __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
        case 3: r = __builtin_amdgcn_update_dpp(0, b, 0x1B, 0xF, 0xF, true); break;   // quad_perm [3,2,1,0]
        case 7: r = __builtin_amdgcn_update_dpp(0, b, 0x141, 0xF, 0xF, true); break;  // row_half_mirror
        case 8: r = __builtin_amdgcn_update_dpp(0, b, 0x128, 0xF, 0xF, true); break;  // row_ror:8
        case 15: r = __builtin_amdgcn_update_dpp(0, b, 0x140, 0xF, 0xF, true); break; // row_mirror
        case 4:   // xor 4 inside a 16-lane row without the LDS pipe: banks 0, 2 take lane + 4, banks 1, 3 lane - 4
            r = __builtin_amdgcn_update_dpp(0, b, 0x104, 0xF, 0x5, false);             // row_shl:4
            r = __builtin_amdgcn_update_dpp(r, b, 0x114, 0xF, 0xA, false);             // row_shr:4
            break;
        case 16: r = __builtin_amdgcn_ds_swizzle(b, 0x401F); break;                    // xor 16
        case 31: r = __builtin_amdgcn_ds_swizzle(b, 0x7C1F); break;                    // xor 31
        default: r = __builtin_amdgcn_ds_bpermute((lane ^ mask) << 2, b); break;       // 32, 63
    }
    return __int_as_float(r);
}

__device__ __forceinline__ void cmp_swap(float& lo, float& hi) {
    const float a = lo, b = hi;
    lo = __builtin_fminf(a, b);
    hi = __builtin_fmaxf(a, b);
}

// compile-time loops over powers of two (a `k <<= 1` loop is not reliably unrolled, and a register
// array indexed by a runtime value would be demoted to scratch)
template <int K, int KMAX, typename F>
__device__ __forceinline__ void for_pow2_up(F&& f) {
    if constexpr (K <= KMAX) {
        f(std::integral_constant<int, K>{});
        for_pow2_up<K * 2, KMAX>(f);
    }
}
template <int J, typename F>
__device__ __forceinline__ void for_pow2_down(F&& f) {
    if constexpr (J > 0) {
        f(std::integral_constant<int, J>{});
        for_pow2_down<J / 2>(f);
    }
}

// Sorts the 64*R values {x[r] of lane l} ascending in index i = r + R*l, for C independent columns.
template <int R, int C>
__device__ __forceinline__ void wave_bitonic_sort(float (&x)[C][R], int lane) {
    const float pinf = __builtin_inff();
    for_pow2_up<2, 64 * R>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        // flip: i <-> i ^ (k-1); the element whose bit (k/2) is clear keeps the minimum
        if constexpr (k <= R) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int p = r ^ (k - 1);
                if (p > r) {
#pragma unroll
                    for (int c = 0; c < C; ++c) cmp_swap(x[c][r], x[c][p]);
                }
            }
        } else {
            constexpr int lane_mask = k / R - 1;
            constexpr int lane_bit = k / (2 * R);
            const float sel = (lane & lane_bit) ? pinf : -pinf;  // upper partner keeps the maximum
#pragma unroll
            for (int r = 0; r < (R + 1) / 2; ++r) {
                const int p = R - 1 - r;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float from_p = lane_xor(x[c][p], lane_mask, lane);
                    if (p != r) {
                        const float from_r = lane_xor(x[c][r], lane_mask, lane);
                        x[c][p] = __builtin_amdgcn_fmed3f(x[c][p], from_r, sel);
                    }
                    x[c][r] = __builtin_amdgcn_fmed3f(x[c][r], from_p, sel);
                }
            }
        }
        // half-cleaners: i <-> i ^ j, j = k/4 ... 1
        for_pow2_down<k / 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j < R) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if ((r & j) == 0) {
#pragma unroll
                        for (int c = 0; c < C; ++c) cmp_swap(x[c][r], x[c][r | j]);
                    }
                }
            } else {
                constexpr int s = j / R;
                const float sel = (lane & s) ? pinf : -pinf;
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float other = lane_xor(x[c][r], s, lane);
                        x[c][r] = __builtin_amdgcn_fmed3f(x[c][r], other, sel);
                    }
                }
            }
        });
    });
}

// Sorts a BITONIC sequence of 64*R values ascending (same index order i = r + R*l): log2(64 R) half-cleaner steps
// instead of the full network.  |x - median| over values sorted by x is such a sequence (it falls, then rises).
template <int R, int C>
__device__ __forceinline__ void wave_bitonic_merge(float (&x)[C][R], int lane) {
    const float pinf = __builtin_inff();
    for_pow2_down<32 * R>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < R) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if ((r & j) == 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c) cmp_swap(x[c][r], x[c][r | j]);
                }
            }
        } else {
            constexpr int s = j / R;
            const float sel = (lane & s) ? pinf : -pinf;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float other = lane_xor(x[c][r], s, lane);
                    x[c][r] = __builtin_amdgcn_fmed3f(x[c][r], other, sel);
                }
            }
        }
    });
}

}  // namespace lanes
}  // namespace byz
