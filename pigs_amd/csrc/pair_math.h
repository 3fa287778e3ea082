// Per-(sample point, Gaussian) arithmetic shared by every kernel of the sampler.
//
// Semantics follow the reference's dense PyTorch twin of its CUDA sampler:
//   order 0  gaussians.sample_gaussians      /root/reference/gaussians.py:48-58
//   order 1  gaussians.gaussian_derivative   /root/reference/gaussians.py:89-101
//   order 2  gaussians.gaussian_derivative2  /root/reference/gaussians.py:103-116 (full Hessian)
//   order 3  its derivative wrt the sample point (shape n,d,d,d,c: model_pn.py:654)
// with  x = s - mu,  p = C x,  q = x.p,  g = exp(-q/2),  w_c = v_c g.
//
// Derivative outputs are symmetric in their derivative indices, so accumulators keep only the
// distinct components (D=2: xx,xy,yy and xxx,xxy,xyy,yyy) and expand on store.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pigs {

constexpr int ORD0 = 1, ORD1 = 2, ORD2 = 4, ORD3 = 8;
// trace of the Hessian (the Laplacian the PDE residuals consume, model_pn.py:614-617) instead of
// the full Hessian: one value per channel in the order-2 slot; never together with ORD2
constexpr int ORD2T = 16;
// A LINEAR RESIDUAL of the sampled field as the only output (mask == ORDR, alone):
//   r[m][c] = a0 u + a1 . grad u + aL (u_xx + u_yy) - target[m][c]
// -- the diffusion / wave residuals of the reference's losses (model_pn.py:612-617, 834-849;
// test_no_mlp.py:127-144: u_t - D lap u with u_t = (u - u_prev) / dt) in one launch and 4 B per point and
// channel instead of u, grad u and the Hessian (28 B).  Its backward is the backward of orders 0, 1 and
// the trace with the incoming gradients a0 gr, a1 gr, aL gr formed on the fly.
constexpr int ORDR = 32;
constexpr int ORDR_AS = ORD0 | ORD1 | ORD2T;      // the order mask whose backward arithmetic a residual uses

template <typename T> struct Resid {
    T a0, a1[2], aL;
    const T* target;      // [M][c] or null
};

template <int D> struct Sym {
    static constexpr int NF = D * (D + 1) / 2;            // distinct 2nd-order components
    static constexpr int N3 = D * (D + 1) * (D + 2) / 6;  // distinct 3rd-order components
};

template <typename T> __device__ __forceinline__ T exp_neg_half(T q);
template <> __device__ __forceinline__ float exp_neg_half<float>(float q) {
    // v_exp_f32 on a pre-scaled argument: exp(-q/2) = 2^(-q * log2(e)/2)
#ifdef PIGS_EXP_SCALE_VGPR
    float k;          // the factor from a register instead of a 32-bit literal (no "volatile": hoisted out of loops)
    asm("v_mov_b32 %0, 0xbf38aa3b" : "=v"(k));
    return __builtin_amdgcn_exp2f(q * k);
#else
    return __builtin_amdgcn_exp2f(q * -0.72134752044448170368f);
#endif
}
template <> __device__ __forceinline__ double exp_neg_half<double>(double q) { return exp(-0.5 * q); }

template <typename T> __device__ __forceinline__ T fma_(T a, T b, T c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float fma_<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Layout of the flat forward accumulator array for a compile-time order mask.
template <int D, int C, int MASK> struct FwdLayout {
    static constexpr int O0 = 0;
    static constexpr int O1 = O0 + ((MASK & ORD0) ? C : 0);
    static constexpr int O2 = O1 + ((MASK & ORD1) ? D * C : 0);
    static constexpr int O3 = O2 + ((MASK & ORD2) ? Sym<D>::NF * C : (MASK & ORD2T) ? C : 0);
    static constexpr int N = O3 + ((MASK & ORD3) ? Sym<D>::N3 * C : 0) + ((MASK & ORDR) ? C : 0);
};

// Pair geometry: x, p, g (and nothing else) for one (point, Gaussian).
template <typename T, int D> struct Pair {
    T x[D], p[D], g;
    __device__ __forceinline__ void eval(const T* s, const T* mu, const T* con) {
        if constexpr (D == 1) {
            x[0] = s[0] - mu[0];
            p[0] = con[0] * x[0];
            g = exp_neg_half<T>(x[0] * p[0]);
        } else {
            x[0] = s[0] - mu[0];
            x[1] = s[1] - mu[1];
            p[0] = fma_<T>(con[1], x[1], con[0] * x[0]);
            p[1] = fma_<T>(con[2], x[1], con[1] * x[0]);
            g = exp_neg_half<T>(fma_<T>(x[1], p[1], x[0] * p[0]));
        }
    }
};

// acc += contributions of one Gaussian to the outputs selected by MASK at one sample point.
// The order-1 accumulator holds +sum(p w); the sign is applied on store.
template <typename T, int D, int C, int MASK>
__device__ __forceinline__ void fwd_accumulate(T* acc, const T* s, const T* mu, const T* con, const T* v,
                                               const Resid<T>* rz = nullptr) {
    using L = FwdLayout<D, C, MASK>;
    Pair<T, D> pr;
    pr.eval(s, mu, con);
    if constexpr (MASK == ORDR) {
        // one polynomial factor per pair, one accumulator per channel: a0 - a1 . p + aL (|p|^2 - tr C)
        T F;
        if constexpr (D == 1) {
            F = fma_<T>(rz->aL, fma_<T>(pr.p[0], pr.p[0], -con[0]), fma_<T>(-rz->a1[0], pr.p[0], rz->a0));
        } else {
            const T ttr = fma_<T>(pr.p[0], pr.p[0], fma_<T>(pr.p[1], pr.p[1], -(con[0] + con[2])));
            F = fma_<T>(rz->aL, ttr, fma_<T>(-rz->a1[0], pr.p[0], fma_<T>(-rz->a1[1], pr.p[1], rz->a0)));
        }
        const T Fg = F * pr.g;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) acc[ch] = fma_<T>(v[ch], Fg, acc[ch]);
        return;
    }
    if constexpr ((MASK & ORD3) != 0) {
        // a pair whose g has underflowed to zero contributes nothing, but its cubic factor can
        // overflow (|p| > 7e12: conics of 1e12 from |rho| -> 1) and 0 * inf would poison the sum
        const bool live = pr.g > T(0);
#pragma unroll
        for (int i = 0; i < D; ++i) pr.p[i] = live ? pr.p[i] : T(0);
    }
    if constexpr (D == 1) {
        const T p = pr.p[0];
        T t2 = 0, t3 = 0;
        if constexpr ((MASK & (ORD2 | ORD2T | ORD3)) != 0) t2 = fma_<T>(p, p, -con[0]);
        if constexpr ((MASK & ORD3) != 0) t3 = p * (T(2) * con[0] - t2);
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const T w = v[ch] * pr.g;
            if constexpr ((MASK & ORD0) != 0) acc[L::O0 + ch] += w;
            if constexpr ((MASK & ORD1) != 0) acc[L::O1 + ch] = fma_<T>(p, w, acc[L::O1 + ch]);
            if constexpr ((MASK & (ORD2 | ORD2T)) != 0) acc[L::O2 + ch] = fma_<T>(t2, w, acc[L::O2 + ch]);   // d = 1: trace = u_xx
            if constexpr ((MASK & ORD3) != 0) acc[L::O3 + ch] = fma_<T>(t3, w, acc[L::O3 + ch]);
        }
    } else {
        const T px = pr.p[0], py = pr.p[1];
        T txx = 0, txy = 0, tyy = 0, t3[4] = {0, 0, 0, 0};
        if constexpr ((MASK & (ORD2 | ORD3)) != 0) {
            txx = fma_<T>(px, px, -con[0]);
            tyy = fma_<T>(py, py, -con[2]);
        }
        if constexpr ((MASK & ORD2) != 0) txy = fma_<T>(px, py, -con[1]);
        T ttr = 0;      // |p|^2 - tr C
        if constexpr ((MASK & ORD2T) != 0) {
            if constexpr ((MASK & ORD3) != 0) ttr = txx + tyy;
            else ttr = fma_<T>(px, px, fma_<T>(py, py, -(con[0] + con[2])));
        }
        if constexpr ((MASK & ORD3) != 0) {
            // C_ij p_k + C_ik p_j + C_jk p_i - p_i p_j p_k, written through t_ij = p_i p_j - C_ij
            const T b2 = T(2) * con[1];
            t3[0] = px * (T(2) * con[0] - txx);   // xxx = 3a px - px^3
            t3[1] = fma_<T>(b2, px, -txx * py);   // xxy = 2b px + a py - px^2 py
            t3[2] = fma_<T>(b2, py, -tyy * px);   // xyy = c px + 2b py - px py^2
            t3[3] = py * (T(2) * con[2] - tyy);   // yyy = 3c py - py^3
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const T w = v[ch] * pr.g;
            if constexpr ((MASK & ORD0) != 0) acc[L::O0 + ch] += w;
            if constexpr ((MASK & ORD1) != 0) {
                acc[L::O1 + 0 * C + ch] = fma_<T>(px, w, acc[L::O1 + 0 * C + ch]);
                acc[L::O1 + 1 * C + ch] = fma_<T>(py, w, acc[L::O1 + 1 * C + ch]);
            }
            if constexpr ((MASK & ORD2) != 0) {
                acc[L::O2 + 0 * C + ch] = fma_<T>(txx, w, acc[L::O2 + 0 * C + ch]);
                acc[L::O2 + 1 * C + ch] = fma_<T>(txy, w, acc[L::O2 + 1 * C + ch]);
                acc[L::O2 + 2 * C + ch] = fma_<T>(tyy, w, acc[L::O2 + 2 * C + ch]);
            }
            if constexpr ((MASK & ORD2T) != 0) acc[L::O2 + ch] = fma_<T>(ttr, w, acc[L::O2 + ch]);
            if constexpr ((MASK & ORD3) != 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[L::O3 + k * C + ch] = fma_<T>(t3[k], w, acc[L::O3 + k * C + ch]);
            }
        }
    }
}

// Expand the symmetric accumulators into the reference layouts
//   out0[M][c], out1[M][d][c], out2[M][d][d][c], out3[M][d][d][d][c]
// for point m.  Null output pointers are skipped (the order was computed but not requested).
// STREAM: non-temporal stores (`nt`): outputs that nothing in the launch reads again leave the L2 as
// they are written instead of staying dirty until the end-of-kernel write-back (the binned forward at
// C3 writes 24 MB: 28.0 -> 26.6 us); the dense kernels' small outputs stay cacheable for the consumer.
template <bool STREAM, typename T>
__device__ __forceinline__ void store_out(T* p, T v) {
    if constexpr (STREAM) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <typename T, int D, int C, int MASK, bool STREAM = false>
__device__ __forceinline__ void fwd_store(const T* acc, int64_t m, T* __restrict__ o0, T* __restrict__ o1,
                                          T* __restrict__ o2, T* __restrict__ o3, const Resid<T>* rz = nullptr) {
    using L = FwdLayout<D, C, MASK>;
    if constexpr (MASK == ORDR) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch)
            store_out<STREAM>(&o0[m * C + ch], rz->target ? acc[ch] - rz->target[m * C + ch] : acc[ch]);
        return;
    }
    if constexpr ((MASK & ORD0) != 0) {
        if (o0) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) store_out<STREAM>(&o0[m * C + ch], acc[L::O0 + ch]);
        }
    }
    if constexpr ((MASK & ORD1) != 0) {
        if (o1) {
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int ch = 0; ch < C; ++ch) store_out<STREAM>(&o1[(m * D + i) * C + ch], -acc[L::O1 + i * C + ch]);
        }
    }
    if constexpr ((MASK & ORD2) != 0) {
        if (o2) {
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j)
#pragma unroll
                    for (int ch = 0; ch < C; ++ch)
                        store_out<STREAM>(&o2[((m * D + i) * D + j) * C + ch], acc[L::O2 + (i + j) * C + ch]);  // D<=2: sym index = i+j
        }
    }
    if constexpr ((MASK & ORD2T) != 0) {      // trace: out2 is [M][c]
        if (o2) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) store_out<STREAM>(&o2[m * C + ch], acc[L::O2 + ch]);
        }
    }
    if constexpr ((MASK & ORD3) != 0) {
        if (o3) {
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j)
#pragma unroll
                    for (int k = 0; k < D; ++k)
#pragma unroll
                        for (int ch = 0; ch < C; ++ch)
                            store_out<STREAM>(&o3[(((m * D + i) * D + j) * D + k) * C + ch], acc[L::O3 + (i + j + k) * C + ch]);
        }
    }
}

// Inverse of fwd_store: read the partial sums of point m back into the accumulators (a cell with
// more than 64 points evaluates each queue flush for every 64-point chunk and parks the partial
// sums in the output rows in between).  Null outputs were never stored and are not wanted: zero.
template <typename T, int D, int C, int MASK>
__device__ __forceinline__ void fwd_load(T* acc, int64_t m, const T* __restrict__ o0, const T* __restrict__ o1,
                                         const T* __restrict__ o2, const T* __restrict__ o3) {
    using L = FwdLayout<D, C, MASK>;
#pragma unroll
    for (int k = 0; k < L::N; ++k) acc[k] = T(0);
    if constexpr ((MASK & ORD0) != 0) {
        if (o0) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[L::O0 + ch] = o0[m * C + ch];
        }
    }
    if constexpr ((MASK & ORD1) != 0) {
        if (o1) {
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int ch = 0; ch < C; ++ch) acc[L::O1 + i * C + ch] = -o1[(m * D + i) * C + ch];
        }
    }
    if constexpr ((MASK & ORD2) != 0) {
        if (o2) {
#pragma unroll
            for (int k = 0; k < Sym<D>::NF; ++k)        // (0,0), (0,1), (1,1): first index 0 until the last
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    const int i = k == Sym<D>::NF - 1 ? D - 1 : 0, j = k - i;
                    acc[L::O2 + k * C + ch] = o2[((m * D + i) * D + j) * C + ch];
                }
        }
    }
    if constexpr ((MASK & ORD2T) != 0) {
        if (o2) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[L::O2 + ch] = o2[m * C + ch];
        }
    }
    if constexpr ((MASK & ORD3) != 0) {
        if (o3) {
#pragma unroll
            for (int k = 0; k < Sym<D>::N3; ++k)        // (0,0,0), (0,0,1), (0,1,1), (1,1,1)
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    const int a = k >= 3 ? 1 : 0, b = k >= 2 ? 1 : 0, c3 = k >= 1 ? 1 : 0;     // D = 2
                    const int i = D == 1 ? 0 : a, j = D == 1 ? 0 : b, l = D == 1 ? 0 : c3;
                    acc[L::O3 + k * C + ch] = o3[(((m * D + i) * D + j) * D + l) * C + ch];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward (VJP) of the selected outputs wrt (means, flat conics, values) for one pair.
//
//   L = sum_m sum_n g sum_c v_c F_c ,  F_c = G0_c - G1_ic p_i + H_ij,c t_ij + K_ijk,c poly3_ijk
// with the incoming gradients symmetrised once per point (Gsym below).  With A = sum_c v_c F_c,
// dA = grad_p A and E = explicit dA/dC:
//   dL/dv_c   += g F_c
//   dL/dmu_l  += g (A p_l - sum_i dA_i C_il)
//   dL/dC_kl  += g (-A x_k x_l / 2 + dA_k x_l + E_kl)      (full matrix; flat = [00, 01+10, 11])
// This is what torch.autograd gives through the reference functions
// (test_derivatives.py:122-124, 208-220, 340-356).
// ---------------------------------------------------------------------------------------------

// Incoming gradients at one sample point, symmetrised over derivative indices.
template <typename T, int D, int C, int MASK> struct Gsym {
    T g0[C], g1[D][C], g2[Sym<D>::NF][C], g3[Sym<D>::N3][C];
    // Null gradient pointers (an order inside the compiled mask whose output received no
    // gradient) read as zero.
    __device__ __forceinline__ void load(int64_t m, const T* __restrict__ G0, const T* __restrict__ G1,
                                         const T* __restrict__ G2, const T* __restrict__ G3) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            if constexpr ((MASK & ORD0) != 0) g0[ch] = G0 ? G0[m * C + ch] : T(0);
            if constexpr ((MASK & ORD1) != 0) {
#pragma unroll
                for (int i = 0; i < D; ++i) g1[i][ch] = G1 ? G1[(m * D + i) * C + ch] : T(0);
            }
            if constexpr ((MASK & ORD2) != 0) {
#pragma unroll
                for (int k = 0; k < Sym<D>::NF; ++k) g2[k][ch] = 0;
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) g2[i + j][ch] += G2 ? G2[((m * D + i) * D + j) * C + ch] : T(0);
            }
            if constexpr ((MASK & ORD2T) != 0) {   // gradient of the trace = gl * identity
                const T gl = G2 ? G2[m * C + ch] : T(0);
#pragma unroll
                for (int k = 0; k < Sym<D>::NF; ++k) g2[k][ch] = (k == 0 || k == Sym<D>::NF - 1) ? gl : T(0);
            }
            if constexpr ((MASK & ORD3) != 0) {
#pragma unroll
                for (int k = 0; k < Sym<D>::N3; ++k) g3[k][ch] = 0;
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j)
#pragma unroll
                        for (int k = 0; k < D; ++k)
                            g3[i + j + k][ch] += G3 ? G3[(((m * D + i) * D + j) * D + k) * C + ch] : T(0);
            }
        }
    }
    // the incoming gradient gr [M][c] of a residual output (MASK = ORDR_AS): g0 = a0 gr, g1 = a1 gr, trace = aL gr
    __device__ __forceinline__ void load_residual(int64_t m, const T* __restrict__ GR, const Resid<T>& rz) {
        static_assert(MASK == ORDR_AS, "a residual's backward runs on orders 0, 1 and the trace");
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const T gr = GR[m * C + ch];
            g0[ch] = rz.a0 * gr;
#pragma unroll
            for (int i = 0; i < D; ++i) g1[i][ch] = rz.a1[i] * gr;
#pragma unroll
            for (int k = 0; k < Sym<D>::NF; ++k) g2[k][ch] = (k == 0 || k == Sym<D>::NF - 1) ? rz.aL * gr : T(0);
        }
    }
};

// Flat backward accumulator: [g_means D][g_conics NF][g_values C]
template <int D, int C> struct BwdLayout {
    static constexpr int MU = 0;
    static constexpr int CON = D;
    static constexpr int VAL = D + Sym<D>::NF;
    static constexpr int N = D + Sym<D>::NF + C;
};

// FAR_GUARD (the dense kernels, which meet every pair, and the binned third-derivative backward,
// whose waves evaluate a queued Gaussian for all 64 points of the cell): where g has underflowed
// to zero the polynomial factors can overflow (p^4 C ~ 1e38 for sigma ~ 1e-4 of the distance)
// and 0 * inf would poison the sums; such a pair contributes exactly nothing, so its x and p are
// zeroed (v_exp_f32 flushes denormals: g is either above 1e-38 or exactly 0).
//
// FACTORED (D = 2, C = 1 only): what depends on the Gaussian alone is left out of the per-pair work
// and applied once per Gaussian by the caller (plan_unpermute_kernel):
//   acc[MU]  = sum_m g (A x - dA)                    -> dL/dmu = v C acc[MU]      (A p - C dA = C (A x - dA))
//   acc[CON] = sum_m g (-A x x^T / 2 + dA x^T + E)   -> dL/dC  = v acc[CON]
//   acc[VAL] = sum_m g F                             (unchanged)
// 38 instead of 46 instructions per pair for orders 0..2.
template <typename T, int D, int C, int MASK, bool FAR_GUARD = false, bool FACTORED = false>
__device__ __forceinline__ void bwd_accumulate(T* acc, const T* s, const T* mu, const T* con, const T* v,
                                               const Gsym<T, D, C, MASK>& G) {
    static_assert(!FACTORED || (D == 2 && C == 1), "the factored form is written for d = 2, c = 1");
    using L = BwdLayout<D, C>;
    Pair<T, D> pr;
    pr.eval(s, mu, con);
    const T g = pr.g;
    if constexpr (FAR_GUARD) {
        const bool live = g > T(0);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            pr.p[i] = live ? pr.p[i] : T(0);
            pr.x[i] = live ? pr.x[i] : T(0);
        }
    }
    if constexpr (D == 1) {
        const T p = pr.p[0], x = pr.x[0], a = con[0];
        const T t2 = fma_<T>(p, p, -a);        // p^2 - a
        const T t3 = p * (T(2) * a - t2);      // 3 a p - p^3
        T A = 0, dA = 0, E = 0;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            T F = 0, dF = 0, eF = 0;
            if constexpr ((MASK & ORD0) != 0) F += G.g0[ch];
            if constexpr ((MASK & ORD1) != 0) { F = fma_<T>(-G.g1[0][ch], p, F); dF -= G.g1[0][ch]; }
            if constexpr ((MASK & (ORD2 | ORD2T)) != 0) {
                F = fma_<T>(G.g2[0][ch], t2, F);
                dF = fma_<T>(T(2) * G.g2[0][ch], p, dF);
                eF -= G.g2[0][ch];
            }
            if constexpr ((MASK & ORD3) != 0) {
                F = fma_<T>(G.g3[0][ch], t3, F);
                dF = fma_<T>(G.g3[0][ch], T(-3) * t2, dF);     // d/dp (3ap - p^3) = 3a - 3p^2
                eF = fma_<T>(T(3) * G.g3[0][ch], p, eF);       // d/da (3ap) = 3p
            }
            acc[L::VAL + ch] = fma_<T>(g, F, acc[L::VAL + ch]);
            A = fma_<T>(v[ch], F, A);
            dA = fma_<T>(v[ch], dF, dA);
            E = fma_<T>(v[ch], eF, E);
        }
        acc[L::MU] = fma_<T>(g, fma_<T>(A, p, -dA * a), acc[L::MU]);
        acc[L::CON] = fma_<T>(g, fma_<T>(T(-0.5) * A * x, x, fma_<T>(dA, x, E)), acc[L::CON]);
    } else {
        const T px = pr.p[0], py = pr.p[1], dx = pr.x[0], dy = pr.x[1];
        const T a = con[0], b = con[1], c = con[2];
        const T txx = fma_<T>(px, px, -a), txy = fma_<T>(px, py, -b), tyy = fma_<T>(py, py, -c);
        T A = 0, dAx = 0, dAy = 0, Exx = 0, Exy = 0, Eyy = 0;   // Exy holds E_01 + E_10
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            T F = 0, dFx = 0, dFy = 0, exx = 0, exy = 0, eyy = 0;
            if constexpr ((MASK & ORD0) != 0) F += G.g0[ch];
            if constexpr ((MASK & ORD1) != 0) {
                F = fma_<T>(-G.g1[0][ch], px, F);
                F = fma_<T>(-G.g1[1][ch], py, F);
                dFx -= G.g1[0][ch];
                dFy -= G.g1[1][ch];
            }
            if constexpr ((MASK & (ORD2 | ORD2T)) != 0) {
                const T hxx = G.g2[0][ch], hxy = G.g2[1][ch], hyy = G.g2[2][ch];
                F = fma_<T>(hxx, txx, F);
                F = fma_<T>(hxy, txy, F);
                F = fma_<T>(hyy, tyy, F);
                dFx = fma_<T>(T(2) * hxx, px, fma_<T>(hxy, py, dFx));
                dFy = fma_<T>(T(2) * hyy, py, fma_<T>(hxy, px, dFy));
                exx -= hxx;
                exy -= hxy;
                eyy -= hyy;
            }
            if constexpr ((MASK & ORD3) != 0) {
                const T k0 = G.g3[0][ch], k1 = G.g3[1][ch], k2 = G.g3[2][ch], k3 = G.g3[3][ch];
                // poly: xxx = 3a px - px^3 ; xxy = 2b px + a py - px^2 py ;
                //       xyy = c px + 2b py - px py^2 ; yyy = 3c py - py^3
                F = fma_<T>(k0, px * (T(2) * a - txx), F);
                F = fma_<T>(k1, fma_<T>(T(2) * b, px, -txx * py), F);
                F = fma_<T>(k2, fma_<T>(T(2) * b, py, -tyy * px), F);
                F = fma_<T>(k3, py * (T(2) * c - tyy), F);
                // d/dpx
                dFx = fma_<T>(k0, T(-3) * txx, dFx);
                dFx = fma_<T>(k1, T(-2) * txy, dFx);
                dFx = fma_<T>(k2, -tyy, dFx);
                // d/dpy
                dFy = fma_<T>(k1, -txx, dFy);
                dFy = fma_<T>(k2, T(-2) * txy, dFy);
                dFy = fma_<T>(k3, T(-3) * tyy, dFy);
                // explicit d/da, d/db (flat: both off-diagonal entries), d/dc
                exx = fma_<T>(T(3) * k0, px, fma_<T>(k1, py, exx));
                exy = fma_<T>(T(2) * k1, px, fma_<T>(T(2) * k2, py, exy));
                eyy = fma_<T>(T(3) * k3, py, fma_<T>(k2, px, eyy));
            }
            acc[L::VAL + ch] = fma_<T>(g, F, acc[L::VAL + ch]);
            if constexpr (C == 1) {
                // one channel: v factors out of A, dA and E -- carry it in the weight instead
                A = F; dAx = dFx; dAy = dFy; Exx = exx; Exy = exy; Eyy = eyy;
            } else {
                A = fma_<T>(v[ch], F, A);
                dAx = fma_<T>(v[ch], dFx, dAx);
                dAy = fma_<T>(v[ch], dFy, dAy);
                Exx = fma_<T>(v[ch], exx, Exx);
                Exy = fma_<T>(v[ch], exy, Exy);
                Eyy = fma_<T>(v[ch], eyy, Eyy);
            }
        }
        if constexpr (FACTORED) {
            // r = -A x / 2;  A x - dA = -(2 r + dA);  conic terms: x_k (r_k + dA_k) + E_kk, and for the
            // off-diagonal x_1 (2 r_0 + dA_0) + dA_1 x_0 + E_01 = -x_1 mx + dA_1 x_0 + E_01
            const T hA = T(-0.5) * A;
            const T rx = hA * dx, ry = hA * dy;
            const T mx = fma_<T>(T(-2), rx, -dAx), my = fma_<T>(T(-2), ry, -dAy);
            acc[L::MU + 0] = fma_<T>(g, mx, acc[L::MU + 0]);
            acc[L::MU + 1] = fma_<T>(g, my, acc[L::MU + 1]);
            acc[L::CON + 0] = fma_<T>(g, fma_<T>(rx + dAx, dx, Exx), acc[L::CON + 0]);
            acc[L::CON + 1] = fma_<T>(g, fma_<T>(-mx, dy, fma_<T>(dAy, dx, Exy)), acc[L::CON + 1]);
            acc[L::CON + 2] = fma_<T>(g, fma_<T>(ry + dAy, dy, Eyy), acc[L::CON + 2]);
            return;
        }
        const T g_ = g;
        const T g = (C == 1) ? g_ * v[0] : g_;          // weight of the mean / conic terms
        // means: g (A p_l - (dA . C)_l)
        acc[L::MU + 0] = fma_<T>(g, fma_<T>(A, px, -fma_<T>(dAx, a, dAy * b)), acc[L::MU + 0]);
        acc[L::MU + 1] = fma_<T>(g, fma_<T>(A, py, -fma_<T>(dAx, b, dAy * c)), acc[L::MU + 1]);
        // flat conic: [G00, G01 + G10, G11]
        const T hA = T(-0.5) * A;
        acc[L::CON + 0] = fma_<T>(g, fma_<T>(hA * dx, dx, fma_<T>(dAx, dx, Exx)), acc[L::CON + 0]);
        acc[L::CON + 1] = fma_<T>(g, fma_<T>(-A * dx, dy, fma_<T>(dAx, dy, fma_<T>(dAy, dx, Exy))), acc[L::CON + 1]);
        acc[L::CON + 2] = fma_<T>(g, fma_<T>(hA * dy, dy, fma_<T>(dAy, dy, Eyy)), acc[L::CON + 2]);
    }
}

}  // namespace pigs
