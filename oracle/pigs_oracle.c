/*
 * CPU oracle (plain C) for the differentiable Gaussian sampler.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py as the checker / reported CPU baseline.  Never linked into, imported by,
 * or called from the product path (pigs_amd/).
 *
 * Restates, pair by pair and with no culling, the reference's dense PyTorch sampler:
 *   order 0  gaussians.sample_gaussians      /root/reference/gaussians.py:48-58
 *   order 1  gaussians.gaussian_derivative   /root/reference/gaussians.py:89-101
 *   order 2  gaussians.gaussian_derivative2  /root/reference/gaussians.py:103-116
 *   order 3  derivative of order 2 wrt the sample point (shape: model_pn.py:654)
 *   backward = torch.autograd through those wrt (means, values, conics)
 *              (test_derivatives.py:122-124, 208-220, 340-356), conic gradient in the
 *              sampler's FLAT layout [xx, xy, yy] (gaussians.py:186-189).
 * Pinned against the tests/golden fixtures (outputs of the reference itself) by tests/test_oracle.py.
 *
 * x = s - mu, p = C x, q = x.p, g = exp(-q/2).  All arithmetic in double.
 * Layouts (row-major, contiguous): means[N][d], conics[N][d(d+1)/2] (upper triangle,
 * row-major), values[N][c], samples[M][d]; out0[M][c], out1[M][d][c], out2[M][d][d][c],
 * out3[M][d][d][d][c].  d <= 3, c <= 8.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 3
#define MAXC 8

static void unpack_conic(int d, const double *flat, double C[MAXD][MAXD]) {
    int k = 0;
    for (int i = 0; i < d; ++i)
        for (int j = i; j < d; ++j) {
            C[i][j] = flat[k];
            C[j][i] = flat[k];
            ++k;
        }
}

/* third-order polynomial (C_ij p_k + C_ik p_j + C_jk p_i - p_i p_j p_k) */
static inline double poly3(double C[MAXD][MAXD], const double *p, int i, int j, int k) {
    return C[i][j] * p[k] + C[i][k] * p[j] + C[j][k] * p[i] - p[i] * p[j] * p[k];
}

int pigs_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int pigs_oracle_forward(int d, int c, int orders_mask, long N, long M,
                        const double *means, const double *conics, const double *values,
                        const double *samples,
                        double *out0, double *out1, double *out2, double *out3) {
    if (d < 1 || d > MAXD || c < 1 || c > MAXC) return 1;
    const int nf = d * (d + 1) / 2;
#pragma omp parallel for schedule(static)
    for (long m = 0; m < M; ++m) {
        double a0[MAXC] = {0}, a1[MAXD][MAXC] = {{0}}, a2[MAXD][MAXD][MAXC] = {{{0}}};
        double a3[MAXD][MAXD][MAXD][MAXC];
        memset(a3, 0, sizeof a3);
        for (long n = 0; n < N; ++n) {
            double C[MAXD][MAXD], x[MAXD], p[MAXD], q = 0.0;
            unpack_conic(d, conics + n * nf, C);
            for (int i = 0; i < d; ++i) x[i] = samples[m * d + i] - means[n * d + i];
            for (int i = 0; i < d; ++i) {
                p[i] = 0.0;
                for (int j = 0; j < d; ++j) p[i] += C[i][j] * x[j];
                q += x[i] * p[i];
            }
            const double g = exp(-0.5 * q);
            for (int ch = 0; ch < c; ++ch) {
                const double w = values[n * c + ch] * g;
                if (orders_mask & 1) a0[ch] += w;
                if (orders_mask & 2)
                    for (int i = 0; i < d; ++i) a1[i][ch] -= p[i] * w;
                if (orders_mask & 4)
                    for (int i = 0; i < d; ++i)
                        for (int j = 0; j < d; ++j) a2[i][j][ch] += (p[i] * p[j] - C[i][j]) * w;
                if (orders_mask & 8)
                    for (int i = 0; i < d; ++i)
                        for (int j = 0; j < d; ++j)
                            for (int k = 0; k < d; ++k) a3[i][j][k][ch] += poly3(C, p, i, j, k) * w;
            }
        }
        for (int ch = 0; ch < c; ++ch) {
            if (orders_mask & 1) out0[m * c + ch] = a0[ch];
            if (orders_mask & 2)
                for (int i = 0; i < d; ++i) out1[(m * d + i) * c + ch] = a1[i][ch];
            if (orders_mask & 4)
                for (int i = 0; i < d; ++i)
                    for (int j = 0; j < d; ++j) out2[((m * d + i) * d + j) * c + ch] = a2[i][j][ch];
            if (orders_mask & 8)
                for (int i = 0; i < d; ++i)
                    for (int j = 0; j < d; ++j)
                        for (int k = 0; k < d; ++k)
                            out3[(((m * d + i) * d + j) * d + k) * c + ch] = a3[i][j][k][ch];
        }
    }
    return 0;
}

/*
 * VJP.  L = sum_m sum_c sum_n v_nc g_mn F_c(m,n) with
 *   F_c = G0_c - G1_ic p_i + G2_ijc (p_i p_j - C_ij) + G3_ijkc poly3_ijk .
 * dL/dv_nc = sum_m g F_c ;  A = sum_c v_c F_c ;  dA = grad_p A ;  E = explicit dA/dC.
 * dL/dmu_l = sum_m g (A p_l - sum_i dA_i C_il)
 * dL/dC_kl = sum_m g (-A x_k x_l / 2 + dA_k x_l + E_kl)   (full matrix), folded to flat.
 */
static int backward_impl(int d, int c, int orders_mask, long N, long M,
                         const double *means, const double *conics, const double *values,
                         const double *samples,
                         const double *g0, const double *g1, const double *g2, const double *g3,
                         double *g_means, double *g_conics, double *g_values, int abs_mode) {
    if (d < 1 || d > MAXD || c < 1 || c > MAXC) return 1;
    const int nf = d * (d + 1) / 2;
#pragma omp parallel for schedule(dynamic, 4)
    for (long n = 0; n < N; ++n) {
        double C[MAXD][MAXD];
        unpack_conic(d, conics + n * nf, C);
        double gm[MAXD] = {0}, gF[MAXD * (MAXD + 1) / 2] = {0}, gv[MAXC] = {0};
        for (long m = 0; m < M; ++m) {
            double x[MAXD], p[MAXD], q = 0.0;
            for (int i = 0; i < d; ++i) x[i] = samples[m * d + i] - means[n * d + i];
            for (int i = 0; i < d; ++i) {
                p[i] = 0.0;
                for (int j = 0; j < d; ++j) p[i] += C[i][j] * x[j];
                q += x[i] * p[i];
            }
            const double g = exp(-0.5 * q);
            double A = 0.0, dA[MAXD] = {0}, E[MAXD][MAXD] = {{0}};
            for (int ch = 0; ch < c; ++ch) {
                const double v = values[n * c + ch];
                double F = 0.0;
                if (orders_mask & 1) F += g0[m * c + ch];
                if (orders_mask & 2)
                    for (int i = 0; i < d; ++i) {
                        const double G = g1[(m * d + i) * c + ch];
                        F -= G * p[i];
                        dA[i] -= v * G;
                    }
                if (orders_mask & 4)
                    for (int i = 0; i < d; ++i)
                        for (int j = 0; j < d; ++j) {
                            const double G = g2[((m * d + i) * d + j) * c + ch];
                            F += G * (p[i] * p[j] - C[i][j]);
                            dA[i] += v * G * p[j];
                            dA[j] += v * G * p[i];
                            E[i][j] -= v * G;
                        }
                if (orders_mask & 8)
                    for (int i = 0; i < d; ++i)
                        for (int j = 0; j < d; ++j)
                            for (int k = 0; k < d; ++k) {
                                const double G = g3[(((m * d + i) * d + j) * d + k) * c + ch];
                                F += G * poly3(C, p, i, j, k);
                                dA[k] += v * G * (C[i][j] - p[i] * p[j]);
                                dA[j] += v * G * (C[i][k] - p[i] * p[k]);
                                dA[i] += v * G * (C[j][k] - p[j] * p[k]);
                                E[i][j] += v * G * p[k];
                                E[i][k] += v * G * p[j];
                                E[j][k] += v * G * p[i];
                            }
                gv[ch] += abs_mode ? fabs(g * F) : g * F;
                A += v * F;
            }
            for (int l = 0; l < d; ++l) {
                double t = A * p[l];
                for (int i = 0; i < d; ++i) t -= dA[i] * C[i][l];
                gm[l] += abs_mode ? fabs(g * t) : g * t;
            }
            /* this pair's contribution to the FLAT conic gradient [G00, G01 + G10, G11] */
            double pc[MAXD][MAXD];
            for (int k = 0; k < d; ++k)
                for (int l = 0; l < d; ++l)
                    pc[k][l] = g * (-0.5 * A * x[k] * x[l] + dA[k] * x[l] + E[k][l]);
            int f = 0;
            for (int i = 0; i < d; ++i)
                for (int j = i; j < d; ++j) {
                    const double t = (i == j) ? pc[i][i] : pc[i][j] + pc[j][i];
                    gF[f++] += abs_mode ? fabs(t) : t;
                }
        }
        for (int l = 0; l < d; ++l) g_means[n * d + l] = gm[l];
        for (int ch = 0; ch < c; ++ch) g_values[n * c + ch] = gv[ch];
        for (int f = 0; f < nf; ++f) g_conics[n * nf + f] = gF[f];
    }
    return 0;
}

int pigs_oracle_backward(int d, int c, int orders_mask, long N, long M,
                         const double *means, const double *conics, const double *values,
                         const double *samples,
                         const double *g0, const double *g1, const double *g2, const double *g3,
                         double *g_means, double *g_conics, double *g_values) {
    return backward_impl(d, c, orders_mask, N, M, means, conics, values, samples, g0, g1, g2, g3,
                         g_means, g_conics, g_values, 0);
}

/*
 * The same sums with every (sample, Gaussian) pair's contribution taken in ABSOLUTE value: the
 * magnitude a float32 accumulation error is measured against (an entry that is a small difference
 * of large contributions cannot be summed to 1e-5 of ITSELF in float32 by any order of summation;
 * tests bound the error of each entry by a few ulp of this sum).  Test infrastructure only.
 */
int pigs_oracle_backward_abs(int d, int c, int orders_mask, long N, long M,
                             const double *means, const double *conics, const double *values,
                             const double *samples,
                             const double *g0, const double *g1, const double *g2, const double *g3,
                             double *g_means, double *g_conics, double *g_values) {
    return backward_impl(d, c, orders_mask, N, M, means, conics, values, samples, g0, g1, g2, g3,
                         g_means, g_conics, g_values, 1);
}
