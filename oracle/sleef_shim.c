/* ORACLE, TEST INFRASTRUCTURE ONLY.  A four-line shim over torch's OWN build of SLEEF (the vendored third_party/sleef that ATen's
 * vectorised CPU kernels call): Sleef_acosf8_u10avx2 / Sleef_atan2f8_u10avx2 are exported by libtorch_cpu.so.  The oracle uses them
 * for the pinned acos rule of the sphere index (oracle/scenerf_oracle.py: OracleConfig.acos_rule) -- the routine the reference's
 * dependency ships, not a restatement of it.  Built by oracle/sleef_acos.py with `gcc -O2 -mavx2 -shared -fPIC` into oracle/_ref/. */
#include <immintrin.h>
#include <stddef.h>
__m256 Sleef_acosf8_u10avx2(__m256);
__m256 Sleef_atan2f8_u10avx2(__m256, __m256);
void oracle_sleef_acosf(const float* x, float* y, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) _mm256_storeu_ps(y + i, Sleef_acosf8_u10avx2(_mm256_loadu_ps(x + i)));
    if (i < n) {
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8];
        for (size_t j = i; j < n; ++j) a[j - i] = x[j];
        _mm256_storeu_ps(b, Sleef_acosf8_u10avx2(_mm256_loadu_ps(a)));
        for (size_t j = i; j < n; ++j) y[j] = b[j - i];
    }
}
void oracle_sleef_atan2f(const float* p, const float* q, float* y, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) _mm256_storeu_ps(y + i, Sleef_atan2f8_u10avx2(_mm256_loadu_ps(p + i), _mm256_loadu_ps(q + i)));
    if (i < n) {
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c[8] = {1, 1, 1, 1, 1, 1, 1, 1}, b[8];
        for (size_t j = i; j < n; ++j) { a[j - i] = p[j]; c[j - i] = q[j]; }
        _mm256_storeu_ps(b, Sleef_atan2f8_u10avx2(_mm256_loadu_ps(a), _mm256_loadu_ps(c)));
        for (size_t j = i; j < n; ++j) y[j] = b[j - i];
    }
}

/* The pinned rule's small matrix products: out[m][r] = fma(A[r][k-1], x[m][k-1], ... fma(A[r][1], x[m][1], A[r][0] * x[m][0])), the
 * k-ordered chain of a BLAS sgemm micro-kernel (what MKL runs for these shapes on the Intel container the golden vectors were minted on;
 * MKL on other CPUs sums some layouts in another order).  A is (rows, k) row-major, x is (M, k) row-major, out is (M, rows). */
#include <math.h>
void oracle_matvec_fma(const float* A, int rows, int k, const float* x, size_t M, float* out) {
    for (size_t m = 0; m < M; ++m)
        for (int r = 0; r < rows; ++r) {
            float acc = A[r * k] * x[m * k];
            for (int j = 1; j < k; ++j) acc = fmaf(A[r * k + j], x[m * k + j], acc);
            out[m * rows + r] = acc;
        }
}
