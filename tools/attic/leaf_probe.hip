// Where does a leaf spend its cycles?  Compiles the product's leaf kernel with phase stamps (s_memtime) and prints the
// per-phase cycle counts of one 128 x 128 factor+inverse.  Dev tool.
#define GPP_LEAF_STAMP 1
#include "../gp-plus_amd/csrc/gpp_leaf.hip"
#include <cstdio>
#include <vector>
#include <cmath>
int main() {
  const int n = 128, ld = 128;
  std::vector<double> A(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = std::exp(-0.05 * (i - j) * (i - j)) + (i == j ? 1e-2 : 0.0);
  double *dA, *dL; int* info;
  (void)hipMalloc(&dA, sizeof(double) * n * n); (void)hipMalloc(&dL, sizeof(double) * n * n); (void)hipMalloc(&info, 4);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    (void)hipMemset(info, 0, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    (void)gpp_launch_leaf(nullptr, dA, ld, dL, ld, n, info, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[64];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_leaf_stamps), sizeof(st));
    if (rep == 2) {
      printf("event time %.1f us; total stamped %.0f cycles\n", ms * 1e3, (double)(st[43] - st[0]));
      printf("load A: (before stamp 0)\n");
      for (int s = 0; s < 8; ++s)
        printf("step %d: park+sync %5llu | diag(wave0) %6llu | sync %5llu | panel+sync %5llu | trailing %5llu\n", s,
               st[2 + 5 * s] - st[1 + 5 * s], st[3 + 5 * s] - st[2 + 5 * s], st[4 + 5 * s] - st[3 + 5 * s],
               st[5 + 5 * s] - st[4 + 5 * s], (s < 7 ? st[1 + 5 * (s + 1)] : st[41]) - st[5 + 5 * s]);
      printf("diag inverses %llu | merge levels %llu | write-out %llu\n", st[44] - st[41], st[42] - st[44], st[43] - st[42]);
      std::vector<double> U(n * n), X(n * n);
      (void)hipMemcpy(U.data(), dA, sizeof(double) * n * n, hipMemcpyDeviceToHost);
      (void)hipMemcpy(X.data(), dL, sizeof(double) * n * n, hipMemcpyDeviceToHost);
      double e1 = 0, e2 = 0, e3 = 0;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
          double s1 = 0, s2 = 0;
          for (int k = 0; k <= j; ++k) s1 += U[k * n + i] * U[k * n + j];          // (L L^T)_ij, L_ik = U_ki
          for (int k = j; k <= i; ++k) s2 += X[i * n + k] * U[j * n + k];          // (X L)_ij
          e1 = std::fmax(e1, std::fabs(s1 - A[i * n + j]));
          e2 = std::fmax(e2, std::fabs(s2 - (i == j ? 1.0 : 0.0)));
          e3 = std::fmax(e3, std::fabs(X[i * n + j] - X[j * n + i]));
        }
      printf("max |LL^T - A| %.2e  max |XL - I| %.2e  mirror %.2e\n", e1, e2, e3);
    }
  }
  return 0;
}
