// Probe: lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 (brute force over candidate maps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k(const double* a, const double* b, const double* c, double* d) {
  int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
int main() {
  std::vector<double> a(64), b(64), c(64), d(64);
  for (int i = 0; i < 64; ++i) { a[i] = sin(1 + 0.7 * i); b[i] = cos(0.3 + 1.3 * i); c[i] = 0.01 * i; }
  double *da, *db, *dc, *dd;
  CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dc, 512)); CK(hipMalloc(&dd, 512));
  CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
  CK(hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice));
  k<<<1, 64>>>(da, db, dc, dd);
  CK(hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost));

  // lane = f0 + 4*f1 + 16*f2 where (f0,f1,f2) is a permutation of the 3 logical indices
  int perms[6][3] = {{0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0}};
  auto lane = [&](const int* p, int blk, int x, int y) { int v[3] = {blk, x, y}; return v[p[0]] + 4 * v[p[1]] + 16 * v[p[2]]; };
  const char* nm[3] = {"blk", "x", "y"};
  for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
    double err = 0;
    for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
      int ld = lane(perms[pd], blk, i, j);
      double s = c[ld];
      for (int kk = 0; kk < 4; ++kk) s += a[lane(perms[pa], blk, i, kk)] * b[lane(perms[pb], blk, kk, j)];
      err = fmax(err, fabs(s - d[ld]));
    }
    if (err < 1e-13) printf("MATCH: A(i,k): lane = %s + 4*%s + 16*%s [x=i,y=k] | B(k,j): lane = %s + 4*%s + 16*%s [x=k,y=j] | D(i,j): lane = %s + 4*%s + 16*%s [x=i,y=j]\n",
      nm[perms[pa][0]], nm[perms[pa][1]], nm[perms[pa][2]], nm[perms[pb][0]], nm[perms[pb][1]], nm[perms[pb][2]], nm[perms[pd][0]], nm[perms[pd][1]], nm[perms[pd][2]]);
  }
  printf("done\n");
  return 0;
}
