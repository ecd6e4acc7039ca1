// C-ABI shim over the reference's stand-alone twin tf_ops/interpolation/interpolate.cpp
// (compiled from where it lies under /root/reference with -Dmain=ref_main_interpolate; see
// Makefile).  TEST INFRASTRUCTURE ONLY: used to pin oracle/dh3d_oracle.c, never by the product.
// Declarations restate the signatures at interpolate.cpp:21, :84, :108.
void threenn_cpu(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx);
void interpolate_cpu(int b, int m, int c, int n, const float *points, const int *idx,
                     const float *weight, float *out);
void interpolate_grad_cpu(int b, int n, int c, int m, const float *grad_out, const int *idx,
                          const float *weight, float *grad_points);
extern "C" {
void ref_threenn_cpu(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist,
                     int *idx) { threenn_cpu(b, n, m, xyz1, xyz2, dist, idx); }
void ref_interpolate_cpu(int b, int m, int c, int n, const float *points, const int *idx,
                         const float *weight, float *out) {
  interpolate_cpu(b, m, c, n, points, idx, weight, out);
}
void ref_interpolate_grad_cpu(int b, int n, int c, int m, const float *grad_out, const int *idx,
                              const float *weight, float *grad_points) {
  interpolate_grad_cpu(b, n, c, m, grad_out, idx, weight, grad_points);
}
}
