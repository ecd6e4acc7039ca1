// C-ABI shim over the reference's stand-alone twin tf_ops/grouping/test/query_ball_point.cpp
// (compiled from where it lies under /root/reference with -Dmain=ref_main_grouping; see Makefile).
// TEST INFRASTRUCTURE ONLY.  Declarations restate query_ball_point.cpp:52, :70.
void group_point_cpu(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                     float *out);
void group_point_grad_cpu(int b, int n, int c, int m, int nsample, const float *grad_out,
                          const int *idx, float *grad_points);
extern "C" {
void ref_group_point_cpu(int b, int n, int c, int m, int nsample, const float *points,
                         const int *idx, float *out) {
  group_point_cpu(b, n, c, m, nsample, points, idx, out);
}
void ref_group_point_grad_cpu(int b, int n, int c, int m, int nsample, const float *grad_out,
                              const int *idx, float *grad_points) {
  group_point_grad_cpu(b, n, c, m, nsample, grad_out, idx, grad_points);
}
}
