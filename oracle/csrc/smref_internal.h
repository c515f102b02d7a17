/* smref_internal.h -- helpers shared by the C restatements (TEST ORACLE / CPU baseline only). */
#ifndef SMREF_INTERNAL_H_
#define SMREF_INTERNAL_H_

typedef struct {
  int n;
  const double* pts; /* [n][3] */
  int* perm;         /* point order */
  int* node_lo;      /* per node: first */
  int* node_hi;      /* per node: last (exclusive) */
  int* node_dim;     /* -1 = leaf */
  double* node_cut;
  int* node_left;
  int* node_right;
  int n_nodes, cap_nodes;
} KdTree;

double now_s(void);
void kd_select(const double* pts, int* idx, int lo, int hi, int k, int dim);
KdTree* kd_build(const double* pts, int n);
void kd_free(KdTree* t);
void kd_nn(const KdTree* t, const double q[3], int* best_id, double* best_d2);
/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 6): A destroyed, V columns = vectors, w = values */
void jacobi_eig(int n, double* A, double* V, double* w);

#endif
