// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or shipped with
// the product (starry_process_amd/).  This translation unit gives the
// reference's own header-only C++ kernels a plain C ABI so that the oracle and
// the golden-vector generator can call the *real* reference code.  The
// reference sources are included BY PATH from /root/reference (see
// oracle/Makefile: -I/root/reference/starry_process/ops/include and the Eigen
// it vendors); nothing from the reference is copied into this repository.
//
// Each wrapper maps caller-owned buffers with Eigen::Map the same way the
// reference's Theano glue does (reference ops/wigner/Rx.cc:37-41,
// ops/wigner/tensordotRz.cc:34-52, ops/wigner/special_tensordotRz.cc:42-59,
// ops/flux/rTA1.cc:21-24, ops/flux/rTA1L.cc:39-58,
// ops/latitude/latitude.cc:47-80).
#include "utils.h"
#include "special.h"
#include "latitude.h"
#include "wigner.h"
#include "flux.h"

using namespace sp::utils;

extern "C" {

int spref_lmax() { return SP__LMAX; }
int spref_umax() { return SP__UMAX; }

void spref_Rx(double theta, double *R, double *dR) {
  Map<Vector<double, SP__NWIG>> Rm(R);
  Map<Vector<double, SP__NWIG>> dRm(dR);
  sp::wigner::computeRx(theta, Rm, dRm);
}

void spref_tensordotRz(const double *M, const double *theta, int K, double *f) {
  Map<RowMatrix<double, Dynamic, SP__N>> Mm(const_cast<double *>(M), K, SP__N);
  Map<Vector<double, Dynamic>> th(const_cast<double *>(theta), K);
  Map<RowMatrix<double, Dynamic, SP__N>> fm(f, K, SP__N);
  sp::wigner::computeTensordotRz(Mm, th, fm);
}

void spref_special_tensordotRz(const double *T, const double *M,
                               const double *theta, int K, double *f) {
  Map<RowMatrix<double, SP__N, SP__N>> Tm(const_cast<double *>(T));
  Map<RowMatrix<double, SP__N, SP__N>> Mm(const_cast<double *>(M));
  Map<Vector<double, Dynamic>> th(const_cast<double *>(theta), K);
  Map<Vector<double, Dynamic>> fm(f, K);
  sp::wigner::computeSpecialTensordotRz(Tm, Mm, th, fm);
}

void spref_rTA1(double *f) {
  Map<Vector<double, SP__N>> fm(f);
  sp::flux::computerTA1(fm);
}

static sp::flux::LimbDark<double> *LD = nullptr;

void spref_rTA1L(const double *u, double *f) {
#if SP__UMAX > 0
  if (LD == nullptr) LD = new sp::flux::LimbDark<double>();
  Map<Vector<double, SP__UMAX>> um(const_cast<double *>(u));
  Map<RowVector<double, SP__N>> fm(f);
  LD->computerTA1L(um, fm);
#else
  (void)u; (void)f;
#endif
}

// reverse-mode kernels (ops/wigner/tensordotRz_rev.cc, special_tensordotRz_rev.cc,
// ops/flux/rTA1L_rev.cc)
void spref_tensordotRz_rev(const double *M, const double *theta, int K, const double *bf,
                           double *bM, double *btheta) {
  Map<RowMatrix<double, Dynamic, SP__N>> Mm(const_cast<double *>(M), K, SP__N);
  Map<Vector<double, Dynamic>> th(const_cast<double *>(theta), K);
  Map<RowMatrix<double, Dynamic, SP__N>> bfm(const_cast<double *>(bf), K, SP__N);
  Map<RowMatrix<double, Dynamic, SP__N>> bMm(bM, K, SP__N);
  Map<Vector<double, Dynamic>> bth(btheta, K);
  sp::wigner::computeTensordotRzGradient(Mm, th, bfm, bMm, bth);
}

void spref_special_tensordotRz_rev(const double *T, const double *M, const double *theta, int K,
                                   const double *bf, double *bM, double *btheta) {
  Map<RowMatrix<double, SP__N, SP__N>> Tm(const_cast<double *>(T));
  Map<RowMatrix<double, SP__N, SP__N>> Mm(const_cast<double *>(M));
  Map<Vector<double, Dynamic>> th(const_cast<double *>(theta), K);
  Map<Vector<double, Dynamic>> bfm(const_cast<double *>(bf), K);
  Map<RowMatrix<double, SP__N, SP__N>> bMm(bM);
  Map<Vector<double, Dynamic>> bth(btheta, K);
  sp::wigner::computeSpecialTensordotRzGradient(Tm, Mm, th, bfm, bMm, bth);
}

void spref_rTA1L_rev(const double *u, const double *bf, double *bu) {
#if SP__UMAX > 0
  if (LD == nullptr) LD = new sp::flux::LimbDark<double>();
  Map<Vector<double, SP__UMAX>> um(const_cast<double *>(u));
  Map<RowVector<double, SP__N>> bfm(const_cast<double *>(bf));
  Map<Vector<double, SP__UMAX>> bum(bu);
  // the reference computes the Jacobian DDp in the forward pass (rTA1L.cc:39-58)
  RowVector<double, SP__N> f;
  LD->computerTA1L(um, f);
  Vector<double, SP__UMAX> uv = um;
  LD->computerTA1L(uv, bfm, bum);
#else
  (void)u; (void)bf; (void)bu;
#endif
}

void spref_latitude(double alpha_in, double beta_in, double *q, double *dqda,
                    double *dqdb, double *Q, double *dQda, double *dQdb) {
  double alpha = alpha_in > 0.0 ? alpha_in : 0.0;
  double beta = beta_in > 0.0 ? beta_in : 0.0;
  Map<Vector<double, SP__N>> qm(q), dqdam(dqda), dqdbm(dqdb);
  Map<RowMatrix<double, SP__N, SP__N>> Qm(Q), dQdam(dQda), dQdbm(dQdb);
  sp::latitude::computeLatitudeIntegrals(alpha, beta, qm, dqdam, dqdbm, Qm,
                                         dQdam, dQdbm);
}

}  // extern "C"
