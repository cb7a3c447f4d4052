// Host build of quantpy_amd/csrc/qt_linesearch.h for tests/test_linesearch_host.py only.
#include "../../quantpy_amd/csrc/qt_linesearch.h"
extern "C" {
typedef void (*phi_cb)(double alpha, double* f, double* g);
// returns 1 accepted / 0 failed; *stp, *fval, *nevals out; mode_out = searcher mode at exit
int qt_host_line_search(phi_cb cb, double phi0, double old_phi0, double derphi0, double* stp_out,
                        double* f_out, int* nevals, int* mode_out) {
  qt::LineSearch ls;
  double stp;
  int r = ls.start(phi0, old_phi0, derphi0, &stp);
  int n = 0;
  double f = phi0, g = derphi0;
  while (r == qt::LS_EVAL && n < 400) {
    cb(stp, &f, &g);
    ++n;
    double next = stp;
    r = ls.advance(stp, f, g, &next);
    if (r == qt::LS_EVAL) stp = next;
  }
  *stp_out = stp;
  *f_out = f;
  *nevals = n;
  *mode_out = ls.mode;
  return r == qt::LS_ACCEPT ? 1 : 0;
}
}
