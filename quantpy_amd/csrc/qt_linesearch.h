// Scalar line-search state machine for the batched BFGS kernel.
//
// What it reproduces: the step-length logic that the reference's MLE reaches through
// scipy.optimize.minimize(method="BFGS")  (reference call site quantpy/tomography/state.py:213):
//   _line_search_wolfe12 = MINPACK-2 dcsrch/dcstep (More' & Thuente) with
//   ftol=c1=1e-4, gtol=c2=0.9, xtol=1e-14, stpmin=1e-100, stpmax=1e100, <=100 trial steps,
//   falling back to the Nocedal-Wright bracketing/zoom search (<=10 + <=11 trial steps).
// The reference's answer is path dependent, so the control flow
// is kept decision-for-decision; only the evaluation scheduling differs: every trial step is
// evaluated once for BOTH the value and the directional derivative (the kernel's gradient is
// analytic), and the searcher is written as an explicit state machine -- one transition per
// evaluation -- so that all trials of a wavefront share a single evaluation call site.
//
// Pure scalar code, compiled for the device by hipcc and for the host by g++ (the host build
// exists only for tests/test_linesearch_host.py, which drives it against SciPy's own
// scalar_search_wolfe1/2 on 1-D functions; it is not reachable from the product path).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define QT_HD __host__ __device__ __forceinline__
#else
#define QT_HD inline
#endif

namespace qt {

enum LsResult : int { LS_EVAL = 0, LS_ACCEPT = 1, LS_FAIL = 2 };

struct LineSearch {
  // ---- problem at alpha = 0
  double phi0, old_phi0, derphi0;
  int mode;  // 0 dcsrch, 1 bracketing phase of the fallback, 2 zoom phase of the fallback
  // ---- dcsrch
  int brackt, stage, evals;
  double gtest, width, width1, stx, fx, gx, sty, fy, gy, stmin, stmax;
  // ---- fallback search
  double alpha0, alpha1, phi_a0, phi_a1, derphi_a0;
  int it;
  double a_lo, a_hi, phi_lo, phi_hi, derphi_lo, phi_rec, a_rec;
  int zi;

  static constexpr double C1 = 1e-4, C2 = 0.9, AMAX = 1e100, AMIN = 1e-100, XTOL = 1e-14;

  // The whole state as SLOTS doubles (integers stored exactly as doubles): the workgroup-per-trial kernels park it
  // in LDS across an evaluation -- every thread holds the same values, thread 0 writes, everybody reads back --
  // instead of keeping ~70 registers per thread alive through the out-of-line NLL call.
  static constexpr int SLOTS = 32;
  QT_HD void save(double* p) const {
    p[0] = phi0; p[1] = old_phi0; p[2] = derphi0; p[3] = mode; p[4] = brackt; p[5] = stage; p[6] = evals;
    p[7] = gtest; p[8] = width; p[9] = width1; p[10] = stx; p[11] = fx; p[12] = gx; p[13] = sty; p[14] = fy;
    p[15] = gy; p[16] = stmin; p[17] = stmax; p[18] = alpha0; p[19] = alpha1; p[20] = phi_a0; p[21] = phi_a1;
    p[22] = derphi_a0; p[23] = it; p[24] = a_lo; p[25] = a_hi; p[26] = phi_lo; p[27] = phi_hi; p[28] = derphi_lo;
    p[29] = phi_rec; p[30] = a_rec; p[31] = zi;
  }
  QT_HD void load(const double* p) {
    phi0 = p[0]; old_phi0 = p[1]; derphi0 = p[2]; mode = (int)p[3]; brackt = (int)p[4]; stage = (int)p[5];
    evals = (int)p[6]; gtest = p[7]; width = p[8]; width1 = p[9]; stx = p[10]; fx = p[11]; gx = p[12]; sty = p[13];
    fy = p[14]; gy = p[15]; stmin = p[16]; stmax = p[17]; alpha0 = p[18]; alpha1 = p[19]; phi_a0 = p[20];
    phi_a1 = p[21]; derphi_a0 = p[22]; it = (int)p[23]; a_lo = p[24]; a_hi = p[25]; phi_lo = p[26]; phi_hi = p[27];
    derphi_lo = p[28]; phi_rec = p[29]; a_rec = p[30]; zi = (int)p[31];
  }

  // Python's min(a, b) / max(a, b): the first argument wins ties and NaN comparisons.
  QT_HD static double dmin(double a, double b) { return b < a ? b : a; }
  QT_HD static double dmax(double a, double b) { return b > a ? b : a; }
  QT_HD static double sgn(double a) { return (a > 0.0) - (a < 0.0); }

  // Initial step of both searches: min(1, 1.01 * 2 (phi0 - old_phi0) / derphi0), 1 if negative.
  QT_HD double first_step() const {
    double a = 1.0;
    if (derphi0 != 0.0) {
      a = dmin(1.0, 1.01 * 2.0 * (phi0 - old_phi0) / derphi0);
      if (a < 0.0) a = 1.0;
    }
    return a;
  }

  // Begin a search along a direction.  Returns LS_EVAL with *stp = first trial step.
  QT_HD int start(double phi0_, double old_phi0_, double derphi0_, double* stp) {
    phi0 = phi0_;
    old_phi0 = old_phi0_;
    derphi0 = derphi0_;
    double a1 = first_step();
    // dcsrch 'START' argument checks; any error sends the caller to the fallback search
    if (a1 < AMIN || a1 > AMAX || !(derphi0 < 0.0)) return start_fallback(stp);
    mode = 0;
    brackt = 0;
    stage = 1;
    evals = 0;
    gtest = C1 * derphi0;
    width = AMAX - AMIN;
    width1 = width / 0.5;
    stx = 0.0; fx = phi0; gx = derphi0;
    sty = 0.0; fy = phi0; gy = derphi0;
    stmin = 0.0;
    stmax = a1 + 4.0 * a1;
    *stp = a1;
    return LS_EVAL;
  }

  QT_HD int start_fallback(double* stp) {
    mode = 1;
    alpha0 = 0.0;
    double a1 = first_step();
    a1 = dmin(a1, AMAX);
    alpha1 = a1;
    phi_a0 = phi0;
    derphi_a0 = derphi0;
    it = 0;
    *stp = a1;
    return LS_EVAL;
  }

  // One safeguarded cubic/quadratic step of More'-Thuente (dcstep).
  QT_HD static void dcstep(double& stx_, double& fx_, double& dx_, double& sty_, double& fy_, double& dy_,
                           double& stp_, double fp, double dp, int& brackt_, double stpmin, double stpmax) {
    const double sgnd = sgn(dp) * sgn(dx_);
    double stpf;
    if (fp > fx_) {
      double theta = 3.0 * (fx_ - fp) / (stp_ - stx_) + dx_ + dp;
      double s = dmax(fabs(theta), dmax(fabs(dx_), fabs(dp)));
      double gamma = s * sqrt((theta / s) * (theta / s) - (dx_ / s) * (dp / s));
      if (stp_ < stx_) gamma = -gamma;
      double p = (gamma - dx_) + theta;
      double q = ((gamma - dx_) + gamma) + dp;
      double r = p / q;
      double stpc = stx_ + r * (stp_ - stx_);
      double stpq = stx_ + ((dx_ / ((fx_ - fp) / (stp_ - stx_) + dx_)) / 2.0) * (stp_ - stx_);
      stpf = (fabs(stpc - stx_) <= fabs(stpq - stx_)) ? stpc : stpc + (stpq - stpc) / 2.0;
      brackt_ = 1;
    } else if (sgnd < 0.0) {
      double theta = 3.0 * (fx_ - fp) / (stp_ - stx_) + dx_ + dp;
      double s = dmax(fabs(theta), dmax(fabs(dx_), fabs(dp)));
      double gamma = s * sqrt((theta / s) * (theta / s) - (dx_ / s) * (dp / s));
      if (stp_ > stx_) gamma = -gamma;
      double p = (gamma - dp) + theta;
      double q = ((gamma - dp) + gamma) + dx_;
      double r = p / q;
      double stpc = stp_ + r * (stx_ - stp_);
      double stpq = stp_ + (dp / (dp - dx_)) * (stx_ - stp_);
      stpf = (fabs(stpc - stp_) > fabs(stpq - stp_)) ? stpc : stpq;
      brackt_ = 1;
    } else if (fabs(dp) < fabs(dx_)) {
      double theta = 3.0 * (fx_ - fp) / (stp_ - stx_) + dx_ + dp;
      double s = dmax(fabs(theta), dmax(fabs(dx_), fabs(dp)));
      double rad = (theta / s) * (theta / s) - (dx_ / s) * (dp / s);
      double gamma = s * sqrt(rad > 0.0 ? rad : 0.0);  // max(0, .): a NaN radicand also gives 0
      if (stp_ > stx_) gamma = -gamma;
      double p = (gamma - dp) + theta;
      double q = (gamma + (dx_ - dp)) + gamma;
      double r = p / q;
      double stpc;
      if (r < 0.0 && gamma != 0.0) stpc = stp_ + r * (stx_ - stp_);
      else if (stp_ > stx_) stpc = stpmax;
      else stpc = stpmin;
      double stpq = stp_ + (dp / (dp - dx_)) * (stx_ - stp_);
      if (brackt_) {
        stpf = (fabs(stpc - stp_) < fabs(stpq - stp_)) ? stpc : stpq;
        if (stp_ > stx_) stpf = dmin(stp_ + 0.66 * (sty_ - stp_), stpf);
        else stpf = dmax(stp_ + 0.66 * (sty_ - stp_), stpf);
      } else {
        stpf = (fabs(stpc - stp_) > fabs(stpq - stp_)) ? stpc : stpq;
        stpf = dmin(dmax(stpf, stpmin), stpmax);
      }
    } else {
      if (brackt_) {
        double theta = 3.0 * (fp - fy_) / (sty_ - stp_) + dy_ + dp;
        double s = dmax(fabs(theta), dmax(fabs(dy_), fabs(dp)));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dy_ / s) * (dp / s));
        if (stp_ > sty_) gamma = -gamma;
        double p = (gamma - dp) + theta;
        double q = ((gamma - dp) + gamma) + dy_;
        double r = p / q;
        stpf = stp_ + r * (sty_ - stp_);
      } else if (stp_ > stx_) stpf = stpmax;
      else stpf = stpmin;
    }
    // interval update, written as value selects + unconditional assignments: as conditional stores through the
    // reference parameters the optimiser turned them into stores to a SELECTED address, i.e. scratch memory
    const bool hi = fp > fx_;
    const bool cross = !hi && sgnd < 0.0;
    const double n_sty = hi ? stp_ : (cross ? stx_ : sty_), n_fy = hi ? fp : (cross ? fx_ : fy_),
                 n_dy = hi ? dp : (cross ? dx_ : dy_);
    const double n_stx = hi ? stx_ : stp_, n_fx = hi ? fx_ : fp, n_dx = hi ? dx_ : dp;
    sty_ = n_sty; fy_ = n_fy; dy_ = n_dy;
    stx_ = n_stx; fx_ = n_fx; dx_ = n_dx;
    stp_ = stpf;
  }

  // Consume the evaluation (f, g = directional derivative) at trial step `stp`.
  // LS_EVAL: evaluate at *next.  LS_ACCEPT: `stp` is the step (value f).  LS_FAIL: no step.
  QT_HD int advance(double stp, double f, double g, double* next) {
    if (mode == 0) {
      int r = advance_dcsrch(stp, f, g, next);
      if (r == LS_FAIL) return start_fallback(next);  // line_search_wolfe2 with the same phi0/old_phi0
      return r;
    }
    return advance_fallback(stp, f, g, next);
  }

  QT_HD int advance_dcsrch(double stp, double f, double g, double* next) {
    ++evals;
    const double ftest = phi0 + stp * gtest;
    if (stage == 1 && f <= ftest && g >= 0.0) stage = 2;
    int warn = 0;
    if (brackt && (stp <= stmin || stp >= stmax)) warn = 1;
    if (brackt && stmax - stmin <= XTOL * stmax) warn = 1;
    if (stp == AMAX && f <= ftest && g <= gtest) warn = 1;
    if (stp == AMIN && (f > ftest || g >= gtest)) warn = 1;
    if (f <= ftest && fabs(g) <= C2 * -derphi0) return LS_ACCEPT;  // convergence overrides a warning
    if (warn) return LS_FAIL;

    // ONE call site of dcstep, on local copies: with two call sites (modified / plain function values) the
    // optimiser merged them into one body fed by SELECTED ADDRESSES, which pinned nine doubles of this struct to
    // scratch memory (72 bytes per lane in every BFGS kernel).  Same arithmetic, same order.
    const bool modified = stage == 1 && f <= fx && f > ftest;
    double sx = stx, sy = sty, fxe = fx, gxe = gx, fye = fy, gye = gy, fe = f, ge = g;
    int br = brackt;
    if (modified) {
      fe = f - stp * gtest;
      fxe = fx - stx * gtest;
      fye = fy - sty * gtest;
      ge = g - gtest;
      gxe = gx - gtest;
      gye = gy - gtest;
    }
    dcstep(sx, fxe, gxe, sy, fye, gye, stp, fe, ge, br, stmin, stmax);
    stx = sx;
    sty = sy;
    brackt = br;
    if (modified) {
      fx = fxe + stx * gtest;
      fy = fye + sty * gtest;
      gx = gxe + gtest;
      gy = gye + gtest;
    } else {
      fx = fxe;
      fy = fye;
      gx = gxe;
      gy = gye;
    }
    if (brackt) {
      if (fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
      width1 = width;
      width = fabs(sty - stx);
      stmin = dmin(stx, sty);
      stmax = dmax(stx, sty);
    } else {
      stmin = stp + 1.1 * (stp - stx);
      stmax = stp + 4.0 * (stp - stx);
    }
    stp = dmin(dmax(stp, AMIN), AMAX);  // NaN propagates like np.clip
    if ((brackt && (stp <= stmin || stp >= stmax)) || (brackt && stmax - stmin <= XTOL * stmax)) stp = stx;
    if (!isfinite(stp)) return LS_FAIL;
    // SciPy gives dcsrch 100 passes including START: the 100th evaluation is made and then
    // discarded unexamined, so the search has failed once 99 evaluations did not converge.
    if (evals >= 99) return LS_FAIL;
    *next = stp;
    return LS_EVAL;
  }

  // a + (-B + sqrt(B^2 - 3 A C)) / (3 A): minimiser of the cubic through (a,fa,fpa), (b,fb), (c,fc).
  // Returns NaN where SciPy's version returns None (it traps divide/overflow/invalid).
  QT_HD static double cubicmin(double a, double fa, double fpa, double b, double fb, double c, double fc) {
    const double C = fpa, db = b - a, dc = c - a;
    const double denom = (db * dc) * (db * dc) * (db - dc);
    if (denom == 0.0) return NAN;
    const double t0 = fb - fa - C * db, t1 = fc - fa - C * dc;
    double A = dc * dc * t0 - db * db * t1;
    double B = -(dc * dc * dc) * t0 + db * db * db * t1;
    A /= denom;
    B /= denom;
    const double radical = B * B - 3.0 * A * C;
    if (!(radical >= 0.0) || A == 0.0) return NAN;
    const double x = a + (-B + sqrt(radical)) / (3.0 * A);
    return isfinite(x) ? x : NAN;
  }

  QT_HD static double quadmin(double a, double fa, double fpa, double b, double fb) {
    const double db = b - a;
    if (db == 0.0) return NAN;
    const double B = (fb - fa - fpa * db) / (db * db);
    if (B == 0.0) return NAN;
    const double x = a - fpa / (2.0 * B);
    return isfinite(x) ? x : NAN;
  }

  QT_HD int zoom_begin(double alo, double ahi, double plo, double phi_h, double dlo, double* next) {
    mode = 2;
    a_lo = alo; a_hi = ahi; phi_lo = plo; phi_hi = phi_h; derphi_lo = dlo;
    phi_rec = phi0;
    a_rec = 0.0;
    zi = 0;
    return zoom_propose(next);
  }

  QT_HD int zoom_propose(double* next) {
    const double dalpha = a_hi - a_lo;
    double a = a_lo, b = a_hi;
    if (dalpha < 0.0) { a = a_hi; b = a_lo; }
    double aj = NAN;
    bool need_quad = true;
    if (zi > 0) {
      const double cchk = 0.2 * dalpha;
      aj = cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec, phi_rec);
      need_quad = isnan(aj) || aj > b - cchk || aj < a + cchk;
    }
    if (need_quad) {
      const double qchk = 0.1 * dalpha;
      aj = quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi);
      if (isnan(aj) || aj > b - qchk || aj < a + qchk) aj = a_lo + 0.5 * dalpha;
    }
    *next = aj;
    return LS_EVAL;
  }

  QT_HD int advance_fallback(double stp, double f, double g, double* next) {
    if (mode == 1) {
      phi_a1 = f;
      const double derphi_a1 = g;
      // (loop head of iteration `it`)
      if (alpha1 == 0.0 || alpha0 > AMAX) return LS_FAIL;
      if (phi_a1 > phi0 + C1 * alpha1 * derphi0 || (phi_a1 >= phi_a0 && it > 0))
        return zoom_begin(alpha0, alpha1, phi_a0, phi_a1, derphi_a0, next);
      if (fabs(derphi_a1) <= -C2 * derphi0) return LS_ACCEPT;
      if (derphi_a1 >= 0.0) return zoom_begin(alpha1, alpha0, phi_a1, phi_a0, derphi_a1, next);
      const double alpha2 = dmin(2.0 * alpha1, AMAX);
      alpha0 = alpha1;
      alpha1 = alpha2;
      phi_a0 = phi_a1;
      derphi_a0 = derphi_a1;
      ++it;
      // SciPy evaluates phi(alpha1) at the end of iteration 9 and then leaves the loop accepting it
      // (derivative unknown there; ours comes with the same evaluation) -> mode 3 marks that case.
      if (it >= 10) mode = 3;
      *next = alpha1;
      return LS_EVAL;
    }
    if (mode == 3) return LS_ACCEPT;
    // zoom: evaluation at a_j = stp
    const double a_j = stp, phi_aj = f, derphi_aj = g;
    if (phi_aj > phi0 + C1 * a_j * derphi0 || phi_aj >= phi_lo) {
      phi_rec = phi_hi; a_rec = a_hi;
      a_hi = a_j; phi_hi = phi_aj;
    } else {
      if (fabs(derphi_aj) <= -C2 * derphi0) return LS_ACCEPT;
      if (derphi_aj * (a_hi - a_lo) >= 0.0) {
        phi_rec = phi_hi; a_rec = a_hi;
        a_hi = a_lo; phi_hi = phi_lo;
      } else {
        phi_rec = phi_lo; a_rec = a_lo;
      }
      a_lo = a_j; phi_lo = phi_aj; derphi_lo = derphi_aj;
    }
    ++zi;
    if (zi > 10) return LS_FAIL;
    return zoom_propose(next);
  }
};

}  // namespace qt
